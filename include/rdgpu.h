/* rdgpu.h -- C-ABI of librdgpu.so, the MI355X (gfx950) engine behind RichDEM's
 * FillDepressions / d8_flow_directions / barnes_flat_resolution_d8 / d8_flow_accum / FA_D8.
 *
 * Plain pointers and sizes only.  Rasters are dense row-major, i = y*width + x,
 * no padding (reference: include/richdem/common/Array2D.hpp:592-595).  All
 * functions return 0 on success and a non-zero code on failure; the message is
 * available from rdgpu_last_error().  The C++ shim include/rdgpu/richdem_gpu.hpp
 * turns non-zero into std::runtime_error, the reference's error convention.
 *
 * Two families:
 *   rdgpu_<op>_<dtype>(host pointers...)       drop-in boundary: H2D, compute, D2H
 *                                              into the SAME host buffer
 *   rdgpu_<op>_dev_<dtype>(device pointers...) HBM-resident variant (bench.py,
 *                                              multi-GPU shards, chaining stages)
 *
 * dtype suffixes: u8 i16 u16 i32 u32 f32 (32-bit-key engine).  f64/i64/u64 are
 * rejected with RDGPU_ERR_UNSUPPORTED in this round.
 *
 * Threading: one host thread at a time per process (the reference functions
 * are not internally re-entrant on shared arrays either).
 */
#ifndef RDGPU_H_
#define RDGPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDGPU_OK 0
#define RDGPU_ERR_HIP 1         /* a HIP runtime call failed (no GPU, OOM, ...) */
#define RDGPU_ERR_ARG 2         /* bad dimensions / null pointer / bad topology  */
#define RDGPU_ERR_UNSUPPORTED 3 /* dtype not supported by this build             */

/* ---- runtime ---------------------------------------------------------------------------- */
const char *rdgpu_last_error(void);
const char *rdgpu_version(void);
int rdgpu_device_count(int *count);
int rdgpu_set_device(int device_id);
/* Free the grow-only device workspace cached between calls. */
int rdgpu_release_workspace(void);

/* ---- FillDepressions<topology>(Array2D<T>&) -----------------------------------------------
 * Replaces richdem::FillDepressions (include/richdem/depressions/depressions.hpp:13-21), i.e.
 * PriorityFlood_Zhou2016 (depressions/Zhou2016.hpp:126-191) for topology 8 and
 * PriorityFlood_Barnes2014<D4> (depressions/Barnes2014.hpp:230-304) for topology 4.
 * In place; NoData is an ordinary elevation, exactly as in the reference. */
int rdgpu_fill_u8(uint8_t *dem, int width, int height, int topology);
int rdgpu_fill_i16(int16_t *dem, int width, int height, int topology);
int rdgpu_fill_u16(uint16_t *dem, int width, int height, int topology);
int rdgpu_fill_i32(int32_t *dem, int width, int height, int topology);
int rdgpu_fill_u32(uint32_t *dem, int width, int height, int topology);
int rdgpu_fill_f32(float *dem, int width, int height, int topology);

int rdgpu_fill_dev_u8(uint8_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i16(int16_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_u16(uint16_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i32(int32_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_u32(uint32_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_f32(float *d_dem, int width, int height, int topology, void *hip_stream);

/* Statistics of the last fill on this process (for DESIGN.md / bench.py reporting). */
typedef struct rdgpu_fill_stats {
  uint64_t cells;       /* width*height                                   */
  uint64_t basins;      /* descent-forest roots (pits) not draining out   */
  uint32_t rounds;      /* Boruvka contraction rounds over the raster     */
  uint32_t jump_passes; /* pointer-jumping passes over the descent forest */
} rdgpu_fill_stats;
int rdgpu_fill_get_stats(rdgpu_fill_stats *out);

/* ---- synthetic input (test/bench input generator) ----------- */
int rdgpu_synth_dem_dev_f32(float *d_dem, int width, int height, int seed, int x0, int y0,
                            float tilt, void *hip_stream);

/* ---- per-kernel timing (HIP events on the launch stream) -----------------------------------
 * rdgpu_profile_enable(1) makes every kernel launch be bracketed by hipEvents on the stream it
 * is launched on.  rdgpu_profile_collect() synchronises and folds the pending event pairs into
 * per-kernel totals.  rdgpu_profile_get() reads one kernel's totals; rdgpu_profile_name(i)
 * enumerates kernel names (NULL past the end).  rdgpu_profile_reset() clears totals. */
int rdgpu_profile_enable(int on);
int rdgpu_profile_collect(void);
int rdgpu_profile_reset(void);
const char *rdgpu_profile_name(int index);
int rdgpu_profile_get(const char *kernel, double *total_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* RDGPU_H_ */
