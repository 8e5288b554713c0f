/* rdgpu.h -- C-ABI of librdgpu.so, the MI355X (gfx950) engine behind RichDEM's
 * FillDepressions / pit_mask / d8_flow_directions / barnes_flat_resolution_d8 / ResolveFlatsEpsilon /
 * d8_flow_accum / FA_D8 / FA_Tarboton / FA_Quinn / FA_Holmgren / FA_Freeman / FlowAccumulation(props).
 *
 * NaN elevations are unsupported input throughout (the reference's heaps and compares give them no defined place either;
 * a float NoData value must be a number -- NaN never equals itself, so NaN "NoData" cells would count as data).
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
 * dtype suffixes: i8 u8 i16 u16 i32 u32 f32 everywhere; f64 i64 u64 additionally for the fill (exact: f64 through the f32
 * engine when the values fit, 64-bit values through dense value ranks otherwise), for d8_flow_directions, flat
 * resolution, ResolveFlatsEpsilon and FA_D8 / FM_D8 (these compare elevations in their own type); the D-infinity / MFD
 * families take i8 ... f64.  The row-block shard entry points of the fill take the 8 / 16 / 32-bit types.
 *
 * Threading: every entry point runs under the lock OF THE DEVICE that is current when it is entered (not a process-wide
 * lock): calls from several host threads on ONE device (e.g. Python threads with the GIL released by the wrapper) are
 * safe and run one after the other -- when the calling thread changes, that device is synchronised first, because its
 * calls share one grow-only workspace -- while threads on DIFFERENT devices run side by side.  The single-process
 * multi-device entry points (rdgpu_*_multi_*, and the plain host entries when RDGPU_DEVICES routes them there) run one
 * orchestrator at a time per process and must not be called from inside one another.
 * Streams: the `_dev_` entry points enqueue on the stream they are given and use that shared per-device workspace for
 * their scratch.  Issue all `_dev_` calls of a process on ONE stream per device, or synchronise between calls issued on
 * different streams -- two calls in flight on two streams would overwrite each other's scratch.  (The handle-based shard
 * entry points keep their state in buffers of their own.)
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
#define RDGPU_ERR_CAPACITY 4    /* rdgpu_fill_graph_solve_dev: a count in d_counts exceeds `cap` -- that shard's edges are
                                   not in d_edges_all; nothing was solved (the caller repeats its exchange with more room) */

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
int rdgpu_fill_i8(int8_t *dem, int width, int height, int topology);
/* 64-bit element types: exact as well -- f64 DEMs whose values fit f32 run the f32 engine, everything
 * else is filled on dense ranks of the values (csrc/fill64.hip). */
int rdgpu_fill_f64(double *dem, int width, int height, int topology);
int rdgpu_fill_i64(int64_t *dem, int width, int height, int topology);
int rdgpu_fill_u64(uint64_t *dem, int width, int height, int topology);

int rdgpu_fill_dev_u8(uint8_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i16(int16_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_u16(uint16_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i32(int32_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_u32(uint32_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_f32(float *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i8(int8_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_f64(double *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_i64(int64_t *d_dem, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_dev_u64(uint64_t *d_dem, int width, int height, int topology, void *hip_stream);

/* HasDepressions<topology>(const Array2D<T>&) (depressions/Barnes2014.hpp:44-103; apps/rd_depressions_has.cpp:14):
 * *out = 1 when the DEM holds a depression, i.e. when FillDepressions<topology> would raise at least one cell (the
 * reference runs the flood of PriorityFlood_Original without raising and stops at the first cell discovered from a higher
 * one, :91-95 -- the same predicate whatever its heap does among equal keys), 0 otherwise.  The DEM is not modified.
 * PriorityFlood_Original<topology> (:136-198) returns the surface of FillDepressions<topology>: rdgpu_fill_<T>.
 *
 * PriorityFlood_Wei2018(Array2D<T>&) (depressions/Wei2018.hpp:154-202): the D8 fill whose seeds are the raster's edge
 * cells AND every data cell next to a NoData cell (InitPriorityQue, :14-50): NoData cells are never altered and NoData
 * regions inside the raster are outlets.  Equal to rdgpu_fill_<T> on rasters without NoData (tests/tests.cpp:259-262). */
#define RDGPU_DECL_VARIANTS(SUF, T)                                                                                  \
  int rdgpu_has_depressions_##SUF(const T *dem, int width, int height, int topology, int *out);                      \
  int rdgpu_has_depressions_dev_##SUF(const T *d_dem, int width, int height, int topology, int *out, void *hip_stream); \
  int rdgpu_fill_wei2018_##SUF(T *dem, T nodata, int width, int height);                                             \
  int rdgpu_fill_wei2018_dev_##SUF(T *d_dem, T nodata, int width, int height, void *hip_stream);
RDGPU_DECL_VARIANTS(u8, uint8_t)
RDGPU_DECL_VARIANTS(i8, int8_t)
RDGPU_DECL_VARIANTS(i16, int16_t)
RDGPU_DECL_VARIANTS(u16, uint16_t)
RDGPU_DECL_VARIANTS(i32, int32_t)
RDGPU_DECL_VARIANTS(u32, uint32_t)
RDGPU_DECL_VARIANTS(f32, float)
RDGPU_DECL_VARIANTS(f64, double)
RDGPU_DECL_VARIANTS(i64, int64_t)
RDGPU_DECL_VARIANTS(u64, uint64_t)
#undef RDGPU_DECL_VARIANTS

/* PriorityFlood_Barnes2014_max_dep<topology>(Array2D<T>&, uint64_t max_dep_size) (depressions/Barnes2014.hpp:844-931;
 * apps/rd_depressions_flood.cpp:16-19 with a non-zero third argument): only depressions of at most max_dep_size
 * cells are filled, the others are left as they are.  "Depression" as the reference counts it: the cells below the
 * level L that one cell of elevation L floods in one run of its pit queue.  Identical to the reference on DEMs without
 * equal elevations and on the reference's goldens (tests/depressions/testdem1.{1,2}.out); when several cells of
 * elevation L touch the same pocket the reference's grouping follows std::priority_queue's pop order, this one the
 * lowest cell index.  Where that can happen is DETECTED on the device: pockets sharing any possible flooding cell form a
 * cluster, a cluster holding a pocket with two or more possible flooding cells is tie-flagged, and inside it the order
 * matters only for pockets within the size limit of clusters beyond it; on all other cells
 * the output does not depend on the pop order (rdgpu_fill_max_dep_get_stats; rdgpu_fill_max_dep_ties_dev_<T> also writes
 * the flagged cells as a uint8 mask -- the parity tests assert that every cell differing from the reference lies in it). */
typedef struct rdgpu_max_dep_stats {
  uint64_t pockets;           /* connected components of the cells the plain fill would raise */
  uint64_t tie_pockets;       /* ... that two or more cells of their spill elevation can flood: the heap's order decides */
  uint64_t tie_cluster_cells; /* cells whose fate the heap's order can decide: pockets no larger than the limit, in tie-flagged
                                 clusters larger than the limit (0: the result is the reference's whatever its heap does) */
  uint64_t pocket_cells;      /* cells the plain fill would raise */
} rdgpu_max_dep_stats;
int rdgpu_fill_max_dep_get_stats(rdgpu_max_dep_stats *out);   /* of the calling thread's last max_dep fill; waits for that
                                                                  call's stream (the `_dev_` entries themselves do not block) */
#define RDGPU_DECL_MAXDEP(SUF, T)                                                                       \
  int rdgpu_fill_max_dep_##SUF(T *dem, int width, int height, int topology, uint64_t max_dep_size);     \
  int rdgpu_fill_max_dep_dev_##SUF(T *d_dem, int width, int height, int topology, uint64_t max_dep_size, void *hip_stream); \
  int rdgpu_fill_max_dep_ties_dev_##SUF(T *d_dem, int width, int height, int topology, uint64_t max_dep_size,              \
                                        uint8_t *d_tie_mask, void *hip_stream);
RDGPU_DECL_MAXDEP(u8, uint8_t)
RDGPU_DECL_MAXDEP(i16, int16_t)
RDGPU_DECL_MAXDEP(u16, uint16_t)
RDGPU_DECL_MAXDEP(i32, int32_t)
RDGPU_DECL_MAXDEP(u32, uint32_t)
RDGPU_DECL_MAXDEP(f32, float)
RDGPU_DECL_MAXDEP(i8, int8_t)
RDGPU_DECL_MAXDEP(f64, double)    /* the 64-bit element types run on dense value ranks (csrc/fill64.hip), as the fill does */
RDGPU_DECL_MAXDEP(i64, int64_t)
RDGPU_DECL_MAXDEP(u64, uint64_t)
#undef RDGPU_DECL_MAXDEP

/* pit_mask<topology>(const Array2D<T>&, Array2D<uint8_t>&) (depressions/Barnes2014.hpp:593-676,
 * apps/rd_depressions_mask.cpp:16): 1 = the cell lies in a depression (the fill would raise it), 0 = not,
 * 3 = NoData.  The DEM is not modified. */
#define RDGPU_DECL_PITMASK(SUF, T)                                                                        \
  int rdgpu_pit_mask_##SUF(const T *dem, T nodata, int width, int height, int topology, uint8_t *mask);   \
  int rdgpu_pit_mask_dev_##SUF(const T *d_dem, T nodata, int width, int height, int topology, uint8_t *d_mask, void *hip_stream);
RDGPU_DECL_PITMASK(u8, uint8_t)
RDGPU_DECL_PITMASK(i16, int16_t)
RDGPU_DECL_PITMASK(u16, uint16_t)
RDGPU_DECL_PITMASK(i32, int32_t)
RDGPU_DECL_PITMASK(u32, uint32_t)
RDGPU_DECL_PITMASK(f32, float)
RDGPU_DECL_PITMASK(i8, int8_t)
RDGPU_DECL_PITMASK(f64, double)
RDGPU_DECL_PITMASK(i64, int64_t)
RDGPU_DECL_PITMASK(u64, uint64_t)
#undef RDGPU_DECL_PITMASK

/* ---- PriorityFloodEpsilon_Barnes2014<topology>(Array2D<T>&) ------------------------------------------------------
 * Replaces richdem::PriorityFloodEpsilon_Barnes2014 (include/richdem/depressions/Barnes2014.hpp:335-420) =
 * FillDepressionsEpsilon (depressions/depressions.hpp:23); Python rd.FillDepressions(epsilon=True) ->
 * rdPFepsilonD8 / rdPFepsilonD4 (wrappers/pyrichdem/src/pywrapper.hpp:34-35).  In place; every depression and flat
 * gets a gradient of one representable step (std::nextafter) per cell towards its outlet.  Floating-point DEMs only:
 * the reference throws "Priority-Flood+Epsilon is only available for floating-point data types!" for the integer
 * types (:424-451) and so does the C++ shim.  NoData cells are never altered and act as outlets of value NoData
 * (what the reference does with NoData regions that touch the raster border, given its precondition that NoData is
 * lower than every data value).
 * The result is the unique surface E = z on the border / NoData, E(c) = max(z(c), nextafter(min over neighbours
 * E(n))) elsewhere; it equals the reference's output on every DEM in which no two cells of the reference's heap tie
 * (with ties the reference's output depends on std::priority_queue's pop order; this surface is then <= it). */
int rdgpu_fill_epsilon_f32(float *dem, float nodata, int width, int height, int topology);
int rdgpu_fill_epsilon_f64(double *dem, double nodata, int width, int height, int topology);
int rdgpu_fill_epsilon_dev_f32(float *d_dem, float nodata, int width, int height, int topology, void *hip_stream);
int rdgpu_fill_epsilon_dev_f64(double *d_dem, double nodata, int width, int height, int topology, void *hip_stream);
typedef struct rdgpu_epsilon_stats {
  uint32_t rounds;           /* tile-relaxation rounds, all attempts                                            */
  uint32_t attempts;         /* 1 unless the assumed slack was too small for the DEM (RDGPU_EPS_SLACK=<steps>)  */
  uint64_t tile_relaxations; /* tiles relaxed, summed over the rounds                                           */
  uint64_t slack;            /* the slack the successful attempt ran with, in representable steps               */
  uint64_t max_lift;         /* largest lift of a cell above the plain fill, in representable steps             */
  uint64_t tie_sources;      /* gradient sources (cells at their own elevation next to a raised cell) whose elevation
                                another source shares: 0 = identical to the reference; otherwise the reference's result
                                follows std::priority_queue's pop order there and this surface is a cell-wise lower
                                bound of it (RDGPU_EPS_TIES=0 skips the count)                                      */
} rdgpu_epsilon_stats;
int rdgpu_fill_epsilon_get_stats(rdgpu_epsilon_stats *out);

/* ---- PriorityFloodWatersheds_Barnes2014<topology>(Array2D<T>&, Array2D<int32_t>& labels, bool alter_elevations) ----
 * Replaces richdem::PriorityFloodWatersheds_Barnes2014 (include/richdem/depressions/Barnes2014.hpp:713-807): labels[i] =
 * the watershed the cell drains to (every data cell of the raster border, and every data cell next to a NoData region
 * connected to it, starts a watershed; labels are numbered from 1 in the order the reference pops their first cell,
 * i.e. by elevation; cells that are never labelled -- NoData connected to the border -- hold -1 = labels.noData()).
 * alter != 0: the DEM is filled as by PriorityFlood_Barnes2014 (:793-794).  Identical to the reference, numbering
 * included, on DEMs without equal elevations among the cells of its heap; with ties the reference's partition follows
 * std::priority_queue's pop order. */
/* PriorityFloodFlowdirs_Barnes2014(elevations, flowdirs) -- depressions/Barnes2014.hpp:483-555: D8 directions of the flood
 * that does not raise the DEM; every cell points at the neighbour that was flooded first (border cells off the raster, NoData
 * cells 0).  Identical to the reference, equal elevations included: the reference's queue breaks ties by insertion order
 * (GridCellZk_low_pq, common/grid_cell.hpp:101-122) and that order is reproduced as a fixed point -- an exact
 * flood of the raster's unique ranks, then passes over (tree of directions, ranks) until both reproduce themselves, a second
 * flood once the ranks rest (3 - 4 passes on float terrain, the breadth-first depth of the largest plateau on integer DEMs;
 * RDGPU_PFD_TREE_ITER=0: a flood in every pass).  The passes are BOUNDED by a count, RDGPU_PFD_TIE_PASSES (default 1000:
 * deterministic -- the same DEM gives the same raster on every machine); a wall-time bound is opt-in
 * (RDGPU_PFD_TIE_SECONDS=<seconds>, checked between passes: a result stopped by the clock is NOT reproducible across
 * machines or host loads).  When a bound stops the passes the call still returns RDGPU_OK, the result is an exact flood of a
 * stable order and stats.unresolved != 0 says how many ranks were still moving (the C++ shim logs one line to stderr, the
 * Python layer raises a RuntimeWarning): callers that need the reference's tie order must read the stats.  RDGPU_PFD_RANKS=0 is the FAST path for callers who do not need the
 * reference's tie order: one flood, ties decided by neighbour number (seconds instead of tens of seconds at 40000^2);
 * stats.unresolved then counts the cells where a tie decided.  One fill per nesting level of the depressions per flood. */
typedef struct rdgpu_pf_flowdirs_stats {
  uint32_t levels;      /* fills run */
  uint32_t twins;       /* cells whose elevation occurs more than once in the raster: 0 => the result is the reference's */
  uint64_t unresolved;  /* twins != 0: cells whose place in the tie order was still moving when the passes ran out (0: the
                           result is the reference's; = twins when no re-rank pass ran at all, RDGPU_PFD_TIE_PASSES=0);
                           RDGPU_PFD_RANKS=0: directions decided by neighbour number */
  uint32_t tie_passes;  /* passes of the tie order's fixed point (rank computations; not every pass floods) */
  uint32_t reserved;
} rdgpu_pf_flowdirs_stats;
int rdgpu_pf_flowdirs_get_stats(rdgpu_pf_flowdirs_stats *out);
#define RDGPU_DECL_PFD(SUF, T)                                                                              \
  int rdgpu_pf_flowdirs_##SUF(const T *dem, T nodata, int width, int height, uint8_t *dirs);                \
  int rdgpu_pf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int width, int height, uint8_t *d_dirs, void *hip_stream); \
  /* building block: the D8 fill with interior outlets (cells flagged in d_outlet drain like border cells) */ \
  int rdgpu_fill_outlets_dev_##SUF(T *d_dem, const uint8_t *d_outlet, int width, int height, void *hip_stream); \
  /* ... d_skip (per 64 x 64 tile, optional): 1 / 2 = nothing but outlets in the tile and around it (1: first time) */ \
  int rdgpu_fill_outlets_skip_dev_##SUF(T *d_dem, const uint8_t *d_outlet, const uint8_t *d_skip, int width, int height, \
                                        void *hip_stream);                                                    \
  /* ... d_lists: [tiles in state 0 or 1 | their 64 x 32 pair-pass tiles (state 0) | tiles in state 0], `stride` entries each, \
     counts3 (host): their lengths -- the raster kernels are launched over the lists only */                  \
  int rdgpu_fill_outlets_lists_dev_##SUF(T *d_dem, const uint8_t *d_outlet, const uint8_t *d_skip, const uint32_t *d_lists, \
                                         uint32_t stride, const uint32_t *counts3, int width, int height, void *hip_stream);
RDGPU_DECL_PFD(u8, uint8_t)
RDGPU_DECL_PFD(i8, int8_t)
RDGPU_DECL_PFD(i16, int16_t)
RDGPU_DECL_PFD(u16, uint16_t)
RDGPU_DECL_PFD(i32, int32_t)
RDGPU_DECL_PFD(u32, uint32_t)
RDGPU_DECL_PFD(f32, float)
#undef RDGPU_DECL_PFD
/* f64 / i64 / u64: on the dense value ranks of the DEM (fill64.hip), no restriction on the values */
#define RDGPU_DECL_PFD64(SUF, T)                                                                            \
  int rdgpu_pf_flowdirs_##SUF(const T *dem, T nodata, int width, int height, uint8_t *dirs);                \
  int rdgpu_pf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
RDGPU_DECL_PFD64(f64, double)
RDGPU_DECL_PFD64(i64, int64_t)
RDGPU_DECL_PFD64(u64, uint64_t)
#undef RDGPU_DECL_PFD64

#define RDGPU_DECL_WS(SUF, T)                                                                                          \
  int rdgpu_watersheds_##SUF(T *dem, T nodata, int width, int height, int topology, int alter, int32_t *labels);      \
  int rdgpu_watersheds_dev_##SUF(T *d_dem, T nodata, int width, int height, int topology, int alter, int32_t *d_labels, void *hip_stream);
RDGPU_DECL_WS(u8, uint8_t)
RDGPU_DECL_WS(i16, int16_t)
RDGPU_DECL_WS(u16, uint16_t)
RDGPU_DECL_WS(i32, int32_t)
RDGPU_DECL_WS(u32, uint32_t)
RDGPU_DECL_WS(f32, float)
RDGPU_DECL_WS(i8, int8_t)
RDGPU_DECL_WS(f64, double)
RDGPU_DECL_WS(i64, int64_t)
RDGPU_DECL_WS(u64, uint64_t)
#undef RDGPU_DECL_WS

/* Environment switches of the fill (read at every call; for tests and A/B timing, results never change):
 *   RDGPU_FILL_EDGES=0         every contraction round is a raster pass (the r01d engine; default: one raster pass,
 *                              then rounds on the component-pair list it records)
 *   RDGPU_FILL_EDGE_CAP=<n>    capacity of that list in records (default min(12 per basin, cells/2)); a list
 *                              that does not fit falls back to raster passes
 *   RDGPU_FILL_DEDUP=0         do not merge the list's records per component pair between rounds
 *   RDGPU_FILL_ROUND_BATCH=<n> contraction rounds enqueued per stream synchronisation (default: all of them at once --
 *                              a fill synchronises twice, rdgpu_fill_stats::host_syncs) */
/* Statistics of the last fill on this process (for DESIGN.md / bench.py reporting). */
typedef struct rdgpu_fill_stats {
  uint64_t cells;       /* width*height                                   */
  uint64_t basins;      /* descent-forest roots (pits) not draining out   */
  uint32_t rounds;      /* Boruvka contraction rounds                     */
  uint32_t jump_passes; /* pointer-jumping passes over the descent forest */
  uint64_t scan_tiles;  /* tiles visited by fill.scan, summed over the rounds */
  uint32_t tile_cells;  /* cells per scan tile                               */
  uint32_t edge_records; /* component-pair records the first raster pass handed to rounds 2.. (0: raster rounds) */
  uint32_t host_syncs;   /* stream synchronisations inside the fill (r06: 2 -- after the descent, after the rounds) */
  uint32_t reserved;
} rdgpu_fill_stats;
int rdgpu_fill_get_stats(rdgpu_fill_stats *out);

/* ---- row-block shards: the tile protocol of programs/parallel_priority_flood over GPUs ---------
 * Mirrors the reference's tiled Priority-Flood (Barnes 2016; programs/parallel_priority_flood/main.cpp:
 * Consumer FirstRound :276-313 / SecondRound :315-330, Producer Calculations :401-547; Zhou2016pf.hpp).
 * The DEM is cut into row blocks; one block per GPU (or one after the other on one GPU).
 *   1. rdgpu_fill_shard_begin_<T>   local phase on the block resident in HBM: fills it against its own
 *                                   perimeter, labels every cell with the cut-row cell ("terminal") or
 *                                   the outside its watershed drains to, and reduces the lowest pass
 *                                   between every pair of adjacent watersheds (the spillover graph).
 *                                   open_top / open_bottom: the first / last row is a cut, not DEM border.
 *   2. rdgpu_fill_shard_export      what the reference's Job1 carries (main.cpp:147-173): the cut rows'
 *                                   elevations (as order-preserving uint32 keys) and the graph edges
 *                                   (a, b, pass) with a/b = terminal ids (top row: x, bottom row:
 *                                   width + x) or 0xFFFFFFFF for the outside.  Host buffers.
 *   3. exchange                     every rank all-gathers (2*width keys + edges) -- RCCL over xGMI in
 *                                   richdem_amd/sharded.py; nothing else crosses GPUs.
 *   4. rdgpu_fill_graph_solve       host: joins the shard graphs along the cuts and floods the label
 *                                   graph from the outside (= the producer's aggregated Priority-Flood).
 *                                   keys/levels: [nshards][2][width]; edges: concatenated triples,
 *                                   edge_offsets[nshards+1] in triples.  Deterministic, so every rank
 *                                   solves redundantly instead of broadcasting (Job2, main.cpp:549-554).
 *   5. rdgpu_fill_shard_finish      raises the block (levels = this shard's [2][width] slice), writes
 *                                   the filled elevations in place and releases the handle.
 * rdgpu_fill_sharded_<T> runs 1-5 shard after shard on one GPU (tiling-invariance tests; DEMs cut
 * this way give bit-identical results to rdgpu_fill_<T>). */
typedef struct rdgpu_fill_shard rdgpu_fill_shard;
int rdgpu_fill_shard_begin_u8(uint8_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_i16(int16_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_u16(uint16_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_i32(int32_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_u32(uint32_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_f32(float *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_begin_i8(int8_t *d_rows, int width, int rows, int topology, int open_top, int open_bottom, void *hip_stream, rdgpu_fill_shard **out);
int rdgpu_fill_shard_edge_count(rdgpu_fill_shard *shard, uint32_t *n_edges);
int rdgpu_fill_shard_export(rdgpu_fill_shard *shard, uint32_t *top_keys, uint32_t *bottom_keys, uint32_t *edges);
int rdgpu_fill_shard_finish(rdgpu_fill_shard *shard, const uint32_t *levels);
int rdgpu_fill_shard_free(rdgpu_fill_shard *shard);
int rdgpu_fill_graph_solve(int nshards, int width, int topology, const uint32_t *keys, const uint32_t *edges,
                           const uint64_t *edge_offsets, uint32_t *levels);
/* Device-resident variants of steps 2, 4 and 5 (nothing but the all-gather leaves HBM):
 *   export_dev : d_keys[2*width] <- cut-row keys, d_edges[3*cap] <- edge triples (count from edge_count)
 *   graph_solve_dev : d_keys_all [nshards][2][width], d_edges_all [nshards][cap][3] with d_counts[nshards]
 *                     valid triples per shard -> d_levels_all [nshards][2][width]; the label graph is
 *                     contracted on the GPU with the fill's own Boruvka kernels
 *   finish_dev : levels = this shard's [2][width] slice, on the device */
int rdgpu_fill_shard_export_dev(rdgpu_fill_shard *shard, uint32_t *d_keys, uint32_t *d_edges, uint32_t cap);
int rdgpu_fill_graph_solve_dev(int nshards, int width, int topology, const uint32_t *d_keys_all,
                               const uint32_t *d_edges_all, const uint32_t *d_counts, uint32_t cap,
                               uint32_t *d_levels_all, void *hip_stream);
int rdgpu_fill_shard_finish_dev(rdgpu_fill_shard *shard, const uint32_t *d_levels);
/* One process, several devices (the reference's tiled driver, programs/parallel_priority_flood/main.cpp:276-330,
 * :401-547, as a library call): row block s of the host raster goes to devices[s] over that device's own PCIe link,
 * is filled locally there (one host thread per device: the devices work side by side), the cut rows and spillover
 * graphs are joined and solved on devices[0] (RDGPU_MULTI_HOST_SOLVE=1: on the host), every device raises its block and
 * returns it.  Same result as rdgpu_fill_<T>, bit for bit.  A device id may be listed more than once (its blocks are
 * then handled in order).  EXPERIMENTAL in one respect: this build's test boxes have one GPU, so the entry is verified
 * with one physical device listed several times -- threads, staging, the joined solve -- not on several devices.
 * rdgpu_fill_<T> itself takes this path when the environment holds RDGPU_DEVICES=<id>,<id>,... with two or more ids,
 * so rdgpu::FillDepressions(Array2D&) and apps/rd_depressions_flood use several GPUs without a change of signature. */
int rdgpu_fill_multi_u8(uint8_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_i16(int16_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_u16(uint16_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_i32(int32_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_u32(uint32_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_f32(float *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_multi_i8(int8_t *dem, int width, int height, int topology, const int *devices, int ndevices);
int rdgpu_fill_sharded_u8(uint8_t *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_i16(int16_t *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_u16(uint16_t *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_i32(int32_t *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_u32(uint32_t *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_f32(float *dem, int width, int height, int topology, int nshards);
int rdgpu_fill_sharded_i8(int8_t *dem, int width, int height, int topology, int nshards);

/* ---- d8_flow_directions(const Array2D<T>&, Array2D<uint8_t>&) ------------------------------
 * Replaces richdem::d8_flow_directions / d8_FlowDir (include/richdem/flowmet/d8_flowdirs.hpp:96-123,
 * :32-74).  dirs[i] in {0 = NO_FLOW, 1..8 = neighbour in the 234/105/876 numbering, 255 =
 * FLOWDIR_NO_DATA} (common/constants.hpp:76-80).  The shim sets flowdirs.setNoData(255). */
int rdgpu_d8_flowdirs_u8(const uint8_t *dem, uint8_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_i16(const int16_t *dem, int16_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_u16(const uint16_t *dem, uint16_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_i32(const int32_t *dem, int32_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_u32(const uint32_t *dem, uint32_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_f32(const float *dem, float nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_f64(const double *dem, double nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_i8(const int8_t *dem, int8_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_i64(const int64_t *dem, int64_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_u64(const uint64_t *dem, uint64_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_d8_flowdirs_dev_u8(const uint8_t *d_dem, uint8_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_i16(const int16_t *d_dem, int16_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_u16(const uint16_t *d_dem, uint16_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_i32(const int32_t *d_dem, int32_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_u32(const uint32_t *d_dem, uint32_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_f32(const float *d_dem, float nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_f64(const double *d_dem, double nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_i8(const int8_t *d_dem, int8_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_i64(const int64_t *d_dem, int64_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_d8_flowdirs_dev_u64(const uint64_t *d_dem, uint64_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);

/* ---- barnes_flat_resolution_d8(Array2D<T>& elevations, Array2D<uint8_t>& flowdirs, alter=false) --
 * Replaces richdem::barnes_flat_resolution_d8 (include/richdem/flats/flat_resolution.hpp:587-605) with
 * alter == false: d8_flow_directions, then resolve_flats_barnes (:447-517), then d8_flow_flats
 * (:96-116).  dirs is written in full; NO_FLOW cells of flats without an outlet stay 0.
 * rdgpu_resolve_flats_* additionally returns the flat_mask of resolve_flats_barnes (identical to the
 * reference's values) and the flat partition (labels[i] = 1 + lowest cell index of the cell's flat,
 * 0 for cells outside drainable flats; the reference numbers flats in scan order instead -- only
 * the partition is defined by the algorithm).  mask / labels may be NULL. */
int rdgpu_flat_resolution_d8_u8(const uint8_t *dem, uint8_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_i16(const int16_t *dem, int16_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_u16(const uint16_t *dem, uint16_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_i32(const int32_t *dem, int32_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_u32(const uint32_t *dem, uint32_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_f32(const float *dem, float nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_f64(const double *dem, double nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_i8(const int8_t *dem, int8_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_i64(const int64_t *dem, int64_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_u64(const uint64_t *dem, uint64_t nodata, int width, int height, uint8_t *dirs);
int rdgpu_flat_resolution_d8_dev_u8(const uint8_t *d_dem, uint8_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_i16(const int16_t *d_dem, int16_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_u16(const uint16_t *d_dem, uint16_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_i32(const int32_t *d_dem, int32_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_u32(const uint32_t *d_dem, uint32_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_f32(const float *d_dem, float nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_f64(const double *d_dem, double nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_i8(const int8_t *d_dem, int8_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_i64(const int64_t *d_dem, int64_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_flat_resolution_d8_dev_u64(const uint64_t *d_dem, uint64_t nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
int rdgpu_resolve_flats_u8(const uint8_t *dem, uint8_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_i16(const int16_t *dem, int16_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_u16(const uint16_t *dem, uint16_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_i32(const int32_t *dem, int32_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_u32(const uint32_t *dem, uint32_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_f32(const float *dem, float nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_f64(const double *dem, double nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_i8(const int8_t *dem, int8_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_i64(const int64_t *dem, int64_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);
int rdgpu_resolve_flats_u64(const uint64_t *dem, uint64_t nodata, int width, int height, uint8_t *dirs, int32_t *mask, int32_t *labels);

/* alter == true (flat_resolution.hpp:597-600 + d8_flats_alter_dem :545-582): the DEM is raised in place by
 * flat_mask increments of nextafterf inside drainable flats, then plain D8 directions are taken on it (the reference
 * applies nextafterf, i.e. float steps, to every element type). */
#define RDGPU_DECL_ALTER(SUF, T)                                                                                       \
  int rdgpu_flat_resolution_d8_alter_##SUF(T *dem, T nodata, int width, int height, uint8_t *dirs);                     \
  int rdgpu_flat_resolution_d8_alter_dev_##SUF(T *d_dem, T nodata, int width, int height, uint8_t *d_dirs, void *hip_stream);
RDGPU_DECL_ALTER(f32, float)
RDGPU_DECL_ALTER(f64, double)
/* integer element types: the reference's step is (T)nextafterf((float)e, numeric_limits<T>::infinity() == 0) -- one
 * towards zero per increment (one float spacing for magnitudes of 2^24 and more) -- reproduced as is */
RDGPU_DECL_ALTER(u8, uint8_t)
RDGPU_DECL_ALTER(i8, int8_t)
RDGPU_DECL_ALTER(i16, int16_t)
RDGPU_DECL_ALTER(u16, uint16_t)
RDGPU_DECL_ALTER(i32, int32_t)
RDGPU_DECL_ALTER(u32, uint32_t)
RDGPU_DECL_ALTER(i64, int64_t)
RDGPU_DECL_ALTER(u64, uint64_t)
#undef RDGPU_DECL_ALTER

/* Environment switches of the directions-only flat resolution (read at every call; A/B timing and tests, results never change):
 *   RDGPU_FLAT_PLANES=0         the two level fields as one int per cell instead of 16 bit planes per 64 x 64 tile (the plane
 *                               engine also steps aside by itself when a level does not fit 16 bits: an open flat more than
 *                               65 000 cells across; flat_mask / labels, alter = true, ResolveFlatsEpsilon and the row-block
 *                               shards always use ints)
 *   RDGPU_FLAT_STATIC=0         the towards search in batches of rounds decided on the host instead of one enqueue
 *   RDGPU_FLAT_CLASS_BITMAPS=1  the classification writes the searches' bitmaps itself (no flag bytes)
 *   RDGPU_FLAT_ASYNC=<n>        a search's rounds hand over to resident wavefronts once a round visits fewer than n tiles
 *                               (default 20000; 0: rounds to the end), RDGPU_FLAT_ASYNC_BLOCKS / _NAP / _STATS tune and report them
 *   RDGPU_FLAT_AWAY_BESIDE=0    the away search after the towards search instead of beside its tail */
typedef struct rdgpu_flat_stats {
  uint64_t low_edges;      /* find_flat_edges: cells with flow next to an equal NO_FLOW cell */
  uint64_t high_edges;     /* NO_FLOW cells next to higher terrain                           */
  uint64_t noflow_cells;   /* cells without a local gradient                                 */
  uint32_t away_levels;    /* tile-relaxation rounds, away-from-higher gradient              */
  uint32_t towards_levels; /* tile-relaxation rounds, towards-lower gradient                 */
} rdgpu_flat_stats;
int rdgpu_flat_get_stats(rdgpu_flat_stats *out);
/* The asynchronous tails of the two level searches of the last flat resolution on this thread (csrc/flats.hip,
 * k_relax_bits_async; no reference counterpart -- the reference's searches are serial queues, flats/flat_resolution.hpp:152-298):
 * tile visits by the resident wavefronts, tiles that were live when the rounds handed over (summed over the launches),
 * launches, launches that gave up and were finished in rounds. */
typedef struct rdgpu_flat_async_stats {
  uint64_t visits;
  uint32_t launches;
  uint32_t failures;
  uint32_t live_tiles;
  uint32_t reserved;
} rdgpu_flat_async_stats;
int rdgpu_flat_get_async_stats(rdgpu_flat_async_stats *out);

/* ResolveFlatsEpsilon(Array2D<T>&) (flats/flats.hpp:21-28; rd.ResolveFlats): the DEM is altered in place so
 * that every flat with an outlet drains -- each interior cell of such a flat is raised by flat_mask increments
 * of std::nextafter(e, numeric_limits<T>::infinity()) in its own type (float / double: towards +inf; integer
 * types: numeric_limits<int>::infinity() is 0, so the reference moves the value towards zero -- kept). */
#define RDGPU_DECL_RFE(SUF, T)                                                    \
  int rdgpu_resolve_flats_epsilon_##SUF(T *dem, T nodata, int width, int height); \
  int rdgpu_resolve_flats_epsilon_dev_##SUF(T *d_dem, T nodata, int width, int height, void *hip_stream);
RDGPU_DECL_RFE(u8, uint8_t)
RDGPU_DECL_RFE(i16, int16_t)
RDGPU_DECL_RFE(u16, uint16_t)
RDGPU_DECL_RFE(i32, int32_t)
RDGPU_DECL_RFE(u32, uint32_t)
RDGPU_DECL_RFE(f32, float)
RDGPU_DECL_RFE(f64, double)
RDGPU_DECL_RFE(i8, int8_t)
RDGPU_DECL_RFE(i64, int64_t)
RDGPU_DECL_RFE(u64, uint64_t)
#undef RDGPU_DECL_RFE

/* ---- flat resolution over row-block shards ---------------------------
 * barnes_flat_resolution_d8 (flats/flat_resolution.hpp:587-605) when the raster is split into row blocks,
 * one per GPU.  d_rows = the shard's own rows plus TWO ghost rows per cut (ghost_top / ghost_bottom are 0
 * at the raster's first / last block, 2 otherwise), device resident and kept alive until _free.
 * Protocol, for phase 0 (towards low edges) and then phase 1 (away from high edges):
 *     repeat { _relax(phase); _boundary(phase, out[2w]);  all-gather;  stop when no gathered row changed;
 *              _inject(phase, last own row of the block above, first own row of the block below) }
 * then  _heights(out[8w]);  all-gather;  rdgpu_flat_graph_solve_dev(gathered, world, w, solved[world*4w]);
 *       _finish(solved + rank*4w, dirs of the own rows).
 * The result equals rdgpu_flat_resolution_d8 of the whole raster restricted to the own rows.           */
typedef struct rdgpu_flat_shard rdgpu_flat_shard;
int rdgpu_flat_shard_begin_u8(const uint8_t *d_rows, uint8_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_i16(const int16_t *d_rows, int16_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_u16(const uint16_t *d_rows, uint16_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_i32(const int32_t *d_rows, int32_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_u32(const uint32_t *d_rows, uint32_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_f32(const float *d_rows, float nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_f64(const double *d_rows, double nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_i8(const int8_t *d_rows, int8_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_i64(const int64_t *d_rows, int64_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_begin_u64(const uint64_t *d_rows, uint64_t nodata, int width, int rows, int ghost_top, int ghost_bottom, void *hip_stream, rdgpu_flat_shard **out);
int rdgpu_flat_shard_relax(rdgpu_flat_shard *shard, int phase);
int rdgpu_flat_shard_boundary(rdgpu_flat_shard *shard, int phase, int32_t *d_out_2w);
int rdgpu_flat_shard_inject(rdgpu_flat_shard *shard, int phase, const int32_t *d_row_above, const int32_t *d_row_below);
int rdgpu_flat_shard_heights(rdgpu_flat_shard *shard, int32_t *d_out_8w);
int rdgpu_flat_graph_solve_dev(const int32_t *d_gathered, int world, int width, int32_t *d_out, void *hip_stream);
int rdgpu_flat_shard_finish(rdgpu_flat_shard *shard, const int32_t *d_heights_4w, uint8_t *d_dirs_out);
int rdgpu_flat_shard_rounds(const rdgpu_flat_shard *shard, int phase);
void rdgpu_flat_shard_free(rdgpu_flat_shard *shard);

/* barnes_flat_resolution_d8(alter = false) of ONE raster over SEVERAL devices of this process: row block s (+ two ghost
 * rows per cut) on devices[s], a host thread per device, the cut rows of the two level fields exchanged from device to
 * device (peer copies; RDGPU_MULTI_HOST_STAGED=1: through the host) until none changes, the flat heights agreed on devices[0]; the result equals rdgpu_flat_resolution_d8_<T> on the whole
 * raster.  A device may be listed more than once.  rdgpu_flat_resolution_d8_<T> takes this path when RDGPU_DEVICES
 * lists two or more ids.  (Verified on one physical device listed several times.) */
#define RDGPU_DECL_FLATS_MULTI(SUF, T) \
  int rdgpu_flat_resolution_d8_multi_##SUF(const T *dem, T nodata, int width, int height, uint8_t *dirs, const int *devices, int ndevices);
RDGPU_DECL_FLATS_MULTI(u8, uint8_t)
RDGPU_DECL_FLATS_MULTI(i8, int8_t)
RDGPU_DECL_FLATS_MULTI(i16, int16_t)
RDGPU_DECL_FLATS_MULTI(u16, uint16_t)
RDGPU_DECL_FLATS_MULTI(i32, int32_t)
RDGPU_DECL_FLATS_MULTI(u32, uint32_t)
RDGPU_DECL_FLATS_MULTI(f32, float)
RDGPU_DECL_FLATS_MULTI(f64, double)
RDGPU_DECL_FLATS_MULTI(i64, int64_t)
RDGPU_DECL_FLATS_MULTI(u64, uint64_t)
#undef RDGPU_DECL_FLATS_MULTI

/* ---- d8_flow_accum(const Array2D<uint8_t>& flowdirs, Array2D<A>& area) ----------------------
 * Replaces richdem::d8_flow_accum (include/richdem/methods/d8_methods.hpp:47-139): area = number of
 * cells draining through each cell (itself included); cells whose direction equals dir_nodata get
 * -1 (area.noData(), :64).  Exact (integer arithmetic) for every output type. */
int rdgpu_d8_flow_accum_i32(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, int32_t *area);
int rdgpu_d8_flow_accum_f32(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, float *area);
int rdgpu_d8_flow_accum_f64(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, double *area);
int rdgpu_d8_flow_accum_dev_i32(const uint8_t *d_dirs, uint8_t dir_nodata, int width, int height, int32_t *d_area, void *hip_stream);
int rdgpu_d8_flow_accum_dev_f32(const uint8_t *d_dirs, uint8_t dir_nodata, int width, int height, float *d_area, void *hip_stream);
int rdgpu_d8_flow_accum_dev_f64(const uint8_t *d_dirs, uint8_t dir_nodata, int width, int height, double *d_area, void *hip_stream);

/* ---- row-block shards of d8_flow_accum: the protocol of programs/parallel_d8_accum over GPUs -------
 * (reference programs/parallel_d8_accum/main.cpp: per-tile accumulation :373-464, paths leaving through the
 * perimeter :270-334, inflow added along the in-tile path :344-370.)  Each rank holds a row block of the
 * uint8 directions plus the adjacent direction row of each neighbouring block (NULL at the DEM edge).
 *   begin    pending-inflow counts (including inflow across the cuts) + all walks that start in the block;
 *            a walk that crosses a cut drops (arrivals << 56 | total) into the outbox slot of the
 *            receiving column
 *   outbox   d_out[2][width] <- {sent up, sent down}; clears the outboxes
 *   inject   the neighbours' outboxes arrive; completed cut-row cells resume walking
 *   (repeat outbox / exchange / inject until every outbox of every rank is empty)
 *   finish   area block in the requested type; releases the handle.
 * Integer arithmetic, exact; the result equals rdgpu_d8_flow_accum_* on the whole raster. */
typedef struct rdgpu_accum_shard rdgpu_accum_shard;
int rdgpu_accum_shard_begin(const uint8_t *d_dirs_rows, uint8_t dir_nodata, int width, int rows,
                            const uint8_t *d_row_above, const uint8_t *d_row_below, void *hip_stream,
                            rdgpu_accum_shard **out);
/* One exchange instead of one per cut crossing (the protocol of programs/parallel_d8_accum/main.cpp:270-464), for
 * directions without loops (every direction raster derived from a DEM):
 *   begin_local  pending counts from the block's own cells only: every cell completes, the outboxes hold what the
 *                block's own cells send across each cut
 *   links        d_links[2][width]: where the flow ENTERING at each cell of the first / last row leaves the block
 *                again -- (1 << 31 if across the lower cut) | receiving column, or -1; *d_pending: cells the local
 *                phase could not complete (a direction loop: fall back to begin / outbox / inject)
 *   (one all-gather of outbox + links; every rank solves the forest over the cut-row cells:
 *    richdem_amd/sharded.py accum_link_solve)
 *   add_paths    the inflow of each entry cell is added along its path inside the block
 *   finish       as above. */
int rdgpu_accum_shard_begin_local(const uint8_t *d_dirs_rows, uint8_t dir_nodata, int width, int rows,
                                  const uint8_t *d_row_above, const uint8_t *d_row_below, void *hip_stream,
                                  rdgpu_accum_shard **out);
int rdgpu_accum_shard_links(rdgpu_accum_shard *shard, int32_t *d_links, unsigned long long *d_pending);
int rdgpu_accum_shard_add_paths(rdgpu_accum_shard *shard, const unsigned long long *d_in_top,
                                const unsigned long long *d_in_bottom);
int rdgpu_accum_shard_outbox(rdgpu_accum_shard *shard, unsigned long long *d_out);
int rdgpu_accum_shard_inject(rdgpu_accum_shard *shard, const unsigned long long *d_from_above,
                             const unsigned long long *d_from_below);
int rdgpu_accum_shard_finish_i32(rdgpu_accum_shard *shard, int32_t *d_area);
int rdgpu_accum_shard_finish_f32(rdgpu_accum_shard *shard, float *d_area);
int rdgpu_accum_shard_finish_f64(rdgpu_accum_shard *shard, double *d_area);
int rdgpu_accum_shard_free(rdgpu_accum_shard *shard);

/* d8_flow_accum of ONE raster over SEVERAL devices of this process (the reference's tiled driver
 * programs/parallel_d8_accum/main.cpp:373-464 as a library call): row block s on devices[s] (a host thread per device,
 * its own PCIe link), one exchange of the cut rows' outboxes and links -- device to device by peer copies, the forest over
 * the cut rows solved on devices[0]; RDGPU_MULTI_HOST_STAGED=1: through the host --, the same result as
 * rdgpu_d8_flow_accum_<A> on the whole raster.  Direction loops fall back to devices[0] alone.  A device may be listed
 * more than once.  rdgpu_d8_flow_accum_<A> takes this path when RDGPU_DEVICES lists two or more ids.
 * (Verified on one physical device listed several times: this build's test boxes have one GPU.) */
int rdgpu_d8_flow_accum_multi_i32(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, int32_t *area, const int *devices, int ndevices);
int rdgpu_d8_flow_accum_multi_f32(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, float *area, const int *devices, int ndevices);
int rdgpu_d8_flow_accum_multi_f64(const uint8_t *dirs, uint8_t dir_nodata, int width, int height, double *area, const int *devices, int ndevices);

/* ---- FA_D8(const Array2D<T>& elevations, Array2D<double>& accum) ----------------------------
 * Replaces richdem::FA_D8 (include/richdem/methods/flow_accumulation.hpp:27) = FM_D8
 * (flowmet/OCallaghan1984.hpp:13-77) + FlowAccumulation (methods/flow_accumulation_generic.hpp:33-100)
 * without materialising the 36 B/cell Array3D.  accum is in/out: on entry the flow each cell
 * generates (1 by default in the reference's callers), on return the accumulation; NoData cells get
 * -1 (ACCUM_NO_DATA).  Exact for integer-valued weights; for general weights the f64 summation order
 * differs from the reference's FIFO order (results agree to f64 rounding, <= 1 ULP after an f32 cast). */
int rdgpu_fa_d8_u8(const uint8_t *dem, uint8_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_i16(const int16_t *dem, int16_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_u16(const uint16_t *dem, uint16_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_i32(const int32_t *dem, int32_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_u32(const uint32_t *dem, uint32_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_f32(const float *dem, float nodata, int width, int height, double *accum);
int rdgpu_fa_d8_f64(const double *dem, double nodata, int width, int height, double *accum);
int rdgpu_fa_d8_i8(const int8_t *dem, int8_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_i64(const int64_t *dem, int64_t nodata, int width, int height, double *accum);
int rdgpu_fa_d8_u64(const uint64_t *dem, uint64_t nodata, int width, int height, double *accum);
/* FM_D8 alone (richdem::FM_D8, flowmet/OCallaghan1984.hpp:81-84) as the 9-float proportions array */
int rdgpu_fm_d8_u8(const uint8_t *dem, uint8_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_i16(const int16_t *dem, int16_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_u16(const uint16_t *dem, uint16_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_i32(const int32_t *dem, int32_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_u32(const uint32_t *dem, uint32_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_f32(const float *dem, float nodata, int width, int height, float *props9);
int rdgpu_fm_d8_f64(const double *dem, double nodata, int width, int height, float *props9);
int rdgpu_fm_d8_i8(const int8_t *dem, int8_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_i64(const int64_t *dem, int64_t nodata, int width, int height, float *props9);
int rdgpu_fm_d8_u64(const uint64_t *dem, uint64_t nodata, int width, int height, float *props9);
int rdgpu_fa_d8_dev_u8(const uint8_t *d_dem, uint8_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_i16(const int16_t *d_dem, int16_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_u16(const uint16_t *d_dem, uint16_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_i32(const int32_t *d_dem, int32_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_u32(const uint32_t *d_dem, uint32_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_f32(const float *d_dem, float nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_f64(const double *d_dem, double nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_i8(const int8_t *d_dem, int8_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_i64(const int64_t *d_dem, int64_t nodata, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_fa_d8_dev_u64(const uint64_t *d_dem, uint64_t nodata, int width, int height, double *d_accum, void *hip_stream);
/* FA_D8 when the CALLER KNOWS that every cell generates a flow of 1 -- it built the accumulation array itself, as
 * apps/rd_flow_accumulation.cpp:13 (Array2D<double> accum(dem, 1)) and rd.FlowAccumulation(weights=None) do: accum is
 * OUTPUT ONLY.  The plain entries above have to read the weights once to find that out (12.8 GB at 40000^2) and, on the
 * host path, upload them (12.8 GB over PCIe); these do neither.  Same result as the plain entry on an array of ones. */
#define RDGPU_DECL_FA_UNIT(SUF, T)                                                                              \
  int rdgpu_fa_d8_unit_##SUF(const T *dem, T nodata, int width, int height, double *accum_out);                 \
  int rdgpu_fa_d8_unit_dev_##SUF(const T *d_dem, T nodata, int width, int height, double *d_accum_out, void *hip_stream);
RDGPU_DECL_FA_UNIT(u8, uint8_t) RDGPU_DECL_FA_UNIT(i8, int8_t) RDGPU_DECL_FA_UNIT(i16, int16_t) RDGPU_DECL_FA_UNIT(u16, uint16_t)
RDGPU_DECL_FA_UNIT(i32, int32_t) RDGPU_DECL_FA_UNIT(u32, uint32_t) RDGPU_DECL_FA_UNIT(f32, float) RDGPU_DECL_FA_UNIT(f64, double)
RDGPU_DECL_FA_UNIT(i64, int64_t) RDGPU_DECL_FA_UNIT(u64, uint64_t)
#undef RDGPU_DECL_FA_UNIT

/* ---- D-infinity (Tarboton 1997) and the generic FlowAccumulation ---------------------------------
 * rdgpu_dinf_flowdirs_<T>   replaces richdem::dinf_flow_directions (include/richdem/flowmet/dinf_flowdirs.hpp
 *                           :128-152): float32 angle in [0, 2*pi), 0 = NO_FLOW, -1 = NoData.
 * rdgpu_fm_tarboton_<T>     replaces richdem::FM_Tarboton / FM_Dinfinity (flowmet/Tarboton1997.hpp:14-144):
 *                           9 floats per cell, index 9*i+n (common/Array3D.hpp:203-206).
 * rdgpu_fa_tarboton_<T>     replaces richdem::FA_Tarboton / FA_Dinfinity (methods/flow_accumulation.hpp:16-17);
 *                           accum in/out like rdgpu_fa_d8_*.
 * rdgpu_flow_accumulation_f64  replaces richdem::FlowAccumulation(const Array3D<float>&, Array2D<double>&)
 *                           (methods/flow_accumulation_generic.hpp:33-100) for ANY proportions array.
 * Angles/proportions use double atan2/sqrt as the reference does; device libm may differ from glibc in the
 * last ulp, so these are specified to <= 1 ULP (f32), not bit-exact; accumulation sums in a different
 * order than the reference's FIFO (exact for integer-valued flows). */
#define RDGPU_DECL_MFD(SUF, T)                                                                          \
  int rdgpu_dinf_flowdirs_##SUF(const T *dem, T nodata, int width, int height, float *angles);           \
  int rdgpu_dinf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int width, int height, float *d_angles, void *hip_stream); \
  int rdgpu_fm_tarboton_##SUF(const T *dem, T nodata, int width, int height, float *props9);             \
  int rdgpu_fa_tarboton_##SUF(const T *dem, T nodata, int width, int height, double *accum);             \
  int rdgpu_fa_tarboton_dev_##SUF(const T *d_dem, T nodata, int width, int height, double *d_accum, void *hip_stream);
RDGPU_DECL_MFD(u8, uint8_t)
RDGPU_DECL_MFD(i16, int16_t)
RDGPU_DECL_MFD(u16, uint16_t)
RDGPU_DECL_MFD(i32, int32_t)
RDGPU_DECL_MFD(u32, uint32_t)
RDGPU_DECL_MFD(f32, float)
RDGPU_DECL_MFD(f64, double)
RDGPU_DECL_MFD(i8, int8_t)
#undef RDGPU_DECL_MFD
/* FM_Holmgren(x) / FM_Freeman(x) / FM_Quinn / FM_D4 proportions and the matching FA_* accumulations
 * (flowmet/Holmgren1994.hpp:14, Freeman1991.hpp:14, Quinn1991.hpp:13, OCallaghan1984.hpp:86;
 * methods/flow_accumulation.hpp:18-20,28).  method: 0 Holmgren, 1 Freeman, 2 Quinn, 3 D4; xparam is the
 * exponent of methods 0 and 1.  accum is in/out as for rdgpu_fa_d8 (in: flow generated per cell).      */
#define RDGPU_DECL_MFD2(SUF, T)                                                                          \
  int rdgpu_fm_mfd_##SUF(const T *dem, T nodata, int width, int height, int method, double xparam, float *props9); \
  int rdgpu_fm_mfd_dev_##SUF(const T *d_dem, T nodata, int width, int height, int method, double xparam, float *d_props9, void *hip_stream); \
  int rdgpu_fa_mfd_##SUF(const T *dem, T nodata, int width, int height, int method, double xparam, double *accum); \
  int rdgpu_fa_mfd_dev_##SUF(const T *d_dem, T nodata, int width, int height, int method, double xparam, double *d_accum, void *hip_stream);
RDGPU_DECL_MFD2(u8, uint8_t)
RDGPU_DECL_MFD2(i16, int16_t)
RDGPU_DECL_MFD2(u16, uint16_t)
RDGPU_DECL_MFD2(i32, int32_t)
RDGPU_DECL_MFD2(u32, uint32_t)
RDGPU_DECL_MFD2(f32, float)
RDGPU_DECL_MFD2(f64, double)
RDGPU_DECL_MFD2(i8, int8_t)
#undef RDGPU_DECL_MFD2
int rdgpu_flow_accumulation_f64(const float *props9, int width, int height, double *accum);
int rdgpu_flow_accumulation_dev_f64(const float *d_props9, int width, int height, double *d_accum, void *hip_stream);
int rdgpu_flow_accumulation_rounds(uint32_t *rounds); /* work-list rounds of the last generic accumulation */

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
