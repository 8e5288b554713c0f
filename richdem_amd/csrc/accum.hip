// accum.hip -- D8 flow accumulation over the flow forest: "last arriver continues".
//
//  * rdgpu_d8_flow_accum_*   replaces d8_flow_accum (reference include/richdem/methods/d8_methods.hpp:47-139)
//  * rdgpu_fa_d8_*           replaces FA_D8 = FM_D8 + FlowAccumulation
//                            (methods/flow_accumulation.hpp:27, flowmet/OCallaghan1984.hpp:13-77,
//                             methods/flow_accumulation_generic.hpp:33-100)
//
// The reference walks the forest in FIFO Kahn order on one thread.  Here every source cell (no inflow)
// starts one GPU thread that carries its total downstream; at each cell it adds its total with ONE
// device-scope atomic that also decrements the cell's pending-inflow count, and only the thread whose
// add completes the cell (the last arriver) continues downstream with the now-final total.  No queues,
// no levels, one launch; work O(cells); the critical path is the longest flow path.
//
//  unit weights (d8_flow_accum): 64-bit word per cell = pending inflows (bits 56..63) | area (bits 0..55),
//      one atomicAdd(word, area - (1<<56)) per step; integer, exact, order independent.
//  f64 weights (FA_D8): f64 atomicAdd on the total + release/acquire decrement of a separate counter.
#include "common.hpp"
#include "flowdirs.hpp"

namespace rdgpu {

constexpr int NTHR = 256;
constexpr unsigned long long CNT1 = 1ull << 56;
constexpr unsigned long long LOWMASK = CNT1 - 1ull;
constexpr unsigned long long SRC = 0xFFull;        // count-field marker of a source cell (unit path)
constexpr uint32_t SRC32 = 0xFFFFFFFEu;            // pending marker of a source cell (f64 path)
constexpr uint32_t NODATA32 = 0xFFFFFFFFu;

// D8 neighbour offsets, numbering 234/105/876 (reference common/constants.hpp:44-45)
__device__ __forceinline__ int d8dx(int n) { return (n == 1 || n == 2 || n == 8) ? -1 : (n >= 4 && n <= 6) ? 1 : 0; }
__device__ __forceinline__ int d8dy(int n) { return (n >= 2 && n <= 4) ? -1 : (n >= 6 && n <= 8) ? 1 : 0; }

// target cell of c under direction n, or -1 when there is none / it is off the grid
__device__ __forceinline__ int64_t flow_target(uint32_t c, int n, int w, int h) {
  if (n < 1 || n > 8) return -1;
  const int x = (int)(c % (uint32_t)w) + d8dx(n), y = (int)(c / (uint32_t)w) + d8dy(n);
  if (x < 0 || y < 0 || x >= w || y >= h) return -1;
  return (int64_t)y * w + x;
}

// number of in-grid neighbours whose direction points at (x, y); NoData-direction cells never flow
__device__ __forceinline__ int inflow_count(const uint8_t *__restrict__ dirs, uint8_t nodata, int x, int y, int w,
                                            int h) {
  int cnt = 0;
#pragma unroll
  for (int n = 1; n <= 8; n++) {
    const int nx = x + d8dx(n), ny = y + d8dy(n);
    if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
    const uint8_t d = dirs[(size_t)ny * w + nx];
    if (d == nodata) continue;
    // neighbour n flows into us iff its direction is the inverse of n (constants.hpp:65 d8_inverse)
    const int inv = n <= 4 ? n + 4 : n - 4;
    if (d == inv) cnt++;
  }
  return cnt;
}

// ---- unit weights ---------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_acc_init_unit(const uint8_t *__restrict__ dirs, uint8_t nodata,
                                                        unsigned long long *word, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    unsigned long long v = 0;
    if (dirs[c] != nodata) {
      const int k = inflow_count(dirs, nodata, x, y, w, h);
      // SRC marks a source: nobody ever adds to it, so the marker is stable while other cells' counts
      // run down to 0 (testing "count == 0" in the walk kernel would race with last arrivals)
      v = ((unsigned long long)(k == 0 ? SRC : k) << 56) | 1ull;  // low field: the cell's own area
    }
    word[c] = v;
  }
}

__global__ __launch_bounds__(NTHR) void k_acc_walk_unit(const uint8_t *__restrict__ dirs, uint8_t nodata,
                                                        unsigned long long *word, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c0 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c0 < n; c0 += stride) {
    uint8_t d = dirs[c0];
    if (d == nodata) continue;
    // plain read is safe: a source's word is never modified by anyone
    if ((word[c0] >> 56) != SRC) continue;  // has inflow: some last arriver will come through here
    uint32_t c = (uint32_t)c0;
    unsigned long long v = 1;
    for (;;) {
      const int64_t t = flow_target(c, d, w, h);                 // d8_methods.hpp:113-122
      if (t < 0) break;
      const uint8_t dt = dirs[t];
      if (dt == nodata) break;                                   // :124-125 flow into NoData is dropped
      const unsigned long long old = atomicAdd(&word[t], v - CNT1);
      if ((old >> 56) != 1) break;                               // not the last inflow of t
      v = (old & LOWMASK) + v;                                   // t's final area
      c = (uint32_t)t;
      d = dt;
    }
  }
}

template <class A>
__global__ __launch_bounds__(NTHR) void k_acc_out_unit(const uint8_t *__restrict__ dirs, uint8_t nodata,
                                                       const unsigned long long *__restrict__ word, A *area,
                                                       uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    if (dirs[c] == nodata) { area[c] = (A)-1; continue; }        // area.noData() == -1, d8_methods.hpp:64,:72-75
    const unsigned long long v = word[c];
    // cells downstream of a direction loop are never completed by the reference either: they keep the
    // sum of the inflows that did arrive, without their own +1 (d8_methods.hpp:104-131)
    const unsigned long long cnt = v >> 56;
    const unsigned long long a = (cnt != 0 && cnt != SRC) ? (v & LOWMASK) - 1 : (v & LOWMASK);
    area[c] = (A)a;
  }
}

// ---- f64 weights ----------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_acc_init_f64(const uint8_t *__restrict__ dirs, uint32_t *pending, int w,
                                                       int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    uint32_t p = NODATA32;
    if (dirs[c] != 255) {
      const int k = inflow_count(dirs, 255, x, y, w, h);
      p = k == 0 ? SRC32 : (uint32_t)k;
    }
    pending[c] = p;
  }
}

__global__ __launch_bounds__(NTHR) void k_acc_walk_f64(const uint8_t *__restrict__ dirs, uint32_t *pending,
                                                       double *acc, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c0 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c0 < n; c0 += stride) {
    uint8_t d = dirs[c0];
    if (d == 255) continue;
    if (pending[c0] != SRC32) continue;   // sources only (a source's counter is never written)
    uint32_t c = (uint32_t)c0;
    double v = acc[c0];               // a source's total is its own generated flow
    for (;;) {
      const int64_t t = flow_target(c, d, w, h);
      if (t < 0) break;
      const uint8_t dt = dirs[t];
      if (dt == 255) break;                                        // flow_accumulation_generic.hpp:85-86
      // :87 (proportion is exactly 1 for D8).  All three steps are device-scope atomic RMWs executed
      // at the memory side: a RETURNING add has completed there before the decrement is issued, and the
      // last arriver's decrement is ordered after every other arriver's decrement, hence after their
      // adds -- no cache write-back / invalidate (release/acquire fences cost ~10x here) is needed.
      const double prev = atomicAdd(&acc[t], v);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");   // the add has returned
      const uint32_t old = __hip_atomic_fetch_sub(&pending[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old != 1) break;
      v = atomicAdd(&acc[t], 0.0);                                 // final total, read at the memory side
      c = (uint32_t)t;
      d = dt;
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_acc_nodata_f64(const uint8_t *__restrict__ dirs, double *acc, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride)
    if (dirs[c] == 255) acc[c] = -1.0;                             // ACCUM_NO_DATA, :95-97
}

// ---- drivers --------------------------------------------------------------------------------
static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

static void check_dims(int w, int h, const char *who) {
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, std::string(who) + ": width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0xFFFF0000ull) throw Error(RDGPU_ERR_ARG, std::string(who) + ": raster too large");
}

template <class A>
void d8_flow_accum_device(const uint8_t *d_dirs, uint8_t nodata, int w, int h, A *d_area, hipStream_t s) {
  if (!d_dirs || !d_area) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum: null pointer");
  check_dims(w, h, "rdgpu_d8_flow_accum");
  const uint64_t n = (uint64_t)w * h;
  unsigned long long *word = Workspace::get().buf<unsigned long long>("accum.word", n);
  RD_LAUNCH("accum.init_unit", k_acc_init_unit, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, nodata, word, w, h);
  RD_LAUNCH("accum.walk_unit", k_acc_walk_unit, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, nodata, word, w, h);
  RD_LAUNCH("accum.out_unit", (k_acc_out_unit<A>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, nodata, word, d_area, n);
}

// d_dirs uses 255 as NoData marker (output of flowdirs_device); d_acc holds the per-cell weights on entry.
void flow_accum_f64_device(const uint8_t *d_dirs, int w, int h, double *d_acc, hipStream_t s) {
  const uint64_t n = (uint64_t)w * h;
  uint32_t *pending = Workspace::get().buf<uint32_t>("accum.pending", n);
  RD_LAUNCH("accum.init_f64", k_acc_init_f64, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, pending, w, h);
  RD_LAUNCH("accum.walk_f64", k_acc_walk_f64, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, pending, d_acc, w, h);
  RD_LAUNCH("accum.nodata_f64", k_acc_nodata_f64, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, d_acc, n);
}

template <class T>
void fa_d8_device(const T *d_z, T nodata, int w, int h, double *d_acc, hipStream_t s) {
  if (!d_z || !d_acc) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_d8: null pointer");
  check_dims(w, h, "rdgpu_fa_d8");
  uint8_t *dirs = Workspace::get().buf<uint8_t>("accum.fmdirs", (size_t)w * h);
  flowdirs_device<T>(d_z, nodata, w, h, dirs, MODE_FM, s);
  flow_accum_f64_device(dirs, w, h, d_acc, s);
}

template <class A>
static void d8_flow_accum_host(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area) {
  if (!dirs || !area) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum: null pointer");
  check_dims(w, h, "rdgpu_d8_flow_accum");
  const size_t n = (size_t)w * h;
  uint8_t *dd = Workspace::get().buf<uint8_t>("host.dirs", n);
  A *da = Workspace::get().buf<A>("host.area", n);
  RD_HIP(hipMemcpy(dd, dirs, n, hipMemcpyHostToDevice));
  d8_flow_accum_device<A>(dd, nodata, w, h, da, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(area, da, n * sizeof(A), hipMemcpyDeviceToHost));
}

template <class T>
static void fa_d8_host(const T *dem, T nodata, int w, int h, double *accum) {
  if (!dem || !accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_d8: null pointer");
  check_dims(w, h, "rdgpu_fa_d8");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  double *da = Workspace::get().buf<double>("host.area", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  RD_HIP(hipMemcpy(da, accum, n * sizeof(double), hipMemcpyHostToDevice));
  fa_d8_device<T>(d, nodata, w, h, da, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(accum, da, n * sizeof(double), hipMemcpyDeviceToHost));
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_ACCUM_API(SUF, A)                                                                                 \
  extern "C" int rdgpu_d8_flow_accum_##SUF(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area) {     \
    return guarded([&] { d8_flow_accum_host<A>(dirs, nodata, w, h, area); });                                \
  }                                                                                                          \
  extern "C" int rdgpu_d8_flow_accum_dev_##SUF(const uint8_t *d_dirs, uint8_t nodata, int w, int h, A *d_area, \
                                               void *stream) {                                               \
    return guarded([&] { d8_flow_accum_device<A>(d_dirs, nodata, w, h, d_area, (hipStream_t)stream); });     \
  }
RD_ACCUM_API(i32, int32_t)
RD_ACCUM_API(f32, float)
RD_ACCUM_API(f64, double)

#define RD_FA_API(SUF, T)                                                                                    \
  extern "C" int rdgpu_fa_d8_##SUF(const T *dem, T nodata, int w, int h, double *accum) {                    \
    return guarded([&] { fa_d8_host<T>(dem, nodata, w, h, accum); });                                        \
  }                                                                                                          \
  extern "C" int rdgpu_fa_d8_dev_##SUF(const T *d_dem, T nodata, int w, int h, double *d_accum, void *stream) { \
    return guarded([&] { fa_d8_device<T>(d_dem, nodata, w, h, d_accum, (hipStream_t)stream); });             \
  }
RD_FA_API(u8, uint8_t)
RD_FA_API(i16, int16_t)
RD_FA_API(u16, uint16_t)
RD_FA_API(i32, int32_t)
RD_FA_API(u32, uint32_t)
RD_FA_API(f32, float)
RD_FA_API(f64, double)
