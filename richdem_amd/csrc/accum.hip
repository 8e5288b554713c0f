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

#include <algorithm>
#include <vector>

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

// ------------------------------------------------------------------------------------------
// Row-block shards of d8_flow_accum (reference programs/parallel_d8_accum/main.cpp: per-tile
// accumulation :373-464, flow leaving a tile through its perimeter :270-334, inflow added along the
// in-tile path :344-370).  Same "last arriver continues" engine; a walk that leaves the shard across a
// cut drops its total into an outbox slot of the receiving cell, (arrivals << 56 | sum), exactly the
// packed format of a cell's word.  The ranks exchange the two outbox rows, inject them (an arrival of
// multiplicity k completes a cell iff its pending count was k) and resume walking -- until no outbox is
// used any more.  Rounds = how often a flow path crosses a cut, not the path length.
// ------------------------------------------------------------------------------------------
struct AccShard {
  const uint8_t *dirs, *above, *below;   // shard rows; last row of the shard above / first row of the one below (or null)
  unsigned long long *word, *out_top, *out_bottom;
  int w, h;
  uint8_t nodata;
};

__device__ __forceinline__ uint8_t accs_dir(const AccShard &s, int x, int y) {
  if (x < 0 || x >= s.w) return s.nodata;            // treated as "never flows in"
  if (y < 0) return s.above ? s.above[x] : s.nodata;
  if (y >= s.h) return s.below ? s.below[x] : s.nodata;
  return s.dirs[(size_t)y * s.w + x];
}

__global__ __launch_bounds__(NTHR) void k_accs_init(AccShard s) {
  const uint64_t n = (uint64_t)s.w * s.h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)s.w), y = (int)(c / (uint64_t)s.w);
    unsigned long long v = 0;
    if (s.dirs[c] != s.nodata) {
      int k = 0;
#pragma unroll
      for (int m = 1; m <= 8; m++) {
        const uint8_t d = accs_dir(s, x + d8dx(m), y + d8dy(m));
        if (d == s.nodata) continue;
        if (d == (m <= 4 ? m + 4 : m - 4)) k++;
      }
      v = ((unsigned long long)(k == 0 ? SRC : k) << 56) | 1ull;
    }
    s.word[c] = v;
  }
}

// continue a walk from cell c (its word is final, total v)
__device__ __forceinline__ void accs_walk(const AccShard &s, uint32_t c, unsigned long long v) {
  uint8_t d = s.dirs[c];
  for (;;) {
    if (d < 1 || d > 8) return;
    const int x = (int)(c % (uint32_t)s.w) + d8dx(d), y = (int)(c / (uint32_t)s.w) + d8dy(d);
    if (x < 0 || x >= s.w) return;                               // off the DEM
    if (y < 0) {                                                 // across the upper cut (or off the DEM)
      if (s.above && s.above[x] != s.nodata) atomicAdd(&s.out_top[x], v + CNT1);
      return;
    }
    if (y >= s.h) {
      if (s.below && s.below[x] != s.nodata) atomicAdd(&s.out_bottom[x], v + CNT1);
      return;
    }
    const uint32_t t = (uint32_t)y * (uint32_t)s.w + (uint32_t)x;
    const uint8_t dt = s.dirs[t];
    if (dt == s.nodata) return;
    const unsigned long long old = atomicAdd(&s.word[t], v - CNT1);
    if ((old >> 56) != 1) return;
    v = (old & LOWMASK) + v;
    c = t;
    d = dt;
  }
}

__global__ __launch_bounds__(NTHR) void k_accs_walk_sources(AccShard s) {
  const uint64_t n = (uint64_t)s.w * s.h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    if (s.dirs[c] == s.nodata) continue;
    if ((s.word[c] >> 56) != SRC) continue;
    accs_walk(s, (uint32_t)c, 1ull);
  }
}

// arrivals from the neighbouring shards: in_top[x] is what the shard above sent to my row 0 cell x
__global__ __launch_bounds__(NTHR) void k_accs_inject(AccShard s, const unsigned long long *__restrict__ in_top,
                                                      const unsigned long long *__restrict__ in_bottom) {
  const int i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= 2 * s.w) return;
  const int x = i % s.w, y = i < s.w ? 0 : s.h - 1;
  const unsigned long long *in = i < s.w ? in_top : in_bottom;
  if (!in) return;
  if (s.h == 1 && i >= s.w && in_top) { /* single-row shard: both boxes hit row 0; handled below */ }
  const unsigned long long pk = in[x];
  if (pk == 0) return;
  const unsigned long long k = pk >> 56, sum = pk & LOWMASK;
  const uint32_t t = (uint32_t)y * (uint32_t)s.w + (uint32_t)x;
  const unsigned long long old = atomicAdd(&s.word[t], sum - (k << 56));
  if ((old >> 56) != k) return;            // other inflows of t are still pending
  accs_walk(s, t, (old & LOWMASK) + sum);
}

template <class A>
__global__ __launch_bounds__(NTHR) void k_accs_out(AccShard s, A *area) {
  const uint64_t n = (uint64_t)s.w * s.h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    if (s.dirs[c] == s.nodata) { area[c] = (A)-1; continue; }
    const unsigned long long v = s.word[c], cnt = v >> 56;
    area[c] = (A)((cnt != 0 && cnt != SRC) ? (v & LOWMASK) - 1 : (v & LOWMASK));
  }
}

}  // namespace rdgpu

struct rdgpu_accum_shard {
  rdgpu::AccShard s;
  hipStream_t stream = nullptr;
  std::vector<void *> owned;
};

namespace rdgpu {

static void accs_free(rdgpu_accum_shard *a) {
  if (!a) return;
  for (void *p : a->owned) (void)hipFree(p);
  delete a;
}

static rdgpu_accum_shard *accs_begin(const uint8_t *d_dirs, uint8_t nodata, int w, int h, const uint8_t *d_above,
                                     const uint8_t *d_below, hipStream_t st) {
  if (!d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_begin: null pointer");
  check_dims(w, h, "rdgpu_accum_shard_begin");
  rdgpu_accum_shard *a = new rdgpu_accum_shard();
  try {
    a->stream = st;
    const uint64_t n = (uint64_t)w * h;
    auto alloc = [&](size_t bytes) {
      void *p = nullptr;
      RD_HIP(hipMalloc(&p, bytes));
      a->owned.push_back(p);
      return p;
    };
    a->s = AccShard{d_dirs, d_above, d_below, (unsigned long long *)alloc(n * 8), (unsigned long long *)alloc((size_t)w * 8),
                    (unsigned long long *)alloc((size_t)w * 8), w, h, nodata};
    RD_HIP(hipMemsetAsync(a->s.out_top, 0, (size_t)w * 8, st));
    RD_HIP(hipMemsetAsync(a->s.out_bottom, 0, (size_t)w * 8, st));
    RD_LAUNCH("accum.shard_init", k_accs_init, dim3(sgrid(n)), dim3(NTHR), 0, st, a->s);
    RD_LAUNCH("accum.shard_walk", k_accs_walk_sources, dim3(sgrid(n)), dim3(NTHR), 0, st, a->s);
  } catch (...) {
    accs_free(a);
    throw;
  }
  return a;
}

}  // namespace rdgpu

using namespace rdgpu;

extern "C" int rdgpu_accum_shard_begin(const uint8_t *d_dirs, uint8_t dir_nodata, int w, int h, const uint8_t *d_row_above,
                                       const uint8_t *d_row_below, void *stream, rdgpu_accum_shard **out) {
  return guarded([&] {
    if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_begin: null output handle");
    *out = accs_begin(d_dirs, dir_nodata, w, h, d_row_above, d_row_below, (hipStream_t)stream);
  });
}

// d_out[2][w]: what this shard sends up (row 0) and down (row 1); the outboxes are cleared.
extern "C" int rdgpu_accum_shard_outbox(rdgpu_accum_shard *a, unsigned long long *d_out) {
  return guarded([&] {
    if (!a || !d_out) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_outbox: null pointer");
    const size_t b = (size_t)a->s.w * 8;
    RD_HIP(hipMemcpyAsync(d_out, a->s.out_top, b, hipMemcpyDeviceToDevice, a->stream));
    RD_HIP(hipMemcpyAsync(d_out + a->s.w, a->s.out_bottom, b, hipMemcpyDeviceToDevice, a->stream));
    RD_HIP(hipMemsetAsync(a->s.out_top, 0, b, a->stream));
    RD_HIP(hipMemsetAsync(a->s.out_bottom, 0, b, a->stream));
  });
}

// d_from_above[w]: the bottom outbox of the shard above; d_from_below[w]: the top outbox of the shard below
extern "C" int rdgpu_accum_shard_inject(rdgpu_accum_shard *a, const unsigned long long *d_from_above,
                                        const unsigned long long *d_from_below) {
  return guarded([&] {
    if (!a) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_inject: null handle");
    if (a->s.h == 1 && d_from_above && d_from_below)
      throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_inject: a shard between two cuts needs at least 2 rows");
    RD_LAUNCH("accum.shard_inject", k_accs_inject, dim3((2 * a->s.w + NTHR - 1) / NTHR), dim3(NTHR), 0, a->stream, a->s,
              d_from_above, d_from_below);
  });
}

#define RD_ACCS_FINISH(SUF, A)                                                                                \
  extern "C" int rdgpu_accum_shard_finish_##SUF(rdgpu_accum_shard *a, A *d_area) {                            \
    if (!a) { set_last_error("rdgpu_accum_shard_finish: null handle"); return RDGPU_ERR_ARG; }                \
    const int rc = guarded([&] {                                                                              \
      if (!d_area) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_finish: null pointer");                      \
      const uint64_t n = (uint64_t)a->s.w * a->s.h;                                                           \
      RD_LAUNCH("accum.shard_out", (k_accs_out<A>), dim3(sgrid(n)), dim3(NTHR), 0, a->stream, a->s, d_area);  \
      RD_HIP(hipStreamSynchronize(a->stream));                                                                \
    });                                                                                                       \
    accs_free(a);                                                                                             \
    return rc;                                                                                                \
  }
RD_ACCS_FINISH(i32, int32_t)
RD_ACCS_FINISH(f32, float)
RD_ACCS_FINISH(f64, double)

extern "C" int rdgpu_accum_shard_free(rdgpu_accum_shard *a) {
  accs_free(a);
  return RDGPU_OK;
}

namespace rdgpu {
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

// FM_D8 as the reference's 9-float proportions array (flowmet/OCallaghan1984.hpp:13-77, Array3D.hpp:203-206)
__global__ __launch_bounds__(256) void k_fm_props(const uint8_t *__restrict__ dirs, float *__restrict__ props, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += stride) {
    const int d = dirs[c];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      float v = -1.0f;                               // NO_FLOW_GEN, :26
      if (k == 0 && d == 255) v = -2.0f;             // NO_DATA_GEN, :37-40
      else if (k == 0 && d >= 1 && d <= 8) v = 0.0f; // HAS_FLOW_GEN, :70
      else if (k == d && d >= 1 && d <= 8) v = 1.0f; // :74
      props[9 * c + k] = v;
    }
  }
}

template <class T>
static void fm_d8_host(const T *dem, T nodata, int w, int h, float *props9) {
  if (!dem || !props9) throw rdgpu::Error(RDGPU_ERR_ARG, "rdgpu_fm_d8: null pointer");
  rdgpu::check_dims(w, h, "rdgpu_fm_d8");
  const size_t n = (size_t)w * h;
  T *d = rdgpu::Workspace::get().buf<T>("host.dem", n);
  uint8_t *dirs = rdgpu::Workspace::get().buf<uint8_t>("accum.fmdirs", n);
  float *p = rdgpu::Workspace::get().buf<float>("host.props", n * 9);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  rdgpu::flowdirs_device<T>(d, nodata, w, h, dirs, rdgpu::MODE_FM, nullptr);
  RD_LAUNCH("accum.fm_props", k_fm_props, dim3(rdgpu::sgrid(n)), dim3(256), 0, (hipStream_t) nullptr, (const uint8_t *)dirs, p,
            (uint64_t)n);
  RD_HIP(hipMemcpy(props9, p, n * 36, hipMemcpyDeviceToHost));
}

#define RD_FA_API(SUF, T)                                                                                    \
  extern "C" int rdgpu_fa_d8_##SUF(const T *dem, T nodata, int w, int h, double *accum) {                    \
    return guarded([&] { fa_d8_host<T>(dem, nodata, w, h, accum); });                                        \
  }                                                                                                          \
  extern "C" int rdgpu_fm_d8_##SUF(const T *dem, T nodata, int w, int h, float *props9) {                     \
    return guarded([&] { fm_d8_host<T>(dem, nodata, w, h, props9); });                                        \
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
