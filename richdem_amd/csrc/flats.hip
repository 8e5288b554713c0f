// flats.hip -- Barnes (2014) flat resolution on MI355X.
//
// Replaces barnes_flat_resolution_d8(elevations, flowdirs, alter=false)
// (reference include/richdem/flats/flat_resolution.hpp:587-605) =
//   d8_flow_directions                       (flowmet/d8_flowdirs.hpp:96-123)       -> flowdirs.hip
//   resolve_flats_barnes                     (flat_resolution.hpp:447-517)
//      find_flat_edges :381-418              -> k_flat_classify   (3x3 stencil)
//      label_this :331-355                   -> k_ccl_*           (lock-free union-find over the
//                                                equal-elevation 8-connected graph; root = lowest index)
//      BuildAwayGradient :152-198            -> k_flat_bfs<AWAY>  (level-synchronous multi-source BFS)
//      BuildTowardsCombinedGradient :241-298 -> k_flat_bfs<TOWARDS>
//   d8_flow_flats / d8_masked_FlowDir :96-116, :42-65 -> k_flat_dirs
//
// The reference's outputs depend only on BFS *levels* and on the *partition* into flats, both of which
// are order independent, so the parallel formulation reproduces flat_mask and the directions exactly.
// Two facts remove label lookups from the hot loops (proved in DESIGN.md section 5):
//   * adjacent cells of equal elevation are always in the same flat, so "same label" == "same elevation";
//   * a NO_FLOW cell is in a labelled (drainable) flat  <=>  the towards-BFS reaches it.
// Labels (union-find roots) are only needed for flat_height[label] = the deepest away level per flat.
#include "common.hpp"
#include "flowdirs.hpp"

#include <algorithm>
#include <cstring>

namespace rdgpu {

constexpr int NTHR = 256;
constexpr int BFS_BLOCKS = 1024;

__device__ __forceinline__ int fdx(int n) { return (n == 1 || n == 2 || n == 8) ? -1 : (n >= 4 && n <= 6) ? 1 : 0; }
__device__ __forceinline__ int fdy(int n) { return (n >= 2 && n <= 4) ? -1 : (n >= 6 && n <= 8) ? 1 : 0; }

// every lane of the wave must call this (pred may be false); returns the slot for lanes with pred
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t *counter) {
  const unsigned long long bal = __ballot(pred);
  if (bal == 0) return 0;
  const int lane = threadIdx.x & 63;
  const int leader = (int)__ffsll((long long)bal) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(bal));
  base = __shfl(base, leader, 64);
  return base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

// ------------------------------------------------------------------------------------------
// find_flat_edges (flat_resolution.hpp:381-418).  counters: [0] low edges, [1] high edges,
// [2] NO_FLOW cells.  WRITE=false only counts.
// ------------------------------------------------------------------------------------------
template <class T, bool WRITE>
__global__ __launch_bounds__(NTHR) void k_flat_classify(const T *__restrict__ z, const uint8_t *__restrict__ dirs,
                                                        int w, int h, uint32_t *low, uint32_t *high,
                                                        uint32_t *counters) {
  const uint64_t n = (uint64_t)w * h;
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  const uint64_t nround = (n + stride - 1) / stride * stride;  // keep whole waves in the loop for ballots
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < nround; c += stride) {
    bool is_low = false, is_high = false, noflow = false;
    if (c < n) {
      const uint8_t d = dirs[c];
      if (d != 255) {
        const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
        const T e = z[c];
        noflow = d == 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
          const int nx = x + fdx(k), ny = y + fdy(k);
          if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
          const size_t ni = (size_t)ny * w + nx;
          const uint8_t dn = dirs[ni];
          if (dn == 255) continue;
          if (!noflow) {
            if (dn == 0 && z[ni] == e) { is_low = true; break; }   // :406-408
          } else {
            if (e < z[ni]) { is_high = true; break; }              // :409-411
          }
        }
      }
    }
    if (WRITE) {
      const uint32_t sl = wave_append(is_low, &counters[0]);
      if (is_low) low[sl] = (uint32_t)c;
      const uint32_t sh = wave_append(is_high, &counters[1]);
      if (is_high) high[sh] = (uint32_t)c;
    } else {
      const unsigned long long bl = __ballot(is_low), bh = __ballot(is_high), bn = __ballot(noflow);
      if ((threadIdx.x & 63) == 0) {
        if (bl) atomicAdd(&counters[0], (uint32_t)__popcll(bl));
        if (bh) atomicAdd(&counters[1], (uint32_t)__popcll(bh));
        if (bn) atomicAdd(&counters[2], (uint32_t)__popcll(bn));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// label_this (:331-355) as connected components of the equal-elevation 8-graph.
// Lock-free union-find: parents always point to a LOWER cell index, so the structure is acyclic under
// any interleaving and the root of a component is its lowest cell index.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uf_find(uint32_t *L, uint32_t x) {
  uint32_t p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    x = p;
    p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}

__device__ __forceinline__ void uf_unite(uint32_t *L, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }   // hook the larger root under the smaller
    const uint32_t old = atomicMin(&L[a], b);
    if (old == a) return;   // a was still a root: done
    a = old;                // a had been hooked meanwhile: what remains is unite(old, b)
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_ccl_init(const T *__restrict__ z, uint32_t *L, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    const T e = z[c];
    uint32_t m = (uint32_t)c;
    // lower-index neighbours in increasing index order: NW, N, NE, W -> the first equal one is the minimum
    if (y > 0) {
      if (x > 0 && z[c - w - 1] == e) m = (uint32_t)(c - w - 1);
      else if (z[c - w] == e) m = (uint32_t)(c - w);
      else if (x < w - 1 && z[c - w + 1] == e) m = (uint32_t)(c - w + 1);
    }
    if (m == (uint32_t)c && x > 0 && z[c - 1] == e) m = (uint32_t)(c - 1);
    L[c] = m;
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_ccl_merge(const T *__restrict__ z, uint32_t *L, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    const T e = z[c];
    if (y > 0) {
      if (x > 0 && z[c - w - 1] == e) uf_unite(L, (uint32_t)c, (uint32_t)(c - w - 1));
      if (z[c - w] == e) uf_unite(L, (uint32_t)c, (uint32_t)(c - w));
      if (x < w - 1 && z[c - w + 1] == e) uf_unite(L, (uint32_t)c, (uint32_t)(c - w + 1));
    }
    if (x > 0 && z[c - 1] == e) uf_unite(L, (uint32_t)c, (uint32_t)(c - 1));
  }
}

__global__ __launch_bounds__(NTHR) void k_ccl_flatten(uint32_t *L, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t r = uf_find(L, (uint32_t)c);
    if (r != L[c]) L[c] = r;
  }
}

// fh[root] = -1: flat without a low edge (label 0 in the reference, :483-487); >= 0: labelled.
__global__ __launch_bounds__(NTHR) void k_flat_mark_low(const uint32_t *__restrict__ low, uint32_t nlow,
                                                        const uint32_t *__restrict__ L, int32_t *fh) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i < nlow) fh[L[low[i]]] = 0;
}

// BFS sources.  AWAY: high edges of labelled flats (:491-500), level 1 -> M = -1, flat_height >= 1.
//               TOWARDS: every low edge, level 1 -> M = 2*1 (:284; low edges never carry an away value).
template <bool AWAY>
__global__ __launch_bounds__(NTHR) void k_flat_seed(const uint32_t *__restrict__ src, uint32_t nsrc,
                                                    const uint32_t *__restrict__ L, int32_t *fh, int32_t *M,
                                                    uint32_t *queue, uint32_t *qtail) {
  const uint32_t i0 = blockIdx.x * NTHR + threadIdx.x;
  bool take = false;
  uint32_t c = 0;
  if (i0 < nsrc) {
    c = src[i0];
    if (AWAY) {
      const uint32_t r = L[c];
      take = fh[r] >= 0;
      if (take) { M[c] = -1; fh[r] = 1; }   // every writer of fh[r] in this launch stores the same value
    } else {
      take = true;
      M[c] = 2;
    }
  }
  const uint32_t slot = wave_append(take, qtail);
  if (take) queue[slot] = c;
}

// One BFS level: expands queue[bounds[lvl&3] .. bounds[(lvl+1)&3]) (cells claimed at level lvl), claims
// unvisited NO_FLOW neighbours of equal elevation at level lvl+1 and appends them.  The last block to
// finish publishes the new tail as bounds[(lvl+2)&3].
template <class T, bool AWAY>
__global__ __launch_bounds__(NTHR) void k_flat_bfs(const T *__restrict__ z, const uint8_t *__restrict__ dirs,
                                                   const uint32_t *__restrict__ L, int32_t *fh, int32_t *M,
                                                   uint32_t *queue, uint32_t *qtail, uint32_t *bounds,
                                                   uint32_t *done, int lvl, int w, int h) {
  const uint32_t start = bounds[lvl & 3], end = bounds[(lvl + 1) & 3];
  const uint32_t stride = gridDim.x * NTHR;
  const uint32_t span = end - start;
  const uint32_t nround = (span + 63u) / 64u * 64u;
  for (uint32_t i = blockIdx.x * NTHR + threadIdx.x; i < nround; i += stride) {
    const bool act = i < span;
    uint32_t c = 0;
    int x = 0, y = 0;
    T e = T();
    if (act) {
      c = queue[start + i];
      x = (int)(c % (uint32_t)w);
      y = (int)(c / (uint32_t)w);
      e = z[c];
    }
#pragma unroll
    for (int k = 1; k <= 8; k++) {
      bool claimed = false;
      uint32_t ni = 0;
      if (act) {
        const int nx = x + fdx(k), ny = y + fdy(k);
        if (nx >= 0 && ny >= 0 && nx < w && ny < h) {                       // labels.inGrid, :189 / :289
          ni = (uint32_t)ny * (uint32_t)w + (uint32_t)nx;
          if (dirs[ni] == 0 && z[ni] == e) {                                // same flat && NO_FLOW, :190-191
            if (AWAY) {
              if (M[ni] == 0) claimed = atomicCAS(&M[ni], 0, -(lvl + 1)) == 0;        // :178-180
            } else {
              const int32_t old = __hip_atomic_load(&M[ni], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (old <= 0) {                                                          // :279
                const int32_t nv = (old != 0 ? fh[L[ni]] + old : 0) + 2 * (lvl + 1);   // :281-284
                claimed = atomicCAS(&M[ni], old, nv) == old;
              }
            }
          }
        }
      }
      if (AWAY && claimed) {
        const uint32_t r = L[ni];
        if (fh[r] != lvl + 1) fh[r] = lvl + 1;   // flat_height[label] = loops (:181); same value from all writers
      }
      const uint32_t slot = wave_append(claimed, qtail);
      if (claimed) queue[slot] = ni;
    }
  }
  // publish the next level's end bound
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t prev = atomicAdd(done, 1u);
    if (prev == gridDim.x - 1) {
      __threadfence();
      bounds[(lvl + 2) & 3] = __hip_atomic_load(qtail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *done = 0;
    }
  }
}

// d8_masked_FlowDir (:42-65) for the NO_FLOW cells of drainable flats (M > 0), d8_flow_flats :96-116.
template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_dirs(const T *__restrict__ z, const int32_t *__restrict__ M,
                                                    uint8_t *dirs, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    if (dirs[c] != 0) continue;
    const int32_t mc = M[c];
    if (mc <= 0) continue;   // flat without outlet: stays NO_FLOW
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;   // interior only (:108-109); cannot happen
    const T e = z[c];
    int32_t m = mc;
    int dir = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
      const size_t ni = (size_t)(y + fdy(k)) * w + (x + fdx(k));
      if (!(z[ni] == e)) continue;                                  // labels(n) != labels(c), :56-57
      const int32_t v = M[ni];
      if (v < m || (v == m && dir > 0 && (dir & 1) == 0 && (k & 1) == 1)) {
        m = v;
        dir = k;
      }
    }
    dirs[c] = (uint8_t)dir;
  }
}

// label export for tests: lowest cell index of the flat + 1, or 0 for unlabelled cells
__global__ __launch_bounds__(NTHR) void k_flat_labels_out(const uint32_t *__restrict__ L, const int32_t *__restrict__ fh,
                                                          int32_t *out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t r = L[c];
    out[c] = fh[r] >= 0 ? (int32_t)(r + 1u) : 0;
  }
}

// ------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------
static rdgpu_flat_stats g_fstats;
static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

template <class T, bool AWAY>
static uint32_t run_bfs(const T *d_z, const uint8_t *d_dirs, const uint32_t *L, int32_t *fh, int32_t *M,
                        uint32_t *queue, uint32_t *ctrl, uint32_t nseed, int w, int h, hipStream_t s) {
  // ctrl: [0] qtail, [1] done counter, [4..7] bounds ring
  uint32_t *hw = Workspace::get().host_words();
  uint32_t init[8] = {nseed, 0, 0, 0, 0, 0, 0, 0};
  init[4 + (1 & 3)] = 0;       // level 1 slice = [bounds[1], bounds[2]) = [0, nseed)
  init[4 + (2 & 3)] = nseed;
  RD_HIP(hipMemcpyAsync(ctrl, init, sizeof(init), hipMemcpyHostToDevice, s));
  RD_HIP(hipStreamSynchronize(s));  // init[] is a stack buffer
  int lvl = 1;
  uint32_t levels = 0;
  if (nseed == 0) return 0;
  for (;;) {
    const int burst = 8;
    for (int b = 0; b < burst; b++, lvl++)
      RD_LAUNCH(AWAY ? "flats.bfs_away" : "flats.bfs_towards", (k_flat_bfs<T, AWAY>), dim3(BFS_BLOCKS), dim3(NTHR), 0, s,
                d_z, d_dirs, L, fh, M, queue, ctrl, ctrl + 4, ctrl + 1, lvl, w, h);
    // after `burst` levels: bounds[(lvl)&3] .. bounds[(lvl+1)&3] is the next slice; empty -> finished
    RD_HIP(hipMemcpyAsync(hw, ctrl, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    levels += burst;
    if (hw[4 + (lvl & 3)] == hw[4 + ((lvl + 1) & 3)]) break;
    if (lvl > (1 << 30)) throw Error(RDGPU_ERR_HIP, "rdgpu flat resolution: BFS did not terminate");
  }
  return levels;
}

// Computes flat_mask (M) for the DEM; d_dirs must hold d8_flow_directions output.
// Returns device pointers (workspace) to M, L, fh through the out parameters.
template <class T>
static void resolve_flats_device(const T *d_z, const uint8_t *d_dirs, int w, int h, int32_t **outM, uint32_t **outL,
                                 int32_t **outFh, hipStream_t s) {
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  uint32_t *ctrl = ws.buf<uint32_t>("flats.ctrl", 16);
  int32_t *M = ws.buf<int32_t>("flats.mask", n);
  *outM = M;
  *outL = nullptr;
  *outFh = nullptr;
  RD_HIP(hipMemsetAsync(M, 0, n * sizeof(int32_t), s));            // flat_mask.setAll(0), :469
  g_fstats = rdgpu_flat_stats{0, 0, 0, 0, 0};

  RD_HIP(hipMemsetAsync(ctrl, 0, 16 * sizeof(uint32_t), s));
  RD_LAUNCH("flats.classify_count", (k_flat_classify<T, false>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, d_dirs, w, h,
            (uint32_t *)nullptr, (uint32_t *)nullptr, ctrl);
  RD_HIP(hipMemcpyAsync(hw, ctrl, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  const uint32_t nlow = hw[0], nhigh = hw[1], nnoflow = hw[2];
  g_fstats.low_edges = nlow;
  g_fstats.high_edges = nhigh;
  g_fstats.noflow_cells = nnoflow;
  if (nlow == 0) return;   // no flats, or none with an outlet (:475-481)

  uint32_t *low = ws.buf<uint32_t>("flats.low", nlow);
  uint32_t *high = ws.buf<uint32_t>("flats.high", (size_t)nhigh + 1);
  RD_HIP(hipMemsetAsync(ctrl, 0, 16 * sizeof(uint32_t), s));
  RD_LAUNCH("flats.classify_write", (k_flat_classify<T, true>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, d_dirs, w, h,
            low, high, ctrl);

  uint32_t *L = ws.buf<uint32_t>("flats.L", n);
  int32_t *fh = ws.buf<int32_t>("flats.fh", n);
  *outL = L;
  *outFh = fh;
  RD_LAUNCH("flats.ccl_init", (k_ccl_init<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, L, w, h);
  RD_LAUNCH("flats.ccl_merge", (k_ccl_merge<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, L, w, h);
  RD_LAUNCH("flats.ccl_flatten", k_ccl_flatten, dim3(sgrid(n)), dim3(NTHR), 0, s, L, n);
  RD_HIP(hipMemsetAsync(fh, 0xFF, n * sizeof(int32_t), s));        // -1 everywhere
  RD_LAUNCH("flats.mark_low", k_flat_mark_low, dim3((nlow + NTHR - 1) / NTHR), dim3(NTHR), 0, s, low, nlow, L, fh);

  const size_t qcap = (size_t)nnoflow + std::max(nlow, nhigh) + 64;
  uint32_t *queue = ws.buf<uint32_t>("flats.queue", qcap);

  // away gradient
  if (nhigh > 0) {
    RD_HIP(hipMemsetAsync(ctrl, 0, 16 * sizeof(uint32_t), s));
    RD_LAUNCH("flats.seed_away", (k_flat_seed<true>), dim3((nhigh + NTHR - 1) / NTHR), dim3(NTHR), 0, s, high, nhigh, L,
              fh, M, queue, ctrl);
    RD_HIP(hipMemcpyAsync(hw, ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    g_fstats.away_levels = run_bfs<T, true>(d_z, d_dirs, L, fh, M, queue, ctrl, hw[0], w, h, s);
  }
  // towards + combined gradient
  RD_HIP(hipMemsetAsync(ctrl, 0, 16 * sizeof(uint32_t), s));
  RD_LAUNCH("flats.seed_towards", (k_flat_seed<false>), dim3((nlow + NTHR - 1) / NTHR), dim3(NTHR), 0, s, low, nlow, L,
            fh, M, queue, ctrl);
  g_fstats.towards_levels = run_bfs<T, false>(d_z, d_dirs, L, fh, M, queue, ctrl, nlow, w, h, s);
}

template <class T>
void flat_resolution_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, hipStream_t s) {
  if (!d_z || !d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: raster too large");
  flowdirs_device<T>(d_z, nodata, w, h, d_dirs, MODE_D8, s);
  int32_t *M, *fh;
  uint32_t *L;
  resolve_flats_device<T>(d_z, d_dirs, w, h, &M, &L, &fh, s);
  if (L)   // there is at least one low edge
    RD_LAUNCH("flats.masked_dirs", (k_flat_dirs<T>), dim3(sgrid((uint64_t)w * h)), dim3(NTHR), 0, s, d_z, M, d_dirs, w, h);
}

template <class T>
static void flat_resolution_host(const T *dem, T nodata, int w, int h, uint8_t *dirs, int32_t *mask, int32_t *labels) {
  if (!dem || !dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: raster too large");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  uint8_t *dd = Workspace::get().buf<uint8_t>("host.dirs", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  hipStream_t s = nullptr;
  flowdirs_device<T>(d, nodata, w, h, dd, MODE_D8, s);
  int32_t *M, *fh;
  uint32_t *L;
  resolve_flats_device<T>(d, dd, w, h, &M, &L, &fh, s);
  if (L) RD_LAUNCH("flats.masked_dirs", (k_flat_dirs<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d, M, dd, w, h);
  RD_HIP(hipStreamSynchronize(s));
  RD_HIP(hipMemcpy(dirs, dd, n, hipMemcpyDeviceToHost));
  if (mask) RD_HIP(hipMemcpy(mask, M, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (labels) {
    if (L) {
      int32_t *lo = Workspace::get().buf<int32_t>("host.labels", n);
      RD_LAUNCH("flats.labels_out", k_flat_labels_out, dim3(sgrid(n)), dim3(NTHR), 0, s, L, fh, lo, (uint64_t)n);
      RD_HIP(hipStreamSynchronize(s));
      RD_HIP(hipMemcpy(labels, lo, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    } else {
      memset(labels, 0, n * sizeof(int32_t));
    }
  }
}

#define RD_INST(T) template void flat_resolution_device<T>(const T *, T, int, int, uint8_t *, hipStream_t);
RD_INST(uint8_t) RD_INST(int16_t) RD_INST(uint16_t) RD_INST(int32_t) RD_INST(uint32_t) RD_INST(float) RD_INST(double)
#undef RD_INST

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FLATS_API(SUF, T)                                                                                   \
  extern "C" int rdgpu_flat_resolution_d8_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs) {         \
    return guarded([&] { flat_resolution_host<T>(dem, nodata, w, h, dirs, nullptr, nullptr); });               \
  }                                                                                                            \
  extern "C" int rdgpu_resolve_flats_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs, int32_t *mask, \
                                           int32_t *labels) {                                                  \
    return guarded([&] { flat_resolution_host<T>(dem, nodata, w, h, dirs, mask, labels); });                   \
  }                                                                                                            \
  extern "C" int rdgpu_flat_resolution_d8_dev_##SUF(const T *d_dem, T nodata, int w, int h, uint8_t *d_dirs,   \
                                                    void *stream) {                                            \
    return guarded([&] { flat_resolution_device<T>(d_dem, nodata, w, h, d_dirs, (hipStream_t)stream); });      \
  }
RD_FLATS_API(u8, uint8_t)
RD_FLATS_API(i16, int16_t)
RD_FLATS_API(u16, uint16_t)
RD_FLATS_API(i32, int32_t)
RD_FLATS_API(u32, uint32_t)
RD_FLATS_API(f32, float)
RD_FLATS_API(f64, double)

extern "C" int rdgpu_flat_get_stats(rdgpu_flat_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = g_fstats;
  return RDGPU_OK;
}
