// fill.hip -- depression filling on MI355X: descent forest -> basins -> raster Boruvka rounds.
//
// Replaces FillDepressions<D8/D4> (reference include/richdem/depressions/depressions.hpp:13-21,
// i.e. PriorityFlood_Zhou2016, depressions/Zhou2016.hpp:126-191, and PriorityFlood_Barnes2014<D4>,
// depressions/Barnes2014.hpp:230-304).  The reference is a serial priority-queue sweep; its
// RESULT is the unique surface
//        W(c) = min over paths c -> raster border of (max elevation on the path)
// (SURVEY.md section 0), computed with comparisons and copies only.  This file computes the same
// surface with a GPU-shaped algorithm (see DESIGN.md section 3 for the proof sketch):
//
//   1. k_descent      every cell points at its lowest (key, index) neighbour if that is lower than
//                     itself in the (key, index) total order; border cells drain "OUT".
//   2. k_chase        pointer chasing with path compression -> every cell knows its pit (root).
//   3. k_count_pits / k_scan_counts / k_assign_pits / k_label_cells
//                     pits get dense basin ids 0..B-1; lab[c] = basin of c (B = the outside).
//                     W(c) = max(z(c), L[basin(c)]) with L = minimax pass height basin -> outside.
//   4. rounds of      k_scan (raster pass: each component's lowest pass to a different component,
//                     one 64-bit atomicMin of (pass height << 32 | neighbour component)),
//                     k_hook (hook every component along its lowest pass; mutual pairs keep the
//                     smaller id as root), k_chase_links (pointer jumping carrying the path
//                     maximum), k_update_basins, k_compact_roots.  This is Boruvka's contraction:
//                     the number of live components at least halves per round.
//   5. k_finalize     z(c) <- max(z(c), acc[lab[c]]).
//
// All elevation work is on order-preserving 32-bit keys (common.hpp Key32), so it is exact for
// u8/i16/u16/i32/u32/f32.  HBM-bound integer/compare work: no MFMA anywhere.
#include "common.hpp"

namespace rdgpu {

constexpr int TW = 64;        // tile width  (cells)  = one wavefront per tile row
constexpr int TH = 16;        // tile height (cells)
constexpr int LW = TW + 2;    // LDS row stride incl. 1-cell halo (66 words: conflict-free rows)
constexpr int LH = TH + 2;
constexpr int NTHR = 256;     // 4 wavefronts
constexpr uint32_t OUTP = 0xFFFFFFFFu;  // descent pointer of a border cell: drains off the raster
constexpr int CELLS_PER_BLOCK = 4096;   // 1-D kernels: 256 threads x 16 cells

static rdgpu_fill_stats g_stats;

// ------------------------------------------------------------------------------------------
// 1. descent pointers
// ------------------------------------------------------------------------------------------
template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_descent(const T *__restrict__ z, uint32_t *__restrict__ ptr,
                                                  int w, int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint32_t sk[LH * LW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
  for (int i = threadIdx.x; i < LH * LW; i += NTHR) {
    const int ly = i / LW, lx = i - ly * LW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    uint32_t k = 0xFFFFFFFFu;
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) k = Key32<T>::to(z[(size_t)gy * w + gx]);
    sk[i] = k;
  }
  __syncthreads();
  const int lx = threadIdx.x & (TW - 1), ly0 = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < TH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const uint32_t c = (uint32_t)gy * (uint32_t)w + (uint32_t)gx;
    uint32_t res;
    if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) {
      res = OUTP;
    } else {
      const uint32_t *p = &sk[(ly + 1) * LW + lx + 1];
      const uint32_t kc = p[0];
      // neighbours in increasing cell-index order; strict '<' keeps the lowest index among ties
      uint32_t bk;
      int boff;
      bool lower_idx;
      if (TOPO == 8) {
        bk = p[-LW - 1]; boff = -w - 1; lower_idx = true;
        uint32_t k;
        k = p[-LW];     if (k < bk) { bk = k; boff = -w; }
        k = p[-LW + 1]; if (k < bk) { bk = k; boff = -w + 1; }
        k = p[-1];      if (k < bk) { bk = k; boff = -1; }
        k = p[1];       if (k < bk) { bk = k; boff = 1; lower_idx = false; }
        k = p[LW - 1];  if (k < bk) { bk = k; boff = w - 1; lower_idx = false; }
        k = p[LW];      if (k < bk) { bk = k; boff = w; lower_idx = false; }
        k = p[LW + 1];  if (k < bk) { bk = k; boff = w + 1; lower_idx = false; }
      } else {
        bk = p[-LW]; boff = -w; lower_idx = true;
        uint32_t k;
        k = p[-1]; if (k < bk) { bk = k; boff = -1; }
        k = p[1];  if (k < bk) { bk = k; boff = 1; lower_idx = false; }
        k = p[LW]; if (k < bk) { bk = k; boff = w; lower_idx = false; }
      }
      const bool take = (bk < kc) || (bk == kc && lower_idx);
      res = take ? (uint32_t)((int64_t)c + boff) : c;
    }
    ptr[c] = res;
  }
}

// ------------------------------------------------------------------------------------------
// 2. pointer chasing with path compression (bounded hops per pass; host repeats while flagged).
// Any value ever stored in ptr[c] is an ancestor of c (or OUTP), so concurrent in-place updates
// and stale cached reads are harmless.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_chase(uint32_t *ptr, uint32_t n, int maxhops, uint32_t *flag) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride) {
    const uint32_t c = (uint32_t)c64;
    uint32_t p = ptr[c];
    if (p == OUTP || p == c) continue;
    const uint32_t p0 = p;
    int hops = 0;
    bool unfinished = false;
    for (;;) {
      const uint32_t q = ptr[p];
      if (q == OUTP) { p = OUTP; break; }
      if (q == p) break;
      p = q;
      if (++hops >= maxhops) { unfinished = true; break; }
    }
    if (p != p0) ptr[c] = p;
    if (unfinished) *flag = 1;
  }
}

// ------------------------------------------------------------------------------------------
// 3. dense basin ids for pits (ptr[c] == c), then per-cell labels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_count_pits(const uint32_t *__restrict__ ptr, uint32_t n,
                                                     uint32_t *__restrict__ counts) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint32_t base = blockIdx.x * CELLS_PER_BLOCK;
  uint32_t cnt = 0;
#pragma unroll 4
  for (int j = 0; j < CELLS_PER_BLOCK / NTHR; j++) {
    const uint32_t c = base + j * NTHR + threadIdx.x;
    if (c < n && ptr[c] == c) cnt++;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// single workgroup: exclusive scan of counts[0..m) in place; total -> *total
__global__ __launch_bounds__(1024) void k_scan_counts(uint32_t *counts, uint32_t m, uint32_t *total) {
  __shared__ uint32_t part[1024];
  const uint32_t chunk = (m + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * chunk, hi = min(lo + chunk, m);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan over 1024 partials
    uint32_t v = (threadIdx.x >= (uint32_t)o) ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t v = counts[i];
    counts[i] = run;
    run += v;
  }
  if (threadIdx.x == 1023) *total = part[1023];
}

__global__ __launch_bounds__(NTHR) void k_assign_pits(const uint32_t *__restrict__ ptr, uint32_t n,
                                                      const uint32_t *__restrict__ offsets,
                                                      uint32_t *__restrict__ lab) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint32_t base = blockIdx.x * CELLS_PER_BLOCK;
  uint32_t run = offsets[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < CELLS_PER_BLOCK / NTHR; j++) {
    const uint32_t c = base + j * NTHR + threadIdx.x;
    const bool pit = c < n && ptr[c] == c;
    const unsigned long long bal = __ballot(pit);
    const uint32_t rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) ws[wv] = __popcll(bal);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NTHR / 64; k++) {
      const uint32_t v = ws[k];
      if (k < wv) woff += v;
      tot += v;
    }
    if (pit) lab[c] = run + woff + rank;
    run += tot;
    __syncthreads();
  }
}

// lab[c] = basin id of c's pit; B for cells draining off the raster.
__global__ __launch_bounds__(NTHR) void k_label_cells(const uint32_t *__restrict__ ptr, uint32_t *lab,
                                                      uint32_t n, uint32_t B) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride) {
    const uint32_t c = (uint32_t)c64;
    const uint32_t p = ptr[c];
    if (p == OUTP) lab[c] = B;
    else if (p != c) lab[c] = lab[p];  // lab[p] was written by k_assign_pits (earlier launch)
  }
}

// ------------------------------------------------------------------------------------------
// 4. Boruvka rounds
// tables (B+1 entries, index B = the outside): cur[b]  current root component of basin b
//                                              acc[b]  max pass key on b's path to cur[b]
//                                              best[r] (pass key << 32 | neighbour root), per root
//                                              link[r] (path max key << 32 | parent root), per root
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_init_tables(uint32_t *cur, uint32_t *acc, unsigned long long *link,
                                                      uint32_t *roots, uint32_t B) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i > B) return;
  cur[i] = i;
  acc[i] = 0;
  link[i] = (unsigned long long)i;
  if (i < B) roots[i] = i;
}

__global__ __launch_bounds__(NTHR) void k_best_reset(const uint32_t *__restrict__ roots, uint32_t nroots,
                                                     unsigned long long *best) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i < nroots) best[roots[i]] = ~0ull;
}

template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_scan(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                               const uint32_t *__restrict__ cur, unsigned long long *best,
                                               int w, int h, uint32_t B, uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint32_t sk[LH * LW];
  __shared__ uint32_t sc[LH * LW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
  for (int i = threadIdx.x; i < LH * LW; i += NTHR) {
    const int ly = i / LW, lx = i - ly * LW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    uint32_t k = 0, comp = B;
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
      const size_t g = (size_t)gy * w + gx;
      k = Key32<T>::to(z[g]);
      comp = cur[lab[g]];
    }
    sk[i] = k;
    sc[i] = comp;
  }
  __syncthreads();
  const int lx = threadIdx.x & (TW - 1), ly0 = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < TH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const int o = (ly + 1) * LW + lx + 1;
    const uint32_t C = sc[o];
    if (C == B) continue;  // already drains to the outside (includes every border cell)
    const uint32_t kc = sk[o];
    unsigned long long cand = ~0ull;
#define RD_NB(off)                                                                     \
  {                                                                                    \
    const uint32_t D = sc[o + (off)];                                                  \
    if (D != C) {                                                                      \
      const uint32_t kn = sk[o + (off)];                                               \
      const unsigned long long e = ((unsigned long long)(kn > kc ? kn : kc) << 32) | D; \
      cand = e < cand ? e : cand;                                                      \
    }                                                                                  \
  }
    RD_NB(-LW) RD_NB(-1) RD_NB(1) RD_NB(LW)
    if (TOPO == 8) { RD_NB(-LW - 1) RD_NB(-LW + 1) RD_NB(LW - 1) RD_NB(LW + 1) }
#undef RD_NB
    if (cand != ~0ull) {
      // cheap (possibly stale) pre-check, then the authoritative atomic
      if (cand < best[C]) atomicMin(&best[C], cand);
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_hook(const uint32_t *__restrict__ roots, uint32_t nroots,
                                               const unsigned long long *__restrict__ best,
                                               unsigned long long *link, uint32_t B) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= nroots) return;
  const uint32_t r = roots[i];
  const unsigned long long b = best[r];
  const uint32_t t = (uint32_t)b;
  bool keep_root = false;
  if (b == ~0ull) keep_root = true;  // no neighbouring component (cannot happen on a connected raster)
  else if (t != B) {
    // mutual lowest pass: the pair merges, the smaller id stays root (the pass heights agree
    // because both sides see the same cell pair)
    keep_root = ((uint32_t)best[t] == r) && (r < t);
  }
  link[r] = keep_root ? (unsigned long long)r : b;
}

// pointer jumping over this round's hook forest, carrying the path maximum in the high word.
__global__ __launch_bounds__(NTHR) void k_chase_links(const uint32_t *__restrict__ roots, uint32_t nroots,
                                                      unsigned long long *link, int maxhops, uint32_t *flag) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= nroots) return;
  const uint32_t r = roots[i];
  unsigned long long l = __hip_atomic_load(&link[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t p = (uint32_t)l, m = (uint32_t)(l >> 32);
  if (p == r) return;
  const uint32_t p0 = p;
  int hops = 0;
  bool unfinished = false;
  for (;;) {
    const unsigned long long lp = __hip_atomic_load(&link[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t pp = (uint32_t)lp;
    if (pp == p) break;
    const uint32_t mm = (uint32_t)(lp >> 32);
    p = pp;
    m = mm > m ? mm : m;
    if (++hops >= maxhops) { unfinished = true; break; }
  }
  if (p != p0)
    __hip_atomic_store(&link[r], ((unsigned long long)m << 32) | p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (unfinished) *flag = 1;
}

__global__ __launch_bounds__(NTHR) void k_update_basins(uint32_t *cur, uint32_t *acc,
                                                        const unsigned long long *__restrict__ link, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  const uint32_t c = cur[b];
  if (c == B) return;
  const unsigned long long l = link[c];
  const uint32_t p = (uint32_t)l, m = (uint32_t)(l >> 32);
  if (p != c) {
    cur[b] = p;
    if (m > acc[b]) acc[b] = m;
  }
}

__global__ __launch_bounds__(NTHR) void k_compact_roots(const uint32_t *__restrict__ roots_in, uint32_t nroots,
                                                        const unsigned long long *__restrict__ link,
                                                        uint32_t *roots_out, uint32_t *counter) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  bool keep = false;
  uint32_t r = 0;
  if (i < nroots) {
    r = roots_in[i];
    keep = (uint32_t)link[r] == r;
  }
  const unsigned long long bal = __ballot(keep);
  if (bal == 0) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == (int)__ffsll((long long)bal) - 1) base = atomicAdd(counter, (uint32_t)__popcll(bal));
  base = __shfl(base, (int)__ffsll((long long)bal) - 1, 64);
  if (keep) roots_out[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
}

// ------------------------------------------------------------------------------------------
// 5. finalize
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(NTHR) void k_finalize(T *z, const uint32_t *__restrict__ lab,
                                                   const uint32_t *__restrict__ acc, uint32_t n, uint32_t B) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride) {
    const uint32_t c = (uint32_t)c64;
    const uint32_t b = lab[c];
    if (b == B) continue;
    const uint32_t L = acc[b];
    if (L > Key32<T>::to(z[c])) z[c] = Key32<T>::from(L);
  }
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

template <class T, int TOPO>
static void fill_device_impl(T *d_z, int w, int h, hipStream_t s) {
  const uint64_t n64 = (uint64_t)w * (uint64_t)h;
  if (n64 > 0xFFFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: raster has more than 2^32-65536 cells");
  const uint32_t n = (uint32_t)n64;
  g_stats = rdgpu_fill_stats{n64, 0, 0, 0};
  if (w <= 2 || h <= 2) return;  // every cell is a border cell: nothing can be filled

  Workspace &ws = Workspace::get();
  uint32_t *ptr = ws.buf<uint32_t>("fill.ptr", n);
  uint32_t *lab = ws.buf<uint32_t>("fill.lab", n);
  const uint32_t nblk = cdiv(n, CELLS_PER_BLOCK);
  uint32_t *counts = ws.buf<uint32_t>("fill.counts", nblk);
  uint32_t *dflags = ws.buf<uint32_t>("fill.flags", 16);  // [0] chase flag, [1] pit total, [2] root counter
  uint32_t *hw = ws.host_words();

  const uint32_t tilesX = cdiv(w, TW), tilesY = cdiv(h, TH), ntiles = tilesX * tilesY;
  const uint32_t tgrid = xcd_grid(ntiles);
  const uint32_t sgrid = min(cdiv(n, NTHR), 256u * 32u);  // grid-stride 1-D kernels

  RD_LAUNCH("fill.descent", (k_descent<T, TOPO>), dim3(tgrid), dim3(NTHR), 0, s, d_z, ptr, w, h, tilesX, ntiles);

  // path compression; each pass shortens every path by >= 32x
  for (;;) {
    RD_HIP(hipMemsetAsync(dflags, 0, sizeof(uint32_t), s));
    RD_LAUNCH("fill.chase", k_chase, dim3(sgrid), dim3(NTHR), 0, s, ptr, n, 32, dflags);
    RD_HIP(hipMemcpyAsync(hw, dflags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    g_stats.jump_passes++;
    if (hw[0] == 0) break;
  }

  RD_LAUNCH("fill.count_pits", k_count_pits, dim3(nblk), dim3(NTHR), 0, s, ptr, n, counts);
  RD_LAUNCH("fill.scan_counts", k_scan_counts, dim3(1), dim3(1024), 0, s, counts, nblk, dflags + 1);
  RD_HIP(hipMemcpyAsync(hw, dflags + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_LAUNCH("fill.assign_pits", k_assign_pits, dim3(nblk), dim3(NTHR), 0, s, ptr, n, counts, lab);
  RD_HIP(hipStreamSynchronize(s));
  const uint32_t B = hw[0];
  g_stats.basins = B;
  if (B == 0) return;  // no interior pits: the DEM has no depressions
  RD_LAUNCH("fill.label_cells", k_label_cells, dim3(sgrid), dim3(NTHR), 0, s, ptr, lab, n, B);

  uint32_t *cur = ws.buf<uint32_t>("fill.cur", (size_t)B + 1);
  uint32_t *acc = ws.buf<uint32_t>("fill.acc", (size_t)B + 1);
  unsigned long long *best = ws.buf<unsigned long long>("fill.best", (size_t)B + 1);
  unsigned long long *link = ws.buf<unsigned long long>("fill.link", (size_t)B + 1);
  uint32_t *rootsA = ws.buf<uint32_t>("fill.rootsA", B);
  uint32_t *rootsB = ws.buf<uint32_t>("fill.rootsB", B);
  RD_LAUNCH("fill.init_tables", k_init_tables, dim3(cdiv((uint64_t)B + 1, NTHR)), dim3(NTHR), 0, s, cur, acc, link,
            rootsA, B);

  uint32_t nroots = B;
  while (nroots > 0) {
    const uint32_t rgrid = cdiv(nroots, NTHR);
    RD_LAUNCH("fill.best_reset", k_best_reset, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, best);
    RD_LAUNCH("fill.scan", (k_scan<T, TOPO>), dim3(tgrid), dim3(NTHR), 0, s, d_z, lab, cur, best, w, h, B, tilesX,
              ntiles);
    RD_LAUNCH("fill.hook", k_hook, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, best, link, B);
    for (;;) {
      RD_HIP(hipMemsetAsync(dflags, 0, sizeof(uint32_t), s));
      RD_LAUNCH("fill.chase_links", k_chase_links, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, link, 32, dflags);
      RD_HIP(hipMemcpyAsync(hw, dflags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      if (hw[0] == 0) break;
    }
    RD_LAUNCH("fill.update_basins", k_update_basins, dim3(cdiv(B, NTHR)), dim3(NTHR), 0, s, cur, acc, link, B);
    RD_HIP(hipMemsetAsync(dflags + 2, 0, sizeof(uint32_t), s));
    RD_LAUNCH("fill.compact_roots", k_compact_roots, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, link, rootsB,
              dflags + 2);
    RD_HIP(hipMemcpyAsync(hw, dflags + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    const uint32_t next = hw[0];
    if (next >= nroots) throw Error(RDGPU_ERR_HIP, "rdgpu_fill: contraction made no progress (internal error)");
    nroots = next;
    std::swap(rootsA, rootsB);
    g_stats.rounds++;
  }

  RD_LAUNCH("fill.finalize", (k_finalize<T>), dim3(sgrid), dim3(NTHR), 0, s, d_z, lab, acc, n, B);
}

template <class T>
static void fill_device(T *d_z, int w, int h, int topology, hipStream_t s) {
  if (!d_z) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: width and height must be positive");
  if (topology == 8) fill_device_impl<T, 8>(d_z, w, h, s);
  else if (topology == 4) fill_device_impl<T, 4>(d_z, w, h, s);
  else throw Error(RDGPU_ERR_ARG, "rdgpu_fill: topology must be 8 or 4");  // depressions.hpp:19-20
}

template <class T>
static void fill_host(T *dem, int w, int h, int topology) {
  if (!dem) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: width and height must be positive");
  const size_t bytes = (size_t)w * h * sizeof(T);
  T *d = Workspace::get().buf<T>("host.dem", (size_t)w * h);
  RD_HIP(hipMemcpy(d, dem, bytes, hipMemcpyHostToDevice));
  fill_device<T>(d, w, h, topology, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, bytes, hipMemcpyDeviceToHost));
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FILL_API(SUF, T)                                                                       \
  extern "C" int rdgpu_fill_##SUF(T *dem, int w, int h, int topology) {                           \
    return guarded([&] { fill_host<T>(dem, w, h, topology); });                                   \
  }                                                                                               \
  extern "C" int rdgpu_fill_dev_##SUF(T *d_dem, int w, int h, int topology, void *stream) {       \
    return guarded([&] { fill_device<T>(d_dem, w, h, topology, (hipStream_t)stream); });          \
  }
RD_FILL_API(u8, uint8_t)
RD_FILL_API(i16, int16_t)
RD_FILL_API(u16, uint16_t)
RD_FILL_API(i32, int32_t)
RD_FILL_API(u32, uint32_t)
RD_FILL_API(f32, float)

extern "C" int rdgpu_fill_get_stats(rdgpu_fill_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = g_stats;
  return RDGPU_OK;
}
