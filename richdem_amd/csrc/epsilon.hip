// epsilon.hip -- PriorityFloodEpsilon_Barnes2014<topo> on MI355X.
//
// Replaces richdem::PriorityFloodEpsilon_Barnes2014 (reference include/richdem/depressions/Barnes2014.hpp:335-420;
// FillDepressionsEpsilon, depressions/depressions.hpp:23; Python rd.FillDepressions(epsilon=True) ->
// rdPFepsilonD8/D4, wrappers/pyrichdem/src/pywrapper.hpp:34-35).  Floating-point DEMs only, as in the reference
// (:424-451 throw for the integer types).
//
// The reference is a serial sweep: a min-heap of cells at their own elevation plus a FIFO of raised cells; a cell is
// closed by the first neighbour processed, c, and becomes max(z, nextafter(c.z, +inf)).  Cells are processed in
// non-decreasing order of their final value (the FIFO's front grows one representable step at a time and the heap
// is consulted whenever the two agree, :375-388), so the result is the Dijkstra-type surface
//
//        E(c) = z(c)                                   on the raster border and on NoData cells (never altered)
//        E(c) = max( z(c), nextafter( min over neighbours n of E(n) ) )       elsewhere,
//
// which has exactly one solution (every step strictly increases).  That surface is what this file computes, in
// parallel, and it IS the reference's output whenever no two cells of the reference's heap hold the same elevation.
// With such ties the reference's own output depends on the order in which std::priority_queue returns them (the cell
// popped first floods its whole depression before the other is looked at); the fixed point is then a lower bound of
// it, cell by cell.  NoData cells act as the reference treats them when they are connected to the raster border
// (processed first, at the NoData value): fixed cells of value NoData; tests/test_epsilon_gpu.py pins all of this
// against the compiled reference.
//
// Method: order-preserving CONTIGUOUS integer keys (nextafter == key + 1, -0.0 and +0.0 share a key) and Bellman-Ford
// relaxation FROM ABOVE: any start d0 >= E relaxes monotonically to E under d <- min(d, max(z, min_n d(n) + 1)).
//   * start: the plain filled surface W (the fill engine of fill.hip) is a lower bound of E.  A cell c keeps its own
//     elevation -- which is then exact -- if it has a neighbour n with W(n) + X < z(c), X being an assumed bound on
//     how far the epsilon gradient lifts any cell above W (a fixed neighbour -- border, NoData -- needs no slack).
//     Everything else (lakes, flats, gentle slopes) starts at +inf.
//   * relaxation: tiles in LDS/registers with an active-tile work list exactly as the flat resolution does it
//     (csrc/flats.hip): an active 64x32 tile is relaxed to its local fixed point, tiles across a changed edge are
//     activated for the next round, until no tile is active.
//   * proof: one stencil pass checks d = max(z, min_n d(n) + 1) on every cell.  The equation has one solution, so a
//     pass without complaint proves the result whatever X was; a complaint means X was too small for this DEM (some
//     lake's gradient spills further over its shore than assumed -- the reference's "false pit cells", :405-406) and
//     the relaxation is repeated with a larger X (the failed attempt is a lower bound of E, so it tells how large).
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

extern "C" int rdgpu_fill_dev_f32(float *, int, int, int, void *);
extern "C" int rdgpu_fill_dev_f64(double *, int, int, int, void *);

namespace rdgpu {

namespace eps {

constexpr int NT = 256;
constexpr int CW = 64, RCH = 32, RBANDS = NT / 64, ROWS = RCH / RBANDS;
constexpr int RW = CW + 2, RH = RCH + 2;
constexpr int HSTEPS = 4;
constexpr int BATCH = 8;

// ---- contiguous order-preserving keys -------------------------------------------------------------------------
template <class T>
struct CKey;
template <>
struct CKey<float> {
  using K = uint32_t;
  static constexpr K INF = 0xFFFFFFF0u;        // "not reached"
  static constexpr K POSINF = 0xFF7FFFFFu;     // key of +infinity: nextafter(+inf) == +inf
  __host__ __device__ static inline K to(float v) {
    uint32_t b = __builtin_bit_cast(uint32_t, v);
    if (b == 0x80000000u) b = 0;
    const uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return k >= 0x80000000u ? k - 1u : k;      // 0x7FFFFFFF (the key -0.0 would have had) is closed up
  }
  __host__ __device__ static inline float from(K c) {
    const uint32_t k = c >= 0x7FFFFFFFu ? c + 1u : c;
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
  }
};
template <>
struct CKey<double> {
  using K = uint64_t;
  static constexpr K INF = 0xFFFFFFFFFFFFFFF0ull;
  static constexpr K POSINF = 0xFFEFFFFFFFFFFFFFull;
  __host__ __device__ static inline K to(double v) {
    uint64_t b = __builtin_bit_cast(uint64_t, v);
    if (b == 0x8000000000000000ull) b = 0;
    const uint64_t k = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    return k >= 0x8000000000000000ull ? k - 1ull : k;
  }
  __host__ __device__ static inline double from(K c) {
    const uint64_t k = c >= 0x7FFFFFFFFFFFFFFFull ? c + 1ull : c;
    const uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __builtin_bit_cast(double, b);
  }
};

// one representable step up (saturating at +infinity; "not reached" stays not reached)
template <class T>
__device__ __forceinline__ typename CKey<T>::K step_up(typename CKey<T>::K m) {
  return m + (m < CKey<T>::POSINF ? 1 : 0);
}

__device__ __forceinline__ uint32_t dpp_left(uint32_t v, uint32_t fill) {   // lane l receives lane l-1's value
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_right(uint32_t v, uint32_t fill) {  // lane l receives lane l+1's value
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t dpp_left(uint64_t v, uint64_t fill) {
  return ((uint64_t)dpp_left((uint32_t)(v >> 32), (uint32_t)(fill >> 32)) << 32) | dpp_left((uint32_t)v, (uint32_t)fill);
}
__device__ __forceinline__ uint64_t dpp_right(uint64_t v, uint64_t fill) {
  return ((uint64_t)dpp_right((uint32_t)(v >> 32), (uint32_t)(fill >> 32)) << 32) | dpp_right((uint32_t)v, (uint32_t)fill);
}
template <class K>
__device__ __forceinline__ K kmin(K a, K b) { return a < b ? a : b; }
template <class K>
__device__ __forceinline__ K kmax(K a, K b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t block_append(bool pred, uint32_t *counter) {
  __shared__ uint32_t wcnt[NT / 64];
  __shared__ uint32_t bbase;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(pred);
  if (lane == 0) wcnt[wv] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    bbase = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  uint32_t off = bbase;
  for (int k = 0; k < wv; k++) off += wcnt[k];
  return off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(NT) void k_tiles_compact(uint8_t *flags, uint32_t ntiles, uint32_t *list, uint32_t *count) {
  const uint32_t i = blockIdx.x * NT + threadIdx.x;
  const bool hit = i < ntiles && flags[i] != 0;
  if (hit) flags[i] = 0;
  const uint32_t slot = block_append(hit, count);
  if (hit) list[slot] = i;
}

// ---- start values --------------------------------------------------------------------------------------------
// D(c) = key(z) for the fixed cells (raster border, NoData) and for the cells with a neighbour n whose final value is
// certainly below z(c): a fixed neighbour lower than c, or any neighbour with W(n) + X < z(c); +inf for all others.
// One 64x32 tile per block; a tile that holds a +inf cell is active in round 1.
template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_eps_init(const T *__restrict__ z, const T *__restrict__ W, T nodata,
                                                 typename CKey<T>::K X, typename CKey<T>::K *__restrict__ D,
                                                 uint8_t *tile_active, int w, int h, uint32_t tilesX, uint32_t ntiles) {
  using K = typename CKey<T>::K;
  __shared__ K sw[RH * RW];   // upper bound of the neighbour's final value
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * CW, y0 = (int)(t / tilesX) * RCH;
  for (int i = threadIdx.x; i < RH * RW; i += NT) {
    const int ly = i / RW, lx = i - ly * RW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    K v = CKey<T>::INF;
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
      const size_t g = (size_t)gy * w + gx;
      const T zz = z[g];
      if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || zz == nodata) v = CKey<T>::to(zz);   // fixed: exact
      else {
        const K wk = CKey<T>::to(W[g]);
        v = (wk <= CKey<T>::POSINF && CKey<T>::POSINF - wk > X) ? wk + X : CKey<T>::INF;
      }
    }
    sw[i] = v;
  }
  __syncthreads();
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  int anyinf = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    const T zz = z[g];
    const K zk = CKey<T>::to(zz);
    K d = zk;
    if (!(gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || zz == nodata)) {
      const int o = (ly + 1) * RW + lx + 1;
      K lo = kmin(kmin(sw[o - RW], sw[o + RW]), kmin(sw[o - 1], sw[o + 1]));
      if (TOPO == 8) lo = kmin(lo, kmin(kmin(sw[o - RW - 1], sw[o - RW + 1]), kmin(sw[o + RW - 1], sw[o + RW + 1])));
      if (!(lo < zk)) { d = CKey<T>::INF; anyinf = 1; }
    }
    D[g] = d;
  }
  if (__syncthreads_or(anyinf) && threadIdx.x == 0) tile_active[t] = 1;
}

// ---- the proof: d == max(z, min_n d(n) + 1) on every relaxed cell -----------------------------------------------------
// out[0] = number of cells where it fails (0: D is the unique fixed point), out[1..2] = the largest d - key(W) seen
// (how far the gradient lifted a cell above the plain fill; a lower bound of the true figure when the proof fails).
template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_eps_check(const T *__restrict__ z, const T *__restrict__ W, T nodata,
                                                  const typename CKey<T>::K *__restrict__ D, int w, int h, uint32_t tilesX,
                                                  uint32_t ntiles, uint32_t *bad, unsigned long long *maxlift) {
  using K = typename CKey<T>::K;
  __shared__ K sd[RH * RW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * CW, y0 = (int)(t / tilesX) * RCH;
  for (int i = threadIdx.x; i < RH * RW; i += NT) {
    const int ly = i / RW, lx = i - ly * RW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    sd[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? D[(size_t)gy * w + gx] : CKey<T>::INF;
  }
  __syncthreads();
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  uint32_t nbad = 0;
  unsigned long long lift = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    const T zz = z[g];
    if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || zz == nodata) continue;
    const int o = (ly + 1) * RW + lx + 1;
    K lo = kmin(kmin(sd[o - RW], sd[o + RW]), kmin(sd[o - 1], sd[o + 1]));
    if (TOPO == 8) lo = kmin(lo, kmin(kmin(sd[o - RW - 1], sd[o - RW + 1]), kmin(sd[o + RW - 1], sd[o + RW + 1])));
    const K f = kmax(CKey<T>::to(zz), step_up<T>(lo));
    const K d = sd[o];
    if (d != f) nbad++;
    const K wk = CKey<T>::to(W[g]);
    if (d < CKey<T>::INF && d > wk) lift = lift > (unsigned long long)(d - wk) ? lift : (unsigned long long)(d - wk);
  }
  for (int o = 32; o > 0; o >>= 1) {
    nbad += __shfl_down(nbad, o, 64);
    const unsigned long long other = __shfl_down(lift, o, 64);
    lift = lift > other ? lift : other;
  }
  if ((threadIdx.x & 63) == 0) {
    if (nbad) atomicAdd(bad, nbad);
    if (lift && lift > __hip_atomic_load(maxlift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlift, lift);
  }
}

// ---- one active tile to its local fixed point ----------------------------------------------------------------------
template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_eps_relax(const T *__restrict__ z, T nodata, typename CKey<T>::K *D,
                                                  const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ count,
                                                  uint8_t *next_active, int w, int h, uint32_t tilesX, uint32_t tilesY) {
  using K = typename CKey<T>::K;
  constexpr K KINF = CKey<T>::INF;
  __shared__ K sd[RH * RW];
  __shared__ K xrow[2][RBANDS][2][CW];
  const uint32_t nact = *count;
  // tiles past the grid (the list grew faster than the host expected) simply stay active for the next round
  for (uint32_t i = gridDim.x + blockIdx.x * NT + threadIdx.x; i < nact; i += gridDim.x * NT) next_active[tiles[i]] = 1;
  if (blockIdx.x >= nact) return;
  const uint32_t t = tiles[blockIdx.x];
  const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
  const int x0 = tx * CW, y0 = ty * RCH;
  {
    // the distances of the tile and of its one-cell ring, all loads in flight together; cells outside the raster
    // read as "not reached"
    constexpr int IPT = (RH * RW + NT - 1) / NT;
    K dv[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = min((int)threadIdx.x + r * NT, RH * RW - 1);
      const int ly = i / RW, lxx = i - ly * RW;
      const int gx = min(max(x0 - 1 + lxx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      dv[r] = __hip_atomic_load(&D[(size_t)gy * w + gx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NT;
      if (i >= RH * RW) continue;
      const int ly = i / RW, lxx = i - ly * RW;
      const int gx = x0 - 1 + lxx, gy = y0 - 1 + ly;
      sd[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? dv[r] : KINF;
    }
  }
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  const int gx = x0 + lx;
  K zk[ROWS];
  uint32_t free_ = 0;   // bit j: the cell is relaxed (inside the raster, not on its border, not NoData)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int gy = y0 + band * ROWS + j;
    zk[j] = 0;
    if (gx < w && gy < h) {
      const T zz = z[(size_t)gy * w + gx];
      zk[j] = CKey<T>::to(zz);
      if (!(gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || zz == nodata)) free_ |= 1u << j;
    }
  }
  __syncthreads();
  K d[ROWS], d0[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; j++) d0[j] = d[j] = sd[(band * ROWS + j + 1) * RW + lx + 1];
  auto ring = [&](int ly /* -1..RCH */, int cx /* -1..CW */) -> K { return sd[(ly + 1) * RW + cx + 1]; };
  K sideL[ROWS], sideR[ROWS];   // what lanes 0 / 63 see in the halo columns (D8: vertical 3-minima; D4: the one cell)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    if (TOPO == 8) {
      sideL[j] = lx == 0 ? kmin(ring(ly - 1, -1), kmin(ring(ly, -1), ring(ly + 1, -1))) : KINF;
      sideR[j] = lx == CW - 1 ? kmin(ring(ly - 1, CW), kmin(ring(ly, CW), ring(ly + 1, CW))) : KINF;
    } else {
      sideL[j] = lx == 0 ? ring(ly, -1) : KINF;
      sideR[j] = lx == CW - 1 ? ring(ly, CW) : KINF;
    }
  }
  const K halo_up = band == 0 ? ring(-1, lx) : KINF, halo_dn = band == RBANDS - 1 ? ring(RCH, lx) : KINF;
  // D8 corner contributions of the rows above / below the band come through the neighbouring lanes' 3-minima, which
  // include `up` / `dn` of THOSE lanes; lanes 0 and 63 add the ring corners through sideL / sideR (rows ly-1..ly+1).

  // smallest value among the neighbours of cell j (D8: 8, D4: 4), given the rows above / below the band
  auto neigh_min = [&](const K up, const K dn, K out[ROWS]) {
    K own[ROWS];   // vertical neighbours in the own column
#pragma unroll
    for (int j = 0; j < ROWS; j++) own[j] = kmin(j ? d[j - 1] : up, j + 1 < ROWS ? d[j + 1] : dn);
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const K col = TOPO == 8 ? kmin(own[j], d[j]) : d[j];   // what the side lanes need from this column
      const K side = kmin(dpp_left(col, sideL[j]), dpp_right(col, sideR[j]));
      out[j] = kmin(own[j], side);
    }
  };

  int changed = 1, it = 0;
  constexpr int IT_CAP = 512;
  for (; it < IT_CAP; it++) {
    // Gauss-Seidel along the strip (vertical neighbours: both topologies)
#pragma unroll
    for (int j = 1; j < ROWS; j++)
      if (free_ & (1u << j)) d[j] = kmin(d[j], kmax(zk[j], step_up<T>(d[j - 1])));
#pragma unroll
    for (int j = ROWS - 2; j >= 0; j--)
      if (free_ & (1u << j)) d[j] = kmin(d[j], kmax(zk[j], step_up<T>(d[j + 1])));
    xrow[it & 1][band][0][lx] = d[0];
    xrow[it & 1][band][1][lx] = d[ROWS - 1];
    if (!__syncthreads_or(changed)) break;
    const K up = band == 0 ? halo_up : xrow[it & 1][band - 1][1][lx];
    const K dn = band == RBANDS - 1 ? halo_dn : xrow[it & 1][band + 1][0][lx];
    // HSTEPS stencil steps per barrier: the sideways exchange is all DPP (registers); the rows above / below the band
    // are one trip stale, which only delays convergence (values are upper bounds and only decrease)
    changed = 0;
#pragma unroll
    for (int sub = 0; sub < HSTEPS; sub++) {
      K mn[ROWS];
      neigh_min(up, dn, mn);
#pragma unroll
      for (int j = 0; j < ROWS; j++) {
        const K cand = kmax(zk[j], step_up<T>(mn[j]));
        if ((free_ & (1u << j)) && cand < d[j]) { d[j] = cand; changed = 1; }
      }
    }
  }
  if (it == IT_CAP && threadIdx.x == 0) next_active[t] = 1;   // iteration cap hit: finish this tile next round
  // write back and wake the tiles across every edge that changed
  int top = 0, bot = 0, lef = 0, rig = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    if (!(free_ & (1u << j)) || d[j] >= d0[j]) continue;
    const int ly = band * ROWS + j;
    __hip_atomic_store(&D[(size_t)(y0 + ly) * w + gx], d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    top |= ly == 0; bot |= ly == RCH - 1; lef |= lx == 0; rig |= lx == CW - 1;
  }
  top = __syncthreads_or(top); bot = __syncthreads_or(bot); lef = __syncthreads_or(lef); rig = __syncthreads_or(rig);
  if (threadIdx.x < 9 && threadIdx.x != 4) {
    const int dx = (int)threadIdx.x % 3 - 1, dy = (int)threadIdx.x / 3 - 1;
    const bool need = (dy < 0 ? top : dy > 0 ? bot : 1) && (dx < 0 ? lef : dx > 0 ? rig : 1) && (top | bot | lef | rig);
    const int ntx = tx + dx, nty = ty + dy;
    if (need && ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) next_active[nty * tilesX + ntx] = 1;
  }
}

// ---- E = decode(D) where it lies above z ------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(NT) void k_eps_final(T *z, T nodata, const typename CKey<T>::K *__restrict__ D, uint64_t n,
                                                  uint32_t *unreached) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const T zz = z[c];
    if (zz == nodata) continue;
    const typename CKey<T>::K d = D[c];
    if (d >= CKey<T>::INF) { *unreached = 1; continue; }   // cannot happen on a connected raster
    if (d > CKey<T>::to(zz)) z[c] = CKey<T>::from(d);
  }
}

struct Stats {
  uint32_t rounds = 0, attempts = 0;
  uint64_t tile_relaxations = 0, slack = 0, max_lift = 0;
};
static Stats g_stats;

template <class T, int TOPO>
static void run(T *d_z, T nodata, const T *d_W, int w, int h, hipStream_t s) {
  using K = typename CKey<T>::K;
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  const uint32_t tilesX = (w + CW - 1) / CW, tilesY = (h + RCH - 1) / RCH, ntiles = tilesX * tilesY;
  K *D = ws.buf<K>("eps.D", n);
  uint8_t *tflags = ws.buf<uint8_t>("eps.tflags", ntiles);
  uint32_t *tlist = ws.buf<uint32_t>("eps.tlist", ntiles);
  uint32_t *ctr = ws.buf<uint32_t>("eps.ctr", BATCH + 8);   // [BATCH] proof failures, [BATCH+2..3] max lift (u64), [BATCH+4] unreached
  g_stats = Stats();
  // assumed bound on the lift above the plain fill, in representable steps (RDGPU_EPS_SLACK overrides: tests)
  const char *env = getenv("RDGPU_EPS_SLACK");
  K X = env ? (K)strtoull(env, nullptr, 10) : (K)(1u << 15);
  for (;;) {
    g_stats.attempts++;
    g_stats.slack = (uint64_t)X;
    RD_HIP(hipMemsetAsync(tflags, 0, ntiles, s));
    RD_HIP(hipMemsetAsync(ctr + BATCH, 0, 8 * sizeof(uint32_t), s));
    RD_LAUNCH("eps.init", (k_eps_init<T, TOPO>), dim3(xcd_grid(ntiles)), dim3(NT), 0, s, (const T *)d_z, d_W, nodata, X, D, tflags,
              w, h, tilesX, ntiles);
    uint32_t grid = ntiles;
    for (bool done = false; !done;) {
      RD_HIP(hipMemsetAsync(ctr, 0, BATCH * sizeof(uint32_t), s));
      for (int b = 0; b < BATCH; b++) {
        RD_LAUNCH("eps.tiles_compact", k_tiles_compact, dim3((ntiles + NT - 1) / NT), dim3(NT), 0, s, tflags, ntiles, tlist, ctr + b);
        RD_LAUNCH("eps.relax", (k_eps_relax<T, TOPO>), dim3(grid), dim3(NT), 0, s, (const T *)d_z, nodata, D, (const uint32_t *)tlist,
                  (const uint32_t *)(ctr + b), tflags, w, h, tilesX, tilesY);
      }
      RD_HIP(hipMemcpyAsync(hw, ctr, BATCH * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      uint32_t most = 0;
      for (int b = 0; b < BATCH; b++) {
        if (hw[b] == 0) { done = true; break; }
        most = std::max(most, hw[b]);
        g_stats.rounds++;
        g_stats.tile_relaxations += hw[b];
      }
      grid = std::min<uint32_t>(ntiles, std::max<uint32_t>(1024u, 2u * most));
      if (g_stats.rounds > (1u << 24)) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_epsilon: relaxation did not terminate");
    }
    RD_LAUNCH("eps.check", (k_eps_check<T, TOPO>), dim3(xcd_grid(ntiles)), dim3(NT), 0, s, (const T *)d_z, d_W, nodata, (const K *)D, w,
              h, tilesX, ntiles, ctr + BATCH, (unsigned long long *)(ctr + BATCH + 2));
    RD_HIP(hipMemcpyAsync(hw, ctr + BATCH, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    const uint64_t lift = (uint64_t)hw[2] | ((uint64_t)hw[3] << 32);
    g_stats.max_lift = lift;
    if (hw[0] == 0) break;   // D = F(D) on every cell: the unique fixed point
    if (X >= CKey<T>::POSINF) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_epsilon: fixed point not reached (internal error)");
    // too small: the failed attempt is a lower bound of E, so the true lift is at least `lift`
    const uint64_t next = std::max<uint64_t>((uint64_t)X * 8, lift * 2 + 16);
    X = next >= (uint64_t)CKey<T>::POSINF ? CKey<T>::POSINF : (K)next;   // POSINF: every interior cell starts at +inf
  }
  RD_LAUNCH("eps.final", (k_eps_final<T>), dim3((uint32_t)std::min<uint64_t>((n + NT - 1) / NT, 256u * 32u)), dim3(NT), 0, s, d_z,
            nodata, (const K *)D, n, ctr + BATCH + 4);
  RD_HIP(hipMemcpyAsync(hw, ctr + BATCH + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  if (hw[0]) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_epsilon: a cell was not reached from the raster border (internal error)");
}

static int fill_plain(float *p, int w, int h, int topo, hipStream_t s) { return rdgpu_fill_dev_f32(p, w, h, topo, s); }
static int fill_plain(double *p, int w, int h, int topo, hipStream_t s) { return rdgpu_fill_dev_f64(p, w, h, topo, s); }

template <class T>
static void fill_epsilon_device(T *d_z, T nodata, int w, int h, int topology, hipStream_t s) {
  if (!d_z) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: width and height must be positive");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: topology must be 8 or 4");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: raster too large");
  if (w <= 2 || h <= 2) return;   // every cell is a border cell
  const size_t n = (size_t)w * h;
  T *W = Workspace::get().buf<T>("eps.W", n);
  RD_HIP(hipMemcpyAsync(W, d_z, n * sizeof(T), hipMemcpyDeviceToDevice, s));
  if (fill_plain(W, w, h, topology, s) != RDGPU_OK) throw Error(RDGPU_ERR_HIP, std::string("rdgpu_fill_epsilon: ") + rdgpu_last_error());
  if (topology == 8) run<T, 8>(d_z, nodata, W, w, h, s);
  else run<T, 4>(d_z, nodata, W, w, h, s);
}

template <class T>
static void fill_epsilon_host(T *dem, T nodata, int w, int h, int topology) {
  if (!dem) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_epsilon: width and height must be positive");
  const size_t bytes = (size_t)w * h * sizeof(T);
  T *d = Workspace::get().buf<T>("host.dem", (size_t)w * h);
  RD_HIP(hipMemcpy(d, dem, bytes, hipMemcpyHostToDevice));
  fill_epsilon_device<T>(d, nodata, w, h, topology, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, bytes, hipMemcpyDeviceToHost));
}

}  // namespace eps
}  // namespace rdgpu

using namespace rdgpu;

extern "C" int rdgpu_fill_epsilon_f32(float *dem, float nodata, int w, int h, int topology) {
  return guarded([&] { eps::fill_epsilon_host<float>(dem, nodata, w, h, topology); });
}
extern "C" int rdgpu_fill_epsilon_f64(double *dem, double nodata, int w, int h, int topology) {
  return guarded([&] { eps::fill_epsilon_host<double>(dem, nodata, w, h, topology); });
}
extern "C" int rdgpu_fill_epsilon_dev_f32(float *d_dem, float nodata, int w, int h, int topology, void *stream) {
  return guarded([&] { eps::fill_epsilon_device<float>(d_dem, nodata, w, h, topology, (hipStream_t)stream); });
}
extern "C" int rdgpu_fill_epsilon_dev_f64(double *d_dem, double nodata, int w, int h, int topology, void *stream) {
  return guarded([&] { eps::fill_epsilon_device<double>(d_dem, nodata, w, h, topology, (hipStream_t)stream); });
}
extern "C" int rdgpu_fill_epsilon_get_stats(rdgpu_epsilon_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  out->rounds = eps::g_stats.rounds;
  out->attempts = eps::g_stats.attempts;
  out->tile_relaxations = eps::g_stats.tile_relaxations;
  out->slack = eps::g_stats.slack;
  out->max_lift = eps::g_stats.max_lift;
  return RDGPU_OK;
}
