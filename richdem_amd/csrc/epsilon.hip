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

#include <vector>

#include <algorithm>
#include <cstdlib>

extern "C" int rdgpu_fill_dev_f32(float *, int, int, int, void *);
extern "C" int rdgpu_fill_dev_f64(double *, int, int, int, void *);

namespace rdgpu {

namespace eps {

constexpr int NT = 256;
constexpr int CW = 64, RCH = 32, RBANDS = NT / 64, ROWS = RCH / RBANDS;
constexpr int RW = CW + 2, RH = RCH + 2;
constexpr int HSTEPS = 8;
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

// The field a relaxation works on: key type, "not reached", the largest key that still steps, and per cell its floor
// (the value it can never go below) and whether it is fixed besides the raster border.
// NoData.  The reference processes a NoData region that touches the raster border first (its border cells are the
// lowest seeds, their NoData neighbours follow through the pit queue) and pushes the data cells next to it into the heap
// at their own elevation: such NoData cells are FIXED cells of value NoData here.  An interior NoData hole is only
// reached when the flood arrives at its ring, and then passes the flood on (Barnes2014.hpp:399-400: it enters the pit
// queue like a lake cell): here such a cell takes part in the relaxation with floor -infinity -- the level travels
// through it, one step per cell -- and is never written.  (Fixed at NoData it would drain every lake around a hole.)
// Which is which: the plain fill W raises an interior hole above NoData and leaves a border-connected region at NoData.
// What is NOT reproduced: the reference leaves those ring cells of a hole at their own elevation that a hole cell
// happens to close before the lake's breadth-first front does -- pits inside a filled lake, and an artefact of its
// queue order; here every ring cell is raised with the lake.
template <class T>
__device__ __forceinline__ bool nodata_is_fixed(T zz, T nodata, T wv) { return zz == nodata && wv == zz; }

template <class T>
struct EpsField {   // the epsilon surface of a float / double DEM: floor = the cell's own elevation
  using K = typename CKey<T>::K;
  static constexpr K INF = CKey<T>::INF, POSINF = CKey<T>::POSINF;
  const T *z;
  T nodata;
  const T *W;
  __device__ __forceinline__ K key(size_t g, bool &fixed) const {
    const T zz = z[g];
    fixed = nodata_is_fixed(zz, nodata, W[g]);
    return zz == nodata && !fixed ? (K)0 : CKey<T>::to(zz);   // (an interior hole: floor -infinity)
  }
};
template <class T>
struct ShedField {   // (level of the plain fill W, steps from where the flood entered the cell's lake or flat): watersheds
  using K = uint64_t;
  static constexpr K INF = 0xFFFFFFFFFFFFFFF0ull, POSINF = 0xFFFFFFFFFFFFFF00ull;
  const T *W;
  __device__ __forceinline__ K key(size_t g, bool &fixed) const {
    fixed = false;
    return (uint64_t)Key32<T>::to(W[g]) << 32;
  }
};

// one representable step up (saturating at +infinity; "not reached" stays not reached)
template <class P>
__device__ __forceinline__ typename P::K step_up(typename P::K m) {
  return m + (m < P::POSINF ? 1 : 0);
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
      if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || nodata_is_fixed(zz, nodata, W[g])) v = CKey<T>::to(zz);   // fixed: exact
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
    const bool border = gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1;
    const bool hole = zz == nodata && !border && !nodata_is_fixed(zz, nodata, W[g]);   // interior NoData: floor -infinity
    const K zk = hole ? (K)0 : CKey<T>::to(zz);
    K d = zk;
    if (!(border || (zz == nodata && !hole))) {
      const int o = (ly + 1) * RW + lx + 1;
      K lo = kmin(kmin(sw[o - RW], sw[o + RW]), kmin(sw[o - 1], sw[o + 1]));
      if (TOPO == 8) lo = kmin(lo, kmin(kmin(sw[o - RW - 1], sw[o - RW + 1]), kmin(sw[o + RW - 1], sw[o + RW + 1])));
      if (!(lo < zk)) { d = CKey<T>::INF; anyinf = 1; }
    }
    D[g] = d;
  }
  if (__syncthreads_or(anyinf) && threadIdx.x == 0) tile_active[t] = 1;
}

constexpr uint32_t TIE_STRIPES = 1024;   // counters of the tie detector (source count, tie count per stripe)
template <class K> struct KeyEmpty { static constexpr K v = (K) ~(K)0; };
__device__ __forceinline__ uint32_t cas_key(uint32_t *p, uint32_t cmp, uint32_t v) { return atomicCAS(p, cmp, v); }
__device__ __forceinline__ uint64_t cas_key(uint64_t *p, uint64_t cmp, uint64_t v) {
  return (uint64_t)atomicCAS(reinterpret_cast<unsigned long long *>(p), (unsigned long long)cmp, (unsigned long long)v);
}

// ---- the proof: d == max(z, min_n d(n) + 1) on every relaxed cell -----------------------------------------------------
// out[0] = number of cells where it fails (0: D is the unique fixed point), out[1..2] = the largest d - key(W) seen
// (how far the gradient lifted a cell above the plain fill; a lower bound of the true figure when the proof fails).
template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_eps_check(const T *__restrict__ z, const T *__restrict__ W, T nodata,
                                                  const typename CKey<T>::K *__restrict__ D, int w, int h, uint32_t tilesX,
                                                  uint32_t ntiles, uint32_t *bad, unsigned long long *maxlift,
                                                  typename CKey<T>::K *tie_table, unsigned long long tie_mask,
                                                  unsigned long long *tie_counts /* [0] sources, [1] ties */) {
  using K = typename CKey<T>::K;
  __shared__ K sd[RH * RW];
  __shared__ uint8_t sr[RH * RW];   // the cell is raised (a data cell above its own elevation): the tie detector's stencil
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * CW, y0 = (int)(t / tilesX) * RCH;
  for (int i = threadIdx.x; i < RH * RW; i += NT) {
    const int ly = i / RW, lx = i - ly * RW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const K dv = in ? D[(size_t)gy * w + gx] : CKey<T>::INF;
    sd[i] = dv;
    uint8_t r = 0;
    if (tie_table && in) {
      const T zz = z[(size_t)gy * w + gx];
      r = (zz != nodata && dv > CKey<T>::to(zz)) ? 1 : 0;
    }
    sr[i] = r;
  }
  __syncthreads();
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  uint32_t nbad = 0, nsrc = 0, nties = 0;
  unsigned long long lift = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    const T zz = z[g];
    const int o = (ly + 1) * RW + lx + 1;
    // Tie detector, on the way: a SOURCE of a gradient is a data cell at its own elevation with a raised neighbour; its
    // key goes into an open-addressing set, and a key that is already there is a tie (see the comment at Stats below).
    if (tie_table && zz != nodata && sd[o] == CKey<T>::to(zz)) {
      int feeds = sr[o - RW] | sr[o + RW] | sr[o - 1] | sr[o + 1];
      if (TOPO == 8) feeds |= sr[o - RW - 1] | sr[o - RW + 1] | sr[o + RW - 1] | sr[o + RW + 1];
      if (feeds) {
        nsrc++;
        const K kc = sd[o];
        if (sizeof(K) == 4 && tie_mask == 0ull) {
          // 32-bit keys, large rasters (tie_mask == 0): one bit per possible key (512 MB), indexed by the key itself -- the sources of a tile lie within
          // a few metres of each other, so their bits share cache lines (a scattering hash table cost 39 ms at S3,
          // a random DRAM access per source; this costs a few)
          uint32_t *bits = reinterpret_cast<uint32_t *>(tie_table);
          const uint32_t bit = 1u << ((uint32_t)kc & 31u);
          if (atomicOr(&bits[(uint32_t)kc >> 5], bit) & bit) nties++;
        } else {
          unsigned long long slot = ((unsigned long long)kc * 0x9E3779B97F4A7C15ull >> 20) & tie_mask;
          for (int probe = 0; probe < 4096; probe++) {   // (bounded: a table that fills up is reported through the source count)
            K v = tie_table[slot];
            if (v == KeyEmpty<K>::v) v = cas_key(&tie_table[slot], KeyEmpty<K>::v, kc);
            if (v == KeyEmpty<K>::v) break;        // inserted
            if (v == kc) { nties++; break; }       // an equal source exists
            slot = (slot + 1) & tie_mask;
          }
        }
      }
    }
    if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || nodata_is_fixed(zz, nodata, W[g])) continue;
    K lo = kmin(kmin(sd[o - RW], sd[o + RW]), kmin(sd[o - 1], sd[o + 1]));
    if (TOPO == 8) lo = kmin(lo, kmin(kmin(sd[o - RW - 1], sd[o - RW + 1]), kmin(sd[o + RW - 1], sd[o + RW + 1])));
    const K f = kmax(zz == nodata ? (K)0 : CKey<T>::to(zz), step_up<EpsField<T>>(lo));
    const K d = sd[o];
    if (d != f) nbad++;
    const K wk = CKey<T>::to(W[g]);
    if (d < CKey<T>::INF && d > wk) lift = lift > (unsigned long long)(d - wk) ? lift : (unsigned long long)(d - wk);
  }
  for (int o = 32; o > 0; o >>= 1) {
    nbad += __shfl_down(nbad, o, 64);
    nsrc += __shfl_down(nsrc, o, 64);
    nties += __shfl_down(nties, o, 64);
    const unsigned long long other = __shfl_down(lift, o, 64);
    lift = lift > other ? lift : other;
  }
  if ((threadIdx.x & 63) == 0) {
    // striped: millions of waves adding to ONE word serialise (~12 ns per same-address atomic: 39 ms at S3)
    unsigned long long *tcs = tie_counts + 2 * ((t * 4 + (threadIdx.x >> 6)) & (TIE_STRIPES - 1));
    if (nsrc) atomicAdd(&tcs[0], (unsigned long long)nsrc);
    if (nties) atomicAdd(&tcs[1], (unsigned long long)nties);
    if (nbad) atomicAdd(bad, nbad);
    if (lift && lift > __hip_atomic_load(maxlift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlift, lift);
  }
}

// ---- one active tile to its local fixed point ----------------------------------------------------------------------
template <class P, int TOPO>
__global__ __launch_bounds__(NT) void k_eps_relax(const P p, typename P::K *D,
                                                  const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ count,
                                                  uint8_t *next_active, int w, int h, uint32_t tilesX, uint32_t tilesY) {
  using K = typename P::K;
  constexpr K KINF = P::INF;
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
      bool fixed;
      zk[j] = p.key((size_t)gy * w + gx, fixed);
      if (!(gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1 || fixed)) free_ |= 1u << j;
    }
  }
  __syncthreads();
  K d[ROWS], d0[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    d0[j] = d[j] = sd[(band * ROWS + j + 1) * RW + lx + 1];
    // a cell that is not relaxed gets its own value as its floor: every candidate max(floor, ...) is then >= d and the
    // updates below need no "is it free" test (this kernel is bound by VALU issue: 4.9e10 instructions per fill at S3)
    if (!(free_ & (1u << j))) zk[j] = d[j];
  }
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
  K sideLR[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; j++) sideLR[j] = kmin(sideL[j], sideR[j]);
  const K halo_up = band == 0 ? ring(-1, lx) : KINF, halo_dn = band == RBANDS - 1 ? ring(RCH, lx) : KINF;
  // D8 corner contributions of the rows above / below the band come through the neighbouring lanes' 3-minima, which
  // include `up` / `dn` of THOSE lanes; lanes 0 and 63 add the ring corners through sideL / sideR (rows ly-1..ly+1).

  // smallest value among the neighbours of cell j (D8: 8, D4: 4), given the rows above / below the band
  auto neigh_min = [&](const K up, const K dn, K out[ROWS]) {
    K own[ROWS];   // vertical neighbours in the own column; D8: the cell itself too (what the side lanes need from this
                   // column, and harmless in the cell's own minimum: a candidate from d itself is d + 1, never below d)
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      own[j] = kmin(j ? d[j - 1] : up, j + 1 < ROWS ? d[j + 1] : dn);
      if (TOPO == 8) own[j] = kmin(own[j], d[j]);
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const K col = TOPO == 8 ? own[j] : d[j];
      // (the shifts fill with the minimum's identity, so that the compiler folds one of them into a v_min_u32_dpp and needs no move per shift; what lanes 0
      // and 63 see beyond the tile, sideLR, is one more minimum -- one instruction less per cell and step)
      const K side = kmin(dpp_left(col, (K) ~(K)0), dpp_right(col, (K) ~(K)0));
      out[j] = kmin(kmin(own[j], side), sideLR[j]);
    }
  };

  int changed = 1, it = 0;
  constexpr int IT_CAP = 512;
  for (; it < IT_CAP; it++) {
    // Gauss-Seidel along the strip (vertical neighbours: both topologies)
#pragma unroll
    for (int j = 1; j < ROWS; j++) d[j] = kmin(d[j], kmax(zk[j], d[j - 1] + 1));   // (unsaturated steps: see the write-back)
#pragma unroll
    for (int j = ROWS - 2; j >= 0; j--) d[j] = kmin(d[j], kmax(zk[j], d[j + 1] + 1));
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
      int moved = 0;
      neigh_min(up, dn, mn);
#pragma unroll
      for (int j = 0; j < ROWS; j++) {
        const K cand = kmax(zk[j], mn[j] + 1);
        if (cand < d[j]) { d[j] = cand; moved = 1; }
      }
      changed |= moved;
      if (!__any(moved)) break;   // the remaining steps of this trip would compute the same values
    }
  }
  if (it == IT_CAP && threadIdx.x == 0) next_active[t] = 1;   // iteration cap hit: finish this tile next round
  // write back and wake the tiles across every edge that changed
  int top = 0, bot = 0, lef = 0, rig = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    // The steps above are plain + 1 (one instruction instead of step_up's two, in a loop bound by instruction issue).  They
    // differ from step_up only past the key of +infinity -- a value counted up from a cell of elevation +inf -- and those
    // values, POSINF < d < "not reached", all stand for POSINF: saturated here, before anything is compared or stored.
    // ("Not reached" + 1 is still above every key and never below a d.)
    const K dj = (d[j] > P::POSINF && d[j] < KINF) ? P::POSINF : d[j];
    if (!(free_ & (1u << j)) || dj >= d0[j]) continue;
    const int ly = band * ROWS + j;
    __hip_atomic_store(&D[(size_t)(y0 + ly) * w + gx], dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---- tie detector -------------------------------------------------------------------------------------------------
// The surface above equals the reference's when no two cells of its heap hold the same elevation.  The cells that
// matter are the SOURCES of a gradient: cells that keep their own elevation (they pass through the heap) and have a
// raised neighbour (whose value is counted from them).  Two sources of equal elevation = the reference's result there
// follows std::priority_queue's pop order (a plateau or a lake entered through several cells of one level gets its
// gradient from whichever of them pops first).  The proof pass (k_eps_check) has the tile and its ring staged anyway: it
// inserts the sources' keys into an open-addressing set, `ties` = sources whose key was already there.  Conservative
// (equal sources far apart need not interact), no extra pass, and what FillDepressions(epsilon=True) warns with.

struct Stats {
  uint32_t rounds = 0, attempts = 0;
  uint64_t tile_relaxations = 0, slack = 0, max_lift = 0, tie_sources = 0;
};
static thread_local Stats g_stats;

// Rounds until no tile is active: compact the active-tile flags into a list + count on the device, relax that list;
// BATCH rounds are enqueued per host read-back, the rounds enqueued past the fixed point see an empty list.
template <class P, int TOPO>
static void relax_until_quiet(const P p, typename P::K *D, uint8_t *tflags, uint32_t *tlist, uint32_t *ctr /* BATCH words */, int w,
                              int h, const char *name, hipStream_t s) {
  uint32_t *hw = Workspace::get().host_words();
  const uint32_t tilesX = (w + CW - 1) / CW, tilesY = (h + RCH - 1) / RCH, ntiles = tilesX * tilesY;
  uint32_t grid = ntiles;
  for (bool done = false; !done;) {
    RD_HIP(hipMemsetAsync(ctr, 0, BATCH * sizeof(uint32_t), s));
    for (int b = 0; b < BATCH; b++) {
      RD_LAUNCH("eps.tiles_compact", k_tiles_compact, dim3((ntiles + NT - 1) / NT), dim3(NT), 0, s, tflags, ntiles, tlist, ctr + b);
      RD_LAUNCH(name, (k_eps_relax<P, TOPO>), dim3(grid), dim3(NT), 0, s, p, D, (const uint32_t *)tlist,
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
    if (g_stats.rounds > (1u << 24)) throw Error(RDGPU_ERR_HIP, "rdgpu: tile relaxation did not terminate");
  }
}

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
  // tie detector (RDGPU_EPS_TIES=0 skips it): one slot per four cells, grown if the DEM has more sources than that
  const char *te = getenv("RDGPU_EPS_TIES");
  const bool ties_on = !(te && te[0] == '0');
  unsigned long long tie_slots = 1024;
  while (ties_on && tie_slots < n / 4) tie_slots *= 2;
  K *tie_table = nullptr;
  bool bitmap = false;   // the tie detector's set is the 2^32-bit map (f32, large rasters): k_eps_check is told by tie_mask == 0
  unsigned long long *tc = ws.buf<unsigned long long>("eps.tiecounts", 2 * TIE_STRIPES);
  std::vector<unsigned long long> htc_v(2 * TIE_STRIPES);
  unsigned long long htc[2] = {0, 0};
  for (;;) {
    g_stats.attempts++;
    g_stats.slack = (uint64_t)X;
    if (ties_on) {
      if (sizeof(K) == 4 && n >= (1ull << 26)) {   // the bitmap over all 2^32 keys (a small raster takes the hashed set: no 512 MB for a 200 x 200 DEM)
        tie_slots = (1ull << 32) / 32;
        tie_table = ws.buf<K>("eps.tiebits", tie_slots);
        RD_HIP(hipMemsetAsync(tie_table, 0, tie_slots * sizeof(K), s));
        bitmap = true;
      } else {
        tie_table = ws.buf<K>("eps.tietable", tie_slots);
        RD_HIP(hipMemsetAsync(tie_table, 0xFF, tie_slots * sizeof(K), s));
      }
      RD_HIP(hipMemsetAsync(tc, 0, 2 * TIE_STRIPES * sizeof(unsigned long long), s));
    }
    RD_HIP(hipMemsetAsync(tflags, 0, ntiles, s));
    RD_HIP(hipMemsetAsync(ctr + BATCH, 0, 8 * sizeof(uint32_t), s));
    RD_LAUNCH("eps.init", (k_eps_init<T, TOPO>), dim3(xcd_grid(ntiles)), dim3(NT), 0, s, (const T *)d_z, d_W, nodata, X, D, tflags,
              w, h, tilesX, ntiles);
    relax_until_quiet<EpsField<T>, TOPO>(EpsField<T>{d_z, nodata, d_W}, D, tflags, tlist, ctr, w, h, "eps.relax", s);
    for (;;) {   // (the proof pass; run again only if the tie detector's set turned out too small for this DEM)
      RD_LAUNCH("eps.check", (k_eps_check<T, TOPO>), dim3(xcd_grid(ntiles)), dim3(NT), 0, s, (const T *)d_z, d_W, nodata, (const K *)D, w,
                h, tilesX, ntiles, ctr + BATCH, (unsigned long long *)(ctr + BATCH + 2), tie_table, bitmap ? 0ull : tie_slots - 1, tc);
      RD_HIP(hipMemcpyAsync(hw, ctr + BATCH, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      if (ties_on) RD_HIP(hipMemcpyAsync(htc_v.data(), tc, 2 * TIE_STRIPES * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      htc[0] = htc[1] = 0;
      if (ties_on)
        for (uint32_t i = 0; i < TIE_STRIPES; i++) { htc[0] += htc_v[2 * i]; htc[1] += htc_v[2 * i + 1]; }
      if (!(ties_on && !bitmap && hw[0] == 0 && htc[0] * 2 > tie_slots)) break;
      while (tie_slots < 4 * htc[0]) tie_slots *= 2;   // more than half full: a larger set, the same D
      tie_table = ws.buf<K>("eps.tietable", tie_slots);
      RD_HIP(hipMemsetAsync(tie_table, 0xFF, tie_slots * sizeof(K), s));
      RD_HIP(hipMemsetAsync(tc, 0, 2 * TIE_STRIPES * sizeof(unsigned long long), s));
      RD_HIP(hipMemsetAsync(ctr + BATCH, 0, 8 * sizeof(uint32_t), s));
    }
    if (ties_on) g_stats.tie_sources = htc[1];
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

// ------------------------------------------------------------------------------------------------------------------
// PriorityFloodWatersheds_Barnes2014<topo>(elevations, labels, alter_elevations) (reference
// depressions/Barnes2014.hpp:713-807).  The reference labels every cell with the label of the cell that CLOSED it -- the
// first of its neighbours to be processed -- and hands out a new label to every data cell that is popped without one
// (:777-778: the cells of the raster border and the data cells next to a NoData region that is connected to it).
// Cells are processed in the order of the level the flood has when it reaches them (the plain fill's W), and the cells
// of one lake in breadth-first order from the cell the flood entered it through (the pit queue is a FIFO, :762-764).
// So with  P(c) = (W(c), steps from the lake's entry)  -- the fixed point of the SAME relaxation as the epsilon fill on
// 64-bit keys, floor (W << 32): a cell with a lower neighbour holds (W, 0), a lake or flat cell one more than its
// lowest neighbour -- the cell that closed c is a neighbour of smallest P, and all such neighbours carry the same
// label.  Labels then are: parent(c) = the lowest-P neighbour, label = the nearest ancestor that starts a label
// (pointer chasing with path compression), numbered in the order of the starters' elevations (their pop order).
// Identical to the reference (labels AND numbering) on DEMs without equal elevations among the heap's cells; with ties
// the partition of the reference depends on std::priority_queue's pop order.
// ------------------------------------------------------------------------------------------------------------------
namespace shed {
using namespace eps;

constexpr uint32_t TERM = 0x80000000u;        // link word: a terminal; low bits = the cell that starts the label
constexpr uint32_t NONE = 0x7FFFFFFFu;        // ... or none (the label stays -1)

template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_ws_init(const T *__restrict__ W, uint64_t *__restrict__ D, uint8_t *tile_active, int w,
                                                int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint32_t sw[RH * RW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * CW, y0 = (int)(t / tilesX) * RCH;
  for (int i = threadIdx.x; i < RH * RW; i += NT) {
    const int ly = i / RW, lx = i - ly * RW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    sw[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? Key32<T>::to(W[(size_t)gy * w + gx]) : 0xFFFFFFFFu;
  }
  __syncthreads();
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  int anyinf = 0;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const int o = (ly + 1) * RW + lx + 1;
    const uint32_t k = sw[o];
    uint64_t d = (uint64_t)k << 32;
    if (!(gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1)) {
      uint32_t lo = kmin(kmin(sw[o - RW], sw[o + RW]), kmin(sw[o - 1], sw[o + 1]));
      if (TOPO == 8) lo = kmin(lo, kmin(kmin(sw[o - RW - 1], sw[o - RW + 1]), kmin(sw[o + RW - 1], sw[o + RW + 1])));
      if (!(lo < k)) { d = ShedField<T>::INF; anyinf = 1; }   // lake / flat cell: its level comes from the relaxation
    }
    D[(size_t)gy * w + gx] = d;
  }
  if (__syncthreads_or(anyinf) && threadIdx.x == 0) tile_active[t] = 1;
}

// link[c]: the lowest-P neighbour (lowest index among equals), or a terminal: TERM | c for a cell that starts a label,
// TERM | NONE for a cell that stays unlabelled.  Without NoData cells the starters are exactly the border cells.
template <class T, int TOPO>
__global__ __launch_bounds__(NT) void k_ws_parent(const T *__restrict__ z, T nodata, const uint64_t *__restrict__ D,
                                                  uint32_t *__restrict__ link, int w, int h, uint32_t *any_nodata) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  int nd = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    const bool isnd = z[c] == nodata;
    nd |= isnd;
    if (x == 0 || y == 0 || x == w - 1 || y == h - 1) {
      link[c] = TERM | (isnd ? NONE : (uint32_t)c);   // :777: a border data cell starts a label, a NoData one none
      continue;
    }
    uint64_t best = ~0ull;
    uint32_t arg = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {   // raster order: the lowest index wins among equals
      if (k == 4) continue;
      const int dx = k % 3 - 1, dy = k / 3 - 1;
      if (TOPO == 4 && dx != 0 && dy != 0) continue;
      const uint64_t q = c + (int64_t)dy * w + dx;
      const uint64_t v = D[q];
      if (v < best) { best = v; arg = (uint32_t)q; }
    }
    link[c] = arg;
  }
  if (__any(nd) && (threadIdx.x & 63) == 0) *any_nodata = 1;
}

// bounded pointer chase with path compression: link[c] <- the terminal word its chain ends in (or a pointer further up
// the chain when the hop budget runs out: the host repeats while flagged)
__global__ __launch_bounds__(NT) void k_ws_chase(uint32_t *link, uint64_t n, int maxhops, uint32_t *flag) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    uint32_t v = link[c];
    if (v & TERM) continue;
    const uint32_t v0 = v;
    bool unfinished = true;
    for (int hops = 0; hops < maxhops; hops++) {
      const uint32_t q = link[v];
      v = q;
      if (q & TERM) { unfinished = false; break; }
    }
    if (v != v0) link[c] = v;
    if (unfinished) *flag = 1;
  }
}

// NoData present.  Pass A (ua): for a NoData cell, does its chain of parents run through NoData cells to a NoData cell
// of the border (then it is never labelled, and a data cell it closes starts a label)?  ua[c] = TERM | 1 yes, TERM | 0
// no (data cells, and what a data cell closed), else the parent.
template <class T>
__global__ __launch_bounds__(NT) void k_ws_nodata_links(const T *__restrict__ z, T nodata, const uint32_t *__restrict__ link,
                                                        uint32_t *__restrict__ ua, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t l = link[c];
    if (z[c] != nodata) ua[c] = TERM | 0u;
    else ua[c] = (l & TERM) ? (TERM | 1u) : l;    // a NoData border cell / an interior NoData cell: ask its parent
  }
}
// Pass B: the final links.  A data cell whose parent is never labelled starts a label; a NoData cell that is never
// labelled is a terminal without one.
template <class T>
__global__ __launch_bounds__(NT) void k_ws_seed_links(const T *__restrict__ z, T nodata, const uint32_t *__restrict__ ua,
                                                      uint32_t *__restrict__ link, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t l = link[c];
    if (l & TERM) continue;                                   // border cells are done
    if (z[c] == nodata) { if (ua[c] == (TERM | 1u)) link[c] = TERM | NONE; }
    else if (ua[l] == (TERM | 1u)) link[c] = TERM | (uint32_t)c;   // (ua of a data parent is TERM | 0)
  }
}

template <class T>
__global__ __launch_bounds__(NT) void k_ws_collect(const T *__restrict__ z, const uint32_t *__restrict__ link, uint64_t n,
                                                   uint64_t *keys, uint32_t *count, uint32_t cap) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c0 = (uint64_t)blockIdx.x * NT; c0 < n; c0 += stride) {   // whole blocks: block_append has barriers
    const uint64_t c = c0 + threadIdx.x;
    const bool hit = c < n && link[c] == (TERM | (uint32_t)c);
    const uint32_t slot = block_append(hit, count);
    if (hit && slot < cap) keys[slot] = ((uint64_t)Key32<T>::to(z[c]) << 32) | (uint64_t)c;
    __syncthreads();   // block_append's LDS words are free again
  }
}

__global__ __launch_bounds__(NT) void k_ws_number(const uint64_t *__restrict__ sorted, uint32_t nseeds, int32_t *labels) {
  const uint32_t i = blockIdx.x * NT + threadIdx.x;
  if (i < nseeds) labels[(uint32_t)sorted[i]] = (int32_t)i + 1;   // clabel starts at 1, :721
}

__global__ __launch_bounds__(NT) void k_ws_apply(const uint32_t *__restrict__ link, int32_t *labels, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t t = link[c] & ~TERM;
    if (t == NONE) labels[c] = -1;                 // labels.noData(), :737-738
    else if (t != (uint32_t)c) labels[c] = labels[t];
  }
}

}  // namespace shed
}  // namespace rdgpu

#include <hipcub/hipcub.hpp>

#define RD_WS_FILL(SUF, T) extern "C" int rdgpu_fill_dev_##SUF(T *, int, int, int, void *);
RD_WS_FILL(u8, uint8_t)
RD_WS_FILL(i16, int16_t)
RD_WS_FILL(u16, uint16_t)
RD_WS_FILL(i32, int32_t)
RD_WS_FILL(u32, uint32_t)
RD_WS_FILL(i8, int8_t)
#undef RD_WS_FILL

namespace rdgpu {
namespace shed {
static int fill_of(uint8_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_u8(p, w, h, t, s); }
static int fill_of(int8_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_i8(p, w, h, t, s); }
static int fill_of(int16_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_i16(p, w, h, t, s); }
static int fill_of(uint16_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_u16(p, w, h, t, s); }
static int fill_of(int32_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_i32(p, w, h, t, s); }
static int fill_of(uint32_t *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_u32(p, w, h, t, s); }
static int fill_of(float *p, int w, int h, int t, hipStream_t s) { return rdgpu_fill_dev_f32(p, w, h, t, s); }

template <class T, int TOPO>
static void run(T *d_z, T nodata, const T *d_W, int w, int h, int32_t *d_labels, hipStream_t s) {
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  const uint32_t tilesX = (w + CW - 1) / CW, tilesY = (h + RCH - 1) / RCH, ntiles = tilesX * tilesY;
  uint64_t *D = ws.buf<uint64_t>("shed.D", n);
  uint8_t *tflags = ws.buf<uint8_t>("eps.tflags", ntiles);
  uint32_t *tlist = ws.buf<uint32_t>("eps.tlist", ntiles);
  uint32_t *ctr = ws.buf<uint32_t>("eps.ctr", BATCH + 8);
  g_stats = Stats();
  g_stats.attempts = 1;
  RD_HIP(hipMemsetAsync(tflags, 0, ntiles, s));
  RD_HIP(hipMemsetAsync(ctr + BATCH, 0, 8 * sizeof(uint32_t), s));
  RD_LAUNCH("shed.init", (k_ws_init<T, TOPO>), dim3(xcd_grid(ntiles)), dim3(NT), 0, s, d_W, D, tflags, w, h, tilesX, ntiles);
  relax_until_quiet<ShedField<T>, TOPO>(ShedField<T>{d_W}, D, tflags, tlist, ctr, w, h, "shed.relax", s);
  const uint32_t sgrid = (uint32_t)std::min<uint64_t>((n + NT - 1) / NT, 256u * 32u);
  uint32_t *link = ws.buf<uint32_t>("shed.link", n);
  RD_LAUNCH("shed.parent", (k_ws_parent<T, TOPO>), dim3(sgrid), dim3(NT), 0, s, (const T *)d_z, nodata, (const uint64_t *)D, link, w,
            h, ctr + BATCH);
  RD_HIP(hipMemcpyAsync(hw, ctr + BATCH, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  auto chase = [&](uint32_t *lk) {
    for (;;) {
      RD_HIP(hipMemsetAsync(ctr + BATCH + 1, 0, sizeof(uint32_t), s));
      RD_LAUNCH("shed.chase", k_ws_chase, dim3(sgrid), dim3(NT), 0, s, lk, n, 32, ctr + BATCH + 1);
      RD_HIP(hipMemcpyAsync(hw, ctr + BATCH + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      if (hw[0] == 0) break;
    }
  };
  if (hw[0]) {   // NoData cells: who is never labelled, and which data cells start a label next to them
    uint32_t *ua = ws.buf<uint32_t>("shed.ua", n);
    RD_LAUNCH("shed.nodata_links", (k_ws_nodata_links<T>), dim3(sgrid), dim3(NT), 0, s, (const T *)d_z, nodata, (const uint32_t *)link,
              ua, n);
    chase(ua);
    RD_LAUNCH("shed.seed_links", (k_ws_seed_links<T>), dim3(sgrid), dim3(NT), 0, s, (const T *)d_z, nodata, (const uint32_t *)ua, link,
              n);
  }
  chase(link);
  // the cells that start a label, in the order of their elevations (= the order the reference pops them in)
  RD_HIP(hipMemsetAsync(ctr + BATCH + 2, 0, sizeof(uint32_t), s));
  uint32_t cap = (uint32_t)std::min<uint64_t>(n, (uint64_t)4 * (w + h) + 1024);
  uint64_t *keys = nullptr;
  uint32_t nseeds = 0;
  for (;;) {
    keys = ws.buf<uint64_t>("shed.keys", (size_t)2 * cap);
    RD_LAUNCH("shed.collect", (k_ws_collect<T>), dim3(sgrid), dim3(NT), 0, s, (const T *)d_z, (const uint32_t *)link, n, keys,
              ctr + BATCH + 2, cap);
    RD_HIP(hipMemcpyAsync(hw, ctr + BATCH + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    nseeds = hw[0];
    if (nseeds <= cap) break;
    cap = nseeds;   // NoData regions made more starters than the border has cells: once more with room for all
    RD_HIP(hipMemsetAsync(ctr + BATCH + 2, 0, sizeof(uint32_t), s));
  }
  uint64_t *sorted = keys + cap;
  if (nseeds) {
    size_t tmp_bytes = 0;
    RD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys, sorted, (int)nseeds, 0, 64, s));
    void *tmp = ws.buf<uint8_t>("shed.sorttmp", tmp_bytes);
    RD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, sorted, (int)nseeds, 0, 64, s));
    RD_LAUNCH("shed.number", k_ws_number, dim3((nseeds + NT - 1) / NT), dim3(NT), 0, s, (const uint64_t *)sorted, nseeds, d_labels);
  }
  RD_LAUNCH("shed.apply", k_ws_apply, dim3(sgrid), dim3(NT), 0, s, (const uint32_t *)link, d_labels, n);
}

template <class T>
static void watersheds_device(T *d_z, T nodata, int w, int h, int topology, int alter, int32_t *d_labels, hipStream_t s) {
  if (!d_z || !d_labels) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: width and height must be positive");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: topology must be 8 or 4");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: raster too large");
  const size_t n = (size_t)w * h;
  T *W = Workspace::get().buf<T>("eps.W", n);
  RD_HIP(hipMemcpyAsync(W, d_z, n * sizeof(T), hipMemcpyDeviceToDevice, s));
  if (fill_of(W, w, h, topology, s) != RDGPU_OK) throw Error(RDGPU_ERR_HIP, std::string("rdgpu_watersheds: ") + rdgpu_last_error());
  if (topology == 8) run<T, 8>(d_z, nodata, W, w, h, d_labels, s);
  else run<T, 4>(d_z, nodata, W, w, h, d_labels, s);
  if (alter) RD_HIP(hipMemcpyAsync(d_z, W, n * sizeof(T), hipMemcpyDeviceToDevice, s));   // :793-794: as PriorityFlood_Barnes2014
}

template <class T>
static void watersheds_host(T *dem, T nodata, int w, int h, int topology, int alter, int32_t *labels) {
  if (!dem || !labels) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  int32_t *dl = Workspace::get().buf<int32_t>("host.labels", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  watersheds_device<T>(d, nodata, w, h, topology, alter, dl, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(labels, dl, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (alter) RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace shed
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
  out->tie_sources = eps::g_stats.tie_sources;
  return RDGPU_OK;
}

#define RD_WS_API(SUF, T)                                                                                              \
  extern "C" int rdgpu_watersheds_##SUF(T *dem, T nodata, int w, int h, int topology, int alter, int32_t *labels) {  \
    return guarded([&] { shed::watersheds_host<T>(dem, nodata, w, h, topology, alter, labels); });                    \
  }                                                                                                                    \
  extern "C" int rdgpu_watersheds_dev_##SUF(T *d_dem, T nodata, int w, int h, int topology, int alter, int32_t *d_labels, \
                                            void *stream) {                                                            \
    return guarded([&] { shed::watersheds_device<T>(d_dem, nodata, w, h, topology, alter, d_labels, (hipStream_t)stream); }); \
  }
RD_WS_API(u8, uint8_t)
RD_WS_API(i16, int16_t)
RD_WS_API(u16, uint16_t)
RD_WS_API(i32, int32_t)
RD_WS_API(u32, uint32_t)
RD_WS_API(f32, float)
RD_WS_API(i8, int8_t)
