// pfdirs.hip -- PriorityFloodFlowdirs_Barnes2014 (reference include/richdem/depressions/Barnes2014.hpp:483-555) without a
// priority queue.
//
// The reference floods the DEM without raising it on a queue ordered by (elevation, insertion counter); a cell's
// direction points at the neighbour that was popped first.  With distinct elevations the pop order is the
// lexicographic order of the sequences
//     key(c) = (F_0(c), F_1(c), ..., F_k(c) = z(c))
// where F_0 is the plain fill (the minimax level from the raster border) and, for a cell that is still below its level
// (F_{j-1}(c) > z(c): "wet"), F_j(c) is the minimax level from the cells through which the flood ENTERS its pocket --
// the wet cells next to THE cell of elevation F_{j-1}(c) -- inside the pocket; a sequence that is a prefix of another one
// comes first (it is that pocket's entry phase).  Proof sketch: every cell below level w is popped before any cell of
// elevation >= w; after the one cell of elevation w the flood runs through the pocket(s) behind it before anything
// higher, and inside them the same argument holds with the pocket's entry cells as sources.
// So: one fill per nesting level -- the compact-label fill of fill.hip with walls (the highest value of the type)
// everywhere outside the wet cells and OUTLETS at the entry cells (rdgpu_fill_outlets_dev_*) -- and per cell a shrinking
// set of candidate neighbours: those with the lowest F_0, among them those with the lowest F_1, ...; a candidate whose
// sequence has ended (its elevation IS the level) is the phase cell itself and wins.  From the second level on walls are
// outlets too, the fill's 64 x 64 tiles that hold nothing but walls are skipped for good, and every kernel is launched over
// compacted lists of the tiles that still have work (k_skip_state).
// EQUAL ELEVATIONS (r04): the reference's stable queue pops them in order of insertion, so its output is a function of the
// DEM -- and it is reproduced: a raster with twins is flooded on its UNIQUE RANKS of (elevation, tie key), and the tie key
// is iterated to the fixed point "discovery time under the flood it induces" (k_tie_* below): equal to the compiled
// reference on its own tie-heavy vectors and, digest for digest, on the 40000 x 40000 bench DEM (1.58e9 cells with a twin;
// r04: 4 floods, 19 s; r05: 2 floods + the tree iteration in pf_flowdirs_device, 10.9 s).  `twins`, `tie_passes`, `unresolved` (ranks still moving when RDGPU_PFD_TIE_PASSES ran out: 0 = exact)
// in rdgpu_pf_flowdirs_get_stats.  RDGPU_PFD_RANKS=0: ties decided inside the levels by neighbour number (r03).
// tests/tools/proto_pf_flowdirs.py is the tie-free algorithm in numpy, checked against the oracle.
#include "common.hpp"

#include <type_traits>

#include <hipcub/hipcub.hpp>

#include <chrono>

#include <limits>

#define RD_OUTLETS(SUF, T) extern "C" int rdgpu_fill_outlets_lists_dev_##SUF(T *, const uint8_t *, const uint8_t *, const uint32_t *, uint32_t, \
                                                                           const uint32_t *, int, int, void *);                          \
                           extern "C" int rdgpu_fill_dev_##SUF(T *, int, int, int, void *);
RD_OUTLETS(u8, uint8_t) RD_OUTLETS(i8, int8_t) RD_OUTLETS(i16, int16_t) RD_OUTLETS(u16, uint16_t) RD_OUTLETS(i32, int32_t)
RD_OUTLETS(u32, uint32_t) RD_OUTLETS(f32, float)
#undef RD_OUTLETS
extern "C" const char *rdgpu_last_error(void);

namespace rdgpu {
namespace pfd {

constexpr int NT = 256;
__device__ __forceinline__ int ndx(int n) { return (n == 1 || n == 2 || n == 8) ? -1 : (n >= 4 && n <= 6) ? 1 : 0; }
__device__ __forceinline__ int ndy(int n) { return (n >= 2 && n <= 4) ? -1 : (n >= 6 && n <= 8) ? 1 : 0; }

template <class T>
__host__ __device__ constexpr T wall_value() {
  return std::numeric_limits<T>::has_infinity ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::max();
}

static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NT - 1) / NT, 256u * 64u); }

// One level of the comparison.  F holds this level's fill.  A cell with more than one candidate keeps those of the lowest
// level; one of them whose own elevation IS that level ends the comparison.  counters[0]: cells still undecided.
constexpr int FT = 64;   // the fill's descent tiles (fill.hip: DW = DH = 64); the kernels here walk the raster tile by tile
template <class T>
__global__ __launch_bounds__(NT) void k_refine(const T *__restrict__ z, const T *__restrict__ F, uint8_t *cand, int w, int h,
                                               unsigned long long *counters, unsigned long long *ambiguous,
                                               const uint32_t *__restrict__ tiles) {
  // one block per tile of the list: the tiles that held a wet cell, ring included, a level ago (a tile of walls holds no
  // undecided cell: an undecided cell's candidates were wet then)
  const uint32_t t = tiles[blockIdx.x];
  const int ftx = (w + FT - 1) / FT;
  const int x = (int)(t % (uint32_t)ftx) * FT + (int)(threadIdx.x & 63), y0 = (int)(t / (uint32_t)ftx) * FT;
  uint32_t open = 0, amb = 0;
  for (int ly = (int)(threadIdx.x >> 6); ly < FT; ly += NT / 64) {
    const int y = y0 + ly;
    if (x >= w || y >= h) continue;
    const size_t c = (size_t)y * w + x;
    const uint32_t m = cand[c];
    if ((m & (m - 1u)) == 0u) continue;   // decided (or a border cell: 0)
    T fv[8], zv[8];
#pragma unroll
    for (int k = 1; k <= 8; k++) {   // (an interior cell: all eight neighbours exist)
      const size_t g = (size_t)(y + ndy(k)) * w + (x + ndx(k));
      fv[k - 1] = F[g];
      zv[k - 1] = z[g];
    }
    T lo = wall_value<T>();
#pragma unroll
    for (int k = 0; k < 8; k++)
      if ((m >> k & 1u) && fv[k] < lo) lo = fv[k];
    uint32_t keep = 0, ended = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool on = (m >> k & 1u) && fv[k] == lo;
      keep |= (on ? 1u : 0u) << k;
      ended |= ((on && zv[k] == lo) ? 1u : 0u) << k;
    }
    if (keep == 0u) keep = m;                                // (every candidate behind a wall: cannot happen without ties)
    const uint32_t out = ended ? (ended & (0u - ended)) : keep;   // the phase cell (one without ties: the lowest bit)
    if (ended & (ended - 1u)) amb++;                              // several cells of that elevation: a tie decided by number
    cand[c] = (uint8_t)out;
    if (out & (out - 1u)) open++;
  }
  for (int o = 32; o > 0; o >>= 1) { open += __shfl_down(open, o, 64); amb += __shfl_down(amb, o, 64); }
  if ((threadIdx.x & 63) == 0 && open) atomicAdd(&counters[(blockIdx.x & 63) * 2], (unsigned long long)open);   // striped
  if ((threadIdx.x & 63) == 0 && amb) atomicAdd(ambiguous, (unsigned long long)amb);
}

// The next level's problem, in place over this level's fill: a wet cell (below its level) keeps its elevation, everything
// else becomes a wall; outlet = wet cell next to THE cell whose elevation is its level -- and every wall (a wall as an
// outlet changes no level: a path over it costs the wall's height; it spares the fill the walls' basins).
// counters[1]: wet cells.  active[]: per 64 x 64 tile of the fill, "a wet cell in the tile or next to it".
template <class T>
__global__ __launch_bounds__(NT) void k_next_level(const T *__restrict__ z, T *F, uint8_t *__restrict__ outlet, uint8_t *active, int w,
                                                   int h, unsigned long long *counters, const uint32_t *__restrict__ tiles, T *last_level) {
  // (the same list: everywhere else the raster is walls and outlets already, and for good)
  const uint32_t t = tiles[blockIdx.x];
  const int ftx = (w + FT - 1) / FT;
  const int x = (int)(t % (uint32_t)ftx) * FT + (int)(threadIdx.x & 63), y0 = (int)(t / (uint32_t)ftx) * FT;
  uint32_t nwet = 0;
  for (int ly = (int)(threadIdx.x >> 6); ly < FT; ly += NT / 64) {
    const int y = y0 + ly;
    if (x >= w || y >= h) continue;
    const size_t c = (size_t)y * w + x;
    const T f = F[c], e = z[c];
    const bool wet = f > e && f < wall_value<T>();
    uint8_t o = 1;
    if (wet) {   // (a wet cell is never on the raster's border)
      nwet++;
      o = 0;
#pragma unroll
      for (int k = 1; k <= 8; k++) o |= z[(size_t)(y + ndy(k)) * w + (x + ndx(k))] == f ? 1 : 0;
      const int tx0 = (x - 1) / FT, tx1 = (x + 1) / FT, ty0 = (y - 1) / FT, ty1 = (y + 1) / FT;
      for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) active[(size_t)ty * ftx + tx] = 1;
    }
    outlet[c] = o;
    if (last_level && wet) last_level[c] = f;   // (r05) overwritten level after level: in the end the level of the cell's INNERMOST pocket
    F[c] = wet ? e : wall_value<T>();
  }
  for (int o = 32; o > 0; o >>= 1) nwet += __shfl_down(nwet, o, 64);
  if ((threadIdx.x & 63) == 0 && nwet) atomicAdd(&counters[(blockIdx.x & 63) * 2 + 1], (unsigned long long)nwet);
}

// skip state of the fill's tiles for the coming level: 0 = has work, 1 = nothing but walls for the first time (its labels
// are written once more), 2 = the same as before.  A tile never gets work again: the wet set only shrinks.  And the
// lists the sparse kernels are launched over: tiles in state 0 or 1 (the fill's descent pass), the 64 x 32 tiles of the
// tiles in state 0 (its pair pass), the tiles in state 0 (its last pass; the two kernels here).  counts: their lengths.
__global__ __launch_bounds__(NT) void k_skip_state(uint8_t *skip, uint8_t *active, uint32_t ntiles, uint32_t ftx, uint32_t scan_rows,
                                                   uint32_t *lists, uint32_t stride, uint32_t *counts) {
  __shared__ uint32_t base[3], cnt[3];
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  uint32_t st = 2, pos[3] = {0, 0, 0};
  if (t < ntiles) {
    st = active[t] ? 0 : (skip[t] ? 2 : 1);
    skip[t] = (uint8_t)st;
    active[t] = 0;
  }
  const uint32_t ty = t / ftx, tx = t - ty * ftx;
  const uint32_t nscan = st == 0 ? (2 * ty + 1 < scan_rows ? 2u : 1u) : 0u;
  if (st <= 1) pos[0] = atomicAdd(&cnt[0], 1u);
  if (nscan) pos[1] = atomicAdd(&cnt[1], nscan);
  if (st == 0) pos[2] = atomicAdd(&cnt[2], 1u);
  __syncthreads();
  if (threadIdx.x < 3) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]) : 0u;
  __syncthreads();
  if (st <= 1) lists[base[0] + pos[0]] = t;
  for (uint32_t k = 0; k < nscan; k++) lists[stride + base[1] + pos[1] + k] = (2 * ty + k) * ftx + tx;
  if (st == 0) lists[2 * (size_t)stride + base[2] + pos[2]] = t;
}

// every tile, for the first level
__global__ __launch_bounds__(NT) void k_all_tiles(uint32_t *list, uint32_t ntiles) {
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  if (t < ntiles) list[t] = t;
}

__global__ __launch_bounds__(128) void k_sum_counters(unsigned long long *counters, unsigned long long *out) {
  __shared__ unsigned long long acc[2];
  if (threadIdx.x < 2) acc[threadIdx.x] = 0;
  __syncthreads();
  atomicAdd(&acc[threadIdx.x & 1], counters[threadIdx.x]);
  __syncthreads();
  if (threadIdx.x < 2) out[threadIdx.x] = acc[threadIdx.x];
  counters[threadIdx.x] = 0;
}

// Equal elevations anywhere in the raster: one bit per possible 32-bit key (512 MB), a cell whose bit is set already has a
// twin.  No twins => the result is the reference's (the proof in the header needs distinct elevations, nothing else).
template <class T>
__global__ __launch_bounds__(NT) void k_twins(const T *__restrict__ z, uint64_t n, uint32_t *bits, unsigned long long *counters) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  uint32_t twins = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const T v = z[c];
    const uint32_t k = Key32<T>::to(v), bit = 1u << (k & 31u);
    // (a cell that holds the type's highest value -- the levels' WALL -- counts as well: the wall must lie above every
    // elevation, and on the ranks, which a count > 0 sends the raster to, it does)
    if ((atomicOr(&bits[k >> 5], bit) & bit) || !(v < wall_value<T>())) twins++;
  }
  for (int o = 32; o > 0; o >>= 1) twins += __shfl_down(twins, o, 64);
  if ((threadIdx.x & 63) == 0 && twins) atomicAdd(&counters[(blockIdx.x & 63) * 2], (unsigned long long)twins);   // striped
}

// candidates of every cell at the start: all eight neighbours of an interior cell, none for a border cell
__global__ __launch_bounds__(NT) void k_init_cand(uint8_t *cand, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    cand[c] = (x == 0 || y == 0 || x == w - 1 || y == h - 1) ? 0 : 0xFF;
  }
}

// the directions: the border cells' fixed ones (:508-528), 0 for NoData cells (:545-548), else the one candidate left
template <class T>
__global__ __launch_bounds__(NT) void k_finish(const T *__restrict__ z, T nodata, const uint8_t *__restrict__ cand, uint8_t *dirs, int w,
                                               int h, unsigned long long *unresolved) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  uint32_t bad = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    int d;
    if (y == 0 || y == h - 1 || x == 0 || x == w - 1) {
      // the reference's writes in their order (:508-528; a later one wins where a raster is a single row or column):
      // top row 3, bottom row 7, the side columns of the rows between 1 / 5, the four corners 2, 4, 8, 6
      d = 0;
      if (y == 0) d = 3;
      if (y == h - 1) d = 7;
      if (y >= 1 && y <= h - 2) {
        if (x == 0) d = 1;
        if (x == w - 1) d = 5;
      }
      if (x == 0 && y == 0) d = 2;
      if (x == w - 1 && y == 0) d = 4;
      if (x == 0 && y == h - 1) d = 8;
      if (x == w - 1 && y == h - 1) d = 6;
    } else if (z[c] == nodata) {
      d = 0;
    } else {
      const uint32_t m = cand[c];
      if (m & (m - 1u)) bad++;
      d = m ? __ffs((int)m) : 0;
    }
    dirs[c] = (uint8_t)d;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(unresolved, (unsigned long long)bad);
}

static thread_local rdgpu_pf_flowdirs_stats g_stats = {0, 0, 0, 0, 0};
static thread_local bool g_rank_pass = false;   // the call runs on the unique ranks of another raster (see pf_flowdirs_device)
// r05: where the flood of a rank raster leaves, per cell, the level of its innermost pocket (null: not wanted).  The header's
// key(c) = (F_0(c), ..., z(c)) IS the path root -> c of the record tree, so the cell of rank F_(k-1)(c) is c's nearest
// ancestor of greater rank: after an exact flood the record tree needs no search at all (k_tie_from_levels).
static thread_local uint32_t *g_last_level = nullptr;

// ---- equal elevations: the flood on UNIQUE RANKS (r04) ----------------------------------------------------------------------
// What this engine can do exactly is a tie-free raster; so a raster with twins is replaced by a rank permutation -- position
// in the order of (elevation, tie key), stable radix sorts of (key, cell) pairs -- and flooded exactly.  First tie key: the
// cell index (the reference's answer for the raster with its ties broken in raster order: at 6000^2 of the bench generator,
// 31 % distinct values, 731 cells differ from the compiled reference instead of the 123 165 of deciding ties late, inside
// the levels); then the discovery times the flood itself implies, to their fixed point (below).
template <class T>
__global__ __launch_bounds__(NT) void k_rank_keys(const T *__restrict__ z, uint32_t *keys, uint32_t *idx, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    keys[c] = Key32<T>::to(z[c]);
    idx[c] = (uint32_t)c;
  }
}
// census on the sorted keys (rasters below 2^26 cells: no 512 MB bitmap for a 200 x 200 DEM -- ADVICE r03): cells whose key
// equals their predecessor's, and cells holding the levels' wall value (see k_twins)
__global__ __launch_bounds__(NT) void k_sorted_twins(const uint32_t *__restrict__ skeys, uint64_t n, uint32_t wall_key,
                                                     unsigned long long *counters) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  uint32_t twins = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride)
    if ((i > 0 && skeys[i] == skeys[i - 1]) || skeys[i] >= wall_key) twins++;
  for (int o = 32; o > 0; o >>= 1) twins += __shfl_down(twins, o, 64);
  if ((threadIdx.x & 63) == 0 && twins) atomicAdd(&counters[(blockIdx.x & 63) * 2], (unsigned long long)twins);
}
__global__ __launch_bounds__(NT) void k_rank_scatter(const uint32_t *__restrict__ sidx, uint32_t *rk, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) rk[sidx[i]] = (uint32_t)i;
}

// ---- equal elevations, EXACTLY (r04): the stable queue's order as a fixed point ---------------------------------------------
// The reference pops equal elevations in the order they were pushed: cell c's tie key is its DISCOVERY time
//     tau(c) = (pop rank R of the cell that closed c, position of c among that cell's pushes in d8_order),
// border cells first, in the order of the reference's two set-up loops (:508-528).  For a tie-free raster the flood's pop
// order has a closed form on the tree of directions: with parent'(c) = c's nearest ancestor of GREATER elevation, a cell's
// key of nested fill levels (header) is the path root -> c of the tree T' so defined, and the lexicographic order of those
// keys, a prefix first, is the PREORDER of T' with children sorted by elevation.  So: flood the unique ranks of (z, tau)
// exactly, read R off the result, form the discovery times, rank again, until the ranks stop moving.  A fixed point is the
// reference's order: if the first m pops of a pass are the reference's, the queue after them holds the same cells with the
// same parents, their new tie keys are their true discovery times, and pop m + 1 is the reference's as well -- and since
// every pass is an exact flood of SOME order, the passes converge from the front.  Passes needed: the longest chain of tie
// decisions that depend on earlier ones -- 2 on float terrain, the breadth-first depth of the largest plateau on integer
// DEMs (tests/test_pfdirs_gpu.py asserts == the compiled reference on the reference's own tie-heavy vectors).
constexpr uint32_t T_ROOT = 0xFFFFFFFFu;

// parent cell in the tree of directions (T_ROOT: a border cell -- pushed before the flood starts)
__global__ __launch_bounds__(NT) void k_tie_parents(const uint8_t *__restrict__ dirs, uint32_t *par, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    const int d = dirs[c];
    const bool border = x == 0 || y == 0 || x == w - 1 || y == h - 1;
    par[c] = (border || d == 0) ? T_ROOT : (uint32_t)((size_t)(y + ndy(d)) * w + (x + ndx(d)));
  }
}

// one round of "nearest ancestor of greater rank" by pointer jumping: every ancestor strictly between c and g[c] has a
// smaller rank than c, so while g[c] is smaller too, g[g[c]] -- whatever another lane has made of it meanwhile -- is the
// next candidate
// (r05) A cell that has found its ancestor marks the pointer (bit 31: cell indices stay below 2^31, T_ROOT carries the bit
// anyway): from then on a round costs it one load instead of three, one of them a gather.  k_tie_g_done strips the marks.
constexpr uint32_t G_DONE = 0x80000000u;
__global__ __launch_bounds__(NT) void k_tie_greater(const uint32_t *__restrict__ rk, uint32_t *g, uint64_t n, uint32_t *changed) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  bool ch = false;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    uint32_t a = g[c];
    if (a & G_DONE) continue;   // the root, or found in an earlier round
    const uint32_t r = rk[c];
    bool done = false;
    for (int hops = 0; hops < 8; hops++) {
      if (rk[a] >= r) { done = true; break; }
      const uint32_t na = __hip_atomic_load(&g[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (na == T_ROOT) { a = T_ROOT; done = true; break; }
      a = na & ~G_DONE;
    }
    __hip_atomic_store(&g[c], done ? (a | G_DONE) : a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ch |= !done;
  }
  if (__any(ch) && (threadIdx.x & 63) == 0) *changed = 1;
}
__global__ __launch_bounds__(NT) void k_tie_g_done(uint32_t *g, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t a = g[c];
    if (a != T_ROOT) g[c] = a & ~G_DONE;
  }
}

// r05: where the search starts.  A cell the PLAIN FILL of the rank raster does not raise has no ancestor of greater rank at all
// -- un-raised cells pop in increasing rank, and a wet ancestor lies below the cell that flooded its pocket, which popped
// before c, hence below c -- yet these cells (half of S3) have the longest chains to walk: to the raster's border.  They start
// at the root; the others at their parent.  (k_tie_greater: 239 rounds, 5.3 s of the function's 18 at S3 before this.)
__global__ __launch_bounds__(NT) void k_tie_g_init(const uint32_t *__restrict__ par, const uint32_t *__restrict__ rk,
                                                   const uint32_t *__restrict__ F0, uint32_t *g, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) g[c] = F0[c] == rk[c] ? T_ROOT : par[c];
}

// r05 (tree iteration, see pf_flowdirs_device): when the tree of directions is NOT the exact flood of the ranks, "the plain fill
// leaves c alone" says nothing about c's ancestors in it.  M(c) = the greatest rank among c's proper ancestors, by pointer
// doubling IN PLACE on one packed word per cell (high: the maximum over the ancestors up to and including `a`; low: a) -- an
// 8-byte load is one consistent (maximum, pointer) pair whatever other lanes have done to it meanwhile.
__global__ __launch_bounds__(NT) void k_tie_maxanc_init(const uint32_t *__restrict__ par, const uint32_t *__restrict__ rk,
                                                        unsigned long long *st, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t p = par[c];
    st[c] = ((unsigned long long)(p == T_ROOT ? 0u : rk[p]) << 32) | p;
  }
}
__global__ __launch_bounds__(NT) void k_tie_maxanc(unsigned long long *st, uint64_t n, uint32_t *changed) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  bool ch = false;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    unsigned long long v = st[c];
    uint32_t a = (uint32_t)v, m = (uint32_t)(v >> 32);
    if (a == T_ROOT) continue;
    for (int hops = 0; hops < 4 && a != T_ROOT; hops++) {
      const unsigned long long u = __hip_atomic_load(&st[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      m = max(m, (uint32_t)(u >> 32));
      a = (uint32_t)u;
    }
    __hip_atomic_store(&st[c], ((unsigned long long)m << 32) | a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ch |= a != T_ROOT;
  }
  if (__any(ch) && (threadIdx.x & 63) == 0) *changed = 1;
}
__global__ __launch_bounds__(NT) void k_tie_g_init_maxanc(const uint32_t *__restrict__ par, const uint32_t *__restrict__ rk,
                                                          const unsigned long long *__restrict__ st, uint32_t *g, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride)
    g[c] = (uint32_t)(st[c] >> 32) < rk[c] ? T_ROOT : par[c];   // (a border cell: no ancestor, maximum 0 -- and its parent is the root anyway)
}

// r05: the record tree straight from an exact flood's levels (see g_last_level): cell of every rank, then parent' = the cell
// whose rank is the level of c's innermost pocket
__global__ __launch_bounds__(NT) void k_tie_rank_cells(const uint32_t *__restrict__ rk, uint32_t *cell_of, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) cell_of[rk[c]] = (uint32_t)c;
}
__global__ __launch_bounds__(NT) void k_tie_from_levels(const uint32_t *__restrict__ last_level, const uint32_t *__restrict__ cell_of,
                                                        uint32_t *g, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t l = last_level[c];
    g[c] = l == T_ROOT ? T_ROOT : cell_of[l];
  }
}

// depth in T' by pointer doubling (ping-pong): anc / dist -> anc2 / dist2
__global__ __launch_bounds__(NT) void k_tie_depth(const uint32_t *__restrict__ anc, const uint32_t *__restrict__ dist, uint32_t *anc2,
                                                  uint32_t *dist2, uint64_t n, uint32_t *changed) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  bool ch = false;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const uint32_t a = anc[c];
    uint32_t d = dist[c], a2 = a;
    if (a != T_ROOT) { d += dist[a]; a2 = anc[a]; ch |= a2 != T_ROOT; }
    anc2[c] = a2;
    dist2[c] = d;
  }
  if (__any(ch) && (threadIdx.x & 63) == 0) *changed = 1;
}

__global__ __launch_bounds__(NT) void k_tie_init(const uint32_t *__restrict__ g, uint32_t *anc, uint32_t *dist, uint32_t *size,
                                                 uint32_t *cells, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    anc[c] = g[c];
    dist[c] = 1;          // depth 1: a child of the virtual root
    size[c] = 1;
    cells[c] = (uint32_t)c;
  }
}

// first index of every depth in the cells sorted by depth: start[d] for d = 1 .. maxd, start[maxd + 1] = n
__global__ __launch_bounds__(NT) void k_tie_bucket_starts(const uint32_t *__restrict__ sdepth, uint64_t n, uint32_t *start) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    const uint32_t d = sdepth[i], dp = i ? sdepth[i - 1] : 0u;
    for (uint32_t k = dp + 1; k <= d; k++) start[k] = (uint32_t)i;
    if (i == n - 1) start[d + 1] = (uint32_t)n;
  }
}

__global__ __launch_bounds__(NT) void k_tie_sizes_up(const uint32_t *__restrict__ cells, uint32_t lo, uint32_t hi,
                                                     const uint32_t *__restrict__ g, uint32_t *size) {
  const uint32_t i = lo + blockIdx.x * NT + threadIdx.x;
  if (i >= hi) return;
  const uint32_t c = cells[i];
  atomicAdd(&size[g[c]], size[c]);   // (depth >= 2: g[c] is a cell)
}

// sibling order: key = (parent' + 1) << 32 | own rank
__global__ __launch_bounds__(NT) void k_tie_sibling_keys(const uint32_t *__restrict__ g, const uint32_t *__restrict__ rk,
                                                         unsigned long long *keys, uint32_t *cells, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    keys[c] = ((unsigned long long)(g[c] + 1u) << 32) | rk[c];
    cells[c] = (uint32_t)c;
  }
}
// in sibling order: the subtree size of every element (for the scan) and, where a run of siblings starts, its index
__global__ __launch_bounds__(NT) void k_tie_gather_sizes(const unsigned long long *__restrict__ skeys, const uint32_t *__restrict__ scells,
                                                         const uint32_t *__restrict__ size, unsigned long long *ssize, uint32_t *head,
                                                         uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    ssize[i] = size[scells[i]];
    head[i] = (i == 0 || (skeys[i] >> 32) != (skeys[i - 1] >> 32)) ? (uint32_t)i : 0u;
  }
}
// off[c] = subtree sizes of c's smaller siblings
__global__ __launch_bounds__(NT) void k_tie_offsets(const uint32_t *__restrict__ scells, const unsigned long long *__restrict__ scan,
                                                    const uint32_t *__restrict__ headpos, uint32_t *off, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) off[scells[i]] = (uint32_t)(scan[i] - scan[headpos[i]]);
}
// preorder rank, one depth at a time from the top: R(c) = R(parent') + 1 + off(c); children of the virtual root: off(c)
// (in place: R[c] holds off(c) until c's depth is processed; its parent' is shallower, i.e. a rank already)
__global__ __launch_bounds__(NT) void k_tie_ranks_down(const uint32_t *__restrict__ cells, uint32_t lo, uint32_t hi,
                                                       const uint32_t *__restrict__ g, uint32_t *R) {
  const uint32_t i = lo + blockIdx.x * NT + threadIdx.x;
  if (i >= hi) return;
  const uint32_t c = cells[i], p = g[c];
  R[c] = (p == T_ROOT ? 0u : R[p] + 1u) + R[c];
}

// discovery time: border cells in the order of the reference's set-up loops (:508-519: for x: (x, 0), (x, h - 1); for
// y = 1 .. h - 2: (0, y), (w - 1, y)), then (pop rank of the closing cell, position among its pushes in d8_order)
__global__ __launch_bounds__(NT) void k_tie_tau(const uint8_t *__restrict__ dirs, const uint32_t *__restrict__ par,
                                                const uint32_t *__restrict__ R, unsigned long long *tau, uint32_t *cells, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  const unsigned long long nb = 2ull * w + 2ull * h;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    unsigned long long t;
    if (y == 0) t = 2ull * x;
    else if (y == h - 1) t = 2ull * x + 1;
    else if (x == 0) t = 2ull * w + 2ull * (y - 1);
    else if (x == w - 1) t = 2ull * w + 2ull * (y - 1) + 1;
    else {
      const int d = dirs[c];                                  // towards the closing cell; it pushed c in direction inverse(d)
      const int inv = d == 0 ? 0 : ((d + 3) & 7) + 1;         // d8_inverse: 1<->5, 2<->6, 3<->7, 4<->8
      const int pos = (inv & 1) ? (inv - 1) >> 1 : 4 + ((inv - 2) >> 1);   // d8_order = 1,3,5,7,2,4,6,8
      const uint32_t p = par[c];
      t = nb + (p == T_ROOT ? 0ull : ((unsigned long long)R[p] + 1ull) * 8ull + (unsigned long long)pos);
    }
    tau[c] = t;
    cells[c] = (uint32_t)c;
  }
}
// r05, the tree iteration's step: the flood pushes a cell when the FIRST of its neighbours is popped, so under the pop ranks R the
// tree of directions must be D(c) = the neighbour of least R.  Rewrites the directions of the interior cells to that (counting
// the cells that change) and forms the discovery times from it: k_tie_tau with the parents read off R.
__global__ __launch_bounds__(NT) void k_tie_tau_argmin(uint8_t *dirs, const uint32_t *__restrict__ R, unsigned long long *tau, uint32_t *cells,
                                                       int w, int h, unsigned long long *changed) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  const unsigned long long nb = 2ull * w + 2ull * h;
  uint32_t m = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    unsigned long long t;
    if (y == 0) t = 2ull * x;
    else if (y == h - 1) t = 2ull * x + 1;
    else if (x == 0) t = 2ull * w + 2ull * (y - 1);
    else if (x == w - 1) t = 2ull * w + 2ull * (y - 1) + 1;
    else {
      uint32_t best = 0xFFFFFFFFu;
      int bd = 1;
#pragma unroll
      for (int d = 1; d <= 8; d++) {   // (pop ranks are distinct: no tie to break)
        const uint32_t r = R[(size_t)(y + ndy(d)) * w + (x + ndx(d))];
        if (r < best) { best = r; bd = d; }
      }
      m += dirs[c] != (uint8_t)bd;
      dirs[c] = (uint8_t)bd;
      const int inv = ((bd + 3) & 7) + 1;                                   // the closing cell pushed c in direction inverse(bd)
      const int pos = (inv & 1) ? (inv - 1) >> 1 : 4 + ((inv - 2) >> 1);   // d8_order = 1,3,5,7,2,4,6,8
      t = nb + ((unsigned long long)best + 1ull) * 8ull + (unsigned long long)pos;
    }
    tau[c] = t;
    cells[c] = (uint32_t)c;
  }
  for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&changed[(blockIdx.x & 63) * 2 + 1], (unsigned long long)m);
}
// first tie key, before any flood: a cell on a slope is most likely closed by its LOWEST neighbour, so equal cells are
// ordered by (that neighbour's key, position among its pushes) -- closer to the insertion order than the cell index, one
// pass of the fixed point saved where ties are local (float terrain); any start converges
__global__ __launch_bounds__(NT) void k_tie_tau0(const uint32_t *__restrict__ zkey, unsigned long long *tau, uint32_t *cells, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  const unsigned long long nb = 2ull * w + 2ull * h;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    unsigned long long t;
    if (y == 0) t = 2ull * x;
    else if (y == h - 1) t = 2ull * x + 1;
    else if (x == 0) t = 2ull * w + 2ull * (y - 1);
    else if (x == w - 1) t = 2ull * w + 2ull * (y - 1) + 1;
    else {
      uint32_t bk = 0xFFFFFFFFu;
      int bd = 1;
#pragma unroll
      for (int k = 1; k <= 8; k++) {   // in d8_order, so that among equal lowest neighbours the first pusher wins
        const int d = k <= 4 ? 2 * k - 1 : 2 * (k - 4);
        const uint32_t kk = zkey[(size_t)(y + ndy(d)) * w + (x + ndx(d))];
        if (kk < bk) { bk = kk; bd = d; }
      }
      const int inv = ((bd + 3) & 7) + 1;                                   // the direction in which that neighbour pushes c
      const int pos = (inv & 1) ? (inv - 1) >> 1 : 4 + ((inv - 2) >> 1);
      t = nb + (unsigned long long)bk * 8ull + (unsigned long long)pos;
    }
    tau[c] = t;
    cells[c] = (uint32_t)c;
  }
}
__global__ __launch_bounds__(NT) void k_tie_gather_keys(const uint32_t *__restrict__ zkey, const uint32_t *__restrict__ order,
                                                        uint32_t *out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) out[i] = zkey[order[i]];
}
// new ranks from the final order; counts the cells whose rank moved
__global__ __launch_bounds__(NT) void k_tie_new_ranks(const uint32_t *__restrict__ order, const uint32_t *__restrict__ rk_old,
                                                      uint32_t *rk_new, uint64_t n, unsigned long long *moved) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  uint32_t m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
    const uint32_t c = order[i];
    rk_new[c] = (uint32_t)i;
    m += rk_old[c] != (uint32_t)i;
  }
  for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&moved[(blockIdx.x & 63) * 2], (unsigned long long)m);
}
struct MaxU32 {
  __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

// interior NoData cells carry no direction (:545-548); the rank pass cannot see which cells those are
template <class T>
__global__ __launch_bounds__(NT) void k_nodata_dirs(const T *__restrict__ z, T nodata, uint8_t *dirs, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NT;
  for (uint64_t c = (uint64_t)blockIdx.x * NT + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    if (x > 0 && y > 0 && x < w - 1 && y < h - 1 && z[c] == nodata) dirs[c] = 0;
  }
}

template <class T>
struct Calls;
#define RD_CALLS(SUF, T)                                                                                             \
  template <>                                                                                                        \
  struct Calls<T> {                                                                                                  \
    static int fill(T *d, int w, int h, void *s) { return rdgpu_fill_dev_##SUF(d, w, h, 8, s); }                     \
    static int fill_outlets(T *d, const uint8_t *o, const uint8_t *k, const uint32_t *l, uint32_t st, const uint32_t *c3, int w,  \
                            int h, void *s) {                                                                        \
      return rdgpu_fill_outlets_lists_dev_##SUF(d, o, k, l, st, c3, w, h, s);                                        \
    }                                                                                                                \
  };
RD_CALLS(u8, uint8_t) RD_CALLS(i8, int8_t) RD_CALLS(i16, int16_t) RD_CALLS(u16, uint16_t) RD_CALLS(i32, int32_t)
RD_CALLS(u32, uint32_t) RD_CALLS(f32, float)
#undef RD_CALLS

static void check_rc(int rc) {
  if (rc != 0) throw Error(rc, std::string("rdgpu_pf_flowdirs: ") + rdgpu_last_error());
}

template <class T>
void pf_flowdirs_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, hipStream_t s) {
  if (!d_z || !d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: raster too large");
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  g_stats = rdgpu_pf_flowdirs_stats{0, 0, 0, 0, 0};
  uint8_t *cand = ws.buf<uint8_t>("pfd.cand", n);
  unsigned long long *counters = ws.buf<unsigned long long>("pfd.counters", 128 + 4);
  unsigned long long *sums = counters + 128;   // [0] undecided, [1] wet, [2] unresolved
  RD_HIP(hipMemsetAsync(counters, 0, (128 + 4) * sizeof(unsigned long long), s));
  bool sorted = false;   // (keys, cells) are already sorted: the census of a small raster
  uint32_t *keys = nullptr, *skeys = nullptr, *idx = nullptr, *sidx = nullptr;
  // Scratch of the rank machinery: six persistent 4-byte arrays (keys, ranks, parents, record-tree parents, sizes / pop
  // ranks, cells by depth) + ONE pool of 24 bytes per cell whose three 8-byte thirds are handed from phase to phase (the
  // first version named every array: 100 bytes per cell, 160 GB at 40000^2, which does not fit beside a caller's rasters)
  uint8_t *pool = nullptr;
  auto pool4 = [&](int third, int half) { return reinterpret_cast<uint32_t *>(pool + ((size_t)third * 2 + half) * n * 4); };
  auto pool8 = [&](int third) { return reinterpret_cast<unsigned long long *>(pool + (size_t)third * n * 8); };
  auto sort_cells = [&]() {   // (LSD radix sort: stable, so equal keys stay in raster order)
    pool = ws.buf<uint8_t>("pfd.t.pool", (size_t)n * 24);
    keys = ws.buf<uint32_t>("pfd.rkeys", n); skeys = pool4(0, 0);
    idx = pool4(0, 1); sidx = pool4(1, 0);
    RD_LAUNCH("pfd.rank_keys", (k_rank_keys<T>), dim3(sgrid(n)), dim3(NT), 0, s, d_z, keys, idx, n);
    const char *ti = getenv("RDGPU_PFD_TIE_INIT");   // =0: equal cells start in raster order
    size_t tb = 0;
    if (!(ti && ti[0] == '0') && w > 2 && h > 2) {
      // equal cells start in the order of (lowest neighbour's key, push position): sort by that, then stably by the key
      unsigned long long *tau = pool8(1), *stau = pool8(2);
      uint32_t *c0 = pool4(0, 1), *c1 = pool4(0, 0);   // (idx = pool4(0, 1) is overwritten: the cells again)
      RD_LAUNCH("pfd.tie.tau0", k_tie_tau0, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)keys, tau, c0, w, h);
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, tau, stau, c0, c1, (int)n, 0, 36, s));
      void *tmp0 = ws.buf("pfd.rtmp", tb);
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp0, tb, tau, stau, c0, c1, (int)n, 0, 36, s));
      uint32_t *zk = pool4(1, 1);   // (tau = pool8(1) is dead)
      RD_LAUNCH("pfd.tie.gather_keys", k_tie_gather_keys, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)keys, (const uint32_t *)c1, zk, n);
      skeys = pool4(2, 0);
      sidx = pool4(2, 1);           // (stau = pool8(2) is dead)
      tb = 0;
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, zk, skeys, c1, sidx, (int)n, 0, 32, s));
      void *tmp1 = ws.buf("pfd.rtmp", tb);
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp1, tb, zk, skeys, c1, sidx, (int)n, 0, 32, s));
    } else {
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, skeys, idx, sidx, (int)n, 0, 32, s));
      void *tmp = ws.buf("pfd.rtmp", tb);
      RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, skeys, idx, sidx, (int)n, 0, 32, s));
    }
    sorted = true;
  };
  {
    const char *env = getenv("RDGPU_PFD_TWINS");   // =0: skip the equal-elevation census
    if (!(env && env[0] == '0') && !g_rank_pass && n < ((uint64_t)1 << 26)) {
      sort_cells();
      RD_LAUNCH("pfd.sorted_twins", k_sorted_twins, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)skeys, n,
                Key32<T>::to(wall_value<T>()), counters);
      RD_LAUNCH("pfd.sum", k_sum_counters, dim3(1), dim3(128), 0, s, counters, sums);
      unsigned long long tw = 0;
      RD_HIP(hipMemcpyAsync(&tw, sums, sizeof tw, hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      g_stats.twins = (uint32_t)std::min<unsigned long long>(tw, 0xFFFFFFFFull);
    } else if (!(env && env[0] == '0') && !g_rank_pass) {   // 512 MB of bits: small beside a raster of >= 2^26 cells
      uint32_t *bits = ws.buf<uint32_t>("pfd.keybits", (size_t)1 << 27);
      RD_HIP(hipMemsetAsync(bits, 0, (size_t)1 << 29, s));
      RD_LAUNCH("pfd.twins", (k_twins<T>), dim3(sgrid(n)), dim3(NT), 0, s, d_z, n, bits, counters);
      RD_LAUNCH("pfd.sum", k_sum_counters, dim3(1), dim3(128), 0, s, counters, sums);
      unsigned long long tw = 0;
      RD_HIP(hipMemcpyAsync(&tw, sums, sizeof tw, hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      g_stats.twins = (uint32_t)std::min<unsigned long long>(tw, 0xFFFFFFFFull);
    }
  }
  {
    const char *re = getenv("RDGPU_PFD_RANKS");   // =0: ties decided inside the levels, by neighbour number (r03; A/B and tests)
    if (g_stats.twins != 0 && !g_rank_pass && !(re && re[0] == '0')) {
      if (!sorted) sort_cells();
      uint32_t *rk = ws.buf<uint32_t>("pfd.rk", n);
      RD_LAUNCH("pfd.rank_scatter", k_rank_scatter, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)sidx, rk, n);
      const rdgpu_pf_flowdirs_stats mine = g_stats;
      const char *tp = getenv("RDGPU_PFD_TIE_PASSES");   // passes of the tie order's fixed point (0: raster order, r04's first version)
      // Each pass settles one more "generation" of tie decisions: 2 - 4 on float terrain, but one breadth-first RING of the
      // largest plateau on integer DEMs -- min(w, h) / 2 passes for an ocean.  Bounded by a COUNT (RDGPU_PFD_TIE_PASSES,
      // default 1000): deterministic, the same DEM gives the same raster on every machine.  A wall-time bound is opt-in
      // (RDGPU_PFD_TIE_SECONDS=<s>, checked between passes; ADVICE r05: a result cut off by the clock is not reproducible
      // across machines or loads, so it is never the default);
      // when a bound stops the iteration the result is an exact flood of SOME stable order, `unresolved` says how many
      // ranks were still moving and the host wrappers warn.
      const uint32_t max_passes = (w <= 2 || h <= 2) ? 0u : tp ? (uint32_t)strtoul(tp, nullptr, 10) : 1000u;   // (no interior cell: no tie to order)
      const char *tsec = getenv("RDGPU_PFD_TIE_SECONDS");
      const double max_seconds = tsec ? atof(tsec) : std::numeric_limits<double>::infinity();
      const auto t_tie0 = std::chrono::steady_clock::now();
      uint32_t passes = 0, levels_total = 0;
      unsigned long long moved = (w <= 2 || h <= 2) ? 0ull : (unsigned long long)g_stats.twins;   // no re-rank pass ran: every twin's place is undecided
      unsigned long long dchanged = 0;
      // r05, THE TREE ITERATION (the default; RDGPU_PFD_TREE_ITER=0: a level flood per pass, r04).  Only the first pass floods by
      // levels.  For ANY tree of directions D and distinct ranks r the
      // machinery below yields R = the order in which a priority queue walks D (pop the least rank, push its children), and the
      // real flood pushes a cell when the first of its neighbours pops: if D(c) = argmin over c's neighbours of R -- checked and
      // enforced by k_tie_tau_argmin -- queue walk and flood have the same frontier after every pop, by induction the same
      // order, and D IS the flood's tree for r.  So the state (D, r) is iterated: R from (D, r); D' = argmin R; the discovery
      // times from (D', R) and r' from them; "no direction changed and no rank moved" certifies the reference's result.  It
      // converges from the front like the passes did: while the first m pops are the reference's, every cell next to one of
      // them gets its true parent and its true discovery time, so pop m + 1 is right in the next iteration (D' stays a forest:
      // a cell's old parent pops before it, so the least R among its neighbours is below its own).  The ranks settle as fast as
      // with a flood per pass (S3: 1.48e9, 101 590, 6, 0 moved), the tree lags behind them (3484, 1845, 802, ... directions
      // still changing, halving per iteration): once (almost) no rank moves -- <= max(64, n / 2^24), RDGPU_PFD_TREE_REFLOOD_MOVED --
      // ONE level flood brings the tree up to date and the next iteration is the check.  A level flood is also run when a bound
      // stops the iteration (the result is then an exact flood of some stable order, as before) and every
      // RDGPU_PFD_TREE_REFLOOD iterations (default 12); at worst every iteration floods: the r04 scheme.  S3
      // (profiles/r05q_pfd_tree_ab.txt): 918 - 1377 levels instead of 1836, 15.1 -> 12.4 s with the flood at "no rank moved".
      const char *tie_env = getenv("RDGPU_PFD_TREE_ITER");
      const bool tree_iter = !(tie_env && tie_env[0] == '0');
      const char *rm_env = getenv("RDGPU_PFD_TREE_REFLOOD_MOVED");
      const unsigned long long reflood_moved = rm_env ? strtoull(rm_env, nullptr, 10) : std::max<unsigned long long>(64ull, n >> 24);
      const char *rf_env = getenv("RDGPU_PFD_TREE_REFLOOD");
      const uint32_t reflood = rf_env ? (uint32_t)strtoul(rf_env, nullptr, 10) : 12u;
      bool exact = false;          // d_dirs is the exact flood of rk
      uint32_t since_flood = 0;
      const char *ll_env = getenv("RDGPU_PFD_LEVEL_TREE");   // =0: the record tree by the ancestor search also after an exact flood (A/B, tests)
      uint32_t *last_level = (ll_env && ll_env[0] == '0') ? nullptr : ws.buf<uint32_t>("pfd.t.last_level", n);
      g_last_level = last_level;
      g_rank_pass = true;
      try {
        for (;;) {
          const bool stop = passes >= max_passes ||
                            (passes && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tie0).count() > max_seconds);
          if (!exact && (passes == 0 || !tree_iter || stop || since_flood >= reflood)) {
            pf_flowdirs_device<uint32_t>(rk, 0xFFFFFFFFu, w, h, d_dirs, s);   // (no rank is 2^32 - 1: n < 2^31)
            levels_total += g_stats.levels;
            exact = true;
            since_flood = 0;
          }
          if (stop) break;
          passes++;
          // ---- the discovery times under this pass's flood, and the ranks of (z, discovery time) -----------------------
          uint32_t *zkey = keys;   // (keys still holds every cell's key: k_rank_keys' output, untouched by the sort)
          uint32_t *par = ws.buf<uint32_t>("pfd.t.par", n), *g = ws.buf<uint32_t>("pfd.t.g", n);
          uint32_t *size = ws.buf<uint32_t>("pfd.t.size", n), *bcells = ws.buf<uint32_t>("pfd.t.bcells", n);
          uint32_t *flag = ws.buf<uint32_t>("pfd.t.flag", 4);
          uint32_t *hw = ws.host_words();
          RD_LAUNCH("pfd.tie.parents", k_tie_parents, dim3(sgrid(n)), dim3(NT), 0, s, (const uint8_t *)d_dirs, par, w, h);
          const char *gi = getenv("RDGPU_PFD_GREATER_INIT");   // =0: every search starts at the parent (r04): A/B and tests
          const bool from_levels = exact && last_level != nullptr;
          if (from_levels) {   // the exact flood's own levels name every cell's nearest greater ancestor: no search
            uint32_t *cell_of = pool4(0, 0);
            RD_LAUNCH("pfd.tie.rank_cells", k_tie_rank_cells, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)rk, cell_of, n);
            RD_LAUNCH("pfd.tie.from_levels", k_tie_from_levels, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)last_level,
                      (const uint32_t *)cell_of, g, n);
          } else if (gi && gi[0] == '0') {
            RD_HIP(hipMemcpyAsync(g, par, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
          } else if (!exact) {   // any tree: the greatest rank among a cell's ancestors says whether it has a greater one
            unsigned long long *st = pool8(0);
            RD_LAUNCH("pfd.tie.maxanc_init", k_tie_maxanc_init, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)par, (const uint32_t *)rk, st, n);
            for (int round = 0;; round++) {
              RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
              RD_LAUNCH("pfd.tie.maxanc", k_tie_maxanc, dim3(sgrid(n)), dim3(NT), 0, s, st, n, flag);
              RD_HIP(hipMemcpyAsync(hw, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
              RD_HIP(hipStreamSynchronize(s));
              if (hw[0] == 0) break;
              if (round > 64) throw Error(RDGPU_ERR_HIP, "rdgpu_pf_flowdirs: the tree of directions holds a loop (internal error)");
            }
            RD_LAUNCH("pfd.tie.g_init", k_tie_g_init_maxanc, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)par, (const uint32_t *)rk,
                      (const unsigned long long *)st, g, n);
          } else {   // the exact flood's tree: one plain fill of the rank raster (the pool is free here) says which cells have no greater ancestor
            uint32_t *F0 = pool4(0, 0);
            RD_HIP(hipMemcpyAsync(F0, rk, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            check_rc(Calls<uint32_t>::fill(F0, w, h, (void *)s));
            RD_LAUNCH("pfd.tie.g_init", k_tie_g_init, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)par, (const uint32_t *)rk, (const uint32_t *)F0,
                      g, n);
          }
          for (int round = 0; !from_levels; round++) {   // nearest ancestor of greater elevation
            RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
            RD_LAUNCH("pfd.tie.greater", k_tie_greater, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)rk, g, n, flag);
            RD_HIP(hipMemcpyAsync(hw, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            RD_HIP(hipStreamSynchronize(s));
            if (hw[0] == 0) break;
            if (round > 10000) throw Error(RDGPU_ERR_HIP, "rdgpu_pf_flowdirs: the record tree did not settle (internal error)");
          }
          if (!from_levels) RD_LAUNCH("pfd.tie.g_done", k_tie_g_done, dim3(sgrid(n)), dim3(NT), 0, s, g, n);
          // ---- depth in T' (pointer doubling), cells bucketed by depth: pool thirds 0 / 1 ping-pong, 2 holds the cells ----
          uint32_t *ancA = pool4(0, 0), *dstA = pool4(0, 1), *ancB = pool4(1, 0), *dstB = pool4(1, 1), *cellsA = pool4(2, 0);
          RD_LAUNCH("pfd.tie.init", k_tie_init, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)g, ancA, dstA, size, cellsA, n);
          for (int round = 0;; round++) {
            RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
            RD_LAUNCH("pfd.tie.depth", k_tie_depth, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)ancA, (const uint32_t *)dstA, ancB,
                      dstB, n, flag);
            std::swap(ancA, ancB);
            std::swap(dstA, dstB);
            RD_HIP(hipMemcpyAsync(hw, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            RD_HIP(hipStreamSynchronize(s));
            if (hw[0] == 0) break;
            if (round > 64) throw Error(RDGPU_ERR_HIP, "rdgpu_pf_flowdirs: depth doubling did not settle (internal error)");
          }
          uint32_t *depth = dstA, *sdepth = dstB;
          size_t tb = 0;
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, depth, sdepth, cellsA, bcells, (int)n, 0, 32, s));
          void *tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, depth, sdepth, cellsA, bcells, (int)n, 0, 32, s));
          RD_HIP(hipMemcpyAsync(hw, sdepth + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
          RD_HIP(hipStreamSynchronize(s));
          const uint32_t maxd = hw[0];
          uint32_t *dstart = ws.buf<uint32_t>("pfd.t.dstart", (size_t)maxd + 2);
          RD_LAUNCH("pfd.tie.buckets", k_tie_bucket_starts, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)sdepth, n, dstart);
          std::vector<uint32_t> hstart((size_t)maxd + 2);
          RD_HIP(hipMemcpyAsync(hstart.data(), dstart, ((size_t)maxd + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
          RD_HIP(hipStreamSynchronize(s));
          for (uint32_t d = maxd; d >= 2; d--) {   // subtree sizes, deepest first
            const uint32_t lo = hstart[d], hi = hstart[d + 1];
            if (hi > lo)
              RD_LAUNCH("pfd.tie.sizes", k_tie_sizes_up, dim3((hi - lo + NT - 1) / NT), dim3(NT), 0, s, (const uint32_t *)bcells, lo, hi,
                        (const uint32_t *)g, size);
          }
          // ---- siblings in order of elevation; off(c) = subtree sizes of c's smaller siblings (the whole pool is free) ----
          unsigned long long *k64 = pool8(0), *sk64 = pool8(1);
          uint32_t *ucells = pool4(2, 0), *scells = pool4(2, 1);
          RD_LAUNCH("pfd.tie.sibling_keys", k_tie_sibling_keys, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)g, (const uint32_t *)rk, k64,
                    ucells, n);
          tb = 0;
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k64, sk64, ucells, scells, (int)n, 0, 64, s));
          tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k64, sk64, ucells, scells, (int)n, 0, 64, s));
          unsigned long long *ssize = k64;   // (the unsorted keys are dead); scanned in place
          uint32_t *head = ucells;           // (the unsorted cells are dead); scanned in place
          RD_LAUNCH("pfd.tie.gather_sizes", k_tie_gather_sizes, dim3(sgrid(n)), dim3(NT), 0, s, (const unsigned long long *)sk64,
                    (const uint32_t *)scells, (const uint32_t *)size, ssize, head, n);
          tb = 0;
          RD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ssize, ssize, (int)n, s));
          tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, ssize, ssize, (int)n, s));
          tb = 0;
          RD_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tb, head, head, MaxU32(), (int)n, s));
          tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceScan::InclusiveScan(tmp, tb, head, head, MaxU32(), (int)n, s));
          uint32_t *R = size;   // offsets first, pop ranks in place (the sizes are dead: ssize holds what was needed of them)
          RD_LAUNCH("pfd.tie.offsets", k_tie_offsets, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)scells,
                    (const unsigned long long *)ssize, (const uint32_t *)head, R, n);
          for (uint32_t d = 1; d <= maxd; d++) {   // preorder ranks, from the top: R(c) = R(parent') + 1 + off(c), in place
            const uint32_t lo = hstart[d], hi = hstart[d + 1];
            if (hi > lo)
              RD_LAUNCH("pfd.tie.ranks", k_tie_ranks_down, dim3((hi - lo + NT - 1) / NT), dim3(NT), 0, s, (const uint32_t *)bcells, lo, hi,
                        (const uint32_t *)g, R);
          }
          // ---- discovery times -> order by (elevation, discovery time): sort by the time, then stably by the key ----------
          unsigned long long *tau = sk64, *stau = k64;
          uint32_t *c0 = ucells, *c1 = scells;
          RD_HIP(hipMemsetAsync(counters, 0, 128 * sizeof(unsigned long long), s));
          if (tree_iter)
            RD_LAUNCH("pfd.tie.tau_argmin", k_tie_tau_argmin, dim3(sgrid(n)), dim3(NT), 0, s, d_dirs, (const uint32_t *)R, tau, c0, w, h, counters);
          else
            RD_LAUNCH("pfd.tie.tau", k_tie_tau, dim3(sgrid(n)), dim3(NT), 0, s, (const uint8_t *)d_dirs, (const uint32_t *)par, (const uint32_t *)R,
                      tau, c0, w, h);
          tb = 0;
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, tau, stau, c0, c1, (int)n, 0, 36, s));
          tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, tau, stau, c0, c1, (int)n, 0, 36, s));
          uint32_t *zk = pool4(1, 0), *zks = pool4(1, 1);   // (tau is dead) keys in discovery order
          RD_LAUNCH("pfd.tie.gather_keys", k_tie_gather_keys, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)zkey, (const uint32_t *)c1, zk, n);
          tb = 0;
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, zk, zks, c1, c0, (int)n, 0, 32, s));
          tmp = ws.buf("pfd.rtmp", tb);
          RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, zk, zks, c1, c0, (int)n, 0, 32, s));
          uint32_t *rk_new = par;   // (the parents are dead)
          RD_LAUNCH("pfd.tie.new_ranks", k_tie_new_ranks, dim3(sgrid(n)), dim3(NT), 0, s, (const uint32_t *)c0, (const uint32_t *)rk, rk_new, n,
                    counters);
          RD_LAUNCH("pfd.sum", k_sum_counters, dim3(1), dim3(128), 0, s, counters, sums);
          unsigned long long mv[2] = {0, 0};   // ranks that moved, directions that changed
          RD_HIP(hipMemcpyAsync(mv, sums, sizeof mv, hipMemcpyDeviceToHost, s));
          RD_HIP(hipStreamSynchronize(s));
          moved = mv[0];
          dchanged = tree_iter ? mv[1] : 0ull;
          if (getenv("RDGPU_PFD_TRACE"))
            fprintf(stderr, "pfd tie pass %u: %llu ranks moved, %llu directions changed (tree %s), record tree depth %u\n", passes, moved, dchanged,
                    exact ? "from the level flood" : "iterated", maxd);
          if (moved == 0 && dchanged == 0) break;   // tree and order reproduce themselves: they are the reference's
          if (moved) RD_HIP(hipMemcpyAsync(rk, rk_new, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
          exact = false;
          since_flood++;
          if (moved <= reflood_moved) since_flood = reflood;   // the order is (all but) final and only the tree lags: one level flood, then the check
        }
      } catch (...) {
        g_rank_pass = false;
        g_last_level = nullptr;
        throw;
      }
      g_rank_pass = false;
      g_last_level = nullptr;
      g_stats.twins = mine.twins;
      g_stats.levels = levels_total;
      g_stats.tie_passes = passes;
      g_stats.unresolved = moved ? moved : dchanged;   // ranks (or, with the order at rest, directions) still moving when the passes ran out (0: the reference's order; no pass at all: the twins)
      RD_LAUNCH("pfd.nodata_dirs", (k_nodata_dirs<T>), dim3(sgrid(n)), dim3(NT), 0, s, d_z, nodata, d_dirs, w, h);
      return;
    }
  }
  RD_LAUNCH("pfd.init", k_init_cand, dim3(sgrid(n)), dim3(NT), 0, s, cand, w, h);
  T *last_level = nullptr;
  if constexpr (std::is_same<T, uint32_t>::value) {
    if (g_rank_pass && g_last_level) {
      last_level = g_last_level;
      RD_HIP(hipMemsetAsync(last_level, 0xFF, n * sizeof(uint32_t), s));   // T_ROOT: never wet -- no ancestor of greater rank
    }
  }
  if (w > 2 && h > 2) {
    T *F = ws.buf<T>("pfd.level", n);
    uint8_t *outlet = ws.buf<uint8_t>("pfd.outlet", n);
    const uint32_t ftiles = (uint32_t)(((w + FT - 1) / FT) * ((h + FT - 1) / FT));
    uint8_t *skip = ws.buf<uint8_t>("pfd.skip", ftiles), *active = ws.buf<uint8_t>("pfd.active", ftiles);
    RD_HIP(hipMemsetAsync(skip, 0, ftiles, s));
    RD_HIP(hipMemsetAsync(active, 0, ftiles, s));
    const char *sparse_env = getenv("RDGPU_PFD_SPARSE");   // =0: every level's fill over the whole raster (A/B and tests)
    const bool sparse = !(sparse_env && sparse_env[0] == '0');
    const uint32_t ftx = (uint32_t)((w + FT - 1) / FT), scan_rows = (uint32_t)((h + 31) / 32), stride = 2 * ftiles;
    uint32_t *lists = ws.buf<uint32_t>("pfd.lists", 3 * (size_t)stride);
    uint32_t *lcounts = ws.buf<uint32_t>("pfd.lcounts", 4);
    uint32_t *cur_tiles = lists + 2 * (size_t)stride;   // the tiles the two kernels here walk: all of them at first
    uint32_t ncur = ftiles;
    RD_LAUNCH("pfd.all_tiles", k_all_tiles, dim3((ftiles + NT - 1) / NT), dim3(NT), 0, s, cur_tiles, ftiles);
    RD_HIP(hipMemcpyAsync(F, d_z, n * sizeof(T), hipMemcpyDeviceToDevice, s));
    check_rc(Calls<T>::fill(F, w, h, (void *)s));
    const bool trace = getenv("RDGPU_PFD_TRACE") != nullptr;   // per level: time, tiles, counts (stderr)
    const auto t_start = std::chrono::steady_clock::now();
    double t_last = 0;
    unsigned long long host[2] = {0, 0}, last_open = ~0ull, last_wet = ~0ull;
    uint32_t hcounts[3] = {0, 0, 0};
    for (;;) {
      g_stats.levels++;
      if (ncur) {
        RD_LAUNCH("pfd.refine", (k_refine<T>), dim3(ncur), dim3(NT), 0, s, d_z, (const T *)F, cand, w, h, counters, sums + 2,
                  (const uint32_t *)cur_tiles);
        RD_LAUNCH("pfd.next_level", (k_next_level<T>), dim3(ncur), dim3(NT), 0, s, d_z, F, outlet, active, w, h, counters,
                  (const uint32_t *)cur_tiles, last_level);
      }
      RD_HIP(hipMemsetAsync(lcounts, 0, 4 * sizeof(uint32_t), s));
      RD_LAUNCH("pfd.skip_state", k_skip_state, dim3((ftiles + NT - 1) / NT), dim3(NT), 0, s, skip, active, ftiles, ftx, scan_rows, lists,
                stride, lcounts);
      RD_LAUNCH("pfd.sum", k_sum_counters, dim3(1), dim3(128), 0, s, counters, sums);
      RD_HIP(hipMemcpyAsync(host, sums, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      RD_HIP(hipMemcpyAsync(hcounts, lcounts, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      if (trace) {
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        fprintf(stderr, "pfd level %u: %.2f ms  tiles walked %u  undecided %llu wet %llu  lists %u %u %u\n", g_stats.levels, now - t_last, ncur,
                host[0], host[1], hcounts[0], hcounts[1], hcounts[2]);
        t_last = now;
      }
      ncur = hcounts[2];
      if ((host[0] == 0 && !last_level) || host[1] == 0) break;      // every cell decided (the levels themselves are wanted: to the last wet cell) / nothing wet any more
      if (host[0] == last_open && host[1] == last_wet) {             // no progress: equal elevations (see the header)
        //  in a rank pass the levels are tie free, the wet count shrinks strictly and this exit cannot be taken with
        // wet cells left; if it ever were, last_level -- and the record tree read off it -- would be silently incomplete
        if (g_rank_pass && last_level && host[1] != 0)
          throw Error(RDGPU_ERR_HIP, "rdgpu_pf_flowdirs: the level flood of a rank raster stalled with wet cells left (internal error)");
        break;
      }
      last_open = host[0];
      last_wet = host[1];
      check_rc(Calls<T>::fill_outlets(F, outlet, skip, sparse ? lists : nullptr, stride, hcounts, w, h, (void *)s));
    }
  }
  RD_LAUNCH("pfd.finish", (k_finish<T>), dim3(sgrid(n)), dim3(NT), 0, s, d_z, nodata, (const uint8_t *)cand, d_dirs, w, h, sums + 2);
  unsigned long long unres = 0;
  RD_HIP(hipMemcpyAsync(&unres, sums + 2, sizeof unres, hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  g_stats.unresolved = unres;
}

template <class T>
static void pf_flowdirs_host(const T *dem, T nodata, int w, int h, uint8_t *dirs) {
  if (!dem || !dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  uint8_t *dd = Workspace::get().buf<uint8_t>("host.dirs", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  pf_flowdirs_device<T>(d, nodata, w, h, dd, nullptr);
  RD_HIP(hipMemcpy(dirs, dd, n, hipMemcpyDeviceToHost));
}

}  // namespace pfd
}  // namespace rdgpu

#define RD_PFD_API(SUF, T)                                                                                                  \
  extern "C" int rdgpu_pf_flowdirs_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs) {                             \
    return rdgpu::guarded([&] { rdgpu::pfd::pf_flowdirs_host<T>(dem, nodata, w, h, dirs); });                              \
  }                                                                                                                         \
  extern "C" int rdgpu_pf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int w, int h, uint8_t *d_dirs, void *stream) {       \
    return rdgpu::guarded([&] { rdgpu::pfd::pf_flowdirs_device<T>(d_dem, nodata, w, h, d_dirs, (hipStream_t)stream); });   \
  }
RD_PFD_API(u8, uint8_t)
RD_PFD_API(i8, int8_t)
RD_PFD_API(i16, int16_t)
RD_PFD_API(u16, uint16_t)
RD_PFD_API(i32, int32_t)
RD_PFD_API(u32, uint32_t)
RD_PFD_API(f32, float)

extern "C" int rdgpu_pf_flowdirs_get_stats(rdgpu_pf_flowdirs_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = rdgpu::pfd::g_stats;
  return 0;
}
