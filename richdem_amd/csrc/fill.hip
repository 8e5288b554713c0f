// fill.hip -- depression filling on MI355X: descent forest -> basins -> one raster pass -> Boruvka rounds on a pair list.
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
//                     itself in the (key, index) total order; border cells drain "OUT".  Paths are
//                     compressed inside a 64x64 tile in LDS, and pits get dense basin ids 0..B-1.
//   2. k_tile_label   lab[c] = basin of c (B = the outside): the few distinct pointers of a tile are chased
//                     to their pits once per tile (k_chase / k_label_cells: fallback for very long chains).
//                     W(c) = max(z(c), L[basin(c)]) with L = minimax pass height basin -> outside.
//   3. rounds of      round 1: k_scan<EMIT> (the one raster pass: the lowest pass of every PAIR of adjacent
//                     components, reduced per tile and appended to a list; each component's lowest pass to a
//                     different component = one 64-bit atomicMin of (pass height << 32 | neighbour component));
//                     rounds 2..: k_edge_round on the pair list (raster k_scan passes as overflow fallback);
//                     k_hook (hook every component along its lowest pass; mutual pairs keep the
//                     smaller id as root), k_chase_links (pointer jumping carrying the path
//                     maximum), k_update_basins, k_compact_roots.  This is Boruvka's contraction:
//                     the number of live components at least halves per round.
//   4. k_finalize     z(c) <- max(z(c), acc[lab[c]]).
//
// All elevation work is on order-preserving 32-bit keys (common.hpp Key32), so it is exact for
// u8/i16/u16/i32/u32/f32.  HBM-bound integer/compare work: no MFMA anywhere.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

struct rdgpu_fill_shard;

namespace rdgpu {

constexpr int TW = 64;        // tile width  (cells)  = one wavefront per tile row
constexpr int TH = 32;        // tile height (cells)
constexpr int LW = TW + 2;    // LDS row stride incl. 1-cell halo (66 words: conflict-free rows)
constexpr int LH = TH + 2;
constexpr int NTHR = 256;     // 4 wavefronts
constexpr uint32_t OUTP = 0xFFFFFFFFu;  // descent pointer of a border cell: drains off the raster
// Component ids carry CLOSED in the top bit when the component never hooks again: the outside (B) and,
// in a row-block shard, the frozen terminal basins of the cut rows (Barnes 2016 tile protocol).
constexpr uint32_t CLOSED = 0x80000000u;
constexpr uint32_t NO_TID = 0xFFFFFFFFu;   // tid[b]: basin b is not a cut-row terminal

static thread_local rdgpu_fill_stats g_stats;   // (per host thread: one thread per device may be inside the library)

// ------------------------------------------------------------------------------------------
// 1. descent pointers, path-compressed inside the tile
// A 64x64 tile (+1 halo) of keys is staged in LDS; every cell picks its descent target; pointers that
// stay inside the tile are chased to their tile-local root with LDS pointer jumping, so what reaches
// HBM already points at a pit, at OUT, or at the first cell OUTSIDE the tile on the cell's path.  The
// global chase (k_chase) then hops tile to tile instead of cell to cell.
// ------------------------------------------------------------------------------------------
constexpr int DW = 64, DH = 64, DLW = DW + 2, DLH = DH + 2;
constexpr uint32_t LAB_PEND = 0x80000000u;   // lab word not resolved yet: LAB_PEND | next cell on the path (OUTP: off the raster)
constexpr uint16_t LTERM_BASE = 0xF000u;   // local pointers >= this: the cell is terminal within the tile (low 4 bits: code)

// Four consecutive cells of a row starting at column gx in one load.  ALIGNED: the raster width is a multiple of 4
// and its base is aligned, so every quad of every row is naturally aligned (the case the compiler may assume and
// the fastest one: the 40000-wide bench DEM).  Otherwise the hardware still takes any element-aligned address for a
// 16 / 8 / 4-byte load (3 % slower when the data happens to be aligned), and the last quad of a row may hang over
// its end: those cells read as `fill`.
template <class U>
struct Quad { U v[4]; };
template <class U, bool ALIGNED>
__device__ __forceinline__ Quad<U> load_quad(const U *__restrict__ row, int gx, int w, U fill) {
  Quad<U> q;
  if (ALIGNED) {
    struct alignas(4 * sizeof(U)) AQ { U v[4]; };
    const AQ a = *reinterpret_cast<const AQ *>(row + gx);
#pragma unroll
    for (int e = 0; e < 4; e++) q.v[e] = a.v[e];
  } else if (gx + 3 < w) {
    __builtin_memcpy(&q, row + gx, sizeof(q));
  } else {
#pragma unroll
    for (int e = 0; e < 4; e++) q.v[e] = gx + e < w ? row[gx + e] : fill;
  }
  return q;
}

template <class T, int TOPO, bool VEC>
__global__ __launch_bounds__(NTHR) void k_descent(const T *__restrict__ z, uint32_t *__restrict__ lab,
                                                  uint32_t *pit_counter,
                                                  int w, int h, uint32_t tilesX, uint32_t ntiles, int open_top,
                                                  int open_bottom, const uint8_t *__restrict__ outlet) {
  // outlet (optional): cells that drain like border cells (rdgpu_fill_outlets_dev_*, pfdirs.hip)
  __shared__ uint32_t sk[DLH * DLW];
  __shared__ uint16_t lp[DH * DW];
  __shared__ uint32_t wtot[NTHR / 64];
  __shared__ uint32_t pbase;
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * DW, y0 = (int)(t / tilesX) * DH;
  {
    // interior columns with quad loads (VEC: naturally aligned quads, else any width -- see load_quad); all
    // loads of the thread are issued before the first is consumed (one memory round trip, not one per trip)
    constexpr int NQ = DLH * (DW / 4), QPT = (NQ + NTHR - 1) / NTHR;
    Quad<T> zq[QPT];   // four cells per load: 16 / 8 / 4 bytes for 4- / 2- / 1-byte elevations
    bool okq[QPT];
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = threadIdx.x + r * NTHR;
      const int ly = i / (DW / 4), q = i - ly * (DW / 4);
      const int gx = x0 + 4 * q, gy = y0 - 1 + ly;
      okq[r] = i < NQ && gy >= 0 && gy < h && gx < w;
      if (okq[r]) zq[r] = load_quad<T, VEC>(z + (size_t)gy * w, gx, w, T());
    }
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = threadIdx.x + r * NTHR;
      if (i >= NQ) continue;
      const int ly = i / (DW / 4), q = i - ly * (DW / 4);
      const int o = ly * DLW + 1 + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; e++) sk[o + e] = (okq[r] && x0 + 4 * q + e < w) ? Key32<T>::to(zq[r].v[e]) : 0xFFFFFFFFu;
    }
    for (int i = threadIdx.x; i < 2 * DLH; i += NTHR) {   // halo columns
      const int ly = i >> 1, lxh = (i & 1) ? DLW - 1 : 0;
      const int gx = x0 - 1 + lxh, gy = y0 - 1 + ly;
      uint32_t kk = 0xFFFFFFFFu;
      if (gx >= 0 && gx < w && gy >= 0 && gy < h) kk = Key32<T>::to(z[(size_t)gy * w + gx]);
      sk[ly * DLW + lxh] = kk;
    }
  }
  __syncthreads();
  // a wavefront owns a band of DH/4 consecutive rows, one column per lane; the 3x3 window of keys slides
  // down the column in registers (3 LDS reads per cell instead of 9)
  const int lx = threadIdx.x & (DW - 1), ly0 = (threadIdx.x >> 6) * (DH / 4);
  const int gx = x0 + lx;
  // local pointers.  A cell whose target stays in the tile stores the target's local index (12 bits); a cell
  // that is terminal within the tile stores LTERM_BASE | code, code = 0 its own root (pit / cut-row terminal),
  // 1..8 the descent neighbour (outside the tile) in raster order, 9 = drains off the raster -- so the global
  // pointer of a terminal never has to be recomputed from the keys.
  {
    uint32_t k0[3], k1[3], k2[3];
#pragma unroll
    for (int e = 0; e < 3; e++) { k0[e] = sk[ly0 * DLW + lx + e]; k1[e] = sk[(ly0 + 1) * DLW + lx + e]; }
#pragma unroll 4
    for (int j = 0; j < DH / 4; j++) {
      const int ly = ly0 + j, gy = y0 + ly;
#pragma unroll
      for (int e = 0; e < 3; e++) k2[e] = sk[(ly + 2) * DLW + lx + e];
      // Branch-free: every case is computed and the pointer is chosen by selects (the nested ifs compiled into five
      // levels of exec-mask branches per cell; this kernel is bound by instruction issue).
      // lowest (key, index) neighbour: visited in increasing index order and compared with strict '<', so the lowest
      // index wins among equal keys; a neighbour of EQUAL key is taken only if its index is lower.  q packs the
      // neighbour's 3x3 position n with its row n / 3 (bits 4-5) and column n % 3 (bits 6-7).
#define RD_Q(n) ((n) | (((n) / 3) << 4) | (((n) % 3) << 6))
      const uint32_t kc = k1[1];
      uint32_t bk, q;
      if (TOPO == 8) {
        bk = k0[0]; q = RD_Q(0);
        { const bool t = k0[1] < bk; bk = t ? k0[1] : bk; q = t ? RD_Q(1) : q; }
        { const bool t = k0[2] < bk; bk = t ? k0[2] : bk; q = t ? RD_Q(2) : q; }
        { const bool t = k1[0] < bk; bk = t ? k1[0] : bk; q = t ? RD_Q(3) : q; }
        { const bool t = k1[2] < bk; bk = t ? k1[2] : bk; q = t ? RD_Q(5) : q; }
        { const bool t = k2[0] < bk; bk = t ? k2[0] : bk; q = t ? RD_Q(6) : q; }
        { const bool t = k2[1] < bk; bk = t ? k2[1] : bk; q = t ? RD_Q(7) : q; }
        { const bool t = k2[2] < bk; bk = t ? k2[2] : bk; q = t ? RD_Q(8) : q; }
      } else {
        bk = k0[1]; q = RD_Q(1);
        { const bool t = k1[0] < bk; bk = t ? k1[0] : bk; q = t ? RD_Q(3) : q; }
        { const bool t = k1[2] < bk; bk = t ? k1[2] : bk; q = t ? RD_Q(5) : q; }
        { const bool t = k2[1] < bk; bk = t ? k2[1] : bk; q = t ? RD_Q(7) : q; }
      }
#undef RD_Q
      const int n = (int)(q & 15u);
      const bool drains = (bk < kc) | ((bk == kc) & (n < 4));
      const int tx = lx + (int)(q >> 6 & 3u) - 1, ty = ly + (int)(q >> 4 & 3u) - 1;
      const bool inside = (tx >= 0) & (tx < DW) & (ty >= 0) & (ty < DH);
      const uint16_t ldrain = inside ? (uint16_t)(ty * DW + tx) : (uint16_t)(LTERM_BASE | (uint16_t)(n < 4 ? n + 1 : n));
      const bool incell = (gx < w) & (gy < h);
      bool border = (gx == 0) | (gx == w - 1) | ((gy == 0) & !open_top) | ((gy == h - 1) & !open_bottom);   // true border
      if (outlet) border |= incell && outlet[(size_t)gy * w + gx] != 0;
      const bool cutrow = (gy == 0) | (gy == h - 1);   // (not a border: the cut row of a row-block shard, a frozen terminal)
      const uint16_t l = !incell ? LTERM_BASE : border ? (uint16_t)(LTERM_BASE | 9) : (cutrow | !drains) ? LTERM_BASE : ldrain;
      lp[ly * DW + lx] = l;
#pragma unroll
      for (int e = 0; e < 3; e++) { k0[e] = k1[e]; k1[e] = k2[e]; }
    }
  }
  __syncthreads();
  // pointer jumping inside the tile; any value ever stored is an ancestor, so races are harmless.
  // A cell whose parent is a tile root is finished for good, so only the still-active cells (bit mask
  // per thread) are revisited: integer VALU + LDS issue, not HBM, bounds this kernel.
  // Every trip reads all of the thread's pointers, then all of their targets' pointers: two batches of
  // independent LDS reads (the latency of a dependent read chain per cell was what bounded this loop).
  // Two hops per trip (c -> p -> q -> r): the barrier, not the LDS reads, is what a trip costs, and the hop distance
  // triples instead of doubling per trip.
  // A cell is finished for good once its pointer names a tile root (or it is one itself); a ROW of the wavefront's band
  // whose 64 cells are all finished is skipped with one scalar test -- this loop is instruction-issue bound (r02a SQ
  // counters: 2.4 k VALU instructions per wavefront, half of them here), and after two or three trips most rows are done.
  uint32_t act = 0;   // bit j: the cell of row ly0 + j may still move
#pragma unroll
  for (int j = 0; j < DH / 4; j++) act |= (lp[(ly0 + j) * DW + lx] < LTERM_BASE ? 1u : 0u) << j;
  for (int it = 0; it < 12; it++) {
    int still = 0;
#pragma unroll
    for (int g = 0; g < DH / 16; g++) {   // groups of four rows: batched LDS reads inside, one scalar test outside
      if (__ballot((act >> (4 * g)) & 15u) == 0ull) continue;   // wave uniform
      uint16_t pv[4], qv[4], rv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) pv[e] = lp[(ly0 + 4 * g + e) * DW + lx];
#pragma unroll
      for (int e = 0; e < 4; e++) qv[e] = lp[pv[e] < LTERM_BASE ? pv[e] : (ly0 + 4 * g + e) * DW + lx];
#pragma unroll
      for (int e = 0; e < 4; e++) rv[e] = lp[qv[e] < LTERM_BASE ? qv[e] : (ly0 + 4 * g + e) * DW + lx];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = 4 * g + e;
        if (pv[e] < LTERM_BASE && qv[e] < LTERM_BASE) {
          // q is not the tile root of the path yet?  then r is a cell further down: jump there and come back
          const bool more = rv[e] < LTERM_BASE;
          lp[(ly0 + j) * DW + lx] = more ? rv[e] : qv[e];
          if (more) still = 1;
          else act &= ~(1u << j);
        } else {
          act &= ~(1u << j);
        }
      }
    }
    if (!__syncthreads_or(still)) break;
  }
  // Pits (and a shard's cut-row terminals) -- the cells that are their own root -- get their dense basin id here:
  // ONE counter add per tile (ids are dense but not in raster order -- nothing depends on their order, the filled
  // surface is unique).
  uint32_t pitmask = 0;
#pragma unroll 4
  for (int j = 0; j < DH / 4; j++) {
    const int ly = ly0 + j, gy = y0 + ly;
    if (gx < w && gy < h && lp[ly * DW + lx] == LTERM_BASE) pitmask |= 1u << j;
  }
  const uint32_t mine = (uint32_t)__popc(pitmask);
  uint32_t incl = mine;   // inclusive prefix over the wavefront
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    pbase = tot ? atomicAdd(pit_counter, tot) : 0;
  }
  __syncthreads();
  uint32_t id = pbase + incl - mine;
  for (int k = 0; k < (int)(threadIdx.x >> 6); k++) id += wtot[k];
  uint32_t *pid = sk;   // the keys are no longer needed: basin id of the pit at local index i
  for (uint32_t m = pitmask; m; m &= m - 1) {
    const int j = __ffs((int)m) - 1;
    pid[(ly0 + j) * DW + lx] = id++;
  }
  __syncthreads();
  // One word per cell: its basin id when its path ends at a pit of this tile (final), LAB_PEND | the first cell
  // outside the tile on its path otherwise, OUTP when it drains off the raster.  k_tile_label resolves the rest.
#pragma unroll 4
  for (int j = 0; j < DH / 4; j++) {
    const int ly = ly0 + j, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const uint16_t p0 = lp[ly * DW + lx];
    const int r = p0 < LTERM_BASE ? (int)p0 : ly * DW + lx;   // the tile root of the cell's path (itself when it is one)
    const uint16_t p = lp[r];                                  // (its own entry again when the cell is a root: no branch)
    const int code = p & 15;
    const uint32_t basin = pid[r];                             // (only meaningful for code 0; any slot is readable)
    const int rx = r & (DW - 1), ry = r >> 6;
    const int n = code <= 4 ? code - 1 : code;   // 3x3 position of the root's descent neighbour (codes 1..8)
    const int nr = n >= 6 ? 2 : n >= 3 ? 1 : 0, nc = n - 3 * nr;
    const uint32_t pend = LAB_PEND | ((uint32_t)(y0 + ry + nr - 1) * (uint32_t)w + (uint32_t)(x0 + rx + nc - 1));
    lab[(size_t)gy * w + gx] = code == 9 ? OUTP : code == 0 ? basin : pend;
  }
}

// ------------------------------------------------------------------------------------------
// 2. labels.  After k_descent a word of lab[] is a basin id (final), LAB_PEND | next cell, or OUTP.  Every value ever
// stored in a word is the cell's basin, or a cell further down its path, or OUTP when the path leaves the raster --
// so concurrent in-place updates and stale cached reads are harmless.
// ------------------------------------------------------------------------------------------
// Fallback for pathologically long tile-to-tile chains: bounded hops per pass with path compression; the host
// repeats while flagged.  Resolves a word as soon as it meets a resolved one.
__global__ __launch_bounds__(NTHR) void k_chase(uint32_t *lab, uint32_t n, int maxhops, uint32_t *flag) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride) {
    const uint32_t c = (uint32_t)c64;
    uint32_t v = lab[c];
    if (!(v & LAB_PEND) || v == OUTP) continue;
    const uint32_t v0 = v;
    int hops = 0;
    bool unfinished = false;
    for (;;) {
      const uint32_t q = lab[v & ~LAB_PEND];
      v = q;
      if (!(q & LAB_PEND) || q == OUTP) break;
      if (++hops >= maxhops) { unfinished = true; break; }
    }
    if (v != v0) lab[c] = v;
    if (unfinished) *flag = 1;
  }
}

// Labels per tile: the unresolved cells of a 64x64 tile share a handful of distinct pointers (the ring cells its
// paths leave through), so the distinct values are collected in an LDS table, each is chased to a resolved word
// ONCE (tile-to-tile hops, global gathers), and the cells read their label from LDS; lab[c] = B for cells draining
// off the raster.  Cells whose path ends at a pit of their own tile were finished by k_descent and are not written.
// A chain longer than maxhops raises the flag and leaves (compressed) pending words: the host then falls back to the
// compressing passes.
constexpr int LT_SLOTS = 1024;
constexpr uint32_t LT_EMPTY = 0xFFFFFFFEu;

__device__ __forceinline__ uint32_t chase_to_label(const uint32_t *lab, uint32_t v, uint32_t B, int maxhops, uint32_t *flag) {
  int hops = 0;
  for (;;) {   // v: a pending word
    const uint32_t q = lab[v & ~LAB_PEND];
    if (q == OUTP) return B;
    if (!(q & LAB_PEND)) return q;
    v = q;
    if (++hops >= maxhops) { *flag = 1; return v; }
  }
}

__global__ __launch_bounds__(NTHR) void k_tile_label(uint32_t *lab, int w, int h, uint32_t tilesX, uint32_t ntiles,
                                                     uint32_t B, int maxhops, uint32_t *flag) {
  __shared__ uint32_t tkey[LT_SLOTS];
  __shared__ uint32_t tval[LT_SLOTS];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * DW, y0 = (int)(t / tilesX) * DH;
  for (int i = threadIdx.x; i < LT_SLOTS; i += NTHR) tkey[i] = LT_EMPTY;
  __syncthreads();
  const int lx = threadIdx.x & (DW - 1), ly0 = threadIdx.x >> 6;
  const int gx = x0 + lx;
  constexpr int CELLS = DH / 4;
  uint32_t pv[CELLS];
  int16_t sl[CELLS];   // table slot, -1: final already / outside the raster, -2: drains off the raster, -3: table full
#pragma unroll
  for (int j = 0; j < CELLS; j++) {   // all loads of the thread in flight together
    const int gy = y0 + ly0 + 4 * j;
    pv[j] = (gx < w && gy < h) ? lab[(size_t)gy * w + gx] : 0u;
  }
#pragma unroll
  for (int j = 0; j < CELLS; j++) {
    sl[j] = -1;
    const uint32_t p = pv[j];
    if (!(p & LAB_PEND)) continue;
    if (p == OUTP) { sl[j] = -2; continue; }
    if (j > 0 && sl[j - 1] >= 0 && pv[j - 1] == p) { sl[j] = sl[j - 1]; continue; }
    uint32_t slot = (p * 0x9E3779B1u) >> 22;
    int found = -3;
#pragma unroll 1
    for (int probe = 0; probe < 16; probe++) {
      uint32_t k = tkey[slot];
      if (k == LT_EMPTY) k = atomicCAS(&tkey[slot], LT_EMPTY, p);
      if (k == LT_EMPTY || k == p) { found = (int)slot; break; }
      slot = (slot + 1) & (LT_SLOTS - 1);
    }
    sl[j] = (int16_t)found;
  }
  __syncthreads();
  {
    // the thread's table slots are chased together: one gather per live chain and trip, all in flight at once
    constexpr int SPT = LT_SLOTS / NTHR;
    uint32_t p[SPT];
    uint32_t live = 0;
#pragma unroll
    for (int r = 0; r < SPT; r++) {
      p[r] = tkey[threadIdx.x + r * NTHR];   // a pending word (or LT_EMPTY)
      if (p[r] != LT_EMPTY) live |= 1u << r;
    }
    for (int hops = 0; live; hops++) {
      if (hops > maxhops) { *flag = 1; break; }   // the words stay pending (compressed): the fallback finishes them
      uint32_t q[SPT];
#pragma unroll
      for (int r = 0; r < SPT; r++) q[r] = (live >> r & 1u) ? lab[p[r] & ~LAB_PEND] : 0u;
#pragma unroll
      for (int r = 0; r < SPT; r++) {
        if (!(live >> r & 1u)) continue;
        if (q[r] == OUTP) { p[r] = B; live &= ~(1u << r); }
        else if (!(q[r] & LAB_PEND)) { p[r] = q[r]; live &= ~(1u << r); }
        else p[r] = q[r];
      }
    }
#pragma unroll
    for (int r = 0; r < SPT; r++) tval[threadIdx.x + r * NTHR] = p[r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CELLS; j++) {
    if (sl[j] == -1) continue;
    const size_t c = (size_t)(y0 + ly0 + 4 * j) * w + gx;
    lab[c] = sl[j] >= 0 ? tval[sl[j]] : sl[j] == -2 ? B : chase_to_label(lab, pv[j], B, maxhops, flag);
  }
}

// ------------------------------------------------------------------------------------------
// 3. after the compressing fallback passes every word is a basin id or OUTP: lab[c] = B for the latter
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_label_cells(uint32_t *lab, uint32_t n, uint32_t B) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride)
    if (lab[c64] == OUTP) lab[c64] = B;
}

// Stream compaction helper: every thread of the (256-thread) block calls it; ONE global atomic per
// block (same-address atomics serialise at ~12 ns each on this chip, so per-wave appends of 10^7
// elements cost milliseconds).  Returns the output slot for threads with pred.
__device__ __forceinline__ uint32_t block_append(bool pred, uint32_t *counter) {
  __shared__ uint32_t wcnt[NTHR / 64];
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

// ------------------------------------------------------------------------------------------
// 4. Boruvka rounds
// tables (B+1 entries, index B = the outside): cur[b]  current root component of basin b
//                                              acc[b]  max pass key on b's path to cur[b]
//                                              best[r] (pass key << 32 | neighbour root), per root
//                                              link[r] (path max key << 32 | parent root), per root
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_init_tables(uint32_t *cur, uint32_t *acc, unsigned long long *link,
                                                      uint32_t *roots, uint32_t *nroots,
                                                      const uint32_t *__restrict__ tid, uint32_t B) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;   // grid covers whole waves past B
  bool open = false;
  if (i <= B) {
    const bool closed = i == B || (tid && tid[i] != NO_TID);
    cur[i] = closed ? (i | CLOSED) : i;
    acc[i] = 0;
    link[i] = (unsigned long long)i;
    open = !closed;
  }
  // live roots = basins that still have to hook (everything but the outside and frozen terminals).  Without
  // terminals that is simply 0..B-1: no compaction (one same-address atomic per block is 0.5 ms for 4e4 blocks)
  if (!tid) {
    if (i < B) roots[i] = i;
    if (i == 0) *nroots = B;
    return;
  }
  const uint32_t slot = block_append(open, nroots);
  if (open) roots[slot] = i;
}

// tid[basin] = terminal id of a cut-row cell's basin (top row: x, bottom row: w + x), NO_TID otherwise
__global__ __launch_bounds__(NTHR) void k_mark_terminals(const uint32_t *__restrict__ lab, uint32_t *tid, int w, int h,
                                                         int open_top, int open_bottom) {
  const int x = blockIdx.x * NTHR + threadIdx.x;
  if (x <= 0 || x >= w - 1) return;
  if (open_top) tid[lab[x]] = (uint32_t)x;
  if (open_bottom) tid[lab[(size_t)(h - 1) * w + x]] = (uint32_t)(w + x);
}

// (r06) dyn != nullptr in the round kernels below: the count is read from the device -- the rounds of a fill are enqueued
// without a host read-back between them, over grids the host sizes from upper bounds (grid-stride loops: a launch
// whose list turns out empty costs a few microseconds).
__global__ __launch_bounds__(NTHR) void k_best_reset(const uint32_t *__restrict__ roots, uint32_t nroots,
                                                     unsigned long long *best, const uint32_t *__restrict__ dyn = nullptr) {
  if (dyn) nroots = *dyn;
  for (uint32_t i = blockIdx.x * NTHR + threadIdx.x; i < nroots; i += gridDim.x * NTHR) best[roots[i]] = ~0ull;
}

// One raster pass of a Boruvka round.  tiles_in == nullptr: all tiles (XCD-banded order); otherwise
// the list of tiles that still contained a component boundary last round (a tile without candidates can
// never produce one again: components only merge, open components only close).  Candidates of the same
// component are first reduced in an LDS table, so a tile issues ~one global atomic per component it
// touches instead of one per boundary cell.
constexpr int SC_SLOTS = 256;

// find-or-insert component C in the block's LDS table; -1 when its probe window is full
__device__ __forceinline__ int tab_slot(uint32_t *tab_id, uint32_t C) {
  uint32_t slot = (C * 0x9E3779B1u) >> 24;
#pragma unroll 1
  for (int probe = 0; probe < 8; probe++) {
    uint32_t id = tab_id[slot];
    if (id == 0xFFFFFFFFu) id = atomicCAS(&tab_id[slot], 0xFFFFFFFFu, C);
    if (id == 0xFFFFFFFFu || id == C) return (int)slot;
    slot = (slot + 1) & (SC_SLOTS - 1);
  }
  return -1;
}

// The component-pair list of the first round (EMIT).  A raster pass is only needed ONCE: while it looks for every
// component's lowest pass it also sees every pair of adjacent components, and the lowest pass of each PAIR is all the
// later rounds need.  The pairs of a tile are reduced in an LDS table and appended to a global list of
// (component, component, pass key) records; rounds 2.. contract that list (k_edge_round) instead of re-reading the
// raster.  The list is split into ESEG segments with their own fill counters: a single counter bumped once per
// tile would serialise (~12 ns per same-address atomic, 7.8e5 tiles).  A pair that finds no slot in the tile's table is
// appended on its own (pair_spill); a segment that runs full raises `overflow`, and the host falls back to raster
// passes for the remaining rounds.
constexpr int PT_SLOTS = 256, PT_PROBES = 16;
constexpr uint32_t ESEG = 8192;
struct EdgeOut {
  uint32_t *a, *b, *k;     // nseg segments of `segcap` records each
  uint32_t *segcount;      // fill count per segment
  uint32_t segcap;
  uint32_t segmask;        // nseg - 1 (nseg: a power of two <= ESEG)
  uint32_t *overflow;
  uint32_t seglimit = 0xFFFFFFFFu;   // k_pairs16: records a segment may take (<= segcap; smaller under RDGPU_FILL_EDGE_CAP)
};

template <int SLOTS = PT_SLOTS>
__device__ __forceinline__ uint32_t pair_home(uint32_t C, uint32_t D) {
  return (((C * 0x9E3779B1u) ^ (D * 0x85EBCA6Bu)) >> 16) & (SLOTS - 1);
}
// find-or-insert the pair (C, D) in the tile's LDS pair table and lower its pass key; false when the probe window is full
template <int SLOTS = PT_SLOTS>
__device__ __forceinline__ bool pair_insert(unsigned long long *pt_pair, uint32_t *pt_key, uint32_t C, uint32_t D, uint32_t key) {
  const unsigned long long pr = ((unsigned long long)C << 32) | D;
  uint32_t slot = pair_home<SLOTS>(C, D);
#pragma unroll 1
  for (int probe = 0; probe < PT_PROBES; probe++) {
    unsigned long long v = pt_pair[slot];
    if (v == ~0ull) {
      v = atomicCAS(&pt_pair[slot], ~0ull, pr);
      if (v == ~0ull) v = pr;
    }
    if (v == pr) {
      if (key < pt_key[slot]) atomicMin(&pt_key[slot], key);
      return true;
    }
    slot = (slot + 1) & (SLOTS - 1);
  }
  return false;
}

// A pair that found no slot in the tile's table goes straight to the tile's segment, one record per atomic: slow, but
// it keeps rasters with hundreds of basins per tile on the pair list (the records are only not merged per tile).
__device__ __forceinline__ void pair_spill(const EdgeOut &eo, unsigned long long *best, uint32_t seg,
                                           uint32_t lo, uint32_t hi, uint32_t key) {
  // (it bypasses the table the components' proposals are derived from, so it proposes on its own)

  const unsigned long long cl = ((unsigned long long)key << 32) | hi, ch = ((unsigned long long)key << 32) | lo;
  if (!(lo & CLOSED) && cl < best[lo]) atomicMin(&best[lo], cl);
  if (!(hi & CLOSED) && ch < best[hi]) atomicMin(&best[hi], ch);
  const uint32_t g = atomicAdd(&eo.segcount[seg], 1u);
  if (g < eo.segcap) {
    const size_t i = (size_t)seg * eo.segcap + g;
    eo.a[i] = lo; eo.b[i] = hi; eo.k[i] = key;
  } else {
    *eo.overflow = 1;
  }
}

// L16 (the compact-label fill below): `lab` holds 16-bit slots (the cell's root within its 64 x 64 descent tile), a cell's
// component is cur[tile_base[its tile] + slot] -- cur is then the node table curN --, and FIRST is not used.
template <class T, int TOPO, bool FIRST, bool VEC, bool EMIT, bool L16 = false>
__global__ __launch_bounds__(NTHR) void k_scan(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                               const uint32_t *__restrict__ cur, unsigned long long *best,
                                               int w, int h, uint32_t B, uint32_t tilesX, uint32_t ntiles,
                                               const uint32_t *__restrict__ tiles_in, uint32_t nwork,
                                               uint8_t *alive_out, EdgeOut eo,
                                               const uint32_t *__restrict__ tile_base = nullptr, uint32_t dtx = 0,
                                               const uint8_t *__restrict__ skip = nullptr) {
  __shared__ __attribute__((aligned(8))) uint32_t sk[LH * LW];
  __shared__ uint32_t sc[LH * LW];
  // The component table: in the pair pass it is only needed AFTER the pairs are reduced, when the keys are dead, so it
  // lives in sk's storage there (with the 256-slot pair table: 25.1 instead of 31.5 KB of LDS, a sixth block per CU).
  __shared__ uint32_t tab_id_s[EMIT ? 1 : SC_SLOTS];
  __shared__ unsigned long long tab_val_s[EMIT ? 1 : SC_SLOTS];
  __shared__ uint8_t tab_cross_s[EMIT ? 1 : SC_SLOTS];
  uint32_t *const tab_id = EMIT ? sk : tab_id_s;
  unsigned long long *const tab_val = EMIT ? reinterpret_cast<unsigned long long *>(sk + SC_SLOTS) : tab_val_s;
  uint8_t *const tab_cross = EMIT ? reinterpret_cast<uint8_t *>(sk + 3 * SC_SLOTS) : tab_cross_s;
  __shared__ uint16_t list[TW * TH];   // LDS offsets of the cells that touch another component
  __shared__ uint32_t nlist;
  __shared__ unsigned long long pt_pair[EMIT ? PT_SLOTS : 1];   // (smaller id << 32 | larger id), ~0 = empty
  __shared__ uint32_t pt_key[EMIT ? PT_SLOTS : 1];              // lowest pass key of the pair
  __shared__ uint32_t pt_n, pt_base, any_open;
  // XCD-banded order in every round; from round 2 on only the tiles that still held a component boundary last
  // round are launched (compacted list: a dead tile costs neither a block nor a flag load)
  const uint32_t wi = xcd_tile(blockIdx.x, nwork);
  if (wi >= nwork) return;
  const uint32_t t = tiles_in ? tiles_in[wi] : wi;
  const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
  if (L16 && skip && skip[(uint32_t)(y0 / DH) * dtx + (uint32_t)(x0 / DW)]) return;   // nothing but outlets here, ring included
  if (!EMIT)
    for (int i = threadIdx.x; i < SC_SLOTS; i += NTHR) { tab_id[i] = 0xFFFFFFFFu; tab_val[i] = ~0ull; tab_cross[i] = 0; }
  if (EMIT)
    for (int i = threadIdx.x; i < PT_SLOTS; i += NTHR) { pt_pair[i] = ~0ull; pt_key[i] = 0xFFFFFFFFu; }
  if (threadIdx.x == 0) { nlist = 0; pt_n = 0; any_open = 0; }
  // FIRST: every basin is still its own component (only used when there are no frozen terminals)
#define RD_COMP(l) (FIRST ? ((l) == B ? (B | CLOSED) : (l)) : cur[(l)])
  {
    // Interior columns with quad loads (VEC: naturally aligned quads, else any width -- see load_quad),
    // the two halo columns with scalar loads.  The kernel is latency bound (SQ counters: waves parked ~75 % of
    // their cycles), so the loads are issued in two batches -- every z / label quad of this thread, then every
    // component gather -- instead of item by item: two dependent memory round trips per tile instead of six.
    constexpr int NQ = LH * (TW / 4);                 // 16-byte items of the tile incl. halo rows
    constexpr int QPT = (NQ + NTHR - 1) / NTHR;       // per thread
    Quad<T> zq[QPT];   // four cells per load: 16 / 8 / 4 bytes for 4- / 2- / 1-byte elevations
    Quad<uint32_t> lq[QPT];
    uint32_t tb[L16 ? QPT : 1];   // L16: node base of the descent tile the quad lies in
    bool ok[QPT];
    const uint16_t *lab16 = reinterpret_cast<const uint16_t *>(lab);
    uint32_t tbU = 0, tbC = 0, tbD = 0, tbL[3] = {0, 0, 0}, tbR[3] = {0, 0, 0};
    if (L16) {
      static_assert(TW == DW && DH % TH == 0, "a scan tile lies inside one descent tile");
      const int dcol = x0 / DW, rU = max(y0 - 1, 0) / DH, rC = y0 / DH, rD = min(y0 + TH, h - 1) / DH;
      const int rows3[3] = {rU, rC, rD};
      tbU = tile_base[(uint32_t)rU * dtx + (uint32_t)dcol];
      tbC = tile_base[(uint32_t)rC * dtx + (uint32_t)dcol];
      tbD = tile_base[(uint32_t)rD * dtx + (uint32_t)dcol];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        tbL[k] = tile_base[(uint32_t)rows3[k] * dtx + (uint32_t)max(dcol - 1, 0)];
        tbR[k] = tile_base[(uint32_t)rows3[k] * dtx + min((uint32_t)dcol + 1u, dtx - 1u)];
      }
    }
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = threadIdx.x + r * NTHR;
      const int ly = i / (TW / 4), q = i - ly * (TW / 4);
      const int gx = x0 + 4 * q, gy = y0 - 1 + ly;
      ok[r] = i < NQ && gy >= 0 && gy < h && gx < w;
      if (ok[r]) {
        const size_t g = (size_t)gy * w + gx;
        if (L16) {
          // (the labels first, the elevations in a loop of their own below: loads return in order, and the component
          // gathers only wait for the labels -- the elevation loads are still in flight behind them)
          const Quad<uint16_t> sq = load_quad<uint16_t, VEC>(lab16 + (g - gx), gx, w, (uint16_t)0);
#pragma unroll
          for (int e = 0; e < 4; e++) lq[r].v[e] = sq.v[e];
          // node base of the quad's descent tile: the scan tile lies inside ONE descent tile (TW == DW, TH divides DH),
          // only its halo rows may belong to the tile above / below -- three block-uniform (scalar) loads
          tb[r] = ly == 0 ? tbU : ly == LH - 1 ? tbD : tbC;
        } else {
          zq[r] = load_quad<T, VEC>(z + (g - gx), gx, w, T());
          lq[r] = load_quad<uint32_t, VEC>(lab + (g - gx), gx, w, B);   // past the row end: the outside's label
        }
      }
    }
    // halo columns: one cell per thread for the first 2 * LH threads
    const bool hcell = threadIdx.x < 2 * LH;
    const int hly = threadIdx.x >> 1, hlx = (threadIdx.x & 1) ? LW - 1 : 0;
    bool hok = false;
    T hz = T();
    uint32_t hl = 0, htb = 0;
    if (hcell) {
      const int gx = x0 - 1 + hlx, gy = y0 - 1 + hly;
      hok = gx >= 0 && gx < w && gy >= 0 && gy < h;
      if (hok) {
        if (L16) {
          hl = lab16[(size_t)gy * w + gx];
          const int k = hly == 0 ? 0 : hly == LH - 1 ? 2 : 1;
          htb = (threadIdx.x & 1) ? (k == 0 ? tbR[0] : k == 1 ? tbR[1] : tbR[2]) : (k == 0 ? tbL[0] : k == 1 ? tbL[1] : tbL[2]);
        } else {
          hl = lab[(size_t)gy * w + gx];
        }
        hz = z[(size_t)gy * w + gx];
      }
    }
    if (L16) {
#pragma unroll
      for (int r = 0; r < QPT; r++) {
        const int i = threadIdx.x + r * NTHR;
        const int ly = i / (TW / 4), q = i - ly * (TW / 4);
        const int gx = x0 + 4 * q, gy = y0 - 1 + ly;
        if (ok[r]) zq[r] = load_quad<T, VEC>(z + (size_t)gy * w, gx, w, T());
      }
    }
    uint32_t cq[QPT][4];
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      // branch-free: an absent quad reads the outside's entry (label B: cur[B] == B | CLOSED), so all gathers of
      // the thread are in flight together
      if (L16) {
        const int q4 = 4 * ((threadIdx.x + r * NTHR) % (TW / 4));
#pragma unroll
        for (int e = 0; e < 4; e++) {   // (absent cells and cells past the row end: node 0 is read, the outside is taken)
          const bool in = ok[r] && x0 + q4 + e < w;
          const uint32_t c = cur[in ? tb[r] + lq[r].v[e] : 0u];
          cq[r][e] = in ? c : (B | CLOSED);
        }
      } else {
        const uint32_t l[4] = {ok[r] ? lq[r].v[0] : B, ok[r] ? lq[r].v[1] : B, ok[r] ? lq[r].v[2] : B, ok[r] ? lq[r].v[3] : B};
#pragma unroll
        for (int e = 0; e < 4; e++) cq[r][e] = RD_COMP(l[e]);
      }
    }
    const uint32_t hlsafe = hok ? hl : B;
    uint32_t hc;
    if (L16) { const uint32_t c = cur[hok ? htb + hl : 0u]; hc = hok ? c : (B | CLOSED); }
    else hc = RD_COMP(hlsafe);
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = threadIdx.x + r * NTHR;
      if (i < NQ) {
        const int ly = i / (TW / 4), q = i - ly * (TW / 4);
        const int o = ly * LW + 1 + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; e++) { sk[o + e] = (ok[r] && x0 + 4 * q + e < w) ? Key32<T>::to(zq[r].v[e]) : 0u; sc[o + e] = cq[r][e]; }
      }
    }
    if (hcell) { sk[hly * LW + hlx] = hok ? Key32<T>::to(hz) : 0u; sc[hly * LW + hlx] = hc; }
  }
#undef RD_COMP
  __syncthreads();
  // Phase 1 -- detect: each wavefront walks a band of TH/4 consecutive rows, one column per lane, with
  // the 3x3 window of component ids carried in registers, and appends the cells that touch another
  // component to an LDS list.  Integer VALU is the bottleneck of this kernel (not HBM), and only
  // ~5-15% of the cells sit on a component boundary, so the expensive candidate evaluation (phase 2)
  // runs densely over that list instead of over every lane.
  const int lx = threadIdx.x & (TW - 1), band = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int ROWS = TH / 4;
  const int gx = x0 + lx;
  const int yb = band * ROWS;   // first tile row of the band
  {
    uint32_t c0[3], c1[3], c2[3];
    {
      const int o = yb * LW + lx;   // LDS row yb is the halo row above tile row yb
#pragma unroll
      for (int e = 0; e < 3; e++) { c0[e] = sc[o + e]; c1[e] = sc[o + LW + e]; }
    }
    unsigned long long bal[ROWS];   // per row: the lanes whose cell touches another component
    unsigned long long balive = 0;
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const int ly = yb + j, gy = y0 + ly;
      {
        const int o = (ly + 2) * LW + lx;
#pragma unroll
        for (int e = 0; e < 3; e++) c2[e] = sc[o + e];
      }
      const uint32_t C = c1[1];
      uint32_t d = (c0[1] ^ C) | (c1[0] ^ C) | (c1[2] ^ C) | (c2[1] ^ C);
      if (TOPO == 8) d |= (c0[0] ^ C) | (c0[2] ^ C) | (c2[0] ^ C) | (c2[2] ^ C);
      // closed: drains to the outside or to a frozen terminal -- never proposes
      bool hit = d != 0 && !(C & CLOSED) && gx < w && gy < h;
      if (EMIT) {
        // the pair pass lists the cells with a foreign E / SE / S / SW neighbour (open or closed: see phase 2);
        // "this tile holds a component boundary" keeps its meaning for the raster fallback
        balive |= __ballot(hit);
        uint32_t df = (c1[2] ^ C) | (c2[1] ^ C);
        if (TOPO == 8) df |= (c2[0] ^ C) | (c2[2] ^ C);
        hit = df != 0 && gx < w && gy < h;
      }
      bal[j] = __ballot(hit);
#pragma unroll
      for (int e = 0; e < 3; e++) { c0[e] = c1[e]; c1[e] = c2[e]; }
    }
    if (EMIT && balive && lane == 0) any_open = 1;
    // one list reservation per wavefront (a returning LDS atomic per row was a dependent chain of eight)
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < ROWS; j++) total += (uint32_t)__popcll(bal[j]);
    if (total) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&nlist, total);
      base = __shfl(base, 0, 64);
#pragma unroll
      for (int j = 0; j < ROWS; j++) {
        if (bal[j] >> lane & 1ull)
          list[base + __popcll(bal[j] & ((1ull << lane) - 1ull))] = (uint16_t)((yb + j + 1) * LW + lx + 1);
        base += (uint32_t)__popcll(bal[j]);
      }
    }
  }
  // (no barrier needed between the detect phase and this: both only read sc)
  // components that also live outside this tile (seen on the halo ring) need the global atomic;
  // a component entirely inside the tile is reduced here completely and can use a plain store
  for (int i = threadIdx.x; !EMIT && i < 2 * LW + 2 * (LH - 2); i += NTHR) {   // (the pair pass always uses the atomic)
    int o;
    if (i < LW) o = i;
    else if (i < 2 * LW) o = (LH - 1) * LW + (i - LW);
    else { const int r = (i - 2 * LW) >> 1; o = (r + 1) * LW + (((i - 2 * LW) & 1) ? LW - 1 : 0); }
    const uint32_t C = sc[o];
    if (!(C & CLOSED)) {
      const int slot = tab_slot(tab_id, C);
      if (slot >= 0) tab_cross[slot] = 1;
    }
  }
  __syncthreads();
  const uint32_t nl = nlist;
  if (EMIT) {
    // Phase 2 (pair pass) -- every adjacent cell pair is recorded exactly once, by its earlier cell in raster order:
    // a listed cell looks at its E, SE, S, SW neighbours (D4: E, S) and the pair is stored as (smaller id, larger
    // id) with its pass height.  Up to two distinct neighbouring components are kept in registers and looked up
    // together: the common case -- the pair is already in its home slot with a pass at least as low -- costs one
    // LDS round trip for the whole cell.  The components' own lowest passes are derived from the pairs afterwards.
    constexpr int NF = TOPO == 8 ? 4 : 2;
    const int foff[4] = {1, TOPO == 8 ? LW + 1 : LW, LW, LW - 1};
    const uint32_t seg = t & eo.segmask;
    for (uint32_t i = threadIdx.x; i < nl; i += NTHR) {
      const int o = list[i];
      const uint32_t C = sc[o], kc = sk[o];
      uint32_t nD[NF], nH[NF];
#pragma unroll
      for (int e = 0; e < NF; e++) { nD[e] = sc[o + foff[e]]; nH[e] = sk[o + foff[e]]; }
      uint32_t pd[2] = {C, C}, pk[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};   // pd == C: unused
#pragma unroll
      for (int e = 0; e < NF; e++) {
        const uint32_t D = nD[e], hn = nH[e] > kc ? nH[e] : kc;   // pass height of the cell pair
        if (D != C && !(C & D & CLOSED)) {   // two closed components never merge
          if (pd[0] == C) pd[0] = D;
          if (D == pd[0]) pk[0] = hn < pk[0] ? hn : pk[0];
          else {
            if (pd[1] == C) pd[1] = D;
            if (D == pd[1]) pk[1] = hn < pk[1] ? hn : pk[1];
            else if (!pair_insert(pt_pair, pt_key, C < D ? C : D, C < D ? D : C, hn)) pair_spill(eo, best, seg, C < D ? C : D, C < D ? D : C, hn);
          }
        }
      }
      uint32_t ps[2], pq[2], lo[2], hi[2];
      unsigned long long pv[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        lo[j] = C < pd[j] ? C : pd[j];
        hi[j] = C < pd[j] ? pd[j] : C;
        ps[j] = pair_home(lo[j], hi[j]);
        pv[j] = pt_pair[ps[j]];
        pq[j] = pt_key[ps[j]];
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (pd[j] != C) {
          const unsigned long long pr = ((unsigned long long)lo[j] << 32) | hi[j];
          if (pv[j] == pr) { if (pk[j] < pq[j]) atomicMin(&pt_key[ps[j]], pk[j]); }
          else if (!pair_insert(pt_pair, pt_key, lo[j], hi[j], pk[j])) pair_spill(eo, best, seg, lo[j], hi[j], pk[j]);
        }
      }
    }
    __syncthreads();
    // the keys are dead from here on: their storage becomes the component table (visible after the next barrier)
    for (int i = threadIdx.x; i < SC_SLOTS; i += NTHR) { tab_id[i] = 0xFFFFFFFFu; tab_val[i] = ~0ull; }
    // compact the occupied slots (`list` is free again)
    const int lane64 = threadIdx.x & 63;
    for (int i = threadIdx.x; i < PT_SLOTS; i += NTHR) {
      const bool occ = pt_pair[i] != ~0ull;
      const unsigned long long bal = __ballot(occ);
      uint32_t base = 0;
      if (lane64 == 0 && bal) base = atomicAdd(&pt_n, (uint32_t)__popcll(bal));
      base = __shfl(base, 0, 64);
      if (occ) list[base + __popcll(bal & ((1ull << lane64) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    // reserve the tile's space in its segment with ONE atomic (its latency hides behind the loop below)
    const uint32_t tot = pt_n;
    if (threadIdx.x == 0) {
      uint32_t ob = 0xFFFFFFFFu;
      if (tot) {
        ob = atomicAdd(&eo.segcount[seg], tot);
        if (ob + tot > eo.segcap) { *eo.overflow = 1; ob = 0xFFFFFFFFu; }
      }
      pt_base = ob;
      alive_out[t] = any_open ? 1 : 0;
    }
    // every pair proposes its pass to its open sides, reduced per component in the LDS table
    for (uint32_t i = threadIdx.x; i < tot; i += NTHR) {
      const int sl = list[i];
      const unsigned long long pr = pt_pair[sl];
      const uint32_t lo = (uint32_t)(pr >> 32), hi = (uint32_t)pr, key = pt_key[sl];
      if (!(lo & CLOSED)) {
        const unsigned long long cand = ((unsigned long long)key << 32) | hi;
        const int slot = tab_slot(tab_id, lo);
        if (slot >= 0) atomicMin(&tab_val[slot], cand);
        else if (cand < best[lo]) atomicMin(&best[lo], cand);
      }
      if (!(hi & CLOSED)) {
        const unsigned long long cand = ((unsigned long long)key << 32) | lo;
        const int slot = tab_slot(tab_id, hi);
        if (slot >= 0) atomicMin(&tab_val[slot], cand);
        else if (cand < best[hi]) atomicMin(&best[hi], cand);
      }
    }
    __syncthreads();
    // (always the atomic: a pair of C with a cell above / left of it is recorded -- and proposed -- by the tile that
    // owns that earlier cell, so even a component entirely inside this tile can have another proposer)
    for (int i = threadIdx.x; i < SC_SLOTS; i += NTHR) {
      const uint32_t C = tab_id[i];
      if (C != 0xFFFFFFFFu) {
        const unsigned long long cand = tab_val[i];
        if (cand < best[C]) atomicMin(&best[C], cand);      // cheap (possibly stale) pre-check first
      }
    }
    const uint32_t ob = pt_base;
    if (ob != 0xFFFFFFFFu) {
      const size_t g0 = (size_t)seg * eo.segcap + ob;
      for (uint32_t i = threadIdx.x; i < tot; i += NTHR) {
        const int sl = list[i];
        const unsigned long long pr = pt_pair[sl];
        eo.a[g0 + i] = (uint32_t)(pr >> 32);
        eo.b[g0 + i] = (uint32_t)pr;
        eo.k[g0 + i] = pt_key[sl];
      }
    }
    return;
  }
  // Phase 2 -- evaluate the boundary cells densely
  for (uint32_t i = threadIdx.x; i < nl; i += NTHR) {
    const int o = list[i];
    const uint32_t C = sc[o], kc = sk[o];
    unsigned long long cand = ~0ull;
    // the ring of neighbours, read once: N, NE, E, SE, S, SW, W, NW (D4: N, E, S, W)
    constexpr int NNB = TOPO == 8 ? 8 : 4;
    const int noff[8] = {-LW, TOPO == 8 ? -LW + 1 : 1, TOPO == 8 ? 1 : LW, TOPO == 8 ? LW + 1 : -1, LW, LW - 1, -1, -LW - 1};
    uint32_t nD[NNB], nH[NNB];
#pragma unroll
    for (int e = 0; e < NNB; e++) { nD[e] = sc[o + noff[e]]; nH[e] = sk[o + noff[e]]; }
#pragma unroll
    for (int e = 0; e < NNB; e++) {
      nH[e] = nH[e] > kc ? nH[e] : kc;   // pass height of the cell pair
      const unsigned long long e_ = ((unsigned long long)(nD[e] != C ? nH[e] : 0xFFFFFFFFu) << 32) | nD[e];
      cand = e_ < cand ? e_ : cand;
    }
    // reduce per component in LDS; fall back to the global atomic when its probe window is full
    const int slot = tab_slot(tab_id, C);
    if (slot >= 0) atomicMin(&tab_val[slot], cand);
    else if (cand < best[C]) atomicMin(&best[C], cand);
  }
  const int any = nl != 0;
  __syncthreads();
  const int alive = any;
  for (int i = threadIdx.x; i < SC_SLOTS; i += NTHR) {
    const uint32_t C = tab_id[i];
    if (C != 0xFFFFFFFFu) {
      const unsigned long long cand = tab_val[i];
      if (cand == ~0ull) continue;   // only seen on the halo ring
      if (!tab_cross[i]) best[C] = cand;                       // nobody else proposes for C
      else if (cand < best[C]) atomicMin(&best[C], cand);      // cheap (possibly stale) pre-check first
    }
  }
  // NB: no global "alive tiles" counter here -- ~10^6 same-address atomics serialise at ~12 ns each
  // (that alone cost 9 ms per pass); k_compact_alive turns the flags into the next round's tile list
  if (threadIdx.x == 0) alive_out[t] = alive ? 1 : 0;
}

// One contraction round on the pair list: every record (a, b, key) is mapped to the current components of its two
// sides; a record inside one component (or between two closed ones) is dropped for good, the others propose
// (key << 32 | other side) to best[] exactly as the raster pass does, and are written to the next round's list as
// (current a, current b, key).  SEG: the input is the segmented list of the raster pass (one segment per block
// range: segcap is a multiple of the 2048 records a block covers); otherwise a dense list of n records.
constexpr int EPT = 8;
constexpr int DT_SLOTS = 4096;   // DEDUP: pair table of a block (2048 records)
// DEDUP (the dense rounds): the records of a block are first merged per component pair in an LDS table -- after a few
// rounds most records of a block connect the same few large components, and without this the lists stop shrinking
// and thousands of lanes hit the same best[] entries with atomics.
template <bool SEG, bool DEDUP>
__global__ __launch_bounds__(NTHR) void k_edge_round(const uint32_t *__restrict__ ea, const uint32_t *__restrict__ eb,
                                                     const uint32_t *__restrict__ ek, uint32_t n,
                                                     const uint32_t *__restrict__ segcount, uint32_t segcap,
                                                     const uint32_t *__restrict__ cur, unsigned long long *best, uint32_t B,
                                                     uint32_t *oa, uint32_t *ob, uint32_t *ok, uint32_t *ocount,
                                                     const uint32_t *__restrict__ dyn = nullptr) {
  __shared__ uint32_t wtot[NTHR / 64];
  __shared__ uint32_t bbase, dn;
  __shared__ unsigned long long dt_pair[DEDUP ? DT_SLOTS : 1];
  __shared__ uint32_t dt_key[DEDUP ? DT_SLOTS : 1];
  __shared__ uint16_t dt_list[DEDUP ? DT_SLOTS : 1];
  if (dyn) n = *dyn;   // (the dense rounds: the record count the round before left on the device)
  for (size_t i0 = (size_t)blockIdx.x * (NTHR * EPT); i0 < (size_t)n; i0 += (size_t)gridDim.x * (NTHR * EPT)) {
  size_t lim = n;   // records of this block's range that exist
  if (SEG) {
    const uint32_t seg = (uint32_t)(i0 / segcap);
    lim = (size_t)seg * segcap + segcount[seg];
    if (i0 >= lim) continue;
  }
  if (DEDUP) {
    for (int i = threadIdx.x; i < DT_SLOTS; i += NTHR) { dt_pair[i] = ~0ull; dt_key[i] = 0xFFFFFFFFu; }
    if (threadIdx.x == 0) dn = 0;
  }
  uint32_t a[EPT], b[EPT], k[EPT];
  bool ok_[EPT];
#pragma unroll
  for (int r = 0; r < EPT; r++) {
    const size_t i = i0 + (size_t)r * NTHR + threadIdx.x;
    ok_[r] = i < lim;
    a[r] = ok_[r] ? ea[i] : B;   // an absent record reads as (outside, outside): dead
    b[r] = ok_[r] ? eb[i] : B;
    k[r] = ok_[r] ? ek[i] : 0u;
  }
  uint32_t ca[EPT], cb[EPT];
#pragma unroll
  for (int r = 0; r < EPT; r++) { ca[r] = cur[a[r] & ~CLOSED]; cb[r] = cur[b[r] & ~CLOSED]; }
  bool live[EPT];
#pragma unroll
  for (int r = 0; r < EPT; r++) live[r] = ca[r] != cb[r] && !(ca[r] & cb[r] & CLOSED);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t nd = 0;
  if (DEDUP) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; r++) {
      if (live[r]) {
        const uint32_t lo = ca[r] < cb[r] ? ca[r] : cb[r], hi = ca[r] < cb[r] ? cb[r] : ca[r];
        if (pair_insert<DT_SLOTS>(dt_pair, dt_key, lo, hi, k[r])) live[r] = false;   // merged; else it stays a record of its own
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DT_SLOTS; i += NTHR) {
      const bool occ = dt_pair[i] != ~0ull;
      const unsigned long long bal = __ballot(occ);
      uint32_t base = 0;
      if (lane == 0 && bal) base = atomicAdd(&dn, (uint32_t)__popcll(bal));
      base = __shfl(base, 0, 64);
      if (occ) dt_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    nd = dn;
    for (uint32_t i = threadIdx.x; i < nd; i += NTHR) {
      const int sl = dt_list[i];
      const unsigned long long pr = dt_pair[sl];
      const uint32_t lo = (uint32_t)(pr >> 32), hi = (uint32_t)pr, key = dt_key[sl];
      const unsigned long long cl = ((unsigned long long)key << 32) | hi, ch = ((unsigned long long)key << 32) | lo;
      if (!(lo & CLOSED) && cl < best[lo]) atomicMin(&best[lo], cl);
      if (!(hi & CLOSED) && ch < best[hi]) atomicMin(&best[hi], ch);
    }
  }
  unsigned long long pa[EPT], pb[EPT];   // current best of either side (possibly stale: only a pre-check)
#pragma unroll
  for (int r = 0; r < EPT; r++) {
    pa[r] = (live[r] && !(ca[r] & CLOSED)) ? best[ca[r]] : 0ull;
    pb[r] = (live[r] && !(cb[r] & CLOSED)) ? best[cb[r]] : 0ull;
  }
#pragma unroll
  for (int r = 0; r < EPT; r++) {
    const unsigned long long candA = ((unsigned long long)k[r] << 32) | cb[r];
    const unsigned long long candB = ((unsigned long long)k[r] << 32) | ca[r];
    if (candA < pa[r]) atomicMin(&best[ca[r]], candA);
    if (candB < pb[r]) atomicMin(&best[cb[r]], candB);
  }
  // survivors -> next list, in order, one global atomic per block
  unsigned long long bal[EPT];
  uint32_t mine = 0;
#pragma unroll
  for (int r = 0; r < EPT; r++) { bal[r] = __ballot(live[r]); mine += (uint32_t)__popcll(bal[r]); }
  if (lane == 0) wtot[wv] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3] + nd;
    bbase = tot ? atomicAdd(ocount, tot) : 0;
  }
  __syncthreads();
  uint32_t off = bbase;
  if (DEDUP) {
    for (uint32_t i = threadIdx.x; i < nd; i += NTHR) {
      const int sl = dt_list[i];
      const unsigned long long pr = dt_pair[sl];
      oa[off + i] = (uint32_t)(pr >> 32); ob[off + i] = (uint32_t)pr; ok[off + i] = dt_key[sl];
    }
    off += nd;
  }
  for (int q = 0; q < wv; q++) off += wtot[q];
#pragma unroll
  for (int r = 0; r < EPT; r++) {
    if (live[r]) {
      const uint32_t g = off + (uint32_t)__popcll(bal[r] & ((1ull << lane) - 1ull));
      oa[g] = ca[r]; ob[g] = cb[r]; ok[g] = k[r];
    }
    off += (uint32_t)__popcll(bal[r]);
  }
  __syncthreads();   // (the next chunk of the block reuses the tables)
  }
}

__global__ __launch_bounds__(NTHR) void k_sum_segments(const uint32_t *__restrict__ segcount, uint32_t nseg, uint32_t *total) {
  __shared__ uint32_t part[NTHR];
  uint32_t v = 0;
  for (uint32_t i = threadIdx.x; i < nseg; i += NTHR) v += segcount[i];
  part[threadIdx.x] = v;
  __syncthreads();
  for (int st = NTHR / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = part[0];
}

// alive flags -> list of tiles for the next round (roughly ascending: blocks append in launch order) + count
__global__ __launch_bounds__(NTHR) void k_compact_alive(const uint8_t *__restrict__ alive, uint32_t ntiles,
                                                        uint32_t *list, uint32_t *count) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  const bool hit = i < ntiles && alive[i] != 0;
  const uint32_t slot = block_append(hit, count);
  if (hit) list[slot] = i;
}

__global__ __launch_bounds__(NTHR) void k_hook(const uint32_t *__restrict__ roots, uint32_t nroots,
                                               const unsigned long long *__restrict__ best,
                                               unsigned long long *link, const uint32_t *__restrict__ dyn = nullptr) {
  if (dyn) nroots = *dyn;
  for (uint32_t i = blockIdx.x * NTHR + threadIdx.x; i < nroots; i += gridDim.x * NTHR) {
    const uint32_t r = roots[i];
    const unsigned long long b = best[r];
    const uint32_t t = (uint32_t)b;
    bool keep_root = false;
    if (b == ~0ull) keep_root = true;  // no neighbouring component (cannot happen on a connected raster)
    else if (!(t & CLOSED)) {
      // mutual lowest pass: the pair merges, the smaller id stays root (the pass heights agree
      // because both sides see the same cell pair).  Closed components never choose, so never mutual.
      keep_root = ((uint32_t)best[t] == r) && (r < t);
    }
    link[r] = keep_root ? (unsigned long long)r : b;
  }
}

// pointer jumping over this round's hook forest, carrying the path maximum in the high word.
__global__ __launch_bounds__(NTHR) void k_chase_links(const uint32_t *__restrict__ roots, uint32_t nroots,
                                                      unsigned long long *link, int maxhops, uint32_t *flag,
                                                      const uint32_t *__restrict__ dyn = nullptr) {
  if (dyn) nroots = *dyn;
  for (uint32_t i = blockIdx.x * NTHR + threadIdx.x; i < nroots; i += gridDim.x * NTHR) {
    const uint32_t r = roots[i];
    unsigned long long l = __hip_atomic_load(&link[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t p = (uint32_t)l, m = (uint32_t)(l >> 32);
    if (p == r) continue;
    const uint32_t p0 = p;
    int hops = 0;
    bool unfinished = false;
    for (;;) {
      if (p & CLOSED) break;   // the outside / a frozen terminal: a root by definition
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
}

__global__ __launch_bounds__(NTHR) void k_update_basins(uint32_t *cur, uint32_t *acc,
                                                        const unsigned long long *__restrict__ link, uint32_t B,
                                                        const uint32_t *__restrict__ dyn = nullptr) {
  if (dyn && *dyn == 0u) return;   // (a round enqueued after the last: nothing hooked)
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  const uint32_t c = cur[b];
  if (c & CLOSED) return;
  const unsigned long long l = link[c];
  const uint32_t p = (uint32_t)l, m = (uint32_t)(l >> 32);
  if (p != c) {
    cur[b] = p;
    if (m > acc[b]) acc[b] = m;
  }
}

constexpr int RPT = 8;   // roots per thread: 2048 per block, so a round over 1e7 roots is 5e3 same-address atomics
__global__ __launch_bounds__(NTHR) void k_compact_roots(const uint32_t *__restrict__ roots_in, uint32_t nroots,
                                                        const unsigned long long *__restrict__ link,
                                                        uint32_t *roots_out, uint32_t *counter,
                                                        const uint32_t *__restrict__ dyn = nullptr) {
  __shared__ uint32_t wtot[NTHR / 64];
  __shared__ uint32_t bbase;
  if (dyn) nroots = *dyn;
  for (uint32_t i0 = blockIdx.x * (NTHR * RPT); i0 < nroots; i0 += gridDim.x * (NTHR * RPT)) {
  uint32_t r[RPT];
  bool ok[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) {
    const uint32_t i = i0 + q * NTHR + threadIdx.x;
    ok[q] = i < nroots;
    r[q] = ok[q] ? roots_in[i] : 0u;
  }
  bool keep[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) keep[q] = ok[q] && (uint32_t)link[r[q]] == r[q];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long bal[RPT];
  uint32_t mine = 0;
#pragma unroll
  for (int q = 0; q < RPT; q++) { bal[q] = __ballot(keep[q]); mine += (uint32_t)__popcll(bal[q]); }
  if (lane == 0) wtot[wv] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    bbase = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  uint32_t off = bbase;
  for (int k = 0; k < wv; k++) off += wtot[k];
#pragma unroll
  for (int q = 0; q < RPT; q++) {
    if (keep[q]) roots_out[off + (uint32_t)__popcll(bal[q] & ((1ull << lane) - 1ull))] = r[q];
    off += (uint32_t)__popcll(bal[q]);
  }
  __syncthreads();   // (the next chunk of the block reuses wtot / bbase)
  }
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
// 6. row-block shards (Barnes 2016 tile protocol; reference programs/parallel_priority_flood/main.cpp,
//    Zhou2016pf.hpp): after the local phase every cell knows its watershed label (a cut-row terminal or
//    the outside) and its locally filled level.  What the global solve needs from a shard is, for every
//    pair of adjacent watersheds, the lowest pass between them (WatershedsMeet, Zhou2016pf.hpp:37-62).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t shard_label(uint32_t c, const uint32_t *__restrict__ lab,
                                                const uint32_t *__restrict__ cur, const uint32_t *__restrict__ tid,
                                                uint32_t B) {
  const uint32_t l = cur[lab[c]] & ~CLOSED;
  return l == B ? NO_TID : tid[l];   // NO_TID doubles as "the outside" in exported edges
}

// open-addressing table of (min(la,lb) << 32 | max(la,lb)) + 1 -> lowest pass key.
// One 64x32 tile per block: the watershed label (three dependent gathers: basin -> component -> terminal id) and
// the locally filled level of every cell of the tile and of its forward halo are fetched ONCE, in three batches,
// into LDS; the pair tests then run on LDS and only cells on a watershed boundary touch the global table.
constexpr int EW = 64, EH = 32, ELW = EW + 2, ELH = EH + 1;
constexpr uint32_t E_INVALID = 0xFFFFFFFEu;   // outside the raster: pairs with it are skipped
template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_shard_edges(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                                      const uint32_t *__restrict__ cur, const uint32_t *__restrict__ acc,
                                                      const uint32_t *__restrict__ tid, uint32_t B, int w, int h,
                                                      unsigned long long *hkeys, uint32_t *hvals, uint32_t hmask,
                                                      uint32_t *overflow, uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint32_t sl[ELH * ELW];
  __shared__ uint32_t sw[ELH * ELW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * EW, y0 = (int)(t / tilesX) * EH;
  constexpr int IPT = (ELH * ELW + NTHR - 1) / NTHR;
  uint32_t lv[IPT], kv[IPT];
  bool ok[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {   // batch 1: basin label and elevation key
    const int i = threadIdx.x + r * NTHR;
    const int ly = i / ELW, lx = i - ly * ELW;
    const int gx = x0 - 1 + lx, gy = y0 + ly;
    ok[r] = i < ELH * ELW && gx >= 0 && gx < w && gy < h;
    lv[r] = B;
    kv[r] = 0;
    if (ok[r]) { lv[r] = lab[(size_t)gy * w + gx]; kv[r] = Key32<T>::to(z[(size_t)gy * w + gx]); }
  }
  uint32_t cv[IPT], av[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {   // batch 2: component and filled level (label B = the outside: cur[B], acc[B] exist)
    cv[r] = cur[lv[r]] & ~CLOSED;
    av[r] = acc[lv[r]];
  }
  uint32_t tv[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) tv[r] = cv[r] == B ? NO_TID : tid[cv[r]];   // batch 3: NO_TID doubles as "the outside"
#pragma unroll
  for (int r = 0; r < IPT; r++) {
    const int i = threadIdx.x + r * NTHR;
    if (i < ELH * ELW) {
      sl[i] = ok[r] ? tv[r] : E_INVALID;
      sw[i] = (lv[r] != B && av[r] > kv[r]) ? av[r] : kv[r];
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & (EW - 1), ly0 = threadIdx.x >> 6;
#pragma unroll 2
  for (int j = 0; j < EH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int o = ly * ELW + lx + 1;
    const uint32_t la = sl[o];
    if (la == E_INVALID) continue;
    const uint32_t wa = sw[o];
    // forward neighbours only (E, SW, S, SE): every adjacent pair is seen once
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (TOPO == 4 && (k == 1 || k == 3)) continue;
      const int q = o + (k == 0 ? 1 : k == 1 ? ELW - 1 : k == 2 ? ELW : ELW + 1);
      const uint32_t lb = sl[q];
      if (lb == la || lb == E_INVALID) continue;
      const uint32_t wb = sw[q];
      const uint32_t pass = wa > wb ? wa : wb;
      const uint32_t lo = la < lb ? la : lb, hi = la < lb ? lb : la;
      const unsigned long long key = (((unsigned long long)lo << 32) | hi) + 1ull;   // never 0 (lo != hi)
      uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & hmask;
      for (uint32_t probe = 0;; probe++) {
        if (probe > hmask) { *overflow = 1; break; }
        unsigned long long cur_k = __hip_atomic_load(&hkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur_k == 0) cur_k = atomicCAS(&hkeys[slot], 0ull, key);
        if (cur_k == 0 || cur_k == key) {
          if (__hip_atomic_load(&hvals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pass) atomicMin(&hvals[slot], pass);
          break;
        }
        slot = (slot + 1) & hmask;
      }
    }
  }
}


__global__ __launch_bounds__(NTHR) void k_shard_edges_compact(const unsigned long long *__restrict__ hkeys,
                                                              const uint32_t *__restrict__ hvals, uint32_t hsize,
                                                              uint32_t *edges, uint32_t *nedges) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;   // grid covers whole waves
  const bool used = i < hsize && hkeys[i] != 0;
  const unsigned long long bal = __ballot(used);
  if (bal == 0) return;
  const int lane = threadIdx.x & 63, leader = (int)__ffsll((long long)bal) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(nedges, (uint32_t)__popcll(bal));
  base = __shfl(base, leader, 64);
  if (used) {
    const unsigned long long k = hkeys[i] - 1ull;
    const uint32_t e = base + __popcll(bal & ((1ull << lane) - 1ull));
    edges[3 * e] = (uint32_t)(k >> 32);
    edges[3 * e + 1] = (uint32_t)k;
    edges[3 * e + 2] = hvals[i];
  }
}

// raise every basin by the globally solved level of its watershed's terminal (SecondRound,
// parallel_priority_flood/main.cpp:315-321): levels[tid] for tid in [0, 2w)
__global__ __launch_bounds__(NTHR) void k_shard_apply(const uint32_t *__restrict__ cur, uint32_t *acc,
                                                      const uint32_t *__restrict__ tid,
                                                      const uint32_t *__restrict__ levels, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  const uint32_t l = cur[b] & ~CLOSED;
  if (l == B) return;   // drains to the true border inside this shard: already final
  const uint32_t g = levels[tid[l]];
  if (g > acc[b]) acc[b] = g;
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Device buffers that outlive the local phase.
struct FillBuffers {
  uint32_t *lab = nullptr, *cur = nullptr, *acc = nullptr, *tid = nullptr;
  uint32_t B = 0;
  bool trivial = false;   // nothing to raise (no interior, or no pits)
  // the compact-label local phase of a row-block shard (r04): no 32-bit label per cell, but the 16-bit slots, the tiles'
  // node bases and counts, node -> basin (curN), and per node its level (lvl) and watershed terminal (nodeW)
  bool compact = false;
  uint16_t *lab16 = nullptr;
  uint32_t *tile_base = nullptr, *tile_count = nullptr, *curN = nullptr, *lvl = nullptr, *nodeW = nullptr;
  unsigned long long *counters = nullptr;
  uint32_t rcap = 0, nstripes = 0, nnmax = 0;
};

struct BufAlloc {   // where persistent buffers come from: the shared workspace, or owned hipMalloc
  bool owned;
  std::vector<void *> *owned_list;
  bool shard_ws = false;   // workspace buffers under their own names ("shard." + name): the one cached shard
  template <class U>
  U *get(const char *name, size_t count) {
    if (!owned && shard_ws) return Workspace::get().buf<U>((std::string("shard.") + name).c_str(), count);
    if (!owned) return Workspace::get().buf<U>(name, count);
    void *p = nullptr;
    RD_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(U)));
    owned_list->push_back(p);
    return static_cast<U *>(p);
  }
};

// Phases 1-4: descent forest, basins, Boruvka rounds.  With open_top/open_bottom the first/last row is
// a cut row of a row-block shard whose cells are frozen terminals.
template <class T, int TOPO>
static void fill_local_phase(const T *d_z, int w, int h, int open_top, int open_bottom, BufAlloc alloc, FillBuffers &fb,
                             hipStream_t s, const uint8_t *outlet = nullptr) {
  const uint64_t n64 = (uint64_t)w * (uint64_t)h;
  if (n64 > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: raster (or shard) has more than 2^31-65536 cells");
  const uint32_t n = (uint32_t)n64;
  g_stats = rdgpu_fill_stats{n64, 0, 0, 0, 0, (uint32_t)(TW * TH), 0};
  fb = FillBuffers();
  const bool sharded = open_top || open_bottom;
  if (w <= 2 || (!sharded && h <= 2)) { fb.trivial = true; return; }  // every cell is a border cell

  Workspace &ws = Workspace::get();
  uint32_t *lab = alloc.get<uint32_t>("fill.lab", n);
  fb.lab = lab;
  uint32_t *dflags = ws.buf<uint32_t>("fill.flags", 16);  // [0] chase flag, [1] pit total, [2] root counter
  uint32_t *hw = ws.host_words();

  const uint32_t tilesX = cdiv(w, TW), tilesY = cdiv(h, TH), ntiles = tilesX * tilesY;
  const uint32_t sgrid = std::min(cdiv(n, NTHR), 256u * 32u);  // grid-stride 1-D kernels

  // descent forest: one word per cell (basin id / pending pointer / off the raster); pits are numbered by the same kernel
  const uint32_t dtx = cdiv(w, DW), dnt = dtx * cdiv(h, DH);
  RD_HIP(hipMemsetAsync(dflags, 0, 2 * sizeof(uint32_t), s));
  // naturally aligned quads when every row starts on a quad boundary; the any-width quad loads otherwise
  const bool vec = (w % 4) == 0 && (reinterpret_cast<uintptr_t>(d_z) % (4 * sizeof(T))) == 0;
  if (vec)
    RD_LAUNCH("fill.descent", (k_descent<T, TOPO, true>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, d_z, lab, dflags + 1, w,
              h, dtx, dnt, open_top, open_bottom, outlet);
  else
    RD_LAUNCH("fill.descent", (k_descent<T, TOPO, false>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, d_z, lab, dflags + 1, w,
              h, dtx, dnt, open_top, open_bottom, outlet);
  RD_HIP(hipMemcpyAsync(hw, dflags + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  const uint32_t B = hw[0];
  g_stats.basins = B;
  fb.B = B;
  if (B == 0) { fb.trivial = true; return; }  // no pits and no terminals: nothing to raise
  RD_LAUNCH("fill.tile_label", k_tile_label, dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, lab, w, h, dtx, dnt, B, 256, dflags);
  RD_HIP(hipMemcpyAsync(hw, dflags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  g_stats.jump_passes = 1;
  if (hw[0] != 0) {
    // pathologically long tile-to-tile chains: compress them (each pass shortens every path >= 32x), relabel
    for (;;) {
      RD_HIP(hipMemsetAsync(dflags, 0, sizeof(uint32_t), s));
      RD_LAUNCH("fill.chase", k_chase, dim3(sgrid), dim3(NTHR), 0, s, lab, n, 32, dflags);
      RD_HIP(hipMemcpyAsync(hw, dflags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      g_stats.jump_passes++;
      if (hw[0] == 0) break;
    }
    RD_LAUNCH("fill.label_cells", k_label_cells, dim3(sgrid), dim3(NTHR), 0, s, lab, n, B);
  }

  uint32_t *cur = alloc.get<uint32_t>("fill.cur", (size_t)B + 1);
  uint32_t *acc = alloc.get<uint32_t>("fill.acc", (size_t)B + 1);
  fb.cur = cur;
  fb.acc = acc;
  uint32_t *tid = nullptr;
  if (sharded) {
    tid = alloc.get<uint32_t>("fill.tid", (size_t)B + 1);
    fb.tid = tid;
    RD_HIP(hipMemsetAsync(tid, 0xFF, ((size_t)B + 1) * sizeof(uint32_t), s));
    RD_LAUNCH("fill.mark_terminals", k_mark_terminals, dim3(cdiv(w, NTHR)), dim3(NTHR), 0, s, lab, tid, w, h, open_top,
              open_bottom);
  }
  unsigned long long *best = ws.buf<unsigned long long>("fill.best", (size_t)B + 1);
  unsigned long long *link = ws.buf<unsigned long long>("fill.link", (size_t)B + 1);
  uint32_t *rootsA = ws.buf<uint32_t>("fill.rootsA", B);
  uint32_t *rootsB = ws.buf<uint32_t>("fill.rootsB", B);
  RD_HIP(hipMemsetAsync(dflags + 2, 0, sizeof(uint32_t), s));
  RD_LAUNCH("fill.init_tables", k_init_tables, dim3(cdiv((uint64_t)B + 1, NTHR)), dim3(NTHR), 0, s, cur, acc, link,
            rootsA, dflags + 2, (const uint32_t *)tid, B);
  RD_HIP(hipMemcpyAsync(hw, dflags + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));

  uint32_t nroots = hw[0];
  // per-tile flag "still holds a component boundary" (ping-pong); round 1 visits every tile
  uint8_t *alive = ws.buf<uint8_t>("fill.alive", ntiles);
  uint32_t *tlist = ws.buf<uint32_t>("fill.tlist", ntiles);
  uint32_t nlive = ntiles;
  bool first = true;
  // The pair list (see k_scan EMIT): round 1 is the only raster pass unless something overflows.  Capacity: 12
  // records per basin (measured at S3: 4.6), at most one per two cells; per segment a multiple of a k_edge_round block.
  // RDGPU_FILL_EDGES=0 keeps every round on the raster; RDGPU_FILL_EDGE_CAP=<records> overrides the list capacity
  // (both exist for the tests of the overflow fallback and for A/B timing).
  const char *env_edges = getenv("RDGPU_FILL_EDGES"), *env_cap = getenv("RDGPU_FILL_EDGE_CAP");
  const bool edges_enabled = !(env_edges && env_edges[0] == '0');
  const char *env_dedup = getenv("RDGPU_FILL_DEDUP");
  const bool dedup = !(env_dedup && env_dedup[0] == '0');
  bool edge_mode = false;           // rounds 2.. run on the pair list
  EdgeOut eo{};
  uint32_t nseg = 1, nedges = 0;
  uint32_t *elist[2] = {nullptr, nullptr};   // ping-pong record buffers: a | b | k planes of ecap[] records
  size_t ecap[2] = {0, 0};
  int ein = 0;
  bool eseg = true;                 // the input list is still the segmented one of the raster pass
  if (edges_enabled && nroots > 0) {
    while (nseg < ESEG && (uint64_t)nseg * 128 <= ntiles) nseg *= 2;
    const uint64_t cap = env_cap ? strtoull(env_cap, nullptr, 10) : std::min<uint64_t>(12ull * B, n / 2) + 2048;
    const uint32_t segcap = cdiv(cdiv(cap, nseg), NTHR * EPT) * (NTHR * EPT);
    ecap[0] = (size_t)nseg * segcap;
    elist[0] = ws.buf<uint32_t>("fill.edges0", 3 * ecap[0]);
    uint32_t *segcount = ws.buf<uint32_t>("fill.segcount", nseg);
    RD_HIP(hipMemsetAsync(segcount, 0, nseg * sizeof(uint32_t), s));
    RD_HIP(hipMemsetAsync(dflags + 4, 0, 2 * sizeof(uint32_t), s));
    eo = EdgeOut{elist[0], elist[0] + ecap[0], elist[0] + 2 * ecap[0], segcount, segcap, nseg - 1, dflags + 5};
  }
  const bool emit = eo.a != nullptr;
  while (nroots > 0) {
    const uint32_t rgrid = cdiv(nroots, NTHR);
    RD_LAUNCH("fill.best_reset", k_best_reset, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, best);
    RD_HIP(hipMemsetAsync(dflags + 3, 0, sizeof(uint32_t), s));
    if (edge_mode) {
      // contract the pair list: in -> out (dense), proposals to best[]
      const int eout = ein ^ 1;
      RD_HIP(hipMemsetAsync(dflags + 4, 0, sizeof(uint32_t), s));
      const uint32_t *ia = elist[ein], *ib = elist[ein] + ecap[ein], *ik = elist[ein] + 2 * ecap[ein];
      uint32_t *oa = elist[eout], *ob = elist[eout] + ecap[eout], *ok = elist[eout] + 2 * ecap[eout];
      // (merging per pair in the first list round was measured slower: 2.5 vs 1.4 ms -- its records are already
      // merged per tile)
      if (eseg)
        RD_LAUNCH("fill.edge_round", (k_edge_round<true, false>), dim3(cdiv(ecap[ein], NTHR * EPT)), dim3(NTHR), 0, s, ia, ib, ik,
                  (uint32_t)ecap[ein], (const uint32_t *)eo.segcount, eo.segcap, (const uint32_t *)cur, best, B, oa, ob, ok,
                  dflags + 4);
      else if (nedges > 0 && dedup)
        RD_LAUNCH("fill.edge_round", (k_edge_round<false, true>), dim3(cdiv(nedges, NTHR * EPT)), dim3(NTHR), 0, s, ia, ib, ik,
                  nedges, (const uint32_t *)nullptr, 0u, (const uint32_t *)cur, best, B, oa, ob, ok, dflags + 4);
      else if (nedges > 0)
        RD_LAUNCH("fill.edge_round", (k_edge_round<false, false>), dim3(cdiv(nedges, NTHR * EPT)), dim3(NTHR), 0, s, ia, ib, ik,
                  nedges, (const uint32_t *)nullptr, 0u, (const uint32_t *)cur, best, B, oa, ob, ok, dflags + 4);
      eseg = false;
      ein = eout;
    } else if (nlive > 0) {
#define RD_SCAN(FIRST_, VEC_, EMIT_, LIST, NWORK)                                                                \
  RD_LAUNCH("fill.scan", (k_scan<T, TOPO, FIRST_, VEC_, EMIT_>), dim3(xcd_grid(NWORK)), dim3(NTHR), 0, s, d_z, lab, cur, best, w, \
            h, B, tilesX, ntiles, (const uint32_t *)(LIST), (uint32_t)(NWORK), alive, eo)
#define RD_SCAN_V(FIRST_, EMIT_, LIST, NWORK)                                                                    \
  { if (vec) RD_SCAN(FIRST_, true, EMIT_, LIST, NWORK); else RD_SCAN(FIRST_, false, EMIT_, LIST, NWORK); }
      const char *env_first = getenv("RDGPU_FILL_FIRST");   // =0: the pair pass gathers cur[label] like a shard's (A/B probe)
      if (first && !sharded && !(env_first && env_first[0] == '0')) { if (emit) RD_SCAN_V(true, true, nullptr, ntiles) else RD_SCAN_V(true, false, nullptr, ntiles) }
      else if (first) { if (emit) RD_SCAN_V(false, true, nullptr, ntiles) else RD_SCAN_V(false, false, nullptr, ntiles) }
      else RD_SCAN_V(false, false, tlist, nlive)
#undef RD_SCAN_V
#undef RD_SCAN
      RD_LAUNCH("fill.compact_alive", k_compact_alive, dim3(cdiv(ntiles, NTHR)), dim3(NTHR), 0, s, (const uint8_t *)alive, ntiles,
                tlist, dflags + 3);
      if (first && emit)
        RD_LAUNCH("fill.sum_segments", k_sum_segments, dim3(1), dim3(NTHR), 0, s, (const uint32_t *)eo.segcount, nseg, dflags + 4);
      g_stats.scan_tiles += first ? ntiles : nlive;
    }
    RD_LAUNCH("fill.hook", k_hook, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, best, link);
    for (;;) {
      RD_HIP(hipMemsetAsync(dflags, 0, sizeof(uint32_t), s));
      RD_LAUNCH("fill.chase_links", k_chase_links, dim3(rgrid), dim3(NTHR), 0, s, rootsA, nroots, link, 32, dflags);
      RD_HIP(hipMemcpyAsync(hw, dflags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      if (hw[0] == 0) break;
    }
    RD_LAUNCH("fill.update_basins", k_update_basins, dim3(cdiv(B, NTHR)), dim3(NTHR), 0, s, cur, acc, link, B);
    RD_HIP(hipMemsetAsync(dflags + 2, 0, sizeof(uint32_t), s));
    RD_LAUNCH("fill.compact_roots", k_compact_roots, dim3(cdiv(nroots, NTHR * RPT)), dim3(NTHR), 0, s, rootsA, nroots, link,
              rootsB, dflags + 2);
    RD_HIP(hipMemcpyAsync(hw, dflags + 2, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    const uint32_t next = hw[0];
    if (next >= nroots) throw Error(RDGPU_ERR_HIP, "rdgpu_fill: contraction made no progress (internal error)");
    nroots = next;
    nlive = hw[1];
    if (first && emit && hw[3] == 0 && nroots > 0) {
      // the raster pass recorded every adjacent component pair: the remaining rounds run on that list
      edge_mode = true;
      nedges = hw[2];
      g_stats.edge_records = nedges;
      ecap[1] = std::max<size_t>(nedges, 1);
      elist[1] = ws.buf<uint32_t>("fill.edges1", 3 * ecap[1]);
    } else if (edge_mode) {
      nedges = hw[2];
    }
    first = false;
    std::swap(rootsA, rootsB);
    g_stats.rounds++;
  }
}

// The same in tile order (64 x 32 tiles, XCD-banded, aligned quads): the cells of a tile share a few dozen basins, so
// the acc[] lines stay in the XCD's L2 while the tile is worked on -- in row order every row segment of a basin
// fetched its line again (5.5 GB of gathers on top of 12.8 GB of rows at S3).  A quad is written back only when one of
// its cells is raised (unchanged cells get their own bits back).
template <class T>
__global__ __launch_bounds__(NTHR) void k_finalize_tiled(T *z, const uint32_t *__restrict__ lab,
                                                         const uint32_t *__restrict__ acc, int w, int h, uint32_t B,
                                                         uint32_t tilesX, uint32_t ntiles) {
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
  const int gx = x0 + 4 * (threadIdx.x & 15), ry = threadIdx.x >> 4;   // 16 quads per row, rows ry and ry + 16
  struct alignas(4 * sizeof(T)) ZQ { T v[4]; };
  ZQ zq[2];
  uint4 lq[2];
  bool ok[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int gy = y0 + ry + 16 * r;
    ok[r] = gx < w && gy < h;
    const size_t g = ok[r] ? (size_t)gy * w + gx : 0;
    zq[r] = *reinterpret_cast<const ZQ *>(z + g);
    lq[r] = *reinterpret_cast<const uint4 *>(lab + g);
  }
  uint32_t L[2][4];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t l[4] = {lq[r].x, lq[r].y, lq[r].z, lq[r].w};
#pragma unroll
    for (int e = 0; e < 4; e++) L[r][e] = acc[l[e]];   // acc[B] (the outside) is 0: never above a key that matters
  }
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t l[4] = {lq[r].x, lq[r].y, lq[r].z, lq[r].w};
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (l[e] != B && L[r][e] > Key32<T>::to(zq[r].v[e])) { zq[r].v[e] = Key32<T>::from(L[r][e]); any = true; }
    }
    if (any && ok[r]) *reinterpret_cast<ZQ *>(z + (size_t)(y0 + ry + 16 * r) * w + gx) = zq[r];
  }
}

template <class T>
static void fill_finalize(T *d_z, int w, int h, const FillBuffers &fb, hipStream_t s) {
  if (fb.trivial) return;
  const uint32_t n = (uint32_t)((uint64_t)w * h);
  if ((w % 4) == 0 && (reinterpret_cast<uintptr_t>(d_z) % (4 * sizeof(T))) == 0) {
    const uint32_t tilesX = cdiv(w, TW), ntiles = tilesX * cdiv(h, TH);
    RD_LAUNCH("fill.finalize", (k_finalize_tiled<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, fb.lab, fb.acc, w, h, fb.B,
              tilesX, ntiles);
    return;
  }
  const uint32_t sgrid = std::min(cdiv(n, NTHR), 256u * 32u);
  RD_LAUNCH("fill.finalize", (k_finalize<T>), dim3(sgrid), dim3(NTHR), 0, s, d_z, fb.lab, fb.acc, n, fb.B);
}

// a shard's finish on compact labels: the nodes' levels from the (raised) basin levels, then the raster pass
template <class T>
static void fill_finalize16(T *d_z, int w, int h, const FillBuffers &fb, hipStream_t s);

static void check_fill_args(const void *p, int w, int h, int topology) {
  if (!p) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: width and height must be positive");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: topology must be 8 or 4");  // depressions.hpp:19-20
}

// pit_mask<topo>(elevations, pit_mask) (reference depressions/Barnes2014.hpp:593-676; apps/rd_depressions_mask.cpp):
// the same flood, recording which cells it would raise instead of raising them.  1 = the cell lies strictly below the
// level it is reached at (W > z), 0 = not in a pit, 3 = NoData (every NoData cell, :668-669).
template <class T>
__global__ __launch_bounds__(NTHR) void k_pit_mask(const T *__restrict__ z, T nodata, const uint32_t *__restrict__ lab,
                                                   const uint32_t *__restrict__ acc, uint8_t *__restrict__ mask, uint32_t n,
                                                   uint32_t B, int trivial) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c64 = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c64 < n; c64 += stride) {
    const uint32_t c = (uint32_t)c64;
    const T e = z[c];
    uint8_t m = 0;
    if (e == nodata) m = 3;
    else if (!trivial) {
      const uint32_t b = lab[c];
      if (b != B && acc[b] > Key32<T>::to(e)) m = 1;
    }
    mask[c] = m;
  }
}

template <class T>
static void pit_mask_device(const T *d_z, T nodata, int w, int h, int topology, uint8_t *d_mask, hipStream_t s) {
  check_fill_args(d_z, w, h, topology);
  if (!d_mask) throw Error(RDGPU_ERR_ARG, "rdgpu_pit_mask: null mask pointer");
  FillBuffers fb;
  BufAlloc ws_alloc{false, nullptr};
  if (topology == 8) fill_local_phase<T, 8>(d_z, w, h, 0, 0, ws_alloc, fb, s);
  else fill_local_phase<T, 4>(d_z, w, h, 0, 0, ws_alloc, fb, s);
  const uint32_t n = (uint32_t)((uint64_t)w * h);
  RD_LAUNCH("fill.pit_mask", (k_pit_mask<T>), dim3(std::min(cdiv(n, NTHR), 256u * 32u)), dim3(NTHR), 0, s, d_z, nodata,
            (const uint32_t *)fb.lab, (const uint32_t *)fb.acc, d_mask, n, fb.B, fb.trivial ? 1 : 0);
}

template <class T>
static void pit_mask_host(const T *dem, T nodata, int w, int h, int topology, uint8_t *mask) {
  check_fill_args(dem, w, h, topology);
  if (!mask) throw Error(RDGPU_ERR_ARG, "rdgpu_pit_mask: null mask pointer");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  uint8_t *dm = Workspace::get().buf<uint8_t>("host.dirs", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  pit_mask_device<T>(d, nodata, w, h, topology, dm, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(mask, dm, n, hipMemcpyDeviceToHost));
}

// ------------------------------------------------------------------------------------------
// PriorityFlood_Barnes2014_max_dep<topo>(elevations, max_dep_size) (reference depressions/Barnes2014.hpp:844-931;
// apps/rd_depressions_flood.cpp:16-19 with a non-zero third argument; goldens tests/depressions/testdem1.{1,2}.out,
// tests/tests.cpp:273-287): the same flood, but a depression is only raised when it has at most max_dep_size cells.
//
// What the reference counts as ONE depression is one run of its pit queue (:893-906): the cell c popped from the heap
// (level L = z(c)) floods every not yet visited cell BELOW L that it can reach through such cells (:918-921), and the
// run's cells are raised to L together, or not at all, when the next cell is popped from the heap.  In terms of the
// plain fill W: the run consists of raised cells (z < W = L); adjacent raised cells always have the same W, so the
// connected components of the raised cells ("pockets") are what can be flooded, each by the FIRST popped cell of
// elevation exactly L next to it, and a run = the pockets that share that cell.
//   * pockets: the raised cells of one basin of the descent forest are connected (every raised cell descends to the pit
//     through lower, hence raised, cells of its basin), so pockets are unions of basins -- a union-find over basins,
//     united wherever two raised cells of different basins touch (k_md_pockets; it also counts the raised cells per
//     basin, one atomic per run of equal labels in a wavefront row).
//   * spawner: the lowest cell index among the un-raised cells of elevation L next to the pocket (k_md_spawn).  When a
//     DEM has no two cells of equal elevation there is exactly one candidate; with ties the reference takes whichever
//     std::priority_queue pops first, which this rule cannot promise to reproduce.
//   * runs: every spawner unites the pockets it spawns (k_md_runs), sizes are summed per run, cells of runs with at
//     most max_dep_size cells are raised (k_md_apply).
// (The reference never applies the run started by the LAST cell popped from its heap, :891/:900 -- the loop ends first.
// On a DEM without equal elevations that cell, the highest un-raised one, cannot have started a run: a pocket has more
// than one rim cell and all of them would have to be the highest cell.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t md_find(uint32_t *par, uint32_t x) {
  uint32_t p = __hip_atomic_load(&par[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    x = p;
    p = __hip_atomic_load(&par[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}
__device__ __forceinline__ void md_unite(uint32_t *par, uint32_t a, uint32_t b) {
  for (;;) {
    a = md_find(par, a);
    b = md_find(par, b);
    if (a == b) return;
    if (a < b) { const uint32_t t = a; a = b; b = t; }   // hook the larger root under the smaller: acyclic under any interleaving
    const uint32_t old = atomicMin(&par[a], b);
    if (old == a) return;
    a = old;
  }
}

__global__ __launch_bounds__(NTHR) void k_md_init(uint32_t *par, uint32_t *cnt, uint32_t *spawn, uint32_t *size, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  par[b] = b; cnt[b] = 0; spawn[b] = 0xFFFFFFFFu; size[b] = 0;
}

template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_md_pockets(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                                     const uint32_t *__restrict__ acc, uint32_t *par, uint32_t *cnt, int w,
                                                     int h, uint32_t B) {
  // one wavefront per 64-cell row segment (grid-stride over segments)
  const uint32_t segsX = ((uint32_t)w + 63u) / 64u;
  const uint64_t nseg = (uint64_t)segsX * (uint64_t)h;
  const int lane = threadIdx.x & 63;
  for (uint64_t sgi = (uint64_t)blockIdx.x * (NTHR / 64) + (threadIdx.x >> 6); sgi < nseg; sgi += (uint64_t)gridDim.x * (NTHR / 64)) {
    const int y = (int)(sgi / segsX), x = (int)(sgi % segsX) * 64 + lane;
    const bool in = x < w;
    const size_t c = (size_t)y * w + (in ? x : 0);
    const uint32_t b = in ? lab[c] : B;
    const bool raised = in && b != B && acc[b] > Key32<T>::to(z[c]);
    if (raised) {
      // forward neighbours: every adjacent pair of raised cells is looked at once
      const int nx[4] = {1, 1, 0, -1}, ny[4] = {0, 1, 1, 1};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (TOPO == 4 && (k == 1 || k == 3)) continue;
        const int xx = x + nx[k], yy = y + ny[k];
        if (xx < 0 || xx >= w || yy >= h) continue;
        const size_t q = (size_t)yy * w + xx;
        const uint32_t bq = lab[q];
        if (bq == b || bq == B) continue;
        if (acc[bq] > Key32<T>::to(z[q])) md_unite(par, b, bq);
      }
    }
    // raised cells per basin: one atomic per run of equal labels along the row segment
    const uint32_t key = raised ? b : 0xFFFFFFFFu;
    const uint32_t left = __shfl_up(key, 1, 64);
    const bool head = raised && (lane == 0 || left != key);
    const unsigned long long heads = __ballot(head), rs = __ballot(raised);
    if (head) {
      const unsigned long long above = lane < 63 ? (heads >> (lane + 1)) : 0ull;
      const int next = above ? lane + __ffsll((long long)above) : 64;   // next run head (of any label)
      // the run ends at the next head or at the first un-raised lane
      const unsigned long long notr = lane < 63 ? ((~rs) >> (lane + 1)) : 0ull;
      const int stop = notr ? lane + __ffsll((long long)notr) : 64;
      const int end = next < stop ? next : stop;
      atomicAdd(&cnt[b], (uint32_t)(end - lane));
    }
  }
}

template <class T, int TOPO, int PASS>
__global__ __launch_bounds__(NTHR) void k_md_spawn(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                                   const uint32_t *__restrict__ acc, uint32_t *par, uint32_t *spawn, int w,
                                                   int h, uint32_t B) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t b = lab[c];
    const uint32_t kz = Key32<T>::to(z[c]);
    if (b != B && acc[b] > kz) continue;   // raised cells are flooded, they do not flood
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    uint32_t first = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (TOPO == 4 && (k & 1)) continue;
      const int dx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
      const int xx = x + dx[k], yy = y + dy[k];
      if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
      const size_t q = (size_t)yy * w + xx;
      const uint32_t bq = lab[q];
      if (bq == B) continue;
      const uint32_t L = acc[bq];
      if (L != kz || !(L > Key32<T>::to(z[q]))) continue;   // a raised neighbour filled to exactly this cell's elevation
      const uint32_t r = md_find(par, bq);
      if (PASS == 0) {
        if ((uint32_t)c < __hip_atomic_load(&spawn[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&spawn[r], (uint32_t)c);
      } else if (__hip_atomic_load(&spawn[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)c) {
        if (first == 0xFFFFFFFFu) first = r;
        else md_unite(par, first, r);        // the pockets this cell floods are one run
      }
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_md_sizes(uint32_t *par, const uint32_t *__restrict__ cnt, uint32_t *size, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  const uint32_t c = cnt[b];
  if (c) atomicAdd(&size[md_find(par, b)], c);
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_md_apply(T *z, const uint32_t *__restrict__ lab, const uint32_t *__restrict__ acc,
                                                   uint32_t *par, const uint32_t *__restrict__ size, uint64_t n, uint32_t B,
                                                   uint64_t max_dep) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t b = lab[c];
    if (b == B) continue;
    const uint32_t L = acc[b];
    if (!(L > Key32<T>::to(z[c]))) continue;
    if ((uint64_t)size[md_find(par, b)] <= max_dep) z[c] = Key32<T>::from(L);
  }
}

// ---- tie detector (r05) -------------------------------------------------------------------------------------------------
// With equal elevations a pocket can have SEVERAL cells able to flood it (un-raised cells of elevation exactly L next to it);
// which of them the reference's std::priority_queue pops first decides which pockets end up in one run, hence whether the
// run passes the size limit.  Everything else about the result is order free.  So: ncand[pocket] = the number of distinct
// candidate cells; pockets that share ANY candidate cell are joined into a CLUSTER (par2, a coarser union-find than the
// runs, which only join through the chosen spawner); a cluster holding a pocket with two or more candidates is tie-flagged.
// Inside an unflagged cluster every pocket has exactly one possible flooding cell, so its runs -- and with them the output
// on all of the cluster's cells -- do not depend on the pop order.  Inside a flagged one the order matters only where the
// size limit can fall either way (k_md_tie_mask): S3, limit 100: 3420 of 4.18 million pockets have two candidates, most of
// them lakes far above the limit.  A difference from the reference on an unmasked cell would be a bug, not a tie
// (tests/test_s3_f2_gpu.py and tests/test_maxdep_gpu.py assert that there is none).
template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_md_ties(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                                  const uint32_t *__restrict__ acc, uint32_t *par, uint32_t *par2, uint32_t *ncand,
                                                  int w, int h, uint32_t B) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t b = lab[c];
    const uint32_t kz = Key32<T>::to(z[c]);
    if (b != B && acc[b] > kz) continue;   // raised cells are flooded, they do not flood
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    uint32_t seen[8];
    int ns = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (TOPO == 4 && (k & 1)) continue;
      const int dx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
      const int xx = x + dx[k], yy = y + dy[k];
      if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
      const size_t q = (size_t)yy * w + xx;
      const uint32_t bq = lab[q];
      if (bq == B) continue;
      const uint32_t L = acc[bq];
      if (L != kz || !(L > Key32<T>::to(z[q]))) continue;   // a raised neighbour filled to exactly this cell's elevation
      const uint32_t r = md_find(par, bq);                  // its pocket
      bool dup = false;
      for (int i = 0; i < ns; i++) dup |= seen[i] == r;
      if (!dup) seen[ns++] = r;
    }
    for (int i = 0; i < ns; i++) {
      atomicAdd(&ncand[seen[i]], 1u);
      if (i) md_unite(par2, seen[0], seen[i]);
    }
  }
}
// pockets (roots of par with a candidate), tie pockets (two or more candidates); flags the clusters of the latter
__global__ __launch_bounds__(NTHR) void k_md_flag(uint32_t *par, uint32_t *par2, const uint32_t *__restrict__ ncand, uint32_t *flag,
                                                  unsigned long long *counts, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  const uint32_t nc = b < B ? ncand[b] : 0u;   // (only pocket roots were counted into)
  if (nc >= 2) flag[md_find(par2, b)] = 1u;
  const unsigned long long p1 = __ballot(nc >= 1), p2 = __ballot(nc >= 2);
  if ((threadIdx.x & 63) == 0) {
    if (p1) atomicAdd(&counts[0], (unsigned long long)__popcll(p1));
    if (p2) atomicAdd(&counts[1], (unsigned long long)__popcll(p2));
  }
}
// cells per pocket (sizeP, at the pocket's root in par) and per cluster (size2, at the cluster's root in par2)
__global__ __launch_bounds__(NTHR) void k_md_tie_sizes(uint32_t *par, uint32_t *par2, const uint32_t *__restrict__ cnt, uint32_t *sizeP,
                                                       uint32_t *size2, uint32_t B) {
  const uint32_t b = blockIdx.x * NTHR + threadIdx.x;
  if (b >= B) return;
  const uint32_t c = cnt[b];
  if (!c) return;
  atomicAdd(&sizeP[md_find(par, b)], c);
  atomicAdd(&size2[md_find(par2, b)], c);
}
// per cell: can the heap's order decide whether this cell is raised?  It lies in a pocket of a tie-flagged cluster, AND the
// size limit can fall either way: a pocket larger than the limit is never raised (any run holding it is larger still), a
// cluster that fits the limit as a whole is always raised (every run is part of it) -- whatever the order.  (mask optional;
// counts those cells and the pocket cells)
template <class T>
__global__ __launch_bounds__(NTHR) void k_md_tie_mask(const T *__restrict__ z, const uint32_t *__restrict__ lab,
                                                      const uint32_t *__restrict__ acc, uint32_t *par, uint32_t *par2,
                                                      const uint32_t *__restrict__ flag, const uint32_t *__restrict__ sizeP,
                                                      const uint32_t *__restrict__ size2, uint64_t max_dep,
                                                      uint8_t *mask, unsigned long long *counts, uint64_t n, uint32_t B) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  unsigned long long nt = 0, np = 0;
  for (uint64_t c0 = (uint64_t)blockIdx.x * NTHR; c0 < n; c0 += stride) {
    const uint64_t c = c0 + threadIdx.x;
    bool pocket = false, tie = false;
    if (c < n) {
      const uint32_t b = lab[c];
      pocket = b != B && acc[b] > Key32<T>::to(z[c]);
      if (pocket) {
        const uint32_t c2 = md_find(par2, b);
        tie = flag[c2] != 0 && (uint64_t)size2[c2] > max_dep && (uint64_t)sizeP[md_find(par, b)] <= max_dep;
      }
      if (mask) mask[c] = tie ? 1 : 0;
    }
    nt += (unsigned long long)__popcll(__ballot(tie));
    np += (unsigned long long)__popcll(__ballot(pocket));
  }
  if ((threadIdx.x & 63) == 0) {
    if (nt) atomicAdd(&counts[2], nt);
    if (np) atomicAdd(&counts[3], np);
  }
}
static thread_local rdgpu_max_dep_stats g_md_stats;
static thread_local unsigned long long *g_md_pinned = nullptr;   // the tie detector's counts of the last call, once its stream got there
static thread_local bool g_md_pending = false;
static thread_local hipStream_t g_md_stream = nullptr;

template <class T, int TOPO>
static void fill_max_dep_device_t(T *d_z, int w, int h, uint64_t max_dep, hipStream_t s, uint8_t *d_tie_mask = nullptr) {
  g_md_stats = rdgpu_max_dep_stats{0, 0, 0, 0};
  g_md_pending = false;
  if (d_tie_mask) RD_HIP(hipMemsetAsync(d_tie_mask, 0, (size_t)w * h, s));
  FillBuffers fb;
  BufAlloc ws_alloc{false, nullptr};
  fill_local_phase<T, TOPO>(d_z, w, h, 0, 0, ws_alloc, fb, s);
  if (fb.trivial) return;
  const uint64_t n = (uint64_t)w * h;
  const uint32_t B = fb.B;
  Workspace &ws = Workspace::get();
  uint32_t *par = ws.buf<uint32_t>("maxdep.par", B), *cnt = ws.buf<uint32_t>("maxdep.cnt", B);
  uint32_t *spawn = ws.buf<uint32_t>("maxdep.spawn", B), *size = ws.buf<uint32_t>("maxdep.size", B);
  const uint32_t bgrid = cdiv(B, NTHR), sgrid = (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u);
  RD_LAUNCH("maxdep.init", k_md_init, dim3(bgrid), dim3(NTHR), 0, s, par, cnt, spawn, size, B);
  RD_LAUNCH("maxdep.pockets", (k_md_pockets<T, TOPO>), dim3(256u * 16u), dim3(NTHR), 0, s, (const T *)d_z, (const uint32_t *)fb.lab,
            (const uint32_t *)fb.acc, par, cnt, w, h, B);
  {   // the tie detector: before the runs join pockets in `par`
    uint32_t *par2 = ws.buf<uint32_t>("maxdep.par2", B), *ncand = ws.buf<uint32_t>("maxdep.ncand", B), *flag = ws.buf<uint32_t>("maxdep.flag", B);
    unsigned long long *counts = ws.buf<unsigned long long>("maxdep.counts", 4);
    RD_HIP(hipMemcpyAsync(par2, par, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    RD_HIP(hipMemsetAsync(ncand, 0, (size_t)B * sizeof(uint32_t), s));
    RD_HIP(hipMemsetAsync(flag, 0, (size_t)B * sizeof(uint32_t), s));
    RD_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), s));
    RD_LAUNCH("maxdep.ties", (k_md_ties<T, TOPO>), dim3(sgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint32_t *)fb.lab,
              (const uint32_t *)fb.acc, par, par2, ncand, w, h, B);
    RD_LAUNCH("maxdep.flag", k_md_flag, dim3(bgrid), dim3(NTHR), 0, s, par, par2, (const uint32_t *)ncand, flag, counts, B);
    uint32_t *sizeP = ws.buf<uint32_t>("maxdep.sizeP", B), *size2 = ws.buf<uint32_t>("maxdep.size2", B);
    RD_HIP(hipMemsetAsync(sizeP, 0, (size_t)B * sizeof(uint32_t), s));
    RD_HIP(hipMemsetAsync(size2, 0, (size_t)B * sizeof(uint32_t), s));
    RD_LAUNCH("maxdep.tie_sizes", k_md_tie_sizes, dim3(bgrid), dim3(NTHR), 0, s, par, par2, (const uint32_t *)cnt, sizeP, size2, B);
    RD_LAUNCH("maxdep.tie_mask", (k_md_tie_mask<T>), dim3(sgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint32_t *)fb.lab,
              (const uint32_t *)fb.acc, par, par2, (const uint32_t *)flag, (const uint32_t *)sizeP, (const uint32_t *)size2, max_dep,
              d_tie_mask, counts, n, B);
    // the counts come back LAZILY (the stream-ordered `_dev_` entry must not block the host): copied into a pinned
    // word block behind the kernels, read by rdgpu_fill_max_dep_get_stats after it has waited for this stream
    if (!g_md_pinned) RD_HIP(hipHostMalloc((void **)&g_md_pinned, 4 * sizeof(unsigned long long)));
    RD_HIP(hipMemcpyAsync(g_md_pinned, counts, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    g_md_pending = true;
    g_md_stream = s;
  }
  RD_LAUNCH("maxdep.spawn", (k_md_spawn<T, TOPO, 0>), dim3(sgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint32_t *)fb.lab,
            (const uint32_t *)fb.acc, par, spawn, w, h, B);
  RD_LAUNCH("maxdep.runs", (k_md_spawn<T, TOPO, 1>), dim3(sgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint32_t *)fb.lab,
            (const uint32_t *)fb.acc, par, spawn, w, h, B);
  RD_LAUNCH("maxdep.sizes", k_md_sizes, dim3(bgrid), dim3(NTHR), 0, s, par, (const uint32_t *)cnt, size, B);
  RD_LAUNCH("maxdep.apply", (k_md_apply<T>), dim3(sgrid), dim3(NTHR), 0, s, d_z, (const uint32_t *)fb.lab, (const uint32_t *)fb.acc,
            par, (const uint32_t *)size, n, B, max_dep);
}

template <class T>
static void fill_max_dep_device(T *d_z, int w, int h, int topology, uint64_t max_dep, hipStream_t s, uint8_t *d_tie_mask = nullptr) {
  check_fill_args(d_z, w, h, topology);
  if (topology == 8) fill_max_dep_device_t<T, 8>(d_z, w, h, max_dep, s, d_tie_mask);
  else fill_max_dep_device_t<T, 4>(d_z, w, h, max_dep, s, d_tie_mask);
}

template <class T>
static void fill_max_dep_host(T *dem, int w, int h, int topology, uint64_t max_dep) {
  check_fill_args(dem, w, h, topology);
  const size_t bytes = (size_t)w * h * sizeof(T);
  T *d = Workspace::get().buf<T>("host.dem", (size_t)w * h);
  RD_HIP(hipMemcpy(d, dem, bytes, hipMemcpyHostToDevice));
  fill_max_dep_device<T>(d, w, h, topology, max_dep, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, bytes, hipMemcpyDeviceToHost));
}

// ==========================================================================================================
// The compact-label fill (r03): 16-bit labels, no label pass over the raster.
//
// The classic path above reads the raster four times (descent, labels, pair pass, finalize: 4.8 x the algorithmic bytes
// at S3) and moves a 32-bit label per cell through three of them.  Here
//   k_descent16      descent pointers and in-tile path compression as k_descent; then every ROOT of the tile -- a pit, a
//                    cell draining off the raster, a cell whose descent neighbour lies outside the tile -- becomes a
//                    NODE: local slot 0.., global id tile_base + slot, G[node] = basin id | PEND | first cell outside
//                    the tile | OUTP.  A cell's label is the SLOT of its root: 2 bytes per cell (lab16).
//   k_resolve_nodes  node -> basin (curN): a pending node follows lab16 / G from tile to tile.  A table pass over ~2 % of
//                    the cell count replaces k_tile_label's raster pass (10.7 GB at S3).
//   k_scan<L16>      the one pair pass, reading z + lab16 (6 B / cell) and gathering the components from curN.
//   rounds           unchanged (pair list).
//   k_finalize16     z <- max(z, level of the cell's node): z + lab16 and the tile's node levels from LDS.
// (Tried first and dropped: the pairs in the descent kernel itself, on slots, with the pairs across tile edges from edge
// strips -- correct, but slot boundaries are several times as many as basin boundaries, since every exit of a tile is a
// slot of its own until it is resolved: 22 ms for the kernel, 80 M records; profiles/r03h_fill_fused_ab.json.)
// Anything the scheme cannot hold (more nodes or records than the buffers) raises a flag and the classic path runs.
// Row-block shards, pit_mask, max_dep and the watershed code keep the classic path (they consume 32-bit labels).
// ==========================================================================================================
// Node and pit ids are handed out per tile with ONE returning atomic -- and same-address device atomics serialise at
// ~12 ns each: 390 625 tiles on one counter are 4.5 ms of the kernel whatever else it does (r04).  So there are up to 32
// STRIPES (tile t uses stripe t & smask), each with its own counter word on its own 128-byte line, its own region of the
// node table (rcap nodes) and its own run of pit numbers; k_stripe_offsets turns the pit numbers into dense basin ids
// afterwards (pitoff[]), which k_resolve_nodes adds when it writes a node's component.
constexpr int FSTRIPES = 32, FSTRIDE = 16;   // counter words are FSTRIDE * 8 bytes apart
struct FusedBuf {
  uint16_t *lab16;                 // [cells] slot of the cell's root within its descent tile
  uint32_t *G;                     // [gcap] node table: stripe s owns [s * rcap, (s + 1) * rcap)
  uint32_t *tile_base, *tile_count;   // [descent tiles]
  unsigned long long *counters;    // [stripe * FSTRIDE] (nodes << 32) | pits of the stripe
  uint32_t gcap, rcap, smask;
  uint32_t *overflow;
  // the first and last column of every descent tile as compact records (key, slot), [tile][side][row]: the pair pass takes
  // its tiles' ring columns from these -- read from the raster a ring column costs a 64-byte sector per row and array
  // for one cell (12.5 GB per pass at S3 once the neighbouring tile's rows are no longer in L2; r04a counters)
  uint32_t *edgeK = nullptr;
  uint16_t *edgeS = nullptr;
};

// after the descent: dense basin ids.  out[0] = basins, out[1] = nodes in use, out[2] = the fullest stripe's node count
__global__ __launch_bounds__(64) void k_stripe_offsets(const unsigned long long *__restrict__ counters, uint32_t nstripes,
                                                       uint32_t *pitoff, uint32_t *out) {
  const uint32_t s = threadIdx.x;
  const unsigned long long c = s < nstripes ? counters[s * FSTRIDE] : 0ull;
  const uint32_t pits = (uint32_t)c, nodes = (uint32_t)(c >> 32);
  uint32_t incl = pits, tot = nodes, mx = nodes;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if ((int)s >= o) incl += v;
    tot += __shfl_xor(tot, o, 64);
    mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64));
  }
  if (s < nstripes) pitoff[s] = incl - pits;
  if (s == 63) { out[0] = incl; out[1] = tot; out[2] = mx; }
}

// OUTLETS: cells flagged in `outlet` drain like the raster's border cells (interior outlets: the restricted fills of
// pfdirs.hip); the instantiation of the plain fill does not look at the pointer.
// There, all outlets of a tile share ONE node (slot 0: a tile of walls has thousands of them), node 0 of the table is a
// global "outside" node, and `skip` (optional, per descent tile) names the tiles that hold nothing but outlets, ring
// included: they get no work -- label 0 everywhere (written once: skip == 1; 2: written before), node 0.
// CUT: the raster is a row-block shard -- the cells of an open first / last row are frozen terminals (own roots, numbered
// like pits, never draining), not border cells.
template <class T, int TOPO, bool VEC, bool OUTLETS = false, bool CUT = false>
__global__ __launch_bounds__(NTHR) void k_descent16(const T *__restrict__ z, FusedBuf fo, int w, int h,
                                                    uint32_t tilesX, uint32_t ntiles, const uint8_t *__restrict__ outlet = nullptr,
                                                    const uint8_t *__restrict__ skip = nullptr,
                                                    const uint32_t *__restrict__ tlist = nullptr, int open_top = 0,
                                                    int open_bottom = 0) {
  // tlist (optional, with OUTLETS): the tiles to work on, one block each (ntiles = the raster's tiles all the same)
  __shared__ uint32_t sk[DLH * DLW];
  // rows of LPD = 66 entries: with 64 two-byte entries every row starts on the same LDS bank and the jumps' gathers --
  // neighbouring columns of different rows -- collide (29 % of the kernel's LDS cycles, r03e)
  constexpr int LPD = DW + 2;
  __shared__ uint16_t lp[DH * LPD];
  __shared__ uint32_t wtot[NTHR / 64];
  __shared__ uint32_t pbase, rbase, fits_s;
  const uint32_t t = (OUTLETS && tlist) ? tlist[blockIdx.x] : xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * DW, y0 = (int)(t / tilesX) * DH;
  if (OUTLETS && skip && skip[t]) {
    if (threadIdx.x == 0) { fo.tile_base[t] = 0; fo.tile_count[t] = 1; }
    if (skip[t] == 1) {
      const int lx_ = threadIdx.x & (DW - 1), gx_ = x0 + lx_;
      for (int ly_ = threadIdx.x >> 6; ly_ < DH; ly_ += NTHR / 64)
        if (gx_ < w && y0 + ly_ < h) fo.lab16[(size_t)(y0 + ly_) * w + gx_] = 0;
    }
    return;
  }
  {
    constexpr int NQ = DLH * (DW / 4), QPT = (NQ + NTHR - 1) / NTHR;
    Quad<T> zq[QPT];
    if (y0 >= 1 && y0 + DH < h && x0 + DW <= w) {   // (block-uniform) every row and quad of the window lies in the raster: no tests
#pragma unroll
      for (int r = 0; r < QPT; r++) {
        const int i = threadIdx.x + r * NTHR;
        const int ly = i / (DW / 4), q = i - ly * (DW / 4);
        if (i < NQ) zq[r] = load_quad<T, VEC>(z + (size_t)(y0 - 1 + ly) * w, x0 + 4 * q, w, T());
      }
#pragma unroll
      for (int r = 0; r < QPT; r++) {
        const int i = threadIdx.x + r * NTHR;
        if (i >= NQ) continue;
        const int ly = i / (DW / 4), q = i - ly * (DW / 4);
        const int o = ly * DLW + 1 + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; e++) sk[o + e] = Key32<T>::to(zq[r].v[e]);
      }
    } else {
      bool okq[QPT];
#pragma unroll
      for (int r = 0; r < QPT; r++) {
        const int i = threadIdx.x + r * NTHR;
        const int ly = i / (DW / 4), q = i - ly * (DW / 4);
        const int gx = x0 + 4 * q, gy = y0 - 1 + ly;
        okq[r] = i < NQ && gy >= 0 && gy < h && gx < w;
        if (okq[r]) zq[r] = load_quad<T, VEC>(z + (size_t)gy * w, gx, w, T());
      }
#pragma unroll
      for (int r = 0; r < QPT; r++) {
        const int i = threadIdx.x + r * NTHR;
        if (i >= NQ) continue;
        const int ly = i / (DW / 4), q = i - ly * (DW / 4);
        const int o = ly * DLW + 1 + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; e++) sk[o + e] = (okq[r] && x0 + 4 * q + e < w) ? Key32<T>::to(zq[r].v[e]) : 0xFFFFFFFFu;
      }
    }
    for (int i = threadIdx.x; i < 2 * DLH; i += NTHR) {   // halo columns
      const int ly = i >> 1, lxh = (i & 1) ? DLW - 1 : 0;
      const int gx = x0 - 1 + lxh, gy = y0 - 1 + ly;
      uint32_t kk = 0xFFFFFFFFu;
      if (gx >= 0 && gx < w && gy >= 0 && gy < h) kk = Key32<T>::to(z[(size_t)gy * w + gx]);
      sk[ly * DLW + lxh] = kk;
    }
  }
  __syncthreads();
  // (the wavefront's index as a SCALAR: everything derived from the row -- bounds tests, LDS row offsets -- then runs on the
  // scalar unit; this kernel is bound by VALU issue, 4 cycles per wave instruction)
  //
  // r04d: the kernel's VALU work per row of 64 cells went 143 -> ~70 instructions by three changes.
  //  * The 8-neighbour minimum WITH its position from row triples: min3 + "first of (a, b, c) equal to it" once per row,
  //    used by the cell above and the cell below; the winner carries one packed word (pointer delta | exit code | which
  //    column / row it lies in), so nothing is decoded afterwards.
  //  * Roots point to THEMSELVES in lp and pointers are byte offsets: a jump is lp[lp[p]] with no compare or select, a
  //    cell is finished when the two reads agree, and the wavefronts iterate on their own rows without barriers (every
  //    value a racing read can see is an ancestor on the cell's path).  The exit codes stay in registers.
  //  * Slots go to a 16-bit array laid over the keys (dead by then) at the SAME offsets as lp: label = slot16[p].
  const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lx = threadIdx.x & (DW - 1), ly0 = wave_s * (DH / 4);
  const int gx = x0 + lx;
  constexpr int RPT = DH / 4;                      // rows per thread
  static_assert(RPT == 16, "four code words of four bytes, 16-bit row masks");
  uint16_t *const slot16 = reinterpret_cast<uint16_t *>(sk);   // valid once the keys are dead (after the edge keys are out)
  const uint32_t selfb0 = (uint32_t)((ly0 * LPD + lx) * 2);
  uint32_t p[RPT];                                 // the cell's pointer: byte offset into lp
  uint32_t codes[RPT / 4] = {0, 0, 0, 0};          // one byte per row: 0 pit, 1..8 exit towards that neighbour, 9 drains off the raster
  uint32_t rootmask = 0, pitmask = 0, outmask = 0;
  {
    // packed word of a candidate: [15:0] byte delta + 134, [19:16] exit code, [26:24] column 0 / 1 / 2, [27] row above, [29] row below
    constexpr uint32_t WT0 = 0u | (1u << 16) | (1u << 24) | (1u << 27), WT1 = 2u | (2u << 16) | (1u << 25) | (1u << 27),
                       WT2 = 4u | (3u << 16) | (1u << 26) | (1u << 27);
    constexpr uint32_t WM0 = (uint32_t)(2 * LPD) | (4u << 16) | (1u << 24), WM2 = (uint32_t)(2 * LPD + 4) | (5u << 16) | (1u << 26);
    constexpr uint32_t BADD = (uint32_t)(4 * LPD) + (5u << 16) + ((1u << 29) - (1u << 27));
    const bool colin = gx < w, colborder = (gx == 0) | (gx == w - 1);
    const uint32_t cmcol = lx == 0 ? (1u << 24) : lx == DW - 1 ? (1u << 26) : 0u;
    uint32_t ta, tb, tc, tm, tW, ma, mb, mc, mm3, mW;   // the triples of the row above and of the cell's row
    auto triple = [&](int r, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &m, uint32_t &W) {
      a = sk[r * DLW + lx];
      b = sk[r * DLW + lx + 1];
      c = sk[r * DLW + lx + 2];
      if (TOPO == 8) {
        m = min(min(a, b), c);
        W = a == m ? WT0 : b == m ? WT1 : WT2;
      } else {
        m = b;
        W = WT1;
      }
    };
    triple(ly0, ta, tb, tc, tm, tW);
    triple(ly0 + 1, ma, mb, mc, mm3, mW);
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int ly = ly0 + j, gy = y0 + ly;   // (scalar)
      uint32_t ba, bb, bc, bm, bW;
      triple(ly + 2, ba, bb, bc, bm, bW);
      const uint32_t kc = mb;
      const bool midfirst = ma <= mc;
      const uint32_t mm = midfirst ? ma : mc;
      const uint32_t bk = min(min(tm, mm), bm);
      const bool isT = tm == bk, isM = mm == bk;
      const uint32_t wM = midfirst ? WM0 : WM2, wB = bW + BADD, wMB = isM ? wM : wB;
      const uint32_t W = isT ? tW : wMB;
      const bool drains = (bk < kc) | ((bk == kc) & (isT | (isM & midfirst)));
      uint32_t cm = cmcol;
      if (j == 0 && ly0 == 0) cm |= 1u << 27;                   // (only a wavefront's first / last row can be the tile's)
      if (j == RPT - 1 && ly0 == DH - RPT) cm |= 1u << 29;
      const bool outside = (W & cm) != 0u;
      const uint32_t self = selfb0 + (uint32_t)(j * LPD * 2);
      const bool incell = colin & (gy < h);
      bool border = colborder | (gy == 0) | (gy == h - 1);
      bool stays = !drains;
      if (CUT) {
        const bool cutrow = ((gy == 0) & (open_top != 0)) | ((gy == h - 1) & (open_bottom != 0));
        border = colborder | ((gy == 0) & !open_top) | ((gy == h - 1) & !open_bottom);
        stays |= cutrow;
      }
      if (OUTLETS) border |= incell && outlet[(size_t)gy * w + gx] != 0;
      const bool selfp = !incell | border | stays | outside;
      const uint32_t rel = selfp ? (uint32_t)(2 * LPD + 2) : (W & 0xFFFFu);   // (the delta's bias: the cell itself)
      p[j] = rel + (self - (uint32_t)(2 * LPD + 2));
      *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lp) + self) = (uint16_t)p[j];
      const uint32_t cexit = (W >> 16) & 15u, cin = stays ? 0u : cexit;
      const uint32_t code = border ? 9u : cin;
      codes[j >> 2] |= code << (8 * (j & 3));
      bool root = incell & selfp;
      if (OUTLETS) {   // the tile's outlets share slot 0
        const bool outroot = root & border;
        outmask |= (outroot ? 1u : 0u) << j;
        root &= !outroot;
      }
      rootmask |= (root ? 1u : 0u) << j;
      pitmask |= ((root & !border & stays) ? 1u : 0u) << j;
      ta = ma; tb = mb; tc = mc; tm = mm3; tW = mW;
      ma = ba; mb = bb; mc = bc; mm3 = bm; mW = bW;
    }
    (void)ta; (void)tb; (void)tc;
  }
  // (the masks and codes are formed HERE, in vector registers: left to the compiler they sink below the jump loop, which
  // keeps sixteen rows of lane masks alive in scalar registers and spills them lane by lane)
  asm volatile("" : "+v"(rootmask), "+v"(pitmask), "+v"(outmask), "+v"(codes[0]), "+v"(codes[1]), "+v"(codes[2]), "+v"(codes[3]));
  // the tile's first and last column: the KEYS of the compact records for the pair pass (before the keys' array is reused)
  if (fo.edgeK && threadIdx.x < 2 * DH) {
    const int side = threadIdx.x >> 6, lye = threadIdx.x & (DH - 1), lxe = side ? DW - 1 : 0;
    fo.edgeK[((size_t)t * 2 + side) * DH + lye] = sk[(lye + 1) * DLW + lxe + 1];
  }
  __syncthreads();
  {   // pointer jumping inside the tile: every wavefront on its own rows until they all point to roots; no barriers
    const char *const lpb = reinterpret_cast<const char *>(lp);
    uint32_t gact = (1u << (RPT / 4)) - 1u;   // (scalar) groups of four rows with a pointer that may still move
    for (int it = 0; gact != 0u && it < DW * DH; it++) {
      asm volatile("" ::: "memory");   // (other wavefronts write lp meanwhile: nothing read from it is kept across passes)
#pragma unroll
      for (int g = 0; g < RPT / 4; g++) {
        if (!(gact >> g & 1u)) continue;
        uint32_t qv[4], rv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { qv[e] = *reinterpret_cast<const uint16_t *>(lpb + p[4 * g + e]); asm("" : "+v"(qv[e])); }
#pragma unroll
        for (int e = 0; e < 4; e++) { rv[e] = *reinterpret_cast<const uint16_t *>(lpb + qv[e]); asm("" : "+v"(rv[e])); }   // (opaque 32-bit values: else the compare is narrowed to 16 bits and every value is masked again for its use as an address)
        unsigned long long moving = 0;   // (ballots of the plain compares: lane masks straight from v_cmp, OR-ed on the scalar unit)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int j = 4 * g + e;
          moving |= __builtin_amdgcn_ballot_w64(rv[e] != qv[e]);   // (equal: qv is a root, the cell is finished)
          p[j] = rv[e];
          *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lp) + selfb0 + (uint32_t)(j * LPD * 2)) = (uint16_t)rv[e];
        }
        if (moving == 0ull) gact &= ~(1u << g);
      }
    }
  }
  // ---- nodes: every root of the tile gets a local slot; pits a dense basin id --------------------------------------
  const uint32_t mine = ((uint32_t)__popc(rootmask) << 16) | (uint32_t)__popc(pitmask);   // (<= 1024 each per wavefront)
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
  __syncthreads();   // (also: every wavefront's pointers are final, the keys are dead)
  if (threadIdx.x == 0) {
    const uint32_t nr = (wtot[0] >> 16) + (wtot[1] >> 16) + (wtot[2] >> 16) + (wtot[3] >> 16) + (OUTLETS ? 1u : 0u);
    const uint32_t np = (wtot[0] & 0xFFFFu) + (wtot[1] & 0xFFFFu) + (wtot[2] & 0xFFFFu) + (wtot[3] & 0xFFFFu);
    const uint32_t st = t & fo.smask;   // the tile's stripe: its own counter, node region and run of pit numbers
    const unsigned long long old = atomicAdd(&fo.counters[st * FSTRIDE], ((unsigned long long)nr << 32) | np);
    const uint32_t rl = (uint32_t)(old >> 32);
    pbase = (uint32_t)old;
    rbase = st * fo.rcap + rl;
    fo.tile_base[t] = rbase;
    fo.tile_count[t] = nr;
    fits_s = (unsigned long long)rl + nr <= fo.rcap ? 1u : 0u;
    if (!fits_s) *fo.overflow = 1;
  }
  {   // (the roots' slots do not depend on the counter: written beside thread 0's atomic)
    uint32_t pre = incl - mine;
    for (int k = 0; k < wave_s; k++) pre += wtot[k];
    uint32_t slot = (pre >> 16) + (OUTLETS ? 1u : 0u);
    if (OUTLETS)
      for (uint32_t m = outmask; m; m &= m - 1)
        *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(slot16) + selfb0 + (uint32_t)((__ffs((int)m) - 1) * LPD * 2)) = 0;   // slot 0
    for (uint32_t m = rootmask; m; m &= m - 1) {
      const int j = __ffs((int)m) - 1;
      *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(slot16) + selfb0 + (uint32_t)(j * LPD * 2)) = (uint16_t)slot;
      slot++;
    }
  }
  __syncthreads();
  const uint32_t nodes0 = rbase;
  if (fits_s != 0) {
    uint32_t pre = incl - mine;
    for (int k = 0; k < wave_s; k++) pre += wtot[k];
    uint32_t slot = (pre >> 16) + (OUTLETS ? 1u : 0u), pit = pbase + (pre & 0xFFFFu);
    if (OUTLETS && threadIdx.x == 0) fo.G[nodes0] = OUTP;
    for (uint32_t m = rootmask; m; m &= m - 1) {
      const int j = __ffs((int)m) - 1;
      const int ly = ly0 + j;
      const uint32_t cw = j < 8 ? (j < 4 ? codes[0] : codes[1]) : (j < 12 ? codes[2] : codes[3]);
      const int code = (int)(cw >> (8 * (j & 3)) & 15u);
      const int n = code <= 4 ? code - 1 : code;   // 3x3 position of the root's descent neighbour (codes 1..8)
      const int nr = n >= 6 ? 2 : n >= 3 ? 1 : 0, nc = n - 3 * nr;
      const uint32_t pend = LAB_PEND | ((uint32_t)(y0 + ly + nr - 1) * (uint32_t)w + (uint32_t)(x0 + lx + nc - 1));
      const uint32_t word = code == 0 ? pit++ : code == 9 ? OUTP : pend;
      fo.G[nodes0 + slot] = word;
      slot++;
    }
  }
  // a cell's label: the slot of its root (its own when it is one)
  {
    const char *const sl = reinterpret_cast<const char *>(slot16);
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int gy = y0 + ly0 + j;
      const uint16_t rv = *reinterpret_cast<const uint16_t *>(sl + p[j]);
      if ((gx < w) & (gy < h)) fo.lab16[(size_t)gy * w + gx] = rv;   // (< 4096: not every cell of a tile can be a root)
    }
    // the tile's first and last column, the SLOTS of the compact records: two wavefronts, one row per lane, coalesced.
    // A cell past the raster's end gets slot 0: the pair pass does read such a record (a ring cell outside the raster is
    // loaded from a clamped position and discarded afterwards), and follows its slot into the node table.
    if (fo.edgeS && threadIdx.x < 2 * DH) {
      const int side = threadIdx.x >> 6, lye = threadIdx.x & (DH - 1), lxe = side ? DW - 1 : 0;
      const uint16_t v = lp[lye * LPD + lxe];
      const bool in = x0 + lxe < w && y0 + lye < h;
      fo.edgeS[((size_t)t * 2 + side) * DH + lye] = in ? *reinterpret_cast<const uint16_t *>(sl + v) : (uint16_t)0;
    }
  }
}

// node -> its basin as a component id: curN[n] = basin, or B | CLOSED for the outside.  A pending node names the first
// cell outside its tile on the path: that cell's node is looked up through lab16 and the tile bases, and so on from
// tile to tile.  A pit's word is its number WITHIN ITS STRIPE: the basin id is that plus the stripe's offset, the stripe
// being that of the node the word was found in.  A chain of several hops is shortened for the others: the node's word
// becomes the last cell of the chain (any value ever stored in G is a valid continuation: concurrent chasers and stale
// reads are harmless).  grid: (nodes of the fullest stripe, stripes).
__global__ __launch_bounds__(NTHR) void k_resolve_nodes(uint32_t *G, const unsigned long long *__restrict__ counters,
                                                        const uint32_t *__restrict__ pitoff, uint32_t rcap,
                                                        const uint16_t *__restrict__ lab16,
                                                        const uint32_t *__restrict__ tile_base, int w, uint32_t tilesX,
                                                        uint32_t B, uint32_t *curN, uint32_t *flag,
                                                        const uint32_t *__restrict__ tid = nullptr) {
  const uint32_t st = blockIdx.y, i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= (uint32_t)(counters[st * FSTRIDE] >> 32)) return;
  const uint32_t n = st * rcap + i;
  uint32_t v = G[n], fn = n, last = 0;
  int hops = 0;
  while ((v & LAB_PEND) && v != OUTP) {
    const uint32_t cell = v & ~LAB_PEND;
    const uint32_t cx = cell % (uint32_t)w, cy = cell / (uint32_t)w;
    fn = tile_base[(cy / DH) * tilesX + cx / DW] + lab16[cell];
    last = cell;
    v = __hip_atomic_load(&G[fn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++hops > (1 << 22)) { *flag = 1; break; }   // (cannot happen: descent paths are loop free)
  }
  if (hops > 1) __hip_atomic_store(&G[n], LAB_PEND | last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == OUTP || (v & LAB_PEND)) { curN[n] = B | CLOSED; return; }
  const uint32_t b = v + pitoff[fn / rcap];
  curN[n] = (tid && tid[b] != NO_TID) ? (b | CLOSED) : b;   // a shard's cut-row terminals are closed components
}

// A shard's cut-row cells are roots of their tiles with a pit number of their own: tid[basin] = terminal id (top row: x,
// bottom row: w + x), straight from the node table (a pit's word needs no chase), before the nodes are resolved.
__global__ __launch_bounds__(NTHR) void k_mark_terminals16(const uint16_t *__restrict__ lab16, const uint32_t *__restrict__ tile_base,
                                                           const uint32_t *__restrict__ G, const uint32_t *__restrict__ pitoff,
                                                           uint32_t rcap, uint32_t *tid, int w, int h, uint32_t dtx, int open_top,
                                                           int open_bottom) {
  const int x = blockIdx.x * NTHR + threadIdx.x;
  if (x <= 0 || x >= w - 1) return;
  if (open_top) {
    const uint32_t node = tile_base[(uint32_t)(x / DW)] + lab16[x];
    tid[G[node] + pitoff[node / rcap]] = (uint32_t)x;
  }
  if (open_bottom) {
    const uint32_t node = tile_base[(uint32_t)((h - 1) / DH) * dtx + (uint32_t)(x / DW)] + lab16[(size_t)(h - 1) * w + x];
    tid[G[node] + pitoff[node / rcap]] = (uint32_t)(w + x);
  }
}

// per node: the terminal id of its watershed after the local rounds (NO_TID: it drains to the true border inside the shard)
__global__ __launch_bounds__(NTHR) void k_node_watersheds(const uint32_t *__restrict__ curN, const uint32_t *__restrict__ cur,
                                                          const uint32_t *__restrict__ tid,
                                                          const unsigned long long *__restrict__ counters, uint32_t rcap, uint32_t B,
                                                          uint32_t *nodeW) {
  const uint32_t st = blockIdx.y, i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= (uint32_t)(counters[st * FSTRIDE] >> 32)) return;
  const uint32_t n = st * rcap + i;
  const uint32_t c = cur[curN[n] & ~CLOSED] & ~CLOSED;
  nodeW[n] = c == B ? NO_TID : tid[c];
}

// k_shard_edges on the compact labels: a cell's watershed terminal and locally filled level come from its NODE
// (tile base + 16-bit slot -> nodeW / lvl) instead of label -> component -> terminal.
template <class T, int TOPO>
__global__ __launch_bounds__(NTHR) void k_shard_edges16(const T *__restrict__ z, const uint16_t *__restrict__ lab16,
                                                        const uint32_t *__restrict__ tile_base, uint32_t dtx,
                                                        const uint32_t *__restrict__ nodeW, const uint32_t *__restrict__ lvl,
                                                        int w, int h, unsigned long long *hkeys, uint32_t *hvals, uint32_t hmask,
                                                        uint32_t *overflow, uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint32_t sl[ELH * ELW];
  __shared__ uint32_t sw[ELH * ELW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * EW, y0 = (int)(t / tilesX) * EH;
  constexpr int IPT = (ELH * ELW + NTHR - 1) / NTHR;
  uint32_t nv[IPT], kv[IPT];
  bool ok[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {   // batch 1: slot, node base of the cell's descent tile, elevation key
    const int i = threadIdx.x + r * NTHR;
    const int ly = i / ELW, lx = i - ly * ELW;
    const int gx = x0 - 1 + lx, gy = y0 + ly;
    ok[r] = i < ELH * ELW && gx >= 0 && gx < w && gy < h;
    const int cx = ok[r] ? gx : 0, cy = ok[r] ? gy : 0;
    const size_t g = (size_t)cy * w + cx;
    nv[r] = tile_base[(uint32_t)(cy / DH) * dtx + (uint32_t)(cx / DW)] + lab16[g];
    kv[r] = Key32<T>::to(z[g]);
  }
  uint32_t wv[IPT], av[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {   // batch 2: the node's watershed terminal and level
    wv[r] = nodeW[nv[r]];
    av[r] = lvl[nv[r]];
  }
#pragma unroll
  for (int r = 0; r < IPT; r++) {
    const int i = threadIdx.x + r * NTHR;
    if (i < ELH * ELW) {
      sl[i] = ok[r] ? wv[r] : E_INVALID;
      sw[i] = av[r] > kv[r] ? av[r] : kv[r];   // (the outside's level is 0)
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & (EW - 1), ly0 = threadIdx.x >> 6;
#pragma unroll 2
  for (int j = 0; j < EH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int o = ly * ELW + lx + 1;
    const uint32_t la = sl[o];
    if (la == E_INVALID) continue;
    const uint32_t wa = sw[o];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (TOPO == 4 && (k == 1 || k == 3)) continue;
      const int q = o + (k == 0 ? 1 : k == 1 ? ELW - 1 : k == 2 ? ELW : ELW + 1);
      const uint32_t lb = sl[q];
      if (lb == la || lb == E_INVALID) continue;
      const uint32_t wb = sw[q];
      const uint32_t pass = wa > wb ? wa : wb;
      const uint32_t lo = la < lb ? la : lb, hi = la < lb ? lb : la;
      const unsigned long long key = (((unsigned long long)lo << 32) | hi) + 1ull;
      uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & hmask;
      for (uint32_t probe = 0;; probe++) {
        if (probe > hmask) { *overflow = 1; break; }
        unsigned long long cur_k = __hip_atomic_load(&hkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur_k == 0) cur_k = atomicCAS(&hkeys[slot], 0ull, key);
        if (cur_k == 0 || cur_k == key) {
          if (__hip_atomic_load(&hvals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pass) atomicMin(&hvals[slot], pass);
          break;
        }
        slot = (slot + 1) & hmask;
      }
    }
  }
}

// level of every node: the final level of its basin (0 for the outside: never above a key that matters)
__global__ __launch_bounds__(NTHR) void k_node_levels(const uint32_t *__restrict__ curN, const uint32_t *__restrict__ acc,
                                                      const unsigned long long *__restrict__ counters, uint32_t rcap,
                                                      uint32_t *lvl) {
  const uint32_t st = blockIdx.y, i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= (uint32_t)(counters[st * FSTRIDE] >> 32)) return;
  const uint32_t n = st * rcap + i;
  lvl[n] = acc[curN[n] & ~CLOSED];
}

// z <- max(z, level of the cell's node).  One block per descent tile: its node levels (<= 4096) in LDS, then z and the
// 16-bit labels in quads; a quad is written back only when one of its cells is raised.
template <class T, bool VEC>
__global__ __launch_bounds__(NTHR) void k_finalize16(T *z, const uint16_t *__restrict__ lab16, const uint32_t *__restrict__ lvl,
                                                     const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ tile_count,
                                                     int w, int h, uint32_t tilesX, uint32_t ntiles,
                                                     const uint8_t *__restrict__ skip = nullptr,
                                                     const uint32_t *__restrict__ tlist = nullptr) {
  __shared__ uint32_t sl[DH * DW];
  const uint32_t t = tlist ? tlist[blockIdx.x] : xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  if (skip && skip[t]) return;
  const int x0 = (int)(t % tilesX) * DW, y0 = (int)(t / tilesX) * DH;
  const int qx = threadIdx.x & 15, ry = threadIdx.x >> 4;   // 16 quads per row, rows ry, ry + 16, ry + 32, ry + 48
  const int gx = x0 + 4 * qx;
  // the thread's quads first (in flight while the node levels arrive), then the tile's levels into LDS
  Quad<T> zqs[DH / 16];
  Quad<uint16_t> lqs[DH / 16];
#pragma unroll
  for (int r = 0; r < DH / 16; r++) {
    const int gy = y0 + ry + 16 * r;
    const bool in = gx < w && gy < h;
    const size_t g = in ? (size_t)gy * w + gx : 0;
    zqs[r] = load_quad<T, VEC>(z + (g - (in ? gx : 0)), in ? gx : 0, w, T());
    lqs[r] = load_quad<uint16_t, VEC>(lab16 + (g - (in ? gx : 0)), in ? gx : 0, w, (uint16_t)0);
  }
  const uint32_t base = tile_base[t], cnt = tile_count[t];
  for (uint32_t i = threadIdx.x; i < cnt; i += NTHR) sl[i] = lvl[base + i];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < DH / 16; r++) {
    const int gy = y0 + ry + 16 * r;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    Quad<T> zq = zqs[r];
    const Quad<uint16_t> lq = lqs[r];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (gx + e >= w) continue;
      const uint32_t L = sl[lq.v[e] & (uint16_t)(DW * DH - 1)];
      if (L > Key32<T>::to(zq.v[e])) { zq.v[e] = Key32<T>::from(L); any = true; }
    }
    if (any) {
      if (VEC) {
        struct alignas(4 * sizeof(T)) AQ { T v[4]; };
        AQ a;
#pragma unroll
        for (int e = 0; e < 4; e++) a.v[e] = zq.v[e];
        *reinterpret_cast<AQ *>(z + g) = a;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (gx + e < w) z[g + e] = zq.v[e];
      }
    }
  }
}

// ==========================================================================================================
// The pair pass of the compact-label fill as a PERSISTENT kernel (r04).
//
// k_scan<L16> spends two thirds of its wave cycles parked: per 64 x 32 tile it goes through a chain of dependent global
// round trips (tile bases -> 16-bit labels -> component gathers from the node table) before its LDS phases can start, and
// a block does nothing else meanwhile.  Here a block stays resident and walks its share of the tiles (its XCD's band,
// strided by the number of blocks per XCD); the NEXT tile's rows, labels, ring cells and the node table of its descent
// tile are loaded into registers while the current tile goes through its LDS phases, so the phases of successive tiles
// follow each other without the round trips in between.  The components of the tile's own cells come from the descent
// tile's node table staged in LDS (one coalesced read of its ~80 entries instead of 2 048 gathers); only the 196 cells
// of the ring, which belong to the neighbouring descent tiles, are gathered one by one.  Phases, tables, records and
// proposals are k_scan<EMIT>'s.
// ==========================================================================================================
constexpr int NT_CAP = 1024;   // node-table entries of one descent tile held in LDS (more: gathered from HBM)
static_assert(NT_CAP * 4 <= TW * TH * 2, "the node table lives in the boundary list's storage");
static_assert(TW == DW && DH % TH == 0, "a scan tile lies inside one descent tile");
constexpr uint32_t NO_TILE = 0xFFFFFFFFu;
constexpr int RING = 2 * LW + 2 * TH;   // ring cells of a tile: two rows of LW, two columns of TH

// A pair that found no slot in the tile's table goes straight to the BLOCK's segment (k_pairs16: one segment per
// resident block, filled through a counter in LDS) and proposes on its own.
__device__ __forceinline__ void pair_spill_local(const EdgeOut &eo, unsigned long long *best, uint32_t seg, uint32_t *seg_fill,
                                                 uint32_t lo, uint32_t hi, uint32_t key) {
  const unsigned long long cl = ((unsigned long long)key << 32) | hi, ch = ((unsigned long long)key << 32) | lo;
  if (!(lo & CLOSED) && cl < best[lo]) atomicMin(&best[lo], cl);
  if (!(hi & CLOSED) && ch < best[hi]) atomicMin(&best[hi], ch);
  const uint32_t g = atomicAdd(seg_fill, 1u);
  if (g < eo.seglimit) {
    const size_t i = (size_t)seg * eo.segcap + g;
    eo.a[i] = lo; eo.b[i] = hi; eo.k[i] = key;
  } else {
    *eo.overflow = 1;
  }
}

template <class T, int TOPO, bool VEC>
__global__ __launch_bounds__(NTHR) void k_pairs16(const T *__restrict__ z, const uint16_t *__restrict__ lab16,
                                                  const uint32_t *__restrict__ curN, unsigned long long *best,
                                                  int w, int h, uint32_t B, uint32_t tilesX,
                                                  const uint32_t *__restrict__ tiles_in, uint32_t nwork,
                                                  EdgeOut eo,
                                                  const uint32_t *__restrict__ tile_base,
                                                  const uint32_t *__restrict__ tile_count, uint32_t dtx,
                                                  const uint8_t *__restrict__ skip, int precheck,
                                                  const uint32_t *__restrict__ edgeK, const uint16_t *__restrict__ edgeS) {
  __shared__ __attribute__((aligned(8))) uint32_t sk[LH * LW];
  __shared__ uint32_t sc[LH * LW];
  __shared__ __attribute__((aligned(4))) uint16_t list[TW * TH];   // boundary cells; before that: the node table
  uint32_t *const ntab = reinterpret_cast<uint32_t *>(list);
  __shared__ unsigned long long pt_pair[PT_SLOTS];
  __shared__ uint32_t pt_key[PT_SLOTS];
  __shared__ uint32_t nlist, pt_n, pt_base, seg_fill;
  uint32_t *const tab_id = sk;   // the component table lives in the keys' storage once the pairs are reduced
  unsigned long long *const tab_val = reinterpret_cast<unsigned long long *>(sk + SC_SLOTS);

  // ---- this block's tiles: its XCD's band, strided over the XCD's blocks (tile j, j + blocks, ...), so that the blocks of
  // an XCD work on neighbouring tiles at any time; RDGPU_FILL_PAIRS_STRIDED=0: a run of consecutive tiles per block
  // (measured 0.15 ms slower; neither keeps the neighbours' rows in the 4 MB L2 for the ring columns -- hence the edge
  // records: 22.9 GB fetched per launch without them, 12.5 with).
  const uint32_t xcd = blockIdx.x & 7u, kb = gridDim.x >> 3, per = (nwork + 7u) / 8u;
  const uint32_t seg = blockIdx.x;   // the block's own segment of the pair list: no counter in HBM to wait for
  const uint32_t run = (per + kb - 1u) / kb;
  const bool strided = (precheck & 2) != 0;
  uint32_t it = strided ? (blockIdx.x >> 3) : (blockIdx.x >> 3) * run;
  const uint32_t it_end = strided ? per : min(it + run, per);
  const uint32_t kstep = strided ? kb : 1u;
  auto next_tile = [&](uint32_t &i) -> uint32_t {
    while (i < it_end) {
      const uint32_t wi = xcd * per + i;
      if (wi >= nwork) break;
      const uint32_t t = tiles_in ? tiles_in[wi] : wi;
      if (!skip) return t;
      const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
      if (!skip[(uint32_t)(y0 / DH) * dtx + (uint32_t)(x0 / DW)]) return t;
      i += kstep;
    }
    i = it_end;
    return NO_TILE;
  };
  if (threadIdx.x == 0) seg_fill = 0;

  // ---- the prefetched tile (registers) -----------------------------------------------------------------------------
  Quad<T> zq[2];
  Quad<uint16_t> lq[2];
  T rz = T();
  uint32_t rl = 0, rtb = 0, rc = 0, tbC = 0, cnt = 0, rkey = 0;
  uint32_t nt[NT_CAP / NTHR];
  // ring cell of a thread (threads >= RING repeat the last one's loads and store nothing).  Everything that depends on
  // the thread index is recomputed from an opaque copy of it in every trip of the tile loop: hoisted out of the loop
  // these few dozen offsets and addresses cost 30 VGPRs, i.e. a third of the resident blocks.
  auto ring_y = [](int tid) { const int rk = min(tid, RING - 1); return rk < LW ? 0 : rk < 2 * LW ? LH - 1 : (rk - 2 * LW < TH ? rk - 2 * LW + 1 : rk - 2 * LW - TH + 1); };
  auto ring_x = [](int tid) { const int rk = min(tid, RING - 1); return rk < LW ? rk : rk < 2 * LW ? rk - LW : (rk - 2 * LW < TH ? 0 : LW - 1); };

  auto load_a = [&](uint32_t t, int tid) {   // everything that only needs the tile's coordinates; branch free (clamped addresses)
    const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
    const uint32_t dC = (uint32_t)(y0 / DH) * dtx + (uint32_t)(x0 / DW);
    tbC = tile_base[dC];
    cnt = tile_count[dC];
    {   // the ring first: its component gather (load_b) then only waits for these
      const int rx = ring_x(tid), ry = ring_y(tid);
      const int gx = min(max(x0 - 1 + rx, 0), w - 1), gy = min(max(y0 - 1 + ry, 0), h - 1);
      // a ring COLUMN cell comes from the neighbouring descent tile's edge records (edgeK / edgeS, when the descent
      // wrote them): the left ring column is that tile's last column, the right one its first
      const bool col = edgeK && ry >= 1 && ry <= TH && tid < RING;
      const uint32_t dtn = (uint32_t)(gy / DH) * dtx + (uint32_t)(gx / DW);
      const size_t e = ((size_t)dtn * 2 + (rx == 0 ? 1 : 0)) * DH + (uint32_t)(gy % DH);
      const size_t g = col ? (size_t)y0 * w + x0 : (size_t)gy * w + gx;   // (a column cell reads nothing new from the raster)
      rtb = tile_base[dtn];
      rl = col ? (uint32_t)edgeS[e] : (uint32_t)lab16[g];
      rkey = col ? edgeK[e] : 0u;
      rz = z[g];
    }
    // the tile's quads: the tile's first cell is a block-uniform (scalar) address, a thread adds a 32-bit byte offset
    const size_t tile0 = (size_t)y0 * w + x0;
    const char *zt = reinterpret_cast<const char *>(z + tile0), *lt = reinterpret_cast<const char *>(lab16 + tile0);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int i = tid + r * NTHR;
      const int ly = i / (TW / 4), q = i - ly * (TW / 4);
      const int gx = x0 + 4 * q, gy = y0 + ly;
      const bool ok = gy < h && gx < w;
      if (VEC) {
        const uint32_t cell = ok ? (uint32_t)ly * (uint32_t)w + 4u * (uint32_t)q : 0u;
        struct alignas(4 * sizeof(T)) AZ { T v[4]; };
        struct alignas(8) AL { uint16_t v[4]; };
        const AL a = *reinterpret_cast<const AL *>(lt + cell * (uint32_t)sizeof(uint16_t));
        const AZ b = *reinterpret_cast<const AZ *>(zt + cell * (uint32_t)sizeof(T));
#pragma unroll
        for (int e = 0; e < 4; e++) { lq[r].v[e] = a.v[e]; zq[r].v[e] = b.v[e]; }
      } else {
        const size_t row = ok ? (size_t)gy * w : 0;
        lq[r] = load_quad<uint16_t, false>(lab16 + row, ok ? gx : 0, w, (uint16_t)0);
        zq[r] = load_quad<T, false>(z + row, ok ? gx : 0, w, T());
        if (!ok)   // (the raster's first cells stand in: THEIR labels belong to another descent tile; slot 0 exists in every tile)
#pragma unroll
          for (int e = 0; e < 4; e++) lq[r].v[e] = 0;
      }
    }
  };
  auto load_b = [&](int tid) {   // what needs load_a's words: the node table, the ring cells' components
    const uint32_t nn = (cnt <= (uint32_t)NT_CAP && cnt) ? cnt : 1u;
#pragma unroll
    for (int k = 0; k < NT_CAP / NTHR; k++) {
      const uint32_t idx = (uint32_t)tid + k * NTHR;
      nt[k] = curN[tbC + min(idx, nn - 1u)];   // (cnt >= 1: the tile holds a cell)
    }
    rc = curN[rtb + rl];
  };

  uint32_t t = next_tile(it);
  if (t == NO_TILE) return;
  load_a(t, (int)threadIdx.x);
  load_b((int)threadIdx.x);
  constexpr int ROWS = TH / 4;
  for (;;) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: see above
    const int lx = tid & (TW - 1), band = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int rly = ring_y(tid), rlx = ring_x(tid);
    const bool isring = tid < RING;
    const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
    // ---- stage 1: the node table of the tile's descent tile; reset the tile's tables --------------------------------
    const bool tabled = cnt <= (uint32_t)NT_CAP;
#pragma unroll
    for (int k = 0; k < NT_CAP / NTHR; k++) ntab[tid + k * NTHR] = nt[k];
    pt_pair[tid] = ~0ull;
    pt_key[tid] = 0xFFFFFFFFu;
    static_assert(PT_SLOTS == NTHR, "one slot per thread");
    if (tid == 0) { nlist = 0; pt_n = 0; }
    __syncthreads();
    // ---- stage 2: keys and components of the tile and its ring into LDS ---------------------------------------------
    {
      uint32_t cq[2][4];
      // (a quad outside the raster holds labels of this descent tile -- its first quad's, or zeros: the lookups need no
      // guard -- and a label is < cnt <= NT_CAP when the table is used)
      if (tabled) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) cq[r][e] = ntab[lq[r].v[e]];
      } else {   // more nodes than the table holds (white noise): gathered
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) cq[r][e] = curN[tbC + (uint32_t)lq[r].v[e]];
      }
      if (x0 + TW <= w && y0 + TH <= h) {   // (block-uniform) the whole tile lies in the raster: no tests
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int i = tid + r * NTHR;
          const int ly = i / (TW / 4), q = i - ly * (TW / 4);
          const int o = (ly + 1) * LW + 1 + 4 * q;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sk[o + e] = Key32<T>::to(zq[r].v[e]);
            sc[o + e] = cq[r][e];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int i = tid + r * NTHR;
          const int ly = i / (TW / 4), q = i - ly * (TW / 4);
          const int gx = x0 + 4 * q, gy = y0 + ly;
          const int o = (ly + 1) * LW + 1 + 4 * q;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const bool in = gy < h && gx + e < w;
            sk[o + e] = in ? Key32<T>::to(zq[r].v[e]) : 0u;
            sc[o + e] = in ? cq[r][e] : (B | CLOSED);
          }
        }
      }
      if (isring) {
        const int gx = x0 - 1 + rlx, gy = y0 - 1 + rly;
        const bool ok = gx >= 0 && gx < w && gy >= 0 && gy < h;
        const bool col = edgeK && rly >= 1 && rly <= TH;
        sk[rly * LW + rlx] = ok ? (col ? rkey : Key32<T>::to(rz)) : 0u;
        sc[rly * LW + rlx] = ok ? rc : (B | CLOSED);
      }
    }
    // ---- the next tile's loads go out now and land while this tile is worked on ---------------------------------------
    it += kstep;
    const uint32_t tn = next_tile(it);
    const uint32_t tl = tn != NO_TILE ? tn : t;   // (no next tile: this one's rows once more -- cheaper than a branch)
    load_a(tl, tid);
    __syncthreads();
    // ---- phase 1: detect (k_scan's) -----------------------------------------------------------------------------------
    const int gxl = x0 + lx;
    const int yb = band * ROWS;
    {
      uint32_t c1[3], c2[3];
      {
        const int o = (yb + 1) * LW + lx;
#pragma unroll
        for (int e = 0; e < 3; e++) c1[e] = sc[o + e];
      }
      unsigned long long bal[ROWS];
      bool bnd[ROWS];
      // (k_scan's "the tile still holds an open boundary" flag feeds the raster rounds, which this path does not have)
      const int rows_in = min(h - y0 - yb, ROWS);   // rows of this band inside the raster
      const bool colin = gxl < w;
#pragma unroll
      for (int j = 0; j < ROWS; j++) {
        const int ly = yb + j;
        {
          const int o = (ly + 2) * LW + lx;
#pragma unroll
          for (int e = 0; e < 3; e++) c2[e] = sc[o + e];
        }
        const uint32_t C = c1[1];
        uint32_t df = (c1[2] ^ C) | (c2[1] ^ C);
        if (TOPO == 8) df |= (c2[0] ^ C) | (c2[2] ^ C);
        bnd[j] = df != 0 && colin && j < rows_in;
        bal[j] = __builtin_amdgcn_ballot_w64(bnd[j]);   // (the builtin: __ballot() costs two VALU instructions)
#pragma unroll
        for (int e = 0; e < 3; e++) c1[e] = c2[e];
      }
      // (the node table's storage becomes the list: every lookup in it was made before the barrier above)
      uint32_t total = 0;
#pragma unroll
      for (int j = 0; j < ROWS; j++) total += (uint32_t)__popcll(bal[j]);
      if (total && !(precheck & 16)) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&nlist, total);
        base = __shfl(base, 0, 64);
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
          if (bnd[j])
            list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[j], 0u))] =
                (uint16_t)((yb + j + 1) * LW + lx + 1);
          base += (uint32_t)__popcll(bal[j]);
        }
      }
    }
    __syncthreads();
    load_b(tid);
    // ---- phase 2: the pairs (k_scan<EMIT>'s) -----------------------------------------------------------------------
    const uint32_t nl = (precheck & 4) ? 0u : nlist;
    constexpr int NF = TOPO == 8 ? 4 : 2;
    const int foff[4] = {1, TOPO == 8 ? LW + 1 : LW, LW, LW - 1};
    for (uint32_t i = tid; i < nl; i += NTHR) {
      const int o = list[i];
      const uint32_t C = sc[o], kc = sk[o];
      uint32_t nD[NF], nH[NF];
#pragma unroll
      for (int e = 0; e < NF; e++) { nD[e] = sc[o + foff[e]]; nH[e] = sk[o + foff[e]]; }
      if (precheck & 64) {   // (timing probe: the gathers alone)
        uint32_t sink = C ^ kc;
#pragma unroll
        for (int e = 0; e < NF; e++) sink ^= nD[e] ^ nH[e];
        if (sink == 0x12345u) nlist = 1;
        continue;
      }
      uint32_t pd[2] = {C, C}, pk[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
      for (int e = 0; e < NF; e++) {
        const uint32_t D = nD[e], hn = nH[e] > kc ? nH[e] : kc;
        if (D != C && !(C & D & CLOSED)) {
          if (pd[0] == C) pd[0] = D;
          if (D == pd[0]) pk[0] = hn < pk[0] ? hn : pk[0];
          else {
            if (pd[1] == C) pd[1] = D;
            if (D == pd[1]) pk[1] = hn < pk[1] ? hn : pk[1];
            else if (!pair_insert(pt_pair, pt_key, C < D ? C : D, C < D ? D : C, hn))
              pair_spill_local(eo, best, seg, &seg_fill, C < D ? C : D, C < D ? D : C, hn);
          }
        }
      }
      if (precheck & 32) {   // (timing probe: gathers + the neighbours' sorting into two components, no table)
        if ((pd[0] ^ pd[1] ^ pk[0] ^ pk[1]) == 0x12345u) nlist = 1;
        continue;
      }
      uint32_t ps[2], pq[2], lo[2], hi[2];
      unsigned long long pv[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        lo[j] = C < pd[j] ? C : pd[j];
        hi[j] = C < pd[j] ? pd[j] : C;
        ps[j] = pair_home(lo[j], hi[j]);
        pv[j] = pt_pair[ps[j]];
        pq[j] = pt_key[ps[j]];
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (pd[j] != C) {
          const unsigned long long pr = ((unsigned long long)lo[j] << 32) | hi[j];
          if (pv[j] == pr) { if (pk[j] < pq[j]) atomicMin(&pt_key[ps[j]], pk[j]); }
          else if (!pair_insert(pt_pair, pt_key, lo[j], hi[j], pk[j])) pair_spill_local(eo, best, seg, &seg_fill, lo[j], hi[j], pk[j]);
        }
      }
    }
    __syncthreads();
    tab_id[tid] = 0xFFFFFFFFu;
    tab_val[tid] = ~0ull;
    static_assert(SC_SLOTS == NTHR, "one slot per thread");
    {
      const bool occ = pt_pair[tid] != ~0ull;
      const unsigned long long bal = __ballot(occ);
      uint32_t base = 0;
      if (lane == 0 && bal) base = atomicAdd(&pt_n, (uint32_t)__popcll(bal));
      base = __shfl(base, 0, 64);
      if (occ) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    }
    __syncthreads();
    const uint32_t tot = (precheck & 8) ? 0u : pt_n;
    if (tid == 0) {
      uint32_t ob = seg_fill;   // (spills are over: nothing else touches the counter until the next tile's pairs)
      if (ob + tot > eo.seglimit) { *eo.overflow = 1; ob = 0xFFFFFFFFu; }
      else seg_fill = ob + tot;
      pt_base = ob;
    }
    for (uint32_t i = tid; i < tot; i += NTHR) {
      const int sl = list[i];
      const unsigned long long pr = pt_pair[sl];
      const uint32_t lo = (uint32_t)(pr >> 32), hi = (uint32_t)pr, key = pt_key[sl];
      if (!(lo & CLOSED)) {
        const unsigned long long cand = ((unsigned long long)key << 32) | hi;
        const int slot = tab_slot(tab_id, lo);
        if (slot >= 0) atomicMin(&tab_val[slot], cand);
        else if (cand < best[lo]) atomicMin(&best[lo], cand);
      }
      if (!(hi & CLOSED)) {
        const unsigned long long cand = ((unsigned long long)key << 32) | lo;
        const int slot = tab_slot(tab_id, hi);
        if (slot >= 0) atomicMin(&tab_val[slot], cand);
        else if (cand < best[hi]) atomicMin(&best[hi], cand);
      }
    }
    __syncthreads();
    {
      const uint32_t C = tab_id[tid];
      if (C != 0xFFFFFFFFu) {
        const unsigned long long cand = tab_val[tid];
        if (precheck & 1) { if (cand < best[C]) atomicMin(&best[C], cand); }
        else atomicMin(&best[C], cand);
      }
    }
    const uint32_t ob = pt_base;
    if (ob != 0xFFFFFFFFu) {
      const size_t g0 = (size_t)seg * eo.segcap + ob;
      for (uint32_t i = tid; i < tot; i += NTHR) {
        const int sl = list[i];
        const unsigned long long pr = pt_pair[sl];
        eo.a[g0 + i] = (uint32_t)(pr >> 32);
        eo.b[g0 + i] = (uint32_t)pr;
        eo.k[g0 + i] = pt_key[sl];
      }
    }
    __syncthreads();   // the tables, the list and the keys' storage are free for the next tile
    if (tn == NO_TILE) break;
    t = tn;
  }
  if (threadIdx.x == 0) eo.segcount[seg] = seg_fill;
}

template <class T>
static void fill_finalize16(T *d_z, int w, int h, const FillBuffers &fb, hipStream_t s) {
  if (fb.trivial) return;
  const uint32_t dtx = cdiv(w, DW), dnt = dtx * cdiv(h, DH);
  const dim3 ngrid(cdiv(std::max(fb.nnmax, 1u), NTHR), fb.nstripes);
  RD_LAUNCH("fill.node_levels", k_node_levels, ngrid, dim3(NTHR), 0, s, (const uint32_t *)fb.curN, (const uint32_t *)fb.acc,
            (const unsigned long long *)fb.counters, fb.rcap, fb.lvl);
  const bool vec = (w % 4) == 0 && (reinterpret_cast<uintptr_t>(d_z) % (4 * sizeof(T))) == 0;
  if (vec)
    RD_LAUNCH("fill.finalize", (k_finalize16<T, true>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, d_z, (const uint16_t *)fb.lab16,
              (const uint32_t *)fb.lvl, (const uint32_t *)fb.tile_base, (const uint32_t *)fb.tile_count, w, h, dtx, dnt,
              (const uint8_t *)nullptr, (const uint32_t *)nullptr);
  else
    RD_LAUNCH("fill.finalize", (k_finalize16<T, false>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, d_z, (const uint16_t *)fb.lab16,
              (const uint32_t *)fb.lvl, (const uint32_t *)fb.tile_base, (const uint32_t *)fb.tile_count, w, h, dtx, dnt,
              (const uint8_t *)nullptr, (const uint32_t *)nullptr);
}

// The compact-label fill's host side.  false: the DEM does not fit the scheme's buffers (more nodes or pair records than
// provided for: e.g. white noise) or it was switched off -- the DEM has not been changed, the classic path runs.
// lists (optional, with skip): device arrays [descent tiles to visit | scan tiles to visit | tiles to finalize], each of
// `stride` entries, and their lengths on the host
// resident blocks of the persistent pair pass: what the kernel's registers and LDS let a CU hold (asked of the runtime;
// RDGPU_FILL_PAIRS_BPC overrides), times the CUs, rounded to a multiple of 8 so that every XCD gets the same number
template <class K>
static int pairs_blocks(K kernel) {
  int dev = 0, cus = 0, bpc = 0;
  RD_HIP(hipGetDevice(&dev));
  RD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const char *e = getenv("RDGPU_FILL_PAIRS_BPC");
  if (e) bpc = atoi(e);
  else RD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kernel, NTHR, 0));
  return std::max(8, (cus * std::max(bpc, 1)) / 8 * 8);
}

struct SparseLists {
  const uint32_t *d = nullptr;
  uint32_t stride = 0, n[3] = {0, 0, 0};
};
// keep (with alloc): the call is a row-block shard's local phase -- open_top / open_bottom name its cut rows, the tables the
// shard needs later (labels, node tables, basins' components and levels, terminal ids) come from `alloc` and are handed
// over in *keep, and the DEM is NOT raised (rdgpu_fill_shard_finish does that, after the levels of the cut-row
// terminals are known).
template <class T, int TOPO>
static bool fill_fused(T *d_z, int w, int h, hipStream_t s, const uint8_t *outlet = nullptr, const uint8_t *skip = nullptr,
                       const SparseLists *lists = nullptr, FillBuffers *keep = nullptr, BufAlloc *alloc = nullptr,
                       int open_top = 0, int open_bottom = 0) {
  const char *fe = getenv("RDGPU_FILL_FUSED");   // =0: the classic four-pass fill (A/B and tests)
  if (fe && fe[0] == '0') return false;
  const char *env_edges = getenv("RDGPU_FILL_EDGES");
  if (env_edges && env_edges[0] == '0') return false;   // (the raster-round fallback lives in the classic path)
  const uint64_t n64 = (uint64_t)w * (uint64_t)h;
  if (n64 > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: raster (or shard) has more than 2^31-65536 cells");
  const uint32_t n = (uint32_t)n64;
  if (w <= 2 || h <= 2) return false;   // every cell is a border cell: the classic path's trivial case
  g_stats = rdgpu_fill_stats{n64, 0, 0, 0, 0, (uint32_t)(TW * TH), 0};
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  const uint32_t dtx = cdiv(w, DW), dty = cdiv(h, DH), dnt = dtx * dty;
  const uint32_t tilesX = cdiv(w, TW), tilesY = cdiv(h, TH), ntiles = tilesX * tilesY;
  const bool vec = (w % 4) == 0 && (reinterpret_cast<uintptr_t>(d_z) % (4 * sizeof(T))) == 0;
  FusedBuf fo;
  fo.gcap = n / 4 + 4096;
  uint32_t nstripes = 1;   // every stripe's region holds at least two tiles of nothing but roots
  while (nstripes < (uint32_t)FSTRIPES && fo.gcap / (2 * nstripes) >= 8192u && dnt >= 8 * nstripes) nstripes *= 2;
  fo.rcap = fo.gcap / nstripes;
  fo.smask = nstripes - 1;
  const bool sharded = keep != nullptr;
  if (sharded && !(open_top || open_bottom)) return false;   // (a block without a cut is the whole raster: the caller's other path)
  // what outlives the call in a shard comes from the shard's allocator
  auto persistent = [&](const char *name, size_t bytes) -> void * {
    return sharded ? (void *)alloc->get<uint8_t>(name, bytes) : ws.buf(name, bytes);
  };
  fo.lab16 = (uint16_t *)persistent("fused.lab16", (size_t)n * 2);
  fo.G = (uint32_t *)persistent("fused.G", (size_t)fo.gcap * 4);
  fo.tile_base = (uint32_t *)persistent("fused.tile_base", (size_t)dnt * 4);
  fo.tile_count = (uint32_t *)persistent("fused.tile_count", (size_t)dnt * 4);
  // [0] chase flag, [2] roots, [3] alive tiles, [4] records, [5] overflow, [8] basins, [9] nodes, [10] nodes of the fullest stripe
  uint32_t *dflags = ws.buf<uint32_t>("fused.flags", 16);
  fo.counters = (unsigned long long *)persistent("fused.counters", (size_t)FSTRIPES * FSTRIDE * 8);
  uint32_t *pitoff = ws.buf<uint32_t>("fused.pitoff", FSTRIPES);
  fo.overflow = dflags + 5;
  uint32_t *curN = (uint32_t *)persistent("fused.curN", (size_t)fo.gcap * 4);
  const char *env_edge = getenv("RDGPU_FILL_EDGECOLS");   // =0: ring columns from the raster (A/B)
  if (!outlet && !(env_edge && env_edge[0] == '0')) {   // (with outlets, tiles are skipped: their records would be stale)
    fo.edgeK = ws.buf<uint32_t>("fused.edgeK", (size_t)dnt * 2 * DH);
    fo.edgeS = ws.buf<uint16_t>("fused.edgeS", (size_t)dnt * 2 * DH);
  }
  RD_HIP(hipMemsetAsync(dflags, 0, 16 * sizeof(uint32_t), s));
  RD_HIP(hipMemsetAsync(fo.counters, 0, (size_t)FSTRIPES * FSTRIDE * sizeof(unsigned long long), s));
  if (outlet) {   // node 0 (the first of stripe 0): the "outside" node of the skipped tiles
    static const uint32_t one = 1u, outp = OUTP;
    RD_HIP(hipMemcpyAsync(reinterpret_cast<uint32_t *>(fo.counters) + 1, &one, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    RD_HIP(hipMemcpyAsync(fo.G, &outp, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  }
  const bool listed = outlet && skip && lists && lists->d;
  const uint32_t *dl = listed ? lists->d : nullptr, *sl_ = listed ? lists->d + lists->stride : nullptr,
                 *fl = listed ? lists->d + 2 * (size_t)lists->stride : nullptr;
  if (listed && lists->n[0] == 0) return true;   // nothing but walls anywhere: nothing to raise
  if (outlet && vec)
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, true, true>), dim3(listed ? lists->n[0] : xcd_grid(dnt)), dim3(NTHR), 0, s,
              (const T *)d_z, fo, w, h, dtx, dnt, outlet, skip, dl, 0, 0);
  else if (outlet)
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, false, true>), dim3(listed ? lists->n[0] : xcd_grid(dnt)), dim3(NTHR), 0, s,
              (const T *)d_z, fo, w, h, dtx, dnt, outlet, skip, dl, 0, 0);
  else if (sharded && vec)
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, true, false, true>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, (const T *)d_z, fo, w, h, dtx,
              dnt, (const uint8_t *)nullptr, (const uint8_t *)nullptr, (const uint32_t *)nullptr, open_top, open_bottom);
  else if (sharded)
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, false, false, true>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, (const T *)d_z, fo, w, h, dtx,
              dnt, (const uint8_t *)nullptr, (const uint8_t *)nullptr, (const uint32_t *)nullptr, open_top, open_bottom);
  else if (vec)
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, true>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, (const T *)d_z, fo, w, h, dtx, dnt,
              (const uint8_t *)nullptr, (const uint8_t *)nullptr, (const uint32_t *)nullptr, 0, 0);
  else
    RD_LAUNCH("fill.descent", (k_descent16<T, TOPO, false>), dim3(xcd_grid(dnt)), dim3(NTHR), 0, s, (const T *)d_z, fo, w, h, dtx, dnt,
              (const uint8_t *)nullptr, (const uint8_t *)nullptr, (const uint32_t *)nullptr, 0, 0);
  RD_LAUNCH("fill.stripe_offsets", k_stripe_offsets, dim3(1), dim3(64), 0, s, (const unsigned long long *)fo.counters, nstripes, pitoff,
            dflags + 8);
  RD_HIP(hipMemcpyAsync(hw, dflags + 4, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  g_stats.host_syncs++;
  if (hw[1] != 0) { if (getenv("RDGPU_FILL_DEBUG")) fprintf(stderr, "fill_fused: node table overflow (nstripes %u rcap %u)\n", nstripes, fo.rcap); return false; }   // more nodes than a stripe's table holds: nothing was written to the DEM
  const uint32_t B = hw[4], NNmax = hw[6];   // basins; nodes of the fullest stripe
  g_stats.basins = B;
  if (sharded) {
    *keep = FillBuffers();
    keep->B = B;
    if (B == 0) { keep->trivial = true; return true; }   // no pits and no terminals: nothing to raise
  }
  if (B == 0) return true;        // no pits: nothing to raise
  const dim3 ngrid(cdiv(std::max(NNmax, 1u), NTHR), nstripes);
  uint32_t *tid = nullptr;
  if (sharded) {
    tid = alloc->get<uint32_t>("fill.tid", (size_t)B + 1);
    RD_HIP(hipMemsetAsync(tid, 0xFF, ((size_t)B + 1) * sizeof(uint32_t), s));
    RD_LAUNCH("fill.mark_terminals", k_mark_terminals16, dim3(cdiv(w, NTHR)), dim3(NTHR), 0, s, (const uint16_t *)fo.lab16,
              (const uint32_t *)fo.tile_base, (const uint32_t *)fo.G, (const uint32_t *)pitoff, fo.rcap, tid, w, h, dtx, open_top,
              open_bottom);
  }
  RD_LAUNCH("fill.resolve_nodes", k_resolve_nodes, ngrid, dim3(NTHR), 0, s, fo.G, (const unsigned long long *)fo.counters,
            (const uint32_t *)pitoff, fo.rcap, (const uint16_t *)fo.lab16, (const uint32_t *)fo.tile_base, w, dtx, B, curN, dflags,
            (const uint32_t *)tid);
  g_stats.jump_passes = 1;
  uint32_t *cur = sharded ? alloc->get<uint32_t>("fill.cur", (size_t)B + 1) : ws.buf<uint32_t>("fill.cur", (size_t)B + 1);
  uint32_t *acc = sharded ? alloc->get<uint32_t>("fill.acc", (size_t)B + 1) : ws.buf<uint32_t>("fill.acc", (size_t)B + 1);
  unsigned long long *best = ws.buf<unsigned long long>("fill.best", (size_t)B + 1);
  unsigned long long *link = ws.buf<unsigned long long>("fill.link", (size_t)B + 1);
  uint32_t *rootsA = ws.buf<uint32_t>("fill.rootsA", B);
  uint32_t *rootsB = ws.buf<uint32_t>("fill.rootsB", B);
  RD_HIP(hipMemsetAsync(dflags + 2, 0, sizeof(uint32_t), s));
  RD_LAUNCH("fill.init_tables", k_init_tables, dim3(cdiv((uint64_t)B + 1, NTHR)), dim3(NTHR), 0, s, cur, acc, link, rootsA,
            dflags + 2, (const uint32_t *)tid, B);
  uint32_t nroots0 = B;
  if (sharded) {   // the frozen terminals are no roots: the list was compacted, its length comes back
    RD_HIP(hipMemcpyAsync(hw, dflags + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    g_stats.host_syncs++;
    nroots0 = hw[0];
  }
  const char *env_pp = getenv("RDGPU_FILL_PAIRS");   // =0: k_scan<L16>, one block per tile (A/B and tests)
  const bool persistent_pairs = !(env_pp && env_pp[0] == '0');
  const uint32_t nwork1 = listed ? lists->n[1] : ntiles;   // scan tiles the pair pass visits
  uint32_t pgrid = 0;
  if (persistent_pairs) {
    static thread_local int pb_cache[2] = {0, 0};   // (per element type and topology: this function is a template)
    int &pb = pb_cache[vec ? 1 : 0];
    if (!pb || getenv("RDGPU_FILL_PAIRS_BPC")) pb = vec ? pairs_blocks(k_pairs16<T, TOPO, true>) : pairs_blocks(k_pairs16<T, TOPO, false>);
    pgrid = std::max(8u, std::min<uint32_t>(xcd_grid(std::max(nwork1, 1u)), (uint32_t)pb));
  }
  // the pair list of the one raster pass: capacity as in the classic path; the persistent pass has one segment per block
  uint32_t nseg = 1;
  while (nseg < ESEG && (uint64_t)nseg * 128 <= ntiles) nseg *= 2;
  if (persistent_pairs) nseg = pgrid;
  const char *env_cap = getenv("RDGPU_FILL_EDGE_CAP");
  const uint64_t cap = env_cap ? strtoull(env_cap, nullptr, 10) : std::min<uint64_t>(12ull * B, n / 2) + 2048;
  const uint32_t segcap = cdiv(cdiv(cap, nseg), NTHR * EPT) * (NTHR * EPT);
  const size_t ecap = (size_t)nseg * segcap;
  uint32_t *elist[2] = {ws.buf<uint32_t>("fill.edges0", 3 * ecap), nullptr};
  size_t pcap[2] = {ecap, 0};
  uint32_t *segcount = ws.buf<uint32_t>("fill.segcount", nseg);
  RD_HIP(hipMemsetAsync(segcount, 0, nseg * sizeof(uint32_t), s));
  EdgeOut eo{elist[0], elist[0] + ecap, elist[0] + 2 * ecap, segcount, segcap, nseg - 1, dflags + 5};
  eo.seglimit = env_cap ? (uint32_t)std::min<uint64_t>(segcap, std::max<uint64_t>(1, cdiv(cap, nseg))) : segcap;   // (the cap given on purpose is enforced to the record: the overflow tests)
  uint8_t *alive = ws.buf<uint8_t>("fill.alive", ntiles);
  int ein = 0;
  bool eseg = true;
  const char *env_dedup = getenv("RDGPU_FILL_DEDUP");
  const bool dedup = !(env_dedup && env_dedup[0] == '0');
  const char *env_pc = getenv("RDGPU_FILL_PRECHECK");
  const char *env_st = getenv("RDGPU_FILL_PAIRS_STRIDED");
  // RDGPU_FILL_PAIRS_ABLATE (timing probes only, compiled in with -DRDGPU_PROBES -- the fill's RESULT IS WRONG with any bit set): 4 = no pair loop (phase 2),
  // 8 = no proposals and no records (phase 3), 16 = no boundary list (phase 1), 32 = the pair loop without its table trips,
  // 64 = the pair loop's gathers alone: what each part costs, tools/probes/pairs_ablate.sh.  r05 at S3 (profiles/r05e_*):
  // staging + detection 2.5 ms, list 0.4, gathers 0.45, sorting the neighbours 1.0, table 1.8, proposals + records 0.5 = 6.1.
  // Built on that and measured SLOWER or equal, not kept: the pair pass on local component ids with a direct triangular pair
  // table (9.5 ms: its LDS atomics and the id lookups cost more than the hash they replace), one table trip per cell with
  // the cells of a second component on a wavefront's own list (6.1-6.2), a two-slot fast path (6.2).
#ifdef RDGPU_PROBES   // (a stray environment variable must not be able to corrupt a production fill)
  const char *env_ab = getenv("RDGPU_FILL_PAIRS_ABLATE");
#else
  const char *env_ab = nullptr;
#endif
  const int precheck = (!(env_pc && env_pc[0] == '0') ? 1 : 0) | (!(env_st && env_st[0] == '0') ? 2 : 0) | (env_ab ? (atoi(env_ab) & 124) : 0);
  // r06: the rounds are enqueued WITHOUT a host read-back between them.  Every round kernel takes its counts from the device
  // (rc[4 r] = live roots entering round r, rc[4 r + 1] = records of the list it contracts) and runs a grid-stride loop over
  // a grid the host sizes from upper bounds: live roots at least halve per round (every one hooks into another component or
  // the outside), the record list never grows.  After a batch of rounds -- as many as the halving bound and the measured
  // shrink (a factor 4 - 9 per round) make plausible -- ONE synchronisation reads every round's counts and the error flags;
  // only a fill that is not finished by then (never seen) enqueues more.  A fill has two synchronisations: after the
  // descent (the basin count sizes the tables) and here.  (r05: one per round, 10 per fill at S3 -- on a busy host each is a
  // scheduling quantum, profiles/README.md r05u.)
  constexpr int MAXR = 48;
  uint32_t *rc = ws.buf<uint32_t>("fused.round_counts", 4 * (MAXR + 2));
  RD_HIP(hipMemsetAsync(rc, 0, 4 * (MAXR + 2) * sizeof(uint32_t), s));
  hw[220] = nroots0;   // (pinned; beyond the words the read-backs use)
  RD_HIP(hipMemcpyAsync(rc + 4, hw + 220, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  pcap[1] = ecap;   // (the second list is sized for whatever the first pass may leave: its count stays on the device)
  elist[1] = ws.buf<uint32_t>("fill.edges1", 3 * pcap[1]);
  const uint32_t gcap = 2048;   // blocks of a grid-stride launch: 8 per CU
  int rdone = 0;                // rounds enqueued so far
  uint32_t rounds_run = 0;
  auto enqueue_round = [&](int r) {   // r = 1, 2, ...
    const uint32_t bound = std::max(1u, r - 1 < 31 ? nroots0 >> (r - 1) : 1u);   // live roots at most halve... at least
    const uint32_t rgrid = std::min(gcap, cdiv(bound, NTHR));
    const uint32_t *nr = rc + 4 * r, *ne = rc + 4 * r + 1;
    uint32_t *nr_next = rc + 4 * (r + 1), *ne_next = rc + 4 * (r + 1) + 1;
    RD_LAUNCH("fill.best_reset", k_best_reset, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, best, nr);
    if (r == 1) {   // round 1: the one raster pass (components gathered from the node table)
      const uint32_t nwork = listed ? lists->n[1] : ntiles;
      if (nwork == 0) {
        // (no tile holds a wet cell although basins exist: cannot happen -- a pit is a wet cell; kept safe)
      } else if (persistent_pairs) {
        if (vec)
          RD_LAUNCH("fill.scan", (k_pairs16<T, TOPO, true>), dim3(pgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint16_t *)fo.lab16,
                    (const uint32_t *)curN, best, w, h, B, tilesX, sl_, nwork, eo, (const uint32_t *)fo.tile_base,
                    (const uint32_t *)fo.tile_count, dtx, skip, precheck, (const uint32_t *)fo.edgeK, (const uint16_t *)fo.edgeS);
        else
          RD_LAUNCH("fill.scan", (k_pairs16<T, TOPO, false>), dim3(pgrid), dim3(NTHR), 0, s, (const T *)d_z, (const uint16_t *)fo.lab16,
                    (const uint32_t *)curN, best, w, h, B, tilesX, sl_, nwork, eo, (const uint32_t *)fo.tile_base,
                    (const uint32_t *)fo.tile_count, dtx, skip, precheck, (const uint32_t *)fo.edgeK, (const uint16_t *)fo.edgeS);
      } else if (vec)
        RD_LAUNCH("fill.scan", (k_scan<T, TOPO, false, true, true, true>), dim3(xcd_grid(nwork)), dim3(NTHR), 0, s, (const T *)d_z,
                  reinterpret_cast<const uint32_t *>(fo.lab16), (const uint32_t *)curN, best, w, h, B, tilesX, ntiles,
                  sl_, nwork, alive, eo, (const uint32_t *)fo.tile_base, dtx, skip);
      else
        RD_LAUNCH("fill.scan", (k_scan<T, TOPO, false, false, true, true>), dim3(xcd_grid(nwork)), dim3(NTHR), 0, s, (const T *)d_z,
                  reinterpret_cast<const uint32_t *>(fo.lab16), (const uint32_t *)curN, best, w, h, B, tilesX, ntiles,
                  sl_, nwork, alive, eo, (const uint32_t *)fo.tile_base, dtx, skip);
      RD_LAUNCH("fill.sum_segments", k_sum_segments, dim3(1), dim3(NTHR), 0, s, (const uint32_t *)eo.segcount, nseg, rc + 4 * 1 + 1);
      g_stats.scan_tiles += ntiles;
    } else {
      const int eout = ein ^ 1;
      const uint32_t *ia = elist[ein], *ib = elist[ein] + pcap[ein], *ik = elist[ein] + 2 * pcap[ein];
      uint32_t *oa = elist[eout], *ob = elist[eout] + pcap[eout], *ok = elist[eout] + 2 * pcap[eout];
      if (eseg)   // round 2 reads the segmented list of the raster pass: the segments' fill counts are on the device
        RD_LAUNCH("fill.edge_round", (k_edge_round<true, false>), dim3(std::min<uint32_t>(4 * gcap, cdiv(ecap, NTHR * EPT))), dim3(NTHR), 0,
                  s, ia, ib, ik, (uint32_t)ecap, (const uint32_t *)segcount, segcap, (const uint32_t *)cur, best, B, oa, ob, ok, ne_next);
      else if (dedup)
        RD_LAUNCH("fill.edge_round", (k_edge_round<false, true>), dim3(gcap), dim3(NTHR), 0, s, ia, ib, ik, 0u,
                  (const uint32_t *)nullptr, 0u, (const uint32_t *)cur, best, B, oa, ob, ok, ne_next, ne);
      else
        RD_LAUNCH("fill.edge_round", (k_edge_round<false, false>), dim3(gcap), dim3(NTHR), 0, s, ia, ib, ik, 0u,
                  (const uint32_t *)nullptr, 0u, (const uint32_t *)cur, best, B, oa, ob, ok, ne_next, ne);
      eseg = false;
      ein = eout;
    }
    RD_LAUNCH("fill.hook", k_hook, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, best, link, nr);
    // One pass: a thread follows its chain of hooks for up to 16384 steps (the chains of a round are a handful of hooks
    // long).  "A chain was left unfinished" stays in dflags[0] for the read-back after the batch and sends the raster to
    // the classic path, whose loop repeats the pass: nothing has been written to the DEM yet.
    RD_LAUNCH("fill.chase_links", k_chase_links, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, link, 1 << 14, dflags, nr);
    RD_LAUNCH("fill.update_basins", k_update_basins, dim3(cdiv(B, NTHR)), dim3(NTHR), 0, s, cur, acc, link, B, nr);
    RD_LAUNCH("fill.compact_roots", k_compact_roots, dim3(std::min(gcap, cdiv(bound, NTHR * RPT))), dim3(NTHR), 0, s, rootsA, 0u, link,
              rootsB, nr_next, nr);
    std::swap(rootsA, rootsB);
  };
  // round 1 writes the records' total where round 2 expects it
  // (k_sum_segments above: rc[4 * 1 + 1] is unused by round 1 itself; round 2 reads the segmented list and needs no count)
  uint32_t *hrc = ws.host_words() + 16;   // (pinned: the counts of every round, read once per batch)
  {
    uint32_t lg = 0;
    while ((1ull << lg) < (unsigned long long)nroots0 + 1ull) lg++;
    int batch = (precheck & 124) ? 1 : std::min(MAXR, std::max(4, (int)(2 * lg + 4) / 5 + 1));   // ~log5.7(roots) + 1: S3 enqueues 11, runs 9
    const char *env_batch = getenv("RDGPU_FILL_ROUND_BATCH");
    if (env_batch) batch = std::min(MAXR, std::max(1, atoi(env_batch)));   // (tests: several batches)
    uint32_t last_live = 0xFFFFFFFFu;
    for (;;) {
      for (int k = 0; k < batch && rdone < MAXR; k++) enqueue_round(++rdone);
      RD_HIP(hipMemcpyAsync(hrc, rc, 4 * (MAXR + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipMemcpyAsync(hw, dflags, 14 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      g_stats.host_syncs++;
      if (getenv("RDGPU_FILL_DEBUG")) fprintf(stderr, "fill_fused: pair pass: %u records, overflow %u, %u tiles on the slow road; %d rounds enqueued\n", hrc[4 + 1], hw[5], hw[12], rdone);
      if (hw[0] != 0) { if (getenv("RDGPU_FILL_DEBUG")) fprintf(stderr, "fill_fused: hook chain unfinished\n"); return false; }
      if (precheck & 124) break;   // RDGPU_FILL_PAIRS_ABLATE (timing probes): the pair pass ran, its output is not used
      if (hw[5] != 0) { if (getenv("RDGPU_FILL_DEBUG")) fprintf(stderr, "fill_fused: pair list overflow (B %u cap %llu nseg %u segcap %u nwork %u)\n", B, (unsigned long long)cap, nseg, segcap, nwork1); return false; }   // the pair list overflowed: the DEM is untouched, the classic path takes over
      rounds_run = 0;
      for (int r = 1; r <= rdone; r++) rounds_run += hrc[4 * r] != 0;
      g_stats.edge_records = hrc[4 + 1];
      const uint32_t live = hrc[4 * (rdone + 1)];
      if (live == 0) break;
      if (live >= last_live || rdone >= MAXR) throw Error(RDGPU_ERR_HIP, "rdgpu_fill: contraction made no progress (internal error)");
      last_live = live;
      if (!env_batch) batch = 4;
    }
  }
  g_stats.rounds += rounds_run;
  uint32_t *lvl = fo.G;   // (the node table is dead: its storage holds the nodes' levels)
  RD_LAUNCH("fill.node_levels", k_node_levels, ngrid, dim3(NTHR), 0, s, (const uint32_t *)curN, (const uint32_t *)acc,
            (const unsigned long long *)fo.counters, fo.rcap, lvl);
  if (sharded) {   // the watershed of every node, and everything rdgpu_fill_shard_export / _finish need
    uint32_t *nodeW = alloc->get<uint32_t>("fused.nodeW", fo.gcap);
    RD_LAUNCH("fill.node_watersheds", k_node_watersheds, ngrid, dim3(NTHR), 0, s, (const uint32_t *)curN, (const uint32_t *)cur,
              (const uint32_t *)tid, (const unsigned long long *)fo.counters, fo.rcap, B, nodeW);
    keep->compact = true;
    keep->cur = cur; keep->acc = acc; keep->tid = tid;
    keep->lab16 = fo.lab16; keep->tile_base = fo.tile_base; keep->tile_count = fo.tile_count;
    keep->curN = curN; keep->lvl = lvl; keep->nodeW = nodeW; keep->counters = fo.counters;
    keep->rcap = fo.rcap; keep->nstripes = nstripes; keep->nnmax = NNmax;
    return true;
  }
  if (listed && lists->n[2] == 0) return true;
  if (vec)
    RD_LAUNCH("fill.finalize", (k_finalize16<T, true>), dim3(listed ? lists->n[2] : xcd_grid(dnt)), dim3(NTHR), 0, s, d_z,
              (const uint16_t *)fo.lab16, (const uint32_t *)lvl, (const uint32_t *)fo.tile_base, (const uint32_t *)fo.tile_count, w, h, dtx,
              dnt, skip, fl);
  else
    RD_LAUNCH("fill.finalize", (k_finalize16<T, false>), dim3(listed ? lists->n[2] : xcd_grid(dnt)), dim3(NTHR), 0, s, d_z,
              (const uint16_t *)fo.lab16, (const uint32_t *)lvl, (const uint32_t *)fo.tile_base, (const uint32_t *)fo.tile_count, w, h, dtx,
              dnt, skip, fl);
  return true;
}

template <class T>
static void fill_device(T *d_z, int w, int h, int topology, hipStream_t s) {
  check_fill_args(d_z, w, h, topology);
  if (topology == 8 ? fill_fused<T, 8>(d_z, w, h, s) : fill_fused<T, 4>(d_z, w, h, s)) return;
  FillBuffers fb;
  BufAlloc ws_alloc{false, nullptr};
  if (topology == 8) fill_local_phase<T, 8>(d_z, w, h, 0, 0, ws_alloc, fb, s);
  else fill_local_phase<T, 4>(d_z, w, h, 0, 0, ws_alloc, fb, s);
  fill_finalize<T>(d_z, w, h, fb, s);
}

// The D8 fill with interior outlets (cells flagged in d_outlet drain like border cells).
template <class T>
static void fill_outlets_device(T *d_z, const uint8_t *d_outlet, const uint8_t *d_skip, int w, int h, hipStream_t s,
                                const SparseLists *lists = nullptr) {
  check_fill_args(d_z, w, h, 8);
  if (!d_outlet) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_outlets: null outlet mask");
  if (w <= 2 || h <= 2) return;   // every cell is a border cell
  if (fill_fused<T, 8>(d_z, w, h, s, d_outlet, d_skip, lists)) return;
  FillBuffers fb;   // (the compact labels ran out of table space, or are switched off: the classic path)
  BufAlloc ws_alloc{false, nullptr};
  fill_local_phase<T, 8>(d_z, w, h, 0, 0, ws_alloc, fb, s, d_outlet);
  fill_finalize<T>(d_z, w, h, fb, s);
}

template <class T>
static void fill_host(T *dem, int w, int h, int topology) {
  check_fill_args(dem, w, h, topology);
  const size_t bytes = (size_t)w * h * sizeof(T);
  T *d = Workspace::get().buf<T>("host.dem", (size_t)w * h);
  RD_HIP(hipMemcpy(d, dem, bytes, hipMemcpyHostToDevice));
  fill_device<T>(d, w, h, topology, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, bytes, hipMemcpyDeviceToHost));
}

}  // namespace rdgpu

// ------------------------------------------------------------------------------------------
// shard handle (C-ABI object)
// ------------------------------------------------------------------------------------------
struct rdgpu_fill_shard {
  int dtype = 0;   // 0 u8, 1 i16, 2 u16, 3 i32, 4 u32, 5 f32, 6 i8
  void *d_dem = nullptr;
  int w = 0, h = 0, topology = 8, open_top = 0, open_bottom = 0;
  hipStream_t stream = nullptr;
  rdgpu::FillBuffers fb;
  std::vector<void *> owned;
  bool cached = false;   // its buffers are the process-wide cached shard workspace (see shard_begin)
  int device = 0;
  uint32_t *d_edges = nullptr;
  uint32_t nedges = 0;
  rdgpu_fill_stats stats{};
};

namespace rdgpu {

template <class T>
struct DtypeCode;
template <> struct DtypeCode<uint8_t> { static constexpr int v = 0; };
template <> struct DtypeCode<int16_t> { static constexpr int v = 1; };
template <> struct DtypeCode<uint16_t> { static constexpr int v = 2; };
template <> struct DtypeCode<int32_t> { static constexpr int v = 3; };
template <> struct DtypeCode<uint32_t> { static constexpr int v = 4; };
template <> struct DtypeCode<float> { static constexpr int v = 5; };
template <> struct DtypeCode<int8_t> { static constexpr int v = 6; };

// One shard per process is the multi-GPU case (one rank, one GPU, one row block), and there a fill must not pay
// hipMalloc / hipFree of its ~5 B/cell of tables on every call: the first live shard keeps its buffers in the
// grow-only workspace (under names of their own); shards begun while it is alive own theirs (tests and tools
// drive many shards from one process).
static bool g_cached_shard_live[64] = {};   // per device: the workspace slots are per device too

static void shard_free(rdgpu_fill_shard *sh) {
  if (!sh) return;
  std::lock_guard<std::recursive_mutex> lock(api_mutex());
  for (void *p : sh->owned) (void)hipFree(p);
  if (sh->cached) {
    g_cached_shard_live[sh->device & 63] = false;
    Workspace::get().unpin();
  }
  delete sh;
}

template <class T, int TOPO>
static void shard_edges(rdgpu_fill_shard *sh) {
  const int w = sh->w, h = sh->h;
  hipStream_t s = sh->stream;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  // adjacent watershed pairs form a planar graph over <= 2w terminals + the outside
  uint32_t hsize = 1024;
  while (hsize < 16u * (uint32_t)w) hsize <<= 1;
  for (;;) {
    unsigned long long *hkeys = ws.buf<unsigned long long>("shard.hkeys", hsize);
    uint32_t *hvals = ws.buf<uint32_t>("shard.hvals", hsize);
    uint32_t *ctr = ws.buf<uint32_t>("shard.ctr", 4);
    RD_HIP(hipMemsetAsync(hkeys, 0, (size_t)hsize * 8, s));
    RD_HIP(hipMemsetAsync(hvals, 0xFF, (size_t)hsize * 4, s));
    RD_HIP(hipMemsetAsync(ctr, 0, 16, s));
    const uint32_t etx = cdiv(w, EW), ent = etx * cdiv(h, EH);
    if (sh->fb.compact)
      RD_LAUNCH("shard.edges", (k_shard_edges16<T, TOPO>), dim3(xcd_grid(ent)), dim3(NTHR), 0, s, (const T *)sh->d_dem,
                (const uint16_t *)sh->fb.lab16, (const uint32_t *)sh->fb.tile_base, cdiv(w, DW), (const uint32_t *)sh->fb.nodeW,
                (const uint32_t *)sh->fb.lvl, w, h, hkeys, hvals, hsize - 1, ctr, etx, ent);
    else
      RD_LAUNCH("shard.edges", (k_shard_edges<T, TOPO>), dim3(xcd_grid(ent)), dim3(NTHR), 0, s,
                (const T *)sh->d_dem, (const uint32_t *)sh->fb.lab, (const uint32_t *)sh->fb.cur,
                (const uint32_t *)sh->fb.acc, (const uint32_t *)sh->fb.tid, sh->fb.B, w, h, hkeys, hvals, hsize - 1, ctr, etx,
                ent);
    uint32_t *edges = (uint32_t *)nullptr;
    if (sh->cached) {
      edges = ws.buf<uint32_t>("shard.edges", (size_t)hsize * 3);
    } else {
      void *p = nullptr;
      RD_HIP(hipMalloc(&p, (size_t)hsize * 12));
      sh->owned.push_back(p);
      edges = (uint32_t *)p;
    }
    RD_LAUNCH("shard.edges_compact", k_shard_edges_compact, dim3(cdiv(hsize, NTHR)), dim3(NTHR), 0, s,
              (const unsigned long long *)hkeys, (const uint32_t *)hvals, hsize, edges, ctr + 1);
    RD_HIP(hipMemcpyAsync(hw, ctr, 8, hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    if (hw[0] == 0 && hw[1] * 2 <= hsize) {
      sh->d_edges = edges;
      sh->nedges = hw[1];
      return;
    }
    if (hsize >= (1u << 28)) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_shard: watershed edge table overflow");
    hsize <<= 2;   // too full (or overflowed): retry with a larger table
  }
}

template <class T>
static rdgpu_fill_shard *shard_begin(T *d_dem, int w, int h, int topology, int open_top, int open_bottom,
                                     hipStream_t s) {
  check_fill_args(d_dem, w, h, topology);
  if ((open_top || open_bottom) && h < 2) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard: a shard needs at least 2 rows");
  rdgpu_fill_shard *sh = new rdgpu_fill_shard();
  try {
    sh->dtype = DtypeCode<T>::v;
    sh->d_dem = d_dem;
    sh->w = w; sh->h = h; sh->topology = topology;
    sh->open_top = open_top ? 1 : 0; sh->open_bottom = open_bottom ? 1 : 0;
    sh->stream = s;
    RD_HIP(hipGetDevice(&sh->device));
    sh->cached = !g_cached_shard_live[sh->device & 63];
    if (sh->cached) {
      g_cached_shard_live[sh->device & 63] = true;
      Workspace::get().pin();   // (its tables live in the workspace: release_workspace() is refused while it is alive)
    }
    BufAlloc alloc{!sh->cached, &sh->owned, sh->cached};
    // the local phase on compact labels (r04: the whole-raster engine with the cut rows as frozen terminals); the classic
    // 32-bit-label phase where that does not apply (RDGPU_SHARD_FUSED=0, tiny blocks, tables that do not fit)
    const char *sf = getenv("RDGPU_SHARD_FUSED");
    bool done = false;
    if (!(sf && sf[0] == '0') && !sh->open_top && !sh->open_bottom) {
      // a block without a cut is the whole raster (one rank): the plain compact-label fill, raised at once -- nothing is
      // left for the exchange or for finish
      done = topology == 8 ? fill_fused<T, 8>(d_dem, w, h, s) : fill_fused<T, 4>(d_dem, w, h, s);
      if (done) { sh->fb = FillBuffers(); sh->fb.trivial = true; }
    } else if (!(sf && sf[0] == '0')) {
      const size_t owned_before = sh->owned.size();
      done = topology == 8 ? fill_fused<T, 8>(d_dem, w, h, s, nullptr, nullptr, nullptr, &sh->fb, &alloc, sh->open_top, sh->open_bottom)
                           : fill_fused<T, 4>(d_dem, w, h, s, nullptr, nullptr, nullptr, &sh->fb, &alloc, sh->open_top, sh->open_bottom);
      if (!done) {
        // the attempt gave up (node table or pair list too small: white noise): what it allocated for itself goes back before
        // the classic phase allocates its own tables, or the shard would hold both until shard_free; the
        // half-filled FillBuffers is reset as well
        RD_HIP(hipStreamSynchronize(s));
        for (size_t i = owned_before; i < sh->owned.size(); i++) (void)hipFree(sh->owned[i]);
        sh->owned.resize(owned_before);
        sh->fb = FillBuffers();
      }
    }
    if (!done) {
      if (topology == 8) fill_local_phase<T, 8>(d_dem, w, h, sh->open_top, sh->open_bottom, alloc, sh->fb, s);
      else fill_local_phase<T, 4>(d_dem, w, h, sh->open_top, sh->open_bottom, alloc, sh->fb, s);
    }
    sh->stats = g_stats;
    if (!sh->fb.trivial && sh->fb.tid) {
      if (topology == 8) shard_edges<T, 8>(sh);
      else shard_edges<T, 4>(sh);
    }
  } catch (...) {
    shard_free(sh);
    throw;
  }
  return sh;
}

template <class T>
static void shard_export_rows(rdgpu_fill_shard *sh, uint32_t *top_keys, uint32_t *bottom_keys) {
  const int w = sh->w, h = sh->h;
  std::vector<T> row(w);
  for (int which = 0; which < 2; which++) {
    uint32_t *out = which == 0 ? top_keys : bottom_keys;
    if (!out) continue;
    const T *src = (const T *)sh->d_dem + (which == 0 ? 0 : (size_t)(h - 1) * w);
    RD_HIP(hipMemcpyAsync(row.data(), src, (size_t)w * sizeof(T), hipMemcpyDeviceToHost, sh->stream));
    RD_HIP(hipStreamSynchronize(sh->stream));
    for (int x = 0; x < w; x++) out[x] = Key32<T>::to(row[x]);
  }
}

template <class T>
static void shard_finish(rdgpu_fill_shard *sh, const uint32_t *levels2w) {
  hipStream_t s = sh->stream;
  if (!sh->fb.trivial) {
    if (sh->fb.tid) {
      if (!levels2w) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_finish: levels required for a shard with cut rows");
      uint32_t *dl = Workspace::get().buf<uint32_t>("shard.levels", (size_t)2 * sh->w);
      RD_HIP(hipMemcpyAsync(dl, levels2w, (size_t)2 * sh->w * 4, hipMemcpyHostToDevice, s));
      RD_LAUNCH("shard.apply", k_shard_apply, dim3(cdiv(sh->fb.B, NTHR)), dim3(NTHR), 0, s,
                (const uint32_t *)sh->fb.cur, sh->fb.acc, (const uint32_t *)sh->fb.tid, (const uint32_t *)dl, sh->fb.B);
    }
    if (sh->fb.compact) fill_finalize16<T>((T *)sh->d_dem, sh->w, sh->h, sh->fb, s);
    else fill_finalize<T>((T *)sh->d_dem, sh->w, sh->h, sh->fb, s);
  }
  RD_HIP(hipStreamSynchronize(s));   // levels2w is caller memory; buffers are freed next
}

// ------------------------------------------------------------------------------------------
// 7. GPU solve of the joined shard graph (device-resident variant of shardgraph.hip's host solve)
// The label graph (cut-row terminals + the outside, watershed edges of every shard, edges across every
// cut) is contracted with the SAME Boruvka machinery as the raster: only the scan differs -- it
// enumerates edges instead of cells.  A 105 ms host Priority-Flood at 8 shards becomes ~1 ms.
//   keys_all  [S][2][w]   cut-row keys of every shard
//   edges_all [S][cap][3] watershed edges of every shard (a, b, pass), counts[S] valid triples each
// Nodes: s*2w + tid; NOUT = S*2w is the outside.
// ------------------------------------------------------------------------------------------
struct GraphDesc {
  const uint32_t *keys_all, *edges_all, *counts;
  uint32_t S, w, cap, topo;
};

__device__ __forceinline__ bool graph_edge(const GraphDesc &g, uint64_t idx, uint32_t &a, uint32_t &b, uint32_t &pass) {
  const uint32_t per = 2u * g.w, NOUT = g.S * per;
  const uint64_t nintra = (uint64_t)g.S * g.cap;
  if (idx < nintra) {
    const uint32_t s = (uint32_t)(idx / g.cap), e = (uint32_t)(idx % g.cap);
    if (e >= g.counts[s]) return false;
    const uint32_t *t = g.edges_all + ((size_t)s * g.cap + e) * 3;
    a = t[0] == NO_TID ? NOUT : s * per + t[0];
    b = t[1] == NO_TID ? NOUT : s * per + t[1];
    pass = t[2];
    return true;
  }
  // edges across the cut between shard s (bottom row) and s+1 (top row), HandleEdge main.cpp:344-378
  idx -= nintra;
  const uint32_t s = (uint32_t)(idx / (3ull * g.w));
  if (s + 1 >= g.S) return false;
  const uint32_t r = (uint32_t)(idx % (3ull * g.w));
  const int x = (int)(r / 3), dx = (int)(r % 3) - 1;
  if (g.topo == 4 && dx != 0) return false;
  const int x2 = x + dx;
  if (x2 < 0 || x2 >= (int)g.w) return false;
  const bool sa = x == 0 || x == (int)g.w - 1, sb = x2 == 0 || x2 == (int)g.w - 1;   // side columns: true border
  if (sa && sb) return false;
  a = sa ? NOUT : s * per + g.w + (uint32_t)x;
  b = sb ? NOUT : (s + 1) * per + (uint32_t)x2;
  const uint32_t ka = g.keys_all[((size_t)s * 2 + 1) * g.w + x], kb = g.keys_all[((size_t)(s + 1) * 2) * g.w + x2];
  pass = ka > kb ? ka : kb;
  return true;
}

// a shard's edge count beyond the capacity of the gathered payload: its edges were not sent (sharded.py, one exchange)
__global__ void k_graph_check_counts(const uint32_t *__restrict__ counts, uint32_t S, uint32_t cap, uint32_t *flag) {
  for (uint32_t s = threadIdx.x; s < S; s += blockDim.x)
    if (counts[s] > cap) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(NTHR) void k_graph_touch(GraphDesc g, uint64_t nslots, uint32_t *closed) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < nslots; i += stride) {
    uint32_t a, b, p;
    if (!graph_edge(g, i, a, b, p)) continue;
    closed[a] = NO_TID;   // NO_TID = "open" in k_init_tables' convention
    closed[b] = NO_TID;
  }
}

__global__ __launch_bounds__(NTHR) void k_graph_scan(GraphDesc g, uint64_t nslots, const uint32_t *__restrict__ cur,
                                                     unsigned long long *best) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < nslots; i += stride) {
    uint32_t a, b, p;
    if (!graph_edge(g, i, a, b, p)) continue;
    const uint32_t ca = cur[a], cb = cur[b];
    if (ca == cb) continue;
    if (!(ca & CLOSED)) {
      const unsigned long long e = ((unsigned long long)p << 32) | cb;
      if (e < best[ca]) atomicMin(&best[ca], e);
    }
    if (!(cb & CLOSED)) {
      const unsigned long long e = ((unsigned long long)p << 32) | ca;
      if (e < best[cb]) atomicMin(&best[cb], e);
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_graph_levels(const uint32_t *__restrict__ cur, const uint32_t *__restrict__ acc,
                                                       const uint32_t *__restrict__ closed, uint32_t *levels, uint32_t NOUT) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= NOUT) return;
  levels[i] = closed[i] == NO_TID ? acc[i] : 0u;   // untouched ids (non-cut rows, side columns) -> 0
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_rows_to_keys(const T *__restrict__ top, const T *__restrict__ bottom,
                                                       uint32_t *keys, int w) {
  const int x = blockIdx.x * NTHR + threadIdx.x;
  if (x >= w) return;
  keys[x] = Key32<T>::to(top[x]);
  keys[w + x] = Key32<T>::to(bottom[x]);
}

static void graph_solve_device(int S, int w, int topo, const uint32_t *d_keys_all, const uint32_t *d_edges_all,
                               const uint32_t *d_counts, uint32_t cap, uint32_t *d_levels_all, hipStream_t s) {
  if (S < 1 || w < 1 || !d_keys_all || !d_levels_all) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_graph_solve_dev: bad arguments");
  const uint32_t per = 2u * (uint32_t)w, NOUT = (uint32_t)S * per;
  RD_HIP(hipMemsetAsync(d_levels_all, 0, (size_t)NOUT * 4, s));
  if (S == 1) return;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  GraphDesc g{d_keys_all, d_edges_all, d_counts, (uint32_t)S, (uint32_t)w, cap, (uint32_t)topo};
  const uint64_t nslots = (uint64_t)S * cap + (uint64_t)(S - 1) * 3ull * (uint64_t)w;
  const uint32_t B = NOUT;   // "basins" = graph nodes, index B = the outside
  uint32_t *cur = ws.buf<uint32_t>("graph.cur", (size_t)B + 1);
  uint32_t *acc = ws.buf<uint32_t>("graph.acc", (size_t)B + 1);
  uint32_t *closed = ws.buf<uint32_t>("graph.closed", (size_t)B + 1);
  unsigned long long *best = ws.buf<unsigned long long>("graph.best", (size_t)B + 1);
  unsigned long long *link = ws.buf<unsigned long long>("graph.link", (size_t)B + 1);
  uint32_t *rootsA = ws.buf<uint32_t>("graph.rootsA", B), *rootsB = ws.buf<uint32_t>("graph.rootsB", B);
  uint32_t *dflags = ws.buf<uint32_t>("graph.flags", 16);
  const uint32_t egrid = (uint32_t)std::min<uint64_t>((nslots + NTHR - 1) / NTHR, 256u * 16u);
  RD_HIP(hipMemsetAsync(closed, 0, ((size_t)B + 1) * 4, s));   // 0 = closed (isolated) until an edge touches it
  RD_LAUNCH("graph.touch", k_graph_touch, dim3(egrid), dim3(NTHR), 0, s, g, nslots, closed);
  RD_HIP(hipMemsetAsync(dflags, 0, 16 * 4, s));
  if (d_counts) hipLaunchKernelGGL(k_graph_check_counts, dim3(1), dim3(64), 0, s, d_counts, (uint32_t)S, cap, dflags + 3);
  RD_LAUNCH("graph.init_tables", k_init_tables, dim3(cdiv((uint64_t)B + 1, NTHR)), dim3(NTHR), 0, s, cur, acc, link, rootsA,
            dflags + 2, (const uint32_t *)closed, B);
  RD_HIP(hipMemcpyAsync(hw, dflags + 2, 8, hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  if (hw[1] != 0) throw Error(RDGPU_ERR_CAPACITY, "rdgpu_fill_graph_solve_dev: a shard holds more edges than the gathered payload has room for");
  const uint32_t nroots0 = hw[0];
  // r06: the rounds as in fill_fused -- counts on the device (rc[4 r] = live roots entering round r), grid-stride round kernels,
  // a batch of rounds per synchronisation (r05: two to three synchronisations PER ROUND; every rank of a sharded fill runs this
  // solve, so its host round trips are on every rank's critical path)
  constexpr int MAXR = 40;
  uint32_t *rc = ws.buf<uint32_t>("graph.round_counts", 4 * (MAXR + 2));
  RD_HIP(hipMemsetAsync(rc, 0, 4 * (MAXR + 2) * sizeof(uint32_t), s));
  hw[220] = nroots0;
  RD_HIP(hipMemcpyAsync(rc + 4, hw + 220, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  uint32_t *hrc = hw + 16;
  int rdone = 0;
  uint32_t lg = 0;
  while ((1ull << lg) < (unsigned long long)nroots0 + 1ull) lg++;
  int batch = std::min(MAXR, std::max(4, (int)(2 * lg + 4) / 5 + 2));
  uint32_t last_live = 0xFFFFFFFFu;
  while (nroots0 > 0) {
    for (int k = 0; k < batch && rdone < MAXR; k++) {
      const int r = ++rdone;
      const uint32_t bound = std::max(1u, r - 1 < 31 ? nroots0 >> (r - 1) : 1u);
      const uint32_t rgrid = std::min(2048u, cdiv(bound, NTHR));
      const uint32_t *nr = rc + 4 * r;
      RD_LAUNCH("graph.best_reset", k_best_reset, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, best, nr);
      RD_LAUNCH("graph.scan", k_graph_scan, dim3(egrid), dim3(NTHR), 0, s, g, nslots, (const uint32_t *)cur, best);
      RD_LAUNCH("graph.hook", k_hook, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, best, link, nr);
      RD_LAUNCH("graph.chase_links", k_chase_links, dim3(rgrid), dim3(NTHR), 0, s, rootsA, 0u, link, 1 << 14, dflags, nr);
      RD_LAUNCH("graph.update", k_update_basins, dim3(cdiv(B, NTHR)), dim3(NTHR), 0, s, cur, acc, link, B, nr);
      RD_LAUNCH("graph.compact_roots", k_compact_roots, dim3(std::min(2048u, cdiv(bound, NTHR * RPT))), dim3(NTHR), 0, s, rootsA, 0u, link,
                rootsB, rc + 4 * (r + 1), nr);
      std::swap(rootsA, rootsB);
    }
    RD_HIP(hipMemcpyAsync(hrc, rc, 4 * (MAXR + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipMemcpyAsync(hw, dflags, 4, hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    if (hw[0] != 0) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_graph_solve_dev: a chain of hooks longer than 16384 (internal error)");
    const uint32_t live = hrc[4 * (rdone + 1)];
    if (live == 0) break;
    if (live >= last_live || rdone >= MAXR)
      throw Error(RDGPU_ERR_HIP, "rdgpu_fill_graph_solve_dev: a cut-row terminal is not connected to the outside");
    last_live = live;
    batch = 4;
  }
  RD_LAUNCH("graph.levels", k_graph_levels, dim3(cdiv(NOUT, NTHR)), dim3(NTHR), 0, s, (const uint32_t *)cur,
            (const uint32_t *)acc, (const uint32_t *)closed, d_levels_all, NOUT);
}

template <class T>
static void shard_export_dev(rdgpu_fill_shard *sh, uint32_t *d_keys, uint32_t *d_edges, uint32_t cap) {
  const int w = sh->w, h = sh->h;
  const T *base = (const T *)sh->d_dem;
  RD_LAUNCH("shard.rows_to_keys", (k_rows_to_keys<T>), dim3(cdiv(w, NTHR)), dim3(NTHR), 0, sh->stream, base,
            base + (size_t)(h - 1) * w, d_keys, w);
  if (sh->nedges) {
    if (!d_edges || cap < sh->nedges) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_export_dev: edge buffer too small");
    RD_HIP(hipMemcpyAsync(d_edges, sh->d_edges, (size_t)sh->nedges * 12, hipMemcpyDeviceToDevice, sh->stream));
  }
}

template <class T>
static void shard_finish_dev(rdgpu_fill_shard *sh, const uint32_t *d_levels2w) {
  hipStream_t s = sh->stream;
  if (!sh->fb.trivial) {
    if (sh->fb.tid) {
      if (!d_levels2w) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_finish_dev: levels required for a shard with cut rows");
      RD_LAUNCH("shard.apply", k_shard_apply, dim3(cdiv(sh->fb.B, NTHR)), dim3(NTHR), 0, s, (const uint32_t *)sh->fb.cur,
                sh->fb.acc, (const uint32_t *)sh->fb.tid, d_levels2w, sh->fb.B);
    }
    if (sh->fb.compact) fill_finalize16<T>((T *)sh->d_dem, sh->w, sh->h, sh->fb, s);
    else fill_finalize<T>((T *)sh->d_dem, sh->w, sh->h, sh->fb, s);
  }
  RD_HIP(hipStreamSynchronize(s));   // the handle's buffers are freed next
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FILL_API(SUF, T)                                                                       \
  extern "C" int rdgpu_fill_multi_##SUF(T *, int, int, int, const int *, int);                    \
  extern "C" int rdgpu_fill_##SUF(T *dem, int w, int h, int topology) {                           \
    const std::vector<int> devs = env_devices();                                                  \
    if (devs.size() > 1 && h >= 2 * (int)devs.size())                                             \
      return rdgpu_fill_multi_##SUF(dem, w, h, topology, devs.data(), (int)devs.size());          \
    return guarded([&] { fill_host<T>(dem, w, h, topology); });                                   \
  }                                                                                               \
  extern "C" int rdgpu_fill_outlets_dev_##SUF(T *d_dem, const uint8_t *d_outlet, int w, int h, void *stream) { \
    return rdgpu::guarded([&] { rdgpu::fill_outlets_device<T>(d_dem, d_outlet, nullptr, w, h, (hipStream_t)stream); }); \
  }                                                                                                \
  extern "C" int rdgpu_fill_outlets_skip_dev_##SUF(T *d_dem, const uint8_t *d_outlet, const uint8_t *d_skip, int w, int h, \
                                                   void *stream) {                                 \
    return rdgpu::guarded([&] { rdgpu::fill_outlets_device<T>(d_dem, d_outlet, d_skip, w, h, (hipStream_t)stream); }); \
  }                                                                                                \
  extern "C" int rdgpu_fill_outlets_lists_dev_##SUF(T *d_dem, const uint8_t *d_outlet, const uint8_t *d_skip,             \
                                                    const uint32_t *d_lists, uint32_t stride, const uint32_t *counts3, int w, \
                                                    int h, void *stream) {                         \
    return rdgpu::guarded([&] {                                                                    \
      rdgpu::SparseLists l;                                                                        \
      l.d = d_lists; l.stride = stride;                                                            \
      for (int k = 0; k < 3; k++) l.n[k] = counts3[k];                                             \
      rdgpu::fill_outlets_device<T>(d_dem, d_outlet, d_skip, w, h, (hipStream_t)stream, &l);       \
    });                                                                                            \
  }                                                                                                \
  extern "C" int rdgpu_fill_dev_##SUF(T *d_dem, int w, int h, int topology, void *stream) {       \
    return guarded([&] { fill_device<T>(d_dem, w, h, topology, (hipStream_t)stream); });          \
  }                                                                                               \
  extern "C" int rdgpu_fill_max_dep_##SUF(T *dem, int w, int h, int topology, uint64_t max_dep_size) { \
    return guarded([&] { fill_max_dep_host<T>(dem, w, h, topology, max_dep_size); });             \
  }                                                                                               \
  extern "C" int rdgpu_fill_max_dep_dev_##SUF(T *d_dem, int w, int h, int topology, uint64_t max_dep_size, void *stream) { \
    return guarded([&] { fill_max_dep_device<T>(d_dem, w, h, topology, max_dep_size, (hipStream_t)stream); }); \
  }                                                                                               \
  extern "C" int rdgpu_fill_max_dep_ties_dev_##SUF(T *d_dem, int w, int h, int topology, uint64_t max_dep_size,            \
                                                   uint8_t *d_tie_mask, void *stream) {                                  \
    return guarded([&] { fill_max_dep_device<T>(d_dem, w, h, topology, max_dep_size, (hipStream_t)stream, d_tie_mask); }); \
  }                                                                                               \
  extern "C" int rdgpu_pit_mask_##SUF(const T *dem, T nodata, int w, int h, int topology, uint8_t *mask) { \
    return guarded([&] { pit_mask_host<T>(dem, nodata, w, h, topology, mask); });                 \
  }                                                                                               \
  extern "C" int rdgpu_pit_mask_dev_##SUF(const T *d_dem, T nodata, int w, int h, int topology, uint8_t *d_mask, \
                                          void *stream) {                                         \
    return guarded([&] { pit_mask_device<T>(d_dem, nodata, w, h, topology, d_mask, (hipStream_t)stream); }); \
  }                                                                                               \
  extern "C" int rdgpu_fill_shard_begin_##SUF(T *d_dem, int w, int h, int topology, int open_top, \
                                              int open_bottom, void *stream, rdgpu_fill_shard **out) { \
    return guarded([&] {                                                                          \
      if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_begin: null output handle");         \
      *out = shard_begin<T>(d_dem, w, h, topology, open_top, open_bottom, (hipStream_t)stream);   \
    });                                                                                           \
  }
RD_FILL_API(u8, uint8_t)
RD_FILL_API(i16, int16_t)
RD_FILL_API(u16, uint16_t)
RD_FILL_API(i32, int32_t)
RD_FILL_API(u32, uint32_t)
RD_FILL_API(f32, float)
RD_FILL_API(i8, int8_t)

#define RD_DISPATCH(sh, CALL)                                                  \
  switch ((sh)->dtype) {                                                       \
    case 0: { using T = uint8_t; CALL; } break;                                \
    case 1: { using T = int16_t; CALL; } break;                                \
    case 2: { using T = uint16_t; CALL; } break;                               \
    case 3: { using T = int32_t; CALL; } break;                                \
    case 4: { using T = uint32_t; CALL; } break;                               \
    case 6: { using T = int8_t; CALL; } break;                                 \
    default: { using T = float; CALL; } break;                                 \
  }

extern "C" int rdgpu_fill_shard_edge_count(rdgpu_fill_shard *sh, uint32_t *n_edges) {
  return guarded([&] {
    if (!sh || !n_edges) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_edge_count: null pointer");
    *n_edges = sh->nedges;
  });
}

extern "C" int rdgpu_fill_shard_export(rdgpu_fill_shard *sh, uint32_t *top_keys, uint32_t *bottom_keys,
                                       uint32_t *edges) {
  return guarded([&] {
    if (!sh) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_export: null handle");
    RD_DISPATCH(sh, shard_export_rows<T>(sh, top_keys, bottom_keys));
    if (edges && sh->nedges) {
      RD_HIP(hipMemcpyAsync(edges, sh->d_edges, (size_t)sh->nedges * 12, hipMemcpyDeviceToHost, sh->stream));
      RD_HIP(hipStreamSynchronize(sh->stream));
    }
  });
}

extern "C" int rdgpu_fill_shard_finish(rdgpu_fill_shard *sh, const uint32_t *levels) {
  if (!sh) { set_last_error("rdgpu_fill_shard_finish: null handle"); return RDGPU_ERR_ARG; }
  std::lock_guard<std::recursive_mutex> lock(api_mutex());
  const int rc = guarded([&] { RD_DISPATCH(sh, shard_finish<T>(sh, levels)); });
  g_stats = sh->stats;
  shard_free(sh);
  return rc;
}

extern "C" int rdgpu_fill_shard_export_dev(rdgpu_fill_shard *sh, uint32_t *d_keys, uint32_t *d_edges, uint32_t cap) {
  return guarded([&] {
    if (!sh || !d_keys) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_shard_export_dev: null pointer");
    RD_DISPATCH(sh, shard_export_dev<T>(sh, d_keys, d_edges, cap));
  });
}

extern "C" int rdgpu_fill_shard_finish_dev(rdgpu_fill_shard *sh, const uint32_t *d_levels) {
  if (!sh) { set_last_error("rdgpu_fill_shard_finish_dev: null handle"); return RDGPU_ERR_ARG; }
  std::lock_guard<std::recursive_mutex> lock(api_mutex());
  const int rc = guarded([&] { RD_DISPATCH(sh, shard_finish_dev<T>(sh, d_levels)); });
  g_stats = sh->stats;
  shard_free(sh);
  return rc;
}

extern "C" int rdgpu_fill_graph_solve_dev(int nshards, int width, int topology, const uint32_t *d_keys_all,
                                          const uint32_t *d_edges_all, const uint32_t *d_counts, uint32_t cap,
                                          uint32_t *d_levels_all, void *stream) {
  return guarded([&] {
    if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_graph_solve_dev: topology must be 8 or 4");
    graph_solve_device(nshards, width, topology, d_keys_all, d_edges_all, d_counts, cap, d_levels_all, (hipStream_t)stream);
  });
}

extern "C" int rdgpu_fill_shard_free(rdgpu_fill_shard *sh) {
  shard_free(sh);
  return RDGPU_OK;
}

extern "C" int rdgpu_fill_max_dep_get_stats(rdgpu_max_dep_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  if (g_md_pending) {
    if (hipStreamSynchronize(g_md_stream) != hipSuccess) { set_last_error("rdgpu_fill_max_dep_get_stats: hipStreamSynchronize failed"); return RDGPU_ERR_HIP; }
    g_md_stats = rdgpu_max_dep_stats{g_md_pinned[0], g_md_pinned[1], g_md_pinned[2], g_md_pinned[3]};
    g_md_pending = false;
  }
  *out = g_md_stats;
  return RDGPU_OK;
}

extern "C" int rdgpu_fill_get_stats(rdgpu_fill_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = g_stats;
  return RDGPU_OK;
}
