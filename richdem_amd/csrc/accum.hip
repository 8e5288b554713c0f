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

#include <string>
#include <type_traits>
#include "flowdirs.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace rdgpu {

constexpr int NTHR = 256;
constexpr unsigned long long CNT1 = 1ull << 56;
constexpr unsigned long long LOWMASK = CNT1 - 1ull;
constexpr unsigned long long SRC = 0xFFull;        // count-field marker of a source cell (unit path)
// single-block unit path: the cell's own direction rides in bits 52-55 of its word, so the returning add of a
// step also delivers the direction of the cell just completed -- ONE dependent memory operation per step
constexpr int DIRSHIFT = 52;
constexpr unsigned long long AREAMASK = (1ull << DIRSHIFT) - 1ull;
// f64 path: pending word = (inflows in total << 12) | (own direction << 8) | count of inflows still to arrive; count byte 0xFE marks a
// source, 0xF0 a NoData cell (at most 8 arrivals ever decrement it: it never reads as 1 or as a source)
constexpr uint32_t SRC32 = 0xFEu;
constexpr uint32_t NODATA32 = 0xF0u;

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

// Stage a (T+2) x (T+2) window of the direction raster in LDS: every load of the thread is issued before the first one is
// consumed (load -> store -> load -> ... paid one memory round trip per loop trip, seventeen in a row: it was most of
// the time of every tile kernel in this file).  Cells outside the raster read as `fill`.
template <int LW_, int NT_>
__device__ __forceinline__ void stage_dirs(const uint8_t *__restrict__ dirs, int w, int h, int x0, int y0, uint8_t fill,
                                           uint8_t *sd) {
  constexpr int N = LW_ * LW_, IPT = (N + NT_ - 1) / NT_;
  uint8_t v[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {
    const int i = min((int)threadIdx.x + r * NT_, N - 1);
    const int ly = i / LW_, lx = i - ly * LW_;
    const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);   // clamped: branch-free loads
    v[r] = dirs[(size_t)gy * w + gx];
  }
#pragma unroll
  for (int r = 0; r < IPT; r++) {
    const int i = (int)threadIdx.x + r * NT_;
    if (i >= N) continue;
    const int ly = i / LW_, lx = i - ly * LW_;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    sd[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? v[r] : fill;
  }
}

// ---- unit weights ---------------------------------------------------------------------------
// Raster-wide walk with lane refill.  One thread per source would leave a wavefront waiting for its single
// longest chain (mean chain ~3 steps, the longest per wave hundreds: 790 ms at 40k x 40k, ~1% lane use).
// Instead each wavefront owns a chunk of cells; a lane whose chain has ended immediately scans the
// chunk for the next source, so every loop trip advances up to 64 chains by one step.
constexpr uint32_t WALK_CHUNK = 8192;   // cells per wavefront
__global__ __launch_bounds__(NTHR) void k_acc_walk_unit(const uint8_t *__restrict__ dirs, uint8_t nodata,
                                                        unsigned long long *word, int w, int h) {
  const uint64_t n = (uint64_t)w * h;
  const uint64_t wave = ((uint64_t)blockIdx.x * NTHR + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t next = wave * WALK_CHUNK;
  const uint64_t end = next + WALK_CHUNK < n ? next + WALK_CHUNK : n;
  if (next >= n) return;
  bool active = false;
  uint32_t c = 0;
  unsigned long long v = 0;
  uint8_t d = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    if (next < end && idle) {
      // idle lanes look at the next cells of the chunk (contiguous -> coalesced)
      const uint64_t my = next + (uint64_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (!active && my < end) {
        const uint8_t dd = dirs[my];
        if (dd != nodata) {
          const unsigned long long wd = word[my];   // a source's word is never modified: plain read
          if ((wd >> 56) == SRC) { active = true; c = (uint32_t)my; v = wd & AREAMASK; d = dd; }
        }
      }
      next += (uint64_t)__popcll(idle);
    } else if (idle == ~0ull) {
      break;   // chunk exhausted and every chain has ended
    }
    if (active) {
      const int64_t t = flow_target(c, d, w, h);                 // d8_methods.hpp:113-122
      if (t < 0) active = false;
      else {
        // A NoData target holds word 0 (count field 0, then 0xFF, 0xFE ... after arrivals): never "1", so the
        // flow into it is dropped (:124-125) without looking its direction up first; k_acc_out ignores its word.
        const unsigned long long old = atomicAdd(&word[t], v - CNT1);
        if ((old >> 56) != 1) active = false;                    // not the last inflow of t
        else { v = (old & AREAMASK) + v; c = (uint32_t)t; d = (uint8_t)((old >> DIRSHIFT) & 15u); }   // t's final area: keep walking
      }
    }
  }
}

// Tile pre-walk: the same last-arriver walk, but confined to one 64x64 tile with the words in LDS
// ((pending << 24) | area in 32 bits).  Global device atomics are the bottleneck of the raster-wide walk
// (~2*10^9/s); here only cells whose 8 neighbours all lie in the tile (everything but the tile's outer ring)
// are ever the TARGET of an add, so their state is private to the block and LDS atomics suffice.  A walk
// stops when its next target is a ring cell or outside the tile; the cell it stopped on is written back as
// a "source" carrying its total, and the raster-wide walk (k_acc_walk_unit) continues from there.
constexpr int AW = 64, AH = 64, ALW = AW + 2, ALH = AH + 2;
__global__ __launch_bounds__(NTHR) void k_acc_tile_prewalk(const uint8_t *__restrict__ dirs, uint8_t nodata,
                                                           unsigned long long *__restrict__ word, int w, int h,
                                                           uint32_t tilesX, uint32_t ntiles) {
  __shared__ uint8_t sd[ALH * ALW];
  __shared__ uint32_t lw[AH * AW];
  constexpr uint32_t LCNT1 = 1u << 24, LMASK = LCNT1 - 1u, LSRC = 0xFFu;
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * AW, y0 = (int)(t / tilesX) * AH;
  stage_dirs<ALW, NTHR>(dirs, w, h, x0, y0, nodata, sd);
  __syncthreads();
  const int lx = threadIdx.x & (AW - 1), ly0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave uniform: row arithmetic on the scalar unit)
  // pending inflows (from anywhere) + own area
  for (int j = 0; j < AH / 4; j++) {
    const int ly = ly0 + 4 * j, o = (ly + 1) * ALW + lx + 1;
    uint32_t v = 0;
    if (x0 + lx < w && y0 + ly < h && sd[o] != nodata) {
      int k = 0;
#pragma unroll
      for (int m = 1; m <= 8; m++) {
        const uint8_t d = sd[o + d8dy(m) * ALW + d8dx(m)];   // halo cells outside the raster hold nodata
        if (d != nodata && d == (m <= 4 ? m + 4 : m - 4)) k++;
      }
      v = ((uint32_t)k << 24) | 1u;
    }
    lw[ly * AW + lx] = v;
  }
  __syncthreads();
  // walks from the tile's sources; sources are fixed before any walk (initial pending == 0)
  uint32_t srcmask = 0;
  for (int j = 0; j < AH / 4; j++) {
    const int ly = ly0 + 4 * j;
    if (x0 + lx < w && y0 + ly < h && sd[(ly + 1) * ALW + lx + 1] != nodata && (lw[ly * AW + lx] >> 24) == 0) srcmask |= 1u << j;
  }
  __syncthreads();
  {
    // a lane whose walk has ended takes its next source in the same trip (see k_acc_link_tile)
    uint32_t m = srcmask, v = 0;
    bool active = false;
    int cx = 0, cy = 0;
    for (;;) {
      if (!active && m) {
        cx = lx; cy = ly0 + 4 * (__ffs((int)m) - 1);
        m &= m - 1;
        v = 1;
        active = true;
      }
      if (!__any(active || m != 0)) break;
      if (active) {
        const uint8_t d = sd[(cy + 1) * ALW + cx + 1];
        bool stalled = false, cont = false;
        if (d >= 1 && d <= 8) {
          const int tx = cx + d8dx(d), ty = cy + d8dy(d);
          const uint8_t dt = sd[(ty + 1) * ALW + tx + 1];      // nodata for cells outside the raster
          if (dt != nodata) {                                    // else: off the DEM / into NoData: dropped
            if (tx >= 1 && tx < AW - 1 && ty >= 1 && ty < AH - 1) {
              // target is strictly inside the tile: all its donors are in this tile -> LDS state is complete
              const uint32_t old = atomicAdd(&lw[ty * AW + tx], v - LCNT1);
              if ((old >> 24) == 1) { v = (old & LMASK) + v; cx = tx; cy = ty; cont = true; }
            } else {
              stalled = true;                                    // ring cell or another tile: continue raster-wide
            }
          }
        }
        if (stalled) lw[cy * AW + cx] = (LSRC << 24) | v;        // completed, still has to pass its total on
        active = cont;
      }
    }
  }
  __syncthreads();
  for (int j = 0; j < AH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const uint32_t v = lw[ly * AW + lx];
    unsigned long long g = 0;
    if (sd[(ly + 1) * ALW + lx + 1] != nodata) {
      uint32_t cnt = v >> 24;
      // a true source that passed its total on inside the tile is simply complete (count 0); a true source
      // that could not (first target outside) keeps the SRC marker it would have got from k_acc_init_unit
      const uint8_t dd = sd[(ly + 1) * ALW + lx + 1];
      g = ((unsigned long long)(cnt == LSRC ? SRC : cnt) << 56) | ((unsigned long long)(dd <= 8 ? dd : 0) << DIRSHIFT) |
          (unsigned long long)(v & LMASK);
    }
    word[(size_t)gy * w + gx] = g;
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
    if (std::is_same<A, unsigned long long>::value) {   // a shard's words: the total, an incomplete cell keeps a count
      area[c] = (A)(((cnt != 0 && cnt != SRC) ? CNT1 : 0ull) | (v & AREAMASK));
      continue;
    }
    const unsigned long long a = (cnt != 0 && cnt != SRC) ? (v & AREAMASK) - 1 : (v & AREAMASK);
    area[c] = (A)a;
  }
}

// ---- unit weights through tile links -------------------------------------------------------------------------
// The raster-wide walk above is bound by its critical path: one lane walks each long river cell by cell, one memory
// round trip per cell (55 ms of the 77 at S3, whose resolved lakes have flow paths of tens of thousands of cells).
// The reference's tiled program (programs/parallel_d8_accum/main.cpp: per-tile accumulation :373-464, perimeter links
// FollowPath :270-334, inflow added along the in-tile path FollowPathAdd :344-370; Barnes 2017) shows the way out, here
// at the granularity of 64 x 64 LDS tiles:
//   1. k_acc_link_tile   every tile on its own: accumulation of the flow that STAYS inside the tile (last-arriver walk
//                        in LDS over all of its cells, donors outside the tile ignored) and, by pointer jumping in LDS,
//                        the cell through which every border cell's path leaves the tile.  Only the tile's 252 border
//                        cells are written: the local total of each EXIT cell (a cell whose direction leaves the tile)
//                        and each border cell's exit.
//   2. the exits form a forest of their own: exit q hands its total to the exit that q's target cell (a border cell of
//                        the neighbouring tile) leaves ITS tile through.  k_acc_link_edges builds those links and their
//                        in-degrees, k_acc_link_walk accumulates along them (the same last-arriver walk; its critical
//                        path is the number of TILES a river crosses, not its cells).
//   3. k_acc_link_final  every tile again: the inflow a border cell receives from outside is the finished total of its
//                        outside donors; with that as extra weight the in-tile accumulation is final, written in the
//                        requested type.
// Direction loops keep the reference's semantics (d8_methods.hpp:104-131: a cell that never becomes free of
// dependencies never passes anything on and keeps the inflow that did arrive, without its own 1): an exit that is not
// complete inside its tile is blocked for good, and a border cell with an unfinished outside donor stays pending.
constexpr int LT = 64, LLW = LT + 2;
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
// The words of the link forest's nodes (exits): pending in-links in bits 40..63, total in bits 0..39.  The per-cell words
// of the raster-wide walk keep their count in 8 bits (a cell has at most 8 donors), but an EXIT can be handed flow by
// every exit of the neighbouring tiles whose path ends at it -- several hundred when a tile funnels everything it
// receives through one cell (r03: the S3 digest test found 1899 cells of FA_D8 wrong where an 8-bit field had wrapped).
// Totals are cell counts of a raster below 2^31 cells: 40 bits are plenty.
constexpr int LK_SHIFT = 40;
constexpr unsigned long long LK_CNT1 = 1ull << LK_SHIFT, LK_LOW = LK_CNT1 - 1ull;
constexpr unsigned long long LK_SRC = 0xFFFFFFull;      // count-field marker: an exit nobody hands anything to
constexpr unsigned long long NOT_A_NODE = 0xFFFFFEull;  // count field of a slot that is not an exit

__device__ __forceinline__ int border_slot(int lx, int ly) {
  if (ly == 0) return lx;
  if (ly == LT - 1) return LT + lx;
  if (lx == 0) return 2 * LT + (ly - 1);
  if (lx == LT - 1) return 2 * LT + (LT - 2) + (ly - 1);
  return -1;
}

// The pointer tables of the tile passes are gathered at random by all 64 lanes; with rows of 64 two-byte entries every row
// starts on the same LDS bank, so lanes that point at neighbouring columns of DIFFERENT rows -- the usual case: flow
// converges -- collide.  Rows of LPS = 66 entries shift the banks by one per row (r03e: SQ_LDS_BANK_CONFLICT was 57 % of
// k_acc_link_tile's LDS cycles, 44 % of k_acc_link_final_sums').  A cell's table index is ly * LPS + lx.
constexpr int LPS = LT + 2;
// ---- the tile passes' common front end (r04d) ---------------------------------------------------------------------
// The staged directions: rows of SDW = 72 bytes with the tile's first column at byte SDO = 4, so that an interior tile is
// staged with aligned 32-bit LDS stores from 32-bit global loads (one byte per load and a division per byte made the
// staging a fifth of k_acc_link_tile's instructions).  Cells outside the raster read as `fill`.
constexpr int SDW = 72, SDO = 4, SDH = LT + 2;
__device__ __forceinline__ void stage_dirs_rows(const uint8_t *__restrict__ dirs, int w, int h, int x0, int y0, uint8_t fill,
                                                uint8_t *sd) {
  if (y0 >= 1 && y0 + LT < h && x0 + LT <= w) {   // (block-uniform) every row of the window lies in the raster
    constexpr int NQ = SDH * (LT / 4), QPT = (NQ + NTHR - 1) / NTHR;
    uint32_t v[QPT];
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i < NQ) __builtin_memcpy(&v[r], dirs + (size_t)(y0 - 1 + (i >> 4)) * w + (x0 + 4 * (i & 15)), 4);   // (any alignment)
    }
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i < NQ) *reinterpret_cast<uint32_t *>(sd + (i >> 4) * SDW + SDO + 4 * (i & 15)) = v[r];
    }
    if (threadIdx.x < 2 * SDH) {   // the two ring columns
      const int ly = (int)threadIdx.x >> 1, side = (int)threadIdx.x & 1;
      const int gx = side ? x0 + LT : x0 - 1;
      sd[ly * SDW + (side ? SDO + LT : SDO - 1)] = (gx >= 0 && gx < w) ? dirs[(size_t)(y0 - 1 + ly) * w + gx] : fill;
    }
  } else {
    for (int i = (int)threadIdx.x; i < SDH * SDH; i += NTHR) {
      const int ly = i / SDH, lx = i - ly * SDH;
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
      sd[ly * SDW + SDO - 1 + lx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? dirs[(size_t)gy * w + gx] : fill;
    }
  }
}
// Per direction 1..8 (index e = d - 1), one byte each, looked up with v_perm_b32 (selector bytes 0..3 pick from the second
// operand, 4..7 from the first, 0x0c gives 0): the target's offset in the staged rows (+73), in the pointer table (+67), and
// which side of the tile it can leave through (1: left, 2: right, 4: top, 8: bottom).
__device__ __forceinline__ uint32_t d8_byte(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
constexpr uint32_t D8_SD_LO = 0x02010048u, D8_SD_HI = 0x9091924Au;   // (dy * SDW + dx) + SDW + 1
constexpr uint32_t D8_LP_LO = 0x02010042u, D8_LP_HI = 0x84858644u;   // (dy * LPS + dx) + LPS + 1
constexpr uint32_t D8_FL_LO = 0x06040501u, D8_FL_HI = 0x09080A02u;
static_assert(SDW == 72 && LPS == 66, "the byte tables above");

__global__ __launch_bounds__(NTHR, 6) void k_acc_link_tile(const uint8_t *__restrict__ dirs, uint8_t nodata, int w, int h,
                                                        uint32_t tilesX, uint32_t ntiles, unsigned long long *nw,
                                                        uint32_t *next, uint8_t *rootslot) {
  // What this pass has to deliver is, per EXIT cell, the number of tile cells whose path leaves the tile through it, and
  // per border cell the exit its path ends at.  Both follow from "the last in-tile cell of every cell's path" (pointer
  // jumping): the exit totals are a histogram over those roots -- one LDS add per cell -- and no accumulation walk is
  // needed here at all (r02c ran the full last-arriver walk in this pass too: 20 of the stage's 47 ms).  Cells that
  // drain into a direction loop inside the tile have no root and are counted nowhere, as in the reference.
  // r04d (the kernel is bound by instruction issue; ~3000 -> ~1600 VALU instructions per wavefront): pointers are byte
  // offsets and an EXIT points to ITSELF, a cell without a target to a self-pointing SINK entry -- a jump is two reads
  // without compares or selects, finished when they agree; the border cells are published one per thread; the target's
  // offsets come from byte tables; the directions are staged 32 bits at a time.
  __shared__ uint32_t cnt[LT * LPS];      // [23:0] cells leaving through this exit, [31:24] the exit's direction
  __shared__ uint16_t lp[LT * LPS + 2];
  uint8_t *const sd = reinterpret_cast<uint8_t *>(cnt);   // the staged directions are dead before the counters are set
  static_assert(SDH * SDW <= LT * LPS * 4, "the staged directions fit into the counters' storage");
  constexpr uint32_t SINK2 = (uint32_t)(LT * LPS * 2);
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * LT, y0 = (int)(t / tilesX) * LT;
  stage_dirs_rows(dirs, w, h, x0, y0, nodata, sd);
  if (threadIdx.x == 0) lp[LT * LPS] = (uint16_t)SINK2;
  __syncthreads();
  const int lx = threadIdx.x & (LT - 1), ly0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave uniform: row arithmetic on the scalar unit)
  constexpr int RPT = LT / 4;
  char *const lpb = reinterpret_cast<char *>(lp);
  const uint32_t self0 = (uint32_t)((ly0 * LPS + lx) * 2);   // row j of the thread: self0 + j * 4 * LPS * 2
  uint32_t p[RPT];          // the cells' pointers (byte offsets into lp)
  uint32_t dex[RPT];        // an exit's direction << 24, else 0: the counters' first value
  {
    const uint32_t cmcol = lx == 0 ? 1u : lx == LT - 1 ? 2u : 0u;
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int ly = ly0 + 4 * j;
      const int o = (ly + 1) * SDW + SDO + lx;
      const uint32_t d = sd[o];
      const uint32_t e = (d - 1u) & 7u, sel = e | 0x0c0c0c00u;
      const bool flows = (d != nodata) & (d - 1u < 8u);
      const uint32_t so = d8_byte(D8_SD_HI, D8_SD_LO, sel), lo = d8_byte(D8_LP_HI, D8_LP_LO, sel), fl = d8_byte(D8_FL_HI, D8_FL_LO, sel);
      const bool into_data = sd[o + (int)so - (SDW + 1)] != nodata;   // else: off the DEM / into NoData: dropped (d8_methods.hpp:113-125)
      uint32_t cm = cmcol;
      if (j == 0 && ly0 == 0) cm |= 4u;                 // (only these two rows of a wavefront can be the tile's first / last)
      if (j == RPT - 1 && ly0 == 3) cm |= 8u;
      const bool leaves = (fl & cm) != 0u, goes = flows & into_data;
      const uint32_t self = self0 + (uint32_t)(j * 4 * LPS * 2);
      const uint32_t inp = self + 2u * lo - (uint32_t)(2 * (LPS + 1));
      const uint32_t ex = leaves ? self : inp;
      p[j] = goes ? ex : SINK2;
      *reinterpret_cast<uint16_t *>(lpb + self) = (uint16_t)p[j];
      dex[j] = (goes & leaves) ? d << 24 : 0u;
    }
  }
  __syncthreads();   // (every thread has read what it needs of sd)
#pragma unroll
  for (int j = 0; j < RPT; j++) cnt[(ly0 + 4 * j) * LPS + lx] = dex[j];
  {
    // two hops per trip with a barrier per trip: twelve trips cover any loop-free path of a 4096-cell tile; what still moves
    // then runs round a direction loop.  A group of four rows whose cells are all finished is skipped with one scalar test.
    uint32_t gact = (1u << (RPT / 4)) - 1u;
#pragma unroll 1
    for (int it = 0; it < 12; it++) {
#pragma unroll
      for (int g = 0; g < RPT / 4; g++) {
        if (!(gact >> g & 1u)) continue;
        uint32_t qv[4], rv[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { qv[e] = *reinterpret_cast<const uint16_t *>(lpb + p[4 * g + e]); asm("" : "+v"(qv[e])); }
#pragma unroll
        for (int e = 0; e < 4; e++) { rv[e] = *reinterpret_cast<const uint16_t *>(lpb + qv[e]); asm("" : "+v"(rv[e])); }   // (opaque 32-bit values: else the compare is narrowed to 16 bits and every value masked again for its use as an address)
        unsigned long long moving = 0;   // (ballots of the plain compares: v_cmp's lane masks, OR-ed on the scalar unit)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int j = 4 * g + e;
          moving |= __builtin_amdgcn_ballot_w64(rv[e] != qv[e]);   // (equal: qv points to itself -- an exit or the sink -- and the cell is finished)
          p[j] = rv[e];
          *reinterpret_cast<uint16_t *>(lpb + self0 + (uint32_t)(j * 4 * LPS * 2)) = (uint16_t)rv[e];
        }
        if (moving == 0ull) gact &= ~(1u << g);
      }
      if (!__syncthreads_or(gact != 0u)) break;
    }
  }
  __syncthreads();
  // every cell adds itself to the exit its path ends at (an exit to itself); a cell whose path ends nowhere, or never
  // ends, adds nothing
  {
    const unsigned long long after = ~((2ull << lx) - 1ull);   // the lanes after this one
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const uint32_t pj = p[j];
      const uint32_t back = *reinterpret_cast<const uint16_t *>(lpb + pj);
      const bool adds = (pj != SINK2) & (back == pj);
      // Neighbouring cells of a row mostly leave through the same exit: a run of lanes with the same target adds its
      // LENGTH once, from its first lane (same-address LDS atomics serialise: SQ_LDS_ADDR_CONFLICT was 26 % of this
      // kernel's LDS cycles).  Lanes that add nothing get a target of their own: runs of length one.
      const uint32_t tgt = adds ? pj : 0x10000u + (uint32_t)lx;
      const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)tgt, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
      const unsigned long long heads = __builtin_amdgcn_ballot_w64(tgt != left);   // (lane 0 is a head: its "left" is the fill)
      if (adds && tgt != left) {
        const unsigned long long above = heads & after;                         // the heads after this lane
        const int end = above ? __ffsll((long long)above) - 1 : 64;             // first lane of the next run
        atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + 2u * tgt), (uint32_t)(end - lx));
      }
    }
  }
  __syncthreads();
  // what the border cells publish, one per thread (252 of the tile's 256 slots; the four spare ones are marked unused)
  {
    const int slot = (int)threadIdx.x;
    const size_t node = (size_t)t * 256 + slot;
    unsigned long long word = NOT_A_NODE << LK_SHIFT;
    uint32_t tn = NO_NODE;
    uint8_t rs = 255;
    if (slot < 4 * LT - 4) {
      const int bx = slot < LT ? slot : slot < 2 * LT ? slot - LT : slot < 3 * LT - 2 ? 0 : LT - 1;
      const int by = slot < LT ? 0 : slot < 2 * LT ? LT - 1 : slot < 3 * LT - 2 ? slot - 2 * LT + 1 : slot - (3 * LT - 2) + 1;
      const uint32_t c2 = (uint32_t)((by * LPS + bx) * 2);
      const uint32_t root = *reinterpret_cast<const uint16_t *>(lpb + c2);
      const uint32_t back = *reinterpret_cast<const uint16_t *>(lpb + root);
      if (root != SINK2 && back == root) {   // the path ends at an exit (the cell itself when it is one)
        const int ri = (int)(root >> 1), ry = ri / LPS, rx = ri - ry * LPS;
        rs = (uint8_t)border_slot(rx, ry);
      }
      if (root == c2) {
        const uint32_t cw = cnt[by * LPS + bx];
        word = (unsigned long long)(cw & 0xFFFFFFu);   // complete by construction: every cell counted has a path to it
        const int d = (int)(cw >> 24);
        const int gx = x0 + bx + d8dx(d), gy = y0 + by + d8dy(d);
        tn = ((uint32_t)(gy / LT) * tilesX + (uint32_t)(gx / LT)) * 256u + (uint32_t)border_slot(gx % LT, gy % LT);
      }
    }
    nw[node] = word;
    next[node] = tn;
    rootslot[node] = rs;
  }
}

// next[q] : the cell q flows to (a border cell of another tile) -> the exit that cell's path leaves its tile through
__global__ __launch_bounds__(NTHR) void k_acc_link_edges(unsigned long long *nw, uint32_t *next,
                                                         const uint8_t *__restrict__ rootslot, uint64_t nnodes) {
  const uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x;
  if (i >= nnodes) return;
  const uint32_t tn = next[i];
  if (tn == NO_NODE) return;
  const uint8_t r = rootslot[tn];
  const uint32_t nx = r == 255 ? NO_NODE : ((tn & ~255u) | r);
  next[i] = nx;
  if (nx != NO_NODE) atomicAdd(&nw[nx], LK_CNT1);
}

__global__ __launch_bounds__(NTHR) void k_acc_link_sources(unsigned long long *nw, uint64_t nnodes) {
  const uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x;
  if (i >= nnodes) return;
  const unsigned long long v = nw[i];
  if ((v >> LK_SHIFT) == 0) nw[i] = v | (LK_SRC << LK_SHIFT);   // an exit nobody hands anything to: a source of the link forest
}

__global__ __launch_bounds__(NTHR) void k_acc_link_walk(unsigned long long *nw, const uint32_t *__restrict__ next,
                                                        uint64_t nnodes) {
  // lane-refill walk over the exits (see k_acc_walk_unit)
  const uint64_t wave = ((uint64_t)blockIdx.x * NTHR + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t nxt = wave * WALK_CHUNK;
  const uint64_t end = nxt + WALK_CHUNK < nnodes ? nxt + WALK_CHUNK : nnodes;
  if (nxt >= nnodes) return;
  bool active = false;
  uint32_t c = 0;
  unsigned long long v = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    if (nxt < end && idle) {
      const uint64_t my = nxt + (uint64_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (!active && my < end) {
        const unsigned long long wd = nw[my];   // a source's word is never modified: plain read
        if ((wd >> LK_SHIFT) == LK_SRC) { active = true; c = (uint32_t)my; v = wd & LK_LOW; }
      }
      nxt += (uint64_t)__popcll(idle);
    } else if (idle == ~0ull) {
      break;
    }
    if (active) {
      const uint32_t t = next[c];
      if (t == NO_NODE) active = false;
      else {
        const unsigned long long old = atomicAdd(&nw[t], v - LK_CNT1);
        if ((old >> LK_SHIFT) != 1) active = false;
        else { v = (old & LK_LOW) + v; c = t; }
      }
    }
  }
}

template <class A>
__device__ __forceinline__ void link_final_walk_tile(const uint32_t t, const uint8_t *__restrict__ dirs, uint8_t nodata, int w, int h,
                                                     uint32_t tilesX, const unsigned long long *__restrict__ nw, A *__restrict__ area) {
  __shared__ uint8_t sd[LLW * LLW];
  // one word per cell: pending donors (bits 56..63) | in-tile target, 0x1FFF for none (bits 43..55) | total (43 bits)
  __shared__ unsigned long long lw[LT * LT];
  __shared__ unsigned long long ext_in[NTHR];
  __shared__ uint8_t ext_blk[NTHR];
  constexpr unsigned long long TMASK = (1ull << 43) - 1ull, NOTGT = 0x1FFFull;
  const int tx0 = (int)(t % tilesX), ty0 = (int)(t / tilesX);
  const int x0 = tx0 * LT, y0 = ty0 * LT;
  stage_dirs<LLW, NTHR>(dirs, w, h, x0, y0, nodata, sd);
  __syncthreads();
  {
    // what the border cells receive from outside, one border cell per thread (all lookups of the block in flight
    // together): finished donors bring their total, unfinished ones (direction loops) block the cell for good
    const int slot = (int)threadIdx.x;
    unsigned long long inflow = 0;
    uint32_t blocked = 0;
    if (slot < 4 * LT - 4) {
      const int bx = slot < LT ? slot : slot < 2 * LT ? slot - LT : slot < 3 * LT - 2 ? 0 : LT - 1;
      const int by = slot < LT ? 0 : slot < 2 * LT ? LT - 1 : slot < 3 * LT - 2 ? slot - 2 * LT + 1 : slot - (3 * LT - 2) + 1;
      const int o = (by + 1) * LLW + bx + 1;
      if (sd[o] != nodata) {
        unsigned long long wv[8];
        bool use[8];
#pragma unroll
        for (int m = 1; m <= 8; m++) {
          const int nx = bx + d8dx(m), ny = by + d8dy(m);
          const uint8_t dn = sd[o + d8dy(m) * LLW + d8dx(m)];
          use[m - 1] = !(nx >= 0 && nx < LT && ny >= 0 && ny < LT) && dn != nodata && dn == (m <= 4 ? m + 4 : m - 4);
          wv[m - 1] = 0;
          if (use[m - 1]) {
            const int gx = x0 + nx, gy = y0 + ny;
            wv[m - 1] = nw[((size_t)(gy / LT) * tilesX + (size_t)(gx / LT)) * 256 + (size_t)border_slot(gx % LT, gy % LT)];
          }
        }
#pragma unroll
        for (int m = 0; m < 8; m++) {
          if (!use[m]) continue;
          const unsigned long long cnt = wv[m] >> LK_SHIFT;
          if (cnt == 0 || cnt == LK_SRC) inflow += wv[m] & LK_LOW;
          else blocked++;
        }
      }
    }
    ext_in[slot] = inflow;
    ext_blk[slot] = (uint8_t)blocked;
  }
  __syncthreads();
  const int lx = threadIdx.x & (LT - 1), ly0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave uniform: row arithmetic on the scalar unit)
  uint32_t tgs[LT / 4];   // in-tile target of the thread's cells (NOTGT: none)
  uint32_t datamask = 0;
#pragma unroll 4
  for (int j = 0; j < LT / 4; j++) {   // every cell without branches (see link_stage); the border cells' inflow follows
    const int ly = ly0 + 4 * j, o = (ly + 1) * LLW + lx + 1;
    const uint8_t d = sd[o];
    const bool data = d != nodata, flows = data && d >= 1 && d <= 8;
    const int dd = flows ? d : 0;
    const int tx = lx + d8dx(dd), ty = ly + d8dy(dd);
    const bool in_tile = flows && tx >= 0 && tx < LT && ty >= 0 && ty < LT && sd[(ty + 1) * LLW + tx + 1] != nodata;
    tgs[j] = in_tile ? (uint32_t)(ty * LT + tx) : (uint32_t)NOTGT;
    datamask |= (data ? 1u : 0u) << j;
    lw[ly * LT + lx] = data ? (((unsigned long long)tgs[j] << 43) | 1ull) : 0ull;
  }
  __syncthreads();
  if (threadIdx.x < 4 * LT - 4) {   // one border cell per thread: what it receives from outside, and what blocks it
    const int slot = (int)threadIdx.x;
    const int bx = slot < LT ? slot : slot < 2 * LT ? slot - LT : slot < 3 * LT - 2 ? 0 : LT - 1;
    const int by = slot < LT ? 0 : slot < 2 * LT ? LT - 1 : slot < 3 * LT - 2 ? slot - 2 * LT + 1 : slot - (3 * LT - 2) + 1;
    const unsigned long long add = ((unsigned long long)ext_blk[slot] << 56) | ext_in[slot];
    if (add && lw[by * LT + bx] != 0) lw[by * LT + bx] += add;   // (a NoData border cell receives nothing: its gather found no donor)
  }
  __syncthreads();
  for (int j = 0; j < LT / 4; j++)   // pending donors, counted from the donors' side (see k_acc_link_tile)
    if (tgs[j] != (uint32_t)NOTGT) atomicAdd(&lw[tgs[j]], CNT1);
  __syncthreads();
  uint32_t srcmask = 0;
#pragma unroll
  for (int j = 0; j < LT / 4; j++) {   // (selects on registers: no branch per cell)
    const int ly = ly0 + 4 * j;
    srcmask |= ((datamask >> j & 1u) & ((uint32_t)(lw[ly * LT + lx] >> 56) == 0 ? 1u : 0u)) << j;
  }
  __syncthreads();   // the sources are fixed before any walk completes a cell
  {
    constexpr uint32_t NT = (uint32_t)NOTGT;
    uint32_t m = srcmask, tg = NT;   // (the target as a 32-bit value: the 64-bit compares and moves doubled the loop's bookkeeping)
    unsigned long long v = 0;
    for (;;) {   // a lane whose walk has ended takes its next source in the same trip (see k_acc_link_tile)
      if (tg == NT && m) {
        const unsigned long long own = lw[(ly0 + 4 * (__ffs((int)m) - 1)) * LT + lx];
        tg = (uint32_t)(own >> 43) & NT;
        v = own & TMASK;
        m &= m - 1;
      }
      if (!__any(tg != NT || m != 0)) break;
      if (tg != NT) {
        const unsigned long long old = atomicAdd(&lw[tg], v - CNT1);
        const bool last = (uint32_t)(old >> 56) == 1u;
        v = last ? (old & TMASK) + v : v;
        tg = last ? (uint32_t)(old >> 43) & NT : NT;
      }
    }
  }
  __syncthreads();
  for (int j = 0; j < LT / 4; j++) {
    const int ly = ly0 + 4 * j, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    A out = (A)-1;                                                  // area.noData() == -1, d8_methods.hpp:64,:72-75
    if (sd[(ly + 1) * LLW + lx + 1] != nodata) {
      const unsigned long long v = lw[ly * LT + lx];
      // cells downstream of a direction loop are never completed by the reference either: they keep the sum of the
      // inflows that did arrive, without their own +1 (d8_methods.hpp:104-131)
      out = (A)((v >> 56) != 0 ? (v & TMASK) - 1ull : (v & TMASK));
      if (std::is_same<A, unsigned long long>::value) out = (A)(((v >> 56) != 0 ? CNT1 : 0ull) | (v & TMASK));   // a shard's words
    }
    area[(size_t)gy * w + gx] = out;
  }
}

// every tile once (RDGPU_ACCUM_SUMS=0: the r02 pass)
template <class A>
__global__ __launch_bounds__(NTHR) void k_acc_link_final(const uint8_t *__restrict__ dirs, uint8_t nodata, int w, int h,
                                                         uint32_t tilesX, uint32_t ntiles,
                                                         const unsigned long long *__restrict__ nw, A *__restrict__ area) {
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  link_final_walk_tile<A>(t, dirs, nodata, w, h, tilesX, nw, area);
}

// only the tiles k_acc_link_final_sums handed over: blocked donors, direction loops (grid-stride over the list)
template <class A>
__global__ __launch_bounds__(NTHR) void k_acc_link_final_list(const uint8_t *__restrict__ dirs, uint8_t nodata, int w, int h,
                                                              uint32_t tilesX, const unsigned long long *__restrict__ nw,
                                                              A *__restrict__ area, const uint32_t *__restrict__ tile_list,
                                                              const uint32_t *__restrict__ tile_count) {
  const uint32_t n = *tile_count;
  for (uint32_t wi = blockIdx.x; wi < n; wi += gridDim.x) {
    link_final_walk_tile<A>(tile_list[wi], dirs, nodata, w, h, tilesX, nw, area);
    __syncthreads();   // (the next tile of the list reuses the arrays)
  }
}

// The final pass without a walk.  With every donor outside the tile finished (no direction loop upstream) and no loop
// inside the tile, a cell's total is the sum of the weights of its in-tile SUBTREE (weight = 1, plus what a border cell
// receives from outside), and subtree sums come out of pointer doubling: S_k(v) = weights of v's descendants closer than
// 2^k; round k hands S_k(u) to u's 2^k-th ancestor, and S_(k+1)(v) = S_k(v) + what arrives (every descendant at distance
// [2^k, 2^(k+1)) is counted by exactly one such u).  ceil(log2(longest in-tile path)) rounds of "read own sum, own
// pointer, the pointer's pointer -- barrier -- one 32-bit LDS add, one pointer store", a cell dropping out as soon as
// its pointer runs off its path; the last-arriver walk this replaces paid one returning 64-bit LDS atomic per cell of
// the longest chain per round of sources, at ~10 % lane use (r02: 20.8 of the stage's 37 ms).  Totals are cell counts
// of a raster below 2^31 cells: 32 bits (24 KB of LDS per tile instead of 38: five blocks per CU).  A tile with a
// blocked donor, or whose pointers still move after twelve doublings (a direction loop), is appended to slow_tiles and
// left to k_acc_link_final, which keeps the reference's partial sums there.
template <class A>
__global__ __launch_bounds__(NTHR, 5) void k_acc_link_final_sums(const uint8_t *__restrict__ dirs, uint8_t nodata, int w, int h,
                                                               uint32_t tilesX, uint32_t ntiles,
                                                               const unsigned long long *__restrict__ nw, A *__restrict__ area,
                                                               uint32_t *slow_tiles, uint32_t *slow_count) {
  // (r04d, like k_acc_link_tile: ancestors are byte offsets, a cell whose path has ended points to a self-pointing SINK
  // entry -- a round reads and hands on without per-cell state tests; targets from byte tables; 32-bit staging)
  __shared__ __attribute__((aligned(4))) uint8_t sd[SDH * SDW];
  __shared__ uint32_t S[LT * LPS + 2];      // (rows of LPS entries: see LPS)
  __shared__ uint16_t anc[LT * LPS + 2];
  constexpr uint32_t SINK2 = (uint32_t)(LT * LPS * 2);
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * LT, y0 = (int)(t / tilesX) * LT;
  stage_dirs_rows(dirs, w, h, x0, y0, nodata, sd);
  if (threadIdx.x == 0) anc[LT * LPS] = (uint16_t)SINK2;
  __syncthreads();
  // what the border cells receive from outside, one border cell per thread (all lookups in flight together)
  unsigned long long inflow = 0;
  uint32_t blocked = 0;
  int bcell = -1;
  {
    const int slot = (int)threadIdx.x;
    if (slot < 4 * LT - 4) {
      const int bx = slot < LT ? slot : slot < 2 * LT ? slot - LT : slot < 3 * LT - 2 ? 0 : LT - 1;
      const int by = slot < LT ? 0 : slot < 2 * LT ? LT - 1 : slot < 3 * LT - 2 ? slot - 2 * LT + 1 : slot - (3 * LT - 2) + 1;
      const int o = (by + 1) * SDW + SDO + bx;
      if (sd[o] != nodata) {
        bcell = by * LPS + bx;
        unsigned long long wv[8];
        bool use[8];
#pragma unroll
        for (int m = 1; m <= 8; m++) {
          const int nx = bx + d8dx(m), ny = by + d8dy(m);
          const uint8_t dn = sd[o + d8dy(m) * SDW + d8dx(m)];
          use[m - 1] = !(nx >= 0 && nx < LT && ny >= 0 && ny < LT) && dn != nodata && dn == (m <= 4 ? m + 4 : m - 4);
          wv[m - 1] = 0;
          if (use[m - 1]) {
            const uint32_t gx = (uint32_t)(x0 + nx), gy = (uint32_t)(y0 + ny);   // (a cell with data: inside the raster)
            static_assert(LT == 64, "shifts and masks below");
            wv[m - 1] = nw[((size_t)(gy >> 6) * tilesX + (size_t)(gx >> 6)) * 256 + (size_t)border_slot((int)(gx & 63u), (int)(gy & 63u))];
          }
        }
#pragma unroll
        for (int m = 0; m < 8; m++) {
          if (!use[m]) continue;
          const unsigned long long cnt = wv[m] >> LK_SHIFT;
          if (cnt == 0 || cnt == LK_SRC) inflow += wv[m] & LK_LOW;
          else blocked++;
        }
      }
    }
  }
  const int lx = threadIdx.x & (LT - 1), ly0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave uniform: row arithmetic on the scalar unit)
  constexpr int RPT = LT / 4;
  char *const ancb = reinterpret_cast<char *>(anc), *const Sb = reinterpret_cast<char *>(S);
  const uint32_t self0 = (uint32_t)((ly0 * LPS + lx) * 2);   // row j of the thread: self0 + j * 4 * LPS * 2 (anc), twice that (S)
  uint32_t a[RPT];         // the 2^k-th ancestor of the thread's cells (byte offset into anc; SINK2: the path is shorter)
  uint32_t datamask = 0;
  {
    const uint32_t cmcol = lx == 0 ? 1u : lx == LT - 1 ? 2u : 0u;
#pragma unroll
    for (int j = 0; j < RPT; j++) {   // every cell without branches
      const int ly = ly0 + 4 * j;
      const int o = (ly + 1) * SDW + SDO + lx;
      const uint32_t d = sd[o];
      const uint32_t e = (d - 1u) & 7u, sel = e | 0x0c0c0c00u;
      const bool data = d != nodata, flows = data & (d - 1u < 8u);
      const uint32_t so = d8_byte(D8_SD_HI, D8_SD_LO, sel), lo = d8_byte(D8_LP_HI, D8_LP_LO, sel), fl = d8_byte(D8_FL_HI, D8_FL_LO, sel);
      const bool into_data = sd[o + (int)so - (SDW + 1)] != nodata;
      uint32_t cm = cmcol;
      if (j == 0 && ly0 == 0) cm |= 4u;
      if (j == RPT - 1 && ly0 == 3) cm |= 8u;
      const bool in_tile = flows & into_data & ((fl & cm) == 0u);
      const uint32_t self = self0 + (uint32_t)(j * 4 * LPS * 2);
      const uint32_t up = self + 2u * lo - (uint32_t)(2 * (LPS + 1));
      a[j] = in_tile ? up : SINK2;
      datamask |= (data ? 1u : 0u) << j;
      *reinterpret_cast<uint32_t *>(Sb + 2u * self) = data ? 1u : 0u;
      *reinterpret_cast<uint16_t *>(ancb + self) = (uint16_t)a[j];
    }
  }
  __syncthreads();
  if (bcell >= 0 && inflow) S[bcell] += (uint32_t)inflow;   // (one thread per border cell)
  if (__syncthreads_or(blocked != 0 || inflow >= (1ull << 31))) {
    if (threadIdx.x == 0) slow_tiles[atomicAdd(slow_count, 1u)] = t;
    return;
  }
  int it = 0;
  uint32_t gact = (1u << (RPT / 4)) - 1u;   // (scalar) groups of four rows with a cell still on its path
#pragma unroll 1
  for (; it < 13; it++) {
    uint32_t sv[RPT], aa[RPT];
#pragma unroll
    for (int g = 0; g < RPT / 4; g++) {   // groups of four rows: one scalar test skips a group that is done
      if (!(gact >> g & 1u)) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = 4 * g + e;
        sv[j] = *reinterpret_cast<const uint32_t *>(Sb + 2u * (self0 + (uint32_t)(j * 4 * LPS * 2)));
        aa[j] = *reinterpret_cast<const uint16_t *>(ancb + a[j]);   // (a finished cell reads the sink's entry: the sink)
        asm("" : "+v"(aa[j]));
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < RPT / 4; g++) {
      if (!(gact >> g & 1u)) continue;
      unsigned long long on = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = 4 * g + e;
        if (a[j] != SINK2) atomicAdd(reinterpret_cast<uint32_t *>(Sb + 2u * a[j]), sv[j]);
        *reinterpret_cast<uint16_t *>(ancb + self0 + (uint32_t)(j * 4 * LPS * 2)) = (uint16_t)aa[j];
        a[j] = aa[j];
        on |= __builtin_amdgcn_ballot_w64(aa[j] != SINK2);
      }
      if (on == 0ull) gact &= ~(1u << g);
    }
    if (!__syncthreads_or(gact != 0u)) break;
  }
  if (it >= 13) {   // 2^12 steps and still on a path: a direction loop inside the tile
    if (threadIdx.x == 0) slow_tiles[atomicAdd(slow_count, 1u)] = t;
    return;
  }
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int ly = ly0 + 4 * j, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(Sb + 2u * (self0 + (uint32_t)(j * 4 * LPS * 2)));
    area[(size_t)gy * w + gx] = (datamask >> j & 1u) ? (A)v : (A)-1;   // area.noData() == -1, d8_methods.hpp:64,:72-75
  }
}

// ---- f64 weights ----------------------------------------------------------------------------
// Tile pre-walk for f64 weights: k_acc_tile_prewalk's scheme with the totals as doubles in LDS.  It also
// replaces a separate inflow-counting pass: the pending words written here are what the raster-wide walk
// expects ((inflows in total << 12) | (direction << 8) | inflows still to arrive / SRC32 / NODATA32).
// LDS executes a wavefront's operations in order, so "add, then decrement, then (if last) read" needs no fences
// here: the last decrementer's read comes after every other arriver's decrement, hence after their adds.
__global__ __launch_bounds__(NTHR) void k_acc_tile_prewalk_f64(const uint8_t *__restrict__ dirs, uint32_t *__restrict__ pending,
                                                               double *__restrict__ acc, int w, int h, uint32_t tilesX,
                                                               uint32_t ntiles) {
  __shared__ uint8_t sd[ALH * ALW];
  __shared__ uint32_t lc[AH * AW];
  __shared__ double lt[AH * AW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * AW, y0 = (int)(t / tilesX) * AH;
  stage_dirs<ALW, NTHR>(dirs, w, h, x0, y0, (uint8_t)255, sd);
  const int lx = threadIdx.x & (AW - 1), ly0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave uniform: row arithmetic on the scalar unit)
  {
    double wv[AH / 4];
#pragma unroll
    for (int j = 0; j < AH / 4; j++) {   // the weights: all loads in flight together
      const int gx = x0 + lx, gy = y0 + ly0 + 4 * j;
      wv[j] = (gx < w && gy < h) ? acc[(size_t)gy * w + gx] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < AH / 4; j++) lt[(ly0 + 4 * j) * AW + lx] = wv[j];
  }
  __syncthreads();
  uint32_t srcmask = 0;
  for (int j = 0; j < AH / 4; j++) {
    const int ly = ly0 + 4 * j, o = (ly + 1) * ALW + lx + 1;
    uint32_t v = NODATA32;
    if (x0 + lx < w && y0 + ly < h && sd[o] != 255) {
      uint32_t k = 0;
#pragma unroll
      for (int m = 1; m <= 8; m++) {
        const uint8_t d = sd[o + d8dy(m) * ALW + d8dx(m)];   // halo cells outside the raster hold 255
        if (d != 255 && d == (m <= 4 ? m + 4 : m - 4)) k++;   // neighbour m flows into us (constants.hpp:65)
      }
      v = (k << 12) | ((uint32_t)(sd[o] <= 8 ? sd[o] : 0) << 8) | k;
      if (k == 0) srcmask |= 1u << j;
    }
    lc[ly * AW + lx] = v;
  }
  __syncthreads();
  {
    // a lane whose walk has ended takes its next source in the same trip (see k_acc_link_tile)
    uint32_t m = srcmask;
    bool active = false;
    int cx = 0, cy = 0;
    double v = 0;
    for (;;) {
      if (!active && m) {
        cx = lx; cy = ly0 + 4 * (__ffs((int)m) - 1);
        m &= m - 1;
        v = lt[cy * AW + cx];
        active = true;
      }
      if (!__any(active || m != 0)) break;
      if (active) {
        const uint8_t d = sd[(cy + 1) * ALW + cx + 1];
        bool stalled = false, cont = false;
        if (d >= 1 && d <= 8) {
          const int tx = cx + d8dx(d), ty = cy + d8dy(d);
          if (sd[(ty + 1) * ALW + tx + 1] != 255) {                // else: off the DEM / into NoData: dropped
            if (tx >= 1 && tx < AW - 1 && ty >= 1 && ty < AH - 1) {
              atomicAdd(&lt[ty * AW + tx], v);
              const uint32_t old = atomicSub(&lc[ty * AW + tx], 1u);
              if ((old & 0xFFu) == 1) {
                v = __hip_atomic_load(&lt[ty * AW + tx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                cx = tx; cy = ty;
                cont = true;
              }
            } else {
              stalled = true;                                      // ring cell or another tile: continue raster-wide
            }
          }
        }
        if (stalled) lc[cy * AW + cx] = (lc[cy * AW + cx] & ~0xFFu) | SRC32;   // complete, still has to pass its total on
        active = cont;
      }
    }
  }
  __syncthreads();
  for (int j = 0; j < AH / 4; j++) {
    const int ly = ly0 + 4 * j, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    const uint32_t v = lc[ly * AW + lx];
    pending[g] = v;
    if (v != NODATA32) acc[g] = lt[ly * AW + lx];
  }
}

__global__ __launch_bounds__(NTHR) void k_acc_walk_f64(uint32_t *pending, double *acc, int w, int h) {
  // lane-refill walk (see k_acc_walk_unit)
  const uint64_t n = (uint64_t)w * h;
  const uint64_t wave = ((uint64_t)blockIdx.x * NTHR + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t next = wave * WALK_CHUNK;
  const uint64_t end = next + WALK_CHUNK < n ? next + WALK_CHUNK : n;
  if (next >= n) return;
  bool active = false;
  uint32_t c = 0;
  double v = 0;
  uint8_t d = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    if (next < end && idle) {
      const uint64_t my = next + (uint64_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (!active && my < end) {
        // sources only; a source's counter and total are never written by anyone else
        const uint32_t p = pending[my];
        if ((p & 0xFFu) == SRC32) { active = true; c = (uint32_t)my; v = acc[my]; d = (uint8_t)((p >> 8) & 15u); }
      }
      next += (uint64_t)__popcll(idle);
    } else if (idle == ~0ull) {
      break;
    }
    if (active) {
      const int64_t t = flow_target(c, d, w, h);
      if (t < 0) active = false;
      else {
        // flow_accumulation_generic.hpp:85-87 (proportion is exactly 1 for D8).  A NoData target is not looked
        // up first: its count byte never reads 1, and its total is overwritten with -1 at the end.
        // All three steps are device-scope atomic RMWs executed at the memory side: a RETURNING add has
        // completed there before the decrement is issued, and the last arriver's decrement is ordered after
        // every other arriver's decrement, hence after their adds -- no cache write-back / invalidate
        // (release/acquire fences cost ~10x here).
        const double prev = atomicAdd(&acc[t], v);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");   // the add has returned
        const uint32_t old = __hip_atomic_fetch_sub(&pending[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 0xFFu) != 1) active = false;
        else {
          // the only inflow of t (the common case): what the add returned was t's own weight, so the total is known
          // without a third round trip; at a confluence the final total is read back at the memory side
          v = ((old >> 12) & 15u) == 1 ? prev + v : atomicAdd(&acc[t], 0.0);
          c = (uint32_t)t;
          d = (uint8_t)((old >> 8) & 15u);
        }
      }
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
  const char *env = getenv("RDGPU_ACCUM_LINKS");   // =0: the raster-wide walk (what the row-block shards run); A/B and tests
  if (!(env && env[0] == '0')) {
    const uint32_t tilesX = (w + LT - 1) / LT, ntiles = tilesX * ((h + LT - 1) / LT);
    const uint64_t nnodes = (uint64_t)ntiles * 256;
    if (nnodes < 0xFFFFFF00ull) {
      Workspace &ws = Workspace::get();
      unsigned long long *nw = ws.buf<unsigned long long>("accum.link_word", nnodes);
      uint32_t *next = ws.buf<uint32_t>("accum.link_next", nnodes);
      uint8_t *rootslot = ws.buf<uint8_t>("accum.link_root", nnodes);
      const uint32_t ngrid = (uint32_t)((nnodes + NTHR - 1) / NTHR);
      RD_LAUNCH("accum.link_tile", k_acc_link_tile, dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_dirs, nodata, w, h, tilesX, ntiles,
                nw, next, rootslot);
      RD_LAUNCH("accum.link_edges", k_acc_link_edges, dim3(ngrid), dim3(NTHR), 0, s, nw, next, (const uint8_t *)rootslot, nnodes);
      RD_LAUNCH("accum.link_sources", k_acc_link_sources, dim3(ngrid), dim3(NTHR), 0, s, nw, nnodes);
      RD_LAUNCH("accum.link_walk", k_acc_link_walk, dim3((uint32_t)(((nnodes + WALK_CHUNK - 1) / WALK_CHUNK + 3) / 4)), dim3(NTHR),
                0, s, nw, (const uint32_t *)next, nnodes);
      const char *sums = getenv("RDGPU_ACCUM_SUMS");   // =0: the last-arriver walk in every tile (r02); A/B and tests
      if (sums && sums[0] == '0') {
        RD_LAUNCH("accum.link_final", (k_acc_link_final<A>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_dirs, nodata, w, h, tilesX,
                  ntiles, (const unsigned long long *)nw, d_area);
        return;
      }
      uint32_t *slow = ws.buf<uint32_t>("accum.link_slow", (size_t)ntiles + 1);   // [0]: count, then the tiles
      RD_HIP(hipMemsetAsync(slow, 0, sizeof(uint32_t), s));
      RD_LAUNCH("accum.link_final", (k_acc_link_final_sums<A>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_dirs, nodata, w, h,
                tilesX, ntiles, (const unsigned long long *)nw, d_area, slow + 1, slow);
      RD_LAUNCH("accum.link_final_loops", (k_acc_link_final_list<A>), dim3(std::min<uint32_t>(ntiles, 2048u)), dim3(NTHR), 0, s,
                d_dirs, nodata, w, h, tilesX, (const unsigned long long *)nw, d_area, (const uint32_t *)(slow + 1),
                (const uint32_t *)slow);
      return;
    }
  }
  unsigned long long *word = Workspace::get().buf<unsigned long long>("accum.word", n);
  {
    const uint32_t tilesX = (w + AW - 1) / AW, ntiles = tilesX * ((h + AH - 1) / AH);
    RD_LAUNCH("accum.tile_prewalk", k_acc_tile_prewalk, dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_dirs, nodata, word, w, h,
              tilesX, ntiles);
  }
  RD_LAUNCH("accum.walk_unit", k_acc_walk_unit, dim3((uint32_t)(((n + WALK_CHUNK - 1) / WALK_CHUNK + 3) / 4)), dim3(NTHR), 0,
            s, d_dirs, nodata, word, w, h);
  RD_LAUNCH("accum.out_unit", (k_acc_out_unit<A>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, nodata, word, d_area, n);
}

// d_dirs uses 255 as NoData marker (output of flowdirs_device); d_acc holds the per-cell weights on entry.
__global__ __launch_bounds__(NTHR) void k_acc_not_all_ones(const double *__restrict__ acc, uint64_t n, uint32_t *flag) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  bool other = false;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) other |= acc[i] != 1.0;
  if (__any(other) && (threadIdx.x & 63) == 0) *flag = 1;
}

template <class A>
void d8_flow_accum_device(const uint8_t *d_dirs, uint8_t nodata, int w, int h, A *d_area, hipStream_t s);

__global__ __launch_bounds__(NTHR) void k_acc_ones(double *acc, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) acc[i] = 1.0;
}

// Weighted accumulation over FM_D8 directions (255 = NoData), in place in d_acc.
// Every weight 1 -- what FA_D8 is called with unless the caller has weights (the Python wrapper's and the apps'
// accum.setAll(1)): the directions come from a DEM, so they are loop-free and the sums are d8_flow_accum's cell counts,
// integers, exact in a double; that goes through the tile links (56 ms instead of 85 at S3).  One read of the weights
// decides.  Anything else: tile prewalk + raster-wide walk on doubles.  (Tile links on doubles were measured too: three
// LDS arrays per tile leave 2-3 blocks per CU, 112 ms at S3 -- not kept.)
static bool unit_check_enabled() {
  const char *unit = getenv("RDGPU_ACCUM_UNIT");   // =0: always the weighted path (A/B and tests)
  return !(unit && unit[0] == '0');
}

// "are all weights 1": enqueued on `on` (which has been made to wait for the caller's stream), answered by unit_check_end
static void unit_check_begin(const double *d_acc, uint64_t n, hipStream_t on) {
  uint32_t *flag = Workspace::get().buf<uint32_t>("accum.unit_flag", 1);
  RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), on));
  RD_LAUNCH("accum.unit_check", k_acc_not_all_ones, dim3(sgrid(n)), dim3(NTHR), 0, on, d_acc, n, flag);
}
static bool unit_check_end(hipStream_t on) {
  uint32_t *flag = Workspace::get().buf<uint32_t>("accum.unit_flag", 1);
  uint32_t other = 1;
  RD_HIP(hipMemcpyAsync(&other, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, on));
  RD_HIP(hipStreamSynchronize(on));
  return other == 0;
}

// unit_known: -1 = find out here; 0 / 1 = the caller has checked the weights already (fa_d8_device does, beside its
// direction pass)
void flow_accum_f64_device(const uint8_t *d_dirs, int w, int h, double *d_acc, hipStream_t s, int unit_known = -1) {
  {
    const uint64_t n = (uint64_t)w * h;
    bool unit = unit_known == 1;
    if (unit_known < 0 && unit_check_enabled()) {
      unit_check_begin(d_acc, n, s);
      unit = unit_check_end(s);
    }
    if (unit) {
      d8_flow_accum_device<double>(d_dirs, (uint8_t)255, w, h, d_acc, s);
      return;
    }
  }
  const uint64_t n = (uint64_t)w * h;
  uint32_t *pending = Workspace::get().buf<uint32_t>("accum.pending", n);
  {
    const uint32_t tilesX = (w + AW - 1) / AW, ntiles = tilesX * ((h + AH - 1) / AH);
    RD_LAUNCH("accum.tile_prewalk_f64", k_acc_tile_prewalk_f64, dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_dirs, pending, d_acc,
              w, h, tilesX, ntiles);
  }
  RD_LAUNCH("accum.walk_f64", k_acc_walk_f64, dim3((uint32_t)(((n + WALK_CHUNK - 1) / WALK_CHUNK + 3) / 4)), dim3(NTHR), 0, s,
            pending, d_acc, w, h);
  RD_LAUNCH("accum.nodata_f64", k_acc_nodata_f64, dim3(sgrid(n)), dim3(NTHR), 0, s, d_dirs, d_acc, n);
}

// unit_weights: the caller GUARANTEES that every cell generates a flow of 1 (it built the array itself: the apps'
// Array2D<double> accum(dem, 1), rd.FlowAccumulation(weights=None)) -- d_acc is then output only and the weights are
// neither read (k_acc_not_all_ones: 12.8 GB at S3) nor, on the host path, uploaded (12.8 GB over PCIe)
template <class T>
void fa_d8_device(const T *d_z, T nodata, int w, int h, double *d_acc, hipStream_t s, bool unit_weights = false) {
  if (!d_z || !d_acc) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_d8: null pointer");
  check_dims(w, h, "rdgpu_fa_d8");
  Workspace &ws = Workspace::get();
  uint8_t *dirs = ws.buf<uint8_t>("accum.fmdirs", (size_t)w * h);
  // the one read of the weights runs on the side stream while the directions are made on the caller's
  int unit = -1;
  if (unit_weights && unit_check_enabled()) {
    flowdirs_device<T>(d_z, nodata, w, h, dirs, MODE_FM, s);
    d8_flow_accum_device<double>(dirs, (uint8_t)255, w, h, d_acc, s);
    return;
  }
  if (unit_weights)   // RDGPU_ACCUM_UNIT=0 (A/B and tests): the weighted engine on an array of ones made here
    RD_LAUNCH("accum.ones", k_acc_ones, dim3(sgrid((uint64_t)w * h)), dim3(NTHR), 0, s, d_acc, (uint64_t)w * h);
  if (unit_check_enabled()) {
    Workspace::SideLane &lane = ws.side_lane(0);
    RD_HIP(hipEventRecord(lane.fork, s));
    RD_HIP(hipStreamWaitEvent(lane.stream, lane.fork, 0));
    unit_check_begin(d_acc, (uint64_t)w * h, lane.stream);
    flowdirs_device<T>(d_z, nodata, w, h, dirs, MODE_FM, s);
    unit = unit_check_end(lane.stream) ? 1 : 0;
  } else {
    flowdirs_device<T>(d_z, nodata, w, h, dirs, MODE_FM, s);
  }
  flow_accum_f64_device(dirs, w, h, d_acc, s, unit);
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
static void fa_d8_host(const T *dem, T nodata, int w, int h, double *accum, bool unit_weights = false) {
  if (!dem || !accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_d8: null pointer");
  check_dims(w, h, "rdgpu_fa_d8");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  double *da = Workspace::get().buf<double>("host.area", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  if (!unit_weights) RD_HIP(hipMemcpy(da, accum, n * sizeof(double), hipMemcpyHostToDevice));
  fa_d8_device<T>(d, nodata, w, h, da, nullptr, unit_weights);
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
  uint8_t local;   // 1: the one-exchange protocol -- pending counts ignore the donors across the cuts
};

__device__ __forceinline__ uint8_t accs_dir(const AccShard &s, int x, int y) {
  if (x < 0 || x >= s.w) return s.nodata;            // treated as "never flows in"
  if (y < 0) return s.above && !s.local ? s.above[x] : s.nodata;
  if (y >= s.h) return s.below && !s.local ? s.below[x] : s.nodata;
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

// one walk step from cell c (final total v); returns false when the walk ends
__device__ __forceinline__ bool accs_step(const AccShard &s, uint32_t &c, unsigned long long &v, uint8_t &d) {
  if (d < 1 || d > 8) return false;
  const int x = (int)(c % (uint32_t)s.w) + d8dx(d), y = (int)(c / (uint32_t)s.w) + d8dy(d);
  if (x < 0 || x >= s.w) return false;                           // off the DEM
  if (y < 0) {                                                   // across the upper cut (or off the DEM)
    if (s.above && s.above[x] != s.nodata) atomicAdd(&s.out_top[x], v + CNT1);
    return false;
  }
  if (y >= s.h) {
    if (s.below && s.below[x] != s.nodata) atomicAdd(&s.out_bottom[x], v + CNT1);
    return false;
  }
  const uint32_t t = (uint32_t)y * (uint32_t)s.w + (uint32_t)x;
  const uint8_t dt = s.dirs[t];
  if (dt == s.nodata) return false;
  const unsigned long long old = atomicAdd(&s.word[t], v - CNT1);
  if ((old >> 56) != 1) return false;
  v = (old & LOWMASK) + v;
  c = t;
  d = dt;
  return true;
}

__global__ __launch_bounds__(NTHR) void k_accs_walk_sources(AccShard s) {
  // lane-refill walk (see k_acc_walk_unit)
  const uint64_t n = (uint64_t)s.w * s.h;
  const uint64_t wave = ((uint64_t)blockIdx.x * NTHR + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint64_t next = wave * WALK_CHUNK;
  const uint64_t end = next + WALK_CHUNK < n ? next + WALK_CHUNK : n;
  if (next >= n) return;
  bool active = false;
  uint32_t c = 0;
  unsigned long long v = 0;
  uint8_t d = 0;
  for (;;) {
    const unsigned long long idle = __ballot(!active);
    if (next < end && idle) {
      const uint64_t my = next + (uint64_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (!active && my < end) {
        const uint8_t dd = s.dirs[my];
        if (dd != s.nodata && (s.word[my] >> 56) == SRC) { active = true; c = (uint32_t)my; v = 1ull; d = dd; }
      }
      next += (uint64_t)__popcll(idle);
    } else if (idle == ~0ull) {
      break;
    }
    if (active) active = accs_step(s, c, v, d);
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

// ---- one exchange (reference programs/parallel_d8_accum/main.cpp: :373-464 the tile's own accumulation, :270-334 where the
// flow entering at a perimeter cell leaves the tile again, :344-370 FollowPathAdd) -------------------------------------
// With local pending counts every cell of a loop-free block completes in begin, and the outboxes hold what the block's
// own cells send across each cut.  What is missing is the flow that ENTERS at a cut-row cell; it travels down that
// cell's path and leaves again at a known place -- its link.  The ranks gather (outbox, links) once, every rank solves
// the small forest over the cut-row cells, and the inflow of each entry is added along its path.
// link of a cut-row cell: (leaves across the lower cut ? 1 << 31 : 0) | receiving column, or -1 (the path ends inside).
__global__ __launch_bounds__(NTHR) void k_accs_links(AccShard s, int32_t *links) {
  const int i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= 2 * s.w) return;
  int x = i % s.w, y = i < s.w ? 0 : s.h - 1;
  int32_t link = -1;
  uint8_t d = s.dirs[(size_t)y * s.w + x];
  if (d != s.nodata) {
    for (uint64_t steps = 0, cap = (uint64_t)s.w * s.h + 1; steps < cap; steps++) {
      if (d < 1 || d > 8) break;
      const int nx = x + d8dx(d), ny = y + d8dy(d);
      if (nx < 0 || nx >= s.w) break;
      if (ny < 0) { if (s.above && s.above[nx] != s.nodata) link = nx; break; }
      if (ny >= s.h) { if (s.below && s.below[nx] != s.nodata) link = (int32_t)(0x80000000u | (uint32_t)nx); break; }
      const uint8_t dn = s.dirs[(size_t)ny * s.w + nx];
      if (dn == s.nodata) break;
      // an incomplete cell lies on or below a direction loop: stop (the protocol falls back anyway) instead of circling
      const unsigned long long k = s.word[(size_t)ny * s.w + nx] >> 56;
      if (k != 0 && k != SRC) break;
      x = nx; y = ny; d = dn;
    }
  }
  links[i] = link;
}

// what the block's own (complete) cut-row cells send across the cuts, in the outbox format of the walk
__global__ __launch_bounds__(NTHR) void k_accs_local_outbox(AccShard s) {
  const int i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= 2 * s.w) return;
  if (s.h == 1 && i >= s.w) return;   // a single row is the first and the last row: once
  const int x = i % s.w, y = i < s.w ? 0 : s.h - 1;
  const uint8_t d = s.dirs[(size_t)y * s.w + x];
  if (d == s.nodata || d < 1 || d > 8) return;
  const int nx = x + d8dx(d), ny = y + d8dy(d);
  if (nx < 0 || nx >= s.w || (ny >= 0 && ny < s.h)) return;
  const unsigned long long v = s.word[(size_t)y * s.w + x];
  if ((v >> 56) != 0) return;   // incomplete (a direction loop upstream): never handed on
  if (ny < 0) { if (s.above && s.above[nx] != s.nodata) atomicAdd(&s.out_top[nx], (v & LOWMASK) + CNT1); }
  else if (s.below && s.below[nx] != s.nodata) atomicAdd(&s.out_bottom[nx], (v & LOWMASK) + CNT1);
}

__global__ __launch_bounds__(NTHR) void k_accs_count_pending(AccShard s, unsigned long long *count) {
  const uint64_t n = (uint64_t)s.w * s.h, stride = (uint64_t)gridDim.x * NTHR;
  uint32_t mine = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const unsigned long long k = s.word[c] >> 56;
    mine += (s.dirs[c] != s.nodata && k != 0 && k != SRC) ? 1u : 0u;
  }
  const unsigned long long b = __ballot(mine != 0);
  if (b) {   // rare: direction loops only
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(count, (unsigned long long)mine);
  }
}

// in_top[x] / in_bottom[x]: the flow entering at cell x of the first / last row from outside; added to every cell of
// that cell's path inside the block (all cells are complete: plain sums, any order)
__global__ __launch_bounds__(NTHR) void k_accs_add_paths(AccShard s, const unsigned long long *__restrict__ in_top,
                                                         const unsigned long long *__restrict__ in_bottom) {
  const int i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= 2 * s.w) return;
  const unsigned long long *in = i < s.w ? in_top : in_bottom;
  if (!in) return;
  int x = i % s.w, y = i < s.w ? 0 : s.h - 1;
  const unsigned long long v = in[x] & LOWMASK;
  if (v == 0) return;
  uint8_t d = s.dirs[(size_t)y * s.w + x];
  if (d == s.nodata) return;
  for (uint64_t steps = 0, cap = (uint64_t)s.w * s.h + 1; steps < cap; steps++) {
    atomicAdd(&s.word[(size_t)y * s.w + x], v);
    if (d < 1 || d > 8) return;
    const int nx = x + d8dx(d), ny = y + d8dy(d);
    if (nx < 0 || nx >= s.w || ny < 0 || ny >= s.h) return;
    const uint8_t dn = s.dirs[(size_t)ny * s.w + nx];
    if (dn == s.nodata) return;
    const unsigned long long k = s.word[(size_t)ny * s.w + nx] >> 56;
    if (k != 0 && k != SRC) return;   // (never with loop-free directions, the only ones this entry is for)
    x = nx; y = ny; d = dn;
  }
}

}  // namespace rdgpu

struct rdgpu_accum_shard {
  rdgpu::AccShard s;
  hipStream_t stream = nullptr;
  int slot = -1;   // which set of workspace buffers this shard holds (several shards may be alive on one device)
};

namespace rdgpu {

// The per-cell words live in the workspace (a hipMalloc / hipFree of 8 B per cell around every accumulation cost more
// than the accumulation's own exchange); a live shard owns one numbered set of buffers until finish / free.
static std::vector<bool> &accs_slots() {   // of the current device (used under that device's API lock)
  static std::mutex mu;
  static std::map<int, std::vector<bool>> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  return per_device[dev];
}

static void accs_free(rdgpu_accum_shard *a) {
  if (!a) return;
  if (a->slot >= 0 && (size_t)a->slot < accs_slots().size()) {
    accs_slots()[a->slot] = false;
    Workspace::get().unpin();
  }
  delete a;
}

static rdgpu_accum_shard *accs_begin(const uint8_t *d_dirs, uint8_t nodata, int w, int h, const uint8_t *d_above,
                                     const uint8_t *d_below, hipStream_t st, bool local = false) {
  if (!d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_begin: null pointer");
  check_dims(w, h, "rdgpu_accum_shard_begin");
  rdgpu_accum_shard *a = new rdgpu_accum_shard();
  try {
    a->stream = st;
    const uint64_t n = (uint64_t)w * h;
    std::vector<bool> &slots = accs_slots();
    size_t k = 0;
    while (k < slots.size() && slots[k]) k++;
    if (k == slots.size()) slots.push_back(false);
    slots[k] = true;
    a->slot = (int)k;
    Workspace::get().pin();   // (the words live in the workspace: release_workspace() is refused while the shard is alive)
    int nbuf = 0;
    auto alloc = [&](size_t bytes) {
      const std::string name = "accum.shard" + std::to_string(k) + "." + std::to_string(nbuf++);
      return Workspace::get().buf(name.c_str(), bytes);
    };
    a->s = AccShard{d_dirs, d_above, d_below, (unsigned long long *)alloc(n * 8), (unsigned long long *)alloc((size_t)w * 8),
                    (unsigned long long *)alloc((size_t)w * 8), w, h, nodata, (uint8_t)(local ? 1 : 0)};
    RD_HIP(hipMemsetAsync(a->s.out_top, 0, (size_t)w * 8, st));
    RD_HIP(hipMemsetAsync(a->s.out_bottom, 0, (size_t)w * 8, st));
    if (local) {
      // the block on its own is a raster like any other: the tile links of d8_flow_accum, written as words
      d8_flow_accum_device<unsigned long long>(d_dirs, nodata, w, h, a->s.word, st);
      RD_LAUNCH("accum.shard_outbox", k_accs_local_outbox, dim3((2 * w + NTHR - 1) / NTHR), dim3(NTHR), 0, st, a->s);
      return a;
    }
    RD_LAUNCH("accum.shard_init", k_accs_init, dim3(sgrid(n)), dim3(NTHR), 0, st, a->s);
    RD_LAUNCH("accum.shard_walk", k_accs_walk_sources, dim3((uint32_t)(((n + WALK_CHUNK - 1) / WALK_CHUNK + 3) / 4)),
              dim3(NTHR), 0, st, a->s);
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

// One-exchange protocol: local pending counts, then outbox as usual, rdgpu_accum_shard_links, one gather, the solve
// (richdem_amd/sharded.py accum_link_solve), rdgpu_accum_shard_add_paths, finish.
extern "C" int rdgpu_accum_shard_begin_local(const uint8_t *d_dirs, uint8_t dir_nodata, int w, int h, const uint8_t *d_row_above,
                                             const uint8_t *d_row_below, void *stream, rdgpu_accum_shard **out) {
  return guarded([&] {
    if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_begin_local: null output handle");
    *out = accs_begin(d_dirs, dir_nodata, w, h, d_row_above, d_row_below, (hipStream_t)stream, true);
  });
}

// d_links[2][w] <- where the flow entering at each cell of the first / last row leaves the block again;
// *d_pending <- the number of cells the local phase could not complete (direction loops: use the iterated protocol)
extern "C" int rdgpu_accum_shard_links(rdgpu_accum_shard *a, int32_t *d_links, unsigned long long *d_pending) {
  return guarded([&] {
    if (!a || !d_links || !d_pending) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_links: null pointer");
    const uint64_t n = (uint64_t)a->s.w * a->s.h;
    RD_HIP(hipMemsetAsync(d_pending, 0, sizeof(unsigned long long), a->stream));
    RD_LAUNCH("accum.shard_links", k_accs_links, dim3((2 * a->s.w + NTHR - 1) / NTHR), dim3(NTHR), 0, a->stream, a->s, d_links);
    RD_LAUNCH("accum.shard_pending", k_accs_count_pending, dim3(sgrid(n)), dim3(NTHR), 0, a->stream, a->s, d_pending);
  });
}

// d_in_top[w] / d_in_bottom[w] (NULL: none): the flow entering at the first / last row from outside
extern "C" int rdgpu_accum_shard_add_paths(rdgpu_accum_shard *a, const unsigned long long *d_in_top,
                                           const unsigned long long *d_in_bottom) {
  return guarded([&] {
    if (!a) throw Error(RDGPU_ERR_ARG, "rdgpu_accum_shard_add_paths: null handle");
    RD_LAUNCH("accum.shard_add_paths", k_accs_add_paths, dim3((2 * a->s.w + NTHR - 1) / NTHR), dim3(NTHR), 0, a->stream, a->s,
              d_in_top, d_in_bottom);
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
  extern "C" int rdgpu_d8_flow_accum_multi_##SUF(const uint8_t *, uint8_t, int, int, A *, const int *, int); \
  extern "C" int rdgpu_d8_flow_accum_##SUF(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area) {     \
    const std::vector<int> devs = env_devices();   /* RDGPU_DEVICES: the node's GPUs, csrc/multi.hip */     \
    if (devs.size() > 1 && h >= (int)devs.size())                                                            \
      return rdgpu_d8_flow_accum_multi_##SUF(dirs, nodata, w, h, area, devs.data(), (int)devs.size());       \
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
  }                                                                                                          \
  extern "C" int rdgpu_fa_d8_unit_##SUF(const T *dem, T nodata, int w, int h, double *accum_out) {           \
    return guarded([&] { fa_d8_host<T>(dem, nodata, w, h, accum_out, true); });                              \
  }                                                                                                          \
  extern "C" int rdgpu_fa_d8_unit_dev_##SUF(const T *d_dem, T nodata, int w, int h, double *d_accum_out, void *stream) { \
    return guarded([&] { fa_d8_device<T>(d_dem, nodata, w, h, d_accum_out, (hipStream_t)stream, true); });   \
  }
RD_FA_API(u8, uint8_t)
RD_FA_API(i16, int16_t)
RD_FA_API(u16, uint16_t)
RD_FA_API(i32, int32_t)
RD_FA_API(u32, uint32_t)
RD_FA_API(f32, float)
RD_FA_API(f64, double)
RD_FA_API(i8, int8_t)
RD_FA_API(i64, int64_t)
RD_FA_API(u64, uint64_t)
