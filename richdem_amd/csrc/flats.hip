// flats.hip -- Barnes (2014) flat resolution on MI355X.
//
// Replaces barnes_flat_resolution_d8(elevations, flowdirs, alter=false)
// (reference include/richdem/flats/flat_resolution.hpp:587-605) =
//   d8_flow_directions                       (flowmet/d8_flowdirs.hpp:96-123)       -> flowdirs.hip
//   resolve_flats_barnes                     (flat_resolution.hpp:447-517)
//      find_flat_edges :381-418              -> k_flat_classify   (3x3 stencil -> flag byte, compacted lists)
//      label_this :331-355                   -> k_ccl_*           (components of the equal-elevation 8-graph:
//                                                in LDS per tile, lock-free union-find across tile borders;
//                                                root = lowest cell index)
//      BuildAwayGradient :152-198            -> k_relax_bits<1>   (breadth-first search on bitmaps, a wavefront per 64 x 64
//      BuildTowardsCombinedGradient :241-298 -> k_relax_bits<2>    tile to its local fixed point: rounds over the active tiles
//                                                while the front is wide, then ONE resident launch that pulls tiles from
//                                                queues -- k_relax_bits_async; the away search runs on a side stream
//                                                beside the towards search's tail)
//   d8_flow_flats / d8_masked_FlowDir :96-116, :42-65 -> k_flat_dirs (full mask), k_flat_dirs_levels (directions-only entry)
// ResolveFlatsEpsilon (flats/flats.hpp:21-28) runs the same searches over FindFlats' cells, with the labels (k_ccl_*) on a
// side stream beside the searches' tail.
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
#include <functional>
#include <type_traits>
#include <cstdlib>
#include <climits>
#include <cstring>

namespace rdgpu {

constexpr int NTHR = 256;

__device__ __forceinline__ int fdx(int n) { return (n == 1 || n == 2 || n == 8) ? -1 : (n >= 4 && n <= 6) ? 1 : 0; }
__device__ __forceinline__ int fdy(int n) { return (n >= 2 && n <= 4) ? -1 : (n >= 6 && n <= 8) ? 1 : 0; }

// NOTE on atomics: same-address device atomics serialise at ~12 ns each on MI355X, so nothing in
// this file funnels per-cell or per-wave work through one counter: lists are built with
// count / scan / fill compaction, BFS appends use ONE atomic per block, flat heights are reduced in
// reverse level order behind a pre-check.

constexpr uint8_t F_LOW = 1, F_HIGH = 2, F_NOFLOW = 4;
constexpr uint8_t F_NEAR = 8;       // NO_FLOW cell next to an equal-elevation cell WITH a direction (a low edge): towards level 2
constexpr uint8_t DIR_GHOST_NOFLOW = 254;   // row-block shards: NO_FLOW cell of a ghost row (feeds, is not relaxed here)

// ------------------------------------------------------------------------------------------
// find_flat_edges (flat_resolution.hpp:381-418) -> one flag byte per cell
// ------------------------------------------------------------------------------------------
constexpr int SW = 64, SH = 16, SLW = SW + 2, SLH = SH + 2;   // stencil tiles (classification, masked directions)

// 64 x 32 tiles; a wavefront owns a band of 8 consecutive rows, a lane one column, and the 3 x 3 windows of elevations
// and directions slide down the column in registers (6 LDS reads per cell instead of 18: the LDS, not HBM, bounded the
// first version of this kernel -- 7.9 ms at S3 for 9.6 GB).
constexpr int KLH = 32, KLLH = KLH + 2;
template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_classify(const T *__restrict__ z, const uint8_t *__restrict__ dirs,
                                                        int w, int h, uint8_t *__restrict__ flags, uint32_t tilesX,
                                                        uint32_t ntiles) {
  __shared__ T sz[KLLH * SLW];
  __shared__ uint8_t sdir[KLLH * SLW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * KLH;
  if (window_inside(x0, y0, w, h, SW, KLH, 1)) {
    stage_window_inside<T, SW, KLH, 1, SLW, NTHR>(z, w, x0, y0, sz);
    stage_window_inside<uint8_t, SW, KLH, 1, SLW, NTHR>(dirs, w, x0, y0, sdir);
  } else {
    constexpr int IPT = (KLLH * SLW + NTHR - 1) / NTHR;
    T zv[IPT];
    uint8_t dv[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {   // all loads of the thread in flight together (clamped addresses)
      const int i = min((int)threadIdx.x + r * NTHR, KLLH * SLW - 1);
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      zv[r] = z[(size_t)gy * w + gx];
      dv[r] = dirs[(size_t)gy * w + gx];
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i >= KLLH * SLW) continue;
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
      const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
      sz[i] = zv[r];
      sdir[i] = in ? dv[r] : (uint8_t)255;   // outside the raster: skipped like NoData
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & (SW - 1), yb = (int)(threadIdx.x >> 6) * (KLH / 4);
  const int gx = x0 + lx;
  T z0[3], z1[3], z2[3];
  uint8_t d0[3], d1[3], d2[3];
  bool v0[3], v1[3], v2[3], n0[3], n1[3], n2[3];
#pragma unroll
  for (int e = 0; e < 3; e++) {
    z0[e] = sz[yb * SLW + lx + e]; z1[e] = sz[(yb + 1) * SLW + lx + e];
    d0[e] = sdir[yb * SLW + lx + e]; d1[e] = sdir[(yb + 1) * SLW + lx + e];
    v0[e] = d0[e] != 255; n0[e] = d0[e] == 0; v1[e] = d1[e] != 255; n1[e] = d1[e] == 0;
  }
#pragma unroll
  for (int j = 0; j < KLH / 4; j++) {
    const int ly = yb + j, gy = y0 + ly;
#pragma unroll
    for (int e = 0; e < 3; e++) {
      z2[e] = sz[(ly + 2) * SLW + lx + e]; d2[e] = sdir[(ly + 2) * SLW + lx + e];
      v2[e] = d2[e] != 255; n2[e] = d2[e] == 0;
    }
    uint8_t f = 0;
    const uint8_t d = d1[1];
    if (d != 255) {
      const bool noflow = d == 0;
      if (noflow) f = F_NOFLOW;
      // a cell WITH flow can only be a low edge if some neighbour is NO_FLOW; a NO_FLOW cell is always
      // interior (edge cells always get a direction), so its 8 neighbours exist
      const T e = z1[1];
      // three questions about the 8 neighbours, accumulated without branches (the per-neighbour early-outs compiled
      // into ~14 exec-mask branches and ~130 scalar mask operations per cell): is one of them higher (:409-411), is
      // one an equal NO_FLOW cell (:406-408), is one an equal cell WITH a direction (a low edge of this cell's flat)
      // (lane masks: the compares are the only vector instructions, the logic runs on the scalar unit; whether a
      // neighbour is valid / NO_FLOW is found once per staged cell -- v?[], n?[] slide down with the window)
      bool higher = false, eq_noflow = false, eq_flow = false;
      auto nb = [&](T zn, bool valid, bool nf) {
        const bool eq = zn == e;
        higher |= valid & !eq;   // (only asked of a NO_FLOW centre, and a valid neighbour of one is never LOWER -- the cell would
                                 // drain there: "not equal" is "higher", one compare per neighbour instead of two)
        eq_noflow |= valid & eq & nf;
        eq_flow |= valid & eq & !nf;
      };
      nb(z0[0], v0[0], n0[0]); nb(z0[1], v0[1], n0[1]); nb(z0[2], v0[2], n0[2]);
      nb(z1[0], v1[0], n1[0]); nb(z1[2], v1[2], n1[2]);
      nb(z2[0], v2[0], n2[0]); nb(z2[1], v2[1], n2[1]); nb(z2[2], v2[2], n2[2]);
      if (noflow ? higher : eq_noflow) f |= noflow ? F_HIGH : F_LOW;
      if (noflow && eq_flow) f |= F_NEAR;
    }
    if (gx < w && gy < h) flags[(size_t)gy * w + gx] = f;
#pragma unroll
    for (int e = 0; e < 3; e++) { z0[e] = z1[e]; z1[e] = z2[e]; d0[e] = d1[e]; d1[e] = d2[e]; v0[e] = v1[e]; v1[e] = v2[e]; n0[e] = n1[e]; n1[e] = n2[e]; }
  }
}

// ---- stream compaction of "flags[c] & mask" into a list of cell indices: count / scan / fill ----
// d8_flow_directions and the classification in ONE pass over the DEM (the directions-only entry): the tile's elevations
// with a halo of two, the directions of the tile and its ring from them (d8_FlowDir, flowmet/d8_flowdirs.hpp:32-74: the
// rule of k_flowdirs<T, MODE_D8>), written for the tile and kept in LDS for the flags.  The two kernels it replaces read
// the DEM twice and the directions once more: 6.15 -> 3.6 ms at S3.
constexpr int FZW = SW + 4, FZH = KLH + 4;   // staged elevations; the directions use the SLW x KLLH arrays of k_flat_classify
template <class T>
__device__ __forceinline__ uint8_t d8_dir_cell(const T *sz, int zx, int zy, int gx, int gy, int w, int h, T nodata) {
  // (zx, zy): the cell in the staged elevations; its 8 neighbours are staged too
  if (gx < 0 || gy < 0 || gx >= w || gy >= h) return 255;   // outside the raster: skipped like NoData by the classification
  const T e = sz[zy * FZW + zx];
  if (e == nodata) return 255;
  if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) {   // :37-54
    if (gx == 0 && gy == 0) return 2;
    if (gx == 0 && gy == h - 1) return 8;
    if (gx == w - 1 && gy == 0) return 4;
    if (gx == w - 1 && gy == h - 1) return 6;
    if (gx == 0) return 1;
    if (gx == w - 1) return 5;
    return gy == 0 ? 3 : 7;
  }
  T m = e;
  int dir = 0;
  bool diag = false;   // the choice so far is a diagonal (see k_flowdirs)
#pragma unroll
  for (int n = 1; n <= 8; n++) {   // :63-71
    const T v = sz[(zy + fdy(n)) * FZW + zx + fdx(n)];
    const bool take = (n & 1) ? ((v < m) | ((v == m) & diag)) : (v < m);
    m = take ? v : m;
    dir = take ? n : dir;
    diag = (n & 1) ? (diag & !take) : (diag | take);
  }
  return (uint8_t)dir;
}

// NEARDIRS (r05): a NO_FLOW cell NEXT TO a low edge of its own flat gets its FINAL direction here.  d8_masked_FlowDir
// (:42-65) sends it to a low edge -- mask 2, below every NO_FLOW cell's -- and which one only depends on WHICH neighbours
// are equal-elevation cells with a direction: the first of them in neighbour order, replaced by the first odd-numbered one
// after it if it is a diagonal (the same tie rule as everywhere).  That is known here and nowhere later without the
// elevations, so the pass after the searches (k_flat_dirs_q) reads no DEM at all.
// FindFlats' pseudo direction of a staged cell (flats/find_flats.hpp:29-69; see k_find_flats): 0 = flat, 1 = not, 255 = NoData
template <class T>
__device__ __forceinline__ uint8_t find_flats_cell(const T *sz, int zx, int zy, int gx, int gy, int w, int h, T nodata) {
  if (gx < 0 || gy < 0 || gx >= w || gy >= h) return 255;
  const T e = sz[zy * FZW + zx];
  if (e == nodata) return 255;
  if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) return 1;
  uint8_t f = 0;
#pragma unroll
  for (int n = 1; n <= 8; n++) {
    const T v = sz[(zy + fdy(n)) * FZW + zx + fdx(n)];
    if (v < e || v == nodata) f = 1;
  }
  return f;
}

// FINDFLATS (r05, ResolveFlatsEpsilon's lean path): the "directions" are FindFlats' pseudo raster and are NOT written -- nothing
// but this classification reads them there: k_find_flats + k_flat_classify (two passes over the DEM, the pseudo raster
// written and read back) in one.
// BITMAPS (r06, the plane engine): instead of a flag byte per cell, the rows of three bitmaps per 64 x 64 search tile -- the
// cells without a direction (M), those of them next to a low edge (the towards seeds), the high edges (the away seeds) --
// and the edge counts: what k_bits_prepare would make of the flags, without writing and re-reading them (1 B -> 3 bits per
// cell; the start kernels of the two searches read 1.5 KB per tile instead of 4 KB, without byte loads).
struct ClassBitmaps {
  unsigned long long *rows = nullptr;   // three bitmaps of nrows = search tiles x 64 row words each: M, near, high
  uint32_t *counts = nullptr;           // 256 stripes of (low edges, high edges, NO_FLOW cells)
  uint32_t nrows = 0, tilesXb = 0;
};
template <class T, bool NEARDIRS = false, bool FINDFLATS = false, bool BITMAPS = false>
__global__ __launch_bounds__(NTHR) void k_dirs_classify(const T *__restrict__ z, T nodata, uint8_t *__restrict__ dirs,
                                                        uint8_t *__restrict__ flags, int w, int h, uint32_t tilesX, uint32_t ntiles,
                                                        ClassBitmaps bm = ClassBitmaps{}) {
  __shared__ T sz[FZH * FZW];
  __shared__ uint8_t sdir[KLLH * SLW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * KLH;
  if (window_inside(x0, y0, w, h, SW, KLH, 2)) {
    stage_window_inside<T, SW, KLH, 2, FZW, NTHR>(z, w, x0, y0, sz);
  } else {
    constexpr int IPT = (FZH * FZW + NTHR - 1) / NTHR;
    T zv[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {   // all loads of the thread in flight together (clamped addresses)
      const int i = min((int)threadIdx.x + r * NTHR, FZH * FZW - 1);
      const int ly = i / FZW, lx = i - ly * FZW;
      const int gx = min(max(x0 - 2 + lx, 0), w - 1), gy = min(max(y0 - 2 + ly, 0), h - 1);
      zv[r] = z[(size_t)gy * w + gx];
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i < FZH * FZW) sz[i] = zv[r];
    }
  }
  __syncthreads();
  // directions of the tile and its ring (SLW x KLLH cells).  The 64 columns of the tile over all KLLH rows: a wavefront a
  // band of rows, a lane a column, the 3 x 3 window slides down in registers; the ring's two columns cell by cell.
  {
    const int lane = threadIdx.x & 63, band = threadIdx.x >> 6;
    const int r_lo = band < 2 ? band * 9 : 18 + (band - 2) * 8, r_n = band < 2 ? 9 : 8;   // 9 + 9 + 8 + 8 = KLLH rows
    static_assert(KLLH == 34 && NTHR == 256, "the bands above");
    const int gx = x0 + lane;
    T r0[3], r1[3], r2[3];
#pragma unroll
    for (int e = 0; e < 3; e++) { r0[e] = sz[r_lo * FZW + lane + 1 + e]; r1[e] = sz[(r_lo + 1) * FZW + lane + 1 + e]; }
    for (int j = 0; j < r_n; j++) {
      const int ry = r_lo + j, gy = y0 - 1 + ry;
#pragma unroll
      for (int e = 0; e < 3; e++) r2[e] = sz[(ry + 2) * FZW + lane + 1 + e];
      const T nbv[9] = {r1[1], r1[0], r0[0], r0[1], r0[2], r1[2], r2[2], r2[1], r2[0]};
      const T e = r1[1];
      int dir = 0;
      const bool in = gx < w && gy >= 0 && gy < h;
      const bool edge = gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1;
      T m = e;
      bool diag = false;   // the choice so far is a diagonal (see k_flowdirs)
#pragma unroll
      for (int n = 1; n <= 8; n++) {   // :63-71 (an edge cell's result is replaced below: its window may lie outside)
        const T v = nbv[n];
        const bool take = (n & 1) ? ((v < m) | ((v == m) & diag)) : (v < m);
        m = take ? v : m;
        dir = take ? n : dir;
        diag = (n & 1) ? (diag & !take) : (diag | take);
      }
      if (edge)   // :37-54
        dir = (gx == 0 && gy == 0) ? 2 : (gx == 0 && gy == h - 1) ? 8 : (gx == w - 1 && gy == 0) ? 4 : (gx == w - 1 && gy == h - 1) ? 6
              : gx == 0 ? 1 : gx == w - 1 ? 5 : gy == 0 ? 3 : 7;
      if (FINDFLATS) {   // find_flats.hpp:48-63: an edge cell is no flat; a lower neighbour or a NoData neighbour makes a cell no flat
        bool notflat = edge;
#pragma unroll
        for (int n = 1; n <= 8; n++) notflat |= (nbv[n] < e) | (nbv[n] == nodata);
        dir = notflat ? 1 : 0;
      }
      if (e == nodata) dir = 255;
      if (!in) dir = 255;
      sdir[ry * SLW + lane + 1] = (uint8_t)dir;
      if (!FINDFLATS && in && ry >= 1 && ry <= KLH) dirs[(size_t)gy * w + gx] = (uint8_t)dir;
#pragma unroll
      for (int e2 = 0; e2 < 3; e2++) { r0[e2] = r1[e2]; r1[e2] = r2[e2]; }
    }
    if (threadIdx.x < 2 * KLLH) {
      const int ly = (int)threadIdx.x % KLLH, lx = threadIdx.x < KLLH ? 0 : SLW - 1;
      sdir[ly * SLW + lx] = FINDFLATS ? find_flats_cell<T>(sz, lx + 1, ly + 1, x0 - 1 + lx, y0 - 1 + ly, w, h, nodata)
                                      : d8_dir_cell<T>(sz, lx + 1, ly + 1, x0 - 1 + lx, y0 - 1 + ly, w, h, nodata);
    }
  }
  __syncthreads();
  // (r05, measured twice: classifying on the VECTOR unit -- x = bits(n) ^ bits(c), masks from the direction bytes with adds and
  // shifts, min / OR over the eight -- removes the ~100 scalar instructions per row that the lane-mask form below leaves
  // (8.3e9 scalar against 8.2e9 vector instructions per launch at S3, profiles/r05m_path40k_sq_summary.csv) and is NOT
  // faster: 5.95 ms against 5.2-5.5, profiles/r05n_flats_ab.txt.  The scalar unit issues beside the vector unit, from other
  // wavefronts; the sum of the two is not what bounds the kernel.)
  // the flags: k_flat_classify's window over (elevations, directions)
  const int lx = threadIdx.x & (SW - 1), yb = (int)(threadIdx.x >> 6) * (KLH / 4);
  const int gx = x0 + lx;
  T z0[3], z1[3], z2[3];
  uint8_t d0[3], d1[3], d2[3];
  bool v0[3], v1[3], v2[3], n0[3], n1[3], n2[3];
  uint32_t nlow = 0;   // (BITMAPS: low edges, per lane)
  unsigned long long keepM = 0, keepN = 0, keepH = 0;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    z0[e] = sz[(yb + 1) * FZW + lx + 1 + e]; z1[e] = sz[(yb + 2) * FZW + lx + 1 + e];
    d0[e] = sdir[yb * SLW + lx + e]; d1[e] = sdir[(yb + 1) * SLW + lx + e];
    v0[e] = d0[e] != 255; n0[e] = d0[e] == 0; v1[e] = d1[e] != 255; n1[e] = d1[e] == 0;
  }
#pragma unroll
  for (int j = 0; j < KLH / 4; j++) {
    const int ly = yb + j, gy = y0 + ly;
#pragma unroll
    for (int e = 0; e < 3; e++) {
      z2[e] = sz[(ly + 3) * FZW + lx + 1 + e]; d2[e] = sdir[(ly + 2) * SLW + lx + e];
      v2[e] = d2[e] != 255; n2[e] = d2[e] == 0;
    }
    uint8_t f = 0;
    bool b_noflow = false, b_high = false, b_near = false, b_low = false;   // (BITMAPS: the flags as lane masks, never a byte)
    const uint8_t d = d1[1];
    if (d != 255) {
      const bool noflow = d == 0;
      if (noflow) f = F_NOFLOW;
      const T e = z1[1];
      // (lane masks: the compares are the only vector instructions, the logic runs on the scalar unit; whether a
      // neighbour is valid / NO_FLOW is found once per staged cell -- v?[], n?[] slide down with the window)
      bool higher = false, eq_noflow = false, eq_flow = false;
      auto nb = [&](T zn, bool valid, bool nf) {
        const bool eq = zn == e;
        higher |= valid & !eq;   // (only asked of a NO_FLOW centre, and a valid neighbour of one is never LOWER -- the cell would
                                 // drain there: "not equal" is "higher", one compare per neighbour instead of two)
        eq_noflow |= valid & eq & nf;
        eq_flow |= valid & eq & !nf;
      };
      nb(z0[0], v0[0], n0[0]); nb(z0[1], v0[1], n0[1]); nb(z0[2], v0[2], n0[2]);
      nb(z1[0], v1[0], n1[0]); nb(z1[2], v1[2], n1[2]);
      nb(z2[0], v2[0], n2[0]); nb(z2[1], v2[1], n2[1]); nb(z2[2], v2[2], n2[2]);
      if (noflow ? higher : eq_noflow) f |= noflow ? F_HIGH : F_LOW;
      if (noflow && eq_flow) f |= F_NEAR;
      if (BITMAPS) { b_noflow = noflow; b_high = noflow & higher; b_near = noflow & eq_flow; b_low = !noflow & eq_noflow; }
      if (NEARDIRS && noflow && eq_flow) {
        // neighbours 1..8 in the 234/105/876 numbering; le: an equal-elevation cell with a direction (a low edge of this flat)
        auto lowedge = [&](T zn, bool valid, bool nf) -> bool { return valid && zn == e && !nf; };
        const bool le[9] = {false,
                            lowedge(z1[0], v1[0], n1[0]), lowedge(z0[0], v0[0], n0[0]), lowedge(z0[1], v0[1], n0[1]),
                            lowedge(z0[2], v0[2], n0[2]), lowedge(z1[2], v1[2], n1[2]), lowedge(z2[2], v2[2], n2[2]),
                            lowedge(z2[1], v2[1], n2[1]), lowedge(z2[0], v2[0], n2[0])};
        int nd = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
          const bool take = le[k] && (nd == 0 || ((nd & 1) == 0 && (k & 1) == 1));
          nd = take ? k : nd;
        }
        if (gx < w && gy < h) dirs[(size_t)gy * w + gx] = (uint8_t)nd;   // (a NO_FLOW cell is interior: its window is complete)
      }
    }
    if (BITMAPS) {   // (a cell outside the raster has d == 255: f == 0)
      const unsigned long long mb = __ballot(b_noflow), nb_ = __ballot(b_near), hb = __ballot(b_high);
      // (the low edges are in no bitmap: counted here, per lane in a vector register -- as a wave-uniform sum it costs the kernel's
      // scalar registers, 37 spilled; the high edges and NO_FLOW cells are counted from the bitmaps by k_planes_prepare_b)
      nlow += b_low ? 1u : 0u;
      if (lx == j) { keepM = mb; keepN = nb_; keepH = hb; }   // lane j keeps row j of the wavefront's band: one store of eight words per bitmap below
    } else if (gx < w && gy < h) flags[(size_t)gy * w + gx] = f;
#pragma unroll
    for (int e = 0; e < 3; e++) { z0[e] = z1[e]; z1[e] = z2[e]; d0[e] = d1[e]; d1[e] = d2[e]; v0[e] = v1[e]; v1[e] = v2[e]; n0[e] = n1[e]; n1[e] = n2[e]; }
  }
  if (BITMAPS) {
    static_assert(SW == 64 && KLH / 4 == 8, "a classification tile is one search tile wide, a wavefront's band eight rows");
    if (lx < KLH / 4) {   // rows y0 + yb .. + 7: consecutive words of one search tile
      const int gy = y0 + yb + lx;
      unsigned long long *const row = bm.rows + ((size_t)(gy >> 6) * bm.tilesXb + (size_t)(x0 >> 6)) * 64 + (size_t)(gy & 63);
      row[0] = keepM; row[bm.nrows] = keepN; row[2 * (size_t)bm.nrows] = keepH;
    }
    // the counts, striped: same-address atomics serialise
    const unsigned long long lowlanes = __ballot(nlow != 0);
    if (lowlanes) {   // (rare: one wavefront in a few has a low edge)
      uint32_t lo = nlow;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) lo += __shfl_xor(lo, o, 64);
      if (lx == 0) atomicAdd(bm.counts + 3 * ((t * 4 + (threadIdx.x >> 6)) & 255u), lo);
    }
  }
}

constexpr int CPB = 4096;   // cells per block
__global__ __launch_bounds__(NTHR) void k_flag_count(const uint8_t *__restrict__ flags, uint8_t mask, uint64_t n,
                                                     uint32_t *__restrict__ counts) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t cnt = 0;
#pragma unroll 4
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t c = base + (uint64_t)j * NTHR + threadIdx.x;
    if (c < n && (flags[c] & mask)) cnt++;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// single workgroup: exclusive scan of counts[0..m) in place; total -> *total
__global__ __launch_bounds__(1024) void k_flag_scan(uint32_t *counts, uint32_t m, uint32_t *total) {
  __shared__ uint32_t part[1024];
  const uint32_t chunk = (m + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * chunk, hi = min(lo + chunk, m);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
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

__global__ __launch_bounds__(NTHR) void k_flag_fill(const uint8_t *__restrict__ flags, uint8_t mask, uint64_t n,
                                                    const uint32_t *__restrict__ offsets, uint32_t *__restrict__ out) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t run = offsets[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t c = base + (uint64_t)j * NTHR + threadIdx.x;
    const bool hit = c < n && (flags[c] & mask);
    const unsigned long long bal = __ballot(hit);
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
    if (hit) out[run + woff + rank] = (uint32_t)c;
    run += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// label_this (:331-355) as connected components of the equal-elevation 8-graph.
//   k_ccl_tile   components inside a 64x32 tile, entirely in LDS (min-label propagation + jumping; an LDS union-find
//                over the lower-index neighbours was measured 3x slower, 46 vs 15 ms: a lake tile is 2048 cells of ONE
//                component, all uniting at once)
//   k_ccl_border lock-free union-find across tile borders only (~5% of the cells touch the atomics)
//   k_ccl_flatten
// Parents always point to a LOWER cell index, so the global structure is acyclic under any interleaving
// and the root of a component is its lowest cell index.
// ------------------------------------------------------------------------------------------
constexpr int CW = 64, CH = 32;

// colZ / colL (optional): the tile's left and right columns (elevation, tile-level label), 2 * CH entries per tile, for
// k_ccl_border2 -- read back from the raster those columns cost a 128-byte line per row and array (25.6 GB at S3).
// A row outside the raster holds the label CCL_NOCELL.
constexpr uint32_t CCL_NOCELL = 0xFFFFFFFFu;
// an element of at most 32 bits as the 32-bit lane value a DPP move carries, and back
template <class T>
__device__ __forceinline__ uint32_t lane_bits(T v) {
  if constexpr (sizeof(T) == 4) { uint32_t u; __builtin_memcpy(&u, &v, 4); return u; }
  else return (uint32_t)v;
}
template <class T>
__device__ __forceinline__ T lane_from_bits(uint32_t u) {
  if constexpr (sizeof(T) == 4) { T v; __builtin_memcpy(&v, &u, 4); return v; }
  else return (T)u;
}
template <class T>
__global__ __launch_bounds__(NTHR) void k_ccl_tile(const T *__restrict__ z, uint32_t *__restrict__ L, int w, int h,
                                                   uint32_t tilesX, uint32_t ntiles, T *__restrict__ colZ, uint32_t *__restrict__ colL) {
  __shared__ T sz[CH * CW];
  __shared__ uint32_t lab[CH * CW];            // union-find parents over the tile's cells (a run's first cell stands for the run)
  __shared__ unsigned long long smask[CH];     // per row: the lanes that start a run
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * CW, y0 = (int)(t / tilesX) * CH;
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  constexpr int ROWS = CH / 4;
  const int gx = x0 + lx;
  // stage the tile; cells outside the raster get a value that matches nothing through the mask below
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j, gy = y0 + ly;
    T v = T();
    if (gx < w && gy < h) v = z[(size_t)gy * w + gx];
    sz[ly * CW + lx] = v;
  }
  __syncthreads();
  // A tile of ONE elevation (open water: 30 % of S3's tiles, 61 % of its NO_FLOW cells) is one component.
  {
    const bool inside = x0 + CW <= w && y0 + CH <= h;
    const T z00 = sz[0];
    bool same = inside;
#pragma unroll
    for (int j = 0; j < ROWS; j++) same &= sz[(band * ROWS + j) * CW + lx] == z00;
    if (__syncthreads_and(same)) {
      const uint32_t root = (uint32_t)y0 * (uint32_t)w + (uint32_t)x0;
#pragma unroll
      for (int j = 0; j < ROWS; j++) L[(size_t)(y0 + band * ROWS + j) * w + gx] = root;
      if (colZ && threadIdx.x < 2 * CH) {
        colZ[(size_t)t * (2 * CH) + threadIdx.x] = z00;
        colL[(size_t)t * (2 * CH) + threadIdx.x] = root;
      }
      return;
    }
  }
  // 8-bit mask of in-tile, in-raster neighbours of equal elevation (bit k-1 for neighbour k)
  uint32_t msk[ROWS];
  int any = 0;
  bool windowed = false;
  if constexpr (sizeof(T) <= 4) if (x0 + CW <= w && y0 + CH <= h) {
    windowed = true;
    // the tile lies inside the raster (block-uniform): the 3 x 3 window slides down the column, the columns beside it come
    // from the neighbouring lanes (DPP), the tile's borders are two constant masks -- 8 compares per cell instead of 8
    // clamped lookups with six bounds tests each (they were half of this kernel's instructions)
    auto left_of = [](T v) -> T {    // lane l receives lane l - 1's value
      return lane_from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(0, (int)lane_bits(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
    };
    auto right_of = [](T v) -> T {   // lane l receives lane l + 1's value
      return lane_from_bits<T>((uint32_t)__builtin_amdgcn_update_dpp(0, (int)lane_bits(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
    };
    const uint32_t colmask = lx == 0 ? ~0x83u : lx == CW - 1 ? ~0x38u : ~0u;   // no W / NW / SW, no NE / E / SE
    const int ly0 = band * ROWS;
    T b0 = ly0 > 0 ? sz[(ly0 - 1) * CW + lx] : T(), b1 = sz[ly0 * CW + lx];
    T a0 = left_of(b0), c0 = right_of(b0), a1 = left_of(b1), c1 = right_of(b1);
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const int ly = ly0 + j;
      const T b2 = ly + 1 < CH ? sz[(ly + 1) * CW + lx] : T();
      const T a2 = left_of(b2), c2 = right_of(b2);
      const T e = b1;
      uint32_t m = (a1 == e ? 1u : 0u) | (a0 == e ? 2u : 0u) | (b0 == e ? 4u : 0u) | (c0 == e ? 8u : 0u) | (c1 == e ? 16u : 0u) |
                   (c2 == e ? 32u : 0u) | (b2 == e ? 64u : 0u) | (a2 == e ? 128u : 0u);
      m &= colmask;
      if (ly == 0) m &= ~0x0Eu;          // (scalar) no NW / N / NE
      if (ly == CH - 1) m &= ~0xE0u;     // no SE / S / SW
      msk[j] = m;
      any |= m != 0;
      a0 = a1; b0 = b1; c0 = c1; a1 = a2; b1 = b2; c1 = c2;
    }
  }
  if (!windowed)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j, gy = y0 + ly;
    uint32_t m = 0;
    if (gx < w && gy < h) {
      const T e = sz[ly * CW + lx];
#pragma unroll
      for (int k = 1; k <= 8; k++) {   // clamped lookups and a mask, no branches (see k_flat_classify)
        const int nlx = lx + fdx(k), nly = ly + fdy(k);
        const bool in = (nlx >= 0) & (nlx < CW) & (nly >= 0) & (nly < CH) & (x0 + nlx < w) & (y0 + nly < h);
        const int cx = min(max(nlx, 0), CW - 1), cy = min(max(nly, 0), CH - 1);
        m |= (uint32_t)(in & (sz[cy * CW + cx] == e)) << (k - 1);
      }
    }
    msk[j] = m;
    any |= m != 0;
  }
  // r05: components from ROW RUNS.  (r01-r04: min-label propagation with pointer jumping, one barrier per sweep, a few dozen
  // sweeps on a lake's shore -- 16.7 ms at S3, the critical path of ResolveFlatsEpsilon.)  A row's runs of equal cells come
  // straight from a ballot (a cell without an equal W neighbour starts one): every cell knows its run's first cell at once.
  // Runs of consecutive rows are joined by a lock-free union-find over the runs' first cells, with the unions cut down
  // to the few that matter: a run's first cell looks at its equal NW and N neighbours, and ANY cell at its equal NE
  // neighbour only if that cell starts a run of the row above -- every other adjacency is implied (an upper cell that does
  // not start a run shares its run with the cell left of it, which the lane to the left or the run's first cell covers).
  // Open water: one union per row.  Then one find per cell.
  uint32_t lab0[ROWS];
  unsigned long long sm_[ROWS];
  const unsigned long long le = ~0ull >> (63 - lx);   // lanes <= this one
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    const unsigned long long st_ = __builtin_amdgcn_ballot_w64(!(msk[j] & 1u));   // (lane 0 always: its W bit is masked off)
    sm_[j] = st_;
    lab0[j] = (uint32_t)(ly * CW + (63 - __clzll((long long)(st_ & le))));
    lab[ly * CW + lx] = lab0[j];
    if (lx == 0) smask[ly] = st_;
  }
  const int any_b = __syncthreads_or(any);   // (block-uniform: does any cell of the tile have an equal neighbour)
  if (any_b) {
    auto lfind = [&](uint32_t x) {   // (parents only ever decrease; other wavefronts lower them meanwhile: relaxed atomic loads)
      uint32_t p_ = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (p_ != x) { x = p_; p_ = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      return x;
    };
    auto lunite = [&](uint32_t a_, uint32_t b_) {
      for (;;) {
        a_ = lfind(a_);
        b_ = lfind(b_);
        if (a_ == b_) return;
        if (a_ < b_) { const uint32_t t_ = a_; a_ = b_; b_ = t_; }   // the larger root goes under the smaller: acyclic under any interleaving
        const uint32_t old_ = atomicMin(&lab[a_], b_);
        if (old_ == a_) return;
        a_ = old_;
      }
    };
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const int ly = band * ROWS + j;
      const uint32_t m = msk[j];
      if (!(m & 0x0Eu)) continue;                                       // no equal cell in the row above (row 0: masked off)
      const unsigned long long su = j ? sm_[j - 1] : smask[ly - 1];    // the row above: its run starts
      auto run_of = [&](int ux) { return (uint32_t)((ly - 1) * CW + (63 - __clzll((long long)(su & (~0ull >> (63 - ux)))))); };
      if ((m & 8u) && (su >> (lx + 1) & 1ull)) lunite(lab0[j], (uint32_t)((ly - 1) * CW + lx + 1));
      if (!(m & 1u)) {
        if (m & 4u) lunite(lab0[j], run_of(lx));
        if (m & 2u) lunite(lab0[j], run_of(lx - 1));
      }
    }
  }
  __syncthreads();
  if (any_b) {   // every cell: its run's root; written back for the column records below
    uint32_t root[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      uint32_t x = lab0[j], p_ = lab[x];
      while (p_ != x) { x = p_; p_ = lab[x]; }
      root[j] = x;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++) lab[(band * ROWS + j) * CW + lx] = root[j];
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const int r = lab[ly * CW + lx];
    L[(size_t)gy * w + gx] = (uint32_t)(y0 + (r >> 6)) * (uint32_t)w + (uint32_t)(x0 + (r & (CW - 1)));
  }
  if (colZ && threadIdx.x < 2 * CH) {
    const int side = threadIdx.x / CH, ly = threadIdx.x % CH, cx = side ? CW - 1 : 0;
    const bool in = x0 + cx < w && y0 + ly < h;
    const int r = lab[ly * CW + cx];
    colZ[(size_t)t * (2 * CH) + threadIdx.x] = sz[ly * CW + cx];
    colL[(size_t)t * (2 * CH) + threadIdx.x] = in ? (uint32_t)(y0 + (r >> 6)) * (uint32_t)w + (uint32_t)(x0 + (r & (CW - 1))) : CCL_NOCELL;
  }
}

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

// unions across tile borders: lower-index neighbours (NW, N, NE, W) that live in another tile.  One thread per
// cell of a tile's top row, left column and right column (128 per tile) -- the other cells have all four
// lower-index neighbours inside their own tile.
template <class T>
__global__ __launch_bounds__(NTHR) void k_ccl_border(const T *__restrict__ z, uint32_t *L, int w, int h, uint32_t tilesX,
                                                     uint32_t ntiles) {
  const uint64_t total = (uint64_t)ntiles * (CW + 2 * CH), stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < total; i += stride) {
    const uint32_t t = (uint32_t)(i / (CW + 2 * CH));
    const int k = (int)(i % (CW + 2 * CH));
    const int tx = k < CW ? k : (k < CW + CH ? 0 : CW - 1);
    const int ty = k < CW ? 0 : (k < CW + CH ? k - CW : k - CW - CH);
    if (k >= CW && ty == 0) continue;      // the corners are in the top row already
    const int x = (int)(t % tilesX) * CW + tx, y = (int)(t / tilesX) * CH + ty;
    if (x >= w || y >= h) continue;
    const uint32_t c = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    const T e = z[c];
    if (y > 0) {
      if (x > 0 && (ty == 0 || tx == 0) && z[c - w - 1] == e) uf_unite(L, c, c - (uint32_t)w - 1u);
      if (ty == 0 && z[c - w] == e) uf_unite(L, c, c - (uint32_t)w);
      if (x < w - 1 && (ty == 0 || tx == CW - 1) && z[c - w + 1] == e) uf_unite(L, c, c - (uint32_t)w + 1u);
    }
    if (x > 0 && tx == 0 && z[c - 1] == e) uf_unite(L, c, c - 1u);
  }
}

// The same unions from the compact column records and with the repeats left out.  Along a lake's tile border every cell
// asks for the SAME union (its tile's component with the neighbouring tile's): a pair of tile-level labels a lane has
// asked for already, or that the lane before it asks for, is skipped -- the lane that keeps it is the first of the run.
//   part A  one thread per cell of a tile's top row: NW, N, NE (the row above, other tiles; coalesced raster reads)
//   part B  one thread per row of a tile's left column: NW, W, SW in the right column of the tile to the left
//           (the NE of a right-column cell is the SW of the left-column cell one row up; the first / last row's
//           NW / SW belong to another tile row: part A of the cell concerned has them as its NW / NE)
__device__ __forceinline__ bool ccl_repeat(unsigned long long key, bool valid, const unsigned long long (&mine)[3], const bool (&mv)[3],
                                           int upto, const unsigned long long (&prev)[3], const bool (&pv)[3], bool has_prev) {
  if (!valid) return true;
  for (int j = 0; j < upto; j++)
    if (mv[j] && mine[j] == key) return true;
  if (has_prev)
    for (int j = 0; j < 3; j++)
      if (pv[j] && prev[j] == key) return true;
  return false;
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_ccl_border2(const T *__restrict__ z, uint32_t *L, const T *__restrict__ colZ,
                                                      const uint32_t *__restrict__ colL, int w, int h, uint32_t tilesX, uint32_t ntiles) {
  const uint64_t nA = (uint64_t)ntiles * CW, total = nA + (uint64_t)ntiles * CH;
  const uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long key[3] = {0, 0, 0};
  bool kv[3] = {false, false, false};
  uint32_t a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  if (i < nA) {
    const uint32_t t = (uint32_t)(i / CW);
    const int tx = (int)(i % CW);
    const int x = (int)(t % tilesX) * CW + tx, y = (int)(t / tilesX) * CH;
    if (x < w && y > 0 && y < h) {
      const uint32_t c = (uint32_t)y * (uint32_t)w + (uint32_t)x;
      const T e = z[c];
      const uint32_t lc = L[c];   // (a member of c's set whatever the unions under way have done to it)
      for (int d = 0; d < 3; d++) {
        const int nx = x + d - 1;
        if (nx < 0 || nx >= w) continue;
        const uint32_t n = c - (uint32_t)w + (uint32_t)(d - 1);
        if (!(z[n] == e)) continue;
        a[d] = lc; b[d] = L[n];
        key[d] = ((unsigned long long)a[d] << 32) | b[d];
        kv[d] = true;
      }
    }
  } else if (i < total) {
    const uint64_t k = i - nA;
    const uint32_t t = (uint32_t)(k / CH);
    const int ly = (int)(k % CH);
    if (t % tilesX != 0) {
      const size_t me = (size_t)t * (2 * CH) + ly, nb = (size_t)(t - 1) * (2 * CH) + CH;   // my left column, its right column
      const uint32_t lc = colL[me];
      if (lc != CCL_NOCELL) {
        const T e = colZ[me];
        for (int d = 0; d < 3; d++) {
          const int ny = ly + d - 1;
          if (ny < 0 || ny >= CH) continue;
          const uint32_t ln = colL[nb + ny];
          if (ln == CCL_NOCELL || !(colZ[nb + ny] == e)) continue;
          a[d] = lc; b[d] = ln;
          key[d] = ((unsigned long long)lc << 32) | ln;
          kv[d] = true;
        }
      }
    }
  }
  unsigned long long pk[3];
  bool pv[3];
  for (int d = 0; d < 3; d++) {
    pk[d] = __shfl_up(key[d], 1, 64);
    pv[d] = __shfl_up((int)kv[d], 1, 64) != 0;
  }
  for (int d = 0; d < 3; d++)
    if (!ccl_repeat(key[d], kv[d], key, kv, d, pk, pv, lane > 0)) uf_unite(L, a[d], b[d]);
}

// k_ccl_flatten four cells a thread, and flat_height's start: a component's deepest away level is collected at its root
// (fh[root] = 0; the other entries of fh are never read by the lean path)
__global__ __launch_bounds__(NTHR) void k_ccl_flatten4(uint32_t *L, int32_t *fh, uint64_t n) {
  const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t q = (uint64_t)blockIdx.x * NTHR + threadIdx.x; q < n4 + (n & 3); q += stride) {
    if (q < n4) {
      uint4 v = reinterpret_cast<const uint4 *>(L)[q];
      uint32_t p[4] = {v.x, v.y, v.z, v.w};
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t c = (uint32_t)(4 * q) + e;
        if (p[e] == c) { fh[c] = 0; continue; }
        const uint32_t r = uf_find(L, p[e]);
        any |= r != p[e];
        p[e] = r;
      }
      if (any) reinterpret_cast<uint4 *>(L)[q] = make_uint4(p[0], p[1], p[2], p[3]);
    } else {
      const uint64_t c = 4 * n4 + (q - n4);
      const uint32_t p = L[c];
      if (p == (uint32_t)c) { fh[c] = 0; continue; }
      const uint32_t r = uf_find(L, p);
      if (r != p) L[c] = r;
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_ccl_flatten(uint32_t *L, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const uint32_t p = L[c];
    if (p == (uint32_t)c) continue;
    const uint32_t r = uf_find(L, p);
    if (r != p) L[c] = r;
  }
}

// fh[root] = -1: flat without a low edge (label 0 in the reference, :483-487); >= 0: labelled.
// the same from the flag bytes (no edge list): every low edge marks its flat as having an outlet
__global__ __launch_bounds__(NTHR) void k_flat_mark_low_flags(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ L,
                                                              int32_t *fh, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride)
    if (flags[c] & F_LOW) fh[L[c]] = 0;
}

__global__ __launch_bounds__(NTHR) void k_flat_mark_low(const uint32_t *__restrict__ low, uint32_t nlow,
                                                        const uint32_t *__restrict__ L, int32_t *fh) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i < nlow) fh[L[low[i]]] = 0;
}

// ------------------------------------------------------------------------------------------
// BuildAwayGradient (:152-198) / BuildTowardsCombinedGradient (:241-298): the reference's levels are
// BFS distances (8-connected, through NO_FLOW cells of the same flat) from the high / low edges.  A
// level-synchronous BFS needs one global step per level -- 20 000+ steps on a 40k x 40k filled DEM whose
// lakes are long and thin -- so distances are computed by tile-local relaxation instead: an active
// 64x32 tile is staged in LDS (elevation, eligibility, distance, 1-cell halo), relaxed to its local fixed
// point d(c) = min(d(c), min over equal-elevation neighbours d(n) + 1), written back, and the tiles
// across a changed edge are activated for the next round.  Distances only ever decrease and stay upper
// bounds, so the fixed point is the exact BFS level; the number of global rounds is the geodesic length in
// TILES, not in cells.
// ------------------------------------------------------------------------------------------
constexpr int32_t DINF = 0x7F7F7F7F;   // memset-able "not reached"
constexpr int RCH = 32;                // relaxation tiles are CW x RCH (the labelling tiles CW x CH)
constexpr int HSTEPS = 8;              // stencil steps per barrier inside a relaxation tile (a step without a move ends them early)
constexpr int RNT = 256, RBANDS = RNT / 64;   // one wavefront per band of RCH / RBANDS rows
constexpr int RELAX_BATCH = 8;         // relaxation rounds enqueued per host read-back (stencil engine)
constexpr int BITS_BATCH = 24;         // ... of the bitmap search: its rounds are short, the read-back's bubble is not

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

// sources get distance 1 and activate their tile.  fh != nullptr: only cells of labelled flats (:491-500)
__global__ __launch_bounds__(NTHR) void k_flat_seed(const uint32_t *__restrict__ src, uint32_t nsrc,
                                                    const uint32_t *__restrict__ L, const int32_t *__restrict__ fh,
                                                    int32_t *D, uint8_t *tile_active, int w, uint32_t tilesX,
                                                    uint32_t tilesY, const int32_t *__restrict__ reach) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i >= nsrc) return;
  const uint32_t c = src[i];
  if (fh && fh[L[c]] < 0) return;   // its flat has no outlet: not a source
  if (reach && reach[c] >= DINF) return;   // same test through the towards distances (row-block shards)
  D[c] = 1;
  // wake the source's tile and the tiles of its 8 neighbours (a source on a tile edge feeds the next tile)
  const int x = (int)(c % (uint32_t)w), y = (int)(c / (uint32_t)w);
  const int tx0 = max(x - 1, 0) / CW, tx1 = min(x + 1, w - 1) / CW;
  const int ty0 = max(y - 1, 0) / RCH, ty1 = (y + 1) / RCH;
  for (int ty = ty0; ty <= ty1; ty++)
    for (int tx = tx0; tx <= tx1; tx++)
      if ((uint32_t)ty < tilesY) tile_active[(uint32_t)ty * tilesX + (uint32_t)tx] = 1;
}

// active-tile flags -> list (flags are cleared for the next round)
__global__ __launch_bounds__(NTHR) void k_tiles_compact(uint8_t *flags, uint32_t ntiles, uint32_t *list,
                                                        uint32_t *count) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  const bool hit = i < ntiles && flags[i] != 0;
  if (hit) flags[i] = 0;
  const uint32_t slot = block_append(hit, count);
  if (hit) list[slot] = i;
}

// lane l receives lane l-1's value (lane 0: fill) / lane l+1's value (lane 63: fill): full-rate DPP wave shifts
__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }

// Start of the towards field, one 64 x RCH tile per block: low edges hold level 1 (they are not relaxed: they have a
// direction), the NO_FLOW cells next to a low edge of their own flat level 2 (:262-275: the low edges are the first
// BFS level), everything else "not reached".  Rows outside [row_lo, row_hi) -- a shard's ghost rows -- are not
// seeded: their levels arrive from their owners.  A tile with a seed wakes itself and its 8 neighbours.
__global__ __launch_bounds__(NTHR) void k_flat_init_towards(const uint8_t *__restrict__ flags, int32_t *__restrict__ D,
                                                            uint8_t *tile_active, int w, int h, int row_lo, int row_hi,
                                                            uint32_t tilesX, uint32_t tilesY) {
  const uint32_t t = blockIdx.x;
  const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
  const int x0 = tx * CW, y0 = ty * RCH;
  const int lx = threadIdx.x & (CW - 1), r0 = threadIdx.x >> 6;
  int seed = 0;
#pragma unroll
  for (int j = 0; j < RCH / 4; j++) {
    const int gx = x0 + lx, gy = y0 + r0 + 4 * j;
    if (gx >= w || gy >= h) continue;
    const size_t g = (size_t)gy * w + gx;
    int32_t v = DINF;
    if (gy >= row_lo && gy < row_hi) {
      const uint8_t f = flags[g];
      if (f & F_LOW) v = 1;
      else if (f & F_NEAR) { v = 2; seed = 1; }
    }
    D[g] = v;
  }
  if (__syncthreads_or(seed) && threadIdx.x < 9) {
    const int ntx = tx + (int)threadIdx.x % 3 - 1, nty = ty + (int)threadIdx.x / 3 - 1;
    if (ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) tile_active[nty * tilesX + ntx] = 1;
  }
}

// One active tile to its local fixed point d(c) = min(d(c), min over the 8 neighbours d(n) + 1) over its NO_FLOW
// cells.  Two adjacent NO_FLOW cells always have the same elevation (neither has a lower neighbour), so among the cells
// that take part the flat graph is the plain 8-grid with the other cells as walls: no elevations are read here at all
// (the low edges enter through the start values, k_flat_init_towards).  Per cell the kernel reads its level and its
// direction byte.
// A wavefront is one 64-column band of ROWS rows: a lane keeps its column strip in registers, sweeps it down and up
// (Gauss-Seidel), takes the two side columns' vertical 3-minima from the neighbouring lanes with DPP wave shifts, and
// only the bands' first/last rows go through LDS (double buffered: one barrier per iteration, shared with the "anything
// changed" vote).  Levels only decrease and stay upper bounds, so the fixed point is the exact BFS level.
// (Measured and dropped, r02, all correct: min-plus scans along the rows inside this kernel -- a front then crosses the
// tile width in one trip -- with ds_bpermute shuffles or with DPP row shifts + v_readlane: 2.5-3x SLOWER at S3, because
// with one stencil step per trip the diagonal fronts of open lakes take 32 trips instead of 8; and one wavefront per
// tile running chamfer sweeps over rows kept in LDS, no barriers at all: 1.4x slower, 89 + 46 ms against 60 + 33 ms; a
// block that follows the front -- goes on with the tile across a changed edge, up to four tiles per launch, minimum
// writes -- saved 7 % of the rounds and doubled the relaxations of the full rounds: 76 + 48 ms against 54 + 26 ms.)
constexpr int32_t DWALL = DINF + 1;   // LDS only: a cell that does not take part
__device__ __forceinline__ void relax_tile(const uint8_t *__restrict__ dirs, int32_t *D, const uint32_t t,
                                           uint8_t *next_active, int w, int h, uint32_t tilesX, uint32_t tilesY, int row_lo,
                                           int row_hi) {
  constexpr int RW = CW + 2, RH = RCH + 2, ROWS = RCH / RBANDS;
  __shared__ int32_t sd[RH * RW];
  __shared__ int32_t xrow[2][RBANDS][2][CW];
  const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
  const int x0 = tx * CW, y0 = ty * RCH;
  {
    // all loads of the thread are issued before the first one is consumed (clamped addresses, branch-free)
    constexpr int IPT = (RH * RW + RNT - 1) / RNT;
    int32_t dv[IPT];
    uint8_t ev[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = min((int)threadIdx.x + r * RNT, RH * RW - 1);
      const int ly = i / RW, lxx = i - ly * RW;
      const int gx = min(max(x0 - 1 + lxx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      const size_t g = (size_t)gy * w + gx;
      dv[r] = __hip_atomic_load(&D[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // neighbours' tiles write it
      ev[r] = dirs[g];
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * RNT;
      if (i >= RH * RW) continue;
      const int ly = i / RW, lxx = i - ly * RW;
      const int gx = x0 - 1 + lxx, gy = y0 - 1 + ly;
      const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
      sd[i] = (in && (ev[r] == 0 || ev[r] == DIR_GHOST_NOFLOW)) ? dv[r] : DWALL;   // only NO_FLOW cells take part (:190-191)
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & (CW - 1), band = threadIdx.x >> 6;
  int32_t d[ROWS], d0[ROWS];
  uint32_t elig = 0;   // bit j: the cell is relaxed here (NO_FLOW, inside the raster, in the own rows)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j, gy = y0 + ly;
    d[j] = sd[(ly + 1) * RW + lx + 1];
    if (d[j] <= DINF && gy >= row_lo && gy < row_hi) elig |= 1u << j;
    d0[j] = d[j];
  }
  if (!__syncthreads_or(elig != 0)) return;
  auto ring = [&](int ly /* -1..RCH */, int cx /* -1..CW */) -> int32_t { return sd[(ly + 1) * RW + cx + 1]; };
  int32_t sideL[ROWS], sideR[ROWS];   // vertical 3-minima of the halo columns (lanes 0 and 63 use them)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int ly = band * ROWS + j;
    sideL[j] = lx == 0 ? imin(ring(ly - 1, -1), imin(ring(ly, -1), ring(ly + 1, -1))) : DWALL;
    sideR[j] = lx == CW - 1 ? imin(ring(ly - 1, CW), imin(ring(ly, CW), ring(ly + 1, CW))) : DWALL;
  }
  const int32_t halo_up = band == 0 ? ring(-1, lx) : DWALL, halo_dn = band == RBANDS - 1 ? ring(RCH, lx) : DWALL;
  int changed = 1, it = 0;
  for (; it < 256; it++) {
    // Gauss-Seidel along the strip
#pragma unroll
    for (int j = 1; j < ROWS; j++)
      if (elig & (1u << j)) d[j] = imin(d[j], d[j - 1] + 1);
#pragma unroll
    for (int j = ROWS - 2; j >= 0; j--)
      if (elig & (1u << j)) d[j] = imin(d[j], d[j + 1] + 1);
    xrow[it & 1][band][0][lx] = d[0];
    xrow[it & 1][band][1][lx] = d[ROWS - 1];
    if (!__syncthreads_or(changed)) break;
    const int32_t up = band == 0 ? halo_up : xrow[it & 1][band - 1][1][lx];
    const int32_t dn = band == RBANDS - 1 ? halo_dn : xrow[it & 1][band + 1][0][lx];
    // HSTEPS stencil steps per barrier: the sideways exchange is all DPP (registers), so a front crosses HSTEPS
    // columns per trip; the rows above / below the band are one trip stale, which only delays, never breaks,
    // convergence (levels are upper bounds and only decrease).  A step in which no lane of the wavefront moved ends
    // the trip's steps (the next ones would compute the same values).  (Measured and dropped: skipping single rows
    // whose neighbourhood did not move -- the per-row scalar branches cost the tail rounds more, 60 -> 72 ms, than
    // they saved the full rounds, 33 -> 30 ms.)
    changed = 0;
#pragma unroll
    for (int sub = 0; sub < HSTEPS; sub++) {
      int32_t m[ROWS];
      int moved = 0;
#pragma unroll
      for (int j = 0; j < ROWS; j++) m[j] = imin(j ? d[j - 1] : up, imin(d[j], j + 1 < ROWS ? d[j + 1] : dn));
#pragma unroll
      for (int j = 0; j < ROWS; j++) {
        const int32_t side = imin(from_left(m[j], sideL[j]), from_right(m[j], sideR[j]));
        const int32_t best = imin(m[j], side) + 1;
        if ((elig & (1u << j)) && best < d[j]) { d[j] = best; moved = 1; }
      }
      changed |= moved;
      if (!__any(moved)) break;
    }
  }
  if (it == 256 && threadIdx.x == 0) next_active[t] = 1;   // iteration cap hit: finish this tile next round
  // Write back, and wake a neighbouring tile only if a cell that moved here can still lower one of ITS cells: the ring
  // staged at the start holds that tile's edge levels, and a ring cell at most one above the new level has nothing to
  // gain (levels only decrease, so a stale ring value can only wake a tile needlessly, never miss one).  Waking on
  // every changed edge made two tiles that merely agree on their common edge wake each other for another round.
  __shared__ uint32_t wake;
  if (threadIdx.x == 0) wake = 0;
  __syncthreads();
  uint32_t mine = 0;   // bit (dy + 1) * 3 + (dx + 1): the tile at (tx + dx, ty + dy)
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    if (!(elig & (1u << j)) || d[j] >= d0[j]) continue;
    const int ly = band * ROWS + j;
    const int gx = x0 + lx, gy = y0 + ly;
    __hip_atomic_store(&D[(size_t)gy * w + gx], d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ly != 0 && ly != RCH - 1 && lx != 0 && lx != CW - 1) continue;
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int ny = ly + dy, nx = lx + dx;
        const int sy = ny < 0 ? -1 : ny >= RCH ? 1 : 0, sx = nx < 0 ? -1 : nx >= CW ? 1 : 0;
        if (!sy && !sx) continue;
        const int32_t r = ring(ny, nx);
        if (r <= DINF && r > d[j] + 1) mine |= 1u << ((sy + 1) * 3 + sx + 1);
      }
  }
  if (mine) atomicOr(&wake, mine);
  __syncthreads();
  if (threadIdx.x < 9 && threadIdx.x != 4 && (wake >> threadIdx.x & 1u)) {
    const int ntx = tx + (int)threadIdx.x % 3 - 1, nty = ty + (int)threadIdx.x / 3 - 1;
    if (ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) next_active[nty * tilesX + ntx] = 1;
  }
}

// One block per listed tile.  The count lives on the device (rounds are enqueued in batches, the host sizes the
// grid from the previous batch): blocks past the count leave at once, and tiles past the grid -- the list grew
// faster than expected -- simply stay active for the next round (relaxation is monotone, order is irrelevant).
__global__ __launch_bounds__(RNT) void k_flat_relax(const uint8_t *__restrict__ dirs, int32_t *D,
                                                    const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ count,
                                                    uint8_t *next_active, int w, int h, uint32_t tilesX, uint32_t tilesY,
                                                    int row_lo, int row_hi) {
  const uint32_t n = *count;
  for (uint32_t i = gridDim.x + blockIdx.x * RNT + threadIdx.x; i < n; i += gridDim.x * RNT) next_active[tiles[i]] = 1;
  if (blockIdx.x >= n) return;
  relax_tile(dirs, D, tiles[blockIdx.x], next_active, w, h, tilesX, tilesY, row_lo, row_hi);
}

// ------------------------------------------------------------------------------------------
// The same fixed point by breadth-first search on BITMAPS, one wavefront per 64 x 64 tile (single device and row-block
// shards alike: see RowWin; k_flat_relax above stays selectable with RDGPU_FLAT_BITS=0).
// Lane r holds row r of the tile as 64-bit masks: the cells that take part (M), those not yet reached (A), the current
// front (F).  One BFS level is   N = dilate8(F) & A   =   two DPP row shifts, two bit shifts and a few ORs for the whole
// tile -- against ~100 VALU instructions per wavefront and stencil step in k_flat_relax, which r02's traces showed to be
// what bounds the stage (instruction issue in the full rounds, a lone wavefront's dependent chain in the ~500 tail
// rounds).  Levels are recorded bit-sliced (plane j collects the cells whose level has bit j set) relative to a base,
// and turned back into one int per cell when the tile is stored.
// Sources of a visit: the tile's seeds (a bitmap, from the flags) and the ring of cells around the tile with the levels
// they hold now -- each is injected at its own level, so the search also runs from several fronts at different depths,
// skipping the gaps between them.  The levels of the tile's own cells are not read at all: they were computed from
// older (higher or equal) ring levels, so recomputing from the current ring can only reproduce or lower them.
// (Measured, r02: holding back tiles whose start level runs ahead of 64..320 levels per round changed nothing -- the
// sweep already is level ordered, one tile per round: at S3 the deepest level is ~2e4 and the towards field takes 349
// rounds; per round a visit's ~35 us of dependent instruction issue is the floor, whatever the number of tiles.)
// ------------------------------------------------------------------------------------------
constexpr int BT = 64;   // bitmap tiles are BT x BT
constexpr int BPLANES = 8;   // level planes: 256 levels per flush
// r06 (flat_planes.inc): a level field kept as bit planes per 64 x 64 tile instead of one int per cell
struct PlaneField {
  unsigned long long *P = nullptr;   // [tile][16][64]
  int32_t *E = nullptr;              // [tile][4][64]
  unsigned long long *R = nullptr;   // [tile][64]
  uint32_t *overflow = nullptr;
  uint8_t *expanded = nullptr;       // per tile: has been visited (its planes exist)
  int32_t max_level = 0xFFF0;        // a level from here on raises `overflow` (the planes hold 16 bits)
};
struct BitsScratch {
  unsigned long long *mbits;   // per tile and row: the cells that take part
  uint8_t *tflags, *expanded;  // tile is active next round / has been visited in this field
  uint32_t *tlist, *ctr, *counts;
  uint32_t tilesX, tilesY, ntiles;
  PlaneField pf;               // P != nullptr: the search runs on planes
  unsigned long long *near = nullptr, *high = nullptr;   // (planes) the two fields' seed rows: the cells next to a low edge, the high edges
};

__device__ __forceinline__ uint32_t dpp_up(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t dpp_dn(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
// lane r receives lane r-1's / lane r+1's mask (0 past the ends)
__device__ __forceinline__ unsigned long long row_above(unsigned long long v) {
  return ((unsigned long long)dpp_up((uint32_t)(v >> 32)) << 32) | dpp_up((uint32_t)v);
}
__device__ __forceinline__ unsigned long long row_below(unsigned long long v) {
  return ((unsigned long long)dpp_dn((uint32_t)(v >> 32)) << 32) | dpp_dn((uint32_t)v);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}
// val in the lanes whose bit is set in the wave-uniform mask, 0 elsewhere: one v_cndmask on the scalar pair
__device__ __forceinline__ uint32_t lanes_of(unsigned long long smask, uint32_t val) {
  uint32_t out;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(out) : "v"(val), "s"(smask));
  return out;
}
// minimum / maximum over the 64 lanes, in every lane: a DPP prefix scan (row shifts + row broadcasts) whose last lane
// holds the result -- six VALU instructions and a v_readlane instead of six LDS permutes
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
  constexpr int32_t I = INT32_MAX;
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) { return -wave_min_i32(-v); }   // (values here are > INT32_MIN)

// The rows a search owns.  Single device: [0, h), nothing beyond.  A row-block shard: [lo, hi) inside a buffer of h rows
// whose rows lo - 1 and hi are GHOST rows -- cells of the neighbouring shard: they are never assigned here, their levels
// arrive from their owner and act exactly like the ring of a tile (the tile grid starts at row lo, so the upper ghost row
// IS the ring of the first tile row; the lower one is the ring below the last own row, wherever in its tile that is).
// gtop / gbot: per tile column, which cells of the ghost row take part (NO_FLOW cells); null: no ghost row.
struct RowWin {
  int lo, hi;
  const unsigned long long *gtop, *gbot;
};

// Per tile and row: the bitmap of the cells that take part; the start of a field's levels (the seeds hold theirs, all
// other cells "not reached"), the tiles to visit first, and the edge counts.
// TOWARDS: seeds = F_NEAR (level 2), D = 1 on the low edges; else seeds = F_HIGH (level 1; with L / fh only those of
// flats that have an outlet, :491-500).  One block per tile, a wavefront per 16 rows, a lane per column.
template <bool TOWARDS, bool WRITE_M>
__global__ __launch_bounds__(NTHR) void k_bits_prepare(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ L,
                                                       const int32_t *__restrict__ fh, int32_t *__restrict__ D,
                                                       unsigned long long *mbits, uint8_t *tile_active, uint32_t *counts,
                                                       int w, RowWin win, const int32_t *__restrict__ reach, uint32_t tilesX,
                                                       uint32_t tilesY) {
  const int h = win.hi;
  const uint32_t t = blockIdx.x;
  const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gx = tx * BT + lane;
  unsigned long long mrow = 0, srow = 0;   // lane j: the masks of row wv * 16 + j
  uint32_t nlow = 0, nhigh = 0, nnoflow = 0;
  uint8_t f[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int gy = win.lo + ty * BT + wv * 16 + j;
    f[j] = (gx < w && gy < h) ? flags[(size_t)gy * w + gx] : (uint8_t)0;
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int gy = win.lo + ty * BT + wv * 16 + j;
    const bool in = gx < w && gy < h;
    bool seed = TOWARDS ? (f[j] & F_NEAR) != 0 : (f[j] & F_HIGH) != 0;
    if (!TOWARDS && fh && seed) seed = fh[L[(size_t)gy * w + gx]] >= 0;
    if (!TOWARDS && reach && seed) seed = reach[(size_t)gy * w + gx] < DINF;   // shards: "the flat has an outlet" through the towards levels
    const unsigned long long mb = __ballot((f[j] & F_NOFLOW) != 0), sb = __ballot(seed);
    if (lane == j) { mrow = mb; srow = sb; }
    if (counts) {
      const unsigned long long lb = __ballot((f[j] & F_LOW) != 0), hb = __ballot((f[j] & F_HIGH) != 0);
      nnoflow += (uint32_t)__popcll(mb);   // (wave uniform sums; lane 0 reports them)
      nlow += (uint32_t)__popcll(lb);
      nhigh += (uint32_t)__popcll(hb);
    }
    if (in) D[(size_t)gy * w + gx] = (TOWARDS && (f[j] & F_LOW)) ? 1 : seed ? (TOWARDS ? 2 : 1) : DINF;
  }
  if (WRITE_M && lane < 16) mbits[(size_t)t * BT + wv * 16 + lane] = mrow;
  const int any = __syncthreads_or(srow != 0);
  if (any && threadIdx.x < 9) {
    const int ntx = tx + (int)threadIdx.x % 3 - 1, nty = ty + (int)threadIdx.x / 3 - 1;
    if (ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) tile_active[nty * tilesX + ntx] = 1;
  }
  if (counts && lane == 0 && (nlow | nhigh | nnoflow)) {   // striped: same-address atomics serialise
    uint32_t *c = counts + 3 * ((t * 4 + wv) & 255u);
    if (nlow) atomicAdd(c, nlow);
    if (nhigh) atomicAdd(c + 1, nhigh);
    if (nnoflow) atomicAdd(c + 2, nnoflow);
  }
}

__global__ __launch_bounds__(NTHR) void k_bits_counts(const uint32_t *__restrict__ counts, unsigned long long *out) {
  __shared__ unsigned long long acc[3];
  if (threadIdx.x < 3) acc[threadIdx.x] = 0;
  __syncthreads();
  for (int k = 0; k < 3; k++) atomicAdd(&acc[k], (unsigned long long)counts[3 * threadIdx.x + k]);
  __syncthreads();
  if (threadIdx.x < 3) out[threadIdx.x] = acc[threadIdx.x];
}

// inclusive prefix / suffix minimum over the 64 lanes, per lane (DPP row shifts inside the rows of 16; the prefix crosses
// the rows with row_bcast, the suffix with three v_readlane) -- checked by tools/probes/scan_probe
__device__ __forceinline__ int32_t wave_prefix_min(int32_t v) {
  constexpr int32_t I = INT32_MAX;
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int32_t wave_suffix_min(int32_t v, int lane) {
  constexpr int32_t I = INT32_MAX;
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x101, 0xf, 0xf, false));   // row_shl:1
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x102, 0xf, 0xf, false));   // row_shl:2
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x104, 0xf, 0xf, false));   // row_shl:4
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x108, 0xf, 0xf, false));   // row_shl:8
  const int32_t r3 = __builtin_amdgcn_readlane(v, 48), r2 = imin(__builtin_amdgcn_readlane(v, 32), r3),
                r1 = imin(__builtin_amdgcn_readlane(v, 16), r2);
  return imin(v, lane < 16 ? r1 : lane < 32 ? r2 : lane < 48 ? r3 : I);
}

// visit statistics of the STATS instantiation (RDGPU_FLAT_TRACE): [0] visits, [1] left at once (nothing new reaches the
// tile), [2] open-water visits, [3] general visits, [4] BFS levels stepped, [5] flushes, [6] rows flushed
// [8..15] (tail launches, < 8000 tiles): visits timed, 10 ns ticks of: setup (ring loads, start level), open-water visit,
// general rows read, level loop incl. flushes, flushes alone, wake test, whole visit; [15] longest visit
__device__ unsigned long long g_relax_stats[16];

#ifdef RDGPU_PROBES
// tools/probes/flat_visit_hist.py: per field (towards first) and tile: [0] visits, [1] visits that had work, [2] BFS levels
// stepped, [3] visits starting BELOW the start level of the tile's previous working visit (that visit ran ahead), [4] that level
__device__ uint32_t *g_probe_hist = nullptr;
extern "C" int rdgpu_probe_flat_hist(uint32_t *d_hist) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_hist), &d_hist, sizeof d_hist) == hipSuccess ? 0 : 1;
}
#endif

// D accesses of a visit.  CO (the asynchronous search, k_relax_bits_async): other wavefronts of the SAME launch write the
// ring cells this visit reads and read the cells it writes, across CUs and XCDs whose L1 / L2 are not coherent with each
// other -- relaxed agent-scope atomics (sc1 loads and write-through stores) on both sides, MI355X_MICROARCH.md's
// "sc1 payload -> vmcnt(0) -> flag" hand-off.  Otherwise (one launch per round): plain loads and stores.
template <bool CO>
__device__ __forceinline__ int32_t ldD(const int32_t *p) {
  if (CO) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool CO>
__device__ __forceinline__ void stD(int32_t *p, int32_t v) {
  if (CO) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// One visit of tile t by one wavefront; returns the 9-bit mask of the neighbouring tiles to wake (bit (dy + 1) * 3 + dx + 1,
// the same in every lane).
template <int SEED_LEVEL, bool STATS, bool CO>
__device__ __forceinline__ uint32_t relax_visit(const unsigned long long *__restrict__ mbits, uint8_t *expanded, int32_t *D,
                                                const uint32_t t, uint16_t *const orow, const bool timed, int w, int h, RowWin win,
                                                uint32_t tilesX, uint32_t tilesY) {
  const int lane = threadIdx.x & 63;
  constexpr int RB = CO ? 32 : 16;   // rows read per batch (the coherent loads come from beyond the L2: fewer, longer batches)
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tkf = 0;
  if (STATS) tk0 = wall_clock64();
  const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
  const int x0 = tx * BT, y0 = win.lo + ty * BT;
  const int brow = min(BT - 1, win.hi - 1 - y0);   // the tile's last own row (63 except in the last tile row)
  const bool hasL = tx > 0, hasR = tx + 1 < (int)tilesX, hasT = ty > 0, hasB = ty + 1 < (int)tilesY;
  const bool gT = !hasT && win.gtop != nullptr, gB = !hasB && win.gbot != nullptr;   // a ghost row instead of a tile
  const size_t tb = (size_t)t * BT;
  const unsigned long long M = mbits[tb + lane];
  // which ring cells take part: the facing rows / columns / corners of the eight neighbouring tiles' masks
  const unsigned long long mT = hasT ? mbits[tb - (size_t)tilesX * BT + (BT - 1)] : gT ? win.gtop[tx] : 0ull;   // (wave uniform)
  const unsigned long long mB = hasB ? mbits[tb + (size_t)tilesX * BT] : gB ? win.gbot[tx] : 0ull;
  const unsigned long long mL = hasL ? mbits[tb - BT + lane] : 0ull;
  const unsigned long long mR = hasR ? mbits[tb + BT + lane] : 0ull;
  const bool eTL = hasL && ((hasT ? mbits[tb - (size_t)tilesX * BT - BT + (BT - 1)] : gT ? win.gtop[tx - 1] : 0ull) >> 63 & 1ull);
  const bool eTR = hasR && ((hasT ? mbits[tb - (size_t)tilesX * BT + BT + (BT - 1)] : gT ? win.gtop[tx + 1] : 0ull) & 1ull);
  const bool eBL = hasL && ((hasB ? mbits[tb + (size_t)tilesX * BT - BT] : gB ? win.gbot[tx - 1] : 0ull) >> 63 & 1ull);
  const bool eBR = hasR && ((hasB ? mbits[tb + (size_t)tilesX * BT + BT] : gB ? win.gbot[tx + 1] : 0ull) & 1ull);
  // ring levels (clamped addresses, all loads in flight together) and the tile's own edge levels as they are now
  const int cx = min(x0 + lane, w - 1), cy = min(y0 + lane, h - 1);
  const int yT = max(y0 - 1, 0), yB = min(y0 + brow + 1, h - 1), xL = max(x0 - 1, 0), xR = min(x0 + BT, w - 1);
  const int yl = min(y0 + brow, h - 1), xl = min(x0 + BT - 1, w - 1);
  int32_t tv = ldD<CO>(&D[(size_t)yT * w + cx]), bv = ldD<CO>(&D[(size_t)yB * w + cx]), lv = ldD<CO>(&D[(size_t)cy * w + xL]), rv = ldD<CO>(&D[(size_t)cy * w + xR]);
  int32_t tl = ldD<CO>(&D[(size_t)yT * w + xL]), tr = ldD<CO>(&D[(size_t)yT * w + xR]), bl = ldD<CO>(&D[(size_t)yB * w + xL]), br = ldD<CO>(&D[(size_t)yB * w + xR]);
  int32_t oldT = ldD<CO>(&D[(size_t)y0 * w + cx]), oldB = ldD<CO>(&D[(size_t)yl * w + cx]), oldL = ldD<CO>(&D[(size_t)cy * w + x0]), oldR = ldD<CO>(&D[(size_t)cy * w + xl]);
  if (!(mT >> lane & 1ull)) tv = DINF;
  if (!(mB >> lane & 1ull)) bv = DINF;
  // (in a last tile row that ends above row 63 the lane below the last own row holds the ghost row's corner cell)
  if (!(lane == brow + 1 ? eBL : (bool)(mL >> 63 & 1ull))) lv = DINF;
  if (!(lane == brow + 1 ? eBR : (bool)(mR & 1ull))) rv = DINF;
  if (!eTL) tl = DINF;
  if (!eTR) tr = DINF;
  if (!eBL) bl = DINF;
  if (!eBR) br = DINF;
  tl = __builtin_amdgcn_readfirstlane(tl); tr = __builtin_amdgcn_readfirstlane(tr);
  bl = __builtin_amdgcn_readfirstlane(bl); br = __builtin_amdgcn_readfirstlane(br);
  // (a cell that does not take part is never stored: its "new" level stays its "old" one)
  int32_t newT = oldT, newB = oldB, newL = oldL, newR = oldR;
  // The level at which each EDGE cell of the tile is reached from the ring: 1 + the lowest of its (up to three, at a
  // corner five) ring neighbours.  Computed once, so a level of the search only compares against these four vectors.
  // (the neighbouring lanes' values by DPP wave shifts, the corners as the fill of lanes 0 / 63)
  const int32_t tL = from_left(tv, tl), tR = from_right(tv, tr), bL = from_left(bv, bl), bR = from_right(bv, br);
  const int32_t lU = from_left(lv, tl), lD = from_right(lv, bl), rU = from_left(rv, tr), rD = from_right(rv, br);
  auto reach = [](int32_t a, int32_t b, int32_t c) { const int32_t m = imin(a, imin(b, c)); return m < DINF ? m + 1 : DINF; };
  const int32_t iT = reach(tL, tv, tR), iB = reach(bL, bv, bR), iL = reach(lU, lv, lD), iR = reach(rU, rv, rD);
  int32_t imax_;
  {
    auto fin = [](int32_t v) { return v < DINF ? v : -1; };
    imax_ = __builtin_amdgcn_readfirstlane(wave_max_i32(max(max(fin(iT), fin(iB)), max(fin(iL), fin(iR)))));
  }
  // Where to start.  Whatever this visit changes lies at or above the lowest level at which the ring now reaches an
  // edge cell earlier than the cell's own level says (a source acts at its level and later, never before); below that
  // the stored levels stand.  A tile visited for the first time also has its seeds to expand.  The cells below the
  // start are read back as "reached", those one below it are the front, and the search goes on from there -- a visit
  // that corrects the top of a tile's range costs that part, not the whole tile.
  const unsigned long long m0 = readlane64(M, 0), m63 = readlane64(M, brow);
  int32_t chg = DINF;
  if ((m0 >> lane & 1ull) && iT < oldT) chg = iT;
  if ((m63 >> lane & 1ull) && iB < oldB) chg = imin(chg, iB);
  if ((M & 1ull) && iL < oldL) chg = imin(chg, iL);
  if ((M >> 63 & 1ull) && iR < oldR) chg = imin(chg, iR);
  int32_t level = __builtin_amdgcn_readfirstlane(wave_min_i32(chg));   // the level being assigned
  if (!CO && !__builtin_amdgcn_readfirstlane((int)expanded[t])) {   // (CO: every tile with seeds has been visited before)
    level = imin(level, SEED_LEVEL + 1);
    if (lane == 0) expanded[t] = 1;
  }
  if (STATS && lane == 0) { atomicAdd(&g_relax_stats[0], 1ull); if (level >= DINF) atomicAdd(&g_relax_stats[1], 1ull); }
#ifdef RDGPU_PROBES
  uint32_t *const ph = g_probe_hist ? g_probe_hist + ((SEED_LEVEL == 2 ? 0u : 5u) * tilesX * tilesY + t) : nullptr;
  const uint32_t pstride = tilesX * tilesY;
  uint32_t psteps = 0;
  if (ph && lane == 0) {
    atomicAdd(&ph[0], 1u);
    if (level < DINF) {
      atomicAdd(&ph[pstride], 1u);
      const uint32_t last = ph[4 * pstride];
      if (last != 0 && (uint32_t)level < last) atomicAdd(&ph[3 * pstride], 1u);
      ph[4 * pstride] = (uint32_t)level;
    }
  }
#endif
  if (level >= DINF) return 0u;   // nothing new reaches this tile
  if (STATS) tk1 = wall_clock64();
  // OPEN WATER: every cell of the tile takes part, so the levels are chessboard distances from the ring (and from what the
  // tile holds already) and two chamfer sweeps give the fixed point exactly: once the ring has entered through the
  // levels at which it reaches the edge cells, any shortest king-move path inside the (convex) tile can be ordered into
  // down/right moves first, up/left moves after.  Lane = column; a row's horizontal pass is a prefix / suffix minimum of
  // (level -/+ column) over the lanes; the rows wait in LDS between the sweeps.  30 % of S3's tiles (61 % of its NO_FLOW
  // cells) are open, and they are the tiles of the tail rounds.  (With the 64 rows in registers instead the allocator
  // spilled both paths of this kernel.)  Levels are kept relative to (level - 1024) in 16 bits, 0xFFFF = not reached; a
  // consistent open tile spans at most 63 levels, anything below the base falls through to the general search.
  const int32_t obase = level - 1024;
  bool open = __all(M == ~0ull);
  if (open) {
    // down: from NW, N, NE, then from W along the row
    int32_t prev = DINF;
    bool fits = true;
    for (int r0 = 0; r0 < BT; r0 += RB) {
      int32_t v[RB];
#pragma unroll
      for (int j = 0; j < RB; j++) v[j] = ldD<CO>(&D[(size_t)(y0 + r0 + j) * w + x0 + lane]);
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const int y = r0 + j;
        int32_t d = v[j];
        fits &= d >= obase;
        const int32_t l0 = __builtin_amdgcn_readlane(iL, y), e0 = __builtin_amdgcn_readlane(iR, y);
        d = imin(d, lane == 0 ? l0 : lane == BT - 1 ? e0 : DINF);   // the ring enters through the edge cells' reach levels
        if (y == 0) d = imin(d, iT);
        if (y == BT - 1) d = imin(d, iB);
        if (y > 0) {
          const int32_t a = __builtin_amdgcn_update_dpp(DINF, prev, 0x138, 0xf, 0xf, false);   // the value of column x - 1
          const int32_t b = __builtin_amdgcn_update_dpp(DINF, prev, 0x130, 0xf, 0xf, false);   // ... of column x + 1
          d = imin(d, imin(imin(a, prev), b) + 1);
        }
        d = wave_prefix_min(d - lane) + lane;
        prev = d;
        const int32_t rel = d - obase;
        orow[y * BT + lane] = (uint16_t)(d >= DINF || rel > 0xFFFE ? 0xFFFF : rel);
      }
    }
    if (STATS && lane == 0 && __all(fits)) atomicAdd(&g_relax_stats[2], 1ull);
    if (__all(fits)) {
      // up: from SW, S, SE, then from E along the row; the finished row goes straight to memory
      int32_t nxt = DINF;
      for (int y = BT - 1; y >= 0; y--) {
        const uint16_t rv16 = orow[y * BT + lane];
        int32_t d = rv16 == 0xFFFF ? DINF : obase + (int32_t)rv16;
        if (y < BT - 1) {
          const int32_t a = __builtin_amdgcn_update_dpp(DINF, nxt, 0x138, 0xf, 0xf, false);
          const int32_t b = __builtin_amdgcn_update_dpp(DINF, nxt, 0x130, 0xf, 0xf, false);
          d = imin(d, imin(imin(a, nxt), b) + 1);
        }
        d = wave_suffix_min(d + lane, lane) - lane;
        nxt = d;
        if (d < DINF) stD<CO>(&D[(size_t)(y0 + y) * w + x0 + lane], d);
        const int32_t l0 = __builtin_amdgcn_readlane(d, 0), e0 = __builtin_amdgcn_readlane(d, BT - 1);
        newL = lane == y ? l0 : newL;   // (the edge columns as one value per lane = row, for the wake test)
        newR = lane == y ? e0 : newR;
        if (y == 0) newT = d;
        if (y == BT - 1) newB = d;
      }
    } else {
      open = false;   // (levels out of the 16-bit window: the general search)
    }
  }
  if (STATS) tk2 = wall_clock64();
  if (timed && open && lane == 0) atomicAdd(&g_relax_stats[9], tk2 - tk1);
  if (!open) {
  if (STATS && lane == 0) atomicAdd(&g_relax_stats[3], 1ull);
  unsigned long long A, F, Rec = 0;
  {
    unsigned long long reached = 0, front = 0;
#pragma unroll
    for (int r0 = 0; r0 < BT; r0 += RB) {
      int32_t v[RB];
#pragma unroll
      for (int j = 0; j < RB; j++) v[j] = ldD<CO>(&D[(size_t)min(y0 + r0 + j, h - 1) * w + cx]);
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const unsigned long long rb = __ballot(v[j] < level), fb = __ballot(v[j] == level - 1);
        if (lane == r0 + j) { reached = rb; front = fb; }
      }
    }
    A = M & ~reached;
    F = M & front;
  }
  if (STATS) { tk3 = wall_clock64(); if (timed && lane == 0) atomicAdd(&g_relax_stats[10], tk3 - tk2); }
  uint32_t Plo[BPLANES], Phi[BPLANES];
#pragma unroll
  for (int j = 0; j < BPLANES; j++) Plo[j] = Phi[j] = 0;
  int32_t base = level, relmax = 0;
  const uint32_t bitv[BPLANES] = {1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u};

  // store the cells recorded since the last flush; keeps the new edge levels for the wake test
  auto flush = [&]() {
    unsigned long long tf0 = 0;
    if (STATS) tf0 = wall_clock64();
    const int np = 32 - __clz(relmax | 1);
    unsigned long long rows = __ballot(Rec != 0);
    if (STATS && lane == 0) { atomicAdd(&g_relax_stats[5], 1ull); atomicAdd(&g_relax_stats[6], (unsigned long long)__popcll(rows)); }
    while (rows) {
      const int r = __ffsll((long long)rows) - 1;
      rows &= rows - 1;
      const unsigned long long rec = readlane64(Rec, r);
      uint32_t val = 0;
#pragma unroll
      for (int j = 0; j < BPLANES; j++)
        if (j < 5 || np > 5) {   // (five planes or all eight: one uniform test per row instead of one per plane)
          const unsigned long long pj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Phi[j], r) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)Plo[j], r);
          val |= lanes_of(pj, bitv[j]);
        }
      if (lanes_of(rec, 1u)) {
        const int32_t v = base + (int32_t)val;
        stD<CO>(&D[(size_t)(y0 + r) * w + x0 + lane], v);
        if (r == 0) newT = v;
        if (r == brow) newB = v;
      }
    }
    if (Rec & 1ull) {
      uint32_t val = 0;
#pragma unroll
      for (int j = 0; j < BPLANES; j++) val |= (Plo[j] & 1u) << j;
      newL = base + (int32_t)val;
    }
    if (Rec >> 63 & 1ull) {
      uint32_t val = 0;
#pragma unroll
      for (int j = 0; j < BPLANES; j++) val |= (Phi[j] >> 31) << j;
      newR = base + (int32_t)val;
    }
    Rec = 0;
#pragma unroll
    for (int j = 0; j < BPLANES; j++) Plo[j] = Phi[j] = 0;
    relmax = 0;
    if (STATS) tkf += wall_clock64() - tf0;
  };

  // Segments of at most 256 levels: the inner loop leaves when the planes are full (or the search is over), the flush
  // happens out here -- inlined INSIDE the level loop it made the compiler copy all sixteen plane registers on every
  // level (phi copies around the rare branch).
  for (bool searching = true; searching;) {
  for (;;) {
    if (level - base >= (1 << BPLANES)) break;   // planes full: flush, then on with this level (tested up here so that
                                                 // the body has ONE path that updates the planes in place)
    // one level: the cells next to the front, plus what the ring and the seeds start at this level, that are still free
    uint32_t nlo, nhi;
    {
      const unsigned long long v = F | row_above(F) | row_below(F);
      const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
      nlo = lo | (lo << 1) | __builtin_amdgcn_alignbit(hi, lo, 1);
      nhi = hi | (hi >> 1) | __builtin_amdgcn_alignbit(hi, lo, 31);
    }
    if (level <= imax_) {
      const unsigned long long T = __ballot(iT == level), B = __ballot(iB == level);
      const unsigned long long tb_ = (lane == 0 ? T : 0ull) | (lane == brow ? B : 0ull);
      nlo |= (uint32_t)tb_ | (iL == level ? 1u : 0u);
      nhi |= (uint32_t)(tb_ >> 32) | (iR == level ? 0x80000000u : 0u);
    }
    nlo &= (uint32_t)A; nhi &= (uint32_t)(A >> 32);
    const unsigned long long N = ((unsigned long long)nhi << 32) | nlo;
    if (STATS && lane == 0) atomicAdd(&g_relax_stats[4], 1ull);
#ifdef RDGPU_PROBES
    psteps++;
#endif
    if (__any(N != 0)) {
      const int32_t rel = level - base;
      relmax = rel;
      A &= ~N;
      Rec |= N;
#pragma unroll
      for (int j = 0; j < BPLANES; j++) {   // branch-free: the mask is all ones when the level has bit j set (a Gray-coded
        // counter per cell -- one plane flip a level, picked by a scalar if-chain -- was measured SLOWER: 31.4 vs 29.7 ms)
        const uint32_t m = (uint32_t)(-(int32_t)((uint32_t)rel >> j & 1u));
        Plo[j] |= nlo & m;
        Phi[j] |= nhi & m;
      }
      F = N;
      level++;
      continue;
    }
    // the front died: on to the next level at which the ring starts one, if any cell is left
    if (!__any(A != 0) || level >= imax_) { searching = false; break; }
    auto pend = [&](int32_t v) { return v > level ? v : DINF; };   // (DINF itself: never)
    const int32_t nx = __builtin_amdgcn_readfirstlane(wave_min_i32(imin(imin(pend(iT), pend(iB)), imin(pend(iL), pend(iR)))));
    if (nx >= DINF) { searching = false; break; }
    level = nx;
    F = 0;
  }
  if (__any(Rec != 0)) flush();
  base = level;
  }
  if (timed && lane == 0) { atomicAdd(&g_relax_stats[11], wall_clock64() - tk3); atomicAdd(&g_relax_stats[12], tkf); }
#ifdef RDGPU_PROBES
  if (ph && lane == 0) atomicAdd(&ph[2 * pstride], psteps);
#endif
  }
  unsigned long long tk4 = 0;
  if (STATS) tk4 = wall_clock64();
  // Wake a neighbouring tile only if an edge cell that moved here can still lower one of ITS cells (see k_flat_relax).
  // Edge cell (r, c) with new level v against the ring cells next to it, whose levels were read at the start.
  uint32_t wake = 0;   // bit (dy + 1) * 3 + dx + 1
  {
    // top / bottom edge rows: lane = column
    auto gain = [](int32_t ring, int32_t v) { return ring > v + 1; };   // (asked only of ring cells that take part)
    const bool movedT = newT < oldT, movedB = newB < oldB, movedL = newL < oldL, movedR = newR < oldR;
    const bool eT = mT >> lane & 1ull, eB = mB >> lane & 1ull;
    const bool eTl = lane > 0 ? (mT >> (lane - 1) & 1ull) : eTL, eTr = lane < 63 ? (mT >> (lane + 1) & 1ull) : eTR;
    const bool eBl = lane > 0 ? (mB >> (lane - 1) & 1ull) : eBL, eBr = lane < 63 ? (mB >> (lane + 1) & 1ull) : eBR;
    if (movedT) {
      if (eT && gain(tv, newT)) wake |= 1u << 1;
      if (eTl && gain(tL, newT)) wake |= lane > 0 ? 1u << 1 : 1u << 0;
      if (eTr && gain(tR, newT)) wake |= lane < 63 ? 1u << 1 : 1u << 2;
    }
    if (movedB) {
      if (eB && gain(bv, newB)) wake |= 1u << 7;
      if (eBl && gain(bL, newB)) wake |= lane > 0 ? 1u << 7 : 1u << 6;
      if (eBr && gain(bR, newB)) wake |= lane < 63 ? 1u << 7 : 1u << 8;
    }
    // left / right edge columns: lane = row (the corner cells' neighbours above / below the tile are the top / bottom rows' business)
    const unsigned long long eLm = __ballot(mL >> 63 & 1ull), eRm = __ballot(mR & 1ull);
    const bool eL = eLm >> lane & 1ull, eR = eRm >> lane & 1ull;
    const bool eLu = lane > 0 ? (eLm >> (lane - 1) & 1ull) : eTL, eLd = lane < 63 ? (eLm >> (lane + 1) & 1ull) : eBL;
    const bool eRu = lane > 0 ? (eRm >> (lane - 1) & 1ull) : eTR, eRd = lane < 63 ? (eRm >> (lane + 1) & 1ull) : eBR;
    if (movedL) {
      if (eL && gain(lv, newL)) wake |= 1u << 3;
      if (eLu && gain(lU, newL)) wake |= lane > 0 ? 1u << 3 : 1u << 0;
      if (eLd && gain(lD, newL)) wake |= lane < 63 ? 1u << 3 : 1u << 6;
    }
    if (movedR) {
      if (eR && gain(rv, newR)) wake |= 1u << 5;
      if (eRu && gain(rU, newR)) wake |= lane > 0 ? 1u << 5 : 1u << 2;
      if (eRd && gain(rD, newR)) wake |= lane < 63 ? 1u << 5 : 1u << 8;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wake |= __shfl_xor(wake, o, 64);
  if (timed && lane == 0) {
    const unsigned long long te = wall_clock64();
    atomicAdd(&g_relax_stats[7], 1ull);
    atomicAdd(&g_relax_stats[8], tk1 - tk0);
    atomicAdd(&g_relax_stats[13], te - tk4);
    atomicAdd(&g_relax_stats[14], te - tk0);
    atomicMax(&g_relax_stats[15], te - tk0);
  }
  return wake;
}

#include "flat_planes.inc"

// the cells of a bitmap, into a stripe of the edge counts
__global__ __launch_bounds__(NTHR) void k_rows_count(const unsigned long long *__restrict__ rows, uint32_t ntiles, uint32_t *counts, int slot) {
  const uint32_t t = blockIdx.x * (NTHR / 64) + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  uint32_t c = (uint32_t)__popcll(rows[(size_t)t * BT + (threadIdx.x & 63)]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(counts + 3 * (t & 255u) + slot, c);
}

template <int SEED_LEVEL, bool STATS = false>
__global__ __launch_bounds__(NTHR, 5) void k_relax_bits(const unsigned long long *__restrict__ mbits, uint8_t *expanded, int32_t *D,
                                                     const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ count,
                                                     uint8_t *next_active, int w, int h, RowWin win, uint32_t tilesX, uint32_t tilesY) {
  // open-water tiles keep their rows here between the two chamfer sweeps, as 16-bit levels relative to a base (8 KB per
  // wavefront: five blocks per CU, the occupancy the kernel's registers allow anyway)
  __shared__ uint16_t open_rows[NTHR / 64][BT * BT];
  const uint32_t n = *count;
  for (uint32_t i = gridDim.x * 4 + blockIdx.x * NTHR + threadIdx.x; i < n; i += gridDim.x * NTHR) next_active[tiles[i]] = 1;
  const uint32_t wi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n) return;
  const int lane = threadIdx.x & 63;
  const uint32_t t = tiles[wi];
  const uint32_t wake = relax_visit<SEED_LEVEL, STATS, false>(mbits, expanded, D, t, open_rows[threadIdx.x >> 6], STATS && n < 8000u, w,
                                                              h, win, tilesX, tilesY);
  if (lane < 9 && lane != 4 && (wake >> lane & 1u)) {
    const int tx = (int)(t % tilesX), ty = (int)(t / tilesX);
    const int ntx = tx + lane % 3 - 1, nty = ty + lane / 3 - 1;
    if (ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) next_active[nty * tilesX + ntx] = 1;
  }
}

// ------------------------------------------------------------------------------------------
// The asynchronous tail of the search.  A launch per round costs the round's SLOWEST visit plus the launch: at S3 the
// towards field needs ~350 rounds, the last 335 of them over fewer than 8000 tiles, at 36-66 us a round (rocprofv3 kernel
// trace, r03c) -- while a visit averages 18 us (this kernel's own clock: 744 000 visits, 13.3 wavefront-seconds).
// The fixed point does not depend on the order of the visits (levels only ever decrease and stay upper bounds), so once
// the front is thin the rounds are dropped: ONE launch of resident wavefronts that pull tiles from queues, visit them and
// push the neighbours they wake, until the queues run dry.  S3: 26.1 -> 17.0 ms for the towards field.
//   tile state   AQ_WAKE: queued, or woken while running;  AQ_RUN: a wavefront is visiting it.  A tile is in the queue
//                at most once and visited by at most one wavefront at a time (two visits of one tile at once could
//                store an older, higher level over a newer one).
//   queue        a ring of tile numbers (AQ_EMPTY = free slot); tickets from tail / head counters; a pusher waits for
//                its slot to be free, a popper for its slot to be filled
//   pending      tiles with a state other than 0; the search is over when it reaches 0
//   data         every D access of these visits is a relaxed agent-scope atomic (ldD / stD), the stores are drained
//                (s_waitcnt vmcnt(0)) before the state words of the neighbours are touched
// A wavefront that finds no work sleeps and polls; a launch that exceeds its tick budget raises the abort word and
// every wavefront leaves (the host reports it: this must not happen, and a hang would cost the whole GPU).
// ------------------------------------------------------------------------------------------
constexpr uint32_t AQ_WAKE = 1u, AQ_RUN = 2u, AQ_EMPTY = 0xFFFFFFFFu, AQ_NONE = 0xFFFFFFFEu, AQ_DONE = 0xFFFFFFFDu;
// One word saturates at ~90 atomics per microsecond on this part (MI355X_MICROARCH.md, "dequeue"), a tail of 600 000 visits
// and 4 000 polling wavefronts on one head / tail / pending word took SECONDS.  So: AQ_NQ queues, tile t in queue
// t % AQ_NQ, wavefront i serving queue i % AQ_NQ, and the termination test on sharded MONOTONE counters: enq[] (tiles
// that left the idle state) and done[] (visits that ended idle).  Reading every done[] first and every enq[] after, equal
// sums mean the search was over when the done[] reads ended -- each enq[] read is at least its value then, and
// enq >= done at all times.  Any idle wavefront may run the test (wavefront 0 often, the others rarely) and raises the
// per-queue finished words that every idle wavefront polls together with its queue's head / tail.
constexpr int AQ_NQ = 64, AQ_STRIDE = 64;   // control words per queue: [0] head [1] tail [2] finished | [32] enq [33] done
constexpr int AQ_G_ABORT = AQ_NQ * AQ_STRIDE, AQ_G_VISITS = AQ_G_ABORT + 32, AQ_G_BUSY = AQ_G_ABORT + 34 /* 64-bit: ticks in visits */,
              AQ_G_SPAN = AQ_G_ABORT + 36 /* longest wavefront lifetime, ticks */, AQ_WORDS = AQ_G_ABORT + 64;
struct AsyncQ {
  uint32_t *q, *state, *ctl;
  uint32_t qmask;   // slots per queue - 1
};

__device__ __forceinline__ uint32_t aq_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr uint32_t AQ_SPIN_CAP = 1u << 22;   // polls of one slot before the launch is given up (never reached unless the protocol is broken)
__device__ __forceinline__ void aq_push(const AsyncQ &Q, uint32_t tile) {
  const uint32_t qi = tile % AQ_NQ;
  const uint32_t slot = atomicAdd(&Q.ctl[qi * AQ_STRIDE + 1], 1u) & Q.qmask;
  uint32_t *const cell = &Q.q[(size_t)qi * (Q.qmask + 1u) + slot];
  for (uint32_t spin = 0; atomicCAS(cell, AQ_EMPTY, tile) != AQ_EMPTY; spin++) {
    if (spin > AQ_SPIN_CAP) { atomicExch(&Q.ctl[AQ_G_ABORT], 2u); return; }
    __builtin_amdgcn_s_sleep(2);
  }
}

// the queues' first content: the list the last compaction made (one thread per entry)
__global__ __launch_bounds__(NTHR) void k_async_init(const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ count, AsyncQ Q) {
  const uint32_t n = *count, i = blockIdx.x * NTHR + threadIdx.x;
  if (i < n) {
    const uint32_t t = tiles[i], qi = t % AQ_NQ;
    const uint32_t slot = atomicAdd(&Q.ctl[qi * AQ_STRIDE + 1], 1u) & Q.qmask;
    Q.q[(size_t)qi * (Q.qmask + 1u) + slot] = t;
    Q.state[t] = AQ_WAKE;
    atomicAdd(&Q.ctl[qi * AQ_STRIDE + 32], 1u);
  }
}

// all lanes: is the search over?  (see above; lane = queue)
__device__ __forceinline__ bool aq_terminated(const AsyncQ &Q, int lane) {
  uint32_t d = aq_load(&Q.ctl[lane * AQ_STRIDE + 33]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o, 64);
  // (the shuffles consumed every done[] value: those loads have returned before the enq[] loads are issued)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t e = aq_load(&Q.ctl[lane * AQ_STRIDE + 32]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) e += __shfl_xor(e, o, 64);
  return e == d;
}

// (r06, the plane visits: 168 registers -- the resident wavefronts share their SIMDs' register files with the other search's
// rounds, which is how the two searches slow each other down: at 128 registers the tail spills and loses 2 ms while the away
// rounds gain 2.4, at 187 the reverse; 22.6 / 23.7 / 22.8 ms for the stage at 168 / 128 / 187)
template <int SEED_LEVEL, bool PL = false>
__global__ __launch_bounds__(NTHR, PL ? 3 : 4) void k_relax_bits_async(const unsigned long long *__restrict__ mbits, int32_t *D, AsyncQ Q, int w,
                                                           int h, RowWin win, uint32_t tilesX, uint32_t tilesY,
                                                           unsigned long long tick_budget, int nap, PlaneField pf = PlaneField{}) {
  static_assert(AQ_NQ == 64, "one lane per queue in the termination test");
  __shared__ uint16_t open_rows[NTHR / 64][PL ? 2 : BT * BT];   // (the plane visits keep nothing in LDS)
  const int lane = threadIdx.x & 63;
  uint16_t *const orow = open_rows[threadIdx.x >> 6];
  const uint32_t me = blockIdx.x * (NTHR / 64) + (threadIdx.x >> 6), nworkers = gridDim.x * (NTHR / 64);
  uint32_t *const homectl = Q.ctl + (size_t)(me % AQ_NQ) * AQ_STRIDE;   // the shard of the counters this wavefront adds to
  // the queue it serves: its own, and -- when the launch has fewer wavefronts than a multiple of AQ_NQ covers -- the
  // next one after every poll that found nothing, so that no queue is left without a wavefront
  uint32_t serve = me % AQ_NQ;
  const unsigned long long t_start = wall_clock64();
  uint32_t idle_polls = 0, visits = 0;
  unsigned long long busy = 0;
  for (;;) {
    uint32_t tile = AQ_NONE;
    uint32_t *const myctl = Q.ctl + (size_t)serve * AQ_STRIDE;
    uint32_t *const myq = Q.q + (size_t)serve * (Q.qmask + 1u);
    if (lane == 0) {
      unsigned long long ht = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(myctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t hd = (uint32_t)ht;
      bool got = false;
      while ((int32_t)((uint32_t)(ht >> 32) - hd) > 0) {
        const uint32_t seen = atomicCAS(&myctl[0], hd, hd + 1u);
        if (seen == hd) { got = true; break; }
        hd = seen;
        ht = (ht & 0xFFFFFFFF00000000ull) | hd;
      }
      if (got) {
        uint32_t *const cell = &myq[hd & Q.qmask];
        uint32_t v, spin = 0;
        while ((v = atomicExch(cell, AQ_EMPTY)) == AQ_EMPTY && ++spin <= AQ_SPIN_CAP) __builtin_amdgcn_s_sleep(1);   // its pusher has the ticket, the store is on its way
        if (v == AQ_EMPTY) { atomicExch(&Q.ctl[AQ_G_ABORT], 3u); tile = AQ_DONE; }
        else { tile = v; atomicExch(&Q.state[tile], AQ_RUN); }   // (queued: AQ_WAKE alone)
      } else if (aq_load(&myctl[2]) != 0u) {
        tile = AQ_DONE;
      }
    }
    // The state word must BE "running" before the ring is read: a neighbour that stores, drains and then finds the tile
    // still "queued" relies on the coming visit to see its stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
    if (tile == AQ_DONE) break;
    if (tile == AQ_NONE) {
      idle_polls++;
      serve = (serve + nworkers) % AQ_NQ;   // (unchanged when the wavefronts cover the queues evenly; else the residues mod gcd(nworkers, 64) are covered by consecutive wavefronts)
      if (me == 0u ? (idle_polls & 3u) == 0u : (idle_polls & 255u) == 0u) {
        const bool late = wall_clock64() - t_start > tick_budget;
        if (late && lane == 0) atomicExch(&Q.ctl[AQ_G_ABORT], 1u);
        const bool stop = late || aq_load(&Q.ctl[AQ_G_ABORT]) != 0u || aq_terminated(Q, lane);
        if (stop) { __hip_atomic_store(&Q.ctl[lane * AQ_STRIDE + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      // back off: a wavefront that has been idle for a while polls less often (the polls share the memory channels with the work)
      for (int k = 0, n = idle_polls < 8u ? 1 : idle_polls < 32u ? 2 * nap : 4 * nap; k < n; k++) __builtin_amdgcn_s_sleep(32);
      continue;
    }
    idle_polls = 0;
    visits++;
    const unsigned long long tv0 = wall_clock64();
    const uint32_t wake = PL ? relax_visit_p<SEED_LEVEL, true>(mbits, pf.expanded, pf, tile, w, h, tilesX, tilesY)
                             : relax_visit<SEED_LEVEL, false, true>(mbits, nullptr, D, tile, orow, false, w, h, win, tilesX, tilesY);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the visit's write-through stores have landed
    busy += wall_clock64() - tv0;
    // Wake the neighbours: an idle one goes into its queue (a queued one will see the stores when it is visited, a running
    // one is queued again by its own wavefront).  Measured and dropped (r03c): the waking wavefront visiting one of the
    // woken tiles itself, without the trip through the queue -- 40.6 vs 39.8 ms at S3: the tail is bound by the number
    // of visits (0.75-1.0 M of 18-26 us for 7500 tiles at the switch), not by the hops of its longest chain.
    if (lane < 9 && lane != 4 && (wake >> lane & 1u)) {
      const int tx = (int)(tile % tilesX), ty = (int)(tile / tilesX);
      const int ntx = tx + lane % 3 - 1, nty = ty + lane / 3 - 1;
      if (ntx >= 0 && nty >= 0 && ntx < (int)tilesX && nty < (int)tilesY) {
        const uint32_t nt = (uint32_t)nty * tilesX + (uint32_t)ntx;
        if (atomicOr(&Q.state[nt], AQ_WAKE) == 0u) {
          atomicAdd(&homectl[32], 1u);
          aq_push(Q, nt);
        }
      }
    }
    if (lane == 0) {
      const uint32_t old = atomicAnd(&Q.state[tile], ~AQ_RUN);
      if (old & AQ_WAKE) aq_push(Q, tile);   // woken while it ran: once more (it never was idle)
      else atomicAdd(&homectl[33], 1u);
    }
  }
  if (lane == 0) {
    if (visits) {
      atomicAdd(&Q.ctl[AQ_G_VISITS], visits);
      atomicAdd(reinterpret_cast<unsigned long long *>(&Q.ctl[AQ_G_BUSY]), busy);
    }
    atomicMax(&Q.ctl[AQ_G_SPAN], (uint32_t)(wall_clock64() - t_start));
  }
}

// flat_height[label] = deepest away level of the flat (:181): atomicMax behind a coherent pre-check
__global__ __launch_bounds__(NTHR) void k_flat_height(const int32_t *__restrict__ A, const uint32_t *__restrict__ L,
                                                      int32_t *fh, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int32_t a = A[c];
    if (a >= DINF) continue;
    const uint32_t r = L[c];
    if (__hip_atomic_load(&fh[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a) atomicMax(&fh[r], a);
  }
}

// the same, four cells a thread
__global__ __launch_bounds__(NTHR) void k_flat_height4(const int32_t *__restrict__ A, const uint32_t *__restrict__ L, int32_t *fh,
                                                       uint64_t n) {
  const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * NTHR;
  auto one = [&](int32_t a, uint32_t r) {
    if (__hip_atomic_load(&fh[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a) atomicMax(&fh[r], a);
  };
  for (uint64_t q = (uint64_t)blockIdx.x * NTHR + threadIdx.x; q < n4 + (n & 3); q += stride) {
    if (q < n4) {
      const int4 a = reinterpret_cast<const int4 *>(A)[q];
      if (a.x >= DINF && a.y >= DINF && a.z >= DINF && a.w >= DINF) continue;
      const uint4 l = reinterpret_cast<const uint4 *>(L)[q];
      if (a.x < DINF) one(a.x, l.x);
      if (a.y < DINF) one(a.y, l.y);
      if (a.z < DINF) one(a.z, l.z);
      if (a.w < DINF) one(a.w, l.w);
    } else {
      const uint64_t c = 4 * n4 + (q - n4);
      if (A[c] < DINF) one(A[c], L[c]);
    }
  }
}

// flat_mask from the two distance fields, in place over the towards distances (:279-284):
//   low edge -> 2;  NO_FLOW cell at towards level t, away level a: (a reached ? flat_height - a : 0) + 2t
__global__ __launch_bounds__(NTHR) void k_flat_combine(int32_t *M /* in: towards level */, const int32_t *__restrict__ A,
                                                       const uint32_t *__restrict__ L, const int32_t *__restrict__ fh,
                                                       uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int32_t t = M[c];
    int32_t m = 0;
    if (t < DINF) {
      const int32_t a = A ? A[c] : DINF;
      m = (a < DINF ? fh[L[c]] - a : 0) + 2 * t;   // a low edge has t = 1 and no away level: 2
    }
    M[c] = m;
  }
}

// d8_masked_FlowDir (:42-65) for the NO_FLOW cells of drainable flats (M > 0), d8_flow_flats :96-116.
template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_dirs(const T *__restrict__ z, const int32_t *__restrict__ M,
                                                    uint8_t *dirs, int w, int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[SLH * SLW];
  __shared__ int32_t sm[SLH * SLW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * SH;
  for (int i = threadIdx.x; i < SLH * SLW; i += NTHR) {
    const int ly = i / SLW, lx = i - ly * SLW;
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
    T v = T();
    int32_t m = 0;
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
      const size_t g = (size_t)gy * w + gx;
      v = z[g];
      m = M[g];
    }
    sz[i] = v;
    sm[i] = m;
  }
  __syncthreads();
  const int lx = threadIdx.x & (SW - 1), ly0 = threadIdx.x >> 6;
  const int off[9] = {0, -1, -SLW - 1, -SLW, -SLW + 1, 1, SLW + 1, SLW, SLW - 1};
#pragma unroll
  for (int j = 0; j < SH / 4; j++) {
    const int ly = ly0 + 4 * j;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) continue;   // interior only (:108-109)
    const int o = (ly + 1) * SLW + lx + 1;
    const int32_t mc = sm[o];
    if (mc <= 0) continue;                 // not in a drainable flat
    const size_t g = (size_t)gy * w + gx;
    if (dirs[g] != 0) continue;            // low edges keep their direction (:112)
    const T e = sz[o];
    int32_t m = mc;
    int dir = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
      if (!(sz[o + off[k]] == e)) continue;                         // labels(n) != labels(c), :56-57
      const int32_t v = sm[o + off[k]];
      if (v < m || (v == m && dir > 0 && (dir & 1) == 0 && (k & 1) == 1)) {
        m = v;
        dir = k;
      }
    }
    dirs[g] = (uint8_t)dir;
  }
}

// Directions straight from the two level fields (the directions-only path of barnes_flat_resolution_d8): inside one flat
// flat_mask = (flat_height - away) + 2 * towards differs from 2 * towards - away by the flat's constant, low edges
// (towards level 1) hold mask 2 -- below every NO_FLOW cell of their flat, whose towards level is at least 2 -- and
// d8_masked_FlowDir (:42-65) only COMPARES masks of cells of one flat.  So neither the flat heights nor the labels
// behind them (k_ccl_*, k_flat_height, k_flat_combine: 45 ms of the stage at S3) are needed for the directions.
template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_dirs_levels(const T *__restrict__ z, const int32_t *__restrict__ TW,
                                                           const int32_t *__restrict__ AW, uint8_t *dirs, int w, int h,
                                                           uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[KLLH * SLW];
  __shared__ int32_t sm[KLLH * SLW];
  constexpr int32_t LOWEDGE = INT32_MIN, NOTFLAT = INT32_MAX;
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * KLH;
  if (window_inside(x0, y0, w, h, SW, KLH, 1)) {
    // the window lies inside the raster: four cells per load (see stage_window_inside), the two level fields combined on the way
    stage_window_inside<T, SW, KLH, 1, SLW, NTHR>(z, w, x0, y0, sz);
    constexpr int QPR = SW / 4, NQ = KLLH * QPR, QPT = (NQ + NTHR - 1) / NTHR, NHC = KLLH * 2;
    struct Q4 { int32_t v[4]; };
    const int32_t *const tb = TW + ((size_t)(y0 - 1) * w + (size_t)(x0 - 1)), *const ab = AW ? AW + ((size_t)(y0 - 1) * w + (size_t)(x0 - 1)) : nullptr;
    Q4 tq[QPT], aq[QPT];
    int32_t th = DINF, ah = DINF;
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      const int ly = i / QPR, qq = i - ly * QPR;
      if (i < NQ) {
        __builtin_memcpy(&tq[r], tb + (uint32_t)(ly * w + 1 + 4 * qq), sizeof(Q4));
        if (ab) __builtin_memcpy(&aq[r], ab + (uint32_t)(ly * w + 1 + 4 * qq), sizeof(Q4));
      }
    }
    static_assert(NHC <= NTHR, "one halo cell per thread");
    if ((int)threadIdx.x < NHC) {
      const int ly = (int)threadIdx.x >> 1, c = (int)threadIdx.x & 1;
      th = tb[(uint32_t)(ly * w + (c ? SW + 1 : 0))];
      if (ab) ah = ab[(uint32_t)(ly * w + (c ? SW + 1 : 0))];
    }
    auto mask = [&](int32_t tv_, int32_t av_) -> int32_t { return tv_ >= DINF ? NOTFLAT : tv_ == 1 ? LOWEDGE : 2 * tv_ - (av_ < DINF ? av_ : 0); };
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      const int ly = i / QPR, qq = i - ly * QPR;
      if (i < NQ) {
#pragma unroll
        for (int e = 0; e < 4; e++) sm[ly * SLW + 1 + 4 * qq + e] = mask(tq[r].v[e], ab ? aq[r].v[e] : DINF);
      }
    }
    if ((int)threadIdx.x < NHC) {
      const int ly = (int)threadIdx.x >> 1, c = (int)threadIdx.x & 1;
      sm[ly * SLW + (c ? SW + 1 : 0)] = mask(th, ah);
    }
  } else {
    constexpr int IPT = (KLLH * SLW + NTHR - 1) / NTHR;
    T zv[IPT];
    int32_t tv[IPT], av[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {   // all loads of the thread in flight together (clamped addresses)
      const int i = min((int)threadIdx.x + r * NTHR, KLLH * SLW - 1);
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      const size_t g = (size_t)gy * w + gx;
      zv[r] = z[g];
      tv[r] = TW[g];
      av[r] = AW ? AW[g] : DINF;
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i >= KLLH * SLW) continue;
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
      const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
      sz[i] = in ? zv[r] : T();
      sm[i] = (!in || tv[r] >= DINF) ? NOTFLAT : tv[r] == 1 ? LOWEDGE : 2 * tv[r] - (av[r] < DINF ? av[r] : 0);
    }
  }
  __syncthreads();
  // a wavefront owns a band of 8 consecutive rows, a lane one column; the 3 x 3 windows slide down in registers
  const int lx = threadIdx.x & (SW - 1), yb = (int)(threadIdx.x >> 6) * (KLH / 4);
  const int gx = x0 + lx;
  T z0[3], z1[3], z2[3];
  int32_t m0[3], m1[3], m2[3];
#pragma unroll
  for (int e = 0; e < 3; e++) {
    z0[e] = sz[yb * SLW + lx + e]; z1[e] = sz[(yb + 1) * SLW + lx + e];
    m0[e] = sm[yb * SLW + lx + e]; m1[e] = sm[(yb + 1) * SLW + lx + e];
  }
#pragma unroll
  for (int j = 0; j < KLH / 4; j++) {
    const int gy = y0 + yb + j;
#pragma unroll
    for (int e = 0; e < 3; e++) { z2[e] = sz[(yb + j + 2) * SLW + lx + e]; m2[e] = sm[(yb + j + 2) * SLW + lx + e]; }
    const int32_t mc = m1[1];
    // interior only (:108-109); a NO_FLOW cell of a drainable flat (low edges keep their direction, :112)
    if (gx > 0 && gy > 0 && gx < w - 1 && gy < h - 1 && mc != NOTFLAT && mc != LOWEDGE) {
      // neighbours 1..8 in the 234/105/876 numbering
      const T zn[9] = {z1[1], z1[0], z0[0], z0[1], z0[2], z1[2], z2[2], z2[1], z2[0]};
      const int32_t mn[9] = {m1[1], m1[0], m0[0], m0[1], m0[2], m1[2], m2[2], m2[1], m2[0]};
      const T e = z1[1];
      int32_t m = mc;
      int dir = 0;
#pragma unroll
      for (int k = 1; k <= 8; k++) {   // selects, no branches (see k_flat_classify)
        const int32_t v = mn[k];
        const bool same = (zn[k] == e) & (v != NOTFLAT);   // labels(n) == labels(c), :56-57 (an equal neighbour outside the flat cannot occur: kept safe)
        const bool take = same & ((v < m) | ((v == m) & (dir > 0) & ((dir & 1) == 0) & ((k & 1) == 1)));
        m = take ? v : m;
        dir = take ? k : dir;
      }
      dirs[(size_t)gy * w + gx] = (uint8_t)dir;
    }
#pragma unroll
    for (int e = 0; e < 3; e++) { z0[e] = z1[e]; z1[e] = z2[e]; m0[e] = m1[e]; m1[e] = m2[e]; }
  }
}

// The same directions WITHOUT the DEM, from 4 bits per cell (r05; the default of the directions-only entry).  What
// d8_masked_FlowDir compares are masks of adjacent cells of one flat, and
//   * two adjacent NO_FLOW cells always lie in one flat (equal elevation: a higher one would have a direction), are reached
//     by the towards search together or not at all, likewise by the away search, and their levels differ by at most one in
//     either field (breadth-first levels of adjacent vertices): |mask(n) - mask(c)| = |2 dT - dA| <= 3, so the masks
//     MODULO 8 order a cell's neighbourhood exactly;
//   * a neighbour WITH a direction only counts as a low edge of the cell's own flat, and a cell that has one (towards level
//     2, exactly the F_NEAR cells) got its final direction in k_dirs_classify<NEARDIRS> already;
//   * a cell takes part iff its towards level lies in [2, DINF): low edges hold 1, everything else DINF.
// So the window is staged as ONE BYTE per cell -- bits 0-2 the mask modulo 8, bit 3 "does not take part", bit 4 "has its
// direction" -- four cells per LDS store, and a neighbour costs a subtraction, two ANDs, an OR and two compares: no
// elevations read or compared (k_flat_dirs_levels: 20.4 GB fetched, 4.5 ms at S3).  Measured and dropped on the way
// (profiles/r05b_flats_bytes_ab.json): byte planes written by the searches beside their 32-bit levels, so that this pass
// reads 3 bytes per cell -- the stores slowed every visit and the start levels (36.1 against 32.6 ms for the stage): the
// stage is bound by instruction issue and dependent visits, not by these bytes.
constexpr int QLW = SW + 8;   // bytes per staged row: the tile's first column at byte 4 (its quads are aligned words), the halo at 3 and 68
__device__ __forceinline__ uint32_t flat_q(int32_t tv, int32_t av) {
  const uint32_t m = (uint32_t)(2 * tv - (av < DINF ? av : 0));
  const bool part = (uint32_t)(tv - 2) < (uint32_t)(DINF - 2);
  return part ? ((m & 7u) | (tv == 2 ? 16u : 0u)) : 8u;
}
__global__ __launch_bounds__(NTHR) void k_flat_dirs_q(const int32_t *__restrict__ TW, const int32_t *__restrict__ AW, uint8_t *dirs,
                                                      int w, int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ __attribute__((aligned(4))) uint8_t sq[KLLH * QLW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * KLH;
  if (window_inside(x0, y0, w, h, SW, KLH, 1)) {
    constexpr int QPR = SW / 4, NQ = KLLH * QPR, QPT = (NQ + NTHR - 1) / NTHR, NHC = KLLH * 2;
    struct Q4 { int32_t v[4]; };
    const int32_t *const tb = TW + ((size_t)(y0 - 1) * w + (size_t)(x0 - 1)), *const ab = AW ? AW + ((size_t)(y0 - 1) * w + (size_t)(x0 - 1)) : nullptr;
    Q4 tq[QPT], aq[QPT];
    int32_t th = DINF, ah = DINF;
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      const int ly = i / QPR, qq = i - ly * QPR;
      if (i < NQ) {
        __builtin_memcpy(&tq[r], tb + (uint32_t)(ly * w + 1 + 4 * qq), sizeof(Q4));
        if (ab) __builtin_memcpy(&aq[r], ab + (uint32_t)(ly * w + 1 + 4 * qq), sizeof(Q4));
      }
    }
    static_assert(NHC <= NTHR, "one halo cell per thread");
    if ((int)threadIdx.x < NHC) {
      const int ly = (int)threadIdx.x >> 1, c = (int)threadIdx.x & 1;
      th = tb[(uint32_t)(ly * w + (c ? SW + 1 : 0))];
      if (ab) ah = ab[(uint32_t)(ly * w + (c ? SW + 1 : 0))];
    }
#pragma unroll
    for (int r = 0; r < QPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      const int ly = i / QPR, qq = i - ly * QPR;
      if (i < NQ) {
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) word |= flat_q(tq[r].v[e], ab ? aq[r].v[e] : DINF) << (8 * e);
        *reinterpret_cast<uint32_t *>(&sq[ly * QLW + 4 + 4 * qq]) = word;
      }
    }
    if ((int)threadIdx.x < NHC) {
      const int ly = (int)threadIdx.x >> 1, c = (int)threadIdx.x & 1;
      sq[ly * QLW + (c ? SW + 4 : 3)] = (uint8_t)flat_q(th, ah);
    }
  } else {
    constexpr int IPT = (KLLH * SLW + NTHR - 1) / NTHR;
    int32_t tv[IPT], av[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {   // all loads of the thread in flight together (clamped addresses)
      const int i = min((int)threadIdx.x + r * NTHR, KLLH * SLW - 1);
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      const size_t g = (size_t)gy * w + gx;
      tv[r] = TW[g];
      av[r] = AW ? AW[g] : DINF;
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i >= KLLH * SLW) continue;
      const int ly = i / SLW, lx = i - ly * SLW;
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
      const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
      sq[ly * QLW + 3 + lx] = in ? (uint8_t)flat_q(tv[r], av[r]) : (uint8_t)8;
    }
  }
  __syncthreads();
  // a wavefront owns a band of 8 consecutive rows, a lane one column; the 3 x 3 windows slide down in registers
  const int lx = threadIdx.x & (SW - 1), yb = (int)(threadIdx.x >> 6) * (KLH / 4);
  const int gx = x0 + lx;
  uint32_t c0[3], c1[3], c2[3];
#pragma unroll
  for (int e = 0; e < 3; e++) { c0[e] = sq[yb * QLW + 3 + lx + e]; c1[e] = sq[(yb + 1) * QLW + 3 + lx + e]; }
#pragma unroll
  for (int j = 0; j < KLH / 4; j++) {
    const int gy = y0 + yb + j;
#pragma unroll
    for (int e = 0; e < 3; e++) c2[e] = sq[(yb + j + 2) * QLW + 3 + lx + e];
    const uint32_t cc = c1[1];
    // interior only (:108-109); a NO_FLOW cell of a drainable flat that has no direction yet
    if (gx > 0 && gy > 0 && gx < w - 1 && gy < h - 1 && (cc & 24u) == 0u) {
      const uint32_t cn[9] = {c1[1], c1[0], c0[0], c0[1], c0[2], c1[2], c2[2], c2[1], c2[0]};   // neighbours 1..8: 234/105/876
      const uint32_t qc = (cc & 7u) - 4u;   // rb = mask(n) - mask(c) + 4 in 1 .. 7 for a neighbour that takes part, >= 8 otherwise
      uint32_t m = 4u;
      int dir = 0;
#pragma unroll
      for (int k = 1; k <= 8; k++) {
        const uint32_t v = cn[k], rb = ((v - qc) & 7u) | (v & 8u);
        const bool take = rb < m || (rb == m && dir > 0 && (dir & 1) == 0 && (k & 1) == 1);
        m = take ? rb : m;
        dir = take ? k : dir;
      }
      dirs[(size_t)gy * w + gx] = (uint8_t)dir;
    }
#pragma unroll
    for (int e = 0; e < 3; e++) { c0[e] = c1[e]; c1[e] = c2[e]; }
  }
}

static inline uint32_t stencil_tiles(int w, int h, uint32_t *tilesX) {
  *tilesX = (uint32_t)((w + SW - 1) / SW);
  return *tilesX * (uint32_t)((h + SH - 1) / SH);
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
static thread_local rdgpu_flat_stats g_fstats;
static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

template <class T>
static void launch_classify(const T *d_z, const uint8_t *d_dirs, int w, int h, uint8_t *flags, hipStream_t s) {
  const uint32_t tilesX = (w + SW - 1) / SW, ntiles = tilesX * ((h + KLH - 1) / KLH);
  RD_LAUNCH("flats.classify", (k_flat_classify<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, d_dirs, w, h, flags, tilesX,
            ntiles);
}
template <class T>
static void launch_masked_dirs(const T *d_z, const int32_t *M, uint8_t *d_dirs, int w, int h, hipStream_t s) {
  uint32_t tilesX;
  const uint32_t ntiles = stencil_tiles(w, h, &tilesX);
  RD_LAUNCH("flats.masked_dirs", (k_flat_dirs<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, M, d_dirs, w, h, tilesX, ntiles);
}
template <class T>
static void launch_ccl_border(const T *d_z, uint32_t *L, int w, int h, hipStream_t s) {
  const uint32_t tilesX = (w + CW - 1) / CW, ntiles = tilesX * ((h + CH - 1) / CH);
  RD_LAUNCH("flats.ccl_border", (k_ccl_border<T>), dim3(sgrid((uint64_t)ntiles * (CW + 2 * CH))), dim3(NTHR), 0, s, d_z, L, w, h,
            tilesX, ntiles);
}

// ---- the two edge lists (and the NO_FLOW count) in ONE count pass and ONE fill pass over the flag raster ----
__global__ __launch_bounds__(NTHR) void k_flag_count3(const uint8_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ cl,
                                                      uint32_t *__restrict__ cn, uint32_t *__restrict__ ch) {
  __shared__ uint32_t ws[3][NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t a = 0, b = 0, c = 0;
#pragma unroll 4
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t i = base + (uint64_t)j * NTHR + threadIdx.x;
    const uint8_t f = i < n ? flags[i] : 0;
    a += (f & F_LOW) != 0; b += (f & F_NOFLOW) != 0; c += (f & F_HIGH) != 0;
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = a; ws[1][threadIdx.x >> 6] = b; ws[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cl[blockIdx.x] = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
    cn[blockIdx.x] = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
    ch[blockIdx.x] = ws[2][0] + ws[2][1] + ws[2][2] + ws[2][3];
  }
}

__global__ __launch_bounds__(NTHR) void k_flag_fill2(const uint8_t *__restrict__ flags, uint64_t n,
                                                     const uint32_t *__restrict__ offl, const uint32_t *__restrict__ offh,
                                                     uint32_t *__restrict__ outl, uint32_t *__restrict__ outh) {
  __shared__ uint32_t ws[2][NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t runl = offl[blockIdx.x], runh = offh[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t c = base + (uint64_t)j * NTHR + threadIdx.x;
    const uint8_t f = c < n ? flags[c] : 0;
    const bool hl = (f & F_LOW) != 0, hh = (f & F_HIGH) != 0;
    const unsigned long long bl = __ballot(hl), bh = __ballot(hh);
    if (lane == 0) { ws[0][wv] = __popcll(bl); ws[1][wv] = __popcll(bh); }
    __syncthreads();
    uint32_t wl = 0, tl = 0, wh = 0, th = 0;
#pragma unroll
    for (int k = 0; k < NTHR / 64; k++) {
      if (k < wv) { wl += ws[0][k]; wh += ws[1][k]; }
      tl += ws[0][k]; th += ws[1][k];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (hl) outl[runl + wl + __popcll(bl & below)] = (uint32_t)c;
    if (hh) outh[runh + wh + __popcll(bh & below)] = (uint32_t)c;
    runl += tl; runh += th;
    __syncthreads();
  }
}

// totals of the three per-block count arrays (the counts-only case needs no offsets: one launch instead of three scans)
__global__ __launch_bounds__(NTHR) void k_flag_totals(const uint32_t *__restrict__ cl, const uint32_t *__restrict__ cn,
                                                      const uint32_t *__restrict__ ch, uint32_t nblk, uint32_t *tot) {
  uint32_t a = 0, b = 0, c = 0;
  for (uint32_t i = blockIdx.x * NTHR + threadIdx.x; i < nblk; i += gridDim.x * NTHR) { a += cl[i]; b += cn[i]; c += ch[i]; }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(tot, a); atomicAdd(tot + 1, b); atomicAdd(tot + 2, c); }
}

static void compact_edges(const uint8_t *flags, uint64_t n, uint32_t **low, uint32_t *nlow, uint32_t **high, uint32_t *nhigh,
                          uint32_t *nnoflow, hipStream_t s, bool lists = true) {
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  const uint32_t nblk = (uint32_t)((n + CPB - 1) / CPB);
  uint32_t *counts = ws.buf<uint32_t>("flats.counts3", 3 * ((size_t)nblk + 1));
  uint32_t *cl = counts, *cn = counts + (nblk + 1), *ch = counts + 2 * ((size_t)nblk + 1);
  RD_LAUNCH("flats.flag_count", k_flag_count3, dim3(nblk), dim3(NTHR), 0, s, flags, n, cl, cn, ch);
  if (!lists) {
    uint32_t *tot = ws.buf<uint32_t>("flats.totals3", 4);
    RD_HIP(hipMemsetAsync(tot, 0, 3 * sizeof(uint32_t), s));
    RD_LAUNCH("flats.flag_totals", k_flag_totals, dim3(64), dim3(NTHR), 0, s, (const uint32_t *)cl, (const uint32_t *)cn,
              (const uint32_t *)ch, nblk, tot);
    RD_HIP(hipMemcpyAsync(hw, tot, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    *nlow = hw[0]; *nnoflow = hw[1]; *nhigh = hw[2];
    *low = *high = nullptr;
    return;
  }
  RD_LAUNCH("flats.flag_scan", k_flag_scan, dim3(1), dim3(1024), 0, s, cl, nblk, cl + nblk);
  RD_LAUNCH("flats.flag_scan", k_flag_scan, dim3(1), dim3(1024), 0, s, cn, nblk, cn + nblk);
  RD_LAUNCH("flats.flag_scan", k_flag_scan, dim3(1), dim3(1024), 0, s, ch, nblk, ch + nblk);
  RD_HIP(hipMemcpyAsync(hw, cl + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipMemcpyAsync(hw + 1, cn + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipMemcpyAsync(hw + 2, ch + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  *nlow = hw[0]; *nnoflow = hw[1]; *nhigh = hw[2];
  *low = *high = nullptr;
  if (*nlow == 0 || !lists) return;   // nothing will be resolved / the bitmap engine takes its seeds from the flags: counts only
  *low = ws.buf<uint32_t>("flats.low", *nlow);
  *high = ws.buf<uint32_t>("flats.highall", std::max<uint32_t>(*nhigh, 1u));
  RD_LAUNCH("flats.flag_fill", k_flag_fill2, dim3(nblk), dim3(NTHR), 0, s, flags, n, (const uint32_t *)cl, (const uint32_t *)ch,
            *low, *high);
}

// list of the cells with flags & mask, in index order; returns the count
static uint32_t compact_flags(const uint8_t *flags, uint8_t mask, uint64_t n, const char *name, uint32_t **out,
                              hipStream_t s) {
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  const uint32_t nblk = (uint32_t)((n + CPB - 1) / CPB);
  uint32_t *counts = ws.buf<uint32_t>("flats.counts", (size_t)nblk + 1);
  RD_LAUNCH("flats.flag_count", k_flag_count, dim3(nblk), dim3(NTHR), 0, s, flags, mask, n, counts);
  RD_LAUNCH("flats.flag_scan", k_flag_scan, dim3(1), dim3(1024), 0, s, counts, nblk, counts + nblk);
  RD_HIP(hipMemcpyAsync(hw, counts + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  const uint32_t total = hw[0];
  *out = nullptr;
  if (out && name && total) {
    *out = ws.buf<uint32_t>(name, total);
    RD_LAUNCH("flats.flag_fill", k_flag_fill, dim3(nblk), dim3(NTHR), 0, s, flags, mask, n, (const uint32_t *)counts, *out);
  }
  return total;
}

// Relaxation rounds until no tile is active.  Rounds are enqueued RELAX_BATCH at a time (compact the active
// tile flags into a list + count on the device, relax that list); the host only reads the counts back once
// per batch, and the rounds enqueued past the fixed point see an empty list.  Returns the rounds that had work.
static uint32_t relax_rounds(const uint8_t *d_dirs, int32_t *D, uint8_t *tflags, uint32_t *tlist,
                             uint32_t *ctr /* RELAX_BATCH words */, int w, int h, int row_lo, int row_hi, const char *name,
                             hipStream_t s) {
  uint32_t *hw = Workspace::get().host_words();
  const uint32_t tilesX = (w + CW - 1) / CW, tilesY = (h + RCH - 1) / RCH, ntiles = tilesX * tilesY;
  const bool trace = getenv("RDGPU_FLAT_TRACE") != nullptr;
  uint32_t rounds = 0, grid = ntiles;   // any tile may be active in the first batch
  for (;;) {
    RD_HIP(hipMemsetAsync(ctr, 0, RELAX_BATCH * sizeof(uint32_t), s));
    for (int b = 0; b < RELAX_BATCH; b++) {
      RD_LAUNCH("flats.tiles_compact", k_tiles_compact, dim3((ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, tflags, ntiles,
                tlist, ctr + b);
      RD_LAUNCH(name, k_flat_relax, dim3(grid), dim3(RNT), 0, s, d_dirs, D, (const uint32_t *)tlist,
                (const uint32_t *)(ctr + b), tflags, w, h, tilesX, tilesY, row_lo, row_hi);
    }
    RD_HIP(hipMemcpyAsync(hw, ctr, RELAX_BATCH * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    uint32_t most = 0;
    for (int b = 0; b < RELAX_BATCH; b++) {
      if (trace) fprintf(stderr, "%s round %u nact %u (grid %u)\n", name, rounds, hw[b], grid);
      if (hw[b] == 0) return rounds;
      most = std::max(most, hw[b]);
      rounds++;
    }
    grid = std::min<uint32_t>(ntiles, std::max<uint32_t>(1024u, 2u * most));
    if (rounds > (1u << 26)) throw Error(RDGPU_ERR_HIP, "rdgpu flat resolution: relaxation did not terminate");
  }
}

struct RelaxScratch {
  uint8_t *tflags;
  uint32_t *tlist, *ctr;
  uint32_t tilesX, tilesY, ntiles;
};
static RelaxScratch relax_scratch(int w, int h) {
  Workspace &ws = Workspace::get();
  RelaxScratch r;
  r.tilesX = (w + CW - 1) / CW; r.tilesY = (h + RCH - 1) / RCH; r.ntiles = r.tilesX * r.tilesY;
  r.tflags = ws.buf<uint8_t>("flats.tflags", r.ntiles);
  r.tlist = ws.buf<uint32_t>("flats.tlist", r.ntiles);
  r.ctr = ws.buf<uint32_t>("flats.tctr", BITS_BATCH);
  return r;
}

// Away levels from the high edges listed in src (level 1) by tile relaxation.  Returns the number of rounds.
static uint32_t run_relax_away(const uint8_t *d_dirs, int32_t *D, const uint32_t *src, uint32_t nsrc, const uint32_t *L,
                               const int32_t *fh_filter, int w, int h, hipStream_t s) {
  const RelaxScratch r = relax_scratch(w, h);
  RD_HIP(hipMemsetAsync(r.tflags, 0, r.ntiles, s));
  RD_LAUNCH("flats.seed", k_flat_seed, dim3((nsrc + NTHR - 1) / NTHR), dim3(NTHR), 0, s, src, nsrc, L, fh_filter, D,
            r.tflags, w, r.tilesX, r.tilesY, (const int32_t *)nullptr);
  return relax_rounds(d_dirs, D, r.tflags, r.tlist, r.ctr, w, h, 0, h, "flats.relax_away", s);
}

// Towards levels from the low edges (flags: F_LOW level 1, F_NEAR level 2).  D is written in full.
static uint32_t run_relax_towards(const uint8_t *d_dirs, const uint8_t *flags, int32_t *D, int w, int h, hipStream_t s) {
  const RelaxScratch r = relax_scratch(w, h);
  RD_HIP(hipMemsetAsync(r.tflags, 0, r.ntiles, s));
  RD_LAUNCH("flats.init_towards", k_flat_init_towards, dim3(r.ntiles), dim3(NTHR), 0, s, flags, D, r.tflags, w, h, 0, h, r.tilesX,
            r.tilesY);
  return relax_rounds(d_dirs, D, r.tflags, r.tlist, r.ctr, w, h, 0, h, "flats.relax_towards", s);
}

// ---- the bitmap engine's rounds (same protocol as relax_rounds: batches of rounds, counts read back per batch) ----
// second = true: the tile flags, lists and counters of a search that runs BESIDE another one (the bitmaps are shared)
static BitsScratch bits_scratch(int w, int h, bool second = false, bool planes = false) {
  Workspace &ws = Workspace::get();
  BitsScratch b;
  b.tilesX = (w + BT - 1) / BT; b.tilesY = (h + BT - 1) / BT; b.ntiles = b.tilesX * b.tilesY;
  b.mbits = ws.buf<unsigned long long>("flats.mbits", (size_t)b.ntiles * BT * (planes ? 3 : 1));   // (planes: M | near | high)
  b.expanded = ws.buf<uint8_t>(second ? "flats.bexp2" : "flats.bexp", b.ntiles);
  b.tflags = ws.buf<uint8_t>(second ? "flats.btflags2" : "flats.btflags", b.ntiles);
  b.tlist = ws.buf<uint32_t>(second ? "flats.btlist2" : "flats.btlist", b.ntiles);
  b.ctr = ws.buf<uint32_t>(second ? "flats.tctr2" : "flats.tctr", BITS_BATCH);
  b.counts = ws.buf<uint32_t>("flats.bcounts", 3 * 256 + 8);
  if (planes) {
    b.pf.P = ws.buf<unsigned long long>(second ? "flats.planes2" : "flats.planes", (size_t)b.ntiles * NPL * BT);
    b.pf.E = ws.buf<int32_t>(second ? "flats.pedges2" : "flats.pedges", (size_t)b.ntiles * 256);
    b.pf.R = ws.buf<unsigned long long>(second ? "flats.preached2" : "flats.preached", (size_t)b.ntiles * BT);
    b.pf.overflow = ws.buf<uint32_t>("flats.poverflow", 4) + (second ? 1 : 0);
    b.pf.expanded = b.expanded;
    if (const char *e = getenv("RDGPU_FLAT_PLANES_MAX")) b.pf.max_level = std::min(0xFFF0, std::max(8, atoi(e)));   // (tests: the overflow path)
    b.near = b.mbits + (size_t)b.ntiles * BT;
    b.high = b.mbits + 2 * (size_t)b.ntiles * BT;
  }
  return b;
}

// RDGPU_FLAT_ASYNC = n: the rounds end and the asynchronous tail (k_relax_bits_async) takes over once a round visits
// fewer than n tiles (default 20000; 0: rounds to the end).
static uint32_t async_threshold() {
  const char *env = getenv("RDGPU_FLAT_ASYNC");
  return env ? (uint32_t)strtoul(env, nullptr, 10) : 20000u;
}

struct AsyncInfo { uint64_t visits; uint32_t launches, failures, live_tiles; };   // (of the last flat resolution: rdgpu_flat_get_async_stats)
static thread_local AsyncInfo g_async_info = {0, 0, 0, 0};

// What relax_rounds_bits / the static search hand to the code around them when the search reaches its tail: mark() before
// the resident launch is enqueued (the place for an event the side work waits on), go() after it (the resident
// wavefronts have their places: now the work that is to run beside them).  Both are called exactly once, also when the
// search ends in its rounds.
struct Beside {
  std::function<void()> mark, go;
};

struct AsyncRun {
  AsyncQ Q;
  const uint32_t *live = nullptr;   // the length of the tile list the queues were filled from
  uint32_t blocks = 0;
  int cus = 0, rate_khz = 0;
};

// compaction of the flags the last round left + the queues' first content + the resident launch, all on s, no host step
template <int SEED_LEVEL>
static AsyncRun async_enqueue(const BitsScratch &b, int32_t *D, int w, int h, const char *name, hipStream_t s, RowWin win, bool second,
                              const Beside *beside) {
  Workspace &ws = Workspace::get();
  uint32_t per_q = 64;   // slots per queue: every tile of a queue at once
  while (per_q < (b.ntiles + AQ_NQ - 1) / AQ_NQ) per_q <<= 1;
  AsyncRun r;
  AsyncQ &Q = r.Q;
  Q.q = ws.buf<uint32_t>(second ? "flats.aq2" : "flats.aq", (size_t)per_q * AQ_NQ);
  Q.state = ws.buf<uint32_t>(second ? "flats.aqstate2" : "flats.aqstate", b.ntiles);
  Q.ctl = ws.buf<uint32_t>(second ? "flats.aqctl2" : "flats.aqctl", AQ_WORDS);
  Q.qmask = per_q - 1;
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  RD_HIP(hipDeviceGetAttribute(&r.cus, hipDeviceAttributeMultiprocessorCount, dev));
  RD_HIP(hipDeviceGetAttribute(&r.rate_khz, hipDeviceAttributeWallClockRate, dev));
  // twenty seconds of wall_clock64 ticks (an attribute that reads 0 is taken as the usual 100 MHz)
  const unsigned long long budget = (unsigned long long)std::max(r.rate_khz, 100000) * 1000ull * 20ull;
  uint32_t *last = b.ctr + (BITS_BATCH - 1);   // (its own counter word: the rounds' words may not have been read back yet)
  r.live = last;
  RD_HIP(hipMemsetAsync(last, 0, sizeof(uint32_t), s));
  RD_LAUNCH("flats.tiles_compact", k_tiles_compact, dim3((b.ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, b.tflags, b.ntiles, b.tlist,
            last);
  RD_HIP(hipMemsetAsync(Q.q, 0xFF, (size_t)per_q * AQ_NQ * sizeof(uint32_t), s));
  RD_HIP(hipMemsetAsync(Q.state, 0, (size_t)b.ntiles * sizeof(uint32_t), s));
  RD_HIP(hipMemsetAsync(Q.ctl, 0, AQ_WORDS * sizeof(uint32_t), s));
  RD_LAUNCH("flats.async_init", k_async_init, dim3((b.ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, (const uint32_t *)b.tlist,
            (const uint32_t *)last, Q);
  // Resident blocks: alone, two per CU were the optimum (S3: 1 / 2 / 3 / 4 per CU 45.5 / 40.6 / 40.9 / 42.7 ms: idle
  // wavefronts poll); with the away search or the labels running BESIDE the tail, fewer leave them room:
  // 256 / 320 / 384 / 448 / 512 / 640 blocks 33.8 / 33.9 / 34.4 / 35.5 / 36.6 / 38.7 ms for the stage, 53.8 ms (320) against
  // 56.4 (512) for ResolveFlatsEpsilon.
  r.blocks = (uint32_t)r.cus * 5u / 4u;
  int nap = 1;
  if (const char *e = getenv("RDGPU_FLAT_ASYNC_BLOCKS")) r.blocks = std::max(1, atoi(e));
  if (const char *e = getenv("RDGPU_FLAT_ASYNC_NAP")) nap = std::min(255, std::max(1, atoi(e)));
  if (beside && beside->mark) beside->mark();
  if (b.pf.P)
    RD_LAUNCH(name, (k_relax_bits_async<SEED_LEVEL, true>), dim3(r.blocks), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits, D, Q,
              w, h, win, b.tilesX, b.tilesY, budget, nap, b.pf);
  else
    RD_LAUNCH(name, (k_relax_bits_async<SEED_LEVEL>), dim3(r.blocks), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits, D, Q, w,
              h, win, b.tilesX, b.tilesY, budget, nap, PlaneField{});
  if (beside && beside->go) beside->go();
  return r;
}

// the end state, checked on the host (s is synchronised here): no abort, every counter pair equal, every queue drained
// Returns false (and says so on stderr) when the launch gave up: the levels are valid upper bounds then, and the caller
// finishes the search in rounds from "every tile active" (the fixed point does not depend on the schedule).
static bool async_check(const AsyncRun &r, const char *name, hipStream_t s) {
  std::vector<uint32_t> all(AQ_WORDS);
  uint32_t live = 0;
  RD_HIP(hipMemcpyAsync(all.data(), r.Q.ctl, AQ_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (r.live) RD_HIP(hipMemcpyAsync(&live, r.live, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  uint64_t enq = 0, done = 0, queued = 0, pushes = 0;
  for (int qi = 0; qi < AQ_NQ; qi++) {
    enq += all[qi * AQ_STRIDE + 32]; done += all[qi * AQ_STRIDE + 33];
    queued += all[qi * AQ_STRIDE + 1] - all[qi * AQ_STRIDE + 0];
    pushes += all[qi * AQ_STRIDE + 1];
  }
  if (all[AQ_G_ABORT] != 0 || enq != done || queued != 0) {
    fprintf(stderr, "rdgpu flat resolution: the asynchronous search (%s) gave up (abort %u, %llu tiles pending, %llu queued); "
                    "finishing in rounds\n", name, all[AQ_G_ABORT], (unsigned long long)(enq - done), (unsigned long long)queued);
    g_async_info.failures++;
    return false;
  }
  g_async_info.visits += all[AQ_G_VISITS];
  g_async_info.live_tiles += live;
  g_async_info.launches++;
  if (getenv("RDGPU_FLAT_TRACE") || getenv("RDGPU_FLAT_ASYNC_STATS"))
    fprintf(stderr, "%s asynchronous tail: %u visits, %llu pushes, %u wavefronts on %d CUs, ticks (%d kHz): in visits %llu, longest wavefront %u\n",
            name, all[AQ_G_VISITS], (unsigned long long)pushes, r.blocks * 4u, r.cus, r.rate_khz,
            (unsigned long long)all[AQ_G_BUSY] | ((unsigned long long)all[AQ_G_BUSY + 1] << 32), all[AQ_G_SPAN]);
  return true;
}

template <int SEED_LEVEL>
static uint32_t relax_rounds_bits(const BitsScratch &b, int32_t *D, int w, int h, const char *name, hipStream_t s,
                                  RowWin win = RowWin{0, -1, nullptr, nullptr}, const Beside *beside = nullptr,
                                  bool allow_async = true) {
  if (win.hi < 0) win.hi = h;   // single device: all rows, no ghost rows
  uint32_t *hw = Workspace::get().host_words();
  const bool trace = getenv("RDGPU_FLAT_TRACE") != nullptr;
  uint32_t async_below = allow_async ? async_threshold() : 0u;
  if (allow_async && getenv("RDGPU_FLAT_ASYNC_FAIL")) async_below = std::max(async_below, 1u);   // (tests: the recovery path below)
  uint32_t rounds = 0, grid = (b.ntiles + 3) / 4;   // any tile may be active in the first batch
  for (;;) {
    // with the asynchronous tail ahead the batches are short (the switch is decided on the host, from the counts)
    const int batch = async_below ? (rounds == 0 ? 6 : 3) : BITS_BATCH;
    RD_HIP(hipMemsetAsync(b.ctr, 0, BITS_BATCH * sizeof(uint32_t), s));
    for (int k = 0; k < batch; k++) {
      RD_LAUNCH("flats.tiles_compact", k_tiles_compact, dim3((b.ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, b.tflags, b.ntiles,
                b.tlist, b.ctr + k);
      if (b.pf.P)
        RD_LAUNCH(name, (k_relax_planes<SEED_LEVEL>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits, b.expanded, b.pf,
                  (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, b.tilesX, b.tilesY);
      else if (trace)
        RD_LAUNCH(name, (k_relax_bits<SEED_LEVEL, true>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits,
                  b.expanded, D, (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, win,
                  b.tilesX, b.tilesY);
      else
        RD_LAUNCH(name, (k_relax_bits<SEED_LEVEL>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits,
                  b.expanded, D, (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, win,
                  b.tilesX, b.tilesY);
    }
    RD_HIP(hipMemcpyAsync(hw, b.ctr, batch * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    uint32_t most = 0;
    for (int k = 0; k < batch; k++) {
      if (trace) fprintf(stderr, "%s round %u nact %u (grid %u)\n", name, rounds, hw[k], grid);
      if (hw[k] == 0 && trace) {
        unsigned long long st[16];
        RD_HIP(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_relax_stats), sizeof st));
        fprintf(stderr, "%s stats (cumulative): visits %llu, left at once %llu, open-water %llu, general %llu, levels stepped %llu, "
                        "flushes %llu, rows flushed %llu\n", name, st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
        fprintf(stderr, "%s tail launches (< 8000 tiles), working visits %llu, 10 ns ticks: setup %llu, open-water %llu, rows read %llu, "
                        "level loop + flushes %llu, flushes %llu, wake %llu, whole %llu, longest %llu\n", name, st[7], st[8], st[9],
                st[10], st[11], st[12], st[13], st[14], st[15]);
      }
      if (hw[k] == 0) {
        if (beside && beside->mark) beside->mark();
        if (beside && beside->go) beside->go();
        return rounds;
      }
      most = std::max(most, hw[k]);
      rounds++;
    }
    if (async_below && hw[batch - 1] < async_below) {
      const AsyncRun run = async_enqueue<SEED_LEVEL>(b, D, w, h, name, s, win, false, beside);
      beside = nullptr;   // (its callbacks have run)
      if (async_check(run, name, s) && !getenv("RDGPU_FLAT_ASYNC_FAIL")) return rounds + 1;
      // the tail gave up (or the test switch says so): rounds to the end, from every tile
      RD_HIP(hipMemsetAsync(b.tflags, 1, b.ntiles, s));
      async_below = 0;
      grid = (b.ntiles + 3) / 4;
      continue;
    }
    grid = std::min<uint32_t>((b.ntiles + 3) / 4, std::max<uint32_t>(256u, (2u * most + 3) / 4));
    if (rounds > (1u << 26)) throw Error(RDGPU_ERR_HIP, "rdgpu flat resolution: relaxation did not terminate");
  }
}

// Towards levels from the low edges, D written in full; write_m: also the bitmap of the cells that take part (shared
// with the away field); counts3 (optional, host): low edges, high edges, NO_FLOW cells.
static uint32_t run_bits_towards(const uint8_t *flags, int32_t *D, bool write_m, unsigned long long *counts3, int w, int h,
                                 hipStream_t s, const Beside *beside = nullptr, bool planes = false) {
  const BitsScratch b = bits_scratch(w, h, false, planes);
  RD_HIP(hipMemsetAsync(b.tflags, 0, b.ntiles, s));
  RD_HIP(hipMemsetAsync(b.expanded, 0, b.ntiles, s));
  uint32_t *cnt = counts3 ? b.counts : nullptr;
  if (cnt && flags) RD_HIP(hipMemsetAsync(cnt, 0, (3 * 256 + 8) * sizeof(uint32_t), s));
  if (planes && !flags) {   // the classification made the bitmaps (mbits, near) and the counts: k_dirs_classify<BITMAPS>
    RD_HIP(hipMemsetAsync(b.pf.overflow, 0, 2 * sizeof(uint32_t), s));
    // (it also counts the NO_FLOW cells and the high edges from their bitmaps: the away search may start later, the counts
    // are read with this search's)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare_b<true>), dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s,
              (const unsigned long long *)b.near, b.pf, b.tflags, h, b.tilesX, b.tilesY, (const unsigned long long *)b.mbits, cnt, 2);
    if (cnt) RD_LAUNCH("flats.bits_counts_high", k_rows_count, dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s,
                       (const unsigned long long *)b.high, b.ntiles, cnt, 1);
  } else if (planes) {
    RD_HIP(hipMemsetAsync(b.pf.overflow, 0, 2 * sizeof(uint32_t), s));   // (both fields' words)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare<true, true>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, b.pf, b.mbits, b.near,
              b.tflags, cnt, w, h, b.tilesX, b.tilesY);
  } else if (write_m)
    RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<true, true>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, (const uint32_t *)nullptr,
              (const int32_t *)nullptr, D, b.mbits, b.tflags, cnt, w, RowWin{0, h, nullptr, nullptr}, (const int32_t *)nullptr,
              b.tilesX, b.tilesY);
  else
    RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<true, false>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, (const uint32_t *)nullptr,
              (const int32_t *)nullptr, D, b.mbits, b.tflags, cnt, w, RowWin{0, h, nullptr, nullptr}, (const int32_t *)nullptr,
              b.tilesX, b.tilesY);
  if (cnt) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(cnt + 3 * 256 + 2);
    RD_LAUNCH("flats.bits_counts", k_bits_counts, dim3(1), dim3(NTHR), 0, s, (const uint32_t *)cnt, out);
    RD_HIP(hipMemcpyAsync(counts3, out, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));   // (read after the rounds' sync)
  }
  return relax_rounds_bits<2>(b, D, w, h, "flats.relax_towards", s, RowWin{0, -1, nullptr, nullptr}, beside);
}

// Away levels from the high edges (with L / fh: only those of flats that have an outlet), D written in full.
static uint32_t run_bits_away(const uint8_t *flags, const uint32_t *L, const int32_t *fh, int32_t *D, bool write_m, int w, int h,
                              hipStream_t s, bool planes = false) {
  const BitsScratch b = bits_scratch(w, h, planes, planes);   // (planes: the field's own planes, beside the towards field's)
  RD_HIP(hipMemsetAsync(b.tflags, 0, b.ntiles, s));
  RD_HIP(hipMemsetAsync(b.expanded, 0, b.ntiles, s));
  if (planes && !flags)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare_b<false>), dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s,
              (const unsigned long long *)b.high, b.pf, b.tflags, h,
              b.tilesX, b.tilesY);
  else if (planes)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare<false, false>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, b.pf, b.mbits,
              (unsigned long long *)nullptr, b.tflags, (uint32_t *)nullptr, w, h, b.tilesX, b.tilesY);
  else if (write_m)
    RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<false, true>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, L, fh, D, b.mbits,
              b.tflags, (uint32_t *)nullptr, w, RowWin{0, h, nullptr, nullptr}, (const int32_t *)nullptr, b.tilesX, b.tilesY);
  else
    RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<false, false>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, L, fh, D, b.mbits,
              b.tflags, (uint32_t *)nullptr, w, RowWin{0, h, nullptr, nullptr}, (const int32_t *)nullptr, b.tilesX, b.tilesY);
  return relax_rounds_bits<1>(b, D, w, h, "flats.relax_away", s);
}

// The away search as ONE enqueue, without a host decision: start levels, AWAY_STATIC_ROUNDS rounds over a full grid, the
// asynchronous tail for whatever is left (any number of rounds before the tail gives the same levels).  It runs on a side
// stream beside the TAIL of the towards search: that tail keeps two resident blocks per CU on dependent visits, the
// away search's rounds are throughput-bound.  RDGPU_FLAT_AWAY_BESIDE=0: after the towards search, on its stream.
constexpr int AWAY_STATIC_ROUNDS = 8;
struct StaticAway {
  BitsScratch b;
  AsyncRun run;
};
static bool away_beside() {
  const char *env = getenv("RDGPU_FLAT_AWAY_BESIDE");
  return !(env && env[0] == '0') && async_threshold() > 0 && getenv("RDGPU_FLAT_TRACE") == nullptr;
}
// r06 (plane engine): the TOWARDS search as one enqueue as well -- start levels (with the edge counts), TOWARDS_STATIC_ROUNDS
// rounds over grids that follow the front's usual shrink (S3: 0.82, 0.44, 0.13, 0.05, 0.03, 0.02 of the tiles), the
// asynchronous tail -- so that a directions-only flat resolution reads nothing back until everything is enqueued: one
// synchronisation per call instead of one per batch of rounds, per tail and per check (on a busy host each is a scheduling
// quantum: profiles/README.md r05u).
constexpr int TOWARDS_STATIC_ROUNDS = 6;
static StaticAway enqueue_towards_static(const uint8_t *flags, unsigned long long *d_counts3, int w, int h, hipStream_t s,
                                         const Beside *beside) {
  static_assert(TOWARDS_STATIC_ROUNDS < BITS_BATCH - 1, "the tail's own counter word is the last one");
  StaticAway st;
  st.b = bits_scratch(w, h, false, true);
  const BitsScratch &b = st.b;
  const RowWin win{0, h, nullptr, nullptr};
  RD_HIP(hipMemsetAsync(b.tflags, 0, b.ntiles, s));
  RD_HIP(hipMemsetAsync(b.expanded, 0, b.ntiles, s));
  RD_HIP(hipMemsetAsync(b.pf.overflow, 0, 2 * sizeof(uint32_t), s));   // (both fields' words)
  if (!flags) {   // the classification made the bitmaps and counted the low edges
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare_b<true>), dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s,
              (const unsigned long long *)b.near, b.pf, b.tflags, h, b.tilesX, b.tilesY, (const unsigned long long *)b.mbits, b.counts, 2);
    RD_LAUNCH("flats.bits_counts_high", k_rows_count, dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s, (const unsigned long long *)b.high,
              b.ntiles, b.counts, 1);
  } else {
    RD_HIP(hipMemsetAsync(b.counts, 0, (3 * 256 + 8) * sizeof(uint32_t), s));
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare<true, true>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, b.pf, b.mbits, b.near,
              b.tflags, b.counts, w, h, b.tilesX, b.tilesY);
  }
  RD_LAUNCH("flats.bits_counts", k_bits_counts, dim3(1), dim3(NTHR), 0, s, (const uint32_t *)b.counts, d_counts3);
  RD_HIP(hipMemsetAsync(b.ctr, 0, BITS_BATCH * sizeof(uint32_t), s));
  for (int k = 0; k < TOWARDS_STATIC_ROUNDS; k++) {
    RD_LAUNCH("flats.tiles_compact", k_tiles_compact, dim3((b.ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, b.tflags, b.ntiles, b.tlist,
              b.ctr + k);
    const uint32_t full = (b.ntiles + 3) / 4, grid = k < 2 ? full : std::max<uint32_t>(256u, full >> (k - 1));
    RD_LAUNCH("flats.relax_towards", (k_relax_planes<2>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits, b.expanded,
              b.pf, (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, b.tilesX, b.tilesY);
  }
  st.run = async_enqueue<2>(b, nullptr, w, h, "flats.relax_towards", s, win, false, beside);   // (mark() before the tail's launch, go() after)
  return st;
}

static StaticAway enqueue_away_static(const uint8_t *flags, int32_t *A, int w, int h, hipStream_t s, bool planes = false) {
  static_assert(AWAY_STATIC_ROUNDS < BITS_BATCH - 1, "the tail's own counter word is the last one");
  StaticAway sa;
  sa.b = bits_scratch(w, h, true, planes);
  const BitsScratch &b = sa.b;
  const RowWin win{0, h, nullptr, nullptr};
  RD_HIP(hipMemsetAsync(b.tflags, 0, b.ntiles, s));
  RD_HIP(hipMemsetAsync(b.expanded, 0, b.ntiles, s));
  if (planes && !flags)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare_b<false>), dim3((b.ntiles + 3) / 4), dim3(NTHR), 0, s,
              (const unsigned long long *)b.high, b.pf, b.tflags, h,
              b.tilesX, b.tilesY);
  else if (planes)
    RD_LAUNCH("flats.bits_prepare", (k_planes_prepare<false, false>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, b.pf, b.mbits,
              (unsigned long long *)nullptr, b.tflags, (uint32_t *)nullptr, w, h, b.tilesX, b.tilesY);
  else
  RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<false, false>), dim3(b.ntiles), dim3(NTHR), 0, s, flags, (const uint32_t *)nullptr,
            (const int32_t *)nullptr, A, b.mbits, b.tflags, (uint32_t *)nullptr, w, win, (const int32_t *)nullptr, b.tilesX, b.tilesY);
  RD_HIP(hipMemsetAsync(b.ctr, 0, BITS_BATCH * sizeof(uint32_t), s));
  for (int k = 0; k < AWAY_STATIC_ROUNDS; k++) {
    RD_LAUNCH("flats.tiles_compact", k_tiles_compact, dim3((b.ntiles + NTHR - 1) / NTHR), dim3(NTHR), 0, s, b.tflags, b.ntiles, b.tlist,
              b.ctr + k);
    // (the front of this search shrinks fast -- S3: 0.86, 0.78, 0.37, 0.17, 0.09, 0.04, 0.02, 0.01 of the tiles -- and a grid
    // that is too small for a round is not an error: the tiles past it stay active for the next one)
    const uint32_t full = (b.ntiles + 3) / 4, grid = k < 2 ? full : std::max<uint32_t>(256u, full >> (k - 1));
    if (planes)
      RD_LAUNCH("flats.relax_away", (k_relax_planes<1>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits, b.expanded,
                b.pf, (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, b.tilesX, b.tilesY);
    else
    RD_LAUNCH("flats.relax_away", (k_relax_bits<1>), dim3(grid), dim3(NTHR), 0, s, (const unsigned long long *)b.mbits,
              b.expanded, A, (const uint32_t *)b.tlist, (const uint32_t *)(b.ctr + k), b.tflags, w, h, win, b.tilesX, b.tilesY);
  }
  sa.run = async_enqueue<1>(b, A, w, h, "flats.relax_away", s, win, true, nullptr);
  return sa;
}
// on a stream that has waited for the side stream: the end state of the tail, and the number of rounds that had work
// (the towards field of the static plane search: the same end-of-search check; *recovered: the tail had given up)
static uint32_t finish_towards_static(const StaticAway &st, int w, int h, hipStream_t s, bool *recovered) {
  uint32_t counts[TOWARDS_STATIC_ROUNDS];
  RD_HIP(hipMemcpyAsync(counts, st.b.ctr, sizeof counts, hipMemcpyDeviceToHost, s));
  const bool ok = async_check(st.run, "flats.relax_towards", s);   // (synchronises s)
  uint32_t rounds = 1;
  for (int k = 0; k < TOWARDS_STATIC_ROUNDS; k++) rounds += counts[k] != 0;
  *recovered = false;
  if (!ok || getenv("RDGPU_FLAT_ASYNC_FAIL")) {
    RD_HIP(hipMemsetAsync(st.b.tflags, 1, st.b.ntiles, s));
    rounds += relax_rounds_bits<2>(st.b, nullptr, w, h, "flats.relax_towards", s, RowWin{0, -1, nullptr, nullptr}, nullptr, false);
    *recovered = true;
  }
  return rounds;
}
static uint32_t finish_away_static(const StaticAway &sa, int32_t *A, int w, int h, hipStream_t s, bool *recovered = nullptr) {
  uint32_t counts[AWAY_STATIC_ROUNDS];
  RD_HIP(hipMemcpyAsync(counts, sa.b.ctr, sizeof counts, hipMemcpyDeviceToHost, s));
  const bool ok = async_check(sa.run, "flats.relax_away", s);   // (synchronises s)
  uint32_t rounds = 1;
  for (int k = 0; k < AWAY_STATIC_ROUNDS; k++) rounds += counts[k] != 0;
  if (recovered) *recovered = false;
  if (!ok || getenv("RDGPU_FLAT_ASYNC_FAIL")) {   // the tail gave up: rounds to the end, from every tile, on this stream
    RD_HIP(hipMemsetAsync(sa.b.tflags, 1, sa.b.ntiles, s));
    rounds += relax_rounds_bits<1>(sa.b, A, w, h, "flats.relax_away", s, RowWin{0, -1, nullptr, nullptr}, nullptr, false);
    if (recovered) *recovered = true;
  }
  return rounds;
}

static bool lean_labels() {
  const char *env = getenv("RDGPU_RFE_LEAN");   // =0: labels, outlet marks and flat heights as the flat_mask path makes them: A/B and tests
  return !(env && env[0] == '0');
}

static bool use_bits_engine() {
  const char *env = getenv("RDGPU_FLAT_BITS");   // =0: the stencil relaxation: A/B and tests
  return !(env && env[0] == '0');
}

// Computes flat_mask (M) for the DEM; d_dirs must hold d8_flow_directions output.
// Returns device pointers (workspace) to M, L, fh through the out parameters.
template <class T>
static void resolve_flats_device(const T *d_z, const uint8_t *d_dirs, int w, int h, int32_t **outM, uint32_t **outL,
                                 int32_t **outFh, hipStream_t s, const int32_t **outA = nullptr, T ff_nodata = T()) {
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  int32_t *M = ws.buf<int32_t>("flats.mask", n);
  *outM = M;
  *outL = nullptr;
  *outFh = nullptr;
  const bool lean = outA && use_bits_engine() && lean_labels();
  if (!lean) RD_HIP(hipMemsetAsync(M, 0, n * sizeof(int32_t), s));  // flat_mask.setAll(0), :469 (lean: the towards search writes every cell)
  g_fstats = rdgpu_flat_stats{0, 0, 0, 0, 0};
  g_async_info = AsyncInfo{0, 0, 0, 0};

  uint8_t *flags = ws.buf<uint8_t>("flats.flags", n);
  if (d_dirs) {
    launch_classify<T>(d_z, d_dirs, w, h, flags, s);
  } else {   // (ResolveFlatsEpsilon's lean path: FindFlats and the classification in one pass; nodata travels in *outA's place)
    if (!lean) throw Error(RDGPU_ERR_ARG, "resolve_flats_device: the fused FindFlats classification needs the lean path");
    const uint32_t tilesX = (w + SW - 1) / SW, ntiles = tilesX * ((h + KLH - 1) / KLH);
    RD_LAUNCH("flats.findflats_classify", (k_dirs_classify<T, false, true>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, ff_nodata,
              (uint8_t *)nullptr, flags, w, h, tilesX, ntiles);
  }
  uint32_t *low = nullptr, *highall = nullptr;
  uint32_t nlow = 0, nhigh_all = 0, nnoflow = 0;
  if (!lean) {
    compact_edges(flags, n, &low, &nlow, &highall, &nhigh_all, &nnoflow, s, !use_bits_engine());
    g_fstats.low_edges = nlow;
    g_fstats.noflow_cells = nnoflow;
    g_fstats.high_edges = nhigh_all;
    if (nlow == 0) return;   // no flats, or none with an outlet (:475-481)
  }

  uint32_t *L = ws.buf<uint32_t>("flats.L", n);
  int32_t *fh = ws.buf<int32_t>("flats.fh", n);
  if (!lean) {
    *outL = L;
    *outFh = fh;
  }
  if (lean) {
    // The caller wants the levels, the labels and the flat heights (ResolveFlatsEpsilon), nothing per flat beyond that: the
    // labels do not depend on the searches here.  "The flat has an outlet" (:491-500) is what the towards levels say anyway
    // (a flat without a low edge is never reached: its cells keep DINF and k_flat_epsilon4 skips them), so the away field
    // starts from EVERY high edge and fh needs no "-1 = no outlet" marks: no k_flat_mark_low pass, no fill of fh.
    const uint32_t tilesX = (w + CW - 1) / CW, ntiles = tilesX * ((h + CH - 1) / CH);
    T *colZ = ws.buf<T>("flats.colz", (size_t)ntiles * 2 * CH);
    uint32_t *colL = ws.buf<uint32_t>("flats.coll", (size_t)ntiles * 2 * CH);
    // ... and so they are made BESIDE the tail of the towards search (and the away search after it), on the device's side
    // stream: the tail keeps two resident blocks per CU busy with dependent visits and leaves the memory system idle, the
    // labelling is three streaming passes.  (Started with the search's first rounds, which are throughput-bound
    // themselves, the labels only took their turn: 64.4 -> 63.0 ms at S3.)  RDGPU_RFE_OVERLAP=0: one stream.
    const char *env = getenv("RDGPU_RFE_OVERLAP");
    const bool beside = !(env && env[0] == '0');
    hipStream_t ls = s;
    Workspace::SideLane *lane = nullptr;   // (lane 0: the labels; lane 1: the away search)
    if (beside) {
      lane = &ws.side_lane(0);
      ls = lane->stream;
      RD_HIP(hipEventRecord(lane->fork, s));   // (the DEM and the workspace are as the caller's stream left them)
    }
    const auto labels = [&]() {
      if (beside) RD_HIP(hipStreamWaitEvent(ls, lane->fork, 0));
      RD_LAUNCH("flats.ccl_tile", (k_ccl_tile<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, ls, d_z, L, w, h, tilesX, ntiles, colZ, colL);
      const uint64_t nthreads = (uint64_t)ntiles * (CW + CH);
      RD_LAUNCH("flats.ccl_border", (k_ccl_border2<T>), dim3((uint32_t)((nthreads + NTHR - 1) / NTHR)), dim3(NTHR), 0, ls, d_z, L,
                (const T *)colZ, (const uint32_t *)colL, w, h, tilesX, ntiles);
      RD_LAUNCH("flats.ccl_flatten", k_ccl_flatten4, dim3(sgrid(n / 4 + 3)), dim3(NTHR), 0, ls, L, fh, n);
      if (beside) RD_HIP(hipEventRecord(lane->join, ls));
    };
    if (!beside) labels();
    // ... the away search too, on a second side stream (see enqueue_away_static), only on request: with the labels already
    // beside the tail a third stream gains nothing (S3: 58.7 ms without, 59.5 with; RDGPU_RFE_AWAY_BESIDE=1)
    const char *env3 = getenv("RDGPU_RFE_AWAY_BESIDE");
    // (the edge counts come with the towards search's start levels, as on the directions path: no counting pass)
    unsigned long long c3[3] = {0, 0, 0};
    const bool away_too = beside && env3 && env3[0] == '1' && away_beside();
    Workspace::SideLane *alane = away_too ? &ws.side_lane(1) : nullptr;
    StaticAway sa;
    int32_t *A = nullptr;
    Beside bs;
    bs.mark = [&]() {
      if (away_too) RD_HIP(hipEventRecord(alane->fork, s));
    };
    bool away_started = false;
    bs.go = [&]() {
      if (beside) labels();
      if (away_too && c3[0] > 0 && c3[1] > 0) {
        away_started = true;
        RD_HIP(hipStreamWaitEvent(alane->stream, alane->fork, 0));
        A = ws.buf<int32_t>("flats.away", n);
        sa = enqueue_away_static(flags, A, w, h, alane->stream);
        RD_HIP(hipEventRecord(alane->join, alane->stream));
      }
    };
    g_fstats.towards_levels = run_bits_towards(flags, M, true, c3, w, h, s, &bs);
    g_fstats.low_edges = c3[0];
    g_fstats.high_edges = c3[1];
    g_fstats.noflow_cells = c3[2];
    if (away_started) {
      RD_HIP(hipStreamWaitEvent(s, alane->join, 0));
      g_fstats.away_levels = finish_away_static(sa, A, w, h, s);
    } else if (c3[0] > 0 && c3[1] > 0) {
      A = ws.buf<int32_t>("flats.away", n);
      g_fstats.away_levels = run_bits_away(flags, nullptr, nullptr, A, false, w, h, s);
    }
    if (beside) RD_HIP(hipStreamWaitEvent(s, lane->join, 0));
    if (c3[0] == 0) return;   // no flats, or none with an outlet (:475-481): *outL stays null, nothing is altered
    *outL = L;
    *outFh = fh;
    if (A) RD_LAUNCH("flats.height", k_flat_height4, dim3(sgrid(n / 4 + 3)), dim3(NTHR), 0, s, (const int32_t *)A, (const uint32_t *)L, fh, n);
    *outA = A;
    return;
  }
  {
    const uint32_t tilesX = (w + CW - 1) / CW, ntiles = tilesX * ((h + CH - 1) / CH);
    RD_LAUNCH("flats.ccl_tile", (k_ccl_tile<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, L, w, h, tilesX, ntiles, (T *)nullptr,
              (uint32_t *)nullptr);
  }
  launch_ccl_border<T>(d_z, L, w, h, s);
  RD_LAUNCH("flats.ccl_flatten", k_ccl_flatten, dim3(sgrid(n)), dim3(NTHR), 0, s, L, n);
  RD_HIP(hipMemsetAsync(fh, 0xFF, n * sizeof(int32_t), s));        // -1 everywhere
  if (low)
    RD_LAUNCH("flats.mark_low", k_flat_mark_low, dim3((nlow + NTHR - 1) / NTHR), dim3(NTHR), 0, s, (const uint32_t *)low,
              nlow, (const uint32_t *)L, fh);
  else
    RD_LAUNCH("flats.mark_low", k_flat_mark_low_flags, dim3(sgrid(n)), dim3(NTHR), 0, s, (const uint8_t *)flags, (const uint32_t *)L,
              fh, n);

  // away gradient: sources = high edges whose flat has a low edge
  int32_t *A = nullptr;
  if (nhigh_all > 0) {
    A = ws.buf<int32_t>("flats.away", n);
    if (use_bits_engine()) {
      g_fstats.away_levels = run_bits_away(flags, L, fh, A, true, w, h, s);
    } else {
      RD_HIP(hipMemsetAsync(A, 0x7F, n * sizeof(int32_t), s));
      g_fstats.away_levels = run_relax_away(d_dirs, A, highall, nhigh_all, L, fh, w, h, s);
    }
    RD_LAUNCH("flats.height", k_flat_height, dim3(sgrid(n)), dim3(NTHR), 0, s, (const int32_t *)A, (const uint32_t *)L, fh,
              n);
  }
  // towards gradient from every low edge, then the combined mask in place
  g_fstats.towards_levels = use_bits_engine() ? run_bits_towards(flags, M, nhigh_all == 0, nullptr, w, h, s)
                                              : run_relax_towards(d_dirs, flags, M, w, h, s);
  if (outA) {   // the caller combines on the fly (k_flat_epsilon): M holds the towards levels
    *outA = A;
    return;
  }
  RD_LAUNCH("flats.combine", k_flat_combine, dim3(sgrid(n)), dim3(NTHR), 0, s, M, (const int32_t *)A, (const uint32_t *)L,
            (const int32_t *)fh, n);
}

static thread_local bool g_planes_overflowed = false;   // (flat_resolution_device: this call repeats a plane search that overflowed)
template <class T>
void flat_resolution_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, hipStream_t s) {
  if (!d_z || !d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8: raster too large");
  const char *env = getenv("RDGPU_FLAT_FULLMASK");
  const char *envf = getenv("RDGPU_FLAT_FUSED_CLASSIFY");   // =0: d8_flow_directions and the classification as two kernels: A/B and tests
  const bool fused_classify = use_bits_engine() && !(env && env[0] == '1') && !(envf && envf[0] == '0');
  if (!fused_classify) flowdirs_device<T>(d_z, nodata, w, h, d_dirs, MODE_D8, s);   // =1: through the full flat_mask (labels, flat heights): A/B and tests
  if (env && env[0] == '1') {
    int32_t *M, *fh;
    uint32_t *L;
    resolve_flats_device<T>(d_z, d_dirs, w, h, &M, &L, &fh, s);
    if (L)   // there is at least one low edge
      launch_masked_dirs<T>(d_z, M, d_dirs, w, h, s);
    return;
  }
  // directions only: the two level fields suffice (k_flat_dirs_levels)
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  g_fstats = rdgpu_flat_stats{0, 0, 0, 0, 0};
  g_async_info = AsyncInfo{0, 0, 0, 0};
  // r05: the last pass from 4 bits per cell, without the DEM (k_flat_dirs_q; the cells next to a low edge get their direction
  // in the classification); RDGPU_FLAT_Q=0: k_flat_dirs_levels over the level planes and the DEM (r02-r04): A/B and tests
  const char *envq = getenv("RDGPU_FLAT_Q");
  const bool qpass = fused_classify && !(envq && envq[0] == '0');
  // r06: the level fields as bit planes per tile (flat_planes.inc); RDGPU_FLAT_PLANES=0, a level beyond 16 bits (an open flat
  // wider than 65 000 cells), or any of the A/B switches above: one int per cell (r02-r05).  The plane engine takes its
  // bitmaps and counts from the flag bytes, or straight from the classification (RDGPU_FLAT_CLASS_BITMAPS=1).
  const char *envp = getenv("RDGPU_FLAT_PLANES"), *envb = getenv("RDGPU_FLAT_CLASS_BITMAPS");
  const bool planes = qpass && use_bits_engine() && !(envp && envp[0] == '0') && !g_planes_overflowed && !getenv("RDGPU_FLAT_TRACE");
  // (measured r06: the start kernels drop from 2.9 to 0.4 ms and 4.8 GB of flag traffic go away, but the classification itself
  // goes from 4.9 to 6.6 ms with the ballots and row words in it -- 22.7 ms either way at S3; so the flags stay the default and
  // RDGPU_FLAT_CLASS_BITMAPS=1 selects the bitmaps: profiles/r06f_flats_class_bitmaps_ab.json)
  const bool class_bitmaps = planes && envb && envb[0] == '1';
  uint8_t *flags = class_bitmaps ? nullptr : ws.buf<uint8_t>("flats.flags", n);
  if (class_bitmaps) {
    const BitsScratch bt = bits_scratch(w, h, false, true);
    ClassBitmaps bm;
    bm.rows = bt.mbits; bm.nrows = bt.ntiles * BT;
    bm.counts = bt.counts; bm.tilesXb = bt.tilesX;
    RD_HIP(hipMemsetAsync(bt.counts, 0, (3 * 256 + 8) * sizeof(uint32_t), s));
    const uint32_t tilesX = (w + SW - 1) / SW, ntiles = tilesX * 2u * (uint32_t)((h + BT - 1) / BT);   // whole search tiles: their rows past the raster read as empty
    RD_LAUNCH("flats.dirs_classify", (k_dirs_classify<T, true, false, true>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, nodata,
              d_dirs, (uint8_t *)nullptr, w, h, tilesX, ntiles, bm);
  } else if (fused_classify) {
    const uint32_t tilesX = (w + SW - 1) / SW, ntiles = tilesX * ((h + KLH - 1) / KLH);
    if (qpass)
      RD_LAUNCH("flats.dirs_classify", (k_dirs_classify<T, true>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, nodata, d_dirs, flags,
                w, h, tilesX, ntiles);
    else
      RD_LAUNCH("flats.dirs_classify", (k_dirs_classify<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, nodata, d_dirs, flags, w, h,
                tilesX, ntiles);
  } else {
    launch_classify<T>(d_z, d_dirs, w, h, flags, s);
  }
  int32_t *TWd = planes ? nullptr : ws.buf<int32_t>("flats.mask", n), *A = nullptr;
  const char *envs = getenv("RDGPU_FLAT_STATIC");   // =0: the towards search in batches of rounds decided on the host (r02-r05): A/B and tests
  if (planes && away_beside() && !(envs && envs[0] == '0')) {
    // Everything enqueued, nothing read back: towards search (start, rounds, tail), the away search beside its tail on a side
    // stream, the last pass -- which looks at the overflow and abort words itself and leaves `dirs` alone when one is set.
    // Then ONE synchronisation: the tails' end states, the overflow words, the counts.
    Workspace::SideLane &lane = ws.side_lane(1);
    unsigned long long *d_c3 = reinterpret_cast<unsigned long long *>(ws.buf<uint32_t>("flats.c3", 8));
    StaticAway sa;
    Beside bs;
    bs.mark = [&]() { RD_HIP(hipEventRecord(lane.fork, s)); };   // (the rounds are done, the bitmaps made: before the tail's launch)
    bs.go = [&]() {                                              // (the resident wavefronts have their places: the away search beside them)
      RD_HIP(hipStreamWaitEvent(lane.stream, lane.fork, 0));
      sa = enqueue_away_static(flags, nullptr, w, h, lane.stream, true);
      RD_HIP(hipEventRecord(lane.join, lane.stream));
    };
    const StaticAway st = enqueue_towards_static(flags, d_c3, w, h, s, &bs);
    RD_HIP(hipStreamWaitEvent(s, lane.join, 0));
    const uint32_t *abT = st.run.Q.ctl + AQ_G_ABORT, *abA = sa.run.Q.ctl + AQ_G_ABORT;
    RD_LAUNCH("flats.dirs_q", k_flat_dirs_qp, dim3(xcd_grid((st.b.ntiles + 3) / 4)), dim3(NTHR), 0, s, st.b.pf, sa.b.pf,
              (const unsigned long long *)st.b.near, 1, d_dirs, w, h, st.b.tilesX, st.b.tilesY, (const uint32_t *)st.b.pf.overflow, abT, abA);
    unsigned long long c3[3] = {0, 0, 0};
    uint32_t over[2] = {0, 0};
    RD_HIP(hipMemcpyAsync(c3, d_c3, sizeof c3, hipMemcpyDeviceToHost, s));
    RD_HIP(hipMemcpyAsync(over, st.b.pf.overflow, sizeof over, hipMemcpyDeviceToHost, s));
    bool recT = false, recA = false;
    g_fstats.towards_levels = finish_towards_static(st, w, h, s, &recT);   // (the synchronisation)
    g_fstats.away_levels = finish_away_static(sa, nullptr, w, h, s, &recA);
    g_fstats.low_edges = c3[0];
    g_fstats.high_edges = c3[1];
    g_fstats.noflow_cells = c3[2];
    if (recT || recA) {   // a tail had given up and was finished in rounds (or the test switch says so): the last pass once more
      RD_HIP(hipMemcpyAsync(over, st.b.pf.overflow, sizeof over, hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
    }
    if (over[0] | over[1]) {   // a level beyond 16 bits: once more, on ints (the last pass left dirs as the classification wrote it)
      g_planes_overflowed = true;
      struct Reset { ~Reset() { g_planes_overflowed = false; } } reset;
      flat_resolution_device<T>(d_z, nodata, w, h, d_dirs, s);
      return;
    }
    if (recT || recA)
      RD_LAUNCH("flats.dirs_q", k_flat_dirs_qp, dim3(xcd_grid((st.b.ntiles + 3) / 4)), dim3(NTHR), 0, s, st.b.pf, sa.b.pf,
                (const unsigned long long *)st.b.near, 1, d_dirs, w, h, st.b.tilesX, st.b.tilesY, (const uint32_t *)nullptr,
                (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    return;
  }
  if (use_bits_engine()) {
    // no edge lists at all: the seeds are bitmaps made from the flags, the counts come with them
    unsigned long long c3[3] = {0, 0, 0};
    // the away search starts when the towards search enters its tail (the counts have arrived by then), on a side stream
    const bool beside = away_beside();
    Workspace::SideLane *lane = beside ? &ws.side_lane(1) : nullptr;
    StaticAway sa;
    bool started = false;
    Beside bs;
    bs.mark = [&]() {
      if (beside) RD_HIP(hipEventRecord(lane->fork, s));   // (the bitmaps are made, the flags and the DEM as the caller left them)
    };
    bs.go = [&]() {
      if (!beside || c3[0] == 0 || c3[1] == 0) return;
      RD_HIP(hipStreamWaitEvent(lane->stream, lane->fork, 0));
      if (!planes) A = ws.buf<int32_t>("flats.away", n);
      sa = enqueue_away_static(flags, A, w, h, lane->stream, planes);
      RD_HIP(hipEventRecord(lane->join, lane->stream));
      started = true;
    };
    g_fstats.towards_levels = run_bits_towards(flags, TWd, true, c3, w, h, s, &bs, planes);
    if (started) RD_HIP(hipStreamWaitEvent(s, lane->join, 0));
    g_fstats.low_edges = c3[0];
    g_fstats.high_edges = c3[1];
    g_fstats.noflow_cells = c3[2];
    if (c3[0] == 0) return;   // no flats, or none with an outlet (:475-481)
    bool have_away = started;
    if (c3[1] > 0 && !started) {
      if (!planes) A = ws.buf<int32_t>("flats.away", n);
      g_fstats.away_levels = run_bits_away(flags, nullptr, nullptr, A, false, w, h, s, planes);
      have_away = true;
    }
    if (started) g_fstats.away_levels = finish_away_static(sa, A, w, h, s);
    if (planes) {
      const BitsScratch bt = bits_scratch(w, h, false, true), ba = bits_scratch(w, h, true, true);
      uint32_t over[2] = {0, 0};
      RD_HIP(hipMemcpyAsync(over, bt.pf.overflow, sizeof over, hipMemcpyDeviceToHost, s));
      RD_HIP(hipStreamSynchronize(s));
      if (over[0] | over[1]) {   // a level beyond 16 bits: once more, on ints (dirs holds what the classification wrote: nothing is lost)
        g_planes_overflowed = true;
        struct Reset { ~Reset() { g_planes_overflowed = false; } } reset;
        flat_resolution_device<T>(d_z, nodata, w, h, d_dirs, s);
        return;
      }
      RD_LAUNCH("flats.dirs_q", k_flat_dirs_qp, dim3(xcd_grid((bt.ntiles + 3) / 4)), dim3(NTHR), 0, s, bt.pf, ba.pf,
                (const unsigned long long *)bt.near, have_away ? 1 : 0, d_dirs, w, h, bt.tilesX, bt.tilesY, (const uint32_t *)nullptr,
                (const uint32_t *)nullptr, (const uint32_t *)nullptr);
      return;
    }
  } else {
    uint32_t *low = nullptr, *highall = nullptr;
    uint32_t nlow = 0, nhigh_all = 0, nnoflow = 0;
    compact_edges(flags, n, &low, &nlow, &highall, &nhigh_all, &nnoflow, s);
    g_fstats.low_edges = nlow;
    g_fstats.noflow_cells = nnoflow;
    g_fstats.high_edges = nhigh_all;
    if (nlow == 0) return;   // no flats, or none with an outlet (:475-481)
    if (nhigh_all > 0) {
      // every high edge seeds (the flats without an outlet are not filtered out: that would need their labels; their
      // cells are never reached by the towards field, so they get no direction either way)
      A = ws.buf<int32_t>("flats.away", n);
      RD_HIP(hipMemsetAsync(A, 0x7F, n * sizeof(int32_t), s));
      g_fstats.away_levels = run_relax_away(d_dirs, A, highall, nhigh_all, nullptr, nullptr, w, h, s);
    }
    g_fstats.towards_levels = run_relax_towards(d_dirs, flags, TWd, w, h, s);
  }
  const uint32_t tilesX = (w + SW - 1) / SW, ntiles = tilesX * ((h + KLH - 1) / KLH);
  if (qpass)
    RD_LAUNCH("flats.dirs_q", k_flat_dirs_q, dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, (const int32_t *)TWd, (const int32_t *)A, d_dirs,
              w, h, tilesX, ntiles);
  else
    RD_LAUNCH("flats.dirs_levels", (k_flat_dirs_levels<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, (const int32_t *)TWd,
              (const int32_t *)A, d_dirs, w, h, tilesX, ntiles);
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
  if (!mask && !labels) {   // directions only: the path without labels and flat heights
    flat_resolution_device<T>(d, nodata, w, h, dd, s);
    RD_HIP(hipStreamSynchronize(s));
    RD_HIP(hipMemcpy(dirs, dd, n, hipMemcpyDeviceToHost));
    return;
  }
  flowdirs_device<T>(d, nodata, w, h, dd, MODE_D8, s);
  int32_t *M, *fh;
  uint32_t *L;
  resolve_flats_device<T>(d, dd, w, h, &M, &L, &fh, s);
  if (L) launch_masked_dirs<T>(d, M, dd, w, h, s);
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

// ------------------------------------------------------------------------------------------
// alter == true: d8_flats_alter_dem (flat_resolution.hpp:545-582) raises every cell of a drainable flat by
// flat_mask increments of nextafterf (in FLOAT precision, whatever the element type -- the reference calls
// nextafterf on every U), then plain d8_flow_directions runs on the altered DEM (:598-600).  Every element type: for
// integer ones the reference's nextafterf(v, numeric_limits<U>::infinity() == 0) walks values towards zero -- hardly
// what its author meant, but it is what the function returns, so it is what comes out here (alter_steps below).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float next_up_n(float v, uint32_t m) {   // nextafterf(v, +inf) applied m times
  uint32_t b = __builtin_bit_cast(uint32_t, v);
  if ((b & 0x7fffffffu) > 0x7f800000u) return v;                  // NaN
  if (b & 0x80000000u) {                                           // negative (incl. -0.0)
    const uint32_t mag = b & 0x7fffffffu;
    if (m <= mag) return __builtin_bit_cast(float, 0x80000000u | (mag - m));   // ... -denorm_min -> -0.0
    b = m - mag;                                                   // -0.0 -> +denorm_min is one step
  } else {
    const uint64_t nb = (uint64_t)b + m;
    b = nb >= 0x7f800000ull ? 0x7f800000u : (uint32_t)nb;          // saturates at +inf
  }
  return __builtin_bit_cast(float, b);
}

// m steps of the reference's `e = nextafterf(e, numeric_limits<U>::infinity())` (:567-568) in element type U.  Floating
// point U: float steps upwards (in FLOAT precision for double too: the reference's TODO).  Integer U: that infinity() is
// 0, so a step is (U)nextafterf((float)e, 0.0f) -- one towards zero while |e| < 2^24, one float spacing beyond (the
// conversion to float rounds to nearest even, as the host's) -- reproduced as is.
template <class T>
__device__ __forceinline__ T alter_steps(T v, uint32_t m) {
  constexpr bool sgn = std::is_signed<T>::value;
  using W = typename std::conditional<sgn, long long, unsigned long long>::type;
  constexpr long long LIM = 1ll << 24;
  W x = (W)v;
  if (sizeof(T) >= 4)
    while (m > 0 && ((long long)x > LIM || (sgn && (long long)x < -LIM) || (!sgn && x > (W)LIM))) {
      x = (W)__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)x) - 1u);   // (the magnitude sits below the sign bit)
      m--;
    }
  if (sgn) {
    const long long xx = (long long)x, mm = (long long)m;
    return (T)(xx > 0 ? (xx > mm ? xx - mm : 0) : (xx < -mm ? xx + mm : 0));
  }
  return (T)(x > (W)m ? x - (W)m : (W)0);
}
template <>
__device__ __forceinline__ float alter_steps<float>(float v, uint32_t m) { return next_up_n(v, m); }
template <>
__device__ __forceinline__ double alter_steps<double>(double v, uint32_t m) { return (double)next_up_n((float)v, m); }

template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_alter(T *z, const int32_t *__restrict__ M, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int32_t m = M[c];
    if (m <= 0) continue;   // cells outside drainable flats (labels == 0) and flat cells with mask 0
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;    // :556-558 interior only
    z[c] = alter_steps<T>(z[c], (uint32_t)m);                      // :567-568
  }
}

template <class T>
void flat_resolution_alter_device(T *d_z, T nodata, int w, int h, uint8_t *d_dirs, hipStream_t s) {
  if (!d_z || !d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_alter: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_alter: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_alter: raster too large");
  flowdirs_device<T>(d_z, nodata, w, h, d_dirs, MODE_D8, s);
  int32_t *M, *fh;
  uint32_t *L;
  resolve_flats_device<T>(d_z, d_dirs, w, h, &M, &L, &fh, s);
  if (L) {
    RD_LAUNCH("flats.alter", (k_flat_alter<T>), dim3(sgrid((uint64_t)w * h)), dim3(NTHR), 0, s, d_z, (const int32_t *)M, w, h);
    flowdirs_device<T>(d_z, nodata, w, h, d_dirs, MODE_D8, s);
  }
}

template <class T>
static void flat_resolution_alter_host(T *dem, T nodata, int w, int h, uint8_t *dirs) {
  if (!dem || !dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_alter: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_alter: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  uint8_t *dd = Workspace::get().buf<uint8_t>("host.dirs", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  flat_resolution_alter_device<T>(d, nodata, w, h, dd, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dirs, dd, n, hipMemcpyDeviceToHost));
  RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));   // the DEM is altered in place
}


// ------------------------------------------------------------------------------------------
// Row-block shards.  A shard works on its rows plus two ghost rows per cut:
// with them d8_flow_directions and the edge classification of every own cell are exact.  What crosses a
// cut is (1) the two distance fields -- each shard relaxes to its local fixed point, the cut rows are
// exchanged, the ghost rows lowered, and that repeats until no cut row changes anywhere -- and (2) the
// deepest away level per flat: local components are glued through the cells both neighbours hold (one
// small union-find over the cut rows, solved redundantly by every rank).  "Has an outlet" is read off the
// towards distances (a NO_FLOW cell is in a drainable flat <=> the towards relaxation reaches it), so the
// towards field is built first and the away sources are filtered through it.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void k_fs_inject(int32_t *D, const int32_t *__restrict__ in, int ghost_row, int ty,
                                                    int w, uint8_t *tile_active, uint32_t tilesX) {
  const int x = blockIdx.x * NTHR + threadIdx.x;
  if (x >= w) return;
  const int32_t v = in[x];
  int32_t *g = &D[(size_t)ghost_row * w + x];
  if (v < *g) {
    *g = v;   // ty: the tile row that holds the own row next to the ghost row
    for (int tx = max(x - 1, 0) / CW; tx <= min(x + 1, w - 1) / CW; tx++) tile_active[(uint32_t)ty * tilesX + (uint32_t)tx] = 1;
  }
}

__global__ __launch_bounds__(NTHR) void k_fs_ghost_dirs(uint8_t *dirs, uint32_t n) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  if (i < n) dirs[i] = dirs[i] == 0 ? DIR_GHOST_NOFLOW : 1;
}

// which cells of a ghost row take part (its NO_FLOW cells), one 64-bit mask per tile column
__global__ __launch_bounds__(64) void k_fs_ghost_mask(const uint8_t *__restrict__ dirs_row, int w, unsigned long long *out) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const unsigned long long m = __ballot(x < w && dirs_row[x] == DIR_GHOST_NOFLOW);
  if (threadIdx.x == 0) out[blockIdx.x] = m;
}

struct CutRows { int row[4]; };   // ghost above, first own, last own, ghost below (-1: no such row)

__global__ __launch_bounds__(NTHR) void k_fs_first(const uint32_t *__restrict__ L, uint32_t *first, CutRows cr, int w, int pass) {
  const int p = blockIdx.x * NTHR + threadIdx.x;
  if (p >= 4 * w) return;
  const int r = cr.row[p / w];
  if (r < 0) return;
  const uint32_t lab = L[(size_t)r * w + (p % w)];
  if (pass == 0) first[lab] = 0xFFFFFFFFu;
  else atomicMin(&first[lab], (uint32_t)p);
}

// out[0..4w) = lowest cut-row position with the same local label; out[4w..8w) = local flat height of that label
__global__ __launch_bounds__(NTHR) void k_fs_export(const uint32_t *__restrict__ L, const uint32_t *__restrict__ first,
                                                    const int32_t *__restrict__ fh, CutRows cr, int w, int32_t *out) {
  const int p = blockIdx.x * NTHR + threadIdx.x;
  if (p >= 4 * w) return;
  const int r = cr.row[p / w];
  int32_t rep = p, v = 0;
  if (r >= 0) {
    const uint32_t lab = L[(size_t)r * w + (p % w)];
    rep = (int32_t)first[lab];
    v = fh[lab];
  }
  out[p] = rep;
  out[4 * w + p] = v;
}

__global__ __launch_bounds__(NTHR) void k_fs_apply(const uint32_t *__restrict__ L, int32_t *fh, const int32_t *__restrict__ in,
                                                   CutRows cr, int w) {
  const int p = blockIdx.x * NTHR + threadIdx.x;
  if (p >= 4 * w) return;
  const int r = cr.row[p / w];
  if (r < 0) return;
  const uint32_t lab = L[(size_t)r * w + (p % w)];
  const int32_t v = in[p];
  if (__hip_atomic_load(&fh[lab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(&fh[lab], v);
}

// ---- the cut-row graph: nodes = (rank, cut-row position); gathered[r] = [4w rep | 4w height] ----
__global__ __launch_bounds__(NTHR) void k_fg_init(const int32_t *__restrict__ gathered, uint32_t *parent, int32_t *val,
                                                  int world, int w) {
  const uint32_t g = blockIdx.x * NTHR + threadIdx.x, per = 4u * (uint32_t)w;
  if (g >= (uint32_t)world * per) return;
  const uint32_t r = g / per;
  parent[g] = r * per + (uint32_t)gathered[(size_t)r * 2 * per + (g - r * per)];
  val[g] = 0;
}

__global__ __launch_bounds__(NTHR) void k_fg_glue(uint32_t *parent, int world, int w) {
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x, per = 4u * (uint32_t)w;
  if (i >= (uint32_t)(world - 1) * 2u * (uint32_t)w) return;
  const uint32_t r = i / (2u * (uint32_t)w), j = i % (2u * (uint32_t)w);
  // the same physical cell seen from both sides: last own row of r == ghost above of r+1;
  // ghost below of r == first own row of r+1
  uf_unite(parent, r * per + 2u * (uint32_t)w + j, (r + 1u) * per + j);
}

__global__ __launch_bounds__(NTHR) void k_fg_max(uint32_t *parent, int32_t *val, const int32_t *__restrict__ gathered,
                                                 int world, int w) {
  const uint32_t g = blockIdx.x * NTHR + threadIdx.x, per = 4u * (uint32_t)w;
  if (g >= (uint32_t)world * per) return;
  const uint32_t r = g / per, root = uf_find(parent, g);
  parent[g] = root;   // (only ever lowers a parent: safe next to concurrent finds)
  const int32_t v = gathered[(size_t)r * 2 * per + per + (g - r * per)];
  if (v > 0) atomicMax(&val[root], v);
}

__global__ __launch_bounds__(NTHR) void k_fg_out(const uint32_t *__restrict__ parent, const int32_t *__restrict__ val,
                                                 int32_t *out, uint32_t total) {
  const uint32_t g = blockIdx.x * NTHR + threadIdx.x;
  if (g < total) out[g] = val[parent[g]];
}

static void flat_graph_solve_device(const int32_t *d_gathered, int world, int w, int32_t *d_out, hipStream_t s) {
  if (!d_gathered || !d_out) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_graph_solve_dev: null pointer");
  if (world <= 0 || w <= 0 || (uint64_t)world * 4u * (uint64_t)w > 0x7FFF0000ull)
    throw Error(RDGPU_ERR_ARG, "rdgpu_flat_graph_solve_dev: bad dimensions");
  const uint32_t total = (uint32_t)world * 4u * (uint32_t)w;
  Workspace &ws = Workspace::get();
  uint32_t *parent = ws.buf<uint32_t>("flatgraph.parent", total);
  int32_t *val = ws.buf<int32_t>("flatgraph.val", total);
  const dim3 grid((total + NTHR - 1) / NTHR), blk(NTHR);
  RD_LAUNCH("flatgraph.init", k_fg_init, grid, blk, 0, s, d_gathered, parent, val, world, w);
  if (world > 1) {
    const uint32_t nglue = (uint32_t)(world - 1) * 2u * (uint32_t)w;
    RD_LAUNCH("flatgraph.glue", k_fg_glue, dim3((nglue + NTHR - 1) / NTHR), blk, 0, s, parent, world, w);
  }
  RD_LAUNCH("flatgraph.max", k_fg_max, grid, blk, 0, s, parent, val, d_gathered, world, w);
  RD_LAUNCH("flatgraph.out", k_fg_out, grid, blk, 0, s, (const uint32_t *)parent, (const int32_t *)val, d_out, total);
}

}  // namespace rdgpu

struct rdgpu_flat_shard {
  int w = 0, rows = 0, gtop = 0, gbot = 0;
  const void *z = nullptr;
  uint8_t *dirs = nullptr, *flags = nullptr;
  uint32_t *L = nullptr, *first = nullptr;
  int32_t *fh = nullptr, *D[2] = {nullptr, nullptr};
  uint32_t *src[2] = {nullptr, nullptr};
  uint32_t nsrc[2] = {0, 0};
  uint8_t *tflags[2] = {nullptr, nullptr};
  uint32_t *tlist = nullptr, *ctr = nullptr;
  // the bitmap engine's state (RDGPU_FLAT_BITS=0: the stencil relaxation instead): cells that take part per tile row,
  // the same for the two ghost rows next to the own rows, "tile visited" per field
  bool bits = false;
  unsigned long long *mbits = nullptr, *gmask[2] = {nullptr, nullptr};
  uint8_t *expanded[2] = {nullptr, nullptr};
  bool seeded[2] = {false, false};
  uint32_t rounds[2] = {0, 0};
  hipStream_t stream = nullptr;
  std::vector<void *> owned;
  void (*relax_fn)(rdgpu_flat_shard *, int) = nullptr;
  void (*finish_fn)(rdgpu_flat_shard *, const int32_t *, uint8_t *) = nullptr;
};

namespace rdgpu {

static CutRows cut_rows(const rdgpu_flat_shard *f) {
  CutRows cr;
  cr.row[0] = f->gtop ? f->gtop - 1 : -1;
  cr.row[1] = f->gtop;
  cr.row[2] = f->rows - f->gbot - 1;
  cr.row[3] = f->gbot ? f->rows - f->gbot : -1;
  return cr;
}

static void fs_free(rdgpu_flat_shard *f) {
  if (!f) return;
  for (void *p : f->owned) (void)hipFree(p);
  delete f;
}

template <class T>
static void fs_relax(rdgpu_flat_shard *f, int phase) {
  hipStream_t s = f->stream;
  const int w = f->w, h = f->rows;
  const uint32_t tilesX = (w + CW - 1) / CW, tilesY = (h + RCH - 1) / RCH;
  const int row_lo = f->gtop, row_hi = h - f->gbot;   // the own rows: ghost rows feed, they are not relaxed here
  int32_t *D = f->D[phase];
  if (f->bits) {
    BitsScratch b;
    b.tilesX = (w + BT - 1) / BT; b.tilesY = (row_hi - row_lo + BT - 1) / BT; b.ntiles = b.tilesX * b.tilesY;
    b.mbits = f->mbits; b.tflags = f->tflags[phase]; b.expanded = f->expanded[phase]; b.tlist = f->tlist; b.ctr = f->ctr;
    b.counts = nullptr;
    const RowWin win{row_lo, row_hi, f->gtop ? f->gmask[0] : nullptr, f->gbot ? f->gmask[1] : nullptr};
    if (!f->seeded[phase]) {
      f->seeded[phase] = true;
      RD_HIP(hipMemsetAsync(b.expanded, 0, b.ntiles, s));
      if (phase == 0)
        RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<true, true>), dim3(b.ntiles), dim3(NTHR), 0, s, (const uint8_t *)f->flags,
                  (const uint32_t *)nullptr, (const int32_t *)nullptr, D, b.mbits, b.tflags, (uint32_t *)nullptr, w, win,
                  (const int32_t *)nullptr, b.tilesX, b.tilesY);
      else   // away sources: the high edges of flats the towards levels reach ("has an outlet")
        RD_LAUNCH("flats.bits_prepare", (k_bits_prepare<false, false>), dim3(b.ntiles), dim3(NTHR), 0, s, (const uint8_t *)f->flags,
                  (const uint32_t *)nullptr, (const int32_t *)nullptr, D, b.mbits, b.tflags, (uint32_t *)nullptr, w, win,
                  (const int32_t *)f->D[0], b.tilesX, b.tilesY);
    }
    f->rounds[phase] += phase ? relax_rounds_bits<1>(b, D, w, h, "flats.relax_away", s, win)
                              : relax_rounds_bits<2>(b, D, w, h, "flats.relax_towards", s, win);
    return;
  }
  if (!f->seeded[phase]) {
    f->seeded[phase] = true;
    if (phase == 0)
      RD_LAUNCH("flats.init_towards", k_flat_init_towards, dim3(tilesX * tilesY), dim3(NTHR), 0, s, (const uint8_t *)f->flags, D,
                f->tflags[0], w, h, row_lo, row_hi, tilesX, tilesY);
    else if (f->nsrc[1])
      RD_LAUNCH("flats.seed", k_flat_seed, dim3((f->nsrc[1] + NTHR - 1) / NTHR), dim3(NTHR), 0, s,
                (const uint32_t *)f->src[1], f->nsrc[1], (const uint32_t *)f->L, (const int32_t *)nullptr, D,
                f->tflags[1], w, tilesX, tilesY, (const int32_t *)f->D[0]);
  }
  f->rounds[phase] += relax_rounds((const uint8_t *)f->dirs, D, f->tflags[phase], f->tlist, f->ctr, w, h, row_lo, row_hi,
                                   phase ? "flats.relax_away" : "flats.relax_towards", s);
}

template <class T>
static void fs_finish(rdgpu_flat_shard *f, const int32_t *d_heights, uint8_t *d_dirs_out) {
  hipStream_t s = f->stream;
  const int w = f->w, h = f->rows;
  const uint64_t n = (uint64_t)w * h;
  if (d_heights)
    RD_LAUNCH("flatshard.apply", k_fs_apply, dim3((4 * w + NTHR - 1) / NTHR), dim3(NTHR), 0, s, (const uint32_t *)f->L, f->fh,
              d_heights, cut_rows(f), w);
  RD_LAUNCH("flats.combine", k_flat_combine, dim3(sgrid(n)), dim3(NTHR), 0, s, f->D[0], (const int32_t *)f->D[1],
            (const uint32_t *)f->L, (const int32_t *)f->fh, n);
  launch_masked_dirs<T>(static_cast<const T *>(f->z), (const int32_t *)f->D[0], f->dirs, w, h, s);
  RD_HIP(hipMemcpyAsync(d_dirs_out, f->dirs + (size_t)f->gtop * w, (size_t)(h - f->gtop - f->gbot) * w, hipMemcpyDeviceToDevice, s));
}

template <class T>
static rdgpu_flat_shard *fs_begin(const T *d_z, T nodata, int w, int rows, int gtop, int gbot, hipStream_t s) {
  if (!d_z) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_begin: null pointer");
  if (w <= 0 || rows <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_begin: width and rows must be positive");
  if ((gtop != 0 && gtop != 2) || (gbot != 0 && gbot != 2) || rows - gtop - gbot < 1)
    throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_begin: a cut needs exactly two ghost rows, and the shard at least one own row");
  const uint64_t n = (uint64_t)w * rows;
  if (n > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_begin: shard too large");
  rdgpu_flat_shard *f = new rdgpu_flat_shard();
  try {
    f->w = w; f->rows = rows; f->gtop = gtop; f->gbot = gbot; f->z = d_z; f->stream = s;
    f->relax_fn = &fs_relax<T>;
    f->finish_fn = &fs_finish<T>;
    auto alloc = [&](size_t bytes) {
      void *p = nullptr;
      RD_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
      f->owned.push_back(p);
      return p;
    };
    const uint32_t tilesX = (w + CW - 1) / CW, ntiles = tilesX * ((rows + CH - 1) / CH);   // labelling tiles
    const uint32_t rtiles = tilesX * ((rows + RCH - 1) / RCH);                               // relaxation tiles
    f->dirs = (uint8_t *)alloc(n);
    f->flags = (uint8_t *)alloc(n);
    f->L = (uint32_t *)alloc(n * 4);
    f->first = (uint32_t *)alloc(n * 4);
    f->fh = (int32_t *)alloc(n * 4);
    f->D[0] = (int32_t *)alloc(n * 4);
    f->D[1] = (int32_t *)alloc(n * 4);
    f->tflags[0] = (uint8_t *)alloc(rtiles);
    f->tflags[1] = (uint8_t *)alloc(rtiles);
    f->tlist = (uint32_t *)alloc((size_t)rtiles * 4);
    f->ctr = (uint32_t *)alloc(BITS_BATCH * sizeof(uint32_t));
    f->bits = use_bits_engine();
    if (f->bits) {   // (64 x 64 tiles over the own rows: never more than the 64 x 32 tiles over all rows)
      const uint32_t btiles = tilesX * (uint32_t)((rows - gtop - gbot + BT - 1) / BT);
      f->mbits = (unsigned long long *)alloc((size_t)btiles * BT * 8);
      f->gmask[0] = (unsigned long long *)alloc((size_t)tilesX * 8);
      f->gmask[1] = (unsigned long long *)alloc((size_t)tilesX * 8);
      f->expanded[0] = (uint8_t *)alloc(btiles);
      f->expanded[1] = (uint8_t *)alloc(btiles);
    }
    flowdirs_device<T>(d_z, nodata, w, rows, f->dirs, MODE_D8, s);
    launch_classify<T>(d_z, (const uint8_t *)f->dirs, w, rows, f->flags, s);
    // sources are own cells only; the ghost rows' distances arrive from their owners
    if (gtop) RD_HIP(hipMemsetAsync(f->flags, 0, (size_t)gtop * w, s));
    if (gbot) RD_HIP(hipMemsetAsync(f->flags + (size_t)(rows - gbot) * w, 0, (size_t)gbot * w, s));
    const uint8_t masks[2] = {F_LOW, F_HIGH};
    for (int ph = 0; ph < 2; ph++) {
      uint32_t *list = nullptr;
      const uint32_t cnt = compact_flags(f->flags, masks[ph], n, ph ? "flats.highall" : "flats.low", &list, s);
      f->nsrc[ph] = cnt;
      if (cnt) {
        f->src[ph] = (uint32_t *)alloc((size_t)cnt * 4);
        RD_HIP(hipMemcpyAsync(f->src[ph], list, (size_t)cnt * 4, hipMemcpyDeviceToDevice, s));
      }
    }
    // ghost cells are never relaxed or given a direction here: their NO_FLOW cells keep feeding (DIR_GHOST_NOFLOW),
    // the others are marked as "has a direction"
    if (gtop) RD_LAUNCH("flatshard.ghost_dirs", k_fs_ghost_dirs, dim3(((size_t)gtop * w + NTHR - 1) / NTHR), dim3(NTHR), 0, s, f->dirs, (uint32_t)gtop * (uint32_t)w);
    if (gbot) RD_LAUNCH("flatshard.ghost_dirs", k_fs_ghost_dirs, dim3(((size_t)gbot * w + NTHR - 1) / NTHR), dim3(NTHR), 0, s, f->dirs + (size_t)(rows - gbot) * w, (uint32_t)gbot * (uint32_t)w);
    if (f->bits) {
      if (gtop) RD_LAUNCH("flatshard.ghost_mask", k_fs_ghost_mask, dim3(tilesX), dim3(64), 0, s, (const uint8_t *)f->dirs + (size_t)(gtop - 1) * w, w, f->gmask[0]);
      if (gbot) RD_LAUNCH("flatshard.ghost_mask", k_fs_ghost_mask, dim3(tilesX), dim3(64), 0, s, (const uint8_t *)f->dirs + (size_t)(rows - gbot) * w, w, f->gmask[1]);
    }
    RD_LAUNCH("flats.ccl_tile", (k_ccl_tile<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, f->L, w, rows, tilesX, ntiles, (T *)nullptr,
              (uint32_t *)nullptr);
    launch_ccl_border<T>(d_z, f->L, w, rows, s);
    RD_LAUNCH("flats.ccl_flatten", k_ccl_flatten, dim3(sgrid(n)), dim3(NTHR), 0, s, f->L, n);
    RD_HIP(hipMemsetAsync(f->fh, 0, n * 4, s));
    RD_HIP(hipMemsetAsync(f->D[0], 0x7F, n * 4, s));
    RD_HIP(hipMemsetAsync(f->D[1], 0x7F, n * 4, s));
    RD_HIP(hipMemsetAsync(f->tflags[0], 0, rtiles, s));
    RD_HIP(hipMemsetAsync(f->tflags[1], 0, rtiles, s));
    RD_HIP(hipStreamSynchronize(s));
  } catch (...) {
    fs_free(f);
    throw;
  }
  return f;
}

static void fs_check(const rdgpu_flat_shard *f, int phase, const char *who) {
  if (!f) throw Error(RDGPU_ERR_ARG, std::string(who) + ": null handle");
  if (phase != 0 && phase != 1) throw Error(RDGPU_ERR_ARG, std::string(who) + ": phase must be 0 (towards) or 1 (away)");
}

static void fs_boundary(rdgpu_flat_shard *f, int phase, int32_t *d_out) {
  if (!d_out) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_boundary: null pointer");
  const CutRows cr = cut_rows(f);
  const size_t rb = (size_t)f->w * 4;
  RD_HIP(hipMemcpyAsync(d_out, f->D[phase] + (size_t)cr.row[1] * f->w, rb, hipMemcpyDeviceToDevice, f->stream));
  RD_HIP(hipMemcpyAsync(d_out + f->w, f->D[phase] + (size_t)cr.row[2] * f->w, rb, hipMemcpyDeviceToDevice, f->stream));
}

static void fs_inject(rdgpu_flat_shard *f, int phase, const int32_t *d_above, const int32_t *d_below) {
  const CutRows cr = cut_rows(f);
  const uint32_t tilesX = (f->w + CW - 1) / CW;
  const dim3 grid((f->w + NTHR - 1) / NTHR), blk(NTHR);
  // the tile row of the own row next to the ghost row, in the engine's tiling (64-row tiles from the first own row / 32-row
  // tiles from row 0)
  auto tile_row = [&](int own_row) { return f->bits ? (own_row - f->gtop) / BT : own_row / RCH; };
  if (d_above && cr.row[0] >= 0)
    RD_LAUNCH("flatshard.inject", k_fs_inject, grid, blk, 0, f->stream, f->D[phase], d_above, cr.row[0], tile_row(cr.row[1]), f->w,
              f->tflags[phase], tilesX);
  if (d_below && cr.row[3] >= 0)
    RD_LAUNCH("flatshard.inject", k_fs_inject, grid, blk, 0, f->stream, f->D[phase], d_below, cr.row[3], tile_row(cr.row[2]), f->w,
              f->tflags[phase], tilesX);
}

static void fs_heights(rdgpu_flat_shard *f, int32_t *d_out) {
  if (!d_out) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_heights: null pointer");
  hipStream_t s = f->stream;
  const uint64_t n = (uint64_t)f->w * f->rows;
  RD_LAUNCH("flats.height", k_flat_height, dim3(sgrid(n)), dim3(NTHR), 0, s, (const int32_t *)f->D[1], (const uint32_t *)f->L,
            f->fh, n);
  const CutRows cr = cut_rows(f);
  const dim3 grid((4 * f->w + NTHR - 1) / NTHR), blk(NTHR);
  RD_LAUNCH("flatshard.first", k_fs_first, grid, blk, 0, s, (const uint32_t *)f->L, f->first, cr, f->w, 0);
  RD_LAUNCH("flatshard.first", k_fs_first, grid, blk, 0, s, (const uint32_t *)f->L, f->first, cr, f->w, 1);
  RD_LAUNCH("flatshard.export", k_fs_export, grid, blk, 0, s, (const uint32_t *)f->L, (const uint32_t *)f->first,
            (const int32_t *)f->fh, cr, f->w, d_out);
}

// ------------------------------------------------------------------------------------------
// ResolveFlatsEpsilon (flats/flats.hpp:21-28) = FindFlats (flats/find_flats.hpp:29-69) + GetFlatMask
// (flats/Barnes2014.hpp:398-467) + ResolveFlatsEpsilon_Barnes2014 (:496-550): what rd.ResolveFlats calls.
// GetFlatMask is the same BFS construction as resolve_flats_barnes, applied to FindFlats' notion of a flat
// cell (interior, no lower neighbour AND no NoData neighbour).  So FindFlats is written as a pseudo direction
// raster (0 = flat, 1 = not, 255 = NoData) and the engine above runs unchanged -- the two facts it leans on
// (adjacent flat cells have equal elevation; drainable <=> reached by the towards gradient) hold for this
// definition too.  Then every interior cell with flat_mask > 0 is raised by flat_mask increments of
// std::nextafter(e, numeric_limits<T>::infinity()) in ITS OWN type: float and double in closed form on the bit
// pattern; for integer T that "infinity" is 0 and the arguments promote to double, so one step moves the value
// by one towards zero -- reproduced as is (it is what the reference returns).
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(NTHR) void k_find_flats(const T *__restrict__ z, T nodata, uint8_t *__restrict__ flats, int w,
                                                     int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[SLH * SLW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * SW, y0 = (int)(t / tilesX) * SH;
  for (int i = threadIdx.x; i < SLH * SLW; i += NTHR) {
    const int ly = i / SLW, lx = i - ly * SLW;
    const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);   // clamped: edge cells are decided by position
    sz[i] = z[(size_t)gy * w + gx];
  }
  __syncthreads();
  const int lx = threadIdx.x & (SW - 1), ly0 = threadIdx.x >> 6;
  const int off[9] = {0, -1, -SLW - 1, -SLW, -SLW + 1, 1, SLW + 1, SLW, SLW - 1};
#pragma unroll
  for (int j = 0; j < SH / 4; j++) {
    const int ly = ly0 + 4 * j, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const int o = (ly + 1) * SLW + lx + 1;
    const T e = sz[o];
    uint8_t f;
    if (e == nodata) f = 255;                                                  // find_flats.hpp:43-46
    else if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) f = 1;          // :48-51
    else {
      f = 0;
#pragma unroll
      for (int k = 1; k <= 8; k++) {
        const T zn = sz[o + off[k]];
        if (zn < e || zn == nodata) f = 1;                                     // :56-63
      }
    }
    flats[(size_t)gy * w + gx] = f;
  }
}

__device__ __forceinline__ double next_up_n64(double v, uint32_t m) {   // nextafter(v, +inf) applied m times
  uint64_t b = __builtin_bit_cast(uint64_t, v);
  if ((b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return v;    // NaN
  if (b >> 63) {
    const uint64_t mag = b & 0x7fffffffffffffffull;
    if (m <= mag) return __builtin_bit_cast(double, 0x8000000000000000ull | (mag - m));
    b = m - mag;
  } else {
    b += m;
    if (b >= 0x7ff0000000000000ull) b = 0x7ff0000000000000ull;
  }
  return __builtin_bit_cast(double, b);
}
template <class T>
__device__ __forceinline__ T epsilon_steps(T v, uint32_t m) {   // m steps towards zero, stopping there
  const long long x = (long long)v, mm = (long long)m;
  return (T)(x > 0 ? (x > mm ? x - mm : 0) : (x < -mm ? x + mm : 0));
}
// 64-bit integers: the reference's step is (T)nextafter((double)v, 0.0).  Up to 2^53 in magnitude that is "one towards
// zero" as above; beyond, (double)v rounds to the double grid and the step is one grid spacing -- reproduced step by
// step while the value is that large (the conversions are the hardware's round-to-nearest-even, as the host's).
__device__ __forceinline__ double toward_zero_1(double d) {   // nextafter(d, 0.0) for |d| >= 2^53
  return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, d) - 1ull);   // (the magnitude sits below the sign bit)
}
template <>
__device__ __forceinline__ int64_t epsilon_steps<int64_t>(int64_t v, uint32_t m) {
  constexpr long long LIM = 1ll << 53;
  long long x = v;
  while (m > 0 && (x > LIM || x < -LIM)) { x = (long long)toward_zero_1((double)x); m--; }
  const long long mm = (long long)m;
  return x > 0 ? (x > mm ? x - mm : 0) : (x < -mm ? x + mm : 0);
}
template <>
__device__ __forceinline__ uint64_t epsilon_steps<uint64_t>(uint64_t v, uint32_t m) {
  constexpr unsigned long long LIM = 1ull << 53;
  unsigned long long x = v;
  while (m > 0 && x > LIM) { x = (unsigned long long)toward_zero_1((double)x); m--; }
  return x > (unsigned long long)m ? x - m : 0ull;
}
template <>
__device__ __forceinline__ float epsilon_steps<float>(float v, uint32_t m) { return next_up_n(v, m); }
template <>
__device__ __forceinline__ double epsilon_steps<double>(double v, uint32_t m) { return next_up_n64(v, m); }

template <class T>
// TW: the towards levels; the mask (k_flat_combine's formula) is formed here, so it is neither written nor read back
__global__ __launch_bounds__(NTHR) void k_flat_epsilon(T *z, const int32_t *__restrict__ TW, const int32_t *__restrict__ A,
                                                       const uint32_t *__restrict__ L, const int32_t *__restrict__ fh, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int32_t t = TW[c];
    if (t >= DINF) continue;   // not in a drainable flat: mask 0
    const int32_t a = A ? A[c] : DINF;
    const int32_t m = (a < DINF ? fh[L[c]] - a : 0) + 2 * t;   // a low edge has t = 1 and no away level: 2
    if (m <= 0) continue;   // unlabelled cells, and labelled cells outside the flat proper (mask 0)
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;    // Barnes2014.hpp:511-512 interior only
    z[c] = epsilon_steps<T>(z[c], (uint32_t)m);                    // :527-528
  }
}

// the same, four cells a thread (16-byte loads of the level fields; a vector of elevations is stored only when one of
// its cells moved)
template <class T>
struct alignas(sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4) ElevQuad { T v[4]; };
template <class T>
__global__ __launch_bounds__(NTHR) void k_flat_epsilon4(T *z, const int32_t *__restrict__ TW, const int32_t *__restrict__ A,
                                                        const uint32_t *__restrict__ L, const int32_t *__restrict__ fh, int w, int h) {
  const uint64_t n = (uint64_t)w * h, n4 = n / 4, stride = (uint64_t)gridDim.x * NTHR;
  auto mask_of = [&](int32_t t, int32_t a, uint32_t l, uint64_t c) -> uint32_t {
    if (t >= DINF) return 0u;
    const int32_t m = (a < DINF ? fh[l] - a : 0) + 2 * t;
    if (m <= 0) return 0u;
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    if (x == 0 || y == 0 || x == w - 1 || y == h - 1) return 0u;
    return (uint32_t)m;
  };
  for (uint64_t q = (uint64_t)blockIdx.x * NTHR + threadIdx.x; q < n4 + (n & 3); q += stride) {
    if (q < n4) {
      const int4 t = reinterpret_cast<const int4 *>(TW)[q];
      if (t.x >= DINF && t.y >= DINF && t.z >= DINF && t.w >= DINF) continue;
      int4 a = make_int4(DINF, DINF, DINF, DINF);
      uint4 l = make_uint4(0, 0, 0, 0);
      if (A) {
        a = reinterpret_cast<const int4 *>(A)[q];
        if (a.x < DINF || a.y < DINF || a.z < DINF || a.w < DINF) l = reinterpret_cast<const uint4 *>(L)[q];
      }
      const uint32_t m[4] = {mask_of(t.x, a.x, l.x, 4 * q), mask_of(t.y, a.y, l.y, 4 * q + 1), mask_of(t.z, a.z, l.z, 4 * q + 2),
                             mask_of(t.w, a.w, l.w, 4 * q + 3)};
      if (!(m[0] | m[1] | m[2] | m[3])) continue;
      ElevQuad<T> e = reinterpret_cast<const ElevQuad<T> *>(z)[q];
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (m[k]) e.v[k] = epsilon_steps<T>(e.v[k], m[k]);
      reinterpret_cast<ElevQuad<T> *>(z)[q] = e;
    } else {
      const uint64_t c = 4 * n4 + (q - n4);
      const int32_t a = A ? A[c] : DINF;
      const uint32_t m = mask_of(TW[c], a, a < DINF ? L[c] : 0u, c);
      if (m) z[c] = epsilon_steps<T>(z[c], m);
    }
  }
}

template <class T>
void resolve_flats_epsilon_device(T *d_z, T nodata, int w, int h, hipStream_t s) {
  if (!d_z) throw Error(RDGPU_ERR_ARG, "rdgpu_resolve_flats_epsilon: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_resolve_flats_epsilon: width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_resolve_flats_epsilon: raster too large");
  // r05: on the lean path (the default) FindFlats is part of the classification pass; RDGPU_RFE_FUSED_FINDFLATS=0 or the other
  // paths: k_find_flats writes the pseudo direction raster first
  const char *envf = getenv("RDGPU_RFE_FUSED_FINDFLATS");
  const bool fused_ff = use_bits_engine() && lean_labels() && !(envf && envf[0] == '0');
  uint8_t *flats = nullptr;
  if (!fused_ff) {
    flats = Workspace::get().buf<uint8_t>("flats.findflats", (size_t)w * h);
    uint32_t tilesX;
    const uint32_t ntiles = stencil_tiles(w, h, &tilesX);
    RD_LAUNCH("flats.find_flats", (k_find_flats<T>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, (const T *)d_z, nodata, flats, w, h,
              tilesX, ntiles);
  }
  int32_t *M, *fh;
  uint32_t *L;
  const int32_t *A = nullptr;
  resolve_flats_device<T>(d_z, flats, w, h, &M, &L, &fh, s, &A, nodata);
  if (L) {
    if (reinterpret_cast<uintptr_t>(d_z) % 16 == 0)
      RD_LAUNCH("flats.epsilon", (k_flat_epsilon4<T>), dim3(sgrid((uint64_t)w * h / 4 + 3)), dim3(NTHR), 0, s, d_z, (const int32_t *)M, A,
                (const uint32_t *)L, (const int32_t *)fh, w, h);
    else   // (a DEM pointer that is not 16-byte aligned: one cell a thread)
      RD_LAUNCH("flats.epsilon", (k_flat_epsilon<T>), dim3(sgrid((uint64_t)w * h)), dim3(NTHR), 0, s, d_z, (const int32_t *)M, A,
                (const uint32_t *)L, (const int32_t *)fh, w, h);
  }
}

template <class T>
static void resolve_flats_epsilon_host(T *dem, T nodata, int w, int h) {
  if (!dem) throw Error(RDGPU_ERR_ARG, "rdgpu_resolve_flats_epsilon: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_resolve_flats_epsilon: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  resolve_flats_epsilon_device<T>(d, nodata, w, h, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
}

#define RD_INST(T) template void flat_resolution_device<T>(const T *, T, int, int, uint8_t *, hipStream_t);
RD_INST(uint8_t) RD_INST(int16_t) RD_INST(uint16_t) RD_INST(int32_t) RD_INST(uint32_t) RD_INST(float) RD_INST(double)
RD_INST(int8_t) RD_INST(int64_t) RD_INST(uint64_t)
#undef RD_INST

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FLATS_API(SUF, T)                                                                                   \
  extern "C" int rdgpu_flat_resolution_d8_multi_##SUF(const T *, T, int, int, uint8_t *, const int *, int);    \
  extern "C" int rdgpu_flat_resolution_d8_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs) {         \
    const std::vector<int> devs = env_devices();   /* RDGPU_DEVICES: the node's GPUs, csrc/multi.hip */       \
    if (devs.size() > 1 && h >= 2 * (int)devs.size())                                                          \
      return rdgpu_flat_resolution_d8_multi_##SUF(dem, nodata, w, h, dirs, devs.data(), (int)devs.size());     \
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
RD_FLATS_API(i8, int8_t)
RD_FLATS_API(i64, int64_t)
RD_FLATS_API(u64, uint64_t)

#define RD_FLATS_ALTER_API(SUF, T)                                                                             \
  extern "C" int rdgpu_flat_resolution_d8_alter_##SUF(T *dem, T nodata, int w, int h, uint8_t *dirs) {         \
    return guarded([&] { flat_resolution_alter_host<T>(dem, nodata, w, h, dirs); });                           \
  }                                                                                                            \
  extern "C" int rdgpu_flat_resolution_d8_alter_dev_##SUF(T *d_dem, T nodata, int w, int h, uint8_t *d_dirs,   \
                                                          void *stream) {                                      \
    return guarded([&] { flat_resolution_alter_device<T>(d_dem, nodata, w, h, d_dirs, (hipStream_t)stream); }); \
  }
RD_FLATS_ALTER_API(f32, float)
RD_FLATS_ALTER_API(f64, double)
RD_FLATS_ALTER_API(u8, uint8_t)
RD_FLATS_ALTER_API(i8, int8_t)
RD_FLATS_ALTER_API(i16, int16_t)
RD_FLATS_ALTER_API(u16, uint16_t)
RD_FLATS_ALTER_API(i32, int32_t)
RD_FLATS_ALTER_API(u32, uint32_t)
RD_FLATS_ALTER_API(i64, int64_t)
RD_FLATS_ALTER_API(u64, uint64_t)


#define RD_RFE_API(SUF, T)                                                                                     \
  extern "C" int rdgpu_resolve_flats_epsilon_##SUF(T *dem, T nodata, int w, int h) {                           \
    return guarded([&] { resolve_flats_epsilon_host<T>(dem, nodata, w, h); });                                 \
  }                                                                                                            \
  extern "C" int rdgpu_resolve_flats_epsilon_dev_##SUF(T *d_dem, T nodata, int w, int h, void *stream) {       \
    return guarded([&] { resolve_flats_epsilon_device<T>(d_dem, nodata, w, h, (hipStream_t)stream); });        \
  }
RD_RFE_API(u8, uint8_t)
RD_RFE_API(i16, int16_t)
RD_RFE_API(u16, uint16_t)
RD_RFE_API(i32, int32_t)
RD_RFE_API(u32, uint32_t)
RD_RFE_API(f32, float)
RD_RFE_API(f64, double)
RD_RFE_API(i8, int8_t)
RD_RFE_API(i64, int64_t)
RD_RFE_API(u64, uint64_t)

#define RD_FLATSHARD_API(SUF, T)                                                                                       \
  extern "C" int rdgpu_flat_shard_begin_##SUF(const T *d_rows, T nodata, int w, int rows, int ghost_top,               \
                                              int ghost_bottom, void *stream, rdgpu_flat_shard **out) {                \
    return guarded([&] {                                                                                               \
      if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_begin: null out pointer");                                \
      *out = fs_begin<T>(d_rows, nodata, w, rows, ghost_top, ghost_bottom, (hipStream_t)stream);                       \
    });                                                                                                                \
  }
RD_FLATSHARD_API(u8, uint8_t)
RD_FLATSHARD_API(i16, int16_t)
RD_FLATSHARD_API(u16, uint16_t)
RD_FLATSHARD_API(i32, int32_t)
RD_FLATSHARD_API(u32, uint32_t)
RD_FLATSHARD_API(f32, float)
RD_FLATSHARD_API(f64, double)
RD_FLATSHARD_API(i8, int8_t)
RD_FLATSHARD_API(i64, int64_t)
RD_FLATSHARD_API(u64, uint64_t)

extern "C" int rdgpu_flat_shard_relax(rdgpu_flat_shard *f, int phase) {
  return guarded([&] { fs_check(f, phase, "rdgpu_flat_shard_relax"); f->relax_fn(f, phase); });
}
extern "C" int rdgpu_flat_shard_boundary(rdgpu_flat_shard *f, int phase, int32_t *d_out) {
  return guarded([&] { fs_check(f, phase, "rdgpu_flat_shard_boundary"); fs_boundary(f, phase, d_out); });
}
extern "C" int rdgpu_flat_shard_inject(rdgpu_flat_shard *f, int phase, const int32_t *d_above, const int32_t *d_below) {
  return guarded([&] { fs_check(f, phase, "rdgpu_flat_shard_inject"); fs_inject(f, phase, d_above, d_below); });
}
extern "C" int rdgpu_flat_shard_heights(rdgpu_flat_shard *f, int32_t *d_out) {
  return guarded([&] { fs_check(f, 0, "rdgpu_flat_shard_heights"); fs_heights(f, d_out); });
}
extern "C" int rdgpu_flat_graph_solve_dev(const int32_t *d_gathered, int world, int w, int32_t *d_out, void *stream) {
  return guarded([&] { flat_graph_solve_device(d_gathered, world, w, d_out, (hipStream_t)stream); });
}
extern "C" int rdgpu_flat_shard_finish(rdgpu_flat_shard *f, const int32_t *d_heights, uint8_t *d_dirs_out) {
  return guarded([&] {
    fs_check(f, 0, "rdgpu_flat_shard_finish");
    if (!d_dirs_out) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_shard_finish: null pointer");
    f->finish_fn(f, d_heights, d_dirs_out);
  });
}
extern "C" int rdgpu_flat_shard_rounds(const rdgpu_flat_shard *f, int phase) {
  return (f && (phase == 0 || phase == 1)) ? (int)f->rounds[phase] : -1;
}
extern "C" void rdgpu_flat_shard_free(rdgpu_flat_shard *f) { fs_free(f); }

extern "C" int rdgpu_flat_get_stats(rdgpu_flat_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = g_fstats;
  return RDGPU_OK;
}
extern "C" int rdgpu_flat_get_async_stats(rdgpu_flat_async_stats *out) {
  if (!out) return RDGPU_ERR_ARG;
  *out = rdgpu_flat_async_stats{g_async_info.visits, g_async_info.launches, g_async_info.failures, g_async_info.live_tiles, 0};
  return RDGPU_OK;
}
