// mfd.hip -- D-infinity flow directions / proportions and the generic (multiple-receiver) flow accumulation.
//
//  * rdgpu_dinf_flowdirs_*        replaces dinf_flow_directions / dinf_FlowDir
//                                 (reference include/richdem/flowmet/dinf_flowdirs.hpp:128-152, :45-115)
//  * rdgpu_fm_tarboton_*          replaces FM_Tarboton = FM_Dinfinity (flowmet/Tarboton1997.hpp:14-144):
//                                 the 9-float-per-cell proportions array (common/Array3D.hpp:203-206)
//  * rdgpu_flow_accumulation_f64  replaces FlowAccumulation(const Array3D<float>&, Array2D<double>&)
//                                 (methods/flow_accumulation_generic.hpp:33-100) for any proportions array
//  * rdgpu_fa_tarboton_*          replaces FA_Tarboton / FA_Dinfinity (methods/flow_accumulation.hpp:16-17)
//                                 without materialising the 36 B/cell array: 5 B/cell (receiver, share)
//
// The slope/angle arithmetic is done in double with atan2/sqrt exactly as written in the reference
// (-ffp-contract=off: no FMA contraction); device libm may differ from glibc in the last ulp of atan2, so
// these outputs are compared with a tolerance (north_star: <= 1 ULP in f32), not bit for bit.
//
// Accumulation: "last arriver continues" generalised to several receivers.  A completed cell pushes
// share*total to each receiver with a returning device-scope atomic add and decrements the receiver's
// pending-donor count; it continues inline with the first receiver it completed and flags the others;
// flagged cells are compacted into the next round's work list.  f64 sums in a different order than the
// reference's FIFO: exact for integer-valued flows, f64 rounding otherwise.
#include "common.hpp"

#include <algorithm>

namespace rdgpu {

constexpr int NTHR = 256;
static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

__device__ __forceinline__ int mdx(int n) { return (n == 1 || n == 2 || n == 8) ? -1 : (n >= 4 && n <= 6) ? 1 : 0; }
__device__ __forceinline__ int mdy(int n) { return (n >= 2 && n <= 4) ? -1 : (n >= 6 && n <= 8) ? 1 : 0; }

// ------------------------------------------------------------------------------------------
// D-infinity per cell, on the 3 x 3 window of elevations (rows y-1, y, y+1; columns x-1, x, x+1).
// Both kernels are LDS-tiled stencils (r04: 64 x 32 tiles + halo, the window slides down a column in registers, as
// flowdirs.hip does): one thread per cell with sixteen neighbour loads and their 64-bit addresses was 36 ms at 40000^2.
// The reference takes atan2 of every facet and then branches on the angle.  Which branch it takes follows from the two
// slopes alone except within rounding of the thresholds (s1, s2 are differences of elevations, never -0.0):
// r < 0 <=> s2 < 0; r > atan2(1, 1) <=> s1 <= 0 or s2 > s1.  So the angle is computed ONCE, for the facet that wins (the
// steepest-facet comparison only needs s), and per facet only where s2 / s1 is within a margin of a threshold.  Results are
// the reference's bit for bit (tests/test_s2_dinf_gpu.py: every band of the 10000^2 raster).
// ------------------------------------------------------------------------------------------
constexpr int DTW = 64, DTH = 32, DLW_ = DTW + 2, DLH_ = DTH + 2;

template <class T>
__device__ __forceinline__ void dinf_stage(const T *__restrict__ z, int w, int h, int x0, int y0, T *sz) {
  constexpr int IPT = (DLH_ * DLW_ + NTHR - 1) / NTHR;
  T zv[IPT];
#pragma unroll
  for (int r = 0; r < IPT; r++) {   // clamped addresses: a cell outside the raster is never used as a neighbour
    const int i = min((int)threadIdx.x + r * NTHR, DLH_ * DLW_ - 1);
    const int ly = i / DLW_, lx = i - ly * DLW_;
    const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
    zv[r] = z[(size_t)gy * w + gx];
  }
#pragma unroll
  for (int r = 0; r < IPT; r++) {
    const int i = (int)threadIdx.x + r * NTHR;
    if (i < DLH_ * DLW_) sz[i] = zv[r];
  }
}

// dinf_FlowDir, flowmet/dinf_flowdirs.hpp:45-115 (facet tables :21-27), interior data cell; win[r][c]
template <class T>
__device__ __forceinline__ float dinf_cell(const T (&win)[3][3]) {
  constexpr int dy_e1[8] = {0, -1, -1, 0, 0, 1, 1, 0}, dx_e1[8] = {1, 0, 0, -1, -1, 0, 0, 1};
  constexpr int dy_e2[8] = {-1, -1, -1, -1, 1, 1, 1, 1}, dx_e2[8] = {1, 1, -1, -1, -1, -1, 1, 1};
  constexpr double ac[8] = {0., 1., 1., 2., 2., 3., 3., 4.}, af[8] = {1., -1., 1., -1., 1., -1., 1., -1.};
  int nmax = -1;
  double smax = 0, rmax = 0, w1 = 0, w2 = 0;   // w1, w2: the winning facet's slopes when its angle is still owed
  bool owed = false;
  const double e0 = (double)win[1][1];
  const double quarter = atan2(1.0, 1.0);
#pragma unroll
  for (int k = 0; k < 8; k++) {                                      // :68-96
    const double e1 = (double)win[1 + dy_e1[k]][1 + dx_e1[k]];
    const double e2 = (double)win[1 + dy_e2[k]][1 + dx_e2[k]];
    const double s1 = (e0 - e1) / 1.0, s2 = (e1 - e2) / 1.0;
    int branch;   // 0: r < 0; 1: r > quarter; 2: in between (r = atan2(s2, s1))
    if (s2 < 0) branch = 0;
    else if (s1 <= 0) branch = (s1 == 0 && s2 == 0) ? 2 : 1;        // atan2(+0, +0) = 0; otherwise r >= pi / 2
    else {
      const double lo = s1 * (1.0 - 0x1p-40), hi = s1 * (1.0 + 0x1p-40);
      if (s2 > hi) branch = 1;
      else if (s2 < lo) branch = 2;
      else branch = atan2(s2, s1) > quarter ? 1 : 2;                 // within rounding of the threshold: as the reference does
    }
    double s;
    if (branch == 0) s = s1;
    else if (branch == 1) s = (e0 - e2) / sqrt(2.0);
    else s = sqrt(s1 * s1 + s2 * s2);
    if (s > smax) {
      smax = s; nmax = k;
      rmax = branch == 1 ? quarter : 0.0;
      owed = branch == 2;
      w1 = s1; w2 = s2;
    }
  }
  if (owed) rmax = atan2(w2, w1);
  double rg = 0;                                                     // NO_FLOW
  if (nmax != -1) rg = af[nmax] * rmax + ac[nmax] * M_PI / 2;
  return (float)rg;
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_dinf_dirs(const T *__restrict__ z, T nodata, float *__restrict__ out, int w,
                                                    int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[DLH_ * DLW_];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * DTW, y0 = (int)(t / tilesX) * DTH;
  dinf_stage<T>(z, w, h, x0, y0, sz);
  __syncthreads();
  const int lx = threadIdx.x & (DTW - 1), yb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (DTH / 4);
  const int gx = x0 + lx;
  T win[3][3];
#pragma unroll
  for (int e = 0; e < 3; e++) { win[0][e] = sz[yb * DLW_ + lx + e]; win[1][e] = sz[(yb + 1) * DLW_ + lx + e]; }
#pragma unroll 2
  for (int j = 0; j < DTH / 4; j++) {
    const int gy = y0 + yb + j;
#pragma unroll
    for (int e = 0; e < 3; e++) win[2][e] = sz[(yb + j + 2) * DLW_ + lx + e];
    float a;
    if (win[1][1] == nodata) a = -1.0f;                                  // dinf_NO_DATA, :147-148
    else if (gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1) {         // :46-63
      double e;
      if (gx == 0 && gy == 0) e = 3 * M_PI / 4;
      else if (gx == 0 && gy == h - 1) e = 5 * M_PI / 4;
      else if (gx == w - 1 && gy == 0) e = 1 * M_PI / 4;
      else if (gx == w - 1 && gy == h - 1) e = 7 * M_PI / 4;
      else if (gx == 0) e = 4 * M_PI / 4;
      else if (gx == w - 1) e = 0 * M_PI / 4;
      else if (gy == 0) e = 2 * M_PI / 4;
      else e = 6 * M_PI / 4;
      a = (float)e;
    } else a = dinf_cell<T>(win);
    if (gx < w && gy < h) out[(size_t)gy * w + gx] = a;
#pragma unroll
    for (int e = 0; e < 3; e++) { win[0][e] = win[1][e]; win[1][e] = win[2][e]; }
  }
}

// ------------------------------------------------------------------------------------------
// FM_Tarboton, flowmet/Tarboton1997.hpp:14-144 in compact form: rcv = first receiver n (0 none, 255
// NoData), share = proportion to neighbour n, share2 = proportion to neighbour nwrap(n + 1).  Interior data cell;
// win[r][c] as above.
// ------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void tarboton_cell(const T (&win)[3][3], T nodata, int &rcv, float &share, float &share2) {
  constexpr int dy_e1[9] = {0, 0, -1, -1, 0, 0, 1, 1, 0}, dx_e1[9] = {0, -1, 0, 0, 1, 1, 0, 0, -1};
  constexpr int dy_e2[9] = {0, -1, -1, -1, -1, 1, 1, 1, 1}, dx_e2[9] = {0, -1, -1, 1, 1, 1, 1, -1, -1};
  constexpr double af[9] = {0, -1., 1., -1., 1., -1., 1., -1., 1.};
  const float dang = (float)atan2(1.0, 1.0);
  rcv = 0;
  share = 0.0f;
  share2 = 0.0f;
  int nmax = -1;
  double smax = 0, w1 = 0, w2 = 0;
  float rmax = 0;
  bool owed = false;
  const double e0 = (double)win[1][1];
#pragma unroll
  for (int n = 1; n <= 8; n++) {                                       // :56-92
    const T v1 = win[1 + dy_e1[n]][1 + dx_e1[n]], v2 = win[1 + dy_e2[n]][1 + dx_e2[n]];
    if (v1 == nodata || v2 == nodata) continue;
    const double e1 = (double)v1, e2 = (double)v2;
    const double s1 = (e0 - e1) / 1.0, s2 = (e1 - e2) / 1.0;
    // which of the reference's three branches (:83-91) the angle r = atan2(s2, s1) falls into follows from the slopes except
    // within a margin of the two thresholds (there: atan2, as the reference); the angle itself is only needed for the
    // facet that wins
    int branch;   // 0: r < 1e-7; 1: r > dang - 1e-7; 2: in between
    if (s2 < 0) branch = 0;
    else if (s1 <= 0) branch = (s1 == 0 && s2 == 0) ? 0 : 1;            // atan2(+0, +0) = 0; otherwise r >= pi / 2
    else if (s2 < s1 * 0.99e-7) branch = 0;                             // tan(1e-7) = 1.0000000000000033e-7
    else if (s2 > s1) branch = 1;                                       // tan(dang - 1e-7) = 0.99999984...
    else if (s2 > s1 * 1.01e-7 && s2 < s1 * 0.9999995) branch = 2;
    else {
      const double r = atan2(s2, s1);
      branch = r < 1e-7 ? 0 : r > dang - 1e-7 ? 1 : 2;
    }
    double s;
    if (branch == 0) s = s1;
    else if (branch == 1) s = (e0 - e2) / sqrt(2.0);
    else s = sqrt(s1 * s1 + s2 * s2);
    if (s > smax) {
      smax = s; nmax = n;
      rmax = branch == 1 ? dang : 0.0f;
      owed = branch == 2;
      w1 = s1; w2 = s2;
    }
  }
  if (nmax == -1) return;
  if (owed) rmax = (float)atan2(w2, w1);
  if (af[nmax] == 1 && rmax == 0) rmax = dang;                          // :99-104
  else if (af[nmax] == 1 && rmax == dang) rmax = 0;
  else if (af[nmax] == 1) rmax = (float)(M_PI / 4 - rmax);
  const int nxt = nmax + 1 == 9 ? 1 : nmax + 1;
  if (rmax == 0) { rcv = nmax; share = 1.0f; }                          // :106-113
  else if (rmax == dang) { rcv = nxt; share = 1.0f; }
  else {
    rcv = nmax | 0x10;                            // 0x10: two receivers
    share = (float)(rmax / (M_PI / 4.));          // props(x,y,nmax), :111
    share2 = (float)(1 - rmax / (M_PI / 4.));     // props(x,y,nwrap(nmax+1)), :112
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_tarboton(const T *__restrict__ z, T nodata, uint8_t *__restrict__ rcv,
                                                   float *__restrict__ sh1, float *__restrict__ sh2, int w, int h,
                                                   uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[DLH_ * DLW_];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * DTW, y0 = (int)(t / tilesX) * DTH;
  dinf_stage<T>(z, w, h, x0, y0, sz);
  __syncthreads();
  const int lx = threadIdx.x & (DTW - 1), yb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (DTH / 4);
  const int gx = x0 + lx;
  T win[3][3];
#pragma unroll
  for (int e = 0; e < 3; e++) { win[0][e] = sz[yb * DLW_ + lx + e]; win[1][e] = sz[(yb + 1) * DLW_ + lx + e]; }
#pragma unroll 2
  for (int j = 0; j < DTH / 4; j++) {
    const int gy = y0 + yb + j;
#pragma unroll
    for (int e = 0; e < 3; e++) win[2][e] = sz[(yb + j + 2) * DLW_ + lx + e];
    int r = 0;
    float p = 0.0f, q = 0.0f;
    if (win[1][1] == nodata) r = 255;                                                     // :44-47
    else if (!(gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1)) tarboton_cell<T>(win, nodata, r, p, q);   // :49-50: edges never flow
    if (gx < w && gy < h) {
      const size_t c = (size_t)gy * w + gx;
      rcv[c] = (uint8_t)r;
      sh1[c] = p;
      sh2[c] = q;
    }
#pragma unroll
    for (int e = 0; e < 3; e++) { win[0][e] = win[1][e]; win[1][e] = win[2][e]; }
  }
}

__global__ __launch_bounds__(NTHR) void k_tarboton_props(const uint8_t *__restrict__ rcv, const float *__restrict__ sh1,
                                                         const float *__restrict__ sh2, float *__restrict__ props,
                                                         uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    float p[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};   // NO_FLOW_GEN, :25
    const int rr = rcv[c], r = rr & 0xF;
    if (rr == 255) p[0] = -2.0f;                         // NO_DATA_GEN, :45
    else if (r >= 1) {
      p[0] = 0.0f;                                       // HAS_FLOW_GEN, :97
      p[r] = sh1[c];
      if (rr & 0x10) p[r == 8 ? 1 : r + 1] = sh2[c];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) props[9 * c + k] = p[k];
  }
}

// ------------------------------------------------------------------------------------------
// proportions accessors
// ------------------------------------------------------------------------------------------
struct PropsAcc {   // the reference's Array3D<float>
  const float *props;
  __device__ __forceinline__ bool nodata(uint64_t c) const { return props[9 * c] == -2.0f; }
  __device__ __forceinline__ float share(uint64_t c, int n) const { return props[9 * c + n]; }
};
struct DinfAcc {    // compact D-infinity form
  const uint8_t *rcv;
  const float *sh1, *sh2;
  __device__ __forceinline__ bool nodata(uint64_t c) const { return rcv[c] == 255; }
  __device__ __forceinline__ float share(uint64_t c, int n) const {
    const int rr = rcv[c], r = rr & 0xF;
    if (rr == 255 || r < 1 || r > 8) return -1.0f;
    if (n == r) return sh1[c];
    if ((rr & 0x10) && n == (r == 8 ? 1 : r + 1)) return sh2[c];
    return -1.0f;
  }
};

constexpr uint32_t PEND_NODATA = 0xFFFFFFFFu;

// deps, flow_accumulation_generic.hpp:47-58 (donors are interior cells only), + the initial ready flags :61-64
template <class ACC>
__global__ __launch_bounds__(NTHR) void k_mfd_init(ACC a, uint32_t *pending, uint8_t *ready, int w, int h) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    uint32_t k = PEND_NODATA;
    uint8_t rdy = 0;
    if (!a.nodata(c)) {
      k = 0;
#pragma unroll
      for (int m = 1; m <= 8; m++) {
        const int dx = x + mdx(m), dy = y + mdy(m);
        if (dx < 1 || dy < 1 || dx >= w - 1 || dy >= h - 1) continue;   // donors: interior cells
        const uint64_t d = (uint64_t)dy * w + dx;
        if (a.nodata(d)) continue;
        if (a.share(d, m <= 4 ? m + 4 : m - 4) > 0) k++;
      }
      rdy = k == 0;
    }
    pending[c] = k;
    ready[c] = rdy;
  }
}

// One round: every thread takes one COMPLETED cell of the round's list and pushes its total to the receivers --
// all returning adds in flight together, one wait, then all counter decrements together: two memory round trips
// per cell however many receivers it has.  It continues inline with the first receiver it completed; the others go
// into the thread's LDS buffer, and at the end the block appends all buffered cells to the next round's list with
// ONE counter add (same-address atomics serialise at ~12 ns on this chip, and full-raster flag compaction cost
// 4 ms per round x hundreds of rounds: r01b FA_Tarboton 1.2 s at 40k x 40k).  A thread whose buffer could not
// take a worst case (8 completions) hands its current cell over to the next round unprocessed.
constexpr int MBUF = 20;
template <class ACC>
__global__ __launch_bounds__(NTHR) void k_mfd_round(ACC a, const uint32_t *__restrict__ list, uint32_t nlist,
                                                    uint32_t *pending, double *acc, uint32_t *next_list,
                                                    uint32_t *next_count, int w, int h) {
  __shared__ uint32_t buf[MBUF][NTHR];
  __shared__ uint32_t wtot[NTHR / 64];
  __shared__ uint32_t bbase;
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  uint32_t nb = 0;
  if (i < nlist) {
    uint32_t c = list[i];
    double v = acc[c];   // completed in an earlier launch (or a source): final
    for (;;) {
      const int x = (int)(c % (uint32_t)w), y = (int)(c / (uint32_t)w);
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) break;   // edge cells never pass flow on (FM_*: :49-50)
      if (nb + 8 > MBUF) { buf[nb++][threadIdx.x] = c; break; }
      float p[8];
      bool nd[8];
#pragma unroll
      for (int n = 1; n <= 8; n++) p[n - 1] = a.share(c, n);
#pragma unroll
      for (int n = 1; n <= 8; n++)   // flow_accumulation_generic.hpp:81-86
        nd[n - 1] = (p[n - 1] > 0) ? a.nodata((uint64_t)(y + mdy(n)) * w + (x + mdx(n))) : true;
      double prev = 0;
#pragma unroll
      for (int n = 1; n <= 8; n++)
        if (!nd[n - 1]) prev += atomicAdd(&acc[(uint64_t)(y + mdy(n)) * w + (x + mdx(n))], (double)p[n - 1] * v);   // :87
      // every add has returned (= was performed at the memory side) before any counter is decremented
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");
      uint32_t old[8];
#pragma unroll
      for (int n = 1; n <= 8; n++)
        if (!nd[n - 1])
          old[n - 1] = __hip_atomic_fetch_sub(&pending[(uint64_t)(y + mdy(n)) * w + (x + mdx(n))], 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
      uint32_t cont = 0xFFFFFFFFu;
#pragma unroll
      for (int n = 1; n <= 8; n++) {
        if (nd[n - 1] || old[n - 1] != 1) continue;
        const uint32_t r = (uint32_t)(y + mdy(n)) * (uint32_t)w + (uint32_t)(x + mdx(n));   // r is complete
        if (cont == 0xFFFFFFFFu) cont = r;          // continue inline with the first one,
        else buf[nb++][threadIdx.x] = r;            // the others are for the next round
      }
      if (cont == 0xFFFFFFFFu) break;
      c = cont;
      v = atomicAdd(&acc[c], 0.0);                  // final total, read at the memory side
    }
  }
  // block-wide exclusive prefix of nb, one counter add per block
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = nb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    bbase = tot ? atomicAdd(next_count, tot) : 0;
  }
  __syncthreads();
  uint32_t off = bbase + incl - nb;
  for (int k = 0; k < wv; k++) off += wtot[k];
  for (uint32_t k = 0; k < nb; k++) next_list[off + k] = buf[k][threadIdx.x];
}

template <class ACC>
__global__ __launch_bounds__(NTHR) void k_mfd_nodata(ACC a, double *acc, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride)
    if (a.nodata(c)) acc[c] = -1.0;                                     // ACCUM_NO_DATA, :95-97
}

// ---- ready flags -> work list (count / scan / fill; flags are cleared) ------------------------------
constexpr int CPB = 4096;
__global__ __launch_bounds__(NTHR) void k_rdy_count(const uint8_t *__restrict__ f, uint64_t n, uint32_t *counts) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t cnt = 0;
#pragma unroll 4
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t c = base + (uint64_t)j * NTHR + threadIdx.x;
    if (c < n && f[c]) cnt++;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_rdy_scan(uint32_t *counts, uint32_t m, uint32_t *total) {
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
__global__ __launch_bounds__(NTHR) void k_rdy_fill(uint8_t *f, uint64_t n, const uint32_t *__restrict__ offsets,
                                                   uint32_t *__restrict__ out) {
  __shared__ uint32_t ws[NTHR / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CPB;
  uint32_t run = offsets[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < CPB / NTHR; j++) {
    const uint64_t c = base + (uint64_t)j * NTHR + threadIdx.x;
    const bool hit = c < n && f[c];
    if (hit) f[c] = 0;
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

// The same step with the round's by-products consumed where they arise (r04f).  In k_mfd_round a cell completed "on the
// side" waits for the next launch, and a launch lasts as long as its longest inline chain: at S3 where every cell drains,
// 2614 launches of 2.8 ms.  Here a wavefront keeps a STACK of such cells in LDS: a lane whose chain has ended takes the
// next cell from it in the same trip of the loop, so a launch ends when everything reachable from its list is done --
// except what a full stack spills to the next launch's list.  All lanes of a wavefront run the loop together (ballots
// decide), the stack pointer is wave-uniform, LDS operations of one wavefront execute in order: no barriers.
constexpr int WCAP = 1536;   // entries of a wavefront's stack (4 x 6 KB of LDS per block)
constexpr uint32_t NOCELL = 0xFFFFFFFFu;
template <class ACC>
__global__ __launch_bounds__(NTHR) void k_mfd_stack(ACC a, const uint32_t *__restrict__ list, uint32_t nlist,
                                                    uint32_t *pending, double *acc, uint32_t *next_list,
                                                    uint32_t *next_count, int w, int h) {
  __shared__ uint32_t stack[NTHR / 64][WCAP];
  volatile uint32_t *const st = stack[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  uint32_t sp = 0;   // (wave-uniform)
  const uint32_t i = blockIdx.x * NTHR + threadIdx.x;
  uint32_t c = i < nlist ? list[i] : NOCELL;
  double v = c != NOCELL ? acc[c] : 0.0;   // completed in an earlier launch (or a source): final
  for (;;) {
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(c == NOCELL);
    if (idle == ~0ull && sp == 0) break;
    if (idle != 0ull && sp != 0) {   // idle lanes take the top of the stack
      const uint32_t k = min((uint32_t)__popcll(idle), sp);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (c == NOCELL && rank < k) {
        c = st[sp - 1u - rank];
        v = atomicAdd(&acc[c], 0.0);       // its final total, read at the memory side
      }
      sp -= k;
    }
    uint32_t oth[8];
    bool isoth[8];
#pragma unroll
    for (int n = 0; n < 8; n++) { oth[n] = 0; isoth[n] = false; }
    uint32_t no = 0;
    if (c != NOCELL) {
      const int x = (int)(c % (uint32_t)w), y = (int)(c / (uint32_t)w);
      uint32_t cont = NOCELL;
      if (!(x == 0 || y == 0 || x == w - 1 || y == h - 1)) {   // edge cells never pass flow on (FM_*: :49-50)
        float p[8];
        bool nd[8];
#pragma unroll
        for (int n = 1; n <= 8; n++) p[n - 1] = a.share(c, n);
#pragma unroll
        for (int n = 1; n <= 8; n++)   // flow_accumulation_generic.hpp:81-86
          nd[n - 1] = (p[n - 1] > 0) ? a.nodata((uint64_t)(y + mdy(n)) * w + (x + mdx(n))) : true;
        double prev = 0;
#pragma unroll
        for (int n = 1; n <= 8; n++)
          if (!nd[n - 1]) prev += atomicAdd(&acc[(uint64_t)(y + mdy(n)) * w + (x + mdx(n))], (double)p[n - 1] * v);   // :87
        // every add has returned (= was performed at the memory side) before any counter is decremented
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");
        uint32_t old[8];
#pragma unroll
        for (int n = 1; n <= 8; n++)
          if (!nd[n - 1])
            old[n - 1] = __hip_atomic_fetch_sub(&pending[(uint64_t)(y + mdy(n)) * w + (x + mdx(n))], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int n = 1; n <= 8; n++) {
          if (nd[n - 1] || old[n - 1] != 1) continue;
          const uint32_t r = (uint32_t)(y + mdy(n)) * (uint32_t)w + (uint32_t)(x + mdx(n));   // r is complete
          if (cont == NOCELL) cont = r;                       // continue inline with the first one,
          else { oth[n - 1] = r; isoth[n - 1] = true; no++; }  // the others go on the stack
        }
      }
      c = cont;
      if (c != NOCELL) v = atomicAdd(&acc[c], 0.0);           // final total, read at the memory side
    }
    if (__builtin_amdgcn_ballot_w64(no != 0u) != 0ull) {   // push the step's by-products: one prefix over the lanes
      uint32_t incl = no;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      uint32_t pos = incl - no;
      if (sp + total <= (uint32_t)WCAP) {
#pragma unroll
        for (int n = 0; n < 8; n++)
          if (isoth[n]) st[sp + pos++] = oth[n];
        sp += total;
      } else {   // the stack cannot take them: the next launch's list
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next_count, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int n = 0; n < 8; n++)
          if (isoth[n]) next_list[base + pos++] = oth[n];
      }
    }
  }
}

static uint32_t g_mfd_rounds = 0;

template <class ACC>
static void mfd_accumulate(ACC a, int w, int h, double *d_acc, hipStream_t s) {
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  uint32_t *pending = ws.buf<uint32_t>("mfd.pending", n);
  uint8_t *ready = ws.buf<uint8_t>("mfd.ready", n);
  uint32_t *list = ws.buf<uint32_t>("mfd.list", n);
  const uint32_t nblk = (uint32_t)((n + CPB - 1) / CPB);
  uint32_t *counts = ws.buf<uint32_t>("mfd.counts", (size_t)nblk + 1);
  RD_LAUNCH("mfd.init", (k_mfd_init<ACC>), dim3(sgrid(n)), dim3(NTHR), 0, s, a, pending, ready, w, h);
  g_mfd_rounds = 0;
  // the sources, once, by flag compaction; afterwards every round appends its successor list itself
  RD_LAUNCH("mfd.ready_count", k_rdy_count, dim3(nblk), dim3(NTHR), 0, s, (const uint8_t *)ready, n, counts);
  RD_LAUNCH("mfd.ready_scan", k_rdy_scan, dim3(1), dim3(1024), 0, s, counts, nblk, counts + nblk);
  RD_HIP(hipMemcpyAsync(hw, counts + nblk, 4, hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  uint32_t nl = hw[0];
  if (nl) RD_LAUNCH("mfd.ready_fill", k_rdy_fill, dim3(nblk), dim3(NTHR), 0, s, ready, n, (const uint32_t *)counts, list);
  uint32_t *list2 = ws.buf<uint32_t>("mfd.list2", n);
  uint32_t *ctr = counts + nblk;
  // Long lists go through k_mfd_round (their by-products are processed next launch, densely packed: a wavefront that
  // works off its own by-products runs at a few active lanes -- S3's filled DEM, 223 launches: 256 ms, one stacked launch
  // 492); once a list is short the launches are what costs (S3 where every cell drains: 2614 launches, 7.2 s) and
  // k_mfd_stack finishes everything that is left.  RDGPU_MFD_STACK_BELOW sets the list length of the switch (0: never).
  const char *se = getenv("RDGPU_MFD_STACK_BELOW");
  const uint32_t stack_below = se ? (uint32_t)strtoul(se, nullptr, 10) : (1u << 22);
  while (nl) {
    const bool stacked = nl < stack_below;
    RD_HIP(hipMemsetAsync(ctr, 0, sizeof(uint32_t), s));
    if (stacked)
      RD_LAUNCH("mfd.round", (k_mfd_stack<ACC>), dim3((nl + NTHR - 1) / NTHR), dim3(NTHR), 0, s, a, (const uint32_t *)list, nl,
                pending, d_acc, list2, ctr, w, h);
    else
      RD_LAUNCH("mfd.round", (k_mfd_round<ACC>), dim3((nl + NTHR - 1) / NTHR), dim3(NTHR), 0, s, a, (const uint32_t *)list, nl,
                pending, d_acc, list2, ctr, w, h);
    RD_HIP(hipMemcpyAsync(hw, ctr, 4, hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    nl = hw[0];
    std::swap(list, list2);
    if (++g_mfd_rounds > (1u << 26)) throw Error(RDGPU_ERR_HIP, "rdgpu flow accumulation did not terminate");
  }
  RD_LAUNCH("mfd.nodata", (k_mfd_nodata<ACC>), dim3(sgrid(n)), dim3(NTHR), 0, s, a, d_acc, n);
}

// ------------------------------------------------------------------------------------------
// FM_Holmgren (flowmet/Holmgren1994.hpp:14-88), FM_Quinn (= Holmgren, x = 1; Quinn1991.hpp:13-17),
// FM_Freeman (Freeman1991.hpp:14-86), FM_D4 = FM_OCallaghan<D4> (OCallaghan1984.hpp:13-77, :86) into the
// reference's 9-float proportions layout.  The arithmetic is written as in the reference (rise in the
// element type's promoted type, gradient and pow in double, Holmgren sums the STORED floats, Freeman the
// doubles, the normalisation multiplies float by double); pow comes from the device libm, so proportions
// are specified to <= 1 ULP (f32) like D-infinity.  FM_D4 keeps the reference's slot numbering: it
// stores the chosen D4 neighbour n = 1..4 (left, up, right, down) in slot n, and FlowAccumulation then reads
// slot n with the D8 offsets -- reproduced as is, because that is what FA_D4 returns.
// ------------------------------------------------------------------------------------------
// pow() as the reference's libm returns it, to the extent the f32 proportions can see: glibc's pow is exact
// whenever x^y is representable, and with short-mantissa gradients (differences of float elevations) and an
// integer exponent, x^y often IS representable and lands exactly on a float rounding tie -- where the device
// libm's last-bit error would flip the stored float.  Integer exponents up to 64 are therefore evaluated by
// binary powering in double-double (one final rounding); everything else goes to the device pow (<= 1 ulp).
struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
  const double p = a.hi * b.hi;
  double e = __builtin_fma(a.hi, b.hi, -p);
  e += a.hi * b.lo + a.lo * b.hi;
  const double hi = p + e;
  return dd{hi, e - (hi - p)};
}
__device__ __forceinline__ double pow_ref(double x, double y) {
  if (y == 1.0) return x;
  if (y >= 2.0 && y <= 64.0 && y == (double)(int)y) {
    int n = (int)y;
    dd r{1.0, 0.0}, b{x, 0.0};
    for (;;) {
      if (n & 1) r = dd_mul(r, b);
      n >>= 1;
      if (!n) break;
      b = dd_mul(b, b);
    }
    return r.hi;
  }
  return pow(x, y);
}

enum { MFD_HOLMGREN = 0, MFD_FREEMAN = 1, MFD_QUINN = 2, MFD_D4 = 3 };

template <class T>
__global__ __launch_bounds__(NTHR) void k_fm_mfd(const T *__restrict__ z, T nodata, float *__restrict__ props, int w, int h,
                                                 int method, double xparam) {
  constexpr double SQ2 = 1.414213562373095048801688724209698078569671875376948;
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t c = (uint64_t)blockIdx.x * NTHR + threadIdx.x; c < n; c += stride) {
    const int x = (int)(c % (uint64_t)w), y = (int)(c / (uint64_t)w);
    float p[9];
#pragma unroll
    for (int k = 0; k < 9; k++) p[k] = -1.0f;
    const T e = z[c];
    if (e == nodata) {
      p[0] = -2.0f;
    } else if (x > 0 && y > 0 && x < w - 1 && y < h - 1) {
      if (method == MFD_D4) {
        int lowest_n = 0;
        T lowest = T();
#pragma unroll
        for (int k = 1; k <= 4; k++) {
          const int dx = k == 1 ? -1 : k == 3 ? 1 : 0, dy = k == 2 ? -1 : k == 4 ? 1 : 0;   // constants.hpp:54-55
          const T ne = z[(size_t)(y + dy) * w + (x + dx)];
          if (ne == nodata || ne >= e) continue;
          if (lowest_n == 0 || ne < lowest) { lowest = ne; lowest_n = k; }
        }
        if (lowest_n) { p[0] = 0.0f; p[lowest_n] = 1.0f; }
      } else {
        double C = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
          const T ne = z[(size_t)(y + mdy(k)) * w + (x + mdx(k))];
          if (ne == nodata) continue;
          if (ne < e) {
            const double rise = e - ne;
            const double run = (k & 1) ? 1.0 : SQ2;
            const double grad = rise / run;
            if (method == MFD_HOLMGREN) {
              p[k] = (float)pow_ref(grad * ((k & 1) ? 0.5 : 0.354), xparam);
              C += p[k];
            } else {
              const double cval = pow_ref(grad, xparam);
              p[k] = (float)cval;
              C += cval;
            }
          }
        }
        if (C > 0) {
          p[0] = 0.0f;
          C = 1 / C;
#pragma unroll
          for (int k = 1; k <= 8; k++) p[k] = p[k] > 0 ? (float)(p[k] * C) : 0.0f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) props[c * 9 + k] = p[k];
  }
}

template <class T>
static void fm_mfd_device(const T *d_z, T nodata, int w, int h, int method, double xparam, float *d_props, hipStream_t s) {
  if (method < 0 || method > 3) throw Error(RDGPU_ERR_ARG, "rdgpu_fm_mfd: method must be 0 (Holmgren), 1 (Freeman), 2 (Quinn) or 3 (D4)");
  if (method == MFD_QUINN) { method = MFD_HOLMGREN; xparam = 1.0; }
  RD_LAUNCH("mfd.fm_mfd", (k_fm_mfd<T>), dim3(sgrid((uint64_t)w * h)), dim3(NTHR), 0, s, d_z, nodata, d_props, w, h, method, xparam);
}

static void check_dims(int w, int h, const char *who) {
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, std::string(who) + ": width and height must be positive");
  if ((uint64_t)w * (uint64_t)h > 0xFFFF0000ull) throw Error(RDGPU_ERR_ARG, std::string(who) + ": raster too large");
}

template <class T>
static DinfAcc tarboton_device(const T *d_z, T nodata, int w, int h, hipStream_t s) {
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  uint8_t *rcv = ws.buf<uint8_t>("mfd.rcv", n);
  float *sh1 = ws.buf<float>("mfd.sh1", n), *sh2 = ws.buf<float>("mfd.sh2", n);
  {
    const uint32_t dtilesX = (w + DTW - 1) / DTW, dntiles = dtilesX * ((h + DTH - 1) / DTH);
    RD_LAUNCH("mfd.tarboton", (k_tarboton<T>), dim3(xcd_grid(dntiles)), dim3(NTHR), 0, s, d_z, nodata, rcv, sh1, sh2, w, h, dtilesX, dntiles);
  }
  return DinfAcc{rcv, sh1, sh2};
}

template <class T>
static void host_dem(const T *dem, int w, int h, T **d) {
  const size_t n = (size_t)w * h;
  *d = Workspace::get().buf<T>("host.dem", n);
  RD_HIP(hipMemcpy(*d, dem, n * sizeof(T), hipMemcpyHostToDevice));
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_MFD_API(SUF, T)                                                                                      \
  extern "C" int rdgpu_dinf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int w, int h, float *d_out, void *st) { \
    return guarded([&] {                                                                                        \
      if (!d_dem || !d_out) throw Error(RDGPU_ERR_ARG, "rdgpu_dinf_flowdirs: null pointer");                    \
      check_dims(w, h, "rdgpu_dinf_flowdirs");                                                                  \
      const uint32_t dtx_ = (w + DTW - 1) / DTW, dnt_ = dtx_ * ((h + DTH - 1) / DTH);                               \
      RD_LAUNCH("mfd.dinf_dirs", (k_dinf_dirs<T>), dim3(xcd_grid(dnt_)), dim3(NTHR), 0, (hipStream_t)st,            \
                d_dem, nodata, d_out, w, h, dtx_, dnt_);                                                        \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_dinf_flowdirs_##SUF(const T *dem, T nodata, int w, int h, float *out) {                  \
    return guarded([&] {                                                                                        \
      if (!dem || !out) throw Error(RDGPU_ERR_ARG, "rdgpu_dinf_flowdirs: null pointer");                        \
      check_dims(w, h, "rdgpu_dinf_flowdirs");                                                                  \
      T *d;                                                                                                     \
      host_dem<T>(dem, w, h, &d);                                                                               \
      float *o = Workspace::get().buf<float>("host.f32out", (size_t)w * h);                                     \
      const uint32_t dtx_ = (w + DTW - 1) / DTW, dnt_ = dtx_ * ((h + DTH - 1) / DTH);                               \
      RD_LAUNCH("mfd.dinf_dirs", (k_dinf_dirs<T>), dim3(xcd_grid(dnt_)), dim3(NTHR), 0, (hipStream_t) nullptr,      \
                (const T *)d, nodata, o, w, h, dtx_, dnt_);                                                     \
      RD_HIP(hipMemcpy(out, o, (size_t)w * h * 4, hipMemcpyDeviceToHost));                                      \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fm_tarboton_##SUF(const T *dem, T nodata, int w, int h, float *props9) {                 \
    return guarded([&] {                                                                                        \
      if (!dem || !props9) throw Error(RDGPU_ERR_ARG, "rdgpu_fm_tarboton: null pointer");                       \
      check_dims(w, h, "rdgpu_fm_tarboton");                                                                    \
      T *d;                                                                                                     \
      host_dem<T>(dem, w, h, &d);                                                                               \
      const DinfAcc a = tarboton_device<T>(d, nodata, w, h, nullptr);                                           \
      const uint64_t n = (uint64_t)w * h;                                                                       \
      float *p = Workspace::get().buf<float>("host.props", n * 9);                                              \
      RD_LAUNCH("mfd.tarboton_props", k_tarboton_props, dim3(sgrid(n)), dim3(NTHR), 0, (hipStream_t) nullptr, a.rcv, a.sh1, \
                a.sh2, p, n);                                                                                   \
      RD_HIP(hipMemcpy(props9, p, n * 36, hipMemcpyDeviceToHost));                                              \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fa_tarboton_dev_##SUF(const T *d_dem, T nodata, int w, int h, double *d_accum, void *st) { \
    return guarded([&] {                                                                                        \
      if (!d_dem || !d_accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_tarboton: null pointer");                    \
      check_dims(w, h, "rdgpu_fa_tarboton");                                                                    \
      const DinfAcc a = tarboton_device<T>(d_dem, nodata, w, h, (hipStream_t)st);                               \
      mfd_accumulate<DinfAcc>(a, w, h, d_accum, (hipStream_t)st);                                               \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fa_tarboton_##SUF(const T *dem, T nodata, int w, int h, double *accum) {                 \
    return guarded([&] {                                                                                        \
      if (!dem || !accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_tarboton: null pointer");                        \
      check_dims(w, h, "rdgpu_fa_tarboton");                                                                    \
      T *d;                                                                                                     \
      host_dem<T>(dem, w, h, &d);                                                                               \
      const size_t n = (size_t)w * h;                                                                           \
      double *da = Workspace::get().buf<double>("host.area", n);                                                \
      RD_HIP(hipMemcpy(da, accum, n * 8, hipMemcpyHostToDevice));                                               \
      const DinfAcc a = tarboton_device<T>(d, nodata, w, h, nullptr);                                           \
      mfd_accumulate<DinfAcc>(a, w, h, da, nullptr);                                                            \
      RD_HIP(hipStreamSynchronize(nullptr));                                                                    \
      RD_HIP(hipMemcpy(accum, da, n * 8, hipMemcpyDeviceToHost));                                               \
    });                                                                                                         \
  }
RD_MFD_API(u8, uint8_t)
RD_MFD_API(i16, int16_t)
RD_MFD_API(u16, uint16_t)
RD_MFD_API(i32, int32_t)
RD_MFD_API(u32, uint32_t)
RD_MFD_API(f32, float)
RD_MFD_API(f64, double)
RD_MFD_API(i8, int8_t)

#define RD_MFD2_API(SUF, T)                                                                                     \
  extern "C" int rdgpu_fm_mfd_dev_##SUF(const T *d_dem, T nodata, int w, int h, int method, double xparam,      \
                                        float *d_props9, void *st) {                                            \
    return guarded([&] {                                                                                        \
      if (!d_dem || !d_props9) throw Error(RDGPU_ERR_ARG, "rdgpu_fm_mfd: null pointer");                        \
      check_dims(w, h, "rdgpu_fm_mfd");                                                                         \
      fm_mfd_device<T>(d_dem, nodata, w, h, method, xparam, d_props9, (hipStream_t)st);                         \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fm_mfd_##SUF(const T *dem, T nodata, int w, int h, int method, double xparam,            \
                                    float *props9) {                                                            \
    return guarded([&] {                                                                                        \
      if (!dem || !props9) throw Error(RDGPU_ERR_ARG, "rdgpu_fm_mfd: null pointer");                            \
      check_dims(w, h, "rdgpu_fm_mfd");                                                                         \
      T *d;                                                                                                     \
      host_dem<T>(dem, w, h, &d);                                                                               \
      const uint64_t n = (uint64_t)w * h;                                                                       \
      float *p = Workspace::get().buf<float>("host.props", n * 9);                                              \
      fm_mfd_device<T>(d, nodata, w, h, method, xparam, p, nullptr);                                            \
      RD_HIP(hipMemcpy(props9, p, n * 36, hipMemcpyDeviceToHost));                                              \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fa_mfd_dev_##SUF(const T *d_dem, T nodata, int w, int h, int method, double xparam,      \
                                        double *d_accum, void *st) {                                            \
    return guarded([&] {                                                                                        \
      if (!d_dem || !d_accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_mfd: null pointer");                         \
      check_dims(w, h, "rdgpu_fa_mfd");                                                                         \
      float *p = Workspace::get().buf<float>("mfd.props", (uint64_t)w * h * 9);                                 \
      fm_mfd_device<T>(d_dem, nodata, w, h, method, xparam, p, (hipStream_t)st);                                \
      mfd_accumulate<PropsAcc>(PropsAcc{p}, w, h, d_accum, (hipStream_t)st);                                    \
    });                                                                                                         \
  }                                                                                                             \
  extern "C" int rdgpu_fa_mfd_##SUF(const T *dem, T nodata, int w, int h, int method, double xparam,            \
                                    double *accum) {                                                            \
    return guarded([&] {                                                                                        \
      if (!dem || !accum) throw Error(RDGPU_ERR_ARG, "rdgpu_fa_mfd: null pointer");                             \
      check_dims(w, h, "rdgpu_fa_mfd");                                                                         \
      T *d;                                                                                                     \
      host_dem<T>(dem, w, h, &d);                                                                               \
      const size_t n = (size_t)w * h;                                                                           \
      double *da = Workspace::get().buf<double>("host.area", n);                                                \
      RD_HIP(hipMemcpy(da, accum, n * 8, hipMemcpyHostToDevice));                                               \
      float *p = Workspace::get().buf<float>("mfd.props", n * 9);                                               \
      fm_mfd_device<T>(d, nodata, w, h, method, xparam, p, nullptr);                                            \
      mfd_accumulate<PropsAcc>(PropsAcc{p}, w, h, da, nullptr);                                                 \
      RD_HIP(hipStreamSynchronize(nullptr));                                                                    \
      RD_HIP(hipMemcpy(accum, da, n * 8, hipMemcpyDeviceToHost));                                               \
    });                                                                                                         \
  }
RD_MFD2_API(u8, uint8_t)
RD_MFD2_API(i16, int16_t)
RD_MFD2_API(u16, uint16_t)
RD_MFD2_API(i32, int32_t)
RD_MFD2_API(u32, uint32_t)
RD_MFD2_API(f32, float)
RD_MFD2_API(f64, double)
RD_MFD2_API(i8, int8_t)

// FlowAccumulation(const Array3D<float>&, Array2D<double>&), methods/flow_accumulation_generic.hpp:33-100
extern "C" int rdgpu_flow_accumulation_dev_f64(const float *d_props9, int w, int h, double *d_accum, void *st) {
  return guarded([&] {
    if (!d_props9 || !d_accum) throw Error(RDGPU_ERR_ARG, "rdgpu_flow_accumulation: null pointer");
    check_dims(w, h, "rdgpu_flow_accumulation");
    mfd_accumulate<PropsAcc>(PropsAcc{d_props9}, w, h, d_accum, (hipStream_t)st);
  });
}
extern "C" int rdgpu_flow_accumulation_f64(const float *props9, int w, int h, double *accum) {
  return guarded([&] {
    if (!props9 || !accum) throw Error(RDGPU_ERR_ARG, "rdgpu_flow_accumulation: null pointer");
    check_dims(w, h, "rdgpu_flow_accumulation");
    const size_t n = (size_t)w * h;
    float *dp = Workspace::get().buf<float>("host.props", n * 9);
    double *da = Workspace::get().buf<double>("host.area", n);
    RD_HIP(hipMemcpy(dp, props9, n * 36, hipMemcpyHostToDevice));
    RD_HIP(hipMemcpy(da, accum, n * 8, hipMemcpyHostToDevice));
    mfd_accumulate<PropsAcc>(PropsAcc{dp}, w, h, da, nullptr);
    RD_HIP(hipStreamSynchronize(nullptr));
    RD_HIP(hipMemcpy(accum, da, n * 8, hipMemcpyDeviceToHost));
  });
}
extern "C" int rdgpu_flow_accumulation_rounds(uint32_t *rounds) {
  if (!rounds) return RDGPU_ERR_ARG;
  *rounds = g_mfd_rounds;
  return RDGPU_OK;
}
