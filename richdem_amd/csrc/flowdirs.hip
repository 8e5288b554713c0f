// flowdirs.hip -- per-cell D8 flow directions, LDS-staged 3x3 stencil.
//
//  * rdgpu_d8_flowdirs_*      replaces d8_flow_directions / d8_FlowDir
//                             (reference include/richdem/flowmet/d8_flowdirs.hpp:96-123, :32-74)
//  * MODE_FM (internal)       the direction rule of FM_OCallaghan<D8> = FM_D8
//                             (flowmet/OCallaghan1984.hpp:13-77): used by FA_D8 (accum.hip) instead of
//                             materialising the 36 B/cell Array3D.
// Elevations are compared in their own type (no keys): the rule involves == ties, so NaN / -0.0 behave
// exactly as in the reference's C++ comparisons.  HBM-bound: 4 B read + 1 B written per cell.
#include "common.hpp"
#include "flowdirs.hpp"

namespace rdgpu {

constexpr int TW = 64, TH = 32, LW = TW + 2, LH = TH + 2, NTHR = 256;

// 64 x 32 tiles (+1 halo) staged in LDS with every load of a thread in flight together; a wavefront owns a band of 8
// consecutive rows, a lane one column, and the 3 x 3 window slides down the column in registers (3 LDS reads per cell).
template <class T, int MODE>
__global__ __launch_bounds__(NTHR) void k_flowdirs(const T *__restrict__ z, T nodata, uint8_t *__restrict__ dirs,
                                                   int w, int h, uint32_t tilesX, uint32_t ntiles) {
  __shared__ T sz[LH * LW];
  const uint32_t t = xcd_tile(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int x0 = (int)(t % tilesX) * TW, y0 = (int)(t / tilesX) * TH;
  if (window_inside(x0, y0, w, h, TW, TH, 1)) {
    stage_window_inside<T, TW, TH, 1, LW, NTHR>(z, w, x0, y0, sz);
  } else {
    constexpr int IPT = (LH * LW + NTHR - 1) / NTHR;
    T zv[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {   // clamped addresses: a cell outside the raster is never used as a neighbour
      const int i = min((int)threadIdx.x + r * NTHR, LH * LW - 1);
      const int ly = i / LW, lx = i - ly * LW;
      const int gx = min(max(x0 - 1 + lx, 0), w - 1), gy = min(max(y0 - 1 + ly, 0), h - 1);
      zv[r] = z[(size_t)gy * w + gx];
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) {
      const int i = (int)threadIdx.x + r * NTHR;
      if (i < LH * LW) sz[i] = zv[r];
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & (TW - 1), yb = (int)(threadIdx.x >> 6) * (TH / 4);
  const int gx = x0 + lx;
  T r0[3], r1[3], r2[3];
#pragma unroll
  for (int e = 0; e < 3; e++) { r0[e] = sz[yb * LW + lx + e]; r1[e] = sz[(yb + 1) * LW + lx + e]; }
#pragma unroll
  for (int j = 0; j < TH / 4; j++) {
    const int gy = y0 + yb + j;
#pragma unroll
    for (int e = 0; e < 3; e++) r2[e] = sz[(yb + j + 2) * LW + lx + e];
    // neighbours 1..8 in the 234/105/876 numbering (reference common/constants.hpp:44-45)
    const T nbv[9] = {r1[1], r1[0], r0[0], r0[1], r0[2], r1[2], r2[2], r2[1], r2[0]};
    const T e = r1[1];
    int dir = 0;  // NO_FLOW (constants.hpp:80)
    const bool edge = gx == 0 || gy == 0 || gx == w - 1 || gy == h - 1;
    if (e == nodata) {
      dir = 255;  // FLOWDIR_NO_DATA (d8_flowdirs.hpp:116-117) / NO_DATA_GEN cell (OCallaghan1984.hpp:37-40)
    } else if (MODE == MODE_D8) {
      if (edge) {  // d8_flowdirs.hpp:37-54: edge cells point off the grid
        if (gx == 0 && gy == 0) dir = 2;
        else if (gx == 0 && gy == h - 1) dir = 8;
        else if (gx == w - 1 && gy == 0) dir = 4;
        else if (gx == w - 1 && gy == h - 1) dir = 6;
        else if (gx == 0) dir = 1;
        else if (gx == w - 1) dir = 5;
        else if (gy == 0) dir = 3;
        else dir = 7;
      } else {  // d8_flowdirs.hpp:63-71
        // (take = v < m, or v == m while the choice so far is a DIAGONAL and n is a cardinal; "is a diagonal" is carried
        // as a flag -- a lane mask on the scalar unit -- instead of being decoded from dir at every cardinal)
        T m = e;
        bool diag = false;
#pragma unroll
        for (int n = 1; n <= 8; n++) {
          const T v = nbv[n];
          const bool take = (n & 1) ? ((v < m) | ((v == m) & diag)) : (v < m);
          m = take ? v : m;
          dir = take ? n : dir;
          diag = (n & 1) ? (diag & !take) : (diag | take);
        }
      }
    } else {  // MODE_FM, OCallaghan1984.hpp:42-74: edges never flow, NoData neighbours are skipped
      if (!edge) {
        T m = e;
#pragma unroll
        for (int n = 1; n <= 8; n++) {
          const T v = nbv[n];
          const bool take = !(v == nodata) & (v < m);   // first strictly-lowest neighbour below the centre
          m = take ? v : m;
          dir = take ? n : dir;
        }
      }
    }
    if (gx < w && gy < h) dirs[(size_t)gy * w + gx] = (uint8_t)dir;
#pragma unroll
    for (int e2 = 0; e2 < 3; e2++) { r0[e2] = r1[e2]; r1[e2] = r2[e2]; }
  }
}

template <class T>
void flowdirs_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, int mode, hipStream_t s) {
  if (!d_z || !d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flowdirs: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flowdirs: width and height must be positive");
  const uint32_t tilesX = (w + TW - 1) / TW, tilesY = (h + TH - 1) / TH, ntiles = tilesX * tilesY;
  if (mode == MODE_D8)
    RD_LAUNCH("flowdirs.d8", (k_flowdirs<T, MODE_D8>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, nodata, d_dirs,
              w, h, tilesX, ntiles);
  else
    RD_LAUNCH("flowdirs.fm_d8", (k_flowdirs<T, MODE_FM>), dim3(xcd_grid(ntiles)), dim3(NTHR), 0, s, d_z, nodata,
              d_dirs, w, h, tilesX, ntiles);
}

#define RD_INST(T) template void flowdirs_device<T>(const T *, T, int, int, uint8_t *, int, hipStream_t);
RD_INST(uint8_t) RD_INST(int16_t) RD_INST(uint16_t) RD_INST(int32_t) RD_INST(uint32_t) RD_INST(float) RD_INST(double)
RD_INST(int8_t) RD_INST(int64_t) RD_INST(uint64_t)
#undef RD_INST

template <class T>
static void flowdirs_host(const T *dem, T nodata, int w, int h, uint8_t *dirs) {
  if (!dem || !dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flowdirs: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flowdirs: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  uint8_t *dd = Workspace::get().buf<uint8_t>("host.dirs", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  flowdirs_device<T>(d, nodata, w, h, dd, MODE_D8, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dirs, dd, n, hipMemcpyDeviceToHost));
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FLOWDIRS_API(SUF, T)                                                                            \
  extern "C" int rdgpu_d8_flowdirs_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs) {            \
    return guarded([&] { flowdirs_host<T>(dem, nodata, w, h, dirs); });                                    \
  }                                                                                                        \
  extern "C" int rdgpu_d8_flowdirs_dev_##SUF(const T *d_dem, T nodata, int w, int h, uint8_t *d_dirs,      \
                                             void *stream) {                                               \
    return guarded([&] { flowdirs_device<T>(d_dem, nodata, w, h, d_dirs, MODE_D8, (hipStream_t)stream); }); \
  }
RD_FLOWDIRS_API(u8, uint8_t)
RD_FLOWDIRS_API(i16, int16_t)
RD_FLOWDIRS_API(u16, uint16_t)
RD_FLOWDIRS_API(i32, int32_t)
RD_FLOWDIRS_API(u32, uint32_t)
RD_FLOWDIRS_API(f32, float)
RD_FLOWDIRS_API(f64, double)
RD_FLOWDIRS_API(i8, int8_t)
RD_FLOWDIRS_API(i64, int64_t)
RD_FLOWDIRS_API(u64, uint64_t)
