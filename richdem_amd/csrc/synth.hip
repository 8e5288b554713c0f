// synth.hip -- seeded fractal value-noise DEM generator (test / bench INPUT only).
// Bit-identical twin of richdem_amd/synth.py::fractal_dem:
// integer lattice hash, f32 arithmetic in a fixed order with explicitly rounded mul/add (no FMA).
#include "common.hpp"

namespace rdgpu {

__device__ __forceinline__ uint32_t hash24(uint32_t ix, uint32_t iy, uint32_t seed) {
  uint32_t h = ix * 0x9E3779B1u;
  h ^= iy * 0x85EBCA77u;
  h ^= seed * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h >> 8;
}

__global__ __launch_bounds__(256) void k_synth(float *z, int w, int h, uint32_t seed, int x0, int y0, float tilt) {
  const uint64_t n = (uint64_t)w * h;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  const float inv24 = 1.0f / 16777216.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int x = (int)(i % (uint64_t)w) + x0, y = (int)(i / (uint64_t)w) + y0;
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < 9; o++) {
      const int period = 512 >> o;
      const uint32_t ix = (uint32_t)(x / period), iy = (uint32_t)(y / period);
      const float tx = __fmul_rn((float)(x % period), 1.0f / (float)period);
      const float ty = __fmul_rn((float)(y % period), 1.0f / (float)period);
      const float sx = __fmul_rn(__fmul_rn(tx, tx), __fsub_rn(3.0f, __fmul_rn(2.0f, tx)));
      const float sy = __fmul_rn(__fmul_rn(ty, ty), __fsub_rn(3.0f, __fmul_rn(2.0f, ty)));
      const float v00 = __fmul_rn((float)hash24(ix, iy, seed + o), inv24);
      const float v10 = __fmul_rn((float)hash24(ix + 1, iy, seed + o), inv24);
      const float v01 = __fmul_rn((float)hash24(ix, iy + 1, seed + o), inv24);
      const float v11 = __fmul_rn((float)hash24(ix + 1, iy + 1, seed + o), inv24);
      const float a = __fadd_rn(v00, __fmul_rn(sx, __fsub_rn(v10, v00)));
      const float b = __fadd_rn(v01, __fmul_rn(sx, __fsub_rn(v11, v01)));
      const float v = __fadd_rn(a, __fmul_rn(sy, __fsub_rn(b, a)));
      acc = __fadd_rn(acc, __fmul_rn(v, 1.0f / (float)(1 << o)));
    }
    acc = __fmul_rn(acc, 1000.0f);
    if (tilt != 0.0f) acc = __fadd_rn(acc, __fmul_rn(tilt, (float)(x + y)));
    z[i] = acc;
  }
}

}  // namespace rdgpu

using namespace rdgpu;

extern "C" int rdgpu_synth_dem_dev_f32(float *d_dem, int w, int h, int seed, int x0, int y0, float tilt,
                                       void *stream) {
  return guarded([&] {
    if (!d_dem || w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_synth_dem: bad arguments");
    const uint64_t n = (uint64_t)w * h;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256u * 64u);
    RD_LAUNCH("synth.dem", k_synth, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_dem, w, h, (uint32_t)seed, x0,
              y0, tilt);
  });
}
