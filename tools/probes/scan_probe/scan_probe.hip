// hipcc --offload-arch=gfx950 -O3 scan_probe.hip -o scan_probe && gpurun -- tools/probes/scan_probe/scan_probe
// checks the DPP prefix / suffix minimum over 64 lanes used by the open-water chamfer sweeps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int32_t wave_prefix_min(int32_t v) {
  constexpr int32_t I = INT32_MAX;
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ int32_t wave_suffix_min(int32_t v, int lane) {
  constexpr int32_t I = INT32_MAX;
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x101, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x102, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x104, 0xf, 0xf, false));
  v = imin(v, __builtin_amdgcn_update_dpp(I, v, 0x108, 0xf, 0xf, false));
  const int32_t r3 = __builtin_amdgcn_readlane(v, 48), r2 = imin(__builtin_amdgcn_readlane(v, 32), r3),
                r1 = imin(__builtin_amdgcn_readlane(v, 16), r2);
  return imin(v, lane < 16 ? r1 : lane < 32 ? r2 : lane < 48 ? r3 : I);
}
__global__ void k(const int32_t *in, int32_t *pre, int32_t *suf) {
  const int lane = threadIdx.x;
  pre[lane] = wave_prefix_min(in[lane]);
  suf[lane] = wave_suffix_min(in[lane], lane);
}
int main() {
  int32_t h[64], p[64], s[64], *di, *dp, *ds;
  hipMalloc(&di, 256); hipMalloc(&dp, 256); hipMalloc(&ds, 256);
  int bad = 0;
  for (int trial = 0; trial < 200; trial++) {
    for (int i = 0; i < 64; i++) h[i] = rand() % 1000 - (trial % 3 == 0 ? i : 0);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dp, ds);
    hipMemcpy(p, dp, 256, hipMemcpyDeviceToHost); hipMemcpy(s, ds, 256, hipMemcpyDeviceToHost);
    int m = INT32_MAX;
    for (int i = 0; i < 64; i++) { m = h[i] < m ? h[i] : m; if (p[i] != m) { if (bad < 5) printf("prefix trial %d lane %d got %d want %d\n", trial, i, p[i], m); bad++; } }
    m = INT32_MAX;
    for (int i = 63; i >= 0; i--) { m = h[i] < m ? h[i] : m; if (s[i] != m) { if (bad < 10) printf("suffix trial %d lane %d got %d want %d\n", trial, i, s[i], m); bad++; } }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
