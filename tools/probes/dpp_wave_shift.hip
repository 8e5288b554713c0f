#include <hip/hip_runtime.h>
__global__ void k(int *o) {
  int v = threadIdx.x;
  int l = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
  int r = __builtin_amdgcn_update_dpp(-2, v, 0x130, 0xf, 0xf, false);
  o[threadIdx.x * 2] = l; o[threadIdx.x * 2 + 1] = r;
}
#include <cstdio>
int main() {
  int *d; (void)hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  int h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; i++) { ok &= h[2 * i] == (i ? i - 1 : -1); ok &= h[2 * i + 1] == (i < 63 ? i + 1 : -2); }
  printf("dpp wave shift semantics %s (lane0 l=%d lane63 r=%d lane5 l=%d r=%d)\n", ok ? "OK" : "WRONG", h[0], h[127], h[10], h[11]);
  return 0;
}
