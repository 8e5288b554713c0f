#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float *rise_in, int method, double xparam, double *dbg, float *out) {
  constexpr double SQ2 = 1.414213562373095048801688724209698078569671875376948;
  float p[9];
  for (int i = 0; i < 9; i++) p[i] = -1.0f;
  double C = 0;
#pragma unroll
  for (int kk = 1; kk <= 8; kk++) {
    const float r = rise_in[kk];
    if (r > 0) {
      const double rise = r;
      const double run = (kk & 1) ? 1.0 : SQ2;
      const double grad = rise / run;
      if (method == 0) {
        const double v = pow(grad * ((kk & 1) ? 0.5 : 0.354), xparam);
        dbg[kk] = v;
        p[kk] = (float)v;
        C += p[kk];
      } else {
        const double cval = pow(grad, xparam);
        p[kk] = (float)cval;
        C += cval;
      }
    }
  }
  dbg[0] = C;
  if (C > 0) {
    p[0] = 0.0f;
    C = 1 / C;
    dbg[9] = C;
#pragma unroll
    for (int kk = 1; kk <= 8; kk++) { dbg[10 + kk] = p[kk]; p[kk] = p[kk] > 0 ? (float)(p[kk] * C) : 0.0f; }
  }
  for (int i = 0; i < 9; i++) out[i] = p[i];
}
int main() {
  float rise[9] = {0, 0.5291748046875f, 2.344970703125f, 1.9635009765625f, 1.555908203125f, 0, 0, 0, 0};
  float *dr, *dout; double *dd;
  (void)hipMalloc(&dr, 36); (void)hipMalloc(&dout, 36); (void)hipMalloc(&dd, 8 * 20);
  (void)hipMemcpy(dr, rise, 36, hipMemcpyHostToDevice);
  k<<<1, 1>>>(dr, 0, 2.0, dd, dout);
  float o[9]; double d[20];
  (void)hipMemcpy(o, dout, 36, hipMemcpyDeviceToHost); (void)hipMemcpy(d, dd, 160, hipMemcpyDeviceToHost);
  for (int i = 1; i <= 4; i++) printf("n=%d pow=%.17g pf=%.9g out=%.9g\n", i, d[i], d[10 + i], o[i]);
  printf("Csum=%.17g Cinv=%.17g hostinv=%.17g\n", d[0], d[9], 1.0 / d[0]);
  return 0;
}
