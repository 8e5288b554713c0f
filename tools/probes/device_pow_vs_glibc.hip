#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const double *x, double y, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = pow(x[i], y);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(n);
  uint64_t s = 12345;
  for (int i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; x[i] = (i & 1) ? 0.9 + (double)(s >> 11) / 9007199254740992.0 * 0.2 : (double)(float)(0.5 + (double)(s >> 11) / 9007199254740992.0); }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  const double ys[] = {2.0, 4.0, 8.0, 3.0, 0.5, 1.1, 1.0, 2.5};
  for (double y : ys) {
    k<<<n / 256, 256>>>(dx, y, dout, n);
    hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    long long maxd = 0, cnt = 0, cnt32 = 0;
    for (int i = 0; i < n; i++) {
      double h = pow(x[i], y);
      int64_t a, b; memcpy(&a, &h, 8); memcpy(&b, &o[i], 8);
      long long d = llabs(a - b);
      if (d) cnt++;
      if (d > maxd) maxd = d;
      if ((float)h != (float)o[i]) cnt32++;
    }
    if (y == 2.0) { double t = 0.98175048828125; hipMemcpy(dx, &t, 8, hipMemcpyHostToDevice); k<<<1, 1>>>(dx, y, dout, 1); double r; hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost); printf("pow(0.98175048828125,2)= %.17g host %.17g\n", r, pow(t, 2.0)); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);}
    printf("y=%g: differing %lld of %d, max ulp(double) %lld, f32-cast differing %lld\n", y, cnt, n, maxd, cnt32);
  }
  return 0;
}
