// What the host boundary can reach: H2D / D2H rates of a pageable and of a registered 6.4 GB host buffer (the 40000^2 float32
// DEM of rdgpu_fill_f32), the cost of hipHostRegister itself, and both directions at once.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/pcie_probe.hip -o tools/probes/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 40000ull * 40000ull * 4ull;
  char *h = (char *)malloc(n), *h2 = (char *)malloc(n);
  memset(h, 1, n); memset(h2, 2, n);
  char *d, *d2;
  CK(hipMalloc(&d, n)); CK(hipMalloc(&d2, n));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int rep = 0; rep < 2; rep++) {
    double t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double a = now() - t;
    t = now(); CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); double b = now() - t;
    printf("pageable   H2D %.1f ms = %.1f GB/s   D2H %.1f ms = %.1f GB/s\n", a * 1e3, n / a / 1e9, b * 1e3, n / b / 1e9);
  }
  double t = now(); CK(hipHostRegister(h, n, hipHostRegisterDefault)); double reg = now() - t;
  t = now(); CK(hipHostRegister(h2, n, hipHostRegisterDefault)); double reg2 = now() - t;
  printf("hipHostRegister of %.2f GB: %.1f ms (second buffer %.1f ms)\n", n / 1e9, reg * 1e3, reg2 * 1e3);
  for (int rep = 0; rep < 2; rep++) {
    t = now(); CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double a = now() - t;
    t = now(); CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double b = now() - t;
    printf("registered H2D %.1f ms = %.1f GB/s   D2H %.1f ms = %.1f GB/s\n", a * 1e3, n / a / 1e9, b * 1e3, n / b / 1e9);
  }
  t = now();
  CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2));
  CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
  double both = now() - t;
  printf("registered, both directions at once: %.1f ms = %.1f GB/s each way\n", both * 1e3, n / both / 1e9);
  t = now(); CK(hipHostUnregister(h)); double un = now() - t;
  printf("hipHostUnregister: %.1f ms\n", un * 1e3);
  return 0;
}
