// variants.hip -- the reference's other names for the depression-filling sweep, on the engines of fill.hip:
//
//   HasDepressions<topo>(const Array2D<T>&)      depressions/Barnes2014.hpp:44-103   (apps/rd_depressions_has.cpp:14)
//   PriorityFlood_Wei2018(Array2D<T>&)           depressions/Wei2018.hpp:154-202 (InitPriorityQue :14-50)
//   (PriorityFlood_Original<topo>, depressions/Barnes2014.hpp:136-198, returns the surface of FillDepressions<topo>: the
//    shim maps it to rdgpu_fill_<T>.)
//
// HasDepressions runs the flood of PriorityFlood_Original without raising anything and stops at the first cell that
// is discovered from a higher one (:91-95).  Up to that moment the two floods are the same sequence of pops, so the answer
// is "the fill raises at least one cell" -- a function of the DEM alone, whatever the heap does among equal keys.
//
// Wei2018 differs from the other fills in ONE respect, its seeds (InitPriorityQue): every NoData cell is flagged before
// the flood starts and never touched, and every data cell NEXT to a NoData cell enters the queue at its own elevation
// beside the raster's edge cells.  So interior NoData regions are outlets: W(c) = min over 8-connected paths of data cells
// from c to a seed of the highest elevation on the path.  A NoData cell may be marked an outlet itself without changing
// anything (all its data neighbours are outlets already, and an outlet keeps its own value), so the surface is the D8
// fill with interior outlets (k_descent16<OUTLETS>) on the mask "NoData, or next to NoData".
#include "common.hpp"

#include <string>
#include <type_traits>

namespace rdgpu {

namespace {
constexpr int NTHR = 256;
inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

// outlet[i] = the cell is NoData or has a NoData cell among its 8 neighbours inside the raster (Wei2018.hpp:26-43)
template <class T>
__global__ __launch_bounds__(NTHR) void k_wei_outlets(const T *__restrict__ z, T nodata, int w, int h, uint8_t *__restrict__ outlet) {
  const uint64_t n = (uint64_t)w * h, stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    const int y = (int)(i / (uint64_t)w), x = (int)(i - (uint64_t)y * w);
    bool o = false;
    for (int dy = -1; dy <= 1; dy++) {
      const int yy = y + dy;
      if (yy < 0 || yy >= h) continue;
      for (int dx = -1; dx <= 1; dx++) {
        const int xx = x + dx;
        if (xx < 0 || xx >= w) continue;
        o |= z[(size_t)yy * w + xx] == nodata;
      }
    }
    outlet[i] = o ? 1 : 0;
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_any_differs(const T *__restrict__ a, const T *__restrict__ b, uint64_t n, uint32_t *flag) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  bool d = false;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) d |= a[i] != b[i];
  if (__any(d) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

void check_raster(const void *p, int w, int h, const char *who) {
  if (!p) throw Error(RDGPU_ERR_ARG, std::string(who) + ": null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, std::string(who) + ": width and height must be positive");
  if ((uint64_t)w * h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, std::string(who) + ": raster has more than 2^31-65536 cells");
}

template <class T, class FillDev>
int has_depressions_device(const T *d_z, int w, int h, int topology, FillDev fill_dev, hipStream_t s) {
  check_raster(d_z, w, h, "rdgpu_has_depressions");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_has_depressions: topology must be 8 or 4");
  const uint64_t n = (uint64_t)w * h;
  Workspace &ws = Workspace::get();
  T *filled = ws.buf<T>("variants.filled", n);
  uint32_t *flag = ws.buf<uint32_t>("variants.flag", 1);
  RD_HIP(hipMemcpyAsync(filled, d_z, n * sizeof(T), hipMemcpyDeviceToDevice, s));
  const int rc = fill_dev(filled, w, h, topology, (void *)s);
  if (rc) throw Error(rc, rdgpu_last_error());
  RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
  hipLaunchKernelGGL((k_any_differs<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, (const T *)filled, n, flag);
  RD_HIP(hipGetLastError());
  uint32_t *hw = ws.host_words();
  RD_HIP(hipMemcpyAsync(hw, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  return hw[0] != 0;
}

template <class T, class OutletsDev>
void wei2018_device(T *d_z, T nodata, int w, int h, OutletsDev outlets_dev, hipStream_t s) {
  check_raster(d_z, w, h, "rdgpu_fill_wei2018");
  const uint64_t n = (uint64_t)w * h;
  uint8_t *outlet = Workspace::get().buf<uint8_t>("variants.outlet", n);
  hipLaunchKernelGGL((k_wei_outlets<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, (const T *)d_z, nodata, w, h, outlet);
  RD_HIP(hipGetLastError());
  const int rc = outlets_dev(d_z, (const uint8_t *)outlet, w, h, (void *)s);
  if (rc) throw Error(rc, rdgpu_last_error());
}

template <class T, class F>
void with_upload(T *dem, int w, int h, bool copy_back, const char *who, F &&fn) {
  check_raster(dem, w, h, who);
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("variants.dem", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  fn(d);
  RD_HIP(hipStreamSynchronize(nullptr));
  if (copy_back) RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
}
}  // namespace

}  // namespace rdgpu

using namespace rdgpu;

#define RD_HASDEP_API(SUF, T)                                                                                     \
  extern "C" int rdgpu_fill_dev_##SUF(T *, int, int, int, void *);                                                \
  extern "C" int rdgpu_has_depressions_dev_##SUF(const T *d_dem, int w, int h, int topology, int *out, void *stream) { \
    return guarded([&] {                                                                                          \
      if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_has_depressions: null result pointer");                         \
      *out = has_depressions_device<T>(d_dem, w, h, topology, rdgpu_fill_dev_##SUF, (hipStream_t)stream);         \
    });                                                                                                           \
  }                                                                                                               \
  extern "C" int rdgpu_has_depressions_##SUF(const T *dem, int w, int h, int topology, int *out) {                \
    return guarded([&] {                                                                                          \
      if (!out) throw Error(RDGPU_ERR_ARG, "rdgpu_has_depressions: null result pointer");                         \
      with_upload<T>(const_cast<T *>(dem), w, h, false, "rdgpu_has_depressions", [&](T *d) {                      \
        *out = has_depressions_device<T>(d, w, h, topology, rdgpu_fill_dev_##SUF, nullptr);                       \
      });                                                                                                         \
    });                                                                                                           \
  }
RD_HASDEP_API(u8, uint8_t)
RD_HASDEP_API(i8, int8_t)
RD_HASDEP_API(i16, int16_t)
RD_HASDEP_API(u16, uint16_t)
RD_HASDEP_API(i32, int32_t)
RD_HASDEP_API(u32, uint32_t)
RD_HASDEP_API(f32, float)
RD_HASDEP_API(f64, double)
RD_HASDEP_API(i64, int64_t)
RD_HASDEP_API(u64, uint64_t)

#define RD_WEI_API(SUF, T)                                                                                        \
  extern "C" int rdgpu_fill_outlets_dev_##SUF(T *, const uint8_t *, int, int, void *);                            \
  extern "C" int rdgpu_fill_wei2018_dev_##SUF(T *d_dem, T nodata, int w, int h, void *stream) {                   \
    return guarded([&] { wei2018_device<T>(d_dem, nodata, w, h, rdgpu_fill_outlets_dev_##SUF, (hipStream_t)stream); }); \
  }                                                                                                               \
  extern "C" int rdgpu_fill_wei2018_##SUF(T *dem, T nodata, int w, int h) {                                       \
    return guarded([&] {                                                                                          \
      with_upload<T>(dem, w, h, true, "rdgpu_fill_wei2018",                                                       \
                     [&](T *d) { wei2018_device<T>(d, nodata, w, h, rdgpu_fill_outlets_dev_##SUF, nullptr); });   \
    });                                                                                                           \
  }
RD_WEI_API(u8, uint8_t)
RD_WEI_API(i8, int8_t)
RD_WEI_API(i16, int16_t)
RD_WEI_API(u16, uint16_t)
RD_WEI_API(i32, int32_t)
RD_WEI_API(u32, uint32_t)
RD_WEI_API(f32, float)
