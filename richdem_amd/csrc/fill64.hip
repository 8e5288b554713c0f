// fill64.hip -- FillDepressions for 64-bit element types (f64, i64, u64).
//
// The fill engine (fill.hip) reduces (pass key << 32 | component) with one 64-bit atomic, so it works on
// 32-bit order-preserving keys.  The fill only compares and copies elevations, so ANY
// order-preserving injection of the DEM's values into 32 bits gives the exact result:
//   * f64 whose values are all exactly representable as f32 (DEMs stored as float64 but measured in
//     float32 / integers -- numpy's default dtype makes this the common case): run the f32 engine;
//   * otherwise: dense ranks.  Sort (key64, cell) pairs (hipCUB radix sort), rank = number of distinct
//     smaller keys (< 2^32 because there are < 2^32 cells), run the u32 engine on the rank raster, map the
//     filled ranks back through the table of distinct keys.
// A cell is rewritten only when its level changed, so untouched cells keep their input bits.
#include "common.hpp"

#include <hipcub/hipcub.hpp>

extern "C" int rdgpu_fill_dev_u32(uint32_t *, int, int, int, void *);
extern "C" int rdgpu_fill_dev_f32(float *, int, int, int, void *);

namespace rdgpu {

constexpr int NTHR = 256;

template <class T>
struct Key64;
template <>
struct Key64<double> {
  __host__ __device__ static inline uint64_t to(double v) {
    uint64_t b = __builtin_bit_cast(uint64_t, v);
    if (b == 0x8000000000000000ull) b = 0;   // -0.0 == +0.0, as in the reference's comparisons
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  }
  __host__ __device__ static inline double from(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __builtin_bit_cast(double, b);
  }
};
template <>
struct Key64<int64_t> {
  __host__ __device__ static inline uint64_t to(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
  __host__ __device__ static inline int64_t from(uint64_t k) { return (int64_t)(k ^ 0x8000000000000000ull); }
};
template <>
struct Key64<uint64_t> {
  __host__ __device__ static inline uint64_t to(uint64_t v) { return v; }
  __host__ __device__ static inline uint64_t from(uint64_t k) { return k; }
};

static inline uint32_t sgrid(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + NTHR - 1) / NTHR, 256u * 32u); }

__global__ __launch_bounds__(NTHR) void k_f64_fits_f32(const double *__restrict__ z, uint64_t n, uint32_t *bad) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  bool b = false;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    const double v = z[i];
    b |= !((double)(float)v == v);   // also true for NaN and for values beyond the f32 range
  }
  if (__any(b) && (threadIdx.x & 63) == 0) *bad = 1;   // at most one store per wave, all the same value
}

__global__ __launch_bounds__(NTHR) void k_f64_to_f32(const double *__restrict__ z, float *f, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) f[i] = (float)z[i];
}

__global__ __launch_bounds__(NTHR) void k_f32_back_to_f64(double *z, const float *__restrict__ f, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    const double v = (double)f[i];
    if (v != z[i]) z[i] = v;
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_keys_iota(const T *__restrict__ z, uint64_t *keys, uint32_t *idx, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    keys[i] = Key64<T>::to(z[i]);
    idx[i] = (uint32_t)i;
  }
}

// head[i] = 1 where a new distinct key starts in the sorted sequence
__global__ __launch_bounds__(NTHR) void k_heads(const uint64_t *__restrict__ sk, uint32_t *head, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride)
    head[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1u : 0u;
}

// rank[i] (inclusive scan of head) - 1 is the dense rank of sorted element i
__global__ __launch_bounds__(NTHR) void k_scatter_ranks(const uint64_t *__restrict__ sk, const uint32_t *__restrict__ sidx,
                                                        const uint32_t *__restrict__ rank, const uint32_t *__restrict__ head,
                                                        uint32_t *rk, uint64_t *uniq, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    const uint32_t r = rank[i] - 1u;
    rk[sidx[i]] = r;
    if (head[i]) uniq[r] = sk[i];
  }
}

template <class T>
__global__ __launch_bounds__(NTHR) void k_ranks_back(T *z, const uint32_t *__restrict__ rk, const uint64_t *__restrict__ uniq,
                                                     uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NTHR;
  for (uint64_t i = (uint64_t)blockIdx.x * NTHR + threadIdx.x; i < n; i += stride) {
    const uint64_t k = uniq[rk[i]];
    if (k != Key64<T>::to(z[i])) z[i] = Key64<T>::from(k);
  }
}

template <class T>
static void fill64_device(T *d_z, int w, int h, int topology, hipStream_t s) {
  if (!d_z) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: width and height must be positive");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: topology must be 8 or 4");
  const uint64_t n = (uint64_t)w * h;
  if (n > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: raster has more than 2^31-65536 cells");
  Workspace &ws = Workspace::get();
  uint32_t *hw = ws.host_words();
  uint32_t *flag = ws.buf<uint32_t>("fill64.flag", 4);

  if (std::is_same<T, double>::value) {
    RD_HIP(hipMemsetAsync(flag, 0, 4, s));
    RD_LAUNCH("fill64.fits_f32", k_f64_fits_f32, dim3(sgrid(n)), dim3(NTHR), 0, s, (const double *)d_z, n, flag);
    RD_HIP(hipMemcpyAsync(hw, flag, 4, hipMemcpyDeviceToHost, s));
    RD_HIP(hipStreamSynchronize(s));
    if (hw[0] == 0) {   // lossless in f32: the f32 engine is exact
      float *f = ws.buf<float>("fill64.f32", n);
      RD_LAUNCH("fill64.to_f32", k_f64_to_f32, dim3(sgrid(n)), dim3(NTHR), 0, s, (const double *)d_z, f, n);
      const int rc = rdgpu_fill_dev_f32(f, w, h, topology, s);
      if (rc) throw Error(rc, rdgpu_last_error());
      RD_LAUNCH("fill64.from_f32", k_f32_back_to_f64, dim3(sgrid(n)), dim3(NTHR), 0, s, (double *)d_z, (const float *)f, n);
      return;
    }
  }
  // dense ranks
  uint64_t *keys = ws.buf<uint64_t>("fill64.keys", n), *skeys = ws.buf<uint64_t>("fill64.skeys", n);
  uint32_t *idx = ws.buf<uint32_t>("fill64.idx", n), *sidx = ws.buf<uint32_t>("fill64.sidx", n);
  RD_LAUNCH("fill64.keys", (k_keys_iota<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, (const T *)d_z, keys, idx, n);
  size_t tb = 0;
  RD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, skeys, idx, sidx, (int)n, 0, 64, s));
  void *tmp = ws.buf("fill64.tmp", tb);
  RD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, skeys, idx, sidx, (int)n, 0, 64, s));
  uint32_t *head = idx;                                // idx is dead after the sort
  uint32_t *rank = reinterpret_cast<uint32_t *>(keys); // so is keys (n * 8 bytes >= n * 4)
  RD_LAUNCH("fill64.heads", k_heads, dim3(sgrid(n)), dim3(NTHR), 0, s, (const uint64_t *)skeys, head, n);
  size_t tb2 = 0;
  RD_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb2, head, rank, (int)n, s));
  void *tmp2 = ws.buf("fill64.tmp2", tb2);
  RD_HIP(hipcub::DeviceScan::InclusiveSum(tmp2, tb2, head, rank, (int)n, s));
  uint32_t *rk = ws.buf<uint32_t>("fill64.rk", n);
  uint64_t *uniq = ws.buf<uint64_t>("fill64.uniq", n);
  RD_LAUNCH("fill64.scatter_ranks", k_scatter_ranks, dim3(sgrid(n)), dim3(NTHR), 0, s, (const uint64_t *)skeys,
            (const uint32_t *)sidx, (const uint32_t *)rank, (const uint32_t *)head, rk, uniq, n);
  const int rc = rdgpu_fill_dev_u32(rk, w, h, topology, s);
  if (rc) throw Error(rc, rdgpu_last_error());
  RD_LAUNCH("fill64.ranks_back", (k_ranks_back<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, (const uint32_t *)rk,
            (const uint64_t *)uniq, n);
}

template <class T>
static void fill64_host(T *dem, int w, int h, int topology) {
  if (!dem) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem64", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  fill64_device<T>(d, w, h, topology, nullptr);
  RD_HIP(hipStreamSynchronize(nullptr));
  RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FILL64_API(SUF, T)                                                                     \
  extern "C" int rdgpu_fill_##SUF(T *dem, int w, int h, int topology) {                           \
    return guarded([&] { fill64_host<T>(dem, w, h, topology); });                                 \
  }                                                                                               \
  extern "C" int rdgpu_fill_dev_##SUF(T *d_dem, int w, int h, int topology, void *stream) {       \
    return guarded([&] { fill64_device<T>(d_dem, w, h, topology, (hipStream_t)stream); });        \
  }
RD_FILL64_API(f64, double)
RD_FILL64_API(i64, int64_t)
RD_FILL64_API(u64, uint64_t)
