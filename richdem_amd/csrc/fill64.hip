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

#include <string>

extern "C" int rdgpu_fill_dev_u32(uint32_t *, int, int, int, void *);
extern "C" int rdgpu_fill_dev_f32(float *, int, int, int, void *);
extern "C" int rdgpu_fill_max_dep_dev_u32(uint32_t *, int, int, int, uint64_t, void *);
extern "C" int rdgpu_pit_mask_dev_u32(const uint32_t *, uint32_t, int, int, int, uint8_t *, void *);
extern "C" int rdgpu_watersheds_dev_u32(uint32_t *, uint32_t, int, int, int, int, int32_t *, void *);

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

// Dense value ranks of a 64-bit raster: rk[cell] = number of distinct smaller keys, uniq[rank] = the key.  Any function of
// the DEM that only compares and copies elevations gives, on the rank raster, the ranks of what it gives on the DEM.
struct Ranks {
  uint32_t *rk;
  uint64_t *uniq;
  uint32_t *last_rank;   // device word: rank of the largest key (number of distinct keys - 1)
};
template <class T>
static Ranks dense_ranks(const T *d_z, uint64_t n, hipStream_t s) {
  Workspace &ws = Workspace::get();
  uint64_t *keys = ws.buf<uint64_t>("fill64.keys", n), *skeys = ws.buf<uint64_t>("fill64.skeys", n);
  uint32_t *idx = ws.buf<uint32_t>("fill64.idx", n), *sidx = ws.buf<uint32_t>("fill64.sidx", n);
  RD_LAUNCH("fill64.keys", (k_keys_iota<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, keys, idx, n);
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
  return Ranks{rk, uniq, rank + (n - 1)};   // (rank[] holds the inclusive scan: its last entry = number of distinct keys)
}

// rank of `key` among the distinct keys, or 0xFFFFFFFF when the raster does not hold it (binary search, one thread)
__global__ void k_rank_of(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ ndistinct, uint64_t key, uint32_t *out) {
  uint32_t lo = 0, hi = *ndistinct;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (uniq[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  *out = (lo < *ndistinct && uniq[lo] == key) ? lo : 0xFFFFFFFFu;
}

// the NoData value as a rank (what the u32 engines compare cells with); 0xFFFFFFFF = no cell holds it
template <class T>
static uint32_t nodata_rank(const Ranks &r, T nodata, hipStream_t s) {
  Workspace &ws = Workspace::get();
  uint32_t *d = ws.buf<uint32_t>("fill64.ndrank", 1);
  hipLaunchKernelGGL(k_rank_of, dim3(1), dim3(1), 0, s, (const uint64_t *)r.uniq, (const uint32_t *)r.last_rank,
                     Key64<T>::to(nodata), d);
  RD_HIP(hipGetLastError());
  uint32_t *hw = ws.host_words();
  RD_HIP(hipMemcpyAsync(hw, d, 4, hipMemcpyDeviceToHost, s));
  RD_HIP(hipStreamSynchronize(s));
  return hw[0];
}

static void check64(const void *p, int w, int h, int topology, const char *who) {
  if (!p) throw Error(RDGPU_ERR_ARG, std::string(who) + ": null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, std::string(who) + ": width and height must be positive");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, std::string(who) + ": topology must be 8 or 4");
  if ((uint64_t)w * h > 0x7FFF0000ull) throw Error(RDGPU_ERR_ARG, std::string(who) + ": raster has more than 2^31-65536 cells");
}

// ---- the other outputs of the sweep for 64-bit element types, on the rank raster ------------------------
// PriorityFlood_Barnes2014_max_dep, pit_mask and PriorityFloodWatersheds only compare and copy elevations (and count
// cells), so they run on the dense ranks with the u32 engine; levels come back through the table of distinct keys.
template <class T>
static void fill_max_dep64_device(T *d_z, int w, int h, int topology, uint64_t max_dep, hipStream_t s) {
  check64(d_z, w, h, topology, "rdgpu_fill_max_dep");
  const uint64_t n = (uint64_t)w * h;
  const Ranks r = dense_ranks<T>(d_z, n, s);
  const int rc = rdgpu_fill_max_dep_dev_u32(r.rk, w, h, topology, max_dep, s);
  if (rc) throw Error(rc, rdgpu_last_error());
  RD_LAUNCH("fill64.ranks_back", (k_ranks_back<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, (const uint32_t *)r.rk,
            (const uint64_t *)r.uniq, n);
}

template <class T>
static void pit_mask64_device(const T *d_z, T nodata, int w, int h, int topology, uint8_t *d_mask, hipStream_t s) {
  check64(d_z, w, h, topology, "rdgpu_pit_mask");
  if (!d_mask) throw Error(RDGPU_ERR_ARG, "rdgpu_pit_mask: null pointer");
  const uint64_t n = (uint64_t)w * h;
  const Ranks r = dense_ranks<T>(d_z, n, s);
  const int rc = rdgpu_pit_mask_dev_u32(r.rk, nodata_rank<T>(r, nodata, s), w, h, topology, d_mask, s);
  if (rc) throw Error(rc, rdgpu_last_error());
}

template <class T>
static void watersheds64_device(T *d_z, T nodata, int w, int h, int topology, int alter, int32_t *d_labels, hipStream_t s) {
  check64(d_z, w, h, topology, "rdgpu_watersheds");
  if (!d_labels) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: null pointer");
  const uint64_t n = (uint64_t)w * h;
  const Ranks r = dense_ranks<T>(d_z, n, s);
  const int rc = rdgpu_watersheds_dev_u32(r.rk, nodata_rank<T>(r, nodata, s), w, h, topology, alter, d_labels, s);
  if (rc) throw Error(rc, rdgpu_last_error());
  if (alter)
    RD_LAUNCH("fill64.ranks_back", (k_ranks_back<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, (const uint32_t *)r.rk,
              (const uint64_t *)r.uniq, n);
}

// PriorityFloodFlowdirs only compares elevations too: the u32 engine (pfdirs.hip) on the ranks
extern "C" int rdgpu_pf_flowdirs_dev_u32(const uint32_t *, uint32_t, int, int, uint8_t *, void *);
template <class T>
static void pf_flowdirs64_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, hipStream_t s) {
  check64(d_z, w, h, 8, "rdgpu_pf_flowdirs");
  if (!d_dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: null pointer");
  const Ranks r = dense_ranks<T>(d_z, (uint64_t)w * h, s);
  const int rc = rdgpu_pf_flowdirs_dev_u32(r.rk, nodata_rank<T>(r, nodata, s), w, h, d_dirs, s);
  if (rc) throw Error(rc, rdgpu_last_error());
}

// PriorityFlood_Wei2018 (depressions/Wei2018.hpp:154-202) compares and copies elevations and tests cells against NoData:
// the u32 engine (variants.hip) on the ranks, NoData as its rank
extern "C" int rdgpu_fill_wei2018_dev_u32(uint32_t *, uint32_t, int, int, void *);
template <class T>
static void wei2018_64_device(T *d_z, T nodata, int w, int h, hipStream_t s) {
  check64(d_z, w, h, 8, "rdgpu_fill_wei2018");
  const uint64_t n = (uint64_t)w * h;
  const Ranks r = dense_ranks<T>(d_z, n, s);
  const int rc = rdgpu_fill_wei2018_dev_u32(r.rk, nodata_rank<T>(r, nodata, s), w, h, s);
  if (rc) throw Error(rc, rdgpu_last_error());
  RD_LAUNCH("fill64.ranks_back", (k_ranks_back<T>), dim3(sgrid(n)), dim3(NTHR), 0, s, d_z, (const uint32_t *)r.rk,
            (const uint64_t *)r.uniq, n);
}

// host-pointer forms: H2D, the device form, D2H of what the call produces
template <class T, class F>
static void with_device_copy(T *dem, int w, int h, bool copy_back, F &&fn) {
  if (!dem) throw Error(RDGPU_ERR_ARG, "rdgpu: null DEM pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu: width and height must be positive");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem64", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  fn(d);
  RD_HIP(hipStreamSynchronize(nullptr));
  if (copy_back) RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
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
  const Ranks r = dense_ranks<T>(d_z, n, s);
  uint32_t *rk = r.rk;
  uint64_t *uniq = r.uniq;
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

#define RD_F2_64_API(SUF, T)                                                                                           \
  extern "C" int rdgpu_fill_max_dep_dev_##SUF(T *d_dem, int w, int h, int topology, uint64_t max_dep_size, void *stream) { \
    return guarded([&] { fill_max_dep64_device<T>(d_dem, w, h, topology, max_dep_size, (hipStream_t)stream); });       \
  }                                                                                                                    \
  extern "C" int rdgpu_fill_max_dep_##SUF(T *dem, int w, int h, int topology, uint64_t max_dep_size) {                 \
    return guarded([&] {                                                                                               \
      with_device_copy<T>(dem, w, h, true, [&](T *d) { fill_max_dep64_device<T>(d, w, h, topology, max_dep_size, nullptr); }); \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" int rdgpu_pit_mask_dev_##SUF(const T *d_dem, T nodata, int w, int h, int topology, uint8_t *d_mask, void *stream) { \
    return guarded([&] { pit_mask64_device<T>(d_dem, nodata, w, h, topology, d_mask, (hipStream_t)stream); });         \
  }                                                                                                                    \
  extern "C" int rdgpu_pit_mask_##SUF(const T *dem, T nodata, int w, int h, int topology, uint8_t *mask) {             \
    return guarded([&] {                                                                                               \
      if (!mask) throw Error(RDGPU_ERR_ARG, "rdgpu_pit_mask: null pointer");                                           \
      with_device_copy<T>(const_cast<T *>(dem), w, h, false, [&](T *d) {                                               \
        uint8_t *dm = Workspace::get().buf<uint8_t>("host.mask64", (size_t)w * h);                                     \
        pit_mask64_device<T>(d, nodata, w, h, topology, dm, nullptr);                                                  \
        RD_HIP(hipStreamSynchronize(nullptr));                                                                         \
        RD_HIP(hipMemcpy(mask, dm, (size_t)w * h, hipMemcpyDeviceToHost));                                             \
      });                                                                                                              \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" int rdgpu_watersheds_dev_##SUF(T *d_dem, T nodata, int w, int h, int topology, int alter, int32_t *d_labels, \
                                            void *stream) {                                                            \
    return guarded([&] { watersheds64_device<T>(d_dem, nodata, w, h, topology, alter, d_labels, (hipStream_t)stream); }); \
  }                                                                                                                    \
  extern "C" int rdgpu_watersheds_##SUF(T *dem, T nodata, int w, int h, int topology, int alter, int32_t *labels) {    \
    return guarded([&] {                                                                                               \
      if (!labels) throw Error(RDGPU_ERR_ARG, "rdgpu_watersheds: null pointer");                                       \
      with_device_copy<T>(dem, w, h, alter != 0, [&](T *d) {                                                           \
        int32_t *dl = Workspace::get().buf<int32_t>("host.labels64", (size_t)w * h);                                   \
        watersheds64_device<T>(d, nodata, w, h, topology, alter, dl, nullptr);                                         \
        RD_HIP(hipStreamSynchronize(nullptr));                                                                         \
        RD_HIP(hipMemcpy(labels, dl, (size_t)w * h * sizeof(int32_t), hipMemcpyDeviceToHost));                         \
      });                                                                                                              \
    });                                                                                                                \
  }
RD_F2_64_API(f64, double)
RD_F2_64_API(i64, int64_t)
RD_F2_64_API(u64, uint64_t)

#define RD_PFD64_API(SUF, T)                                                                                           \
  extern "C" int rdgpu_pf_flowdirs_dev_##SUF(const T *d_dem, T nodata, int w, int h, uint8_t *d_dirs, void *stream) {  \
    return guarded([&] { pf_flowdirs64_device<T>(d_dem, nodata, w, h, d_dirs, (hipStream_t)stream); });                \
  }                                                                                                                    \
  extern "C" int rdgpu_pf_flowdirs_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs) {                        \
    return guarded([&] {                                                                                               \
      if (!dirs) throw Error(RDGPU_ERR_ARG, "rdgpu_pf_flowdirs: null pointer");                                        \
      with_device_copy<T>(const_cast<T *>(dem), w, h, false, [&](T *d) {                                               \
        uint8_t *dd = Workspace::get().buf<uint8_t>("host.mask64", (size_t)w * h);                                     \
        pf_flowdirs64_device<T>(d, nodata, w, h, dd, nullptr);                                                         \
        RD_HIP(hipStreamSynchronize(nullptr));                                                                         \
        RD_HIP(hipMemcpy(dirs, dd, (size_t)w * h, hipMemcpyDeviceToHost));                                             \
      });                                                                                                              \
    });                                                                                                                \
  }
RD_PFD64_API(f64, double)
RD_PFD64_API(i64, int64_t)
RD_PFD64_API(u64, uint64_t)

#define RD_WEI64_API(SUF, T)                                                                                           \
  extern "C" int rdgpu_fill_wei2018_dev_##SUF(T *d_dem, T nodata, int w, int h, void *stream) {                        \
    return guarded([&] { wei2018_64_device<T>(d_dem, nodata, w, h, (hipStream_t)stream); });                           \
  }                                                                                                                    \
  extern "C" int rdgpu_fill_wei2018_##SUF(T *dem, T nodata, int w, int h) {                                            \
    return guarded([&] { with_device_copy<T>(dem, w, h, true, [&](T *d) { wei2018_64_device<T>(d, nodata, w, h, nullptr); }); }); \
  }
RD_WEI64_API(f64, double)
RD_WEI64_API(i64, int64_t)
RD_WEI64_API(u64, uint64_t)
