// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrappers around the UNMODIFIED reference headers under
// /root/reference/include (never copied into this repo).  Built by
// oracle/Makefile into oracle/_ref/libref.so when /root/reference is present
// (-std=c++17 -O3 -fopenmp -DNDEBUG -DRICHDEM_NO_PROGRESS).
// Used to (1) pin oracle/oracle.c against the real reference, (2) generate the
// golden fixtures in tests/golden/, (3) optionally serve as bench.py's
// cpu_baseline ("kind": "reference").  Never on the product path.
//
// Each wrapper wraps caller memory with the reference's own wrap constructor
// Array2D(T*,w,h) (common/Array2D.hpp:344-352) so the algorithms run on exactly
// the bytes the caller passed.
#include <richdem/common/Array2D.hpp>
#include <richdem/common/Array3D.hpp>
#include <richdem/depressions/depressions.hpp>
#include <richdem/flowmet/d8_flowdirs.hpp>
#include <richdem/flowmet/dinf_flowdirs.hpp>
#include <richdem/flats/flat_resolution.hpp>
#include <richdem/flats/flats.hpp>
#include <richdem/methods/d8_methods.hpp>
#include <richdem/methods/flow_accumulation.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <omp.h>

using namespace richdem;

namespace {

template <class T>
void ref_fill(T *dem, int w, int h, int variant) {
  Array2D<T> a(dem, w, h);
  switch (variant) {
  case 0: FillDepressions<Topology::D8>(a); break;            // = PriorityFlood_Zhou2016 (depressions.hpp:13-21)
  case 1: PriorityFlood_Barnes2014<Topology::D8>(a); break;   // depressions/Barnes2014.hpp:230
  case 2: FillDepressions<Topology::D4>(a); break;            // = PriorityFlood_Barnes2014<D4>
  case 3: PriorityFlood_Wei2018(a); break;                    // depressions/Wei2018.hpp:154
  case 4: PriorityFlood_Original<Topology::D8>(a); break;     // depressions/Barnes2014.hpp:136
  }
}

template <class T>
void ref_d8_flowdirs(const T *dem, T nodata, int w, int h, uint8_t *out) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<uint8_t> fd;
  d8_flow_directions(a, fd);   // flowmet/d8_flowdirs.hpp:96-123
  std::memcpy(out, fd.data(), (size_t)w * h);
}

template <class T>
void ref_flat_resolution(T *dem, T nodata, int w, int h, uint8_t *out, int alter) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  Array2D<uint8_t> fd;
  barnes_flat_resolution_d8(a, fd, alter != 0);   // flats/flat_resolution.hpp:587-605
  std::memcpy(out, fd.data(), (size_t)w * h);
}

// Exposes the intermediate flat_mask / labels of resolve_flats_barnes
// (flats/flat_resolution.hpp:447-517) so the oracle restatement can be pinned
// on more than the final directions.
template <class T>
void ref_resolve_flats(const T *dem, T nodata, int w, int h, uint8_t *dirs_out,
                       int32_t *mask_out, int32_t *labels_out) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<uint8_t> fd;
  d8_flow_directions(a, fd);
  Array2D<int32_t> flat_mask, labels;
  resolve_flats_barnes(a, fd, flat_mask, labels);
  std::memcpy(dirs_out, fd.data(), (size_t)w * h);
  std::memcpy(mask_out, flat_mask.data(), (size_t)w * h * 4);
  std::memcpy(labels_out, labels.data(), (size_t)w * h * 4);
}

template <class A>
void ref_d8_flow_accum(const uint8_t *dirs, uint8_t nodata, int w, int h, A *out) {
  Array2D<uint8_t> fd(const_cast<uint8_t *>(dirs), w, h);
  fd.setNoData(nodata);
  Array2D<A> area;
  // The reference's dependency pass does a non-atomic ++dependency under
  // "omp parallel for" (methods/d8_methods.hpp:68-90); run it on one thread so
  // the parity run is deterministic.
  const int nt = omp_get_max_threads();
  omp_set_num_threads(1);
  d8_flow_accum(fd, area);    // methods/d8_methods.hpp:47-139
  omp_set_num_threads(nt);
  std::memcpy(out, area.data(), (size_t)w * h * sizeof(A));
}

template <class T>
void ref_fa_d8(const T *dem, T nodata, int w, int h, double *accum) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<double> acc(accum, w, h);
  FA_D8(a, acc);              // methods/flow_accumulation.hpp:27
}

template <class T>
void ref_fm_d8(const T *dem, T nodata, int w, int h, float *props9) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array3D<float> props(a);
  FM_D8(a, props);            // flowmet/OCallaghan1984.hpp:81-84
  std::memcpy(props9, props.getData(), (size_t)w * h * 9 * sizeof(float));
}

template <class T>
void ref_fm_mfd(const T *dem, T nodata, int w, int h, int method, double xparam, float *props9) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array3D<float> props(a);
  switch (method) {
    case 0: FM_Holmgren(a, props, xparam); break;                 // flowmet/Holmgren1994.hpp:14
    case 1: FM_Freeman(a, props, xparam); break;                  // flowmet/Freeman1991.hpp:14
    case 2: FM_Quinn(a, props); break;                            // flowmet/Quinn1991.hpp:13
    default: FM_D4(a, props); break;                              // flowmet/OCallaghan1984.hpp:86
  }
  std::memcpy(props9, props.getData(), (size_t)w * h * 9 * sizeof(float));
}

template <class T>
void ref_fa_mfd(const T *dem, T nodata, int w, int h, int method, double xparam, double *accum) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<double> acc(accum, w, h);
  switch (method) {
    case 0: FA_Holmgren(a, acc, xparam); break;                   // methods/flow_accumulation.hpp:18
    case 1: FA_Freeman(a, acc, xparam); break;
    case 2: FA_Quinn(a, acc); break;
    default: FA_D4(a, acc); break;
  }
}

template <class T>
void ref_dinf_flowdirs(const T *dem, T nodata, int w, int h, float *out) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<float> fd;
  dinf_flow_directions(a, fd);   // flowmet/dinf_flowdirs.hpp:128-152
  std::memcpy(out, fd.data(), (size_t)w * h * sizeof(float));
}

template <class T>
void ref_fm_tarboton(const T *dem, T nodata, int w, int h, float *props9) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array3D<float> props(a);
  FM_Tarboton(a, props);         // flowmet/Tarboton1997.hpp:14-144
  std::memcpy(props9, props.getData(), (size_t)w * h * 9 * sizeof(float));
}

template <class T>
void ref_fa_tarboton(const T *dem, T nodata, int w, int h, double *accum) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<double> acc(accum, w, h);
  FA_Tarboton(a, acc);           // methods/flow_accumulation.hpp:16
}

} // namespace

#define REF_ELEV_API(SUF, T)                                                                     \
  extern "C" void ref_fill_##SUF(T *dem, int w, int h, int variant) { ref_fill<T>(dem, w, h, variant); } \
  extern "C" void ref_d8_flowdirs_##SUF(const T *dem, T nodata, int w, int h, uint8_t *out) {    \
    ref_d8_flowdirs<T>(dem, nodata, w, h, out);                                                  \
  }                                                                                              \
  extern "C" void ref_flat_resolution_##SUF(T *dem, T nodata, int w, int h, uint8_t *out, int alter) { \
    ref_flat_resolution<T>(dem, nodata, w, h, out, alter);                                       \
  }                                                                                              \
  extern "C" void ref_resolve_flats_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs,   \
                                          int32_t *mask, int32_t *labels) {                      \
    ref_resolve_flats<T>(dem, nodata, w, h, dirs, mask, labels);                                 \
  }                                                                                              \
  extern "C" void ref_fa_d8_##SUF(const T *dem, T nodata, int w, int h, double *accum) {         \
    ref_fa_d8<T>(dem, nodata, w, h, accum);                                                      \
  }                                                                                              \
  extern "C" void ref_fm_d8_##SUF(const T *dem, T nodata, int w, int h, float *props9) {         \
    ref_fm_d8<T>(dem, nodata, w, h, props9);                                                     \
  }                                                                                              \
  extern "C" void ref_dinf_flowdirs_##SUF(const T *dem, T nodata, int w, int h, float *out) {    \
    ref_dinf_flowdirs<T>(dem, nodata, w, h, out);                                                \
  }                                                                                              \
  extern "C" void ref_fm_tarboton_##SUF(const T *dem, T nodata, int w, int h, float *props9) {   \
    ref_fm_tarboton<T>(dem, nodata, w, h, props9);                                               \
  }                                                                                              \
  extern "C" void ref_fa_tarboton_##SUF(const T *dem, T nodata, int w, int h, double *accum) {   \
    ref_fa_tarboton<T>(dem, nodata, w, h, accum);                                                \
  }                                                                                              \
  extern "C" void ref_fm_mfd_##SUF(const T *dem, T nodata, int w, int h, int method, double xparam, float *props9) { \
    ref_fm_mfd<T>(dem, nodata, w, h, method, xparam, props9);                                    \
  }                                                                                              \
  extern "C" void ref_fa_mfd_##SUF(const T *dem, T nodata, int w, int h, int method, double xparam, double *accum) { \
    ref_fa_mfd<T>(dem, nodata, w, h, method, xparam, accum);                                     \
  }

REF_ELEV_API(u8, uint8_t)
REF_ELEV_API(i8, int8_t)
REF_ELEV_API(i16, int16_t)
REF_ELEV_API(u16, uint16_t)
REF_ELEV_API(i32, int32_t)
REF_ELEV_API(u32, uint32_t)
REF_ELEV_API(f32, float)
REF_ELEV_API(f64, double)
REF_ELEV_API(i64, int64_t)
REF_ELEV_API(u64, uint64_t)

extern "C" void ref_d8_flow_accum_i32(const uint8_t *dirs, uint8_t nodata, int w, int h, int32_t *out) {
  ref_d8_flow_accum<int32_t>(dirs, nodata, w, h, out);
}
extern "C" void ref_d8_flow_accum_f32(const uint8_t *dirs, uint8_t nodata, int w, int h, float *out) {
  ref_d8_flow_accum<float>(dirs, nodata, w, h, out);
}
extern "C" void ref_d8_flow_accum_f64(const uint8_t *dirs, uint8_t nodata, int w, int h, double *out) {
  ref_d8_flow_accum<double>(dirs, nodata, w, h, out);
}

// Generic FlowAccumulation over a caller-supplied 9-float proportions array
// (methods/flow_accumulation_generic.hpp:33-100).  Array3D has no wrap
// constructor, so the proportions are copied in.
extern "C" void ref_flow_accumulation_f64(const float *props9, int w, int h, double *accum) {
  Array2D<double> acc(accum, w, h);
  Array3D<float> props(acc);
  std::memcpy(props.getData(), props9, (size_t)w * h * 9 * sizeof(float));
  props.setNoData(NO_DATA_GEN);
  FlowAccumulation(props, acc);
}

// native raster format, common/Array2D.hpp:209-281 (saveToCache / Array2D(filename, native=true))
template <class T>
int ref_native_save(const char *path, const T *data, int w, int h, T nodata, const double *gt6, const char *proj) {
  try {
    Array2D<T> a(const_cast<T *>(data), w, h);
    a.setNoData(nodata);
    a.geotransform.assign(gt6, gt6 + 6);
    a.projection = proj;
    a.saveToCache(path);
    return 0;
  } catch (...) { return 1; }
}
template <class T>
int ref_native_load(const char *path, T *data, int *w, int *h, T *nodata, double *gt6, char *proj, int projcap) {
  try {
    Array2D<T> a(std::string(path), true);
    if (data && (a.width() != *w || a.height() != *h)) return 2;
    *w = a.width(); *h = a.height(); *nodata = a.noData();
    for (int i = 0; i < 6; i++) gt6[i] = a.geotransform[i];
    std::snprintf(proj, projcap, "%s", a.projection.c_str());
    if (data) std::memcpy(data, a.data(), (size_t)a.width() * a.height() * sizeof(T));
    return 0;
  } catch (...) { return 1; }
}
#define REF_NATIVE_API(SUF, T)                                                                                        \
  extern "C" int ref_native_save_##SUF(const char *path, const T *data, int w, int h, T nodata, const double *gt6,    \
                                       const char *proj) { return ref_native_save<T>(path, data, w, h, nodata, gt6, proj); } \
  extern "C" int ref_native_load_##SUF(const char *path, T *data, int *w, int *h, T *nodata, double *gt6, char *proj, \
                                       int projcap) { return ref_native_load<T>(path, data, w, h, nodata, gt6, proj, projcap); }
REF_NATIVE_API(u8, uint8_t)
REF_NATIVE_API(i32, int32_t)
REF_NATIVE_API(f32, float)
REF_NATIVE_API(f64, double)

// ResolveFlatsEpsilon, flats/flats.hpp:21-28 (what rd.ResolveFlats calls, pywrapper.hpp:37)
template <class T>
void ref_resolve_flats_epsilon(T *dem, T nodata, int w, int h) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  ResolveFlatsEpsilon(a);
}
#define REF_RFE_API(SUF, T) \
  extern "C" void ref_resolve_flats_epsilon_##SUF(T *dem, T nodata, int w, int h) { ref_resolve_flats_epsilon<T>(dem, nodata, w, h); }
REF_RFE_API(u8, uint8_t)
REF_RFE_API(i8, int8_t)
REF_RFE_API(i64, int64_t)
REF_RFE_API(u64, uint64_t)
REF_RFE_API(i16, int16_t)
REF_RFE_API(u16, uint16_t)
REF_RFE_API(i32, int32_t)
REF_RFE_API(u32, uint32_t)
REF_RFE_API(f32, float)
REF_RFE_API(f64, double)

// pit_mask<topo>, depressions/Barnes2014.hpp:593-676 (apps/rd_depressions_mask.cpp)
template <class T>
void ref_pit_mask(const T *dem, T nodata, int w, int h, int topo, uint8_t *out) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  a.setNoData(nodata);
  Array2D<uint8_t> m;
  if (topo == 8) pit_mask<Topology::D8>(a, m);
  else pit_mask<Topology::D4>(a, m);
  std::memcpy(out, m.data(), (size_t)w * h);
}
#define REF_PM_API(SUF, T) \
  extern "C" void ref_pit_mask_##SUF(const T *dem, T nodata, int w, int h, int topo, uint8_t *out) { ref_pit_mask<T>(dem, nodata, w, h, topo, out); }
REF_PM_API(u8, uint8_t)
REF_PM_API(i8, int8_t)
REF_PM_API(i16, int16_t)
REF_PM_API(u16, uint16_t)
REF_PM_API(i32, int32_t)
REF_PM_API(u32, uint32_t)
REF_PM_API(f32, float)
REF_PM_API(f64, double)

// ---- the other outputs of the Priority-Flood sweep -------------------------------------------------
// PriorityFloodEpsilon_Barnes2014<topo> (depressions/Barnes2014.hpp:335-420; floating point only, :424-451)
template <class T>
void ref_fill_epsilon(T *dem, T nodata, int w, int h, int topo) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  if (topo == 8) PriorityFloodEpsilon_Barnes2014<Topology::D8>(a);
  else PriorityFloodEpsilon_Barnes2014<Topology::D4>(a);
}
extern "C" void ref_fill_epsilon_f32(float *dem, float nodata, int w, int h, int topo) { ref_fill_epsilon<float>(dem, nodata, w, h, topo); }
extern "C" void ref_fill_epsilon_f64(double *dem, double nodata, int w, int h, int topo) { ref_fill_epsilon<double>(dem, nodata, w, h, topo); }

// PriorityFloodWatersheds_Barnes2014<topo>(elevations, labels, alter_elevations) (:713-807)
template <class T>
void ref_watersheds(T *dem, T nodata, int w, int h, int topo, int alter, int32_t *labels) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  Array2D<int32_t> lab;
  if (topo == 8) PriorityFloodWatersheds_Barnes2014<Topology::D8>(a, lab, alter != 0);
  else PriorityFloodWatersheds_Barnes2014<Topology::D4>(a, lab, alter != 0);
  std::memcpy(labels, lab.data(), (size_t)w * h * sizeof(int32_t));
}
// PriorityFlood_Barnes2014_max_dep<topo>(elevations, max_dep_size) (:844-931; apps/rd_depressions_flood.cpp:16-19)
template <class T>
void ref_fill_max_dep(T *dem, int w, int h, int topo, uint64_t max_dep_size) {
  Array2D<T> a(dem, w, h);
  if (topo == 8) PriorityFlood_Barnes2014_max_dep<Topology::D8>(a, max_dep_size);
  else PriorityFlood_Barnes2014_max_dep<Topology::D4>(a, max_dep_size);
}
// PriorityFloodFlowdirs_Barnes2014(elevations, flowdirs) (:483-555)
template <class T>
void ref_pf_flowdirs(T *dem, T nodata, int w, int h, uint8_t *dirs) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  Array2D<d8_flowdir_t> fd;
  PriorityFloodFlowdirs_Barnes2014(a, fd);
  std::memcpy(dirs, fd.data(), (size_t)w * h);
}
// PriorityFlood_Wei2018 with a NoData value (Wei2018.hpp:154-202; InitPriorityQue :14-50 seeds the neighbours of NoData cells),
// PriorityFlood_Original<topo> (Barnes2014.hpp:136-198), HasDepressions<topo> (:44-103)
template <class T>
void ref_fill_wei2018(T *dem, T nodata, int w, int h) {
  Array2D<T> a(dem, w, h);
  a.setNoData(nodata);
  PriorityFlood_Wei2018(a);
}
template <class T>
void ref_fill_original(T *dem, int w, int h, int topo) {
  Array2D<T> a(dem, w, h);
  if (topo == 8) PriorityFlood_Original<Topology::D8>(a);
  else PriorityFlood_Original<Topology::D4>(a);
}
template <class T>
int ref_has_depressions(const T *dem, int w, int h, int topo) {
  Array2D<T> a(const_cast<T *>(dem), w, h);
  return (topo == 8 ? HasDepressions<Topology::D8>(a) : HasDepressions<Topology::D4>(a)) ? 1 : 0;
}
#define REF_VARIANTS_API(SUF, T)                                                                                              \
  extern "C" void ref_fill_wei2018_##SUF(T *dem, T nodata, int w, int h) { ref_fill_wei2018<T>(dem, nodata, w, h); }          \
  extern "C" void ref_fill_original_##SUF(T *dem, int w, int h, int topo) { ref_fill_original<T>(dem, w, h, topo); }          \
  extern "C" int ref_has_depressions_##SUF(const T *dem, int w, int h, int topo) { return ref_has_depressions<T>(dem, w, h, topo); }
REF_VARIANTS_API(u8, uint8_t)
REF_VARIANTS_API(i8, int8_t)
REF_VARIANTS_API(i16, int16_t)
REF_VARIANTS_API(u16, uint16_t)
REF_VARIANTS_API(i32, int32_t)
REF_VARIANTS_API(u32, uint32_t)
REF_VARIANTS_API(f32, float)
REF_VARIANTS_API(f64, double)
REF_VARIANTS_API(i64, int64_t)
REF_VARIANTS_API(u64, uint64_t)
#define REF_F2_API(SUF, T)                                                                                                    \
  extern "C" void ref_watersheds_##SUF(T *dem, T nodata, int w, int h, int topo, int alter, int32_t *labels) {                \
    ref_watersheds<T>(dem, nodata, w, h, topo, alter, labels);                                                                \
  }                                                                                                                           \
  extern "C" void ref_fill_max_dep_##SUF(T *dem, int w, int h, int topo, uint64_t max_dep_size) {                             \
    ref_fill_max_dep<T>(dem, w, h, topo, max_dep_size);                                                                       \
  }                                                                                                                           \
  extern "C" void ref_pf_flowdirs_##SUF(T *dem, T nodata, int w, int h, uint8_t *dirs) { ref_pf_flowdirs<T>(dem, nodata, w, h, dirs); }
REF_F2_API(u8, uint8_t)
REF_F2_API(i8, int8_t)
REF_F2_API(i16, int16_t)
REF_F2_API(u16, uint16_t)
REF_F2_API(i32, int32_t)
REF_F2_API(u32, uint32_t)
REF_F2_API(f32, float)
REF_F2_API(f64, double)
