// rdgpu/richdem_gpu.hpp -- C++ shim: the reference's function names and Array2D<T> signatures over
// the C-ABI of librdgpu.so (include/rdgpu.h).  Host code stays C++; HIP lives only behind the C-ABI.
//
// Every function is templated on the ARRAY type, so the same shim binds to
//   * richdem::Array2D<T>  -- inside the reference tree (apps/rd_depressions_flood.cpp,
//                             apps/rd_flow_accumulation.cpp, apps/rd_d8_flowdirs.cpp compile unchanged
//                             apart from the namespace of the call), and
//   * rdgpu::Array2D<T>    -- the stand-alone container in rdgpu/Array2D.hpp.
// Required of the array type: data(), width(), height(), noData(), setNoData(), resize(other, val),
// templateCopy(other) -- all present in the reference's Array2D (common/Array2D.hpp).
//
// Contracts kept from the reference:
//   * all functions are void, modify / size their outputs exactly as the reference does, and report
//     failure by throwing std::runtime_error (the C-ABI's non-zero codes are converted here);
//   * the DEM buffer is never freed or reallocated: results are copied back into data();
//   * works for owning and for wrapping (externally owned) arrays.
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../rdgpu.h"

namespace rdgpu {
namespace detail {

inline void check(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + rdgpu_last_error());
}

template <class A>
using elem_t = typename std::remove_cv<typename std::remove_pointer<decltype(std::declval<A &>().data())>::type>::type;

[[noreturn]] inline void unsupported(const char *fn) {
  throw std::runtime_error(std::string(fn) + ": element type not supported by the MI355X engine");
}

// ---- overload sets: element type -> C-ABI entry point ---------------------------------------
#define RDGPU_SHIM_ELEV(SUF, T)                                                                            \
  inline int c_fill(T *p, int w, int h, int t) { return rdgpu_fill_##SUF(p, w, h, t); }
RDGPU_SHIM_ELEV(u8, uint8_t)
RDGPU_SHIM_ELEV(i16, int16_t)
RDGPU_SHIM_ELEV(u16, uint16_t)
RDGPU_SHIM_ELEV(i32, int32_t)
RDGPU_SHIM_ELEV(u32, uint32_t)
RDGPU_SHIM_ELEV(f32, float)
RDGPU_SHIM_ELEV(f64, double)
RDGPU_SHIM_ELEV(i64, int64_t)
RDGPU_SHIM_ELEV(u64, uint64_t)
RDGPU_SHIM_ELEV(i8, int8_t)
#undef RDGPU_SHIM_ELEV
template <class T>
int c_fill(T *, int, int, int) { unsupported("FillDepressions"); }

#define RDGPU_SHIM_STENCIL(SUF, T)                                                                         \
  inline int c_flowdirs(const T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_d8_flowdirs_##SUF(p, nd, w, h, o); } \
  inline int c_flatres(const T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_flat_resolution_d8_##SUF(p, nd, w, h, o); } \
  inline int c_fa_d8(const T *p, T nd, int w, int h, double *a) { return rdgpu_fa_d8_##SUF(p, nd, w, h, a); } \
  inline int c_fa_d8_unit(const T *p, T nd, int w, int h, double *a) { return rdgpu_fa_d8_unit_##SUF(p, nd, w, h, a); } \
  inline int c_fa_dinf(const T *p, T nd, int w, int h, double *a) { return rdgpu_fa_tarboton_##SUF(p, nd, w, h, a); } \
  inline int c_fa_mfd(const T *p, T nd, int w, int h, int m, double x, double *a) { return rdgpu_fa_mfd_##SUF(p, nd, w, h, m, x, a); } \
  inline int c_rfe(T *p, T nd, int w, int h) { return rdgpu_resolve_flats_epsilon_##SUF(p, nd, w, h); } \
  inline int c_dinf(const T *p, T nd, int w, int h, float *o) { return rdgpu_dinf_flowdirs_##SUF(p, nd, w, h, o); } \
  inline int c_fm_d8(const T *p, T nd, int w, int h, float *o) { return rdgpu_fm_d8_##SUF(p, nd, w, h, o); } \
  inline int c_fm_dinf(const T *p, T nd, int w, int h, float *o) { return rdgpu_fm_tarboton_##SUF(p, nd, w, h, o); } \
  inline int c_fm_mfd(const T *p, T nd, int w, int h, int m, double x, float *o) { return rdgpu_fm_mfd_##SUF(p, nd, w, h, m, x, o); }
RDGPU_SHIM_STENCIL(u8, uint8_t)
RDGPU_SHIM_STENCIL(i16, int16_t)
RDGPU_SHIM_STENCIL(u16, uint16_t)
RDGPU_SHIM_STENCIL(i32, int32_t)
RDGPU_SHIM_STENCIL(u32, uint32_t)
RDGPU_SHIM_STENCIL(f32, float)
RDGPU_SHIM_STENCIL(f64, double)
RDGPU_SHIM_STENCIL(i8, int8_t)
#undef RDGPU_SHIM_STENCIL
#define RDGPU_SHIM_STENCIL64(SUF, T)                                                                       \
  inline int c_flowdirs(const T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_d8_flowdirs_##SUF(p, nd, w, h, o); } \
  inline int c_flatres(const T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_flat_resolution_d8_##SUF(p, nd, w, h, o); } \
  inline int c_fa_d8(const T *p, T nd, int w, int h, double *a) { return rdgpu_fa_d8_##SUF(p, nd, w, h, a); } \
  inline int c_fa_d8_unit(const T *p, T nd, int w, int h, double *a) { return rdgpu_fa_d8_unit_##SUF(p, nd, w, h, a); } \
  inline int c_rfe(T *p, T nd, int w, int h) { return rdgpu_resolve_flats_epsilon_##SUF(p, nd, w, h); } \
  inline int c_fm_d8(const T *p, T nd, int w, int h, float *o) { return rdgpu_fm_d8_##SUF(p, nd, w, h, o); }
RDGPU_SHIM_STENCIL64(i64, int64_t)
RDGPU_SHIM_STENCIL64(u64, uint64_t)
#undef RDGPU_SHIM_STENCIL64
template <class T>
int c_flowdirs(const T *, T, int, int, uint8_t *) { unsupported("d8_flow_directions"); }
template <class T>
int c_flatres(const T *, T, int, int, uint8_t *) { unsupported("barnes_flat_resolution_d8"); }
template <class T>
int c_fa_d8(const T *, T, int, int, double *) { unsupported("FA_D8"); }
template <class T>
int c_fa_d8_unit(const T *, T, int, int, double *) { unsupported("FA_D8"); }
template <class T>
int c_fa_dinf(const T *, T, int, int, double *) { unsupported("FA_Tarboton"); }
template <class T>
int c_dinf(const T *, T, int, int, float *) { unsupported("dinf_flow_directions"); }
template <class T>
int c_fa_mfd(const T *, T, int, int, int, double, double *) { unsupported("FA_Holmgren / FA_Freeman / FA_Quinn / FA_D4"); }
template <class T>
int c_rfe(T *, T, int, int) { unsupported("ResolveFlatsEpsilon"); }
template <class T>
int c_fm_d8(const T *, T, int, int, float *) { unsupported("FM_D8"); }
template <class T>
int c_fm_dinf(const T *, T, int, int, float *) { unsupported("FM_Tarboton"); }
template <class T>
int c_fm_mfd(const T *, T, int, int, int, double, float *) { unsupported("FM_Holmgren / FM_Freeman / FM_Quinn / FM_D4"); }

#define RDGPU_SHIM_VARIANTS(SUF, T)                                                                          \
  inline int c_hasdep(const T *p, int w, int h, int t, int *o) { return rdgpu_has_depressions_##SUF(p, w, h, t, o); } \
  inline int c_fill_wei(T *p, T nd, int w, int h) { return rdgpu_fill_wei2018_##SUF(p, nd, w, h); }
RDGPU_SHIM_VARIANTS(u8, uint8_t) RDGPU_SHIM_VARIANTS(i8, int8_t) RDGPU_SHIM_VARIANTS(i16, int16_t)
RDGPU_SHIM_VARIANTS(u16, uint16_t) RDGPU_SHIM_VARIANTS(i32, int32_t) RDGPU_SHIM_VARIANTS(u32, uint32_t)
RDGPU_SHIM_VARIANTS(f32, float) RDGPU_SHIM_VARIANTS(f64, double) RDGPU_SHIM_VARIANTS(i64, int64_t)
RDGPU_SHIM_VARIANTS(u64, uint64_t)
#undef RDGPU_SHIM_VARIANTS
template <class T>
int c_hasdep(const T *, int, int, int, int *) { unsupported("HasDepressions"); }
template <class T>
int c_fill_wei(T *, T, int, int) { unsupported("PriorityFlood_Wei2018"); }

#define RDGPU_SHIM_PITMASK(SUF, T) \
  inline int c_pitmask(const T *p, T nd, int w, int h, int topo, uint8_t *m) { return rdgpu_pit_mask_##SUF(p, nd, w, h, topo, m); }
RDGPU_SHIM_PITMASK(u8, uint8_t)
RDGPU_SHIM_PITMASK(i16, int16_t)
RDGPU_SHIM_PITMASK(u16, uint16_t)
RDGPU_SHIM_PITMASK(i32, int32_t)
RDGPU_SHIM_PITMASK(u32, uint32_t)
RDGPU_SHIM_PITMASK(f32, float)
RDGPU_SHIM_PITMASK(i8, int8_t)
RDGPU_SHIM_PITMASK(f64, double)
RDGPU_SHIM_PITMASK(i64, int64_t)
RDGPU_SHIM_PITMASK(u64, uint64_t)
#undef RDGPU_SHIM_PITMASK
template <class T>
int c_pitmask(const T *, T, int, int, int, uint8_t *) { unsupported("pit_mask"); }

#define RDGPU_SHIM_MAXDEP(SUF, T) \
  inline int c_fill_maxdep(T *p, int w, int h, int t, uint64_t m) { return rdgpu_fill_max_dep_##SUF(p, w, h, t, m); }
RDGPU_SHIM_MAXDEP(u8, uint8_t)
RDGPU_SHIM_MAXDEP(i16, int16_t)
RDGPU_SHIM_MAXDEP(u16, uint16_t)
RDGPU_SHIM_MAXDEP(i32, int32_t)
RDGPU_SHIM_MAXDEP(u32, uint32_t)
RDGPU_SHIM_MAXDEP(f32, float)
RDGPU_SHIM_MAXDEP(i8, int8_t)
RDGPU_SHIM_MAXDEP(f64, double)
RDGPU_SHIM_MAXDEP(i64, int64_t)
RDGPU_SHIM_MAXDEP(u64, uint64_t)
#undef RDGPU_SHIM_MAXDEP
template <class T>
int c_fill_maxdep(T *, int, int, int, uint64_t) { unsupported("PriorityFlood_Barnes2014_max_dep"); }

#define RDGPU_SHIM_WS(SUF, T) \
  inline int c_watersheds(T *p, T nd, int w, int h, int t, int alter, int32_t *l) { return rdgpu_watersheds_##SUF(p, nd, w, h, t, alter, l); }
RDGPU_SHIM_WS(u8, uint8_t)
RDGPU_SHIM_WS(i16, int16_t)
RDGPU_SHIM_WS(u16, uint16_t)
RDGPU_SHIM_WS(i32, int32_t)
RDGPU_SHIM_WS(u32, uint32_t)
RDGPU_SHIM_WS(f32, float)
RDGPU_SHIM_WS(i8, int8_t)
RDGPU_SHIM_WS(f64, double)
RDGPU_SHIM_WS(i64, int64_t)
RDGPU_SHIM_WS(u64, uint64_t)
#undef RDGPU_SHIM_WS
template <class T>
int c_watersheds(T *, T, int, int, int, int, int32_t *) { unsupported("PriorityFloodWatersheds_Barnes2014"); }

#define RDGPU_SHIM_PFD(SUF, T) \
  inline int c_pf_flowdirs(const T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_pf_flowdirs_##SUF(p, nd, w, h, o); }
RDGPU_SHIM_PFD(u8, uint8_t) RDGPU_SHIM_PFD(i8, int8_t) RDGPU_SHIM_PFD(i16, int16_t) RDGPU_SHIM_PFD(u16, uint16_t)
RDGPU_SHIM_PFD(i32, int32_t) RDGPU_SHIM_PFD(u32, uint32_t) RDGPU_SHIM_PFD(f32, float)
RDGPU_SHIM_PFD(f64, double) RDGPU_SHIM_PFD(i64, int64_t) RDGPU_SHIM_PFD(u64, uint64_t)
#undef RDGPU_SHIM_PFD
template <class T>
int c_pf_flowdirs(const T *, T, int, int, uint8_t *) { unsupported("PriorityFloodFlowdirs_Barnes2014"); }

inline int c_fill_eps(float *p, float nd, int w, int h, int t) { return rdgpu_fill_epsilon_f32(p, nd, w, h, t); }
inline int c_fill_eps(double *p, double nd, int w, int h, int t) { return rdgpu_fill_epsilon_f64(p, nd, w, h, t); }
template <class T>
int c_fill_eps(T *, T, int, int, int) {   // depressions/Barnes2014.hpp:424-451
  throw std::runtime_error("Priority-Flood+Epsilon is only available for floating-point data types!");
}

#define RDGPU_SHIM_ALTER(SUF, T) \
  inline int c_flatres_alter(T *p, T nd, int w, int h, uint8_t *o) { return rdgpu_flat_resolution_d8_alter_##SUF(p, nd, w, h, o); }
RDGPU_SHIM_ALTER(f32, float) RDGPU_SHIM_ALTER(f64, double) RDGPU_SHIM_ALTER(u8, uint8_t) RDGPU_SHIM_ALTER(i8, int8_t)
RDGPU_SHIM_ALTER(i16, int16_t) RDGPU_SHIM_ALTER(u16, uint16_t) RDGPU_SHIM_ALTER(i32, int32_t) RDGPU_SHIM_ALTER(u32, uint32_t)
RDGPU_SHIM_ALTER(i64, int64_t) RDGPU_SHIM_ALTER(u64, uint64_t)
#undef RDGPU_SHIM_ALTER

inline int c_accum(const uint8_t *d, uint8_t nd, int w, int h, int32_t *a) { return rdgpu_d8_flow_accum_i32(d, nd, w, h, a); }
inline int c_accum(const uint8_t *d, uint8_t nd, int w, int h, float *a) { return rdgpu_d8_flow_accum_f32(d, nd, w, h, a); }
inline int c_accum(const uint8_t *d, uint8_t nd, int w, int h, double *a) { return rdgpu_d8_flow_accum_f64(d, nd, w, h, a); }
template <class A>
int c_accum(const uint8_t *, uint8_t, int, int, A *) { unsupported("d8_flow_accum"); }

// Topology enumerators: richdem::Topology and rdgpu::Topology both declare {D8, D4} in this order
// (reference common/constants.hpp:97-100).
template <auto topo>
constexpr int topology_code() {
  static_assert(std::is_enum<decltype(topo)>::value, "topology must be Topology::D8 or Topology::D4");
  return static_cast<int>(topo) == 0 ? 8 : 4;
}

}  // namespace detail

// ---- depressions -------------------------------------------------------------------------------
// richdem::PriorityFlood_Zhou2016(Array2D<T>&)            depressions/Zhou2016.hpp:126-191
template <class A>
void PriorityFlood_Zhou2016(A &dem) {
  detail::check(detail::c_fill(dem.data(), dem.width(), dem.height(), 8), "PriorityFlood_Zhou2016");
}

// richdem::PriorityFlood_Barnes2014<topo>(Array2D<T>&)    depressions/Barnes2014.hpp:230-304
template <auto topo, class A>
void PriorityFlood_Barnes2014(A &dem) {
  detail::check(detail::c_fill(dem.data(), dem.width(), dem.height(), detail::topology_code<topo>()),
                "PriorityFlood_Barnes2014");
}

// richdem::FillDepressions<topo>(Array2D<T>&)             depressions/depressions.hpp:13-21
template <auto topo, class A>
void FillDepressions(A &dem) {
  detail::check(detail::c_fill(dem.data(), dem.width(), dem.height(), detail::topology_code<topo>()), "FillDepressions");
}

// richdem::PriorityFlood_Original<topo>(Array2D<T>&)      depressions/Barnes2014.hpp:136-198: the surface of
// FillDepressions<topo> (every flood of this family returns it, tests/tests.cpp:233-271)
template <auto topo, class A>
void PriorityFlood_Original(A &dem) {
  detail::check(detail::c_fill(dem.data(), dem.width(), dem.height(), detail::topology_code<topo>()), "PriorityFlood_Original");
}

// richdem::PriorityFlood_Wei2018(Array2D<T>&)             depressions/Wei2018.hpp:154-202: D8; NoData cells are left alone
// and the data cells next to them are seeds beside the raster's edge cells (InitPriorityQue, :14-50)
template <class A>
void PriorityFlood_Wei2018(A &dem) {
  using T = detail::elem_t<A>;
  if (dem.width() == 0 || dem.height() == 0) return;
  detail::check(detail::c_fill_wei((T *)dem.data(), dem.noData(), dem.width(), dem.height()), "PriorityFlood_Wei2018");
}

// richdem::HasDepressions<topo>(const Array2D<T>&)        depressions/Barnes2014.hpp:44-103 (apps/rd_depressions_has.cpp:14)
template <auto topo, class A>
bool HasDepressions(const A &elevations) {
  if (elevations.width() == 0 || elevations.height() == 0) return false;   // (an empty queue: "No depressions found.")
  int found = 0;
  detail::check(detail::c_hasdep(elevations.data(), elevations.width(), elevations.height(), detail::topology_code<topo>(), &found),
                "HasDepressions");
  return found != 0;
}

// richdem::PriorityFlood_Barnes2014_max_dep<topo>(Array2D<T>&, uint64_t max_dep_size)   depressions/Barnes2014.hpp:844-931
// (apps/rd_depressions_flood.cpp:16-19): only depressions of at most max_dep_size cells are filled
template <auto topo, class A>
void PriorityFlood_Barnes2014_max_dep(A &dem, uint64_t max_dep_size) {
  using T = detail::elem_t<A>;
  if (dem.width() == 0 || dem.height() == 0) return;
  detail::check(detail::c_fill_maxdep((T *)dem.data(), dem.width(), dem.height(), detail::topology_code<topo>(), max_dep_size),
                "PriorityFlood_Barnes2014_max_dep");
}

// richdem::PriorityFloodWatersheds_Barnes2014<topo>(Array2D<T>&, Array2D<int32_t>&, bool alter_elevations)
// depressions/Barnes2014.hpp:713-807: labels resized to the DEM, NoData -1 (:737-738)
template <auto topo, class A, class L>
void PriorityFloodWatersheds_Barnes2014(A &elevations, L &labels, bool alter_elevations) {
  using T = detail::elem_t<A>;
  static_assert(std::is_same<detail::elem_t<L>, int32_t>::value, "PriorityFloodWatersheds_Barnes2014: labels must be Array2D<int32_t>");
  labels.resize(elevations.width(), elevations.height(), -1);
  labels.setNoData(-1);
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::check(detail::c_watersheds((T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(),
                                     detail::topology_code<topo>(), alter_elevations ? 1 : 0, labels.data()),
                "PriorityFloodWatersheds_Barnes2014");
}

// richdem::PriorityFloodEpsilon_Barnes2014<topo>(Array2D<T>&)   depressions/Barnes2014.hpp:335-420; integer element
// types throw std::runtime_error as the reference's specialisations do (:424-451)
template <auto topo, class A>
void PriorityFloodEpsilon_Barnes2014(A &dem) {
  using T = detail::elem_t<A>;
  if (dem.width() == 0 || dem.height() == 0) return;
  detail::check(detail::c_fill_eps((T *)dem.data(), dem.noData(), dem.width(), dem.height(), detail::topology_code<topo>()),
                "PriorityFloodEpsilon_Barnes2014");
}

// richdem::FillDepressionsEpsilon<topo>(Array2D<T>&)      depressions/depressions.hpp:23
template <auto topo, class A>
void FillDepressionsEpsilon(A &dem) {
  PriorityFloodEpsilon_Barnes2014<topo>(dem);
}

// ---- flow directions ---------------------------------------------------------------------------
namespace detail {
template <class E, class F, class Fn>
void dirs_into(const E &elevations, F &flowdirs, Fn fn, const char *who) {
  using U = elem_t<F>;
  flowdirs.resize(elevations);          // d8_flowdirs.hpp:107
  flowdirs.setNoData((U)255);           // FLOWDIR_NO_DATA, d8_flowdirs.hpp:109
  const int w = elevations.width(), h = elevations.height();
  if (w == 0 || h == 0) return;
  if constexpr (std::is_same<U, uint8_t>::value) {
    check(fn(elevations.data(), elevations.noData(), w, h, flowdirs.data()), who);
  } else {  // the reference allows any integer U: compute in u8, widen
    std::vector<uint8_t> tmp((size_t)w * h);
    check(fn(elevations.data(), elevations.noData(), w, h, tmp.data()), who);
    for (size_t i = 0; i < tmp.size(); i++) flowdirs.data()[i] = (U)tmp[i];
  }
}
}  // namespace detail

// richdem::PriorityFloodFlowdirs_Barnes2014(const Array2D<T>&, Array2D<d8_flowdir_t>&)   depressions/Barnes2014.hpp:483-555
// flowdirs resized to the DEM, NoData = NO_FLOW (0) (:495-496).  Identical to the reference, equal elevations included
// (rdgpu.h; rdgpu_pf_flowdirs_get_stats says how many passes the tie order took).
template <class E, class F>
void PriorityFloodFlowdirs_Barnes2014(const E &elevations, F &flowdirs) {
  static_assert(std::is_same<detail::elem_t<F>, uint8_t>::value, "PriorityFloodFlowdirs_Barnes2014: flowdirs must be Array2D<d8_flowdir_t>");
  flowdirs.resize(elevations.width(), elevations.height());
  flowdirs.setNoData((uint8_t)0);
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::check(detail::c_pf_flowdirs(elevations.data(), elevations.noData(), elevations.width(), elevations.height(), flowdirs.data()),
                "PriorityFloodFlowdirs_Barnes2014");
  rdgpu_pf_flowdirs_stats st;
  if (rdgpu_pf_flowdirs_get_stats(&st) == 0 && st.unresolved != 0)   // (the reference's RDLOG_WARN channel is stderr too)
    std::fprintf(stderr, "W PriorityFloodFlowdirs_Barnes2014: the order of %llu of %u equal-elevation cells had not settled when the "
                         "tie-order passes were stopped (RDGPU_PFD_TIE_PASSES / RDGPU_PFD_TIE_SECONDS); directions are the "
                         "reference's only where those ties do not decide\n", (unsigned long long)st.unresolved, st.twins);
}

// richdem::d8_flow_directions(const Array2D<T>&, Array2D<U>&)   flowmet/d8_flowdirs.hpp:96-123
template <class E, class F>
void d8_flow_directions(const E &elevations, F &flowdirs) {
  using T = detail::elem_t<E>;
  detail::dirs_into(elevations, flowdirs,
                    [](const T *p, T nd, int w, int h, uint8_t *o) { return detail::c_flowdirs(p, nd, w, h, o); },
                    "d8_flow_directions");
}

// richdem::barnes_flat_resolution_d8(Array2D<T>&, Array2D<U>&, bool alter)   flats/flat_resolution.hpp:587-605
template <class E, class F>
void barnes_flat_resolution_d8(E &elevations, F &flowdirs, bool alter) {
  using T = detail::elem_t<E>;
  if (alter) {   // flat_resolution.hpp:597-600: the DEM itself is raised, then plain D8 directions
    using U = detail::elem_t<F>;
    flowdirs.resize(elevations);
    flowdirs.setNoData((U)255);
    if (elevations.width() > 0 && elevations.height() > 0) {
      std::vector<uint8_t> tmp((size_t)elevations.width() * elevations.height());
      detail::check(detail::c_flatres_alter(elevations.data(), elevations.noData(), elevations.width(),
                                            elevations.height(), tmp.data()),
                    "barnes_flat_resolution_d8");
      for (size_t i = 0; i < tmp.size(); i++) flowdirs.data()[i] = (U)tmp[i];
    }
    flowdirs.templateCopy(elevations);
    return;
  }
  detail::dirs_into(elevations, flowdirs,
                    [](const T *p, T nd, int w, int h, uint8_t *o) { return detail::c_flatres(p, nd, w, h, o); },
                    "barnes_flat_resolution_d8");
  flowdirs.templateCopy(elevations);    // flat_resolution.hpp:604
}

// accum_t other than double (the reference's FA_* / FlowAccumulation are templated on it,
// methods/flow_accumulation.hpp:14-28, flow_accumulation_generic.hpp:33-34): the engine accumulates in double; an
// Array2D<float> / Array2D<int32_t> / ... accumulation array is staged through a double copy (weights in, result out,
// converted as a C++ assignment would).  Equal to the reference whenever every partial sum is exactly representable in
// accum_t -- unit or integer-valued weights with totals below 2^24 for float and (because the reference forms
// `float proportion * accum_t` before it adds) for the integer types as well; beyond that the reference's own value is a
// product of float rounding in its queue's order.
namespace detail {
template <class G, class Fn>
void with_double_accum(G &accum, Fn fn) {
  using A = elem_t<G>;
  static_assert(std::is_arithmetic<A>::value, "the accumulation array must hold an arithmetic type");
  if constexpr (std::is_same<A, double>::value) {
    fn(accum.data());
  } else {
    const size_t n = (size_t)accum.width() * (size_t)accum.height();
    std::vector<double> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = (double)accum.data()[i];
    fn(tmp.data());
    for (size_t i = 0; i < n; i++) accum.data()[i] = (A)tmp[i];
  }
}
}  // namespace detail

// ---- accumulation ------------------------------------------------------------------------------
// richdem::d8_flow_accum(const Array2D<T>& flowdirs, Array2D<U>& area)   methods/d8_methods.hpp:47-139
template <class F, class G>
void d8_flow_accum(const F &flowdirs, G &area) {
  static_assert(std::is_same<detail::elem_t<const F>, uint8_t>::value || std::is_same<detail::elem_t<F>, uint8_t>::value,
                "d8_flow_accum: flow directions must be uint8_t (d8_flowdir_t)");
  using U = detail::elem_t<G>;
  area.resize(flowdirs, (U)0);          // d8_methods.hpp:63
  area.setNoData((U)-1);                // d8_methods.hpp:64
  if (flowdirs.width() == 0 || flowdirs.height() == 0) return;
  detail::check(detail::c_accum(flowdirs.data(), flowdirs.noData(), flowdirs.width(), flowdirs.height(), area.data()),
                "d8_flow_accum");
}

// richdem::FA_D8(const Array2D<elev_t>&, Array2D<accum_t>&)   methods/flow_accumulation.hpp:27
// accum is in/out: pre-loaded with the flow each cell generates.
template <class E, class G>
void FA_D8(const E &elevations, G &accum) {
  using T = detail::elem_t<E>;
  accum.setNoData((detail::elem_t<G>)-1);   // ACCUM_NO_DATA, flow_accumulation_generic.hpp:40
  if (accum.width() != elevations.width() || accum.height() != elevations.height())   // :42-43
    throw std::runtime_error("Accumulation array must have same dimensions as proportions array!");
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::with_double_accum(accum, [&](double *a) {
    detail::check(detail::c_fa_d8((const T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(), a),
                  "FA_D8");
  });
}

// richdem::FA_Tarboton / FA_Dinfinity(const Array2D<elev_t>&, Array2D<accum_t>&)   methods/flow_accumulation.hpp:16-17
template <class E, class G>
void FA_Tarboton(const E &elevations, G &accum) {
  using T = detail::elem_t<E>;
  accum.setNoData((detail::elem_t<G>)-1);
  if (accum.width() != elevations.width() || accum.height() != elevations.height())
    throw std::runtime_error("Accumulation array must have same dimensions as proportions array!");
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::with_double_accum(accum, [&](double *a) {
    detail::check(detail::c_fa_dinf((const T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(), a),
                  "FA_Tarboton");
  });
}
template <class E, class G>
void FA_Dinfinity(const E &elevations, G &accum) { FA_Tarboton(elevations, accum); }

// FA_D8 for a caller that built `accum` as an array of ones itself (apps/rd_flow_accumulation.cpp:13: Array2D<double>
// accum(dem, 1)) and says so: rdgpu::FA_D8(dem, accum, rdgpu::unit_weights).  accum is then output only -- nothing is read
// from it, nothing uploaded (rdgpu_fa_d8_unit_<T>); the result equals FA_D8 on an array of ones.
struct unit_weights_t {};
constexpr unit_weights_t unit_weights{};
template <class E, class G>
void FA_D8(const E &elevations, G &accum, unit_weights_t) {
  using T = detail::elem_t<E>;
  accum.setNoData((detail::elem_t<G>)-1);
  if (accum.width() != elevations.width() || accum.height() != elevations.height())
    throw std::runtime_error("Accumulation array must have same dimensions as proportions array!");
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::with_double_accum(accum, [&](double *a) {
    detail::check(detail::c_fa_d8_unit((const T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(), a),
                  "FA_D8");
  });
}

// richdem::FA_Holmgren / FA_Quinn / FA_Freeman / FA_D4   methods/flow_accumulation.hpp:18-20, :28
namespace detail {
template <class E, class G>
void fa_mfd(const E &elevations, G &accum, int method, double xparam, const char *who) {
  using T = elem_t<E>;
  accum.setNoData((elem_t<G>)-1);
  if (accum.width() != elevations.width() || accum.height() != elevations.height())
    throw std::runtime_error("Accumulation array must have same dimensions as proportions array!");
  if (elevations.width() == 0 || elevations.height() == 0) return;
  with_double_accum(accum, [&](double *a) {
    check(c_fa_mfd((const T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(), method, xparam, a), who);
  });
}
}  // namespace detail
template <class E, class G>
void FA_Holmgren(const E &elevations, G &accum, double xparam) { detail::fa_mfd(elevations, accum, 0, xparam, "FA_Holmgren"); }
template <class E, class G>
void FA_Freeman(const E &elevations, G &accum, double xparam) { detail::fa_mfd(elevations, accum, 1, xparam, "FA_Freeman"); }
template <class E, class G>
void FA_Quinn(const E &elevations, G &accum) { detail::fa_mfd(elevations, accum, 2, 1.0, "FA_Quinn"); }
template <class E, class G>
void FA_D4(const E &elevations, G &accum) { detail::fa_mfd(elevations, accum, 3, 1.0, "FA_D4"); }
template <class E, class G>
void FA_OCallaghanD4(const E &elevations, G &accum) { FA_D4(elevations, accum); }
template <class E, class G>
void FA_OCallaghanD8(const E &elevations, G &accum) { FA_D8(elevations, accum); }

// ---- flow proportions (the Array3D<float> on-the-wire format of rd.FlowProportions) ----------------------------
// richdem::FM_D8 / FM_OCallaghan<D8> (flowmet/OCallaghan1984.hpp:13-84), FM_D4 / FM_OCallaghan<D4> (:86),
// FM_Tarboton / FM_Dinfinity (flowmet/Tarboton1997.hpp:14-144), FM_Holmgren(x) (Holmgren1994.hpp:14),
// FM_Freeman(x) (Freeman1991.hpp:14), FM_Quinn (Quinn1991.hpp:13).  `props` is sized by the caller (nine float
// slots per cell, rdgpu::Array3D<float> or richdem::Array3D<float>); its NoData becomes NO_DATA_GEN = -2.
namespace detail {
// the slot buffer of a proportions array: richdem::Array3D has getData() (common/Array3D.hpp:165), rdgpu::Array3D data()
template <class P>
auto raw3(P &p, int) -> decltype(p.getData()) { return p.getData(); }
template <class P>
auto raw3(P &p, long) -> decltype(p.data()) { return p.data(); }

template <class E, class P, class Fn>
void fm_into(const E &elevations, P &props, Fn fn, const char *who) {
  static_assert(std::is_same<decltype(raw3(props, 0)), float *>::value, "FM_*: the proportions array must be Array3D<float>");
  props.setNoData(-2.0f);               // NO_DATA_GEN, common/constants.hpp:85
  if (props.width() != elevations.width() || props.height() != elevations.height())
    throw std::runtime_error(std::string(who) + ": proportions array must have the dimensions of the elevations");
  if (elevations.width() == 0 || elevations.height() == 0) return;
  check(fn(elevations.data(), elevations.noData(), elevations.width(), elevations.height(), raw3(props, 0)), who);
}
}  // namespace detail
template <class E, class P>
void FM_D8(const E &elevations, P &props) {
  using T = detail::elem_t<E>;
  detail::fm_into(elevations, props, [](const T *p, T nd, int w, int h, float *o) { return detail::c_fm_d8(p, nd, w, h, o); }, "FM_D8");
}
template <class E, class P>
void FM_Tarboton(const E &elevations, P &props) {
  using T = detail::elem_t<E>;
  detail::fm_into(elevations, props, [](const T *p, T nd, int w, int h, float *o) { return detail::c_fm_dinf(p, nd, w, h, o); }, "FM_Tarboton");
}
template <class E, class P>
void FM_Dinfinity(const E &elevations, P &props) { FM_Tarboton(elevations, props); }
namespace detail {
template <class E, class P>
void fm_mfd(const E &elevations, P &props, int method, double xparam, const char *who) {
  using T = elem_t<E>;
  fm_into(elevations, props,
          [method, xparam](const T *p, T nd, int w, int h, float *o) { return c_fm_mfd(p, nd, w, h, method, xparam, o); }, who);
}
}  // namespace detail
template <class E, class P>
void FM_Holmgren(const E &elevations, P &props, double xparam) { detail::fm_mfd(elevations, props, 0, xparam, "FM_Holmgren"); }
template <class E, class P>
void FM_Freeman(const E &elevations, P &props, double xparam) { detail::fm_mfd(elevations, props, 1, xparam, "FM_Freeman"); }
template <class E, class P>
void FM_Quinn(const E &elevations, P &props) { detail::fm_mfd(elevations, props, 2, 1.0, "FM_Quinn"); }
template <class E, class P>
void FM_D4(const E &elevations, P &props) { detail::fm_mfd(elevations, props, 3, 1.0, "FM_D4"); }
// FM_OCallaghan<topo>: Topology::D8 -> FM_D8, Topology::D4 -> FM_D4 (OCallaghan1984.hpp:81-90)
template <auto topo, class E, class P>
void FM_OCallaghan(const E &elevations, P &props) {
  if (detail::topology_code<topo>() == 8) FM_D8(elevations, props);
  else FM_D4(elevations, props);
}

// richdem::FlowAccumulation(const Array3D<float>&, Array2D<A>&)   methods/flow_accumulation_generic.hpp:33-100
// accum is in/out: pre-loaded with the flow each cell generates.
template <class P, class G>
void FlowAccumulation(const P &props, G &accum) {
  accum.setNoData((detail::elem_t<G>)-1);   // ACCUM_NO_DATA, :40
  if (accum.width() != props.width() || accum.height() != props.height())   // :42-43
    throw std::runtime_error("Accumulation array must have same dimensions as proportions array!");
  if (props.width() == 0 || props.height() == 0) return;
  const float *p9 = detail::raw3(const_cast<P &>(props), 0);   // (the reference's accessor is non-const; read only)
  detail::with_double_accum(accum, [&](double *a) {
    detail::check(rdgpu_flow_accumulation_f64(p9, props.width(), props.height(), a), "FlowAccumulation");
  });
}

// richdem::pit_mask<topo>(const Array2D<T>&, Array2D<uint8_t>&)   depressions/Barnes2014.hpp:593-676
template <auto topo, class E, class M>
void pit_mask(const E &elevations, M &mask) {
  using T = detail::elem_t<E>;
  static_assert(std::is_same<detail::elem_t<M>, uint8_t>::value, "pit_mask: the mask must be Array2D<uint8_t>");
  mask.resize(elevations.width(), elevations.height());   // :619
  mask.setNoData(3);                                       // :620
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::check(detail::c_pitmask((const T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height(),
                                  detail::topology_code<topo>(), mask.data()),
                "pit_mask");
}

// richdem::ResolveFlatsEpsilon(Array2D<T>&)   flats/flats.hpp:21-28 (pywrapper.hpp:37 rdResolveFlatsEpsilon)
template <class E>
void ResolveFlatsEpsilon(E &elevations) {
  using T = detail::elem_t<E>;
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::check(detail::c_rfe((T *)elevations.data(), elevations.noData(), elevations.width(), elevations.height()),
                "ResolveFlatsEpsilon");
}

// richdem::dinf_flow_directions(const Array2D<T>&, Array2D<float>&)   flowmet/dinf_flowdirs.hpp:128-152
template <class E, class F>
void dinf_flow_directions(const E &elevations, F &flowdirs) {
  using T = detail::elem_t<E>;
  static_assert(std::is_same<detail::elem_t<F>, float>::value, "dinf_flow_directions: flowdirs must be Array2D<float>");
  flowdirs.resize(elevations);          // :137
  flowdirs.setNoData(-1.0f);            // dinf_NO_DATA, :138
  if (elevations.width() == 0 || elevations.height() == 0) return;
  detail::check(detail::c_dinf((const T *)elevations.data(), elevations.noData(), elevations.width(),
                               elevations.height(), flowdirs.data()),
                "dinf_flow_directions");
}

}  // namespace rdgpu
