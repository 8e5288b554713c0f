// pywrapper_gpu.cpp -- the `_richdem` extension module of the reference's Python package, on the MI355X engine.
//
// The reference's wrapper (wrappers/pyrichdem/src/pywrapper.cpp, pywrapper.hpp) binds its header-only C++ library
// to Python as `_richdem`; `richdem/__init__.py` does `import _richdem` and nothing else touches C++.  This file
// builds a module with the same name and the same binding surface -- Array2D_<type>, Array3D_float, the rd*/FA_*/
// FM_* functions -- over the C++ shim (include/rdgpu/richdem_gpu.hpp -> librdgpu.so), so the reference's own
// `richdem/__init__.py` runs FillDepressions / FlowAccumulation / FlowProportions / FlowAccumFromProps /
// ResolveFlats on the GPU unchanged (the epsilon fill included).  What lies outside SURVEY.md section 8 (breaching, terrain
// attributes, the random Rho8/Rho4 family, terrain generation, the depression hierarchy) is bound too, and raises
// std::runtime_error("... outside the scope of the MI355X engine"): the surface is complete, the failure loud.
//
// Differences from the reference binding, on purpose:
//   * wrapping a numpy array never copies: dtype and C-contiguity must match (the reference's forcecast would wrap
//     a temporary), and the wrapper keeps the array alive;
//   * the GIL is released while the engine runs.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/stl_bind.h>

#include <cstdint>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <rdgpu/Array2D.hpp>
#include <rdgpu/Array3D.hpp>
#include <rdgpu/richdem_gpu.hpp>

namespace py = pybind11;
using rdgpu::Array2D;
using rdgpu::Array3D;
using rdgpu::Topology;

namespace {

[[noreturn]] void out_of_scope(const char *name) {
  throw std::runtime_error(std::string(name) +
                           ": outside the scope of the MI355X engine (depression filling, D8 directions and flat "
                           "resolution, D8 / D-infinity / MFD flow accumulation)");
}

// a function the reference binds but the engine does not implement: any arguments, always raises
void def_out_of_scope(py::module_ &m, const char *name) {
  if (py::hasattr(m, name)) return;   // bound once, not per element type
  std::string n = name;
  m.def(name, [n](const py::args &, const py::kwargs &) { out_of_scope(n.c_str()); }, "Outside the scope of the MI355X engine: raises.");
}

using release = py::call_guard<py::gil_scoped_release>;

// rd.FillDepressions(epsilon=True) on a DEM with equal elevations among its gradient sources (every integer-valued DEM):
// the reference's surface follows std::priority_queue's pop order there, the engine's is the order-free lower bound.  The
// device counts those sources; the caller is told how many (a RuntimeWarning, raised with the GIL held).
void warn_epsilon_ties() {
  rdgpu_epsilon_stats st;
  if (rdgpu_fill_epsilon_get_stats(&st) != 0 || st.tie_sources == 0) return;
  const std::string msg = "FillDepressions(epsilon=True): " + std::to_string(st.tie_sources) +
                          " gradient sources share their elevation with another one; the reference's Priority-Flood+Epsilon "
                          "resolves such ties by std::priority_queue's pop order, the MI355X engine returns the order-free "
                          "surface (a cell-wise lower bound of the reference's)";
  if (PyErr_WarnEx(PyExc_RuntimeWarning, msg.c_str(), 1) < 0) throw py::error_already_set();
}

struct DepressionRecord {
  static constexpr std::uint32_t NONE = 0xFFFFFFFFu;
  std::uint32_t pit_cell = NONE, out_cell = NONE, parent = NONE, odep = NONE, geolink = NONE;
  double pit_elev = std::numeric_limits<double>::infinity(), out_elev = std::numeric_limits<double>::infinity();
  std::uint32_t lchild = NONE, rchild = NONE;
  bool ocean_parent = false;
  std::vector<std::uint32_t> ocean_linked;
  std::uint32_t dep_label = 0, cell_count = 0;
  double dep_vol = 0, water_vol = 0, total_elevation = 0;
};

// reference pywrapper.hpp:27-82 (TemplatedFunctionsWrapper<T>)
template <class T>
void bind_functions(py::module_ &m) {
  m.def("rdFillDepressionsD8", [](Array2D<T> &dem) { rdgpu::PriorityFlood_Zhou2016(dem); }, release(),
        "Fill all depressions, D8 (PriorityFlood_Zhou2016's result).");
  m.def("rdFillDepressionsD4", [](Array2D<T> &dem) { rdgpu::PriorityFlood_Barnes2014<Topology::D4>(dem); }, release(),
        "Fill all depressions, D4 (PriorityFlood_Barnes2014<D4>'s result).");
  // not in the reference's module (it binds the default fill only): the other names of the sweep, for scripts that call them
  m.def("rdPriorityFloodOriginalD8", [](Array2D<T> &dem) { rdgpu::PriorityFlood_Original<Topology::D8>(dem); }, release(),
        "PriorityFlood_Original<D8> (depressions/Barnes2014.hpp:136-198).");
  m.def("rdPriorityFloodOriginalD4", [](Array2D<T> &dem) { rdgpu::PriorityFlood_Original<Topology::D4>(dem); }, release(),
        "PriorityFlood_Original<D4>.");
  m.def("rdPriorityFloodWei2018", [](Array2D<T> &dem) { rdgpu::PriorityFlood_Wei2018(dem); }, release(),
        "PriorityFlood_Wei2018 (depressions/Wei2018.hpp:154-202): NoData cells are left alone, their neighbours are seeds.");
  m.def("rdHasDepressionsD8", [](const Array2D<T> &dem) { return rdgpu::HasDepressions<Topology::D8>(dem); }, release(),
        "HasDepressions<D8> (depressions/Barnes2014.hpp:44-103).");
  m.def("rdHasDepressionsD4", [](const Array2D<T> &dem) { return rdgpu::HasDepressions<Topology::D4>(dem); }, release(),
        "HasDepressions<D4>.");
  // pywrapper.hpp:34-35 (floating-point element types; the others raise as the reference's specialisations do)
  // (the engine runs with the GIL released; the tie census of the call is turned into a Python RuntimeWarning afterwards)
  m.def("rdPFepsilonD8", [](Array2D<T> &dem) {
    { py::gil_scoped_release nogil; rdgpu::PriorityFloodEpsilon_Barnes2014<Topology::D8>(dem); }
    warn_epsilon_ties();
  }, "Fill all depressions with epsilon.");
  m.def("rdPFepsilonD4", [](Array2D<T> &dem) {
    { py::gil_scoped_release nogil; rdgpu::PriorityFloodEpsilon_Barnes2014<Topology::D4>(dem); }
    warn_epsilon_ties();
  }, "Fill all depressions with epsilon.");
  m.def("rdResolveFlatsEpsilon", [](Array2D<T> &dem) { rdgpu::ResolveFlatsEpsilon(dem); }, release(),
        "Raise the cells of drainable flats by the smallest representable steps (ResolveFlatsEpsilon).");

  m.def("FA_Tarboton", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_Tarboton(dem, accum); }, release());
  m.def("FA_Dinfinity", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_Dinfinity(dem, accum); }, release());
  m.def("FA_Holmgren", [](const Array2D<T> &dem, Array2D<double> &accum, double x) { rdgpu::FA_Holmgren(dem, accum, x); }, release());
  m.def("FA_Quinn", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_Quinn(dem, accum); }, release());
  m.def("FA_Freeman", [](const Array2D<T> &dem, Array2D<double> &accum, double x) { rdgpu::FA_Freeman(dem, accum, x); }, release());
  m.def("FA_D8", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_D8(dem, accum); }, release());
  m.def("FA_D4", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_D4(dem, accum); }, release());
  m.def("FA_OCallaghanD8", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_OCallaghanD8(dem, accum); }, release());
  m.def("FA_OCallaghanD4", [](const Array2D<T> &dem, Array2D<double> &accum) { rdgpu::FA_OCallaghanD4(dem, accum); }, release());

  m.def("FM_Tarboton", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_Tarboton(dem, props); }, release());
  m.def("FM_Dinfinity", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_Dinfinity(dem, props); }, release());
  m.def("FM_Holmgren", [](const Array2D<T> &dem, Array3D<float> &props, double x) { rdgpu::FM_Holmgren(dem, props, x); }, release());
  m.def("FM_Quinn", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_Quinn(dem, props); }, release());
  m.def("FM_Freeman", [](const Array2D<T> &dem, Array3D<float> &props, double x) { rdgpu::FM_Freeman(dem, props, x); }, release());
  m.def("FM_OCallaghanD8", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_OCallaghan<Topology::D8>(dem, props); }, release());
  m.def("FM_OCallaghanD4", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_OCallaghan<Topology::D4>(dem, props); }, release());
  m.def("FM_D8", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_D8(dem, props); }, release());
  m.def("FM_D4", [](const Array2D<T> &dem, Array3D<float> &props) { rdgpu::FM_D4(dem, props); }, release());
}

// the numpy array behind `src`, zero-copy: exact dtype, C-contiguous, `dims` dimensions
template <class T>
py::array_t<T> exact_array(py::handle src, int dims, const char *dims_message) {
  if (!py::isinstance<py::array>(src)) throw std::runtime_error("Unable to convert array to RichDEM object!");
  py::array a = py::reinterpret_borrow<py::array>(src);
  if (!a.dtype().is(py::dtype::of<T>()) || !(a.flags() & py::array::c_style))
    throw std::runtime_error("Unable to convert array to RichDEM object! (the element type must match and the array must be C-contiguous: wrapping never copies)");
  if (a.ndim() != dims) throw std::runtime_error(dims_message);
  return py::reinterpret_borrow<py::array_t<T>>(src);
}

// setNoData takes any Python number (reference pywrapper.hpp:141-150: one overload per C++ number type)
template <class A, class T, class C>
void def_set_nodata(C &cls) {
  cls.def("setNoData", [](A &a, std::int64_t v) { a.setNoData((T)v); })
      .def("setNoData", [](A &a, std::uint64_t v) { a.setNoData((T)v); })
      .def("setNoData", [](A &a, double v) { a.setNoData((T)v); });
}

// reference pywrapper.hpp:87-190 (TemplatedArrayWrapper<T>)
template <class T>
void bind_array2d(py::module_ &m, const std::string &tname) {
  using A = Array2D<T>;
  py::class_<A> cls(m, ("Array2D_" + tname).c_str(), py::dynamic_attr());
  cls.def(py::init<>())
      .def(py::init<typename A::xy_t, typename A::xy_t, T>())
      .def(py::init([](py::handle src) {
             auto buf = exact_array<T>(src, 2, "Array must have two dimensions!");
             return new A(buf.mutable_data(), (typename A::xy_t)buf.shape(1), (typename A::xy_t)buf.shape(0));
           }),
           py::keep_alive<1, 2>())   // the wrapper does not own the cells: keep the numpy array alive
      .def("size", &A::size)
      .def("width", &A::width)
      .def("height", &A::height)
      .def("empty", &A::empty)
      .def("noData", &A::noData)
      .def("min", &A::min)
      .def("max", &A::max)
      .def_readwrite("geotransform", &A::geotransform)
      .def_readwrite("projection", &A::projection)
      .def_readwrite("metadata", &A::metadata)
      .def("copy", [](const A &a) { return A(a); })
      .def("__repr__",
           [tname](const A &a) {
             return "<RichDEM array: type=" + tname + ", width=" + std::to_string(a.width()) + ", height=" +
                    std::to_string(a.height()) + ", owned=" + std::to_string(a.owned()) + ">";
           })
      .def("__call__",
           [](const A &a, int x, int y) {
             if (!a.inGrid(x, y)) throw py::index_error("cell outside the raster");
             return a(x, y);
           })
      .def("__call__", [](const A &a, std::uint32_t i) {
        if (i >= a.size()) throw py::index_error("cell outside the raster");
        return a(i);
      });
  def_set_nodata<A, T>(cls);
}

void bind_array3d(py::module_ &m) {
  using A = Array3D<float>;
  py::class_<A> cls(m, "Array3D_float", py::dynamic_attr());
  cls.def(py::init<>())
      .def(py::init<A::xy_t, A::xy_t, float>())
      .def(py::init([](py::handle src) {
             auto buf = exact_array<float>(src, 3, "Array must have three dimensions!");
             if (buf.shape(2) != A::LAYERS) throw std::runtime_error("Array must have nine slots per cell: shape (height, width, 9)!");
             return new A(buf.mutable_data(), (A::xy_t)buf.shape(1), (A::xy_t)buf.shape(0));   // (y, x, slot) order
           }),
           py::keep_alive<1, 2>())
      .def("size", &A::size)
      .def("width", &A::width)
      .def("height", &A::height)
      .def("empty", &A::empty)
      .def("noData", &A::noData)
      .def_readwrite("geotransform", &A::geotransform)
      .def_readwrite("projection", &A::projection)
      .def_readwrite("metadata", &A::metadata)
      .def("copy", [](const A &a) { return A(a); })
      .def("__repr__",
           [](const A &a) {
             return "<RichDEM 3D array: type=float, width=" + std::to_string(a.width()) + ", height=" +
                    std::to_string(a.height()) + ", owned=" + std::to_string(a.owned()) + ">";
           })
      .def("__call__",
           [](const A &a, int x, int y, int n) {
             if (x < 0 || y < 0 || x >= a.width() || y >= a.height() || n < 0 || n >= A::LAYERS) throw py::index_error("slot outside the array");
             return a(x, y, n);
           })
      .def("getIN", [](const A &a, std::uint32_t i, int n) {
        if (i >= a.size() || n < 0 || n >= A::LAYERS) throw py::index_error("slot outside the array");
        return a.getIN(i, n);
      });
  def_set_nodata<A, float>(cls);
}

}  // namespace

PYBIND11_MODULE(_richdem, m) {
  m.doc() = "RichDEM's internal calculation library on the MI355X engine (librdgpu.so): drop-in for the reference's _richdem";

  m.attr("NO_FLOW") = 0;   // richdem::NO_FLOW, common/constants.hpp:80
  m.attr("engine") = "rdgpu";

  py::bind_map<std::map<std::string, std::string>>(m, "MapStringString");   // pywrapper.cpp:23

#define RDGPU_FOR_TYPES(X)                                                                                       \
  X(float, "float") X(double, "double") X(std::int8_t, "int8_t") X(std::int16_t, "int16_t") X(std::int32_t, "int32_t") \
  X(std::int64_t, "int64_t") X(std::uint8_t, "uint8_t") X(std::uint16_t, "uint16_t") X(std::uint32_t, "uint32_t")      \
  X(std::uint64_t, "uint64_t")
#define X(T, NAME) bind_array2d<T>(m, NAME);
  RDGPU_FOR_TYPES(X)   // the array classes first: the functions' signatures name them
#undef X
  bind_array3d(m);
#define X(T, NAME) bind_functions<T>(m);
  RDGPU_FOR_TYPES(X)
#undef X
#undef RDGPU_FOR_TYPES

  // methods/flow_accumulation_generic.hpp:33-100 (pywrapper.cpp:50)
  m.def("FlowAccumulation", [](const Array3D<float> &props, Array2D<double> &accum) { rdgpu::FlowAccumulation(props, accum); },
        release(), "Flow accumulation from nine flow proportions per cell; accum is in/out (in: flow generated per cell).");

  m.def("rdHash", []() { return std::string(rdgpu_version()); }, "Version of the engine (the reference returns its git hash).");
  m.def("rdCompileTime", []() { return std::string(__DATE__ " " __TIME__); }, "Build time of this module.");

  for (const char *name : {"rdBreachDepressionsD8", "rdBreachDepressionsD4", "TA_SPI", "TA_CTI",
                           "TA_slope_riserun", "TA_slope_percentage", "TA_slope_degrees", "TA_slope_radians", "TA_aspect",
                           "TA_curvature", "TA_planform_curvature", "TA_profile_curvature", "FA_FairfieldLeymarieD8",
                           "FA_FairfieldLeymarieD4", "FA_Rho8", "FA_Rho4", "FM_FairfieldLeymarieD8", "FM_FairfieldLeymarieD4",
                           "FM_Rho8", "FM_Rho4", "generate_perlin_terrain"})
    def_out_of_scope(m, name);

  // pywrapper.cpp:134-170: the depression-hierarchy submodule (constants kept, functions out of scope)
  py::module_ dh = m.def_submodule("depression_hierarchy", "Depression hierarchies (outside the scope of the MI355X engine)");
  dh.attr("NO_PARENT") = 0xFFFFFFFFu;   // depressions/depression_hierarchy.hpp:34-35, :163-166
  dh.attr("NO_VALUE") = 0xFFFFFFFFu;
  dh.attr("NO_DEP") = 0xFFFFFFFFu;
  dh.attr("OCEAN") = 0u;
  // the record type of the hierarchy (depression_hierarchy.hpp:41-95): a plain data holder; richdem/__init__.py names it
  // in its annotations at import time
  py::class_<DepressionRecord>(dh, "Depression")
      .def(py::init<>())
      .def_readwrite("pit_cell", &DepressionRecord::pit_cell)
      .def_readwrite("out_cell", &DepressionRecord::out_cell)
      .def_readwrite("parent", &DepressionRecord::parent)
      .def_readwrite("odep", &DepressionRecord::odep)
      .def_readwrite("geolink", &DepressionRecord::geolink)
      .def_readwrite("pit_elev", &DepressionRecord::pit_elev)
      .def_readwrite("out_elev", &DepressionRecord::out_elev)
      .def_readwrite("lchild", &DepressionRecord::lchild)
      .def_readwrite("rchild", &DepressionRecord::rchild)
      .def_readwrite("ocean_parent", &DepressionRecord::ocean_parent)
      .def_readwrite("ocean_linked", &DepressionRecord::ocean_linked)
      .def_readwrite("dep_label", &DepressionRecord::dep_label)
      .def_readwrite("cell_count", &DepressionRecord::cell_count)
      .def_readwrite("dep_vol", &DepressionRecord::dep_vol)
      .def_readwrite("water_vol", &DepressionRecord::water_vol)
      .def_readwrite("total_elevation", &DepressionRecord::total_elevation);
  def_out_of_scope(dh, "get_depression_hierarchy");
  def_out_of_scope(dh, "fill_spill_merge");
}
