// rdgpu/Array3D.hpp -- stand-alone counterpart of the reference's flow-proportion container
// (reference include/richdem/common/Array3D.hpp): nine float slots per cell, cell-major,
//     index(x, y, n) = (y * width + x) * 9 + n                                   (Array3D.hpp:204-206)
// slot 0 is the cell's state (NO_FLOW_GEN -1, HAS_FLOW_GEN 0, NoData), slots 1..8 the share of the cell's flow
// sent to neighbour n (flowmet/*.hpp).  It is the on-the-wire type of rd.FlowProportions / FlowAccumFromProps
// (wrappers/pyrichdem/src/pywrapper.cpp:52-131) and of the C-ABI's `props9` arguments.
// Only what the hot path and the Python binding need: owning or wrapping storage, NoData, georeferencing.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "Array2D.hpp"

namespace rdgpu {

template <class T>
class Array3D {
public:
  typedef int32_t xy_t;
  typedef uint32_t i_t;
  typedef uint8_t n_t;
  static constexpr int LAYERS = 9;

  std::vector<double> geotransform;
  std::string projection;
  std::map<std::string, std::string> metadata;

  Array3D() = default;
  Array3D(xy_t width, xy_t height, const T &val = T()) { resize(width, height, val); }

  // Wrap caller memory: (height, width, 9) C-contiguous, as numpy hands it over (pywrapper.cpp:84-85).
  Array3D(T *data0, xy_t width, xy_t height) : ptr_(data0), w_(width), h_(height), owned_(false) {
    if (!data0 || width <= 0 || height <= 0) throw std::runtime_error("Array3D: cannot wrap an empty buffer");
  }

  // Same dimensions and georeferencing as a raster (reference Array3D(const Array2D<U>&, val), Array3D.hpp:152-160)
  template <class U>
  explicit Array3D(const Array2D<U> &other, const T &val = T()) {
    resize(other.width(), other.height(), val);
    geotransform = other.geotransform;
    projection = other.projection;
    metadata = other.metadata;
  }

  Array3D(const Array3D &o) { *this = o; }
  Array3D &operator=(const Array3D &o) {
    if (this == &o) return *this;
    w_ = o.w_;
    h_ = o.h_;
    no_data_ = o.no_data_;
    geotransform = o.geotransform;
    projection = o.projection;
    metadata = o.metadata;
    const size_t n = (size_t)LAYERS * (size_t)w_ * (size_t)h_;
    store_.reset(n ? new T[n] : nullptr);
    ptr_ = store_.get();
    owned_ = true;
    for (size_t i = 0; i < n; i++) ptr_[i] = o.ptr_[i];
    return *this;
  }
  Array3D(Array3D &&) = default;
  Array3D &operator=(Array3D &&) = default;

  void resize(xy_t width, xy_t height, const T &val = T()) {
    if (!owned_) throw std::runtime_error("Array3D: cannot resize a wrapped buffer");
    if (width < 0 || height < 0) throw std::runtime_error("Array3D: negative dimension");
    const size_t n = (size_t)LAYERS * (size_t)width * (size_t)height;
    store_.reset(n ? new T[n] : nullptr);
    ptr_ = store_.get();
    w_ = width;
    h_ = height;
    for (size_t i = 0; i < n; i++) ptr_[i] = val;
  }

  T *data() { return ptr_; }
  const T *data() const { return ptr_; }
  i_t size() const { return (i_t)w_ * (i_t)h_; }   // cells, not slots (Array3D.hpp:168)
  xy_t width() const { return w_; }
  xy_t height() const { return h_; }
  bool empty() const { return size() == 0; }
  bool owned() const { return owned_; }
  T noData() const { return no_data_; }
  void setNoData(const T &v) { no_data_ = v; }

  T &operator()(xy_t x, xy_t y, int n) { return ptr_[((size_t)y * (size_t)w_ + (size_t)x) * LAYERS + (size_t)n]; }
  T operator()(xy_t x, xy_t y, int n) const { return ptr_[((size_t)y * (size_t)w_ + (size_t)x) * LAYERS + (size_t)n]; }
  T &getIN(i_t i, int n) { return ptr_[(size_t)i * LAYERS + (size_t)n]; }
  T getIN(i_t i, int n) const { return ptr_[(size_t)i * LAYERS + (size_t)n]; }

private:
  std::unique_ptr<T[]> store_;
  T *ptr_ = nullptr;
  xy_t w_ = 0, h_ = 0;
  bool owned_ = true;
  T no_data_ = (T)-1;
};

}  // namespace rdgpu
