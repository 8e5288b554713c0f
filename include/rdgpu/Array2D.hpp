// rdgpu/Array2D.hpp -- a small row-major raster container with the interface subset of
// richdem::Array2D<T> that the hot path touches (reference include/richdem/common/Array2D.hpp:89-1416,
// ManagedVector.hpp:11-181).  Written from scratch for stand-alone use of the GPU engine (tests, tools,
// the GPU box where the reference tree is absent).  Inside the reference tree you do NOT need this file:
// the shim in rdgpu/richdem_gpu.hpp is templated on the array type and binds to richdem::Array2D<T>
// directly.
//
// Contract mirrored from the reference:
//   * dense row-major storage, i = y*width + x, no padding (Array2D.hpp:592-595);
//   * xy_t = int32_t, i_t = uint32_t (Array2D.hpp:100-101);
//   * a raster can WRAP caller memory without owning it (Array2D.hpp:344-352); resizing a wrapping
//     raster throws (ManagedVector.hpp:158-172);
//   * NoData value defaults to -1 (Array2D.hpp:114); geotransform / projection / metadata travel
//     with resize(other) and templateCopy (Array2D.hpp:873-878, 1102-1108).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rdgpu {

enum class Topology { D8, D4 };  // same enumerators as richdem::Topology (common/constants.hpp:97-100)

template <class T>
class Array2D {
public:
  typedef int32_t xy_t;
  typedef uint32_t i_t;
  static const i_t NO_I = 0xFFFFFFFFu;

  std::vector<double> geotransform;
  std::string projection;
  std::map<std::string, std::string> metadata;
  std::string basename;

  Array2D() = default;

  Array2D(xy_t width, xy_t height, const T &val = T()) { resize(width, height, val); }

  // Wrap caller memory (row-major, width*height elements).  Not owned, never freed, never resized.
  Array2D(T *data0, xy_t width, xy_t height) : ptr_(data0), w_(width), h_(height), owned_(false) {
    if (!data0 || width <= 0 || height <= 0) throw std::runtime_error("Array2D: cannot wrap an empty buffer");
  }

  // Same dimensions and georeferencing as `other`, every cell set to val.
  template <class U>
  Array2D(const Array2D<U> &other, const T &val = T()) {
    resize(other, val);
    metadata = other.metadata;
    basename = other.basename;
  }

  Array2D(const Array2D &o) { *this = o; }
  Array2D &operator=(const Array2D &o) {
    if (this == &o) return *this;
    store_.reset();
    owned_ = true;
    w_ = o.w_;
    h_ = o.h_;
    no_data_ = o.no_data_;
    geotransform = o.geotransform;
    projection = o.projection;
    metadata = o.metadata;
    basename = o.basename;
    const size_t n = (size_t)w_ * (size_t)h_;
    store_.reset(n ? new T[n] : nullptr);
    ptr_ = store_.get();
    std::copy(o.ptr_, o.ptr_ + n, ptr_);
    return *this;
  }
  Array2D(Array2D &&) = default;
  Array2D &operator=(Array2D &&) = default;

  T *data() { return ptr_; }
  const T *data() const { return ptr_; }
  i_t size() const { return (i_t)w_ * (i_t)h_; }
  xy_t width() const { return w_; }
  xy_t height() const { return h_; }
  bool empty() const { return size() == 0; }
  bool owned() const { return owned_; }

  T noData() const { return no_data_; }
  void setNoData(const T &v) { no_data_ = v; }

  void setAll(const T &val) { std::fill(ptr_, ptr_ + (size_t)w_ * (size_t)h_, val); }

  void resize(xy_t width, xy_t height, const T &val = T()) {
    if (!owned_) throw std::runtime_error("Cannot resize unowned memory!");
    if (width < 0 || height < 0) throw std::runtime_error("Array2D: negative dimensions");
    const size_t n = (size_t)width * (size_t)height;
    if (n != (size_t)w_ * (size_t)h_ || !ptr_) {
      store_.reset(n ? new T[n] : nullptr);
      ptr_ = store_.get();
    }
    w_ = width;
    h_ = height;
    setAll(val);
  }

  template <class U>
  void resize(const Array2D<U> &other, const T &val = T()) {
    resize(other.width(), other.height(), val);
    geotransform = other.geotransform;
    projection = other.projection;
  }

  template <class U>
  void templateCopy(const Array2D<U> &other) {
    geotransform = other.geotransform;
    projection = other.projection;
    basename = other.basename;
    metadata = other.metadata;
  }

  i_t xyToI(xy_t x, xy_t y) const { return (i_t)y * (i_t)w_ + (i_t)x; }
  void iToxy(i_t i, xy_t &x, xy_t &y) const {
    x = (xy_t)(i % (i_t)w_);
    y = (xy_t)(i / (i_t)w_);
  }
  // flat index of the neighbour (dx, dy) of cell i, NO_I when it is off the grid
  i_t nToI(i_t i, xy_t dx, xy_t dy) const {
    const xy_t x = (xy_t)(i % (i_t)w_) + dx, y = (xy_t)(i / (i_t)w_) + dy;
    return inGrid(x, y) ? xyToI(x, y) : NO_I;
  }
  bool inGrid(xy_t x, xy_t y) const { return x >= 0 && y >= 0 && x < w_ && y < h_; }
  bool isEdgeCell(xy_t x, xy_t y) const { return x == 0 || y == 0 || x == w_ - 1 || y == h_ - 1; }
  bool isNoData(xy_t x, xy_t y) const { return ptr_[xyToI(x, y)] == no_data_; }
  bool isNoData(i_t i) const { return ptr_[i] == no_data_; }

  T &operator()(i_t i) { return ptr_[i]; }
  T operator()(i_t i) const { return ptr_[i]; }
  T &operator()(xy_t x, xy_t y) { return ptr_[xyToI(x, y)]; }
  T operator()(xy_t x, xy_t y) const { return ptr_[xyToI(x, y)]; }

  // same cells and same NoData value (reference Array2D.hpp:649-658)
  bool operator==(const Array2D<T> &o) const {
    if (w_ != o.w_ || h_ != o.h_ || no_data_ != o.no_data_) return false;
    return std::equal(ptr_, ptr_ + (size_t)w_ * (size_t)h_, o.ptr_);
  }

  i_t countval(const T &val) const { return (i_t)std::count(ptr_, ptr_ + (size_t)w_ * (size_t)h_, val); }

  // multiply every data cell (reference Array2D.hpp:1406-1410)
  void scale(double x) {
    const size_t n = (size_t)w_ * (size_t)h_;
    for (size_t i = 0; i < n; i++)
      if (ptr_[i] != no_data_) ptr_[i] = (T)(ptr_[i] * x);
  }

  // |cell width * cell height| from the geotransform; 1 when there is none
  double getCellArea() const { return geotransform.size() >= 6 ? std::abs(geotransform[1] * geotransform[5]) : 1.0; }

private:
  template <class U>
  friend class Array2D;
  std::unique_ptr<T[]> store_;
  T *ptr_ = nullptr;
  xy_t w_ = 0, h_ = 0;
  bool owned_ = true;
  T no_data_ = (T)-1;
};

}  // namespace rdgpu
