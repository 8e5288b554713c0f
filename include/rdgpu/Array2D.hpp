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
//     with resize(other) and templateCopy (Array2D.hpp:873-878, 1102-1108);
//   * the reference's NATIVE on-disk format (saveToCache / loadNative, Array2D.hpp:209-281, uncompressed
//     build): files written here load in the reference with Array2D<T>(filename, true) and vice versa.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
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

  // Load a raster from the reference's native format (reference Array2D(filename, native=true, ...),
  // Array2D.hpp:421-431).  native == false would mean GDAL, which this container does not link.
  explicit Array2D(const std::string &filename, bool native = true, bool load_data = true) {
    if (!native) throw std::runtime_error("RichDEM was not compiled with GDAL!");
    loadNative(filename, load_data);
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

  // extremes over the cells that are not NoData (reference Array2D.hpp:516-535; numeric_limits when there are none)
  T min() const {
    T m = std::numeric_limits<T>::max();
    for (size_t i = 0, n = (size_t)w_ * (size_t)h_; i < n; i++)
      if (!(ptr_[i] == no_data_) && ptr_[i] < m) m = ptr_[i];
    return m;
  }
  T max() const {
    T m = std::numeric_limits<T>::lowest();
    for (size_t i = 0, n = (size_t)w_ * (size_t)h_; i < n; i++)
      if (!(ptr_[i] == no_data_) && ptr_[i] > m) m = ptr_[i];
    return m;
  }

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

  // Native format (reference saveToCache, Array2D.hpp:209-246): int32 height, width, x offset, y offset; uint32
  // number of data cells (0xFFFFFFFF = not counted); T NoData; 6 doubles geotransform; size_t projection length +
  // bytes; width*height cells row-major.  A missing geotransform is written as the reference's fallback
  // {1000, 1, 0, 1000, 0, -1} (Array2D.hpp:149).
  void saveToCache(const std::string &filename) const {
    std::ofstream out(filename, std::ios::binary | std::ios::trunc);
    if (!out.good()) throw std::logic_error("Failed to open cache file '" + filename + "'.");
    const xy_t zero = 0;
    const i_t ndc = NO_I;
    double gt[6] = {1000., 1., 0., 1000., 0., -1.};
    if (geotransform.size() >= 6) std::copy(geotransform.begin(), geotransform.begin() + 6, gt);
    const std::string::size_type plen = projection.size();
    out.write(reinterpret_cast<const char *>(&h_), sizeof(xy_t));
    out.write(reinterpret_cast<const char *>(&w_), sizeof(xy_t));
    out.write(reinterpret_cast<const char *>(&zero), sizeof(xy_t));
    out.write(reinterpret_cast<const char *>(&zero), sizeof(xy_t));
    out.write(reinterpret_cast<const char *>(&ndc), sizeof(i_t));
    out.write(reinterpret_cast<const char *>(&no_data_), sizeof(T));
    out.write(reinterpret_cast<const char *>(gt), sizeof(gt));
    out.write(reinterpret_cast<const char *>(&plen), sizeof(plen));
    out.write(projection.data(), (std::streamsize)plen);
    out.write(reinterpret_cast<const char *>(ptr_), (std::streamsize)((size_t)w_ * (size_t)h_ * sizeof(T)));
    if (!out.good()) throw std::runtime_error("Failed to write native file '" + filename + "'!");
  }
  void saveNative(const std::string &filename) const { saveToCache(filename); }

  // reference loadNative, Array2D.hpp:251-281
  void loadNative(const std::string &filename, bool load_data = true) {
    std::ifstream in(filename, std::ios::in | std::ios::binary);
    if (!in.good()) throw std::runtime_error("Failed to load native file '" + filename + "!");
    xy_t hh = 0, ww = 0, xoff = 0, yoff = 0;
    i_t ndc = 0;
    std::string::size_type plen = 0;
    in.read(reinterpret_cast<char *>(&hh), sizeof(xy_t));
    in.read(reinterpret_cast<char *>(&ww), sizeof(xy_t));
    in.read(reinterpret_cast<char *>(&xoff), sizeof(xy_t));
    in.read(reinterpret_cast<char *>(&yoff), sizeof(xy_t));
    in.read(reinterpret_cast<char *>(&ndc), sizeof(i_t));
    in.read(reinterpret_cast<char *>(&no_data_), sizeof(T));
    geotransform.resize(6);
    in.read(reinterpret_cast<char *>(geotransform.data()), 6 * sizeof(double));
    in.read(reinterpret_cast<char *>(&plen), sizeof(plen));
    if (!in.good() || hh < 0 || ww < 0 || plen > (1u << 26)) throw std::runtime_error("Failed to load native file '" + filename + "!");
    projection.resize(plen, ' ');
    in.read(&projection[0], (std::streamsize)plen);
    if (load_data) {
      resize(ww, hh);
      in.read(reinterpret_cast<char *>(ptr_), (std::streamsize)((size_t)ww * (size_t)hh * sizeof(T)));
      if (!in.good()) throw std::runtime_error("Failed to load native file '" + filename + "!");
    }
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
