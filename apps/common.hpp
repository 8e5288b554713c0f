// Shared by the apps: dispatch on the element type named on the command line.  The reference's apps read the type
// from GDAL (apps/router.hpp); the native raster format does not store it, so it is an argument here (default f32).
#pragma once
#include <rdgpu/Array2D.hpp>
#include <rdgpu/richdem_gpu.hpp>

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>

namespace apps {
using rdgpu::Array2D;
using rdgpu::Topology;

template <template <class> class F, class... Args>
int route(const std::string &type, Args &&...args) {
  if (type == "f32") return F<float>::run(args...);
  if (type == "f64") return F<double>::run(args...);
  if (type == "u8") return F<uint8_t>::run(args...);
  if (type == "i16") return F<int16_t>::run(args...);
  if (type == "u16") return F<uint16_t>::run(args...);
  if (type == "i32") return F<int32_t>::run(args...);
  if (type == "u32") return F<uint32_t>::run(args...);
  std::cerr << "Unknown element type '" << type << "' (u8 i16 u16 i32 u32 f32 f64)" << std::endl;
  return -1;
}

inline int guarded_main(int (*body)(int, char **), int argc, char **argv) {
  try {
    return body(argc, argv);
  } catch (const std::exception &e) {
    std::cerr << "E " << e.what() << std::endl;
    return 1;
  }
}
}  // namespace apps
