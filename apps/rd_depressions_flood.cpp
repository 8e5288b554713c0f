// rd_depressions_flood on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_depressions_flood.cpp:11-23 (PerformAlgorithm: PriorityFlood_Zhou2016 when the maximum
// depression size is 0, PriorityFlood_Barnes2014_max_dep<D8> otherwise) and its usage text (:31-38).
#include "common.hpp"

#include <cstdlib>

template <class T>
struct Flood {
  static int run(const std::string &in, const std::string &out, uint64_t max_dep_size) {
    apps::Array2D<T> elevation(in, true);
    if (max_dep_size == 0) rdgpu::PriorityFlood_Zhou2016(elevation);
    else rdgpu::PriorityFlood_Barnes2014_max_dep<apps::Topology::D8>(elevation, max_dep_size);
    elevation.saveToCache(out);
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 3 || argc > 5) {
    std::cerr << "Eliminate all depressions via flooding." << std::endl;
    std::cerr << argv[0] << " <Input native raster> <Output native raster> [<Maximum Depression Size> = 0] [element type = f32]" << std::endl;
    std::cerr << "\t<Maximum Depression Size> - Depressions larger than this are not flooded." << std::endl;
    std::cerr << "                              Use `0` to flood all depressions.            " << std::endl;
    return -1;
  }
  // (round 1 took the element type as the third argument: still accepted)
  std::string type = "f32";
  uint64_t max_dep_size = 0;   // (the reference takes it as uint64_t: depressions/Barnes2014.hpp:844)
  for (int i = 3; i < argc; i++) {
    const std::string a = argv[i];
    if (!a.empty() && a.find_first_not_of("0123456789") == std::string::npos) max_dep_size = (uint64_t)std::stoull(a);
    else type = a;
  }
  return apps::route<Flood>(type, std::string(argv[1]), std::string(argv[2]), max_dep_size);
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
