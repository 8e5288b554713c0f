// rd_depressions_flood on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_depressions_flood.cpp:11-23 (PerformAlgorithm: PriorityFlood_Zhou2016 when no maximum
// depression size is given); the bounded-depth variant (_max_dep) is order dependent and not provided.
#include "common.hpp"

template <class T>
struct Flood {
  static int run(const std::string &in, const std::string &out) {
    apps::Array2D<T> elevation(in, true);
    rdgpu::PriorityFlood_Zhou2016(elevation);
    elevation.saveToCache(out);
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 3 || argc > 4) {
    std::cerr << "Fill all depressions" << std::endl;
    std::cerr << argv[0] << " <Input native raster> <Output native raster> [element type: f32]" << std::endl;
    return -1;
  }
  return apps::route<Flood>(argc == 4 ? argv[3] : "f32", std::string(argv[1]), std::string(argv[2]));
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
