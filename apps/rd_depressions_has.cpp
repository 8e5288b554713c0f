// rd_depressions_has on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_depressions_has.cpp:10-21 (PerformAlgorithm: HasDepressions<Topology::D8>(elevation), then
// "m Depressions found." / "m No depressions found." on stdout).
#include "common.hpp"

template <class T>
struct Has {
  static int run(const std::string &in) {
    apps::Array2D<T> elevation(in, true);
    if (rdgpu::HasDepressions<apps::Topology::D8>(elevation))
      std::cout << "m Depressions found." << std::endl;
    else
      std::cout << "m No depressions found." << std::endl;
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 2 || argc > 3) {
    std::cerr << argv[0] << " <Input native raster> [element type: f32]" << std::endl;
    return -1;
  }
  return apps::route<Has>(argc == 3 ? argv[2] : "f32", std::string(argv[1]));
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
