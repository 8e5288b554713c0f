// rd_depressions_mask on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_depressions_mask.cpp:10-21 (PerformAlgorithm: Array2D<uint8_t> mask(elevation);
// pit_mask<Topology::D8>(elevation, mask)): 1 = the cell lies in a depression, 0 = it does not, 3 = NoData.
#include "common.hpp"

template <class T>
struct Mask {
  static int run(const std::string &in, const std::string &out) {
    apps::Array2D<T> elevation(in, true);
    apps::Array2D<uint8_t> mask(elevation);
    rdgpu::pit_mask<apps::Topology::D8>(elevation, mask);
    mask.saveToCache(out);
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 3 || argc > 4) {
    std::cerr << "Return a raster in which 1 indicates depressions, 0 indicates non-depressions, and 3 indicates NoData." << std::endl;
    std::cerr << argv[0] << " <Input native raster> <Output native raster> [element type: f32]" << std::endl;
    return -1;
  }
  return apps::route<Mask>(argc == 4 ? argv[3] : "f32", std::string(argv[1]), std::string(argv[2]));
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
