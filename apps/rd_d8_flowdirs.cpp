// rd_d8_flowdirs on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_d8_flowdirs.cpp:12-25: PriorityFlood_Barnes2014<D8>, then barnes_flat_resolution_d8
// (alter = false); the output is the uint8 direction raster (NoData 255).
#include "common.hpp"

template <class T>
struct Flowdirs {
  static int run(const std::string &in, const std::string &out) {
    apps::Array2D<T> elevations(in, true);
    rdgpu::PriorityFlood_Barnes2014<apps::Topology::D8>(elevations);
    apps::Array2D<uint8_t> flowdirs;
    rdgpu::barnes_flat_resolution_d8(elevations, flowdirs, false);
    flowdirs.saveToCache(out);
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 3 || argc > 4) {
    std::cerr << "Calculate D8 flow directions (depressions filled, flats resolved)" << std::endl;
    std::cerr << argv[0] << " <Input native raster> <Output native raster (uint8)> [element type: f32]" << std::endl;
    return -1;
  }
  return apps::route<Flowdirs>(argc == 4 ? argv[3] : "f32", std::string(argv[1]), std::string(argv[2]));
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
