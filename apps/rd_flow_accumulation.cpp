// rd_flow_accumulation on the GPU engine, native raster files instead of GDAL ones.
// Mirrors reference apps/rd_flow_accumulation.cpp:9-44: accum(dem, 1); FA_<algorithm>(dem, accum); scale by the
// cell area; save.  Algorithm numbers as in the reference; Rho8 (2) draws from a process-global random engine and
// 7-9 are disabled in the reference itself.
#include "common.hpp"

template <class T>
struct Accumulate {
  static int run(const std::string &in, const std::string &out, int algorithm, float param) {
    apps::Array2D<T> dem(in, true);
    apps::Array2D<double> accum(dem, 1);
    switch (algorithm) {
      case 1: rdgpu::FA_D8(dem, accum); break;                  // D8  - O'Callaghan/Marks (1984)
      case 3: rdgpu::FA_Quinn(dem, accum); break;               // MD8 - Quinn (1991)
      case 4: rdgpu::FA_Holmgren(dem, accum, param); break;     // MD8 - Holmgren (1994)
      case 5: rdgpu::FA_Freeman(dem, accum, param); break;      // MD8 - Freeman (1991)
      case 6: rdgpu::FA_Tarboton(dem, accum); break;            // Dinf - Tarboton (1997)
      default: std::cerr << "This FA is not provided by the GPU engine." << std::endl; return -1;
    }
    accum.scale(accum.getCellArea());
    accum.saveToCache(out);
    return 0;
  }
};

static int body(int argc, char **argv) {
  if (argc < 4 || argc > 6) {
    std::cerr << "Calculate flow accumulation in terms of upstream area" << std::endl;
    std::cerr << argv[0] << " <DEM native raster> <Output native raster (float64)> <Algorithm #> [Parameter] [element type: f32]" << std::endl;
    std::cerr << " 1: D8   3: Quinn   4: Holmgren (x)   5: Freeman (p)   6: Dinf" << std::endl;
    return -1;
  }
  const int algorithm = std::stoi(argv[3]);
  const bool needs = algorithm == 4 || algorithm == 5;
  if (needs && argc < 5) { std::cerr << "Algorithm requires a parameter!" << std::endl; return -1; }
  const float param = needs ? std::stof(argv[4]) : 0.0f;
  const int ti = needs ? 5 : 4;
  return apps::route<Accumulate>(argc > ti ? argv[ti] : "f32", std::string(argv[1]), std::string(argv[2]), algorithm, param);
}
int main(int argc, char **argv) { return apps::guarded_main(body, argc, argv); }
