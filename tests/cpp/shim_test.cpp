// tests/cpp/shim_test.cpp -- drives librdgpu.so through the C++ shim (rdgpu/richdem_gpu.hpp) exactly as
// a RichDEM app would, and checks the results against the CPU oracle (oracle/liboracle.so, checker only).
// With -DRDGPU_TEST_WITH_RICHDEM the same code is compiled against the reference's own
// richdem::Array2D<T> (compile check of the drop-in claim; only where /root/reference exists).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#ifdef RDGPU_TEST_WITH_RICHDEM
#include <richdem/common/Array2D.hpp>
#include <richdem/common/Array3D.hpp>
#include <richdem/common/constants.hpp>
template <class T>
using Arr = richdem::Array2D<T>;
using Arr3 = richdem::Array3D<float>;
using Topo = richdem::Topology;
static float *slots(Arr3 &p) { return p.getData(); }
#else
#include <rdgpu/Array2D.hpp>
#include <rdgpu/Array3D.hpp>
template <class T>
using Arr = rdgpu::Array2D<T>;
using Arr3 = rdgpu::Array3D<float>;
using Topo = rdgpu::Topology;
static float *slots(Arr3 &p) { return p.data(); }
#endif
#include <rdgpu/richdem_gpu.hpp>

extern "C" {
void orc_fill_f32(float *, int, int, int);
void orc_fill_i32(int32_t *, int, int, int);
void orc_fill_f64(double *, int, int, int);
void orc_fill_wei2018_f32(float *, float, int, int);
int orc_has_depressions_f32(const float *, int, int, int);
void orc_flat_resolution_f32(const float *, float, int, int, uint8_t *);
void orc_flat_resolution_alter_f32(float *, float, int, int, uint8_t *);
void orc_d8_flowdirs_f32(const float *, float, int, int, uint8_t *);
void orc_pf_flowdirs_f32(const float *, float, int, int, uint8_t *);
void orc_d8_flow_accum_f64(const uint8_t *, uint8_t, int, int, double *);
void orc_d8_flow_accum_i32(const uint8_t *, uint8_t, int, int, int32_t *);
void orc_fa_mfd_f32(const float *dem, float nodata, int w, int h, int method, double xparam, double *accum);
void orc_resolve_flats_epsilon_f32(float *dem, float nodata, int w, int h);
void orc_pit_mask_f32(const float *dem, float nodata, int w, int h, int topo, uint8_t *mask);
void orc_fa_d8_f32(const float *, float, int, int, double *);
void orc_fm_d8_f32(const float *dem, float nodata, int w, int h, float *props9);
void orc_fm_mfd_f32(const float *dem, float nodata, int w, int h, int method, double xparam, float *props9);
}

static int failures = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)

static float noise(int x, int y) {   // cheap deterministic terrain with pits and flats
  unsigned h = (unsigned)x * 2654435761u ^ (unsigned)y * 40503u;
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  return (float)((x / 16 + y / 16) * 4 + (int)(h % 23));
}

int main() {
  const int w = 301, h = 203;
  Arr<float> dem(w, h, 0.0f);
  dem.setNoData(-9999.0f);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) dem(x, y) = noise(x, y);
  std::vector<float> ref(dem.data(), dem.data() + (size_t)w * h);

  // FillDepressions<D8> / PriorityFlood_Zhou2016 / D4, in place on an owning array
  {
    Arr<float> a = dem;
    rdgpu::FillDepressions<Topo::D8>(a);
    std::vector<float> e = ref;
    orc_fill_f32(e.data(), w, h, 8);
    EXPECT(std::memcmp(a.data(), e.data(), e.size() * 4) == 0);
    Arr<float> b = dem;
    rdgpu::PriorityFlood_Zhou2016(b);
    EXPECT(b == a);
    Arr<float> c = dem;
    rdgpu::FillDepressions<Topo::D4>(c);
    std::vector<float> e4 = ref;
    orc_fill_f32(e4.data(), w, h, 4);
    EXPECT(std::memcmp(c.data(), e4.data(), e4.size() * 4) == 0);
    Arr<float> d4 = dem;
    rdgpu::PriorityFlood_Barnes2014<Topo::D4>(d4);
    EXPECT(d4 == c);
  }
  // the sweep's other names (reference tests/tests.cpp:233-271): PriorityFlood_Original<topo>, PriorityFlood_Wei2018 with NoData
  // holes as outlets (Wei2018.hpp:14-50), HasDepressions<topo> (apps/rd_depressions_has.cpp:14)
  {
    Arr<float> a = dem, b = dem;
    rdgpu::PriorityFlood_Original<Topo::D8>(a);
    rdgpu::FillDepressions<Topo::D8>(b);
    EXPECT(a == b);
    Arr<float> hole = dem;
    for (int y = 90; y < 104; y++)
      for (int x = 120; x < 180; x++) hole(x, y) = -9999.0f;
    std::vector<float> e(hole.data(), hole.data() + (size_t)w * h);
    orc_fill_wei2018_f32(e.data(), -9999.0f, w, h);
    Arr<float> plain = hole;
    rdgpu::PriorityFlood_Wei2018(hole);
    EXPECT(std::memcmp(hole.data(), e.data(), e.size() * 4) == 0);
    rdgpu::FillDepressions<Topo::D8>(plain);
    EXPECT(!(plain == hole));
    EXPECT(rdgpu::HasDepressions<Topo::D8>(dem) == (orc_has_depressions_f32(dem.data(), w, h, 8) != 0));
    EXPECT(rdgpu::HasDepressions<Topo::D8>(dem));
    EXPECT(!rdgpu::HasDepressions<Topo::D8>(b));
    EXPECT(!rdgpu::HasDepressions<Topo::D4>(Arr<float>(5, 4, 1.0f)));
  }
  // wrapping (externally owned) integer memory: the numpy -> Array2D(T*,w,h) path of the Python wrapper
  {
    std::vector<int32_t> buf((size_t)w * h), e;
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (int32_t)ref[i];
    e = buf;
    Arr<int32_t> wrapped(buf.data(), w, h);
    rdgpu::FillDepressions<Topo::D8>(wrapped);
    orc_fill_i32(e.data(), w, h, 8);
    EXPECT(wrapped.data() == buf.data());
    EXPECT(std::memcmp(buf.data(), e.data(), e.size() * 4) == 0);
  }
  // PriorityFloodFlowdirs_Barnes2014 on a DEM without equal elevations (a permutation of 0 .. w*h-1 scattered by a prime)
  {
    Arr<float> p(w, h, 0.0f);
    p.setNoData(-9999.0f);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) p(x, y) = (float)(((uint64_t)(y * w + x) * 2654435761ull) % 1000003ull);
    Arr<uint8_t> fd;
    rdgpu::PriorityFloodFlowdirs_Barnes2014(p, fd);
    EXPECT(fd.width() == w && fd.height() == h && fd.noData() == 0);
    std::vector<uint8_t> ef((size_t)w * h);
    orc_pf_flowdirs_f32(p.data(), -9999.0f, w, h, ef.data());
    EXPECT(std::memcmp(fd.data(), ef.data(), ef.size()) == 0);
  }

  // rd_d8_flowdirs chain: fill -> barnes_flat_resolution_d8 -> d8_flow_accum
  {
    Arr<float> a = dem;
    rdgpu::PriorityFlood_Barnes2014<Topo::D8>(a);
    Arr<uint8_t> dirs;
    rdgpu::barnes_flat_resolution_d8(a, dirs, false);
    EXPECT(dirs.width() == w && dirs.height() == h && dirs.noData() == 255);
    std::vector<uint8_t> ed((size_t)w * h);
    orc_flat_resolution_f32(a.data(), -9999.0f, w, h, ed.data());
    EXPECT(std::memcmp(dirs.data(), ed.data(), ed.size()) == 0);

    Arr<uint8_t> raw;
    rdgpu::d8_flow_directions(a, raw);
    std::vector<uint8_t> er((size_t)w * h);
    orc_d8_flowdirs_f32(a.data(), -9999.0f, w, h, er.data());
    EXPECT(std::memcmp(raw.data(), er.data(), er.size()) == 0);

    Arr<double> area;
    rdgpu::d8_flow_accum(dirs, area);
    EXPECT(area.noData() == -1.0 && area.width() == w);
    std::vector<double> ea((size_t)w * h);
    orc_d8_flow_accum_f64(ed.data(), 255, w, h, ea.data());
    EXPECT(std::memcmp(area.data(), ea.data(), ea.size() * 8) == 0);
    Arr<int32_t> areai;
    rdgpu::d8_flow_accum(dirs, areai);
    std::vector<int32_t> ei((size_t)w * h);
    orc_d8_flow_accum_i32(ed.data(), 255, w, h, ei.data());
    EXPECT(std::memcmp(areai.data(), ei.data(), ei.size() * 4) == 0);

    // alter = true: the DEM itself is raised (nextafterf steps), then plain D8
    Arr<float> b = a;
    Arr<uint8_t> adirs;
    rdgpu::barnes_flat_resolution_d8(b, adirs, true);
    std::vector<float> eb(a.data(), a.data() + (size_t)w * h);
    std::vector<uint8_t> ead((size_t)w * h);
    orc_flat_resolution_alter_f32(eb.data(), -9999.0f, w, h, ead.data());
    EXPECT(std::memcmp(b.data(), eb.data(), eb.size() * 4) == 0);
    EXPECT(std::memcmp(adirs.data(), ead.data(), ead.size()) == 0);
  }
  // rd_flow_accumulation: Array2D<double> accum(dem, 1); FA_D8(dem, accum)
  {
    Arr<double> accum(dem, 1.0);
    rdgpu::FA_D8(dem, accum);
    std::vector<double> e((size_t)w * h, 1.0);
    orc_fa_d8_f32(dem.data(), -9999.0f, w, h, e.data());
    EXPECT(std::memcmp(accum.data(), e.data(), e.size() * 8) == 0);
    EXPECT(accum.noData() == -1.0);
    Arr<double> wrong(3, 3, 1.0);
    bool threw = false;
    try { rdgpu::FA_D8(dem, wrong); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw);
    // accum_t = float / int32_t (the reference's FA_D8<elev_t, accum_t> is templated on it): totals below 2^24 are exact
    Arr<float> af(dem, 1.0f);
    Arr<int32_t> ai(dem, 1);
    rdgpu::FA_D8(dem, af);
    rdgpu::FA_D8(dem, ai);
    bool same = af.noData() == -1.0f && ai.noData() == -1;
    for (size_t i = 0; i < e.size() && same; i++) same = af.data()[i] == (float)e[i] && ai.data()[i] == (int32_t)e[i];
    EXPECT(same);
  }
  // rd_flow_accumulation's other deterministic methods: FA_Quinn / FA_Holmgren / FA_Freeman / FA_D4
  {
    Arr<double> a1(dem, 1.0), a2(dem, 1.0), a3(dem, 1.0), a4(dem, 1.0);
    rdgpu::FA_Quinn(dem, a1);
    rdgpu::FA_Holmgren(dem, a2, 2.0);
    rdgpu::FA_Freeman(dem, a3, 1.1);
    rdgpu::FA_D4(dem, a4);
    const Arr<double> *got[4] = {&a1, &a2, &a3, &a4};
    const int method[4] = {2, 0, 1, 3};
    const double xp[4] = {1.0, 2.0, 1.1, 1.0};
    for (int m = 0; m < 4; m++) {
      std::vector<double> e((size_t)w * h, 1.0);
      orc_fa_mfd_f32(dem.data(), -9999.0f, w, h, method[m], xp[m], e.data());
      bool close = true;
      for (size_t i = 0; i < e.size(); i++) close &= std::fabs(got[m]->data()[i] - e[i]) <= 2e-6 * std::fabs(e[i]);
      EXPECT(close);
      EXPECT(got[m]->noData() == -1.0);
    }
  }
  // double DEMs (the Python wrapper's default dtype): lossless-f32 path and value-rank path
  {
    Arr<double> d(w, h, 0.0), g(w, h, 0.0);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) { d(x, y) = noise(x, y); g(x, y) = noise(x, y) + 1e-9 * ((x * 7 + y * 13) % 11); }
    std::vector<float> e = ref;
    orc_fill_f32(e.data(), w, h, 8);
    rdgpu::FillDepressions<Topo::D8>(d);
    bool same = true;
    for (size_t i = 0; i < e.size(); i++) same &= d.data()[i] == (double)e[i];
    EXPECT(same);
    std::vector<double> eg(g.data(), g.data() + (size_t)w * h);
    orc_fill_f64(eg.data(), w, h, 8);
    rdgpu::FillDepressions<Topo::D8>(g);
    EXPECT(std::memcmp(g.data(), eg.data(), eg.size() * 8) == 0);
  }
  // rd_depressions_mask: pit_mask<Topology::D8>(elevation, mask)
  {
    Arr<uint8_t> mask;
    rdgpu::pit_mask<Topo::D8>(dem, mask);
    std::vector<uint8_t> e((size_t)w * h);
    orc_pit_mask_f32(dem.data(), -9999.0f, w, h, 8, e.data());
    EXPECT(mask.width() == w && mask.height() == h && mask.noData() == 3);
    EXPECT(std::memcmp(mask.data(), e.data(), e.size()) == 0);
  }
  // rd.ResolveFlats: ResolveFlatsEpsilon(Array2D<T>&) on the filled DEM
  {
    Arr<float> a(dem);
    a.setNoData(-9999.0f);
    for (size_t i = 0; i < ref.size(); i++) a.data()[i] = ref[i];
    rdgpu::FillDepressions<Topo::D8>(a);
    std::vector<float> e(a.data(), a.data() + (size_t)w * h);
    orc_resolve_flats_epsilon_f32(e.data(), -9999.0f, w, h);
    rdgpu::ResolveFlatsEpsilon(a);
    EXPECT(std::memcmp(a.data(), e.data(), e.size() * 4) == 0);
  }
  // rd.FlowProportions / rd.FlowAccumFromProps: FM_*(elevations, Array3D<float>&) then FlowAccumulation(props, accum)
  {
    Arr<float> filled(dem);
    filled.setNoData(-9999.0f);
    orc_fill_f32(filled.data(), w, h, 8);
    Arr3 props(filled);
    rdgpu::FM_D8(filled, props);
    EXPECT(props.noData() == -2.0f);
    std::vector<float> e((size_t)w * h * 9);
    orc_fm_d8_f32(filled.data(), -9999.0f, w, h, e.data());
    EXPECT(std::memcmp(slots(props), e.data(), e.size() * 4) == 0);
    Arr<double> accum(filled, 1.0), ea(filled, 1.0);
    rdgpu::FlowAccumulation(props, accum);
    orc_fa_d8_f32(filled.data(), -9999.0f, w, h, ea.data());
    EXPECT(accum.noData() == -1.0);
    EXPECT(std::memcmp(accum.data(), ea.data(), (size_t)w * h * 8) == 0);
    rdgpu::FM_OCallaghan<Topo::D8>(filled, props);
    EXPECT(std::memcmp(slots(props), e.data(), e.size() * 4) == 0);
    rdgpu::FM_Quinn(filled, props);
    orc_fm_mfd_f32(filled.data(), -9999.0f, w, h, 2, 1.0, e.data());
    size_t bad = 0;
    for (size_t i = 0; i < e.size(); i++)
      if (std::fabs(slots(props)[i] - e[i]) > 3e-7f * std::fabs(e[i])) bad++;
    EXPECT(bad == 0);
    rdgpu::FM_Holmgren(filled, props, 2.0);
    rdgpu::FM_Freeman(filled, props, 1.1);
    rdgpu::FM_D4(filled, props);
    rdgpu::FM_Tarboton(filled, props);
    Arr<double> wrong(w + 1, h, 1.0);
    bool threw = false;
    try { rdgpu::FlowAccumulation(props, wrong); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw);
  }
  // native raster format round trip (reference saveToCache / Array2D(filename, native=true), Array2D.hpp:209-281)
  {
    Arr<float> a(dem);
    a.setNoData(-9999.0f);
    a.geotransform = {10., 2., 0., 20., 0., -2.};
    a.projection = "PROJCS[test]";
    const std::string path = "/tmp/rdgpu_shim_test_native.rd";
    a.saveToCache(path);
    Arr<float> b(path, true);
    EXPECT(b == a);
    EXPECT(b.geotransform == a.geotransform && b.projection == a.projection);
    bool threw = false;
    try { Arr<float> c(std::string("/nonexistent/file.rd"), true); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw);
    std::remove(path.c_str());
  }
  // int8_t DEMs (bound by the reference, wrappers/pyrichdem/src/pywrapper.cpp:25-45) run; what the engine does not take
  // -> std::runtime_error, the reference's error convention (Priority-Flood+Epsilon on integers: Barnes2014.hpp:424-451)
  {
    Arr<int8_t> d(8, 8, 1);
    d(3, 3) = -5;
    rdgpu::FillDepressions<Topo::D8>(d);
    EXPECT(d(3, 3) == 1);
    bool threw = false;
    try { rdgpu::PriorityFloodEpsilon_Barnes2014<Topo::D8>(d); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw);
    // 64-bit element types run on dense value ranks (csrc/fill64.hip): pit_mask, max_dep, watersheds as well as the fill
    Arr<int64_t> e(8, 8, (int64_t)1 << 40);
    e(3, 3) = -((int64_t)1 << 41);
    e(3, 4) = 5;
    Arr<uint8_t> m;
    rdgpu::pit_mask<Topo::D8>(e, m);
    EXPECT(m(3, 3) == 1 && m(3, 4) == 1 && m(0, 0) == 0 && m(2, 2) == 0);
    rdgpu::PriorityFlood_Barnes2014_max_dep<Topo::D8>(e, 1);      // the pit has two cells: left alone
    EXPECT(e(3, 3) == -((int64_t)1 << 41));
    rdgpu::PriorityFlood_Barnes2014_max_dep<Topo::D8>(e, 2);
    EXPECT(e(3, 3) == ((int64_t)1 << 40) && e(3, 4) == ((int64_t)1 << 40));
    // barnes_flat_resolution_d8(alter = true) on an integer DEM: the reference's towards-zero steps (flat_resolution.hpp:567)
    Arr<int32_t> fl(9, 9, 50);
    for (int x = 0; x < 9; x++) fl(x, 8) = 10;                    // the flat drains over its lower edge
    Arr<uint8_t> fd;
    rdgpu::barnes_flat_resolution_d8(fl, fd, true);
    EXPECT(fl(4, 4) == 42 && fl(4, 1) == 46 && fl(4, 7) == 48 && fd(4, 4) == 0 && fd(1, 1) == 6);   // (what the reference returns)
  }
  // the other outputs of the sweep through the shim: epsilon fill, bounded depressions, watershed labels
  {
    Arr<float> d(9, 9, 10.0f);
    d(4, 4) = 1.0f; d(4, 5) = 2.0f;
    Arr<float> e = d;
    rdgpu::PriorityFloodEpsilon_Barnes2014<Topo::D8>(e);
    EXPECT(e(4, 4) > 10.0f && e(4, 5) > 10.0f && e(0, 0) == 10.0f);
    Arr<float> f = d;
    rdgpu::PriorityFlood_Barnes2014_max_dep<Topo::D8>(f, 1);      // the pit has two cells: left alone
    EXPECT(f(4, 4) == 1.0f);
    rdgpu::PriorityFlood_Barnes2014_max_dep<Topo::D8>(f, 2);
    EXPECT(f(4, 4) == 10.0f && f(4, 5) == 10.0f);
    Arr<int32_t> lab;
    Arr<float> g = d;
    rdgpu::PriorityFloodWatersheds_Barnes2014<Topo::D8>(g, lab, true);
    EXPECT(lab.width() == 9 && lab.noData() == -1 && lab(0, 0) >= 1 && g(4, 4) == 10.0f);
  }
  std::printf(failures ? "shim_test: %d FAILURES\n" : "shim_test: all checks passed\n", failures);
  return failures ? 1 : 0;
}
