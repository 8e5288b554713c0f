// multi.hip -- d8_flow_accum of one raster over SEVERAL devices driven by this one process.
//
// The library form of the reference's tiled accumulation driver (programs/parallel_d8_accum/main.cpp: the tile's own
// accumulation :373-464, where the flow entering at a perimeter cell leaves the tile again :270-334, FollowPathAdd
// :344-370), on row blocks, with the ONE-exchange protocol of richdem_amd/sharded.py::d8_flow_accum_sharded:
//   1. every device (a host thread each, common.hpp per_device): upload its rows of the direction raster plus one ghost
//      row per cut, accumulate the block on its own (rdgpu_accum_shard_begin_local), report per cut-row cell what its own
//      cells send across the cut (outbox) and where flow entering there leaves the block again (links);
//   2. this thread: the forest over the 2 * width * blocks cut-row cells (Kahn): the inflow of every entry cell;
//   3. every device: the inflows added along their paths (rdgpu_accum_shard_add_paths), the block written in the
//      requested type and copied back.
// Direction loops (never in directions derived from a DEM) are reported by step 1 / 2; the raster then goes through
// the single-device entry on devices[0], which keeps the reference's partial sums below a loop.
// A device may be listed more than once (its blocks are handled in order: what the one-GPU tests do).
#include "common.hpp"

#include <algorithm>
#include <string>
#include <vector>

extern "C" {
int rdgpu_d8_flow_accum_i32(const uint8_t *, uint8_t, int, int, int32_t *);
int rdgpu_d8_flow_accum_f32(const uint8_t *, uint8_t, int, int, float *);
int rdgpu_d8_flow_accum_f64(const uint8_t *, uint8_t, int, int, double *);
}

namespace rdgpu {

constexpr unsigned long long LOW56 = (1ull << 56) - 1ull;

static int finish_typed(rdgpu_accum_shard *a, int32_t *d) { return rdgpu_accum_shard_finish_i32(a, d); }
static int finish_typed(rdgpu_accum_shard *a, float *d) { return rdgpu_accum_shard_finish_f32(a, d); }
static int finish_typed(rdgpu_accum_shard *a, double *d) { return rdgpu_accum_shard_finish_f64(a, d); }
static int whole_typed(const uint8_t *d, uint8_t nd, int w, int h, int32_t *a) { return rdgpu_d8_flow_accum_i32(d, nd, w, h, a); }
static int whole_typed(const uint8_t *d, uint8_t nd, int w, int h, float *a) { return rdgpu_d8_flow_accum_f32(d, nd, w, h, a); }
static int whole_typed(const uint8_t *d, uint8_t nd, int w, int h, double *a) { return rdgpu_d8_flow_accum_f64(d, nd, w, h, a); }

// The forest over the cut-row cells.  boxes [S][2][w]: what every block's OWN cells send up (0) / down (1); links
// [S][2][w]: where the flow entering at a first (0) / last (1) row cell leaves its block again ((1 << 31 if across the
// lower cut) | receiving column, -1: nowhere).  inflow [S][2][w]: the flow entering every first / last row cell from
// outside its block.  false: the links form a loop across the cuts.
static bool link_solve(int S, int w, const std::vector<unsigned long long> &boxes, const std::vector<int32_t> &links,
                       std::vector<unsigned long long> &inflow) {
  const size_t n = (size_t)S * 2 * w;
  inflow.assign(n, 0ull);
  auto at = [&](int s, int r, int x) { return ((size_t)s * 2 + r) * w + x; };
  for (int s = 0; s < S; s++)
    for (int x = 0; x < w; x++) {
      if (s > 0) inflow[at(s, 0, x)] = boxes[at(s - 1, 1, x)] & LOW56;       // my first row receives what the block above sent down
      if (s + 1 < S) inflow[at(s, 1, x)] = boxes[at(s + 1, 0, x)] & LOW56;   // my last row: what the block below sent up
    }
  std::vector<int64_t> dst(n, -1);
  std::vector<uint32_t> indeg(n, 0);
  for (int s = 0; s < S; s++)
    for (int r = 0; r < 2; r++)
      for (int x = 0; x < w; x++) {
        const int32_t lk = links[at(s, r, x)];
        if (lk == -1) continue;
        const bool down = lk < 0;
        const int col = (int)((uint32_t)lk & 0x7FFFFFFFu);
        const int t = down ? s + 1 : s - 1;
        if (t < 0 || t >= S || col >= w) throw Error(RDGPU_ERR_HIP, "rdgpu_d8_flow_accum_multi: link out of range (internal error)");
        dst[at(s, r, x)] = (int64_t)at(t, down ? 0 : 1, col);
        indeg[(size_t)dst[at(s, r, x)]]++;
      }
  std::vector<size_t> queue;
  queue.reserve(n);
  size_t alive = 0;
  for (size_t i = 0; i < n; i++) {
    if (dst[i] < 0) continue;
    alive++;
    if (indeg[i] == 0) queue.push_back(i);
  }
  // (nodes without an outgoing link only receive; they need no turn of their own)
  for (size_t q = 0; q < queue.size(); q++) {
    const size_t i = queue[q], d = (size_t)dst[i];
    inflow[d] += inflow[i];
    alive--;
    if (--indeg[d] == 0 && dst[d] >= 0) queue.push_back(d);
  }
  return alive == 0;
}

template <class A>
static void d8_flow_accum_multi_host(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area, const int *devices, int ndev) {
  if (!dirs || !area || !devices) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum_multi: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum_multi: width and height must be positive");
  if (ndev < 1 || h < ndev) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum_multi: need at least one row per device");
  int ndevices = 0;
  RD_HIP(hipGetDeviceCount(&ndevices));
  for (int s = 0; s < ndev; s++)
    if (devices[s] < 0 || devices[s] >= ndevices) throw Error(RDGPU_ERR_ARG, "rdgpu_d8_flow_accum_multi: no such device");
  const int S = ndev;
  std::vector<int> r0(S + 1);
  for (int s = 0; s <= S; s++) r0[s] = (int)((int64_t)h * s / S);
  std::vector<rdgpu_accum_shard *> sh(S, nullptr);
  std::vector<hipStream_t> st(S, nullptr);   // ONE stream per device (its blocks share the device's named scratch)
  std::vector<char> owns(S, 0);              // the first block of a device owns the stream
  std::vector<uint8_t *> d_ext(S, nullptr);
  std::vector<unsigned long long> boxes((size_t)S * 2 * w), inflow, pending(S, 0);
  std::vector<int32_t> links((size_t)S * 2 * w);
  int home = 0;
  RD_HIP(hipGetDevice(&home));
  auto cleanup = [&]() noexcept {
    for (int s = 0; s < S; s++) {
      if (hipSetDevice(devices[s]) != hipSuccess) continue;
      if (sh[s]) { rdgpu_accum_shard_free(sh[s]); sh[s] = nullptr; }
      if (st[s] && owns[s]) { (void)hipStreamSynchronize(st[s]); (void)hipStreamDestroy(st[s]); }
      st[s] = nullptr;
    }
    (void)hipSetDevice(home);
  };
  bool loops = false;
  try {
    per_device(devices, S, [&](int, const std::vector<int> &mine) {
      Workspace &ws = Workspace::get();
      RD_HIP(hipStreamCreateWithFlags(&st[mine[0]], hipStreamNonBlocking));
      owns[mine[0]] = 1;
      for (int s : mine) st[s] = st[mine[0]];
      for (int s : mine) {
        const int rows = r0[s + 1] - r0[s], ga = s > 0 ? 1 : 0, gb = s + 1 < S ? 1 : 0;
        d_ext[s] = ws.buf<uint8_t>(("multi.dirs." + std::to_string(s)).c_str(), (size_t)(rows + 2) * w);
        // own rows and the ghost rows above / below in one copy (they are contiguous in the caller's raster)
        RD_HIP(hipMemcpyAsync(d_ext[s] + (size_t)(1 - ga) * w, dirs + (size_t)(r0[s] - ga) * w, (size_t)(rows + ga + gb) * w,
                              hipMemcpyHostToDevice, st[s]));
      }
      for (int s : mine) {
        const int rows = r0[s + 1] - r0[s];
        uint8_t *own = d_ext[s] + (size_t)w;
        int rc = rdgpu_accum_shard_begin_local(own, nodata, w, rows, s > 0 ? d_ext[s] : nullptr,
                                               s + 1 < S ? own + (size_t)rows * w : nullptr, st[s], &sh[s]);
        if (rc) throw Error(rc, rdgpu_last_error());
        unsigned long long *d_box = ws.buf<unsigned long long>(("multi.box." + std::to_string(s)).c_str(), (size_t)2 * w + 1);
        int32_t *d_links = ws.buf<int32_t>(("multi.links." + std::to_string(s)).c_str(), (size_t)2 * w);
        rc = rdgpu_accum_shard_outbox(sh[s], d_box);
        if (rc) throw Error(rc, rdgpu_last_error());
        rc = rdgpu_accum_shard_links(sh[s], d_links, d_box + 2 * w);
        if (rc) throw Error(rc, rdgpu_last_error());
        RD_HIP(hipMemcpyAsync(&boxes[(size_t)s * 2 * w], d_box, (size_t)2 * w * 8, hipMemcpyDeviceToHost, st[s]));
        RD_HIP(hipMemcpyAsync(&pending[s], d_box + 2 * w, 8, hipMemcpyDeviceToHost, st[s]));
        RD_HIP(hipMemcpyAsync(&links[(size_t)s * 2 * w], d_links, (size_t)2 * w * 4, hipMemcpyDeviceToHost, st[s]));
      }
      RD_HIP(hipStreamSynchronize(st[mine[0]]));
    });
    for (int s = 0; s < S; s++) loops |= pending[s] != 0;
    if (!loops) loops = !link_solve(S, w, boxes, links, inflow);
    if (!loops) {
      per_device(devices, S, [&](int, const std::vector<int> &mine) {
        Workspace &ws = Workspace::get();
        for (int s : mine) {
          const int rows = r0[s + 1] - r0[s];
          unsigned long long *d_in = ws.buf<unsigned long long>(("multi.box." + std::to_string(s)).c_str(), (size_t)2 * w + 1);
          RD_HIP(hipMemcpyAsync(d_in, &inflow[(size_t)s * 2 * w], (size_t)2 * w * 8, hipMemcpyHostToDevice, st[s]));
          int rc = rdgpu_accum_shard_add_paths(sh[s], s > 0 ? d_in : nullptr, s + 1 < S ? d_in + w : nullptr);
          if (rc) throw Error(rc, rdgpu_last_error());
          A *d_area = ws.buf<A>(("multi.area." + std::to_string(s)).c_str(), (size_t)rows * w);
          rdgpu_accum_shard *p = sh[s];
          sh[s] = nullptr;
          rc = finish_typed(p, d_area);   // synchronises the block's stream
          if (rc) throw Error(rc, rdgpu_last_error());
          RD_HIP(hipMemcpyAsync(area + (size_t)r0[s] * w, d_area, (size_t)rows * w * sizeof(A), hipMemcpyDeviceToHost, st[s]));
        }
        RD_HIP(hipStreamSynchronize(st[mine[0]]));   // a failed copy must not return RDGPU_OK with a partly written raster
        RD_HIP(hipStreamDestroy(st[mine[0]]));
        for (int s : mine) { st[s] = nullptr; owns[s] = 0; }
      });
    }
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
  if (loops) {   // direction loops: the whole raster on one device, with the reference's partial sums
    DeviceGuard g(devices[0]);
    multi_route_off() = true;   // (RDGPU_DEVICES would send the call straight back here)
    const int rc = whole_typed(dirs, nodata, w, h, area);
    multi_route_off() = false;
    if (rc) throw Error(rc, rdgpu_last_error());
  }
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_ACCUM_MULTI_API(SUF, A)                                                                              \
  extern "C" int rdgpu_d8_flow_accum_multi_##SUF(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area,    \
                                                 const int *devices, int ndev) {                                \
    return unlocked([&] { d8_flow_accum_multi_host<A>(dirs, nodata, w, h, area, devices, ndev); });             \
  }
RD_ACCUM_MULTI_API(i32, int32_t)
RD_ACCUM_MULTI_API(f32, float)
RD_ACCUM_MULTI_API(f64, double)
