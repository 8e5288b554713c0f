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

// ---- the same forest on the GPU (r05: the exchange of the multi-device accumulation stays on the devices) -------------------
// gathered boxes / links in the layout above, on devices[0].  inflow[v] = what v's own neighbours across the cut sent +
// the inflow of every node whose chain of links passes through v: subtree sums of a forest, by pointer doubling --
// S_(k+1)(v) = S_k(v) + sum of S_k(u) over the u whose 2^k-th successor is v (every node upstream at distance [2^k, 2^(k+1))
// arrives through exactly one such u).  2 * width * blocks nodes (640 000 at S3 / 8), <= 40 rounds; a chain that has not
// ended then is a loop across the cuts (reported, as by link_solve).
constexpr uint32_t LK_NONE = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void k_ml_init(const unsigned long long *__restrict__ boxes, const int32_t *__restrict__ links, int S,
                                                 int w, unsigned long long *sum, uint32_t *succ, uint32_t *bad) {
  const uint32_t n = (uint32_t)S * 2u * (uint32_t)w, i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int s = (int)(i / (2u * (uint32_t)w)), r = (int)(i / (uint32_t)w) & 1, x = (int)(i % (uint32_t)w);
  auto at = [&](int s_, int r_, int x_) { return ((uint32_t)s_ * 2u + (uint32_t)r_) * (uint32_t)w + (uint32_t)x_; };
  unsigned long long v = 0;
  if (r == 0 && s > 0) v = boxes[at(s - 1, 1, x)] & LOW56;          // my first row receives what the block above sent down
  if (r == 1 && s + 1 < S) v = boxes[at(s + 1, 0, x)] & LOW56;      // my last row: what the block below sent up
  sum[i] = v;
  const int32_t lk = links[i];
  uint32_t d = LK_NONE;
  if (lk != -1) {
    const bool down = lk < 0;
    const int col = (int)((uint32_t)lk & 0x7FFFFFFFu), t = down ? s + 1 : s - 1;
    if (t < 0 || t >= S || col >= w) *bad = 1u;
    else d = at(t, down ? 0 : 1, col);
  }
  succ[i] = d;
}
__global__ __launch_bounds__(256) void k_ml_round(const unsigned long long *__restrict__ sum_in, unsigned long long *sum_out,
                                                  const uint32_t *__restrict__ succ_in, uint32_t *succ_out, uint32_t n, uint32_t *alive) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t d = succ_in[i];
  if (d != LK_NONE) {
    const unsigned long long v = sum_in[i];
    if (v) atomicAdd(&sum_out[d], v);
    const uint32_t dd = succ_in[d];
    succ_out[i] = dd;
    if (dd != LK_NONE) *alive = 1u;
  } else {
    succ_out[i] = LK_NONE;
  }
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
  // r05: boxes and links go from their devices to devices[0] by peer copies behind events, the forest is solved there on the
  // GPU and every block's inflows are pushed back -- the host sees one word per block (does it hold a loop).
  // RDGPU_MULTI_HOST_STAGED=1: through the host vectors above and the host Kahn order (r02-r04): A/B and tests.
  const char *hstaged = getenv("RDGPU_MULTI_HOST_STAGED");
  const bool staged = (hstaged && hstaged[0] == '1') || S == 1;
  std::vector<hipEvent_t> ev(S, nullptr);
  std::vector<unsigned long long *> d_boxv(S, nullptr);
  std::vector<int32_t *> d_linkv(S, nullptr);
  hipEvent_t ev_in = nullptr;
  hipStream_t s0 = nullptr;
  int home = 0;
  RD_HIP(hipGetDevice(&home));
  auto cleanup = [&]() noexcept {
    for (int s = 0; s < S; s++) {
      if (hipSetDevice(devices[s]) != hipSuccess) continue;
      if (sh[s]) { rdgpu_accum_shard_free(sh[s]); sh[s] = nullptr; }
      if (st[s] && owns[s]) { (void)hipStreamSynchronize(st[s]); (void)hipStreamDestroy(st[s]); }
      st[s] = nullptr;
      if (ev[s]) { (void)hipEventDestroy(ev[s]); ev[s] = nullptr; }
    }
    if (hipSetDevice(devices[0]) == hipSuccess) {
      if (s0) { (void)hipStreamSynchronize(s0); (void)hipStreamDestroy(s0); s0 = nullptr; }
      if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
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
        RD_HIP(hipMemcpyAsync(&pending[s], d_box + 2 * w, 8, hipMemcpyDeviceToHost, st[s]));
        if (staged) {
          RD_HIP(hipMemcpyAsync(&boxes[(size_t)s * 2 * w], d_box, (size_t)2 * w * 8, hipMemcpyDeviceToHost, st[s]));
          RD_HIP(hipMemcpyAsync(&links[(size_t)s * 2 * w], d_links, (size_t)2 * w * 4, hipMemcpyDeviceToHost, st[s]));
        } else {
          d_boxv[s] = d_box;
          d_linkv[s] = d_links;
          RD_HIP(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming));
          RD_HIP(hipEventRecord(ev[s], st[s]));
        }
      }
      RD_HIP(hipStreamSynchronize(st[mine[0]]));
    });
    for (int s = 0; s < S; s++) loops |= pending[s] != 0;
    if (!loops && staged) loops = !link_solve(S, w, boxes, links, inflow);
    if (!loops && !staged) {
      DeviceGuard g(devices[0]);
      Workspace &ws = Workspace::get();
      const uint32_t nn = (uint32_t)S * 2u * (uint32_t)w, grid = (nn + 255u) / 256u;
      unsigned long long *d_boxes = ws.buf<unsigned long long>("multi.l.boxes", nn), *sumA = ws.buf<unsigned long long>("multi.l.sumA", nn),
                         *sumB = ws.buf<unsigned long long>("multi.l.sumB", nn);
      int32_t *d_links_all = ws.buf<int32_t>("multi.l.links", nn);
      uint32_t *succA = ws.buf<uint32_t>("multi.l.succA", nn), *succB = ws.buf<uint32_t>("multi.l.succB", nn), *flags = ws.buf<uint32_t>("multi.l.flags", 2);
      uint32_t *hw = ws.host_words();
      for (int s = 1; s < S; s++) { enable_peer_access(devices[0], devices[s]); enable_peer_access(devices[s], devices[0]); }
      RD_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
      for (int s = 0; s < S; s++) {
        RD_HIP(hipStreamWaitEvent(s0, ev[s], 0));
        RD_HIP(hipMemcpyPeerAsync(d_boxes + (size_t)s * 2 * w, devices[0], d_boxv[s], devices[s], (size_t)2 * w * 8, s0));
        RD_HIP(hipMemcpyPeerAsync(d_links_all + (size_t)s * 2 * w, devices[0], d_linkv[s], devices[s], (size_t)2 * w * 4, s0));
      }
      RD_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), s0));
      RD_LAUNCH("multi.link_init", k_ml_init, dim3(grid), dim3(256), 0, s0, (const unsigned long long *)d_boxes, (const int32_t *)d_links_all, S, w,
                sumA, succA, flags + 1);
      for (int round = 0;; round++) {
        RD_HIP(hipMemcpyAsync(sumB, sumA, (size_t)nn * 8, hipMemcpyDeviceToDevice, s0));
        RD_HIP(hipMemsetAsync(flags, 0, sizeof(uint32_t), s0));
        RD_LAUNCH("multi.link_round", k_ml_round, dim3(grid), dim3(256), 0, s0, (const unsigned long long *)sumA, sumB, (const uint32_t *)succA,
                  succB, nn, flags);
        std::swap(sumA, sumB);
        std::swap(succA, succB);
        RD_HIP(hipMemcpyAsync(hw, flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s0));
        RD_HIP(hipStreamSynchronize(s0));
        if (hw[1]) throw Error(RDGPU_ERR_HIP, "rdgpu_d8_flow_accum_multi: link out of range (internal error)");
        if (!hw[0]) break;
        if (round >= 40) { loops = true; break; }   // a chain of links longer than any forest over these nodes: a loop across the cuts
      }
      if (!loops) {
        for (int s = 0; s < S; s++)   // every block's inflows onto its device (its outbox buffer is free again)
          RD_HIP(hipMemcpyPeerAsync(d_boxv[s], devices[s], sumA + (size_t)s * 2 * w, devices[0], (size_t)2 * w * 8, s0));
        RD_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
        RD_HIP(hipEventRecord(ev_in, s0));
      }
    }
    if (!loops) {
      per_device(devices, S, [&](int, const std::vector<int> &mine) {
        Workspace &ws = Workspace::get();
        for (int s : mine) {
          const int rows = r0[s + 1] - r0[s];
          unsigned long long *d_in = ws.buf<unsigned long long>(("multi.box." + std::to_string(s)).c_str(), (size_t)2 * w + 1);
          if (staged) RD_HIP(hipMemcpyAsync(d_in, &inflow[(size_t)s * 2 * w], (size_t)2 * w * 8, hipMemcpyHostToDevice, st[s]));
          else RD_HIP(hipStreamWaitEvent(st[s], ev_in, 0));   // (the inflows arrived in this very buffer)
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

// ------------------------------------------------------------------------------------------------------------------
// barnes_flat_resolution_d8(alter = false) of one raster over several devices.  The reference resolves flats on one
// node only; the row-block protocol is this engine's own (flats.hip, "Row-block shards"): a block
// holds its rows plus TWO ghost rows per cut, relaxes each level field to its local fixed point, the cut rows are
// exchanged and the ghost rows lowered until no cut row changes any more (towards field first, then away), the deepest
// away level per flat is agreed on through one small union-find over the cut rows (on devices[0]), and every block
// writes the directions of its own rows.  Driver = richdem_amd/sharded.py::flat_exchange, here for the devices of one
// process: the relaxations synchronise with their device per batch of rounds, hence a host thread per device.  (r05) What the
// blocks exchange -- cut rows, heights, solved levels -- is copied from device to device.
// ------------------------------------------------------------------------------------------------------------------
#define RD_FS_BEGIN(SUF, T)                                                                                        \
  static int fs_begin_typed(const T *d, T nd, int w, int rows, int gt, int gb, hipStream_t st, rdgpu_flat_shard **o) { \
    return rdgpu_flat_shard_begin_##SUF(d, nd, w, rows, gt, gb, (void *)st, o);                                      \
  }
RD_FS_BEGIN(u8, uint8_t) RD_FS_BEGIN(i8, int8_t) RD_FS_BEGIN(i16, int16_t) RD_FS_BEGIN(u16, uint16_t) RD_FS_BEGIN(i32, int32_t)
RD_FS_BEGIN(u32, uint32_t) RD_FS_BEGIN(f32, float) RD_FS_BEGIN(f64, double) RD_FS_BEGIN(i64, int64_t) RD_FS_BEGIN(u64, uint64_t)
#undef RD_FS_BEGIN

// r05: a block's received cut rows against the ones it received the round before (and kept for the next comparison)
__global__ __launch_bounds__(256) void k_cut_changed(const int32_t *__restrict__ in, int32_t *prev, uint32_t lo, uint32_t hi, int first,
                                                     uint32_t *flag) {
  const uint32_t i = lo + blockIdx.x * 256u + threadIdx.x;
  if (i >= hi) return;
  const int32_t v = in[i];
  if (first || prev[i] != v) *flag = 1u;
  prev[i] = v;
}

template <class T>
static void flat_resolution_multi_host(const T *dem, T nodata, int w, int h, uint8_t *dirs, const int *devices, int ndev) {
  if (!dem || !dirs || !devices) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_multi: null pointer");
  if (w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_multi: width and height must be positive");
  if (ndev < 1 || (ndev > 1 && h / ndev < 2)) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_multi: need >= 2 rows per device");
  int ndevices = 0;
  RD_HIP(hipGetDeviceCount(&ndevices));
  for (int s = 0; s < ndev; s++)
    if (devices[s] < 0 || devices[s] >= ndevices) throw Error(RDGPU_ERR_ARG, "rdgpu_flat_resolution_d8_multi: no such device");
  const int S = ndev;
  std::vector<int> r0(S + 1);
  for (int s = 0; s <= S; s++) r0[s] = (int)((int64_t)h * s / S);
  std::vector<rdgpu_flat_shard *> sh(S, nullptr);
  std::vector<T *> d_ext(S, nullptr);
  std::vector<int32_t> cut((size_t)S * 2 * w), prev, heights((size_t)S * 8 * w), solved((size_t)S * 4 * w);
  const char *hstaged = getenv("RDGPU_MULTI_HOST_STAGED");
  const bool staged = (hstaged && hstaged[0] == '1') || S == 1;
  int home = 0;
  RD_HIP(hipGetDevice(&home));
  auto cleanup = [&]() noexcept {
    for (int s = 0; s < S; s++) {
      if (!sh[s] || hipSetDevice(devices[s]) != hipSuccess) continue;
      rdgpu_flat_shard_free(sh[s]);
      sh[s] = nullptr;
    }
    (void)hipSetDevice(home);
  };
  auto name = [](const char *what, int s) { return std::string("multi.") + what + "." + std::to_string(s); };
  try {
    per_device(devices, S, [&](int, const std::vector<int> &mine) {
      Workspace &ws = Workspace::get();
      for (int s : mine) {   // own rows + two ghost rows per cut: contiguous in the caller's raster
        const int gt = s > 0 ? 2 : 0, gb = s + 1 < S ? 2 : 0, rows = r0[s + 1] - r0[s] + gt + gb;
        d_ext[s] = ws.buf<T>(name("dem", s).c_str(), (size_t)rows * w);
        RD_HIP(hipMemcpy(d_ext[s], dem + (size_t)(r0[s] - gt) * w, (size_t)rows * w * sizeof(T), hipMemcpyHostToDevice));
        const int rc = fs_begin_typed(d_ext[s], nodata, w, rows, gt, gb, nullptr, &sh[s]);
        if (rc) throw Error(rc, rdgpu_last_error());
      }
    });
    // r05: the cut rows, the heights and the solved levels go from device to device (peer copies); the host sees one word per
    // block and round ("did a row I received change").  RDGPU_MULTI_HOST_STAGED=1: through the host vectors (r03-r04).
    std::vector<int32_t *> d_cut(S, nullptr);
    std::vector<char> changed(S, 0);
    if (!staged)
      for (int s = 0; s < S; s++)
        for (int t = std::max(0, s - 1); t <= std::min(S - 1, s + 1); t++)
          if (t != s) enable_peer_access(devices[s], devices[t]);
    for (int phase = 0; phase < 2; phase++) {   // towards the low edges first: it also tells which flats have an outlet
      prev.clear();
      for (int round = 0;; round++) {
        per_device(devices, S, [&](int, const std::vector<int> &mine) {
          Workspace &ws = Workspace::get();
          for (int s : mine) {
            int rc = rdgpu_flat_shard_relax(sh[s], phase);
            if (rc) throw Error(rc, rdgpu_last_error());
            int32_t *d_b = ws.buf<int32_t>(name("cut", s).c_str(), (size_t)2 * w);
            d_cut[s] = d_b;
            rc = rdgpu_flat_shard_boundary(sh[s], phase, d_b);
            if (rc) throw Error(rc, rdgpu_last_error());
            RD_HIP(hipStreamSynchronize(nullptr));
            if (staged) RD_HIP(hipMemcpy(&cut[(size_t)s * 2 * w], d_b, (size_t)2 * w * 4, hipMemcpyDeviceToHost));
          }
        });
        if (staged && !prev.empty() && prev == cut) break;   // no cut row changed anywhere: the field is final
        per_device(devices, S, [&](int, const std::vector<int> &mine) {
          Workspace &ws = Workspace::get();
          for (int s : mine) {
            int32_t *d_in = ws.buf<int32_t>(name("cutin", s).c_str(), (size_t)2 * w);
            // the last own row of the block above, the first own row of the block below
            if (staged) {
              if (s > 0) RD_HIP(hipMemcpy(d_in, &cut[((size_t)(s - 1) * 2 + 1) * w], (size_t)w * 4, hipMemcpyHostToDevice));
              if (s + 1 < S) RD_HIP(hipMemcpy(d_in + w, &cut[((size_t)(s + 1) * 2) * w], (size_t)w * 4, hipMemcpyHostToDevice));
            } else {
              if (s > 0) RD_HIP(hipMemcpyPeer(d_in, devices[s], d_cut[s - 1] + w, devices[s - 1], (size_t)w * 4));
              if (s + 1 < S) RD_HIP(hipMemcpyPeer(d_in + w, devices[s], d_cut[s + 1], devices[s + 1], (size_t)w * 4));
              int32_t *d_prev = ws.buf<int32_t>(name("cutprev", s).c_str(), (size_t)2 * w);
              uint32_t *flag = ws.buf<uint32_t>(name("cutflag", s).c_str(), 1);
              const uint32_t lo = s > 0 ? 0u : (uint32_t)w, hi = s + 1 < S ? 2u * (uint32_t)w : (uint32_t)w;
              uint32_t hv = 0;
              RD_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), nullptr));
              if (hi > lo)
                RD_LAUNCH("multi.cut_changed", k_cut_changed, dim3((hi - lo + 255u) / 256u), dim3(256), 0, nullptr, (const int32_t *)d_in, d_prev,
                          lo, hi, round == 0 ? 1 : 0, flag);
              RD_HIP(hipMemcpy(&hv, flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
              changed[s] = hv != 0;
              if (!changed[s]) continue;   // the same rows as last round: injecting them again changes nothing
            }
            const int rc = rdgpu_flat_shard_inject(sh[s], phase, s > 0 ? d_in : nullptr, s + 1 < S ? d_in + w : nullptr);
            if (rc) throw Error(rc, rdgpu_last_error());
          }
        });
        if (staged) prev = cut;
        else if (std::none_of(changed.begin(), changed.end(), [](char c) { return c != 0; })) break;   // no received row changed anywhere
      }
    }
    std::vector<int32_t *> d_hv(S, nullptr);
    per_device(devices, S, [&](int, const std::vector<int> &mine) {
      Workspace &ws = Workspace::get();
      for (int s : mine) {
        int32_t *d_h = ws.buf<int32_t>(name("heights", s).c_str(), (size_t)8 * w);
        d_hv[s] = d_h;
        const int rc = rdgpu_flat_shard_heights(sh[s], d_h);
        if (rc) throw Error(rc, rdgpu_last_error());
        RD_HIP(hipStreamSynchronize(nullptr));
        if (staged) RD_HIP(hipMemcpy(&heights[(size_t)s * 8 * w], d_h, (size_t)8 * w * 4, hipMemcpyDeviceToHost));
      }
    });
    int32_t *d_solved_all = nullptr;
    {
      DeviceGuard g(devices[0]);
      Workspace &ws = Workspace::get();
      int32_t *d_g = ws.buf<int32_t>("multi.heights_all", heights.size()), *d_o = ws.buf<int32_t>("multi.solved_all", solved.size());
      if (staged) RD_HIP(hipMemcpy(d_g, heights.data(), heights.size() * 4, hipMemcpyHostToDevice));
      else
        for (int s = 0; s < S; s++) {
          enable_peer_access(devices[0], devices[s]);
          enable_peer_access(devices[s], devices[0]);
          RD_HIP(hipMemcpyPeer(d_g + (size_t)s * 8 * w, devices[0], d_hv[s], devices[s], (size_t)8 * w * 4));
        }
      const int rc = rdgpu_flat_graph_solve_dev(d_g, S, w, d_o, nullptr);
      if (rc) throw Error(rc, rdgpu_last_error());
      RD_HIP(hipStreamSynchronize(nullptr));
      if (staged) RD_HIP(hipMemcpy(solved.data(), d_o, solved.size() * 4, hipMemcpyDeviceToHost));
      d_solved_all = d_o;
    }
    per_device(devices, S, [&](int, const std::vector<int> &mine) {
      Workspace &ws = Workspace::get();
      for (int s : mine) {
        const int rows = r0[s + 1] - r0[s];
        int32_t *d_s = ws.buf<int32_t>(name("solved", s).c_str(), (size_t)4 * w);
        uint8_t *d_d = ws.buf<uint8_t>(name("dirs", s).c_str(), (size_t)rows * w);
        if (staged) RD_HIP(hipMemcpy(d_s, &solved[(size_t)s * 4 * w], (size_t)4 * w * 4, hipMemcpyHostToDevice));
        else RD_HIP(hipMemcpyPeer(d_s, devices[s], d_solved_all + (size_t)s * 4 * w, devices[0], (size_t)4 * w * 4));
        const int rc = rdgpu_flat_shard_finish(sh[s], d_s, d_d);   // (does not release the handle)
        if (rc) throw Error(rc, rdgpu_last_error());
        RD_HIP(hipStreamSynchronize(nullptr));
        RD_HIP(hipMemcpy(dirs + (size_t)r0[s] * w, d_d, (size_t)rows * w, hipMemcpyDeviceToHost));
        rdgpu_flat_shard_free(sh[s]);
        sh[s] = nullptr;
      }
    });
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
}

}  // namespace rdgpu

using namespace rdgpu;

#define RD_FLATS_MULTI_API(SUF, T)                                                                              \
  extern "C" int rdgpu_flat_resolution_d8_multi_##SUF(const T *dem, T nodata, int w, int h, uint8_t *dirs,      \
                                                      const int *devices, int ndev) {                           \
    return unlocked([&] { flat_resolution_multi_host<T>(dem, nodata, w, h, dirs, devices, ndev); });            \
  }
RD_FLATS_MULTI_API(u8, uint8_t) RD_FLATS_MULTI_API(i8, int8_t) RD_FLATS_MULTI_API(i16, int16_t) RD_FLATS_MULTI_API(u16, uint16_t)
RD_FLATS_MULTI_API(i32, int32_t) RD_FLATS_MULTI_API(u32, uint32_t) RD_FLATS_MULTI_API(f32, float) RD_FLATS_MULTI_API(f64, double)
RD_FLATS_MULTI_API(i64, int64_t) RD_FLATS_MULTI_API(u64, uint64_t)

#define RD_ACCUM_MULTI_API(SUF, A)                                                                              \
  extern "C" int rdgpu_d8_flow_accum_multi_##SUF(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area,    \
                                                 const int *devices, int ndev) {                                \
    return unlocked([&] { d8_flow_accum_multi_host<A>(dirs, nodata, w, h, area, devices, ndev); });             \
  }
RD_ACCUM_MULTI_API(i32, int32_t)
RD_ACCUM_MULTI_API(f32, float)
RD_ACCUM_MULTI_API(f64, double)
