// shardgraph.hip -- host side of the row-block shard protocol for the fill.
//
// Mirrors the producer of the reference's tiled Priority-Flood (Barnes 2016;
// programs/parallel_priority_flood/main.cpp: ProducerSpecifics::Calculations :401-547, HandleEdge /
// HandleCorner :344-398): the per-shard watershed graphs are joined along the cut rows and a
// Priority-Flood on that small graph, started from the outside, gives every cut-row watershed its
// final level.  Shards are row blocks, so every cut is a full row and the "corner" cases of the
// reference's 2-D tiling reduce to the diagonal neighbours x-1 / x+1 across a cut.
//
// Plain host C++ (no kernels): runs identically on a box without a GPU, which is how the
// world_size-2 gloo tests exercise it.
#include "common.hpp"

#include <algorithm>
#include <exception>
#include <map>
#include <queue>
#include <thread>
#include <utility>
#include <vector>

struct rdgpu_fill_shard;
extern "C" {
int rdgpu_fill_shard_edge_count(rdgpu_fill_shard *sh, uint32_t *n_edges);
int rdgpu_fill_shard_export(rdgpu_fill_shard *sh, uint32_t *top_keys, uint32_t *bottom_keys, uint32_t *edges);
int rdgpu_fill_shard_finish(rdgpu_fill_shard *sh, const uint32_t *levels);
int rdgpu_fill_shard_free(rdgpu_fill_shard *sh);
int rdgpu_fill_graph_solve_dev(int nshards, int width, int topology, const uint32_t *d_keys_all, const uint32_t *d_edges_all,
                               const uint32_t *d_counts, uint32_t cap, uint32_t *d_levels_all, void *stream);
}

namespace rdgpu {

constexpr uint32_t OUT_TID = 0xFFFFFFFFu;

// keys:   [nshards][2][w]  order-preserving keys of each shard's top (0) and bottom (1) row
// edges:  concatenated (a, b, pass) triples per shard, a/b = terminal ids in [0, 2w) (top row: x,
//         bottom row: w + x) or OUT_TID; edge_offsets[s] .. edge_offsets[s+1] are shard s's triples
// levels: [nshards][2][w]  out: final level key of every cut-row terminal (0 where not a terminal)
static void graph_solve(int nshards, int w, int topology, const uint32_t *keys, const uint32_t *edges,
                        const uint64_t *edge_offsets, uint32_t *levels) {
  if (nshards < 1 || w < 1 || !keys || !edge_offsets || !levels) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_graph_solve: bad arguments");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_graph_solve: topology must be 8 or 4");
  const size_t per = (size_t)2 * w;
  const uint32_t NOUT = (uint32_t)((size_t)nshards * per);   // node id of the outside
  std::fill(levels, levels + (size_t)nshards * per, 0u);
  if (nshards == 1) return;

  struct E { uint32_t a, b, w; };
  std::vector<E> el;
  el.reserve((size_t)edge_offsets[nshards] + (size_t)(nshards - 1) * w * 3);
  auto node = [&](int s, uint32_t tid) -> uint32_t {
    if (tid == OUT_TID) return NOUT;
    if (tid >= per) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_graph_solve: terminal id out of range");
    return (uint32_t)((size_t)s * per + tid);
  };
  for (int s = 0; s < nshards; s++)
    for (uint64_t e = edge_offsets[s]; e < edge_offsets[s + 1]; e++)
      el.push_back(E{node(s, edges[3 * e]), node(s, edges[3 * e + 1]), edges[3 * e + 2]});
  // edges across each cut: bottom row of shard s  <->  top row of shard s+1 (HandleEdge, main.cpp:344-378)
  for (int s = 0; s + 1 < nshards; s++) {
    const uint32_t *kb = keys + ((size_t)s * 2 + 1) * w, *kt = keys + ((size_t)(s + 1) * 2) * w;
    for (int x = 0; x < w; x++) {
      const uint32_t a = (x == 0 || x == w - 1) ? NOUT : node(s, (uint32_t)(w + x));   // side columns are true border
      for (int dx = -1; dx <= 1; dx++) {
        if (topology == 4 && dx != 0) continue;
        const int x2 = x + dx;
        if (x2 < 0 || x2 >= w) continue;
        const uint32_t b = (x2 == 0 || x2 == w - 1) ? NOUT : node(s + 1, (uint32_t)x2);
        if (a == NOUT && b == NOUT) continue;
        el.push_back(E{a, b, std::max(kb[x], kt[x2])});
      }
    }
  }
  // CSR adjacency
  const size_t nn = (size_t)NOUT + 1;
  std::vector<uint64_t> off(nn + 1, 0);
  for (const E &e : el) { off[e.a + 1]++; off[e.b + 1]++; }
  for (size_t i = 0; i < nn; i++) off[i + 1] += off[i];
  std::vector<std::pair<uint32_t, uint32_t>> adj(off[nn]);
  {
    std::vector<uint64_t> pos(off.begin(), off.end() - 1);
    for (const E &e : el) {
      adj[pos[e.a]++] = {e.b, e.w};
      adj[pos[e.b]++] = {e.a, e.w};
    }
  }
  // Priority-Flood on the graph from the outside (main.cpp:498-546): level[v] = min over paths of max pass
  std::vector<uint32_t> lvl(nn, 0xFFFFFFFFu);
  std::vector<uint8_t> done(nn, 0);
  typedef std::pair<uint32_t, uint32_t> QE;   // (level, node)
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
  lvl[NOUT] = 0;
  pq.push({0u, NOUT});
  while (!pq.empty()) {
    const QE t = pq.top();
    pq.pop();
    const uint32_t u = t.second;
    if (done[u]) continue;
    done[u] = 1;
    for (uint64_t i = off[u]; i < off[u + 1]; i++) {
      const uint32_t v = adj[i].first, cand = std::max(t.first, adj[i].second);
      if (!done[v] && cand < lvl[v]) {
        lvl[v] = cand;
        pq.push({cand, v});
      }
    }
  }
  for (int s = 0; s < nshards; s++)
    for (int r = 0; r < 2; r++) {
      const bool cut = (r == 0) ? s > 0 : s + 1 < nshards;
      if (!cut) continue;
      for (int x = 1; x < w - 1; x++) {
        const size_t i = (size_t)s * per + (size_t)r * w + x;
        if (lvl[i] == 0xFFFFFFFFu) throw Error(RDGPU_ERR_HIP, "rdgpu_fill_graph_solve: a cut-row terminal is not connected to the outside");
        levels[i] = lvl[i];
      }
    }
}

// Whole-DEM fill through the shard protocol on ONE GPU, shard after shard: the same code path the
// multi-GPU ranks run, minus the all-gather.  Used for tiling-invariance tests and as an out-of-core
// style entry point.
template <class T, class Begin>
static void fill_sharded_host(T *dem, int w, int h, int topology, int nshards, Begin begin) {
  if (!dem || w <= 0 || h <= 0) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_sharded: bad arguments");
  if (nshards < 1 || (nshards > 1 && h / nshards < 2)) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_sharded: need >= 2 rows per shard");
  const size_t n = (size_t)w * h;
  T *d = Workspace::get().buf<T>("host.dem", n);
  RD_HIP(hipMemcpy(d, dem, n * sizeof(T), hipMemcpyHostToDevice));
  std::vector<rdgpu_fill_shard *> sh(nshards, nullptr);
  std::vector<int> r0(nshards + 1);
  for (int s = 0; s <= nshards; s++) r0[s] = (int)((int64_t)h * s / nshards);
  try {
    const size_t per = (size_t)2 * w;
    std::vector<uint32_t> keys((size_t)nshards * per, 0), levels((size_t)nshards * per, 0), edges;
    std::vector<uint64_t> offs(nshards + 1, 0);
    for (int s = 0; s < nshards; s++) {
      const int rc = begin(d + (size_t)r0[s] * w, w, r0[s + 1] - r0[s], topology, s > 0, s + 1 < nshards, &sh[s]);
      if (rc) throw Error(rc, rdgpu_last_error());
      uint32_t ne = 0;
      rdgpu_fill_shard_edge_count(sh[s], &ne);
      offs[s + 1] = offs[s] + ne;
      edges.resize((size_t)offs[s + 1] * 3);
      const int rc2 = rdgpu_fill_shard_export(sh[s], &keys[(size_t)s * per], &keys[(size_t)s * per + w],
                                              ne ? &edges[(size_t)offs[s] * 3] : nullptr);
      if (rc2) throw Error(rc2, rdgpu_last_error());
    }
    graph_solve(nshards, w, topology, keys.data(), edges.data(), offs.data(), levels.data());
    for (int s = 0; s < nshards; s++) {
      rdgpu_fill_shard *p = sh[s];
      sh[s] = nullptr;
      const int rc = rdgpu_fill_shard_finish(p, &levels[(size_t)s * per]);
      if (rc) throw Error(rc, rdgpu_last_error());
    }
  } catch (...) {
    for (auto *p : sh) if (p) rdgpu_fill_shard_free(p);
    throw;
  }
  RD_HIP(hipDeviceSynchronize());
  RD_HIP(hipMemcpy(dem, d, n * sizeof(T), hipMemcpyDeviceToHost));
}

// The same protocol over SEVERAL devices driven by this one process (the role of the reference's
// programs/parallel_priority_flood producer + consumers, main.cpp:276-330, :401-547, :700-800, and of its exchange layer
// include/richdem/common/communication.hpp): row block s lives on devices[s], is uploaded over that device's own PCIe link
// and filled locally there, the cut rows and spillover graphs meet on devices[0], the solved levels go back, and every
// device raises and returns its block.
// One HOST THREAD PER DEVICE: the local phase synchronises with its device several times per Boruvka round, so issuing
// the blocks from one thread would run them one after another (r02).  A worker holds its device's API lock for its
// phase (common.hpp: locks are per device), so the devices run side by side and a second caller on one of them waits.
// A device may be listed more than once: its blocks are then handled in order by that device's worker (what the
// one-GPU tests do -- they exercise the threads, the events, the peer copies and the joined solve, not the concurrency).
//
// THE EXCHANGE NEVER TOUCHES THE HOST (r05): every block exports its cut-row keys and edge triples into a buffer on ITS
// device (rdgpu_fill_shard_export_dev) and records an event; devices[0]'s stream waits for the events, pulls the exports
// into the joined layout with hipMemcpyPeerAsync (xGMI where peer access is available -- enabled here -- else the runtime's
// staging), solves the joined graph with the raster's own Boruvka kernels (rdgpu_fill_graph_solve_dev, 0.7 ms at S3 / 8
// blocks), pushes each block's 2 * width levels to its device and records one event that the blocks' streams wait for
// before rdgpu_fill_shard_finish_dev.  Only the edge COUNTS (one word per block) are read by the host, to size the joined
// buffer.  RDGPU_MULTI_HOST_STAGED=1: the r02-r04 exchange through host vectors and blocking copies (A/B and tests);
// RDGPU_MULTI_HOST_SOLVE=1 (implies it): the joined graph solved by the host Priority-Flood above.
template <class T, class Begin>
static void fill_multi_host(T *dem, int w, int h, int topology, const int *devices, int ndev, Begin begin) {
  if (!dem || w <= 0 || h <= 0 || !devices) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_multi: bad arguments");
  if (ndev < 1 || (ndev > 1 && h / ndev < 2)) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_multi: need >= 2 rows per device");
  if (topology != 8 && topology != 4) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_multi: topology must be 8 or 4");
  int ndevices = 0;
  RD_HIP(hipGetDeviceCount(&ndevices));
  for (int s = 0; s < ndev; s++)
    if (devices[s] < 0 || devices[s] >= ndevices) throw Error(RDGPU_ERR_ARG, "rdgpu_fill_multi: no such device");
  std::vector<rdgpu_fill_shard *> sh(ndev, nullptr);
  std::vector<T *> blk(ndev, nullptr);
  std::vector<hipStream_t> st(ndev, nullptr);
  std::vector<hipEvent_t> ev(ndev, nullptr);       // block s: its export is in its device buffer
  hipEvent_t ev_levels = nullptr;                  // devices[0]: every block's levels are on its device
  hipStream_t s0 = nullptr;                        // devices[0]: the exchange and the joined solve
  std::vector<uint32_t *> d_exp(ndev, nullptr), d_lev(ndev, nullptr);
  std::vector<uint32_t> counts(ndev, 0);
  std::vector<int> r0(ndev + 1);
  for (int s = 0; s <= ndev; s++) r0[s] = (int)((int64_t)h * s / ndev);
  const size_t per = (size_t)2 * w;
  const char *hsolve = getenv("RDGPU_MULTI_HOST_SOLVE"), *hstaged = getenv("RDGPU_MULTI_HOST_STAGED");
  const bool host_solve = hsolve && hsolve[0] == '1';
  const bool staged = host_solve || (hstaged && hstaged[0] == '1') || ndev == 1;
  std::vector<uint32_t> keys, levels;
  std::vector<std::vector<uint32_t>> edges(ndev);
  if (staged) { keys.assign((size_t)ndev * per, 0); levels.assign((size_t)ndev * per, 0); }
  auto cleanup = [&]() noexcept {   // error path only: best effort
    for (int s = 0; s < ndev; s++) {
      if (hipSetDevice(devices[s]) != hipSuccess) continue;
      if (sh[s]) rdgpu_fill_shard_free(sh[s]);
      if (st[s]) { (void)hipStreamSynchronize(st[s]); (void)hipStreamDestroy(st[s]); }
      if (ev[s]) (void)hipEventDestroy(ev[s]);
    }
    if (hipSetDevice(devices[0]) == hipSuccess) {
      if (s0) { (void)hipStreamSynchronize(s0); (void)hipStreamDestroy(s0); }
      if (ev_levels) (void)hipEventDestroy(ev_levels);
    }
  };
  int home = 0;
  RD_HIP(hipGetDevice(&home));
  try {
    if (!staged)
      for (int s = 1; s < ndev; s++) { enable_peer_access(devices[0], devices[s]); enable_peer_access(devices[s], devices[0]); }
    // 1. every device: upload its blocks (own PCIe link, own stream), local phase; the export stays on the device
    per_device(devices, ndev, [&](int, const std::vector<int> &mine) {
      for (int s : mine) {
        const size_t cells = (size_t)(r0[s + 1] - r0[s]) * w;
        blk[s] = Workspace::get().buf<T>(("multi.block." + std::to_string(s)).c_str(), cells);
        RD_HIP(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        RD_HIP(hipMemcpyAsync(blk[s], dem + (size_t)r0[s] * w, cells * sizeof(T), hipMemcpyHostToDevice, st[s]));
      }
      for (int s : mine) {
        const int rc = begin(blk[s], w, r0[s + 1] - r0[s], topology, s > 0, s + 1 < ndev, st[s], &sh[s]);
        if (rc) throw Error(rc, rdgpu_last_error());
        uint32_t ne = 0;
        rdgpu_fill_shard_edge_count(sh[s], &ne);
        counts[s] = ne;
        if (staged) {
          edges[s].resize((size_t)ne * 3);
          const int rc2 = rdgpu_fill_shard_export(sh[s], &keys[(size_t)s * per], &keys[(size_t)s * per + w],
                                                  ne ? edges[s].data() : nullptr);
          if (rc2) throw Error(rc2, rdgpu_last_error());
        } else {
          // [2 * w keys | 3 * ne edge words] and the block's levels, in buffers of this device
          d_exp[s] = Workspace::get().buf<uint32_t>(("multi.export." + std::to_string(s)).c_str(), per + (size_t)std::max(ne, 1u) * 3);
          d_lev[s] = Workspace::get().buf<uint32_t>(("multi.levels." + std::to_string(s)).c_str(), per);
          const int rc2 = rdgpu_fill_shard_export_dev(sh[s], d_exp[s], ne ? d_exp[s] + per : nullptr, ne);
          if (rc2) throw Error(rc2, rdgpu_last_error());
          RD_HIP(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming));
          RD_HIP(hipEventRecord(ev[s], st[s]));
        }
      }
    });
    if (!staged) {
      // 2. the exchange and the joined solve on devices[0]: events and peer copies, no host buffer
      DeviceGuard g(devices[0]);
      uint32_t cap = 1;
      for (int s = 0; s < ndev; s++) cap = std::max(cap, counts[s]);
      Workspace &ws = Workspace::get();
      uint32_t *d_keys = ws.buf<uint32_t>("multi.keys", (size_t)ndev * per), *d_edges = ws.buf<uint32_t>("multi.edges", (size_t)ndev * cap * 3);
      uint32_t *d_counts = ws.buf<uint32_t>("multi.counts", ndev), *d_levels = ws.buf<uint32_t>("multi.levels", (size_t)ndev * per);
      RD_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
      RD_HIP(hipMemcpyAsync(d_counts, counts.data(), (size_t)ndev * 4, hipMemcpyHostToDevice, s0));   // (the one host word per block)
      for (int s = 0; s < ndev; s++) {
        RD_HIP(hipStreamWaitEvent(s0, ev[s], 0));
        RD_HIP(hipMemcpyPeerAsync(d_keys + (size_t)s * per, devices[0], d_exp[s], devices[s], per * 4, s0));
        if (counts[s])
          RD_HIP(hipMemcpyPeerAsync(d_edges + (size_t)s * cap * 3, devices[0], d_exp[s] + per, devices[s], (size_t)counts[s] * 12, s0));
      }
      const int rc = rdgpu_fill_graph_solve_dev(ndev, w, topology, d_keys, d_edges, d_counts, cap, d_levels, (void *)s0);
      if (rc) throw Error(rc, rdgpu_last_error());
      for (int s = 0; s < ndev; s++)
        RD_HIP(hipMemcpyPeerAsync(d_lev[s], devices[s], d_levels + (size_t)s * per, devices[0], per * 4, s0));
      RD_HIP(hipEventCreateWithFlags(&ev_levels, hipEventDisableTiming));
      RD_HIP(hipEventRecord(ev_levels, s0));
    } else if (ndev > 1 && !host_solve) {
      DeviceGuard g(devices[0]);
      uint32_t cap = 1;
      for (int s = 0; s < ndev; s++) cap = std::max(cap, counts[s]);
      std::vector<uint32_t> padded((size_t)ndev * cap * 3, 0);
      for (int s = 0; s < ndev; s++) std::copy(edges[s].begin(), edges[s].end(), padded.begin() + (size_t)s * cap * 3);
      Workspace &ws = Workspace::get();
      uint32_t *d_keys = ws.buf<uint32_t>("multi.keys", keys.size()), *d_edges = ws.buf<uint32_t>("multi.edges", padded.size());
      uint32_t *d_counts = ws.buf<uint32_t>("multi.counts", ndev), *d_levels = ws.buf<uint32_t>("multi.levels", levels.size());
      RD_HIP(hipMemcpy(d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
      RD_HIP(hipMemcpy(d_edges, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
      RD_HIP(hipMemcpy(d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice));
      const int rc = rdgpu_fill_graph_solve_dev(ndev, w, topology, d_keys, d_edges, d_counts, cap, d_levels, nullptr);
      if (rc) throw Error(rc, rdgpu_last_error());
      RD_HIP(hipStreamSynchronize(nullptr));
      RD_HIP(hipMemcpy(levels.data(), d_levels, levels.size() * 4, hipMemcpyDeviceToHost));
    } else {
      std::vector<uint64_t> offs(ndev + 1, 0);
      std::vector<uint32_t> flat;
      for (int s = 0; s < ndev; s++) {
        offs[s + 1] = offs[s] + edges[s].size() / 3;
        flat.insert(flat.end(), edges[s].begin(), edges[s].end());
      }
      graph_solve(ndev, w, topology, keys.data(), flat.data(), offs.data(), levels.data());
    }
    // 3. every device: raise its blocks, copy them back, and only then report success
    per_device(devices, ndev, [&](int, const std::vector<int> &mine) {
      for (int s : mine) {
        rdgpu_fill_shard *p = sh[s];
        sh[s] = nullptr;
        int rc;
        if (staged) {
          rc = rdgpu_fill_shard_finish(p, &levels[(size_t)s * per]);   // synchronises the block's stream
        } else {
          RD_HIP(hipStreamWaitEvent(st[s], ev_levels, 0));              // the levels have arrived on this device
          rc = rdgpu_fill_shard_finish_dev(p, d_lev[s]);                // synchronises the block's stream
        }
        if (rc) throw Error(rc, rdgpu_last_error());
        RD_HIP(hipMemcpyAsync(dem + (size_t)r0[s] * w, blk[s], (size_t)(r0[s + 1] - r0[s]) * w * sizeof(T), hipMemcpyDeviceToHost, st[s]));
      }
      for (int s : mine) {
        RD_HIP(hipStreamSynchronize(st[s]));   // a failed copy must not return RDGPU_OK with a partly updated DEM
        RD_HIP(hipStreamDestroy(st[s]));
        st[s] = nullptr;
        if (ev[s]) { RD_HIP(hipEventDestroy(ev[s])); ev[s] = nullptr; }
      }
    });
    if (s0) {
      DeviceGuard g(devices[0]);
      RD_HIP(hipStreamSynchronize(s0));
      RD_HIP(hipStreamDestroy(s0));
      s0 = nullptr;
      RD_HIP(hipEventDestroy(ev_levels));
      ev_levels = nullptr;
    }
  } catch (...) {
    cleanup();
    (void)hipSetDevice(home);
    throw;
  }
  (void)hipSetDevice(home);
}

}  // namespace rdgpu

using namespace rdgpu;

extern "C" int rdgpu_fill_graph_solve(int nshards, int width, int topology, const uint32_t *keys, const uint32_t *edges,
                                      const uint64_t *edge_offsets, uint32_t *levels) {
  return guarded([&] { graph_solve(nshards, width, topology, keys, edges, edge_offsets, levels); });
}

#define RD_SHARDED_API(SUF, T)                                                                              \
  extern "C" int rdgpu_fill_shard_begin_##SUF(T *, int, int, int, int, int, void *, rdgpu_fill_shard **);   \
  extern "C" int rdgpu_fill_sharded_##SUF(T *dem, int w, int h, int topology, int nshards) {                \
    return guarded([&] {                                                                                    \
      fill_sharded_host<T>(dem, w, h, topology, nshards,                                                    \
                           [](T *d, int w_, int h_, int t, int ot, int ob, rdgpu_fill_shard **o) {          \
                             return rdgpu_fill_shard_begin_##SUF(d, w_, h_, t, ot, ob, nullptr, o);         \
                           });                                                                              \
    });                                                                                                     \
  }                                                                                                         \
  extern "C" int rdgpu_fill_multi_##SUF(T *dem, int w, int h, int topology, const int *devices, int ndev) { \
    return unlocked([&] {                                                                                   \
      fill_multi_host<T>(dem, w, h, topology, devices, ndev,                                                \
                         [](T *d, int w_, int h_, int t, int ot, int ob, hipStream_t st, rdgpu_fill_shard **o) { \
                           return rdgpu_fill_shard_begin_##SUF(d, w_, h_, t, ot, ob, (void *)st, o);        \
                         });                                                                                \
    });                                                                                                     \
  }
RD_SHARDED_API(u8, uint8_t)
RD_SHARDED_API(i16, int16_t)
RD_SHARDED_API(u16, uint16_t)
RD_SHARDED_API(i32, int32_t)
RD_SHARDED_API(u32, uint32_t)
RD_SHARDED_API(f32, float)
RD_SHARDED_API(i8, int8_t)
