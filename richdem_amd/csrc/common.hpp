// common.hpp -- shared host/device plumbing of librdgpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdgpu.h"

namespace rdgpu {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &m);

#define RD_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      throw ::rdgpu::Error(RDGPU_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e_) + \
                                              " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

// Every C-ABI entry point runs under the lock of the device that is current when it is entered: the grow-only
// workspace (scratch named per device) must not be used by two host threads at once (e.g. two Python threads calling the
// `_richdem` module, which releases the GIL around every call).  Calls on ONE device are serialised here, and when the
// calling thread changes that device is synchronised first, so that the previous caller's asynchronous `_dev` work has
// left the shared scratch buffers.  Calls on DIFFERENT devices hold different locks and run side by side: one host thread
// per device is how the single-process multi-device entry points (rdgpu_*_multi_*) drive a node.  What is shared across
// devices has locks of its own (the workspace's slot map, the profiler) or is per thread (statistics, last error).
std::recursive_mutex &api_mutex();            // of the current device
std::recursive_mutex &api_mutex(int device);
void api_enter();   // call with api_mutex() held

// Runs fn(), maps exceptions to the C-ABI return code convention.
template <class F>
int guarded(F &&fn) {
  std::lock_guard<std::recursive_mutex> lock(api_mutex());
  try {
    api_enter();
    fn();
    return RDGPU_OK;
  } catch (const Error &e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return RDGPU_ERR_HIP;
  }
}

// The same mapping of exceptions to return codes WITHOUT a device lock: for entry points that only orchestrate
// per-device workers, each of which takes its own device's lock for its phase (rdgpu_*_multi_*).  The blocks of such a
// call are staged in workspace buffers named by shard index and the device locks are released between the phases, so two
// orchestrators at once would overwrite each other's blocks: ONE of them runs at a time (r04).  No caller of
// these entries holds a device lock (the plain entries route here BEFORE they lock), so the order orchestrator -> device
// cannot deadlock.
inline std::mutex &orchestrator_mutex() {
  static std::mutex m;
  return m;
}
inline bool &orchestrating() {   // this thread is inside a multi-device entry: another one from here would wait for itself
  static thread_local bool inside = false;
  return inside;
}
template <class F>
int unlocked(F &&fn) {
  if (orchestrating()) {   // (the lock below is not recursive -- say so instead of deadlocking)
    set_last_error("rdgpu_*_multi_*: a multi-device entry was called from inside another one on the same thread");
    return RDGPU_ERR_ARG;
  }
  std::lock_guard<std::mutex> one_orchestrator(orchestrator_mutex());
  struct Flag { Flag() { orchestrating() = true; } ~Flag() { orchestrating() = false; } } flag;
  try {
    fn();
    return RDGPU_OK;
  } catch (const Error &e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return RDGPU_ERR_HIP;
  }
}

// ------------------------------------------------------------------------------------------
// grow-only device workspace, cached across calls (bench loops must not hipMalloc per step)
// ------------------------------------------------------------------------------------------
class Workspace {
public:
  static Workspace &get();
  // Returns a buffer of at least `bytes` on the CURRENT device, identified by (device, name); contents are undefined.
  void *buf(const char *name, size_t bytes);
  template <class T>
  T *buf(const char *name, size_t count) {
    return static_cast<T *>(buf(name, count * sizeof(T)));
  }
  // pinned host scalar slots for small read-backs
  uint32_t *host_words();
  // A second stream of the current device (non-blocking, created on first use, kept for the life of the process) with a
  // fork and a join event: work that does not depend on what the caller's stream is doing runs beside it --
  // record fork on the caller's stream, let side wait for it, enqueue, record join on side, let the caller's stream wait.
  struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
  };
  SideLane &side_lane(int k = 0);   // k = 0, 1: two lanes per device
  // Frees every buffer.  Refused (Error) while a shard handle whose tables live in the workspace is alive: the handle
  // would keep dangling device pointers (rdgpu_*_shard_begin pins, _finish / _free unpins).
  void release();
  void pin() { std::lock_guard<std::mutex> g(mu_); pins_++; }
  void unpin() { std::lock_guard<std::mutex> g(mu_); if (pins_ > 0) pins_--; }

private:
  struct Slot {
    void *p = nullptr;
    size_t cap = 0;
  };
  std::mutex mu_;   // the maps below are shared by the threads of different devices
  std::map<std::string, Slot> slots_;
  std::map<int, uint32_t *> host_words_;   // per device: a read-back of one device must not land in another's words
  std::map<std::pair<int, int>, SideLane> side_;
  int pins_ = 0;
};

// ------------------------------------------------------------------------------------------
// one host thread per device (the single-process multi-device entry points)
// ------------------------------------------------------------------------------------------
struct DeviceGuard {   // the worker's device: current + locked for the scope
  int prev = 0;
  std::unique_lock<std::recursive_mutex> lock;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    RD_HIP(hipSetDevice(dev));
    lock = std::unique_lock<std::recursive_mutex>(api_mutex(dev));
    api_enter();
  }
  ~DeviceGuard() {
    lock.unlock();
    (void)hipSetDevice(prev);
  }
};

// direct copies between two devices of this process (xGMI) where the platform allows them; without it a peer copy is staged
// by the runtime
inline void enable_peer_access(int a, int b) {
  if (a == b) return;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); return; }
  DeviceGuard g(a);
  (void)hipDeviceEnablePeerAccess(b, 0);   // (hipErrorPeerAccessAlreadyEnabled is fine)
  (void)hipGetLastError();
}

// runs job(device, its shard indices) for every distinct device, each in a thread of its own; rethrows the first failure
template <class Job>
inline void per_device(const int *devices, int ndev, Job job) {
  std::vector<int> order;                       // distinct devices in order of first appearance
  std::map<int, std::vector<int>> shards;
  for (int s = 0; s < ndev; s++) {
    if (!shards.count(devices[s])) order.push_back(devices[s]);
    shards[devices[s]].push_back(s);
  }
  if (order.size() == 1) {   // one device (listed once or several times): no worker thread, the caller does the work
    DeviceGuard g(order[0]);
    job(order[0], shards[order[0]]);
    return;
  }
  std::vector<std::exception_ptr> err(order.size());
  std::vector<const std::vector<int> *> mine(order.size());   // (looked up here: the workers must not touch the map)
  for (size_t k = 0; k < order.size(); k++) mine[k] = &shards.at(order[k]);
  std::vector<std::thread> th;
  for (size_t k = 0; k < order.size(); k++)
    th.emplace_back([&, k] {
      try {
        DeviceGuard g(order[k]);
        job(order[k], *mine[k]);
      } catch (...) {
        err[k] = std::current_exception();
      }
    });
  for (auto &t : th) t.join();
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
}

// RDGPU_DEVICES=0,1,2,...: the host-pointer entry points (what rdgpu::FillDepressions(Array2D&), rdgpu::d8_flow_accum and
// the apps call) spread their row blocks over these devices (rdgpu_*_multi_*); unset or one id: the current device.
// A multi-device driver that falls back to one device switches the routing off for its own thread meanwhile.
inline bool &multi_route_off() {
  static thread_local bool off = false;
  return off;
}
inline std::vector<int> env_devices() {
  std::vector<int> v;
  const char *e = getenv("RDGPU_DEVICES");
  if (!e || multi_route_off()) return v;
  for (const char *p = e; *p;) {
    char *end = nullptr;
    const long id = strtol(p, &end, 10);
    if (end == p) break;
    v.push_back((int)id);
    p = *end == ',' ? end + 1 : end;
    if (*end != ',') break;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// per-kernel timing with HIP events on the launch stream
// ------------------------------------------------------------------------------------------
class Profiler {
public:
  static Profiler &get();
  bool enabled = false;
  void begin(const char *name, hipStream_t s);
  void end(hipStream_t s);
  void collect();
  void reset();
  struct Tot {
    double ms = 0;
    uint64_t n = 0;
  };
  std::map<std::string, Tot> totals;
  std::vector<std::string> names() const;

private:
  struct Pending {
    std::string name;
    hipEvent_t a, b;
    int device;
  };
  std::mutex mu_;
  std::vector<Pending> pending_;
  std::map<int, std::vector<hipEvent_t>> pool_;   // events belong to the device they were created on
  hipEvent_t take(int device);
};

// Launch a kernel, bracketed by events when profiling is on.
#define RD_LAUNCH(name, kern, grid, block, shmem, stream, ...)                                    \
  do {                                                                                            \
    ::rdgpu::Profiler &pf_ = ::rdgpu::Profiler::get();                                           \
    if (pf_.enabled) pf_.begin(name, stream);                                                     \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                            \
    if (pf_.enabled) pf_.end(stream);                                                             \
    RD_HIP(hipGetLastError());                                                                    \
  } while (0)

// ------------------------------------------------------------------------------------------
// order-preserving 32-bit keys: a < b  <=>  key(a) < key(b) (unsigned compare)
// Fill / directions only ever compare and copy elevations, so working on
// keys is exact for every dtype.  Floats: sign-flip trick; -0.0f is canonicalised to +0.0f so
// that it compares equal, as it does in the reference.  NaN elevations are unsupported input
// (the reference's Zhou2016 heap has no NaN ordering either).
// ------------------------------------------------------------------------------------------
template <class T>
struct Key32;

template <>
struct Key32<float> {
  __host__ __device__ static inline uint32_t to(float v) {
    uint32_t b = __builtin_bit_cast(uint32_t, v);
    if (b == 0x80000000u) b = 0;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
  __host__ __device__ static inline float from(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
  }
};
template <>
struct Key32<int32_t> {
  __host__ __device__ static inline uint32_t to(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
  __host__ __device__ static inline int32_t from(uint32_t k) { return (int32_t)(k ^ 0x80000000u); }
};
template <>
struct Key32<uint32_t> {
  __host__ __device__ static inline uint32_t to(uint32_t v) { return v; }
  __host__ __device__ static inline uint32_t from(uint32_t k) { return k; }
};
template <>
struct Key32<int16_t> {
  __host__ __device__ static inline uint32_t to(int16_t v) { return (uint32_t)((int32_t)v + 32768); }
  __host__ __device__ static inline int16_t from(uint32_t k) { return (int16_t)((int32_t)k - 32768); }
};
template <>
struct Key32<uint16_t> {
  __host__ __device__ static inline uint32_t to(uint16_t v) { return v; }
  __host__ __device__ static inline uint16_t from(uint32_t k) { return (uint16_t)k; }
};
template <>
struct Key32<int8_t> {
  __host__ __device__ static inline uint32_t to(int8_t v) { return (uint32_t)((int32_t)v + 128); }
  __host__ __device__ static inline int8_t from(uint32_t k) { return (int8_t)((int32_t)k - 128); }
};
template <>
struct Key32<uint8_t> {
  __host__ __device__ static inline uint32_t to(uint8_t v) { return v; }
  __host__ __device__ static inline uint8_t from(uint32_t k) { return (uint8_t)k; }
};

// XCD-aware block -> tile mapping: the dispatcher places block b on XCD b % 8 (performance
// observation only, MI355X_MICROARCH.md); giving each XCD a contiguous band of tiles keeps halo
// rows, per-basin tables and atomics of neighbouring tiles in ONE XCD's L2.  Any mapping is
// correct; this one is only faster.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t ntiles) {
  const uint32_t per = (ntiles + 7u) / 8u;
  uint32_t t = (b & 7u) * per + (b >> 3);
  return t;  // may be >= ntiles for the ragged tail: caller must bounds-check
}
inline uint32_t xcd_grid(uint32_t ntiles) { return ((ntiles + 7u) / 8u) * 8u; }

// Stage the (TH_ + 2 H_) x (TW_ + 2 H_) window around the tile whose first cell is (x0, y0) into LDS rows of LW_ elements
// (the tile's first column at element H_ of a row) -- for a window that lies INSIDE the raster, which the caller tests with
// window_inside(): four elements per load for the tile's columns (element-aligned: any raster width), the 2 H_ halo
// columns cell by cell, every load of the thread issued before the first store.  The stencil kernels staged cell by cell
// with a division and two clamps per cell (r01-r04c): a quarter of their instructions, and they are bound by instruction
// issue (profiles/r04e_path40k_sq_summary.csv).  Tiles on the raster's border keep that path.
__device__ __forceinline__ bool window_inside(int x0, int y0, int w, int h, int tw, int th, int halo) {
  return x0 >= halo && y0 >= halo && x0 + tw + halo <= w && y0 + th + halo <= h;
}
template <class T, int TW_, int TH_, int H_, int LW_, int NT_>
__device__ __forceinline__ void stage_window_inside(const T *__restrict__ src, int w, int x0, int y0, T *dst) {
  constexpr int ROWS = TH_ + 2 * H_, QPR = TW_ / 4, NQ = ROWS * QPR, QPT = (NQ + NT_ - 1) / NT_;
  constexpr int NHC = ROWS * 2 * H_, HPT = (NHC + NT_ - 1) / NT_;
  static_assert(TW_ % 4 == 0, "four elements per load");
  struct Q4 { T v[4]; };
  const T *const base = src + ((size_t)(y0 - H_) * w + (size_t)(x0 - H_));   // the window's first cell (block-uniform)
  Q4 q[QPT];
  T hv[HPT];
#pragma unroll
  for (int r = 0; r < QPT; r++) {
    const int i = (int)threadIdx.x + r * NT_;
    const int ly = i / QPR, qq = i - ly * QPR;
    if (i < NQ) __builtin_memcpy(&q[r], base + (uint32_t)(ly * w + H_ + 4 * qq), sizeof(Q4));
  }
#pragma unroll
  for (int r = 0; r < HPT; r++) {
    const int i = (int)threadIdx.x + r * NT_;
    const int ly = i / (2 * H_), c = i - ly * (2 * H_);
    if (i < NHC) hv[r] = base[(uint32_t)(ly * w + (c < H_ ? c : TW_ + c))];
  }
#pragma unroll
  for (int r = 0; r < QPT; r++) {
    const int i = (int)threadIdx.x + r * NT_;
    const int ly = i / QPR, qq = i - ly * QPR;
    if (i < NQ) {
#pragma unroll
      for (int e = 0; e < 4; e++) dst[ly * LW_ + H_ + 4 * qq + e] = q[r].v[e];
    }
  }
#pragma unroll
  for (int r = 0; r < HPT; r++) {
    const int i = (int)threadIdx.x + r * NT_;
    const int ly = i / (2 * H_), c = i - ly * (2 * H_);
    if (i < NHC) dst[ly * LW_ + (c < H_ ? c : TW_ + c)] = hv[r];
  }
}

}  // namespace rdgpu
