// common.hpp -- shared host/device plumbing of librdgpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
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

// Every C-ABI entry point runs under ONE process-wide lock: the grow-only workspace, the statistics and the
// profiler are process globals, so two host threads (e.g. two Python threads calling the `_richdem` module, which
// releases the GIL around every call) must not be inside the library at the same time.  Calls from different
// threads are serialised here; when the calling thread changes, the device is synchronised first so that the
// previous caller's asynchronous `_dev` work has left the shared scratch buffers.
std::recursive_mutex &api_mutex();
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
  // Frees every buffer.  Refused (Error) while a shard handle whose tables live in the workspace is alive: the handle
  // would keep dangling device pointers (rdgpu_*_shard_begin pins, _finish / _free unpins).
  void release();
  void pin() { pins_++; }
  void unpin() { if (pins_ > 0) pins_--; }

private:
  struct Slot {
    void *p = nullptr;
    size_t cap = 0;
  };
  std::map<std::string, Slot> slots_;
  uint32_t *host_words_ = nullptr;
  int pins_ = 0;
};

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
  };
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> pool_;
  hipEvent_t take();
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

}  // namespace rdgpu
