// runtime.hip -- error state, device workspace, HIP-event profiler, runtime C-ABI entry points.
#include "common.hpp"

#include <algorithm>
#include <thread>

namespace rdgpu {

static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }

std::recursive_mutex &api_mutex(int device) {
  static std::mutex mu;
  static std::map<int, std::recursive_mutex *> locks;   // (never freed: a lock may be held at exit)
  std::lock_guard<std::mutex> g(mu);
  std::recursive_mutex *&m = locks[device];
  if (!m) m = new std::recursive_mutex();
  return *m;
}

std::recursive_mutex &api_mutex() {
  int dev = 0;
  (void)hipGetDevice(&dev);   // (no device at all: everything shares lock 0 and fails later with a proper error)
  return api_mutex(dev);
}

void api_enter() {
  // per device: which host thread used it last (held under that device's lock)
  static std::mutex mu;
  static std::map<int, std::thread::id> last;
  static thread_local int depth_guard = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::thread::id me = std::this_thread::get_id();
  bool sync = false;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = last.find(dev);
    sync = it != last.end() && it->second != me && depth_guard == 0;
    last[dev] = me;
  }
  if (sync) {
    // another host thread used this device last: its asynchronous work may still read the shared scratch
    depth_guard = 1;
    (void)hipDeviceSynchronize();
    depth_guard = 0;
  }
}

// ---- Workspace ----------------------------------------------------------------------------
Workspace &Workspace::get() {
  static Workspace w;
  return w;
}

void *Workspace::buf(const char *name, size_t bytes) {
  // slots are per device: after rdgpu_set_device(1) (or under a torch.cuda.device context) a call must not launch on
  // device 1 against scratch that lives in device 0's HBM
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  Slot *sp;
  {
    std::lock_guard<std::mutex> g(mu_);
    sp = &slots_[std::to_string(dev) + ":" + name];   // (map nodes are stable; a slot is only used under its device's API lock)
  }
  Slot &s = *sp;
  if (bytes > s.cap) {
    if (s.p) RD_HIP(hipFree(s.p));
    s.p = nullptr;
    s.cap = 0;
    // grow with a little slack so slightly different basin counts between calls do not realloc
    size_t want = bytes + bytes / 16 + 256;
    RD_HIP(hipMalloc(&s.p, want));
    s.cap = want;
  }
  return s.p;
}

uint32_t *Workspace::host_words() {
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu_);
  uint32_t *&p = host_words_[dev];
  if (!p) RD_HIP(hipHostMalloc((void **)&p, 256 * sizeof(uint32_t), hipHostMallocDefault));
  return p;
}

Workspace::SideLane &Workspace::side_lane(int k) {
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu_);
  SideLane &l = side_[std::make_pair(dev, k)];
  if (!l.stream) {
    RD_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    RD_HIP(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
    RD_HIP(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
  }
  return l;
}

void Workspace::release() {
  std::lock_guard<std::mutex> g(mu_);
  if (pins_ > 0)
    throw Error(RDGPU_ERR_ARG, "rdgpu_release_workspace: " + std::to_string(pins_) +
                                   " shard handle(s) still hold workspace buffers (finish or free them first)");
  int home = 0, ndev = 0;
  (void)hipGetDevice(&home);
  (void)hipGetDeviceCount(&ndev);
  for (int d = 0; d < ndev; d++) {   // buffers of every device this process touched
    (void)hipSetDevice(d);
    (void)hipDeviceSynchronize();
  }
  (void)hipSetDevice(home);
  for (auto &kv : slots_)
    if (kv.second.p) (void)hipFree(kv.second.p);
  slots_.clear();
  for (auto &kv : host_words_)
    if (kv.second) (void)hipHostFree(kv.second);
  host_words_.clear();
}

// ---- Profiler -----------------------------------------------------------------------------
Profiler &Profiler::get() {
  static Profiler p;
  return p;
}

hipEvent_t Profiler::take(int device) {
  std::vector<hipEvent_t> &pool = pool_[device];
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e;
  RD_HIP(hipEventCreate(&e));
  return e;
}

void Profiler::begin(const char *name, hipStream_t s) {
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu_);
  Pending p;
  p.name = name;
  p.device = dev;
  p.a = take(dev);
  p.b = take(dev);
  RD_HIP(hipEventRecord(p.a, s));
  pending_.push_back(p);
}

void Profiler::end(hipStream_t s) {
  int dev = 0;
  RD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu_);
  for (size_t i = pending_.size(); i-- > 0;)   // the calling thread's open entry: the latest one of its device
    if (pending_[i].device == dev) { RD_HIP(hipEventRecord(pending_[i].b, s)); return; }
}

void Profiler::collect() {
  std::lock_guard<std::mutex> g(mu_);
  int home = 0;
  (void)hipGetDevice(&home);
  for (auto &p : pending_) {
    (void)hipSetDevice(p.device);
    RD_HIP(hipEventSynchronize(p.b));
    float ms = 0;
    RD_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    Tot &t = totals[p.name];
    t.ms += ms;
    t.n += 1;
    pool_[p.device].push_back(p.a);
    pool_[p.device].push_back(p.b);
  }
  (void)hipSetDevice(home);
  pending_.clear();
}

void Profiler::reset() {
  collect();
  std::lock_guard<std::mutex> g(mu_);
  totals.clear();
}

std::vector<std::string> Profiler::names() const {
  std::vector<std::string> v;
  for (auto &kv : totals) v.push_back(kv.first);
  return v;
}

}  // namespace rdgpu

using namespace rdgpu;

extern "C" {

const char *rdgpu_last_error(void) { return g_last_error.c_str(); }
const char *rdgpu_version(void) { return "rdgpu 0.1 (gfx950)"; }

int rdgpu_device_count(int *count) {
  return guarded([&] {
    if (!count) throw Error(RDGPU_ERR_ARG, "rdgpu_device_count: null pointer");
    RD_HIP(hipGetDeviceCount(count));
  });
}

int rdgpu_set_device(int id) {
  return guarded([&] { RD_HIP(hipSetDevice(id)); });
}

int rdgpu_release_workspace(void) {
  return guarded([&] { Workspace::get().release(); });
}

int rdgpu_profile_enable(int on) {
  Profiler::get().enabled = on != 0;
  return RDGPU_OK;
}
int rdgpu_profile_collect(void) {
  return guarded([&] { Profiler::get().collect(); });
}
int rdgpu_profile_reset(void) {
  return guarded([&] { Profiler::get().reset(); });
}
const char *rdgpu_profile_name(int index) {
  static thread_local std::string s;
  auto v = Profiler::get().names();
  if (index < 0 || (size_t)index >= v.size()) return nullptr;
  s = v[index];
  return s.c_str();
}
int rdgpu_profile_get(const char *kernel, double *total_ms, uint64_t *launches) {
  auto &t = Profiler::get().totals;
  auto it = t.find(kernel ? kernel : "");
  if (it == t.end()) {
    if (total_ms) *total_ms = 0;
    if (launches) *launches = 0;
    return RDGPU_ERR_ARG;
  }
  if (total_ms) *total_ms = it->second.ms;
  if (launches) *launches = it->second.n;
  return RDGPU_OK;
}

}  // extern "C"
