// nullhip.cpp -- a do-nothing stand-in for the HIP runtime entry points libgtn_amd.so uses,
// LD_PRELOADed by tools/nullhip/host_step to time the HOST side of a training step in a
// container without a GPU (allocation, graph construction, uploads, launch bookkeeping,
// teardown).  Kernels are not executed, so every value the "device" would produce is garbage:
// this is a DIAGNOSTIC for host overhead only -- never a fallback, never loaded by the package,
// the tests, smoke() or bench.py.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

// NULLHIP_DEVICES=8: eight "devices".  What a device IS here: the thread-local current device (hipSetDevice), a stream
// handle that knows its device, every allocation tagged with the device that was current, and counters per device
// (launches, bytes allocated, launches made on a stream while ANOTHER device was current = what the real runtime
// rejects) -- enough to check the engine's per-device contexts (runtime.h) and the sharding of a batch over the GPUs
// of a node (tests/native/multidev_test.cpp) without hardware.  The RCCL entry points comm.cpp looks up are here too,
// doing the collective on host memory.
namespace {
thread_local dim3 t_grid, t_block;
thread_local size_t t_shmem;
thread_local hipStream_t t_stream;
thread_local int t_dev = 0;
int g_dummy;
constexpr int kMaxDev = 16;
const int g_ndev = [] {
  const char* e = std::getenv("NULLHIP_DEVICES");
  const int n = e ? std::atoi(e) : 1;
  return n < 1 ? 1 : (n > kMaxDev ? kMaxDev : n);
}();
struct Stream {
  int dev;
};
std::atomic<long> g_launches[kMaxDev], g_wrong_device[kMaxDev], g_alloc_bytes[kMaxDev];
int dev_of_stream(hipStream_t s) {
  if (!s || s == reinterpret_cast<hipStream_t>(&g_dummy)) return t_dev;
  return reinterpret_cast<Stream*>(s)->dev;
}
}
extern "C" {
// counters for the tests (dlsym'd by name)
long nullhip_launches(int d) { return d >= 0 && d < kMaxDev ? g_launches[d].load() : -1; }
long nullhip_wrong_device_launches(int d) { return d >= 0 && d < kMaxDev ? g_wrong_device[d].load() : -1; }
long nullhip_alloc_bytes(int d) { return d >= 0 && d < kMaxDev ? g_alloc_bytes[d].load() : -1; }
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = g_ndev; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= g_ndev) return hipErrorInvalidDevice;
  t_dev = d;
  return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "nullhip"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
// "device" memory = what hipMalloc handed out (so that the engine's device-pointer detection sees it as such)
static std::mutex g_range_mu;
static std::map<uintptr_t, std::pair<size_t, int>> g_ranges;  // start -> (bytes, device current at hipMalloc)
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_range_mu);
  auto it = g_ranges.upper_bound(reinterpret_cast<uintptr_t>(p));
  if (it == g_ranges.begin()) return hipErrorInvalidValue;
  --it;
  if (reinterpret_cast<uintptr_t>(p) >= it->first + it->second.first) return hipErrorInvalidValue;
  std::memset(a, 0, sizeof(*a));
  a->type = hipMemoryTypeDevice;
  a->device = it->second.second;
  return hipSuccess;
}
// NULLHIP_ZERO=1: zero-filled allocations and full-size copies -- deterministic (all-zero) "device"
// results, for comparing the HOST behaviour of two builds on the same test program
static const bool g_zero = std::getenv("NULLHIP_ZERO") != nullptr;
// NULLHIP_TRACE=1: one line per device operation on stderr (copies with kind and bytes, fills, launches by kernel
// name, synchronisations) -- the dependent chain a small step puts on the stream
static const bool g_trace = std::getenv("NULLHIP_TRACE") != nullptr;
static std::map<const void*, const char*>& kernels() {  // (filled by the fat binaries' constructors: before our statics)
  static auto* m = new std::map<const void*, const char*>();
  return *m;
}
hipError_t hipMalloc(void** p, size_t n) {
  if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
  if (g_zero) std::memset(*p, 0, n ? n : 256);
  {
    std::lock_guard<std::mutex> lk(g_range_mu);
    g_ranges[reinterpret_cast<uintptr_t>(*p)] = {n ? n : 256, t_dev};
  }
  g_alloc_bytes[t_dev].fetch_add(long(n));
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  {
    std::lock_guard<std::mutex> lk(g_range_mu);
    g_ranges.erase(reinterpret_cast<uintptr_t>(p));
  }
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return posix_memalign(p, 256, n ? n : 256) ? hipErrorOutOfMemory : hipSuccess; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
  if (g_trace) std::fprintf(stderr, "nullhip: memcpyAsync kind %d bytes %zu\n", int(k), n);
  // big tensor copies are the GPU's job; the host only pays the call
  if (g_zero || n <= (1u << 20)) std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) {
  if (g_trace) std::fprintf(stderr, "nullhip: memcpy kind %d bytes %zu\n", int(k), n);
  std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  if (g_trace) std::fprintf(stderr, "nullhip: memsetAsync bytes %zu\n", n);
  if (g_zero || n <= (1u << 20)) std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(new Stream{t_dev});  // (never destroyed: the engine keeps its streams)
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) {
  if (g_trace) std::fprintf(stderr, "nullhip: streamSynchronize\n");
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // the "GPU" is never busy
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(&g_dummy); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(&g_dummy); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) {
  if (g_trace) std::fprintf(stderr, "nullhip: streamWaitEvent\n");
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) {
  if (g_trace) std::fprintf(stderr, "nullhip: eventRecord\n");
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) {
  if (g_trace) std::fprintf(stderr, "nullhip: eventSynchronize\n");
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t, hipStream_t st) {
  const int d = dev_of_stream(st);
  // the engine's own small copies and fills (misc.hip: copy_small_kernel, fill_i32_kernel) stand where hipMemcpyAsync /
  // hipMemsetAsync stood, and those this device carries out: so these two kernels are, too
  {
    auto it = kernels().find(f);
    if (it != kernels().end() && args) {
      if (std::strstr(it->second, "copy_small_kernel")) {
        std::memmove(*static_cast<void**>(args[0]), *static_cast<void**>(args[1]), *static_cast<size_t*>(args[2]));
      } else if (std::strstr(it->second, "fill_i32_kernel")) {
        int* p = *static_cast<int**>(args[0]);
        const int v = *static_cast<int*>(args[1]);
        const size_t n = *static_cast<size_t*>(args[2]);
        if (g_zero || n <= (1u << 18)) std::fill(p, p + n, v);
      }
    }
  }
  if (g_trace) {
    auto it = kernels().find(f);
    std::fprintf(stderr, "nullhip: launch %s grid %u block %u\n", it == kernels().end() ? "?" : it->second, g.x * g.y * g.z, b.x);
  }
  g_launches[d].fetch_add(1);
  if (d != t_dev) g_wrong_device[d].fetch_add(1);  // the real runtime: hipErrorInvalidResourceHandle
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host, char*, const char* name, unsigned, void*, void*, void*, void*, int*) {
  kernels()[host] = name;
}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) {
  t_grid = g; t_block = b; t_shmem = sh; t_stream = st;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
  *g = t_grid; *b = t_block; *sh = t_shmem; *st = t_stream;
  return hipSuccess;
}

// ---- the slice of RCCL that gtn_amd/csrc/comm.cpp looks up (dlsym), on host memory: every "device" buffer is
// ordinary memory here, so a collective is a loop.  Group semantics as far as comm.cpp uses them: the calls between
// ncclGroupStart and ncclGroupEnd are collected and carried out at the end.
namespace {
struct NComm {
  int rank, n;
};
struct Pending {
  int kind;  // 0 gather, 1 reduce
  const void* send;
  void* recv;
  size_t count;
  NComm* c;
};
thread_local std::vector<Pending>* t_group = nullptr;
}
int ncclCommInitAll(void** comms, int n, const int*) {
  for (int k = 0; k < n; ++k) comms[k] = new NComm{k, n};
  return 0;
}
int ncclCommDestroy(void* c) {
  delete static_cast<NComm*>(c);
  return 0;
}
int ncclGroupStart() {
  if (!t_group) t_group = new std::vector<Pending>();
  t_group->clear();
  return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int, void* comm, hipStream_t) {
  if (!t_group) return 1;
  t_group->push_back({0, send, recv, count, static_cast<NComm*>(comm)});
  return 0;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int, int, void* comm, hipStream_t) {
  if (!t_group) return 1;
  t_group->push_back({1, send, recv, count, static_cast<NComm*>(comm)});
  return 0;
}
int ncclGroupEnd() {
  if (!t_group) return 1;
  std::vector<Pending>& g = *t_group;
  if (g.empty()) return 0;
  const size_t n = g.size(), count = g[0].count;
  if (g[0].kind == 0) {
    for (size_t k = 0; k < n; ++k)  // every receiver gets every sender's block, at the sender's rank
      for (size_t j = 0; j < n; ++j)
        if (static_cast<char*>(g[k].recv) + 4 * count * size_t(g[j].c->rank) != g[j].send)
          std::memmove(static_cast<char*>(g[k].recv) + 4 * count * size_t(g[j].c->rank), g[j].send, 4 * count);
  } else {
    std::vector<float> sum(count, 0.0f);
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < count; ++i) sum[i] += static_cast<const float*>(g[j].send)[i];
    for (size_t k = 0; k < n; ++k) std::memcpy(g[k].recv, sum.data(), 4 * count);
  }
  g.clear();
  return 0;
}
const char* ncclGetErrorString(int) { return "nullhip rccl"; }
}
