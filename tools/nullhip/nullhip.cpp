// nullhip.cpp -- a do-nothing stand-in for the HIP runtime entry points libgtn_amd.so uses,
// LD_PRELOADed by tools/nullhip/host_step to time the HOST side of a training step in a
// container without a GPU (allocation, graph construction, uploads, launch bookkeeping,
// teardown).  Kernels are not executed, so every value the "device" would produce is garbage:
// this is a DIAGNOSTIC for host overhead only -- never a fallback, never loaded by the package,
// the tests, smoke() or bench.py.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
thread_local dim3 t_grid, t_block;
thread_local size_t t_shmem;
thread_local hipStream_t t_stream;
int g_dummy;
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "nullhip"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
// "device" memory = what hipMalloc handed out (so that the engine's device-pointer detection sees it as such)
static std::mutex g_range_mu;
static std::map<uintptr_t, size_t> g_ranges;
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_range_mu);
  auto it = g_ranges.upper_bound(reinterpret_cast<uintptr_t>(p));
  if (it == g_ranges.begin()) return hipErrorInvalidValue;
  --it;
  if (reinterpret_cast<uintptr_t>(p) >= it->first + it->second) return hipErrorInvalidValue;
  std::memset(a, 0, sizeof(*a));
  a->type = hipMemoryTypeDevice;
  return hipSuccess;
}
// NULLHIP_ZERO=1: zero-filled allocations and full-size copies -- deterministic (all-zero) "device"
// results, for comparing the HOST behaviour of two builds on the same test program
static const bool g_zero = std::getenv("NULLHIP_ZERO") != nullptr;
hipError_t hipMalloc(void** p, size_t n) {
  if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
  if (g_zero) std::memset(*p, 0, n ? n : 256);
  {
    std::lock_guard<std::mutex> lk(g_range_mu);
    g_ranges[reinterpret_cast<uintptr_t>(*p)] = n ? n : 256;
  }
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
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  // big tensor copies are the GPU's job; the host only pays the call
  if (g_zero || n <= (1u << 20)) std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  if (g_zero || n <= (1u << 20)) std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(&g_dummy); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // the "GPU" is never busy
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(&g_dummy); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(&g_dummy); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) {
  t_grid = g; t_block = b; t_shmem = sh; t_stream = st;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
  *g = t_grid; *b = t_block; *sh = t_shmem; *st = t_stream;
  return hipSuccess;
}
}
