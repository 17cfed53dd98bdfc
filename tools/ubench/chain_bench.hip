// chain_bench.hip -- what does ONE link of a chain of dependent launches cost on this GPU, and what does a
// grid-wide barrier inside a persistent kernel cost instead?  (Sizing of the per-time-step kernels of
// lazy.hip / maxplus.hip: 1000 dependent launches per sweep.)   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__global__ void k_empty(float* p) {
  if (p == nullptr && threadIdx.x == 12345) p[0] = 1.0f;
}
// every workgroup writes 4 KB (1 MB per launch at 256 workgroups): dirty lines for the boundary to flush
__global__ void k_write(float* p, int it) { p[(size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4 + (it & 3)] = float(it); }
// every workgroup reads `kb` KB of a shared 2.5 MB region written by the previous launch, then writes 4 KB
__global__ void k_readwrite(const float* __restrict__ in, float* out, int kb, int it) {
  const float4* q = reinterpret_cast<const float4*>(in);
  const int n16 = kb * 64;  // float4s
  const int total = 2560 * 64;
  float acc = 0.0f;
  int at = (blockIdx.x * 977 + threadIdx.x) % total;
  for (int i = threadIdx.x; i < n16; i += blockDim.x) {
    const float4 v = q[at];
    acc += v.x + v.y + v.z + v.w;
    at += blockDim.x;
    if (at >= total) at -= total;
  }
  out[size_t(blockIdx.x) * blockDim.x + threadIdx.x] = acc + float(it);
}

// a kernel that is busy for about `us` microseconds (wall clock counter at 100 MHz), then writes 4 KB per workgroup
__global__ void k_busy(float* p, int it, int us) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 100ll * us) __builtin_amdgcn_s_sleep(2);
  p[(size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4 + (it & 3)] = float(it);
}

// persistent: `steps` rounds of (write 4 KB, grid barrier, read what a neighbour wrote)
__global__ void k_persistent(float* buf, unsigned* counter, int steps, float* sink) {
  const unsigned nwg = gridDim.x;
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s) {
    float* plane = buf + size_t(s & 1) * nwg * blockDim.x;
    plane[size_t(blockIdx.x) * blockDim.x + threadIdx.x] = float(s) + acc * 1e-9f;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = unsigned(s + 1) * nwg;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
    const unsigned nb = (blockIdx.x + 37) % nwg;
    acc += __hip_atomic_load(&plane[size_t(nb) * blockDim.x + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sink[size_t(blockIdx.x) * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2000;
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;
  float *a, *b;
  unsigned* ctr;
  CK(hipMalloc(&a, 64 << 20));
  CK(hipMalloc(&b, 64 << 20));
  CK(hipMalloc(&ctr, 4));
  CK(hipMemset(a, 0, 64 << 20));
  CK(hipMemset(b, 0, 64 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) {
    for (int i = 0; i < 50; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us per launch\n", name, ms * 1e3 / n);
  };
  timeit("empty kernel, 256 threads", [&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st, a); });
  timeit("empty kernel, 512 threads", [&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(512), 0, st, a); });
  timeit("write 4 KB per workgroup", [&](int i) { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, st, a, i); });
  for (int kb : {16, 64, 148}) {
    char nm[64];
    snprintf(nm, sizeof nm, "read %d KB per workgroup + write", kb);
    timeit(nm, [&](int i) {
      hipLaunchKernelGGL(k_readwrite, dim3(wgs), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, kb, i);
    });
  }
  // the same chain replayed from a graph
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, st, a, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < n / 100; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us per launch\n", "write 4 KB, chain of 100 as a graph", ms * 1e3 / (n / 100 * 100));
  }
  // does a graph still pay when the kernels are long enough for the host to keep up?  (7 us busy + 4 KB writes)
  {
    timeit("busy 7 us + write, stream launches", [&](int i) { hipLaunchKernelGGL(k_busy, dim3(wgs), dim3(256), 0, st, a, i, 7); });
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_busy, dim3(wgs), dim3(256), 0, st, a, i, 7);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < n / 100; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us per launch\n", "busy 7 us + write, chain of 100 as a graph", ms * 1e3 / (n / 100 * 100));
  }
  // persistent kernel with a grid barrier per step (all workgroups must be resident: cooperative launch)
  for (int th : {256, 512}) {
    CK(hipMemset(ctr, 0, 4));
    int steps = n;
    void* args[] = {&a, &ctr, &steps, &b};
    CK(hipEventRecord(e0, st));
    CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_persistent), dim3(wgs), dim3(th), args, 0, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("persistent, grid barrier per step, %d threads   %8.2f us per step\n", th, ms * 1e3 / n);
  }
  return 0;
}
