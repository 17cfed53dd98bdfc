// global_load_lds_dwordx4 on gfx950: where does lane i's 16 bytes land?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_test.hip -o tools/ubench/lds_dma_test && tools/ubench/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float buf[2048];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = -1.0f;
  __syncthreads();
  // wave w: lane l fetches src[(w * 64 + (63 - l)) * 4 .. +4) -- a permuted gather -- into the wave's KB of LDS
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (wave * 64 + (63 - lane)) * 4),
                                   (__attribute__((address_space(3))) void*)(buf + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = buf[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = float(i);
  float *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, o);
  std::vector<float> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  // expectation if lane l's 16 bytes land at base + 16 l: r[w*256 + 4l + j] == (w*64 + 63 - l)*4 + j
  int bad = 0;
  for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j)
    bad += r[w * 256 + 4 * l + j] != float((w * 64 + 63 - l) * 4 + j);
  printf("lane-major 16-byte landing: %s (%d mismatches); first values %g %g %g %g | %g\n", bad ? "NO" : "yes", bad, r[0], r[1], r[2], r[3], r[4]);
  return 0;
}
