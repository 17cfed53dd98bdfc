// Latency / throughput micro-benchmarks that shaped band.hip (diagnostic tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lat.hip -o tools/ubench/lat && tools/ubench/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 2048
__device__ __forceinline__ float dpp(float old, float x, int) { return x; }

template <int MODE>
__global__ void k(float* out, long long* cyc, float seed) {
  float x = seed + threadIdx.x, y = seed * 0.5f + threadIdx.x, z = 1.0f;
  __shared__ float sh[512];
  sh[threadIdx.x] = x;
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N_IT; ++i) {
    if (MODE == 0) {  // dependent v_add
      x = x + y;
    } else if (MODE == 1) {  // dependent wave_shr:1
      x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, false)) + y;
    } else if (MODE == 2) {  // dependent row_shr:1
      x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, false)) + y;
    } else if (MODE == 3) {  // dependent exp2
      x = __builtin_amdgcn_exp2f(x) * 0.25f;
    } else if (MODE == 4) {  // dependent log2
      x = __builtin_amdgcn_logf(x + 3.0f);
    } else if (MODE == 5) {  // s_barrier only
      asm volatile("s_barrier" ::: "memory");
      x += y;
    } else if (MODE == 6) {  // lds write + barrier + read (neighbour exchange)
      sh[threadIdx.x] = x;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      x = sh[(threadIdx.x + 255) & 255] + y;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (MODE == 7) {  // ds_bpermute shift
      x = __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x + 63) & 63) << 2, __float_as_int(x))) + y;
    } else if (MODE == 8) {  // 8 independent v_add (issue rate)
      x += y; z += y; y += 1.0f; x += z; z += x; x += 2.0f; z += 3.0f; y += z;
    } else if (MODE == 9) {  // wave_shr via row_bcast + row_shr (2 row-level DPP ops)
      int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xe, 0x1, false);
      t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x111, 0xf, 0xf, false);
      x = __int_as_float(t) + y;
    } else if (MODE == 10) {  // dependent max3/med3/min3 + sub
      float a = fmaxf(fmaxf(x, y), z), b = __builtin_amdgcn_fmed3f(x, y, z);
      x = b - a + x;
    } else if (MODE == 11) {  // lds read dependent chain (address from data)
      x = sh[(__float_as_int(x) >> 2) & 255] + 1.0f;
    } else if (MODE >= 20 && MODE <= 25) {
      // one time step of the banded forward recursion (band.hip), 8 per iteration, dependent through x
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float p1, p2;
        if (MODE == 21) {
          p1 = x + 1.0f;
          p2 = p1 + 1.0f;
        } else if (MODE == 24) {  // row-level dpp only
          p1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(x), 0x111, 0xf, 0xf, false));
          p2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(p1), 0x111, 0xf, 0xf, false));
        } else {
          p1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(x), 0x138, 0xf, 0xf, false));
          p2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(p1), 0x138, 0xf, 0xf, false));
        }
        const float x2 = p2 + z;
        const float mx = fmaxf(fmaxf(x, p1), x2), md = __builtin_amdgcn_fmed3f(x, p1, x2), mn = fminf(fminf(x, p1), x2);
        float r;
        if (MODE == 22) r = mx + (1.0f + (md - mx) + (mn - mx));
        else if (MODE == 25) r = mx + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(md - mx));  // one exp only
        else r = mx + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(md - mx) + __builtin_amdgcn_exp2f(mn - mx));
        x = r + y;
        if (MODE == 23) {  // a second, independent node in the same lane (NPL = 2)
          const float q1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(z), 0x138, 0xf, 0xf, false));
          const float m2 = fmaxf(fmaxf(z, q1), p1), d2 = __builtin_amdgcn_fmed3f(z, q1, p1), n2 = fminf(fminf(z, q1), p1);
          z = m2 + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d2 - m2) + __builtin_amdgcn_exp2f(n2 - m2)) + y;
        }
      }
    } else if (MODE == 12) {  // lds atomic add (no return), distinct addresses
      asm volatile("ds_add_f32 %0, %1" ::"v"((unsigned)(threadIdx.x * 4)), "v"(y) : "memory");
      x += y;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + sh[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads) {
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  k<MODE><<<blocks, threads>>>(out, cyc, 1.0f);
  k<MODE><<<blocks, threads>>>(out, cyc, 1.0f);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += double(v);
  printf("%-44s blocks %4d x %3d : %7.1f clock64 ticks / iteration\n", name, blocks, threads, s / blocks / N_IT);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int blocks = cfg == 0 ? 1 : 512, threads = 256;
    run<0>("dependent v_add_f32", blocks, threads);
    run<8>("8 independent-ish v_add_f32", blocks, threads);
    run<1>("dependent dpp wave_shr:1 + add", blocks, threads);
    run<2>("dependent dpp row_shr:1 + add", blocks, threads);
    run<9>("wave shift by row_bcast:15 + row_shr:1 + add", blocks, threads);
    run<7>("ds_bpermute shift + add", blocks, threads);
    run<3>("dependent v_exp_f32 + mul", blocks, threads);
    run<4>("dependent add + v_log_f32", blocks, threads);
    run<10>("max3 + med3 + sub + add chain", blocks, threads);
    run<5>("s_barrier + add (4 waves)", blocks, threads);
    run<6>("lds write, barrier, read, barrier", blocks, threads);
    run<11>("dependent lds read + add", blocks, threads);
    run<12>("ds_add_f32 + add", blocks, threads);
    run<20>("8 x banded step (2 wave_shr, lse3)", blocks, threads);
    run<21>("8 x banded step, no dpp", blocks, threads);
    run<24>("8 x banded step, row_shr instead of wave_shr", blocks, threads);
    run<22>("8 x banded step, no exp/log", blocks, threads);
    run<25>("8 x banded step, one exp", blocks, threads);
    run<23>("8 x banded step, 2 nodes per lane", blocks, threads);
  }
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wclk = 0;
  hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, wall clock rate %d kHz (clock64 = s_memtime)\n", clk, wclk);
  return 0;
}
