// What ONE SIMD sustains when several waves issue to it (diagnostic tool, not part of the library): the cost table behind
// DESIGN section 13.3's "the backward band sweep is bound by its vector instruction count".
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_tput.hip -o tools/ubench/valu_tput && tools/ubench/valu_tput
// One workgroup of 4 W waves on one CU (W per SIMD), every wave runs the same loop of 8 independent chains of one
// operation; reported: SIMD cycles per wave instruction = elapsed / (iterations * 8 * W).
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_IT 2048

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
  float f[16];
  double d[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = 1.0f + seed * (i + 1) + threadIdx.x * 1e-6f;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = 1.0 + seed * (i + 1);
  const float fm = 1.0f - 1e-7f * seed, fa = 1e-9f * seed;
  __shared__ float lds[4096];
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 1024] = seed;
  const int sacc = 0;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
      if (MODE == 1) {  // packed: two floats per lane and instruction
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x = {f[2 * i], f[2 * i + 1]}, m = {fm, fm}, a = {fa, fa};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(a));
        f[2 * i] = x.x, f[2 * i + 1] = x.y;
      }
      if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(double(fa)));
      if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
      if (MODE == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
      if (MODE == 6) {  // a vector instruction and a scalar one, alternating
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
        asm volatile("s_add_i32 s20, s20, 1" ::: "s20");
      }
      if (MODE == 7) {  // a vector instruction and an LDS read, alternating
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
        float v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(int(threadIdx.x * 4)));
        f[8 + i] = v;
      }
      if (MODE == 8) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fm));
      if (MODE == 10) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(f[i]) : "v"(fm));
      if (MODE == 11) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
    }
    if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = clock64();
  float r = float(sacc);
#pragma unroll
  for (int i = 0; i < 16; ++i) r += f[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += float(d[i]);
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
double run(int wps) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4 * 1024);
  hipMalloc(&cyc, 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256 * wps), 0, 0, out, cyc, 0.0f);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  hipFree(out);
  hipFree(cyc);
  return double(c) / (double(N_IT) * 8 * wps);
}

template <int MODE>
void row(const char* name) {
  printf("  %-44s %6.2f %6.2f %6.2f %6.2f\n", name, run<MODE>(1), run<MODE>(2), run<MODE>(3), run<MODE>(4));
}

int main() {
  printf("SIMD cycles per wave instruction at 1 / 2 / 3 / 4 waves per SIMD\n");
  row<0>("v_fma_f32");
  row<1>("v_pk_fma_f32 (two floats per lane)");
  row<2>("v_exp_f32");
  row<8>("v_log_f32");
  row<3>("v_add_f64");
  row<11>("v_cvt_f32_f64");
  row<4>("v_mov_b32_dpp wave_shr:1");
  row<5>("v_max3_f32");
  row<9>("v_cndmask_b32");
  row<10>("v_lshl_add_u32");
  row<6>("v_fma_f32 + s_add_i32 (per pair)");
  row<7>("v_fma_f32 + ds_read_b32 (per pair)");
  return 0;
}
