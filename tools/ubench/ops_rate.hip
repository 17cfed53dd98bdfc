// Issue rate of the VALU operations a linear-domain band sweep would be made of, and one time step of the backward
// recursion in its two forms (diagnostic tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ops_rate.hip -o tools/ubench/ops_rate && tools/ubench/ops_rate
// One wave per workgroup, one workgroup: cycles per wave instruction with eight independent chains (issue cost)
// and with one dependent chain (latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 1024

__device__ __forceinline__ float shl1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double shl1d(double x, double fill) {
  const long long b = __double_as_longlong(x), f = __double_as_longlong(fill);
  const int lo = __builtin_amdgcn_update_dpp(int(f), int(b), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(int(f >> 32), int(b >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
  const float s = seed + threadIdx.x * 1e-3f;
  double d[8];
  float f[8];
  int e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d[i] = 1.0 + s * (i + 1);
    f[i] = 1.0f + s * (i + 1);
    e[i] = i - 3 + int(seed);
  }
  const double dm = 1.0 - 1e-9 * seed, da = 1e-12 * seed;
  const float fm = 1.0f - 1e-7f * seed, fa = 1e-9f * seed;
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (MODE >= 100) ? 0 : i;  // >= 100: one dependent chain
      const int m = MODE % 100;
      if (m == 0) f[j] = __builtin_fmaf(f[j], fm, fa);
      else if (m == 1) d[j] = __builtin_fma(d[j], dm, da);
      else if (m == 2) d[j] = d[j] * dm;
      else if (m == 3) d[j] = d[j] + da;
      else if (m == 4) d[j] = __builtin_amdgcn_ldexp(d[j], e[j] & 1);
      else if (m == 5) f[j] = float(d[j]) + f[j], d[j] += 0.0;  // cvt_f32_f64 (+ an add)
      else if (m == 6) d[j] = double(f[j]) * dm, f[j] += fa;      // cvt_f64_f32 (+ a mul, an add)
      else if (m == 7) f[j] = __builtin_amdgcn_exp2f(f[j]) * 0.25f;
      else if (m == 8) f[j] = __builtin_amdgcn_ldexpf(f[j], e[j] & 1);
      else if (m == 9) e[j] += __builtin_amdgcn_frexp_exp(d[j]), d[j] = __builtin_amdgcn_frexp_mant(d[j]) + 1.0;
      else if (m == 10) e[j] += __builtin_amdgcn_frexp_expf(f[j]), f[j] = __builtin_amdgcn_frexp_mantf(f[j]) + 1.0f;
      else if (m == 11) f[j] = shl1(f[j], fa);
      else if (m == 12) d[j] = shl1d(d[j], da);
      else if (m == 13) f[j] = __builtin_amdgcn_logf(f[j] + 3.0f);
      else if (m == 14) e[j] = max(max(e[j], e[(j + 1) & 7]), e[(j + 2) & 7]) - 1;
      else if (m == 15) f[j] = __builtin_amdgcn_fractf(f[j]) + 1.5f;
    }
  }
  long long t1 = clock64();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += f[i] + float(d[i]) + float(e[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- one time step of the backward band recursion (UNIT graph, G's arc gradients wanted), eight per iteration, a
// dependent chain through the steps.  STEP 0: log domain as band_backward_body.inc has it.  STEP 1: linear domain,
// float64 values against a wave-uniform scale (2^x by v_exp_f32 of the fraction + ldexp), arc gradients in float64.
// STEP 2: linear domain, float32 values with a per-lane integer exponent.
template <int STEP>
__global__ void kstep(float* out, long long* cyc, float seed, const float* __restrict__ tab) {
  const int l = threadIdx.x & 63;
  __shared__ float ev_s[64 * 64], al_s[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ev_s[i] = tab[i & 511], al_s[i] = tab[512 + (i & 511)];
  __syncthreads();
  float acc[3] = {0, 0, 0};
  double acc64[3] = {0, 0, 0};
  float post = 0.0f;
  const float wo2 = (l & 1) ? 0.0f : -1e30f;
  const float m2 = (l & 1) ? 1.0f : 0.0f;
  float b = seed * l * 1e-3f;       // log domain
  double bl = 1.0 + seed * l;       // linear, float64
  float bm = 0.5f + seed * l;       // linear, float32 mantissa
  int bx = int(seed);               // ... and exponent
  const float dl = seed;
  const int Di = int(seed);
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < N_IT; ++it) {
    float ev[8], al[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ev[i] = ev_s[((it * 8 + i) & 63) * 64 + l], al[i] = al_s[((it * 8 + i) & 63) * 64 + l];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (STEP == 0) {
        const float q = ev[i] + b;
        const float n1 = shl1(q, dl), n2 = shl1(n1, dl);
        const float y0 = q, y1 = n1, y2 = n2 + wo2;
        const float mx = fmaxf(fmaxf(y0, y1), y2);
        const float f = __builtin_amdgcn_exp2f(al[i] + mx + dl);
        const float e0 = __builtin_amdgcn_exp2f(y0 - mx), e1 = __builtin_amdgcn_exp2f(y1 - mx), e2 = __builtin_amdgcn_exp2f(y2 - mx);
        const float S = e0 + e1 + e2;
        acc[0] += e0 * f, acc[1] += e1 * f, acc[2] += e2 * f;
        b = mx + __builtin_amdgcn_logf(S);
        post += f * S;
      } else if (STEP == 1) {
        const double q = double(ev[i]) * bl;  // (ev would be stored as a float probability)
        const double n1 = shl1d(q, 0.0), n2 = shl1d(n1, 0.0);
        const double t2 = n2 * double(m2);
        const double S = q + n1 + t2;
        bl = S;
        // posterior factor 2^(al + D): fraction by v_exp_f32, integer part by ldexp
        const float x = al[i] + dl;
        const float xf = __builtin_amdgcn_fractf(x);
        const int xi = int(x - xf) + Di;
        const double F = __builtin_amdgcn_ldexp(double(__builtin_amdgcn_exp2f(xf)), xi);
        post += float(S * F);
        acc64[0] = __builtin_fma(q, F, acc64[0]);
        acc64[1] = __builtin_fma(n1, F, acc64[1]);
        acc64[2] = __builtin_fma(t2, F, acc64[2]);
      } else if (STEP == 3) {
        const float q = ev[i] * bm;
        const float n1 = shl1(q, 0.0f), n2 = shl1(n1, 0.0f);
        const float t2 = n2 * m2;
        const float S = q + n1 + t2;
        bm = S;
        const float f = __builtin_amdgcn_exp2f(al[i] + dl);
        post += f * S;
        acc[0] += q * f, acc[1] += n1 * f, acc[2] += t2 * f;
      } else {
        const float qm = ev[i] * bm;
        const int qe = bx;
        const float n1m = shl1(qm, 0.0f), n2m = shl1(n1m, 0.0f);
        const int n1e = __float_as_int(shl1(__int_as_float(qe), __int_as_float(-1000000)));
        const int n2e = __float_as_int(shl1(__int_as_float(n1e), __int_as_float(-1000000))) + ((l & 1) ? 0 : -1000000);
        const int E = max(max(qe, n1e), n2e);
        const float t0 = __builtin_amdgcn_ldexpf(qm, qe - E), t1 = __builtin_amdgcn_ldexpf(n1m, n1e - E),
                    t2 = __builtin_amdgcn_ldexpf(n2m, n2e - E);
        const float S = t0 + t1 + t2;
        bm = __builtin_amdgcn_frexp_mantf(S);
        bx = S == 0.0f ? -1000000 : E + __builtin_amdgcn_frexp_expf(S);
        const float f = __builtin_amdgcn_exp2f(al[i] + dl + float(E + Di));
        post += f * S;
        acc[0] += t0 * f, acc[1] += t1 * f, acc[2] += t2 * f;
      }
    }
    if (STEP == 3) {  // the wave-uniform rescale, once per 8 steps here
      int mx = __float_as_int(bm);
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
      bm = __builtin_amdgcn_ldexpf(bm, 127 + 60 - ((mx >> 23) & 0xff));
    }
    if (STEP == 1 && (it & 0) == 0) {  // the wave-uniform rescale, once per 8 steps here
      const int hi = int(__double_as_longlong(bl) >> 32);
      int mx = hi;
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
      const int sh = 1023 + 500 - ((mx >> 20) & 0x7ff);
      bl = __builtin_amdgcn_ldexp(bl, sh);
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = post + acc[0] + acc[1] + acc[2] + float(acc64[0] + acc64[1] + acc64[2]) + b + float(bl) + bm + bx;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
double run(int waves) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4 * 64 * waves);
  hipMalloc(&cyc, 8 * waves);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.0f);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.0f);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  hipFree(out);
  hipFree(cyc);
  return double(c) / (N_IT * 8);
}
template <int STEP>
double run_step(int waves) {
  float *out, *tab;
  long long* cyc;
  hipMalloc(&out, 4 * 64 * waves);
  hipMalloc(&tab, 4 * 1024);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i < 512 ? (STEP == 0 ? -2.0f - (i % 7) : 0.01f * (1 + i % 7)) : -3.0f - (i % 5);
  hipMemcpy(tab, h.data(), 4096, hipMemcpyHostToDevice);
  hipMalloc(&cyc, 8 * waves);
  hipLaunchKernelGGL(kstep<STEP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.0f, tab);
  hipLaunchKernelGGL(kstep<STEP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 0.0f, tab);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  return double(c) / (N_IT * 8);
}

int main() {
  const char* names[] = {"v_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_ldexp_f64", "v_cvt_f32_f64 + add", "v_cvt_f64_f32 + mul + add",
                         "v_exp_f32 + mul", "v_ldexp_f32", "v_frexp_{exp,mant}_f64 + 2 adds", "v_frexp_{exp,mant}_f32 + 2 adds",
                         "dpp wave_shl1 f32", "dpp wave_shl1 f64 (2 movs)", "v_log_f32 + add", "2 v_max_i32 + sub", "v_fract_f32 + add"};
  double ind[16], dep[16];
  ind[0] = run<0>(1), ind[1] = run<1>(1), ind[2] = run<2>(1), ind[3] = run<3>(1), ind[4] = run<4>(1), ind[5] = run<5>(1), ind[6] = run<6>(1),
  ind[7] = run<7>(1), ind[8] = run<8>(1), ind[9] = run<9>(1), ind[10] = run<10>(1), ind[11] = run<11>(1), ind[12] = run<12>(1), ind[13] = run<13>(1),
  ind[14] = run<14>(1), ind[15] = run<15>(1);
  dep[0] = run<100>(1), dep[1] = run<101>(1), dep[2] = run<102>(1), dep[3] = run<103>(1), dep[4] = run<104>(1), dep[5] = run<105>(1),
  dep[6] = run<106>(1), dep[7] = run<107>(1), dep[8] = run<108>(1), dep[9] = run<109>(1), dep[10] = run<110>(1), dep[11] = run<111>(1),
  dep[12] = run<112>(1), dep[13] = run<113>(1), dep[14] = run<114>(1), dep[15] = run<115>(1);
  printf("cycles per wave64 operation group, one wave: 8 independent chains | 1 dependent chain\n");
  for (int i = 0; i < 16; ++i) printf("  %-34s %6.1f %6.1f\n", names[i], ind[i], dep[i]);
  printf("v_fma_f32, 8 independent chains, cycles per operation of one wave at 1 / 4 / 8 / 16 / 32 waves per workgroup: %.1f %.1f %.1f %.1f %.1f\n",
         run<0>(1), run<0>(4), run<0>(8), run<0>(16), run<0>(32));
  printf("v_exp_f32 + mul, the same: %.1f %.1f %.1f %.1f %.1f\n", run<7>(1), run<7>(4), run<7>(8), run<7>(16), run<7>(32));
  printf("backward step (UNIT, arc gradients), one wave, dependent through the steps: cycles per step\n");
  for (int waves : {1, 4, 8, 16}) {
    printf(" %d wave(s) per workgroup (%d per SIMD): cycles per step of one wave\n", waves, (waves + 3) / 4);
    printf("  log domain (band.hip)             %6.1f\n", run_step<0>(waves));
    printf("  linear float64, wave scale        %6.1f\n", run_step<1>(waves));
    printf("  linear float32, per-lane exponent %6.1f\n", run_step<2>(waves));
    printf("  linear float32, wave scale        %6.1f\n", run_step<3>(waves));
  }
  return 0;
}
