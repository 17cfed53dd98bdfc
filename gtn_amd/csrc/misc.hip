// misc.hip -- linear-chain (emissions) kernels and small batched helpers.
//
// forwardScore / viterbiScore of linearGraph(M, C) (gtn/creations.cpp:20-33) is
//   score = sum_t reduce_c w[t][c]      (reduce = log-sum-exp | max)
// because node t+1's only predecessor is node t (shortest.cpp:118-137 applied to
// a chain).  So the emissions normaliser of a CTC/ASG loss is a pure streaming
// row reduction over the [M][C] weight tensor: one wave64 per row, float4
// loads, shuffle reductions; HBM traffic = 4*M*C bytes (+ the same again to
// write the gradient in the backward kernel).
#include <hip/hip_runtime.h>

#include <climits>

#include "kernels.h"

namespace gtnx {
namespace {

#define NEG_INF (-__builtin_huge_valf())
#define POS_INF (__builtin_huge_valf())
constexpr int kBlock = 256;
constexpr int kSplits = 8;  // row chunks per graph => n*8 workgroups

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// row reduce with one wave; returns (lse or max) and, for tropical, the first arg-max
template <bool TROPICAL>
__device__ __forceinline__ float row_reduce(const float* __restrict__ row, int C, int lane, int* argmax) {
  float mx = NEG_INF;
  int am = INT_MAX;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    if (v > mx) {
      mx = v;
      am = c;
    }
  }
  if (TROPICAL) {
    // first max in label order (in-arcs are label-ascending, strict '>')
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(mx, o, 64);
      const int a2 = __shfl_xor(am, o, 64);
      if (v2 > mx || (v2 == mx && a2 < am)) {
        mx = v2;
        am = a2;
      }
    }
    if (argmax) *argmax = (mx > NEG_INF) ? am : -1;
    return C > 0 ? mx : NEG_INF;
  }
  mx = wave_max(mx);
  if (C == 0) return NEG_INF;
  if (mx == POS_INF || mx == NEG_INF) return mx;
  float sum = 0.0f;
  for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
  sum = wave_sum(sum);
  return mx + log1pf(sum - 1.0f);
}

// ---- log-semiring rows held in registers (C % 4 == 0, C <= 1024) ------------------
// One wave reduces TWO rows per iteration: each lane owns up to four float4 pieces of
// a row (one 16-byte load each, all in flight together), the max and the exp-sum go
// across the wave with DPP row shifts / broadcasts (VALU speed; the __shfl_xor form
// is six dependent LDS permutes per reduction), and the row is read from HBM once --
// the backward kernel produces the gradient from the same registers.
#define GTNX_DPP_F(x, op, old, ctrl, rmask) \
  x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float dpp_add(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_max_dpp(float x) {
  GTNX_DPP_F(x, fmaxf, x, 0x111, 0xf);
  GTNX_DPP_F(x, fmaxf, x, 0x112, 0xf);
  GTNX_DPP_F(x, fmaxf, x, 0x114, 0xf);
  GTNX_DPP_F(x, fmaxf, x, 0x118, 0xf);
  GTNX_DPP_F(x, fmaxf, x, 0x142, 0xa);
  GTNX_DPP_F(x, fmaxf, x, 0x143, 0xc);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x111, 0xf);
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x112, 0xf);
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x114, 0xf);
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x118, 0xf);
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x142, 0xa);
  GTNX_DPP_F(x, dpp_add, 0.0f, 0x143, 0xc);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

constexpr int kRowPieces = 4;  // float4 pieces per lane: C <= 4 * 256
constexpr int kRowsPerIter = 2;

template <bool BWD>
__global__ __launch_bounds__(kBlock) void linear_rows_kernel(const LinArgs* __restrict__ args) {
  const LinArgs a = args[blockIdx.x / kSplits];
  const int split = blockIdx.x % kSplits;
  const int rows_per = (a.M + kSplits - 1) / kSplits;
  const int r0 = split * rows_per, r1 = min(a.M, r0 + rows_per);
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int C = a.C;
  const float delta = BWD ? *a.delta : 0.0f;
  float acc = 0.0f;
  for (int rb = r0 + wave * kRowsPerIter; rb < r1; rb += (kBlock / 64) * kRowsPerIter) {
    gtnx_f4 v[kRowsPerIter][kRowPieces];
#pragma unroll
    for (int q = 0; q < kRowsPerIter; ++q) {
      const int r = min(rb + q, r1 - 1);
      const GTNX_G float* row = a.w + (size_t)r * C;
#pragma unroll
      for (int k = 0; k < kRowPieces; ++k) {
        const int c = lane * 4 + k * 256;
        v[q][k] = gtnx_f4{NEG_INF, NEG_INF, NEG_INF, NEG_INF};
        if (k * 256 < C && c < C) v[q][k] = *reinterpret_cast<const GTNX_G gtnx_f4*>(row + c);
      }
    }
#pragma unroll
    for (int q = 0; q < kRowsPerIter; ++q) {
      float mx = NEG_INF;
#pragma unroll
      for (int k = 0; k < kRowPieces; ++k)
        if (k * 256 < C)  // wave-uniform: pieces past the row end cost nothing
          mx = fmaxf(fmaxf(fmaxf(v[q][k].x, v[q][k].y), fmaxf(v[q][k].z, v[q][k].w)), mx);
      mx = wave_max_dpp(mx);
      float red = mx;
      if (mx != POS_INF && mx != NEG_INF) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < kRowPieces; ++k)
          if (k * 256 < C)
            sum += expf(v[q][k].x - mx) + expf(v[q][k].y - mx) + expf(v[q][k].z - mx) + expf(v[q][k].w - mx);
        sum = wave_sum_dpp(sum);
        red = mx + log1pf(sum - 1.0f);
      }
      const bool real = rb + q < r1;
      if (!BWD) {
        if (real) acc += red;
      } else if (real) {
        GTNX_G float* grow = a.grad + (size_t)(rb + q) * C;
#pragma unroll
        for (int k = 0; k < kRowPieces; ++k) {
          const int c = lane * 4 + k * 256;
          if (c < C) {
            gtnx_f4 g;
            g.x = expf(v[q][k].x - red) * delta;  // exp(score[t] + w - score[t+1])
            g.y = expf(v[q][k].y - red) * delta;
            g.z = expf(v[q][k].z - red) * delta;
            g.w = expf(v[q][k].w - red) * delta;
            GTNX_G gtnx_f4* gp = reinterpret_cast<GTNX_G gtnx_f4*>(grow + c);
            if (a.accumulate) {
              const gtnx_f4 o = *gp;
              g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
            }
            *gp = g;
          }
        }
      }
    }
  }
  if (BWD) return;
  __shared__ float sh[kBlock / 64];
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
    a.partial[split] = t;
  }
}

template <bool TROPICAL>
__global__ __launch_bounds__(kBlock) void linear_forward_kernel(const LinArgs* __restrict__ args) {
  const LinArgs a = args[blockIdx.x / kSplits];
  const int split = blockIdx.x % kSplits;
  const int rows_per = (a.M + kSplits - 1) / kSplits;
  const int r0 = split * rows_per, r1 = min(a.M, r0 + rows_per);
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  float acc = 0.0f;
  for (int r = r0 + wave; r < r1; r += kBlock / 64)
    acc += row_reduce<TROPICAL>(a.w + (size_t)r * a.C, a.C, lane, nullptr);
  __shared__ float sh[kBlock / 64];
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
    a.partial[split] = t;
  }
}

__global__ void linear_finish_kernel(const LinArgs* __restrict__ args, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const LinArgs a = args[i];
  float t;
  if (a.M == 0) {
    t = NEG_INF;  // start node is not accepting: empty accept reduction
  } else {
    t = 0.0f;
    for (int s = 0; s < kSplits; ++s) t += a.partial[s];
  }
  *a.out_score = t;
}

template <bool TROPICAL>
__global__ __launch_bounds__(kBlock) void linear_backward_kernel(const LinArgs* __restrict__ args) {
  const LinArgs a = args[blockIdx.x / kSplits];
  const int split = blockIdx.x % kSplits;
  const int rows_per = (a.M + kSplits - 1) / kSplits;
  const int r0 = split * rows_per, r1 = min(a.M, r0 + rows_per);
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const float delta = *a.delta;
  for (int r = r0 + wave; r < r1; r += kBlock / 64) {
    const float* row = a.w + (size_t)r * a.C;
    float* grow = a.grad + (size_t)r * a.C;
    int am = -1;
    const float red = row_reduce<TROPICAL>(row, a.C, lane, &am);
    for (int c = lane; c < a.C; c += 64) {
      float g;
      if (TROPICAL)
        g = (c == am) ? 1.0f : 0.0f;
      else
        g = expf(row[c] - red);  // exp(score[t] + w - score[t+1])
      grow[c] = a.accumulate ? grow[c] + g * delta : g * delta;
    }
  }
}

__global__ void fill_i32_kernel(int* p, int v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void fill_f32_kernel(float* p, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// (args == nullptr: a single record, handed over in the kernel's own argument block -- the per-graph functions on one
// utterance, where a 24-byte table copied to the device is one more operation of a dependent chain)
__global__ void scalar_combine_kernel(const ScalarArgs* __restrict__ args, ScalarArgs one, int n, float sa, float sb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ScalarArgs a = args ? args[i] : one;
  float v = sa * (*a.a);
  if (a.b) v += sb * (*a.b);
  *a.out = v;
  if (a.mirror) *a.mirror = v;  // (pinned host memory: item() of this result waits for the stream and reads it there)
}
// a small copy as a kernel of our own (setWeights from a device pointer, a table out of a pinned block): the runtime's
// hipMemcpyAsync costs the host twice what a launch does, and a single utterance through the per-graph functions waits
// for the host at the head of its chain
__global__ void copy_small_kernel(void* __restrict__ dst, const void* __restrict__ src, size_t bytes, int vec) {
  const size_t i0 = blockIdx.x * size_t(blockDim.x) + threadIdx.x, stride = size_t(gridDim.x) * blockDim.x;
  if (vec == 16) {
    for (size_t i = i0; i < bytes / 16; i += stride) static_cast<uint4*>(dst)[i] = static_cast<const uint4*>(src)[i];
  } else if (vec == 4) {
    for (size_t i = i0; i < bytes / 4; i += stride) static_cast<uint32_t*>(dst)[i] = static_cast<const uint32_t*>(src)[i];
  } else {
    for (size_t i = i0; i < bytes; i += stride) static_cast<uint8_t*>(dst)[i] = static_cast<const uint8_t*>(src)[i];
  }
}
// a scalar op's gradient function: o0 = s0 * d, o1 = s1 * d (o1 may be null); seed != null: d is the seed of a
// backward pass from this result (1, written to *seed: autograd.cpp:57-62) -- seed, and both inputs' gradients, in
// one launch (what scalar_seed_kernel is to a batch record)
__global__ void scalar_fan_kernel(const ScalarFanArgs* __restrict__ args, ScalarFanArgs one, int n, float s0, float s1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ScalarFanArgs a = args ? args[i] : one;
  float d = 1.0f;
  if (a.seed) *a.seed = 1.0f;
  else d = *a.d;
  *a.o0 = s0 * d;
  if (a.o1) *a.o1 = s1 * d;
}

__global__ void axpy_batch_kernel(const AxpyArgs* __restrict__ args, int atomic) {
  const AxpyArgs a = args[blockIdx.y];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < a.n; i += stride) {
    if (atomic == 2)
      a.dst[i] = a.scale * a.src[i];  // copy mode
    else if (atomic)
      atomicAdd(a.dst + i, a.scale * a.src[i]);
    else
      a.dst[i] += a.scale * a.src[i];
  }
}

// one grid row per segment: HBM-bound copy, 16 bytes per lane when both addresses are 16-byte aligned
__global__ void copy_segments_kernel(const CopySeg* __restrict__ segs) {
  const CopySeg sg = segs[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(sg.dst) | reinterpret_cast<uintptr_t>(sg.src)) & 15) == 0) {
    const int64_t n16 = sg.bytes >> 4;
    const uint4* s = static_cast<const uint4*>(sg.src);
    uint4* d = static_cast<uint4*>(sg.dst);
    for (int64_t k = i; k < n16; k += stride) d[k] = s[k];
    const int64_t n4 = sg.bytes >> 2;
    for (int64_t k = (n16 << 2) + i; k < n4; k += stride)
      static_cast<uint32_t*>(sg.dst)[k] = static_cast<const uint32_t*>(sg.src)[k];
  } else {
    const int64_t n4 = sg.bytes >> 2;
    for (int64_t k = i; k < n4; k += stride) static_cast<uint32_t*>(sg.dst)[k] = static_cast<const uint32_t*>(sg.src)[k];
  }
}

__global__ void gather_scalars_kernel(const float* const* __restrict__ ptrs, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *ptrs[i];
}

__global__ void scatter_add_kernel(const ScatterArgs* __restrict__ args) {
  const ScatterArgs a = args[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
    atomicAdd(a.grad + a.idx[i], a.delta[i]);
}

__global__ void linear_materialize_kernel(int M, int C, int* src, int* dst, int* il, int* ol) {
  const size_t A = (size_t)M * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < A; i += stride) {
    const int m = int(i / C), c = int(i % C);
    src[i] = m;
    dst[i] = m + 1;
    il[i] = c;
    ol[i] = c;
  }
}

int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return int(g);
}

} // namespace

void launch_linear_forward(const LinArgs* d, int n, int tropical, int vec_rows, hipStream_t st) {
  if (n <= 0) return;
  if (!tropical && vec_rows)
    hipLaunchKernelGGL(linear_rows_kernel<false>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
  else if (tropical)
    hipLaunchKernelGGL(linear_forward_kernel<true>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL(linear_forward_kernel<false>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
  hipLaunchKernelGGL(linear_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d, n);
}

void launch_linear_backward(const LinArgs* d, int n, int tropical, int vec_rows, hipStream_t st) {
  if (n <= 0) return;
  if (!tropical && vec_rows)
    hipLaunchKernelGGL(linear_rows_kernel<true>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
  else if (tropical)
    hipLaunchKernelGGL(linear_backward_kernel<true>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL(linear_backward_kernel<false>, dim3(n * kSplits), dim3(kBlock), 0, st, d);
}

namespace {
__global__ void vec_axpby_kernel(float* out, const float* a, const float* b, size_t n, float sa, float sb, int accumulate) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    float v = sa * a[i];
    if (b) v += sb * b[i];
    out[i] = accumulate ? out[i] + v : v;
  }
}
} // namespace
void launch_vec_axpby(float* out, const float* a, const float* b, size_t n, float sa, float sb, int accumulate, hipStream_t st) {
  if (n) hipLaunchKernelGGL(vec_axpby_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, a, b, n, sa, sb, accumulate);
}
// seed of a backward pass over a scalar op's result (batch.cpp: batch_backward): root = 1, and what the op's own
// gradient function would add to its inputs next (s0 / s1 times that 1) in the same launch
__global__ void scalar_seed_kernel(float* root, float* g0, float s0, int acc0, float* g1, float s1, int acc1, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    root[i] = 1.0f;
    if (g0) g0[i] = acc0 ? g0[i] + s0 : s0;
    if (g1) g1[i] = acc1 ? g1[i] + s1 : s1;
  }
}
void launch_scalar_seed(float* root, float* g0, float s0, int acc0, float* g1, float s1, int acc1, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(scalar_seed_kernel, dim3(grid_for(n)), dim3(256), 0, st, root, g0, s0, acc0, g1, s1, acc1, n);
}
void launch_fill_i32(int* p, int v, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, v, n);
}
void launch_fill_f32(float* p, float v, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, v, n);
}
void launch_scalar_combine(const ScalarArgs* d, int n, float sa, float sb, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(scalar_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, ScalarArgs{}, n, sa, sb);
}
void launch_scalar_combine_one(const ScalarArgs& a, float sa, float sb, hipStream_t st) {
  hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, st, static_cast<const ScalarArgs*>(nullptr), a, 1, sa, sb);
}
void launch_copy_small(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (!bytes) return;
  const uintptr_t both = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | uintptr_t(bytes);
  const int vec = both % 16 == 0 ? 16 : both % 4 == 0 ? 4 : 1;
  hipLaunchKernelGGL(copy_small_kernel, dim3(grid_for(bytes / size_t(vec), 256, 512)), dim3(256), 0, st, dst, src, bytes, vec);
}
void launch_scalar_fan(const ScalarFanArgs* d, int n, float s0, float s1, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(scalar_fan_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, ScalarFanArgs{}, n, s0, s1);
}
void launch_scalar_fan_one(const ScalarFanArgs& a, float s0, float s1, hipStream_t st) {
  hipLaunchKernelGGL(scalar_fan_kernel, dim3(1), dim3(64), 0, st, static_cast<const ScalarFanArgs*>(nullptr), a, 1, s0, s1);
}
void launch_axpy_batch(const AxpyArgs* d, int n, int64_t maxn, int atomic, hipStream_t st) {
  if (n <= 0 || maxn <= 0) return;
  hipLaunchKernelGGL(axpy_batch_kernel, dim3(grid_for((size_t)maxn, 256, 1024), n), dim3(256), 0, st, d, atomic);
}
void launch_copy_segments(const CopySeg* d, int n, int64_t max_bytes, hipStream_t st) {
  if (n <= 0 || max_bytes <= 0) return;
  // ~8 workgroups per segment at most: a region's batch has hundreds of segments
  hipLaunchKernelGGL(copy_segments_kernel, dim3(grid_for((size_t)(max_bytes >> 4) + 1, 256, 8), n), dim3(256), 0, st, d);
}
void launch_gather_scalars(const float* const* d_ptrs, float* out, int n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(gather_scalars_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_ptrs, out, n);
}
void launch_scatter_add(const ScatterArgs* d, int n, int maxn, hipStream_t st) {
  if (n <= 0 || maxn <= 0) return;
  hipLaunchKernelGGL(scatter_add_kernel, dim3(grid_for((size_t)maxn, 256, 256), n), dim3(256), 0, st, d);
}
void launch_linear_materialize(int M, int C, int* src, int* dst, int* il, int* ol, hipStream_t st) {
  const size_t A = (size_t)M * C;
  if (A) hipLaunchKernelGGL(linear_materialize_kernel, dim3(grid_for(A)), dim3(256), 0, st, M, C, src, dst, il, ol);
}

} // namespace gtnx
