// lazy_pair.hip -- forwardScore and its gradient over chain o G for a SMALL G, one
// workgroup per (utterance, G) pair, without ever building the product.
//
// This is the CTC shape: every utterance has its OWN
// target graph (a few hundred nodes, <= 4 arcs per node), so the batched time-step kernels
// of lazy.hip (one G shared by the batch) do not apply, and building the lattice
// (compose.cpp:377-522) only to stream it twice (shortest.cpp:86-170 and its gradFunc,
// shortest.cpp:33-62; compose.cpp:496-518) moves 70x more bytes than the problem holds:
// the lattice is T shifted copies of G, and everything that varies with t is one
// emission row.  Here
//   * lane n owns node n of G for the whole sweep; its (<= 4) in-arcs (forward) or
//     out-arcs (backward) -- neighbour, matched label, weight -- live in registers;
//   * alpha[t] / beta[t+1] are a two-row LDS ring (one barrier per time step);
//   * emission rows are staged through LDS in chunks, the next chunk's loads in
//     flight while the current one is consumed (coalesced dwords, every byte of the
//     emissions read exactly once per sweep);
//   * alpha is kept in HBM ([T+1][N] per pair) for the backward sweep, which reads it
//     back through registers one chunk ahead;
//   * the emission gradient of time step t is summed in an LDS row (ds_add_f32; the
//     hottest label -- CTC's blank -- is pre-summed across the wave by DPP) and stored
//     exactly once, coalesced, while step t-1 runs; the gradient of an arc of G is
//     a register accumulator of its owning lane, stored once at the end.
// HBM traffic per pair: forward 4TC + 4TN, backward 8TC + 4TN (+ G, a few KB).
// The sweeps are latency-bound (T dependent steps of ~LDS-latency work each), which
// is why two or more workgroups share a CU.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr float NEG_INF = -__builtin_inff();
constexpr float POS_INF = __builtin_inff();
constexpr int KC = 4;   // arcs per node held in registers (host checks max degree <= KC)
constexpr int SR = 16;  // emission floats a lane stages per chunk: a chunk is <= SR * BLK floats

// max + log(sum exp(. - max)): sum >= 1, so the raw v_log_f32 is as accurate in absolute
// terms as the reference's log1p form; an all -inf (or +inf) row keeps its max
__device__ __forceinline__ float lse4(const float (&x)[KC]) {
  const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
  const bool fin = fabsf(mx) != POS_INF;
  const float m2 = fin ? mx : 0.0f;
  float sum = 0.0f;
#pragma unroll
  for (int j = 0; j < KC; ++j) sum += __expf(x[j] - m2);
  return fin ? mx + 0.69314718f * __builtin_amdgcn_logf(sum) : mx;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e.
// it would wait every time step for the alpha / gradient-row store just issued (and for
// the next chunk's staged loads) to come back from HBM.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define GTNX_PAIR_DPP_ADD(x, ctrl, rmask) \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
// wave64 sum by DPP row shifts / broadcasts; all lanes active; lane 63 holds the total
__device__ __forceinline__ float wave_sum_to_lane63(float x) {
  GTNX_PAIR_DPP_ADD(x, 0x111, 0xf);  // row_shr:1
  GTNX_PAIR_DPP_ADD(x, 0x112, 0xf);  // row_shr:2
  GTNX_PAIR_DPP_ADD(x, 0x114, 0xf);  // row_shr:4
  GTNX_PAIR_DPP_ADD(x, 0x118, 0xf);  // row_shr:8
  GTNX_PAIR_DPP_ADD(x, 0x142, 0xa);  // row_bcast:15
  GTNX_PAIR_DPP_ADD(x, 0x143, 0xc);  // row_bcast:31
  return x;
}

template <int BLK>
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();  // red may still be read from an earlier reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int i = 1; i < BLK / 64; ++i) m = fmaxf(m, red[i]);
  return m;
}
template <int BLK>
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int i = 1; i < BLK / 64; ++i) s += red[i];
  return s;
}

template <int BLK>
__device__ __forceinline__ int block_max_int(int v, int* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int m = red[0];
#pragma unroll
  for (int i = 1; i < BLK / 64; ++i) m = max(m, red[i]);
  return m;
}

// registers of one lane's arcs; unused slots: weight -inf, label 0, neighbour 0
struct ArcRegs {
  int other[KC];
  int lab[KC];
  int aid[KC];
  float w[KC];
};
__device__ __forceinline__ void load_arcs(const LazyPair& a, int n, bool out, ArcRegs& r) {
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    r.other[k] = 0;
    r.lab[k] = 0;
    r.aid[k] = -1;
    r.w[k] = NEG_INF;
  }
  if (n >= a.g.N) return;
  const GTNX_G int* off = out ? a.g.out_off : a.g.in_off;
  const GTNX_G gtnx_i4* rec = out ? a.g.out_rec : a.g.in_rec;
  const int r0 = off[n], deg = off[n + 1] - r0;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < deg) {
      const gtnx_i4 q = rec[r0 + k];  // {ilabel, olabel, neighbour, arc id}
      const int l = a.chain_first ? q.x : q.y;
      if (l >= 0 && l < a.C) {
        r.other[k] = q.z;
        r.lab[k] = l;
        r.aid[k] = q.w;
        r.w[k] = a.g.w[q.w];
      }
    }
  }
}

// --------------------------------------------------------------------------
// forward: alpha[t+1][n] = (+)_k alpha[t][src_k] + w_k + em[t][lab_k]
// --------------------------------------------------------------------------
template <int BLK>
__global__ __launch_bounds__(BLK) void lazy_pair_forward_kernel(const LazyPair* __restrict__ pairs) {
  const LazyPair a = pairs[blockIdx.x];
  const int N = a.g.N, T = a.T, C = a.C;
  const int n = threadIdx.x;
  extern __shared__ float lds[];
  float* ring = lds;            // [2][BLK]
  float* ebuf = lds + 2 * BLK;  // [2][SR * BLK]
  __shared__ float red[BLK / 64];
  constexpr int CH = SR * BLK;
  const int R = min(CH / C, 64);  // time steps per chunk (host: C <= CH)
  const bool live = n < N;
  ArcRegs in;
  load_arcs(a, n, false, in);
  const uint8_t fl = live ? a.g.nflags[n] : uint8_t(0);

  float stage[SR];
  auto fetch = [&](int t0) {
    const int cnt = max(0, min(R, T - t0)) * C;
    const GTNX_G float* p = a.em + int64_t(t0) * C;
#pragma unroll
    for (int i = 0; i < SR; ++i) {
      const int idx = i * BLK + n;
      stage[i] = idx < cnt ? p[idx] : 0.0f;
    }
  };
  auto park = [&](int buf) {
    float* e = ebuf + buf * CH;
#pragma unroll
    for (int i = 0; i < SR; ++i) e[i * BLK + n] = stage[i];
  };

  float v = (live && (fl & NF_START)) ? 0.0f : NEG_INF;
  ring[n] = v;
  if (live) a.alpha[n] = v;
  fetch(0);
  park(0);
  __syncthreads();
  int cur = 0, buf = 0;
  for (int t0 = 0; t0 < T; t0 += R) {
    const bool more = t0 + R < T;
    if (more) fetch(t0 + R);
    const int rows = min(R, T - t0);
    for (int r = 0; r < rows; ++r) {
      const float* e = ebuf + buf * CH + r * C;
      const float* ap = ring + cur * BLK;
      float x[KC];
#pragma unroll
      for (int k = 0; k < KC; ++k) x[k] = ap[in.other[k]] + in.w[k] + e[in.lab[k]];
      v = lse4(x);
      ring[(cur ^ 1) * BLK + n] = v;
      if (r == rows - 1 && more) park(buf ^ 1);  // nobody reads that buffer during this chunk
      if (live) a.alpha[int64_t(t0 + r + 1) * N + n] = v;
      lds_barrier();
      cur ^= 1;
    }
    buf ^= 1;
  }
  // score = (+) over accept nodes of alpha[T]  (shortest.cpp:153-167)
  const float f = (live && (fl & NF_ACCEPT)) ? v : NEG_INF;
  const float m = block_max<BLK>(f, red);
  const bool fin = fabsf(m) != POS_INF;
  const float s = block_sum<BLK>(fin ? __expf(f - m) : 0.0f, red);
  if (n == 0) a.score[0] = fin ? m + __logf(s) : m;
}

// --------------------------------------------------------------------------
// backward: beta[t][n] = (+)_k w_k + em[t][lab_k] + beta[t+1][dst_k]; the posterior of
// arc k at step t is exp(alpha[t][n] + w_k + em[t][lab_k] + beta[t+1][dst_k] - score)
// --------------------------------------------------------------------------
template <int BLK, int R>
__global__ __launch_bounds__(BLK) void lazy_pair_backward_kernel(const LazyPair* __restrict__ pairs) {
  const LazyPair a = pairs[blockIdx.x];
  const int N = a.g.N, T = a.T, C = a.C;
  const int n = threadIdx.x;
  extern __shared__ float lds[];
  constexpr int CH = SR * BLK;
  float* ring = lds;                // [2][BLK]
  float* ebuf = lds + 2 * BLK;      // [2][CH]
  float* grow = ebuf + 2 * CH;      // [2][C] emission-gradient rows of two consecutive steps
  __shared__ float red[BLK / 64];
  const bool live = n < N;
  const float Z = a.score[0];
  const float dl = a.delta[0];
  if (!(fabsf(Z) != POS_INF) || T <= 0) {
    // no accepting path (or no time step): every gradient is zero (grad_fixed is pre-zeroed)
    if (a.grad_em)
      for (int64_t i = n; i < int64_t(T) * C; i += BLK) a.grad_em[i] = 0.0f;
    return;
  }
  ArcRegs out;
  load_arcs(a, n, true, out);
  float facc[KC] = {0.0f, 0.0f, 0.0f, 0.0f};
  const uint8_t fl = live ? a.g.nflags[n] : uint8_t(0);

  // If all in-arcs of every node carry ONE matched label (CTC targets, any acceptor built
  // state-per-symbol), the emission gradient needs one term per NODE, not per arc:
  //   d score / d em[t][l] = sum over nodes n with label l of exp(alpha[t+1][n] + beta[t+1][n] - score)
  // -- 2.5x fewer LDS atomics on a CTC target.  nlab: that label (-1: no matching in-arc).
  int nlab = -1;
  bool same = true;
  if (live) {
    const int r0 = a.g.in_off[n], r1 = a.g.in_off[n + 1];
    for (int k = r0; k < r1; ++k) {
      const gtnx_i4 q = a.g.in_rec[k];
      const int l = a.chain_first ? q.x : q.y;
      if (l < 0 || l >= C) continue;
      if (nlab < 0) nlab = l;
      same = same && l == nlab;
    }
  }
  const bool moore = __syncthreads_and(same) != 0;

  // hottest matched label of G (CTC: blank, half of the nodes): summed by DPP, not by the LDS atomic unit
  int* hist = reinterpret_cast<int*>(grow);
  for (int c = n; c < 2 * C; c += BLK) grow[c] = 0.0f;
  __syncthreads();
  if (moore) {
    if (nlab >= 0) atomicAdd(&hist[nlab], 1);
  } else {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (out.aid[k] >= 0) atomicAdd(&hist[out.lab[k]], 1);
  }
  __syncthreads();
  int key = 0;
  for (int c = n; c < C; c += BLK) {
    const int h = hist[c];
    if (h >= 8) key = max(key, (h << 13) | c);  // C <= SR * 512 = 8192; h <= 4 * 512
  }
  const int hk = block_max_int<BLK>(key, reinterpret_cast<int*>(red));
  __syncthreads();  // red is reused below
  const int hot = hk ? (hk & 8191) : -1;
  for (int c = n; c < C; c += BLK) grow[c] = 0.0f;

  float stage[SR];
  auto fetch = [&](int t0) {
    const int cnt = max(0, min(R, T - t0)) * C;
    const GTNX_G float* p = a.em + int64_t(t0) * C;
#pragma unroll
    for (int i = 0; i < SR; ++i) {
      const int idx = i * BLK + n;
      stage[i] = idx < cnt ? p[idx] : 0.0f;
    }
  };
  auto park = [&](int buf) {
    float* e = ebuf + buf * CH;
#pragma unroll
    for (int i = 0; i < SR; ++i) e[i * BLK + n] = stage[i];
  };
  float a_cur[R], a_nxt[R];
  auto fetch_alpha = [&](int t0, float (&dst)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) dst[j] = (live && t0 + j < T) ? a.alpha[int64_t(t0 + j) * N + n] : NEG_INF;
  };

  float b_hi = (live && (fl & NF_ACCEPT)) ? 0.0f : NEG_INF;  // beta[t+1][n], this lane's previous result
  float a_hi = live ? a.alpha[int64_t(T) * N + n] : NEG_INF;   // alpha[t+1][n]
  const bool want_fixed = a.grad_fixed != nullptr;
  ring[n] = b_hi;  // beta[T]
  const int t_last = ((T - 1) / R) * R;
  fetch(t_last);
  fetch_alpha(t_last, a_cur);
  park(0);
  __syncthreads();
  int cur = 0, buf = 0, gr = 0;
  for (int t0 = t_last; t0 >= 0; t0 -= R) {
    const bool more = t0 > 0;
    if (more) {
      fetch(t0 - R);
      fetch_alpha(t0 - R, a_nxt);
    }
#pragma unroll
    for (int j = R - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (t < T) {  // uniform
        const float* e = ebuf + buf * CH + j * C;
        const float* bn = ring + cur * BLK;
        float x[KC], p[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) x[k] = out.w[k] + e[out.lab[k]] + bn[out.other[k]];
        // beta[t][n] = m + log sum_k e_k with e_k = exp(x_k - m); the same e_k give the arc
        // posteriors: exp(alpha[t][n] + x_k - Z) = e_k * exp(alpha[t][n] + m - Z), and the
        // second factor is <= 1 (alpha + beta <= Z), so nothing overflows
        const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        const bool fin = fabsf(mx) != POS_INF;
        const float m2 = fin ? mx : 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          p[k] = __expf(x[k] - m2);
          sum += p[k];
        }
        const float b_lo = fin ? mx + 0.69314718f * __builtin_amdgcn_logf(sum) : mx;
        ring[(cur ^ 1) * BLK + n] = b_lo;
        float hv = 0.0f;
        float* gcur = grow + gr * C;
        if (!moore || want_fixed) {
          const float sc = fin ? __expf(a_cur[j] + mx - Z) * dl : 0.0f;
#pragma unroll
          for (int k = 0; k < KC; ++k) {
            p[k] *= sc;
            facc[k] += p[k];
          }
        }
        if (moore) {
          // one term per node: its posterior at time t+1 goes to row t under its in-label
          const float g = nlab >= 0 ? __expf(a_hi + b_hi - Z) * dl : 0.0f;
          if (nlab == hot) hv = g;
          else if (g != 0.0f) atomicAdd(&gcur[nlab], g);
        } else {
#pragma unroll
          for (int k = 0; k < KC; ++k) {
            if (out.lab[k] == hot) hv += p[k];
            else if (p[k] != 0.0f) atomicAdd(&gcur[out.lab[k]], p[k]);
          }
        }
        if (hot >= 0) {
          hv = wave_sum_to_lane63(hv);
          if ((n & 63) == 63 && hv != 0.0f) atomicAdd(&gcur[hot], hv);
        }
        a_hi = a_cur[j];
        b_hi = b_lo;
        // land the next chunk before this step's stores: the wait in front of it then only
        // covers memory operations issued at least one step ago
        if (j == 0 && more) park(buf ^ 1);
        // the row of step t+1 is complete (barrier of that step): store it, clear it
        if (t + 1 < T) {
          float* gprev = grow + (gr ^ 1) * C;
          for (int c = n; c < C; c += BLK) {
            const float g = gprev[c];
            gprev[c] = 0.0f;
            if (a.grad_em) a.grad_em[int64_t(t + 1) * C + c] = g;
          }
        }
        lds_barrier();
        cur ^= 1;
        gr ^= 1;
      }
    }
    buf ^= 1;
#pragma unroll
    for (int j = 0; j < R; ++j) a_cur[j] = a_nxt[j];
  }
  if (a.grad_em) {
    const float* g0 = grow + (gr ^ 1) * C;  // the row of step 0
    for (int c = n; c < C; c += BLK) a.grad_em[c] = g0[c];
  }
  if (a.grad_fixed) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (out.aid[k] >= 0) a.grad_fixed[out.aid[k]] = facc[k];
  }
}

template <int BLK>
size_t pair_lds_bytes(int C, bool backward) {
  return sizeof(float) * (size_t(2 * BLK) + size_t(2 * SR * BLK) + (backward ? size_t(2 * C) : 0));
}

// The sweeps are latency-bound and share a CU's VALU / LDS-atomic unit: n workgroups finish
// soonest when every CU holds ceil(n / CUs) of them.  The dispatcher packs as many as fit,
// so the LDS request is padded until exactly that many fit.
size_t balanced_lds(size_t need, int n, int cus) {
  const int per_cu = std::max(1, (n + cus - 1) / std::max(cus, 1));
  const size_t cap = (160 * 1024 - 1024) / size_t(per_cu) / 256 * 256;  // largest request of which per_cu fit
  return std::max(need, cap);
}

template <int BLK, int R>
void launch_bwd(const LazyPair* d_pairs, int n, int C, int cus, hipStream_t st) {
  hipLaunchKernelGGL((lazy_pair_backward_kernel<BLK, R>), dim3(n), dim3(BLK),
                     balanced_lds(pair_lds_bytes<BLK>(C, true), n, cus), st, d_pairs);
}

} // namespace

int lazy_pair_max_nodes() { return 512; }
int lazy_pair_max_degree() { return KC; }
int lazy_pair_block(int max_nodes) { return max_nodes <= 256 ? 256 : 512; }
int lazy_pair_max_labels(int block) { return SR * block; }  // one emission row must fit a chunk

namespace {
void pair_attrs() {
  static std::atomic<uint64_t> attr_done{0};
  gtnx_first_on_device first{attr_done};
  if (!first) return;
  // 512-lane workgroups (and C in the thousands) need more than the default 64 KB of dynamic LDS
  const int lim = 160 * 1024 - 512;
  const void* fns[] = {reinterpret_cast<const void*>(lazy_pair_backward_kernel<256, 16>),
                       reinterpret_cast<const void*>(lazy_pair_backward_kernel<256, 4>),
                       reinterpret_cast<const void*>(lazy_pair_backward_kernel<256, 1>),
                       reinterpret_cast<const void*>(lazy_pair_backward_kernel<512, 16>),
                       reinterpret_cast<const void*>(lazy_pair_backward_kernel<512, 4>),
                       reinterpret_cast<const void*>(lazy_pair_backward_kernel<512, 1>),
                       reinterpret_cast<const void*>(lazy_pair_forward_kernel<256>),
                       reinterpret_cast<const void*>(lazy_pair_forward_kernel<512>)};
  for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
}
} // namespace

void launch_lazy_pair_forward(const LazyPair* d_pairs, int n, int block, int C, int cus, hipStream_t st) {
  if (n <= 0) return;
  pair_attrs();
  if (block == 256)
    hipLaunchKernelGGL(lazy_pair_forward_kernel<256>, dim3(n), dim3(256),
                       balanced_lds(pair_lds_bytes<256>(C, false), n, cus), st, d_pairs);
  else
    hipLaunchKernelGGL(lazy_pair_forward_kernel<512>, dim3(n), dim3(512),
                       balanced_lds(pair_lds_bytes<512>(C, false), n, cus), st, d_pairs);
}

void launch_lazy_pair_backward(const LazyPair* d_pairs, int n, int block, int C, int cus, hipStream_t st) {
  if (n <= 0) return;
  pair_attrs();
  // steps per chunk: the largest of 16 / 4 / 1 whose emission rows fit one chunk
  const int ch = SR * block;
  const int r = 16 * C <= ch ? 16 : (4 * C <= ch ? 4 : 1);
  if (block == 256) {
    if (r == 16) launch_bwd<256, 16>(d_pairs, n, C, cus, st);
    else if (r == 4) launch_bwd<256, 4>(d_pairs, n, C, cus, st);
    else launch_bwd<256, 1>(d_pairs, n, C, cus, st);
  } else {
    if (r == 16) launch_bwd<512, 16>(d_pairs, n, C, cus, st);
    else if (r == 4) launch_bwd<512, 4>(d_pairs, n, C, cus, st);
    else launch_bwd<512, 1>(d_pairs, n, C, cus, st);
  }
}

} // namespace gtnx
