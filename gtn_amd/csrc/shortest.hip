// shortest.hip -- forwardScore / viterbiScore / viterbiPath and their gradients
// as persistent, level-scheduled CDNA4 kernels.
//
// Replaces gtn/functions/shortest.cpp:86-188 (Kahn-queue shortest distance),
// :33-82 (its gradient) and :190-272 (shortest path).
//
// Execution model (MI355X): ONE workgroup per graph walks the graph's dependency
// levels in order; a batch of B graphs is ONE launch of B workgroups (B >= 512
// fills the 256 CUs twice over).  Within a level every node is independent: G
// lanes of a wave64 cooperate on one node (G = 1, 8 or 64 by mean in-degree),
// stream its in-arc row (src position, weight) from the CSR in HBM with
// coalesced loads, gather the source scores, and reduce max / arg-max / sum-exp
// with wave shuffles.  Levels are separated by one workgroup barrier; scores are
// written once (4 B/node) and re-read through L1/L2.
// Algorithmic HBM bytes per graph: 8*A + 8*N (row offsets, src id + weight per
// in-arc, score write) -- the figure bench.py's roofline uses.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "kernels.h"

namespace gtnx {

namespace {

constexpr int kBlock = 256;
#define NEG_INF (-__builtin_huge_valf())
#define POS_INF (__builtin_huge_valf())

template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

// lexicographic (value desc, rank asc) arg-max across G lanes; NaN never wins,
// -inf never claims an arg (shortest.cpp:124-127 starts from max = -inf, strict >)
template <int G>
__device__ __forceinline__ void group_argmax(float& v, int& rank, int& payload) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    float v2 = __shfl_xor(v, o, G);
    int r2 = __shfl_xor(rank, o, G);
    int p2 = __shfl_xor(payload, o, G);
    if (v2 > v || (v2 == v && r2 < rank)) {
      v = v2;
      rank = r2;
      payload = p2;
    }
  }
}

__device__ __forceinline__ float finish_lse(float mx, float sum_exp, int cnt) {
  // shortest.cpp:102-114
  if (cnt == 0) return NEG_INF;
  if (mx == POS_INF || mx == NEG_INF) return mx;
  return mx + log1pf(sum_exp - 1.0f);
}

// --------------------------------------------------------------------------
// forward sweep
// --------------------------------------------------------------------------
template <int MODE, int G>
__global__ __launch_bounds__(kBlock) void sd_forward_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  const DSched s = a.s;
  const int tid = threadIdx.x;
  const int sub = tid % G, grp = tid / G;
  constexpr int NGRP = kBlock / G;
  const bool tie_by_arc = (s.flags & SCHED_TIE_BY_ARC) != 0;
  float* __restrict__ scores = a.scores;

  for (int l = 0; l < s.L; ++l) {
    const int lo = s.level_off[l], hi = s.level_off[l + 1];
    for (int p = lo + grp; p < hi; p += NGRP) {
      const int r0 = s.row_off[p], r1 = s.row_off[p + 1];
      const bool is_start = (s.pflags[p] & NF_START) != 0;
      float mx = NEG_INF;
      int best_rank = INT_MAX, best = -1;
      for (int k = r0 + sub; k < r1; k += G) {
        const float wt = s.in_w ? s.in_w[k] : a.w[s.in_arc[k]];
        const float sc = scores[s.in_srcpos[k]] + wt;
        if (MODE == SD_LOG) {
          mx = fmaxf(mx, sc);  // NaN-ignoring like the strict '>' scan
        } else {
          const int rank = MODE == SD_PATH ? (s.in_rank ? s.in_rank[k] : s.in_arc[k])
                                           : (tie_by_arc ? s.in_arc[k] : k);
          if (sc > mx || (sc == mx && sc > NEG_INF && rank < best_rank)) {
            mx = sc;
            best_rank = rank;
            best = MODE == SD_PATH ? k : s.in_arc[k];
          }
        }
      }
      if (MODE == SD_LOG) {
        mx = group_max<G>(mx);
      } else {
        if (!(mx > NEG_INF)) { best_rank = INT_MAX; best = -1; }
        group_argmax<G>(mx, best_rank, best);
      }
      if (is_start) {
        // the start node's virtual 0.0 in-score: listed last for the score
        // (shortest.cpp:129-135), first for the path relaxation (:200-206)
        if (MODE == SD_PATH) {
          if (!(mx > 0.0f)) { mx = 0.0f; best = -1; }
        } else if (0.0f > mx) {
          mx = 0.0f;
          best = -1;
        }
      }
      float out;
      const int cnt = (r1 - r0) + (is_start ? 1 : 0);
      if (MODE == SD_LOG) {
        float sum = 0.0f;
        if (cnt > 0 && mx != POS_INF && mx != NEG_INF) {
          for (int k = r0 + sub; k < r1; k += G) {
            const float wt = s.in_w ? s.in_w[k] : a.w[s.in_arc[k]];
            sum += expf(scores[s.in_srcpos[k]] + wt - mx);
          }
          sum = group_sum<G>(sum);
          if (is_start) sum += expf(0.0f - mx);
        }
        out = finish_lse(mx, sum, cnt);
      } else {
        // tropical: the max; path mode keeps -inf for unreachable nodes (:196)
        out = (cnt == 0) ? NEG_INF : mx;
      }
      if (MODE != SD_PATH && (s.pflags[p] & NF_ORPHAN)) out = 0.0f;  // shortest.cpp:89 zero-init
      if (sub == 0) {
        scores[p] = out;
        if (MODE != SD_LOG) a.argmax[p] = best;
      }
    }
    __syncthreads();
  }

  // ---- accept reduction (shortest.cpp:148-159 / :226-237)
  __shared__ float sh_v[kBlock];
  __shared__ int sh_k[kBlock];
  float mx = NEG_INF;
  int bestk = INT_MAX;
  for (int k = tid; k < s.n_accept; k += kBlock) {
    const float v = scores[s.acc_pos[k]];
    if (v > mx) {
      mx = v;
      bestk = k;
    }
  }
  sh_v[tid] = mx;
  sh_k[tid] = bestk;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) {
      const float v2 = sh_v[tid + o];
      const int k2 = sh_k[tid + o];
      if (v2 > sh_v[tid] || (v2 == sh_v[tid] && k2 < sh_k[tid])) {
        sh_v[tid] = v2;
        sh_k[tid] = k2;
      }
    }
    __syncthreads();
  }
  mx = sh_v[0];
  bestk = sh_k[0];
  __syncthreads();
  float sum = 0.0f;
  if (MODE == SD_LOG && s.n_accept > 0 && mx != POS_INF && mx != NEG_INF) {
    for (int k = tid; k < s.n_accept; k += kBlock) sum += expf(scores[s.acc_pos[k]] - mx);
  }
  sh_v[tid] = sum;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) sh_v[tid] += sh_v[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    float out = (MODE == SD_LOG) ? finish_lse(mx, sh_v[0], s.n_accept)
                                 : (s.n_accept == 0 ? NEG_INF : mx);
    SdResult r;
    r.score = out;
    r.max_final = mx;
    r.argmax_final = (bestk == INT_MAX || !(mx > NEG_INF)) ? -1 : s.acc_pos[bestk];
    r.pad = 0;
    *a.result = r;
    if (a.out_score) *a.out_score = out;
  }
}

// --------------------------------------------------------------------------
// backward sweep (pull form over the transposed rows; no atomics):
//   nodeGrad[u] = acceptTerm(u) + sum over out-arcs a = (u -> v) of g_a
//   g_a (log)      = nodeGrad[v] * exp(score[u] + w_a - score[v])
//   g_a (tropical) = nodeGrad[v] if a is v's arg-max in-arc else 0
// which is shortest.cpp:62-80 regrouped by source node.
// --------------------------------------------------------------------------
template <int MODE, int G>
__global__ __launch_bounds__(kBlock) void sd_backward_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  const DSched s = a.s;
  const int tid = threadIdx.x;
  const int sub = tid % G, grp = tid / G;
  constexpr int NGRP = kBlock / G;
  const SdResult res = *a.result;
  const float delta = *a.delta;
  const float* __restrict__ scores = a.scores;
  float* __restrict__ ng = a.node_grad;
  const float denom = (MODE == SD_LOG) ? expf(res.score - res.max_final) : 0.0f;

  for (int l = s.L - 1; l >= 0; --l) {
    const int lo = s.level_off[l], hi = s.level_off[l + 1];
    for (int p = lo + grp; p < hi; p += NGRP) {
      const int r0 = s.out_off[p], r1 = s.out_off[p + 1];
      const float su = scores[p];
      float acc = 0.0f;
      for (int k = r0 + sub; k < r1; k += G) {
        const int v = s.out_dstpos[k];
        const int arc = s.out_arc ? s.out_arc[k] : k;
        float g;
        if (MODE == SD_LOG) {
          g = ng[v] * expf(su + a.w[arc] - scores[v]);
        } else {
          g = (a.argmax[v] == arc) ? ng[v] : 0.0f;
        }
        a.arc_grad[arc] = g * delta;
        acc += g;
      }
      acc = group_sum<G>(acc);
      if (sub == 0) {
        if (s.pflags[p] & NF_ACCEPT) {
          // shortest.cpp:49-60
          acc += (MODE == SD_LOG) ? expf(su - res.max_final) / denom
                                  : (p == res.argmax_final ? 1.0f : 0.0f);
        }
        ng[p] = acc;
      }
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------
// viterbiPath pointer chase (shortest.cpp:239-260); one lane per graph
// --------------------------------------------------------------------------
__global__ void path_chase_kernel(const PathArgs* __restrict__ args, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PathArgs a = args[i];
  int cur = a.result->argmax_final;
  const int has_node = cur != -1;
  int len = 0;
  // first pass: length
  for (int c = cur; c != -1 && a.argmax[c] != -1 && len < a.cap;) {
    const int k = a.argmax[c];
    c = a.s.in_srcpos[k];
    ++len;
  }
  int pos = len;
  for (int c = cur; c != -1 && a.argmax[c] != -1 && pos > 0;) {
    const int k = a.argmax[c];
    const int arc = a.s.in_arc[k];
    --pos;
    a.path_arcs[pos] = arc;
    if (a.g.kind == KIND_LINEAR) {
      a.path_il[pos] = arc % a.g.C;
      a.path_ol[pos] = arc % a.g.C;
    } else {
      a.path_il[pos] = a.g.il[arc];
      a.path_ol[pos] = a.g.ol[arc];
    }
    a.path_w[pos] = a.g.w[arc];
    c = a.s.in_srcpos[k];
  }
  a.path_len[0] = len;
  a.path_len[1] = has_node;
}

template <int MODE>
void launch_fwd_mode(const SdArgs* d, int n, int g, hipStream_t st) {
  if (g >= 64)
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 64>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 8)
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 8>), dim3(n), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 1>), dim3(n), dim3(kBlock), 0, st, d);
}
template <int MODE>
void launch_bwd_mode(const SdArgs* d, int n, int g, hipStream_t st) {
  if (g >= 64)
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 64>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 8)
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 8>), dim3(n), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 1>), dim3(n), dim3(kBlock), 0, st, d);
}

int pick_group(int avg_deg_x16) {
  if (avg_deg_x16 >= 24 * 16) return 64;
  if (avg_deg_x16 >= 4 * 16) return 8;
  return 1;
}

} // namespace

void launch_sd_forward(const SdArgs* d_args, int n, int mode, int /*max_level_width*/,
                       int avg_in_degree_x16, hipStream_t st) {
  if (n <= 0) return;
  const int g = pick_group(avg_in_degree_x16);
  if (mode == SD_LOG)
    launch_fwd_mode<SD_LOG>(d_args, n, g, st);
  else if (mode == SD_TROPICAL)
    launch_fwd_mode<SD_TROPICAL>(d_args, n, g, st);
  else
    launch_fwd_mode<SD_PATH>(d_args, n, g, st);
}

void launch_sd_backward(const SdArgs* d_args, int n, int mode, int avg_out_degree_x16, hipStream_t st) {
  if (n <= 0) return;
  const int g = pick_group(avg_out_degree_x16);
  if (mode == SD_LOG)
    launch_bwd_mode<SD_LOG>(d_args, n, g, st);
  else
    launch_bwd_mode<SD_TROPICAL>(d_args, n, g, st);
}

void launch_path_chase(const PathArgs* d_args, int n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(path_chase_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_args, n);
}

} // namespace gtnx
