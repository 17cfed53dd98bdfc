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

// G = 256: the whole workgroup reduces ONE node's row (rows of hundreds of arcs on a level of one or two nodes:
// benchmarks/functions.cpp makeLinear(1000, 1000)) -- wave reductions, then the four partial results through LDS.
// Every lane of the workgroup is in the same iteration of the node loop (one group), so the barriers are uniform.
template <>
__device__ __forceinline__ float group_max<256>(float v) {
  __shared__ float sh[4];
  v = group_max<64>(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return v;
}
template <>
__device__ __forceinline__ float group_sum<256>(float v) {
  __shared__ float sh[4];
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return v;
}
template <>
__device__ __forceinline__ void group_argmax<256>(float& v, int& rank, int& payload) {
  __shared__ float shv[4];
  __shared__ int shr[4], shp[4];
  group_argmax<64>(v, rank, payload);
  if ((threadIdx.x & 63) == 0) {
    shv[threadIdx.x >> 6] = v;
    shr[threadIdx.x >> 6] = rank;
    shp[threadIdx.x >> 6] = payload;
  }
  __syncthreads();
  v = shv[0];
  rank = shr[0];
  payload = shp[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (shv[w] > v || (shv[w] == v && shr[w] < rank)) {
      v = shv[w];
      rank = shr[w];
      payload = shp[w];
    }
  __syncthreads();
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
// DEEP, THIN DAGs (log semiring): thousands of dependency levels of one or two nodes each --
// benchmarks/functions.cpp's makeRandomDAG(20000, 200000) has a node per level.  The generic kernels above pay
// four dependent global round trips and a workgroup barrier per level (1.6 us each: 32 ms for that graph, the
// reference on one core: 5.1 ms).  Here ONE WAVE walks the positions in order -- position order is level order,
// so everything a node reads is finished, and a wave's LDS traffic is in order: no barrier at all -- with
//   * the whole score vector in LDS (<= kDeepP positions), so the only value a node waits for is an LDS read;
//   * the rows (source position, weight) staged through LDS a chunk of up to 64 nodes / kDeepStage arcs at a
//     time: those loads do not depend on scores, so a chunk's worth is issued at once and their latency is paid
//     once per chunk instead of once per level.
// The backward twin walks the positions downwards with the node gradients in LDS and (destination position,
// weight, destination score) staged.
// --------------------------------------------------------------------------
constexpr int kDeepP = 24576;
constexpr int kDeepStage = 2048;
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// wave64 reductions by DPP row shifts / row broadcasts (VALU speed: a __shfl_xor butterfly is six LDS-crossbar
// permutes, several hundred cycles per node here); every lane must be active, the result is returned to all
#define GTNX_DPP_F(x, ctrl, rmask, old) \
  __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float deep_wave_sum(float x) {
  x += GTNX_DPP_F(x, 0x111, 0xf, 0.0f);
  x += GTNX_DPP_F(x, 0x112, 0xf, 0.0f);
  x += GTNX_DPP_F(x, 0x114, 0xf, 0.0f);
  x += GTNX_DPP_F(x, 0x118, 0xf, 0.0f);
  x += GTNX_DPP_F(x, 0x142, 0xa, 0.0f);
  x += GTNX_DPP_F(x, 0x143, 0xc, 0.0f);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float deep_wave_max(float x) {  // (lanes without a source keep their own value)
  x = fmaxf(x, GTNX_DPP_F(x, 0x111, 0xf, x));
  x = fmaxf(x, GTNX_DPP_F(x, 0x112, 0xf, x));
  x = fmaxf(x, GTNX_DPP_F(x, 0x114, 0xf, x));
  x = fmaxf(x, GTNX_DPP_F(x, 0x118, 0xf, x));
  x = fmaxf(x, GTNX_DPP_F(x, 0x142, 0xa, x));
  x = fmaxf(x, GTNX_DPP_F(x, 0x143, 0xc, x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
#undef GTNX_DPP_F

__global__ __launch_bounds__(64) void sd_forward_deep_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  const DSched s = a.s;
  const int lane = threadIdx.x;
  extern __shared__ float deep_lds[];
  float* lsc = deep_lds;                                            // [P] scores by position
  int* st_sp = reinterpret_cast<int*>(deep_lds + ((s.P + 3) & ~3));  // [kDeepStage] source positions
  float* st_w = reinterpret_cast<float*>(st_sp + kDeepStage);       // [kDeepStage] weights
  auto finish_node = [&](int p, int fl, int deg, float mx, float sum) {
    const bool is_start = (fl & NF_START) != 0;
    const int cnt = deg + (is_start ? 1 : 0);
    // shortest.cpp:102-114 with the hardware's exp / log (a node is a dependent chain here: latency is the cost)
    float out = cnt == 0 ? NEG_INF : ((mx == POS_INF || mx == NEG_INF) ? mx : mx + __logf(sum));
    if (fl & NF_ORPHAN) out = 0.0f;  // shortest.cpp:89 zero-init
    if (lane == 0) {
      lsc[p] = out;
      a.scores[p] = out;
    }
    wave_lds_fence();
  };
  for (int p0 = 0; p0 < s.P;) {
    const int pm = min(p0 + lane, s.P - 1);
    const int my_r0 = s.row_off[pm], my_r1 = s.row_off[pm + 1], my_fl = s.pflags[pm];
    const int base = __shfl(my_r0, 0);
    const bool fits = p0 + lane < s.P && my_r1 - base <= kDeepStage;
    const unsigned long long fm = __ballot(fits);
    const int nc = fm == ~0ull ? 64 : __ffsll((long long)~fm) - 1;  // leading nodes whose rows fit the stage together
    if (nc == 0) {
      // a row longer than the stage: straight from global memory, the wave over its arcs
      const int r0 = base, r1 = __shfl(my_r1, 0), fl = __shfl(my_fl, 0);
      const bool is_start = (fl & NF_START) != 0;
      float mx = NEG_INF;
      for (int k = r0 + lane; k < r1; k += 64)
        mx = fmaxf(mx, lsc[s.in_srcpos[k]] + (s.in_w ? s.in_w[k] : a.w[s.in_arc[k]]));
      mx = deep_wave_max(mx);
      if (is_start && 0.0f > mx) mx = 0.0f;
      float sum = 0.0f;
      if (mx != POS_INF && mx != NEG_INF) {
        for (int k = r0 + lane; k < r1; k += 64)
          sum += expf(lsc[s.in_srcpos[k]] + (s.in_w ? s.in_w[k] : a.w[s.in_arc[k]]) - mx);
        sum = deep_wave_sum(sum);
        if (is_start) sum += expf(0.0f - mx);
      }
      finish_node(p0, fl, r1 - r0, mx, sum);
      p0 += 1;
      continue;
    }
    const int narcs = __shfl(my_r1, nc - 1) - base;
    for (int e = lane; e < narcs; e += 64) {
      const int k = base + e;
      st_sp[e] = s.in_srcpos[k];
      st_w[e] = s.in_w ? s.in_w[k] : a.w[s.in_arc[k]];
    }
    wave_lds_fence();
    // the row of the NEXT node is fetched from the stage while this one is reduced (its addresses do not depend
    // on scores; only the score lookup does)
    int nx_sp = 0;
    float nx_w = 0.0f;
    {
      const int r0 = __builtin_amdgcn_readlane(my_r0, 0) - base, r1 = __builtin_amdgcn_readlane(my_r1, 0) - base;
      if (lane < r1 - r0) {
        nx_sp = st_sp[r0 + lane];
        nx_w = st_w[r0 + lane];
      }
    }
    for (int i = 0; i < nc; ++i) {
      const int r0 = __builtin_amdgcn_readlane(my_r0, i) - base, r1 = __builtin_amdgcn_readlane(my_r1, i) - base, fl = __builtin_amdgcn_readlane(my_fl, i);
      const bool is_start = (fl & NF_START) != 0;
      const int deg = r1 - r0;
      const int cu_sp = nx_sp;
      const float cu_w = nx_w;
      float mx = NEG_INF, sum = 0.0f;
      float sc = NEG_INF;
      if (deg <= 64 && lane < deg) sc = lsc[cu_sp] + cu_w;
      if (i + 1 < nc) {
        const int q0 = __builtin_amdgcn_readlane(my_r0, i + 1) - base, q1 = __builtin_amdgcn_readlane(my_r1, i + 1) - base;
        if (lane < q1 - q0) {
          nx_sp = st_sp[q0 + lane];
          nx_w = st_w[q0 + lane];
        }
      }
      if (deg <= 64) {  // the row in registers
        mx = deep_wave_max(sc);
        if (is_start && 0.0f > mx) mx = 0.0f;
        if (mx != POS_INF && mx != NEG_INF) {
          sum = deep_wave_sum(lane < deg ? __expf(sc - mx) : 0.0f);
          if (is_start) sum += __expf(0.0f - mx);
        }
      } else {
        for (int k = r0 + lane; k < r1; k += 64) mx = fmaxf(mx, lsc[st_sp[k]] + st_w[k]);
        mx = deep_wave_max(mx);
        if (is_start && 0.0f > mx) mx = 0.0f;
        if (mx != POS_INF && mx != NEG_INF) {
          for (int k = r0 + lane; k < r1; k += 64) sum += __expf(lsc[st_sp[k]] + st_w[k] - mx);
          sum = deep_wave_sum(sum);
          if (is_start) sum += __expf(0.0f - mx);
        }
      }
      finish_node(p0 + i, fl, deg, mx, sum);
    }
    p0 += nc;
  }
  // ---- accept reduction (shortest.cpp:148-159)
  float mx = NEG_INF;
  int bestk = INT_MAX;
  for (int k = lane; k < s.n_accept; k += 64) {
    const float v = lsc[s.acc_pos[k]];
    if (v > mx) {
      mx = v;
      bestk = k;
    }
  }
  if (!(mx > NEG_INF)) bestk = INT_MAX;
  {
    int rk = bestk, pl = bestk;
    group_argmax<64>(mx, rk, pl);
    bestk = pl;
  }
  float sum = 0.0f;
  if (s.n_accept > 0 && mx != POS_INF && mx != NEG_INF)
    for (int k = lane; k < s.n_accept; k += 64) sum += expf(lsc[s.acc_pos[k]] - mx);
  sum = deep_wave_sum(sum);
  if (lane == 0) {
    const float out = finish_lse(mx, sum, s.n_accept);
    SdResult r;
    r.score = out;
    r.max_final = mx;
    r.argmax_final = (bestk == INT_MAX || !(mx > NEG_INF)) ? -1 : s.acc_pos[bestk];
    r.pad = 0;
    *a.result = r;
    if (a.out_score) *a.out_score = out;
  }
}

__global__ __launch_bounds__(64) void sd_backward_deep_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  const DSched s = a.s;
  const int lane = threadIdx.x;
  extern __shared__ float deep_lds[];
  float* lng = deep_lds;                                            // [P] node gradients by position
  int* st_dp = reinterpret_cast<int*>(deep_lds + ((s.P + 3) & ~3));  // [kDeepStage] destination positions
  float* st_w = reinterpret_cast<float*>(st_dp + kDeepStage);       // [kDeepStage] weights
  float* st_sc = st_w + kDeepStage;                                 // [kDeepStage] forward scores of the destinations
  int* st_arc = reinterpret_cast<int*>(st_sc + kDeepStage);         // [kDeepStage] arc ids
  const SdResult res = *a.result;
  const float delta = *a.delta;
  const float denom = expf(res.score - res.max_final);
  auto finish_node = [&](int p, int fl, float su, float acc) {
    if (lane == 0) {
      if (fl & NF_ACCEPT) acc += expf(su - res.max_final) / denom;  // shortest.cpp:49-60
      lng[p] = acc;
      a.node_grad[p] = acc;
    }
    wave_lds_fence();
  };
  for (int hi = s.P; hi > 0;) {
    // lane j looks at node hi-1-j (descending positions)
    const int pm = max(hi - 1 - lane, 0);
    const int my_r0 = s.out_off[pm], my_r1 = s.out_off[pm + 1], my_fl = s.pflags[pm];
    const float my_su = a.scores[pm];
    const int top = __shfl(my_r1, 0);  // rows of descending nodes are contiguous downwards from here
    const bool fits = hi - 1 - lane >= 0 && top - my_r0 <= kDeepStage;
    const unsigned long long fm = __ballot(fits);
    const int nc = fm == ~0ull ? 64 : __ffsll((long long)~fm) - 1;
    if (nc == 0) {
      const int p = hi - 1, r0 = __shfl(my_r0, 0), r1 = top, fl = __shfl(my_fl, 0);
      const float su = __shfl(my_su, 0);
      float acc = 0.0f;
      for (int k = r0 + lane; k < r1; k += 64) {
        const int v = s.out_dstpos[k];
        const int arc = s.out_arc ? s.out_arc[k] : k;
        const float g = lng[v] * expf(su + a.w[arc] - a.scores[v]);
        a.arc_grad[arc] = g * delta;
        acc += g;
      }
      acc = deep_wave_sum(acc);
      finish_node(p, fl, su, acc);
      hi -= 1;
      continue;
    }
    const int lowest = __shfl(my_r0, nc - 1);
    const int narcs = top - lowest;
    for (int e = lane; e < narcs; e += 64) {
      const int k = lowest + e;
      const int v = s.out_dstpos[k];
      const int arc = s.out_arc ? s.out_arc[k] : k;
      st_dp[e] = v;
      st_arc[e] = arc;
      st_w[e] = a.w[arc];
      st_sc[e] = a.scores[v];
    }
    wave_lds_fence();
    for (int i = 0; i < nc; ++i) {
      const int r0 = __builtin_amdgcn_readlane(my_r0, i) - lowest, r1 = __builtin_amdgcn_readlane(my_r1, i) - lowest, fl = __builtin_amdgcn_readlane(my_fl, i);
      const float su = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_su), i));
      float acc = 0.0f;
      for (int k = r0 + lane; k < r1; k += 64) {
        const float g = lng[st_dp[k]] * expf(su + st_w[k] - st_sc[k]);
        a.arc_grad[st_arc[k]] = g * delta;
        acc += g;
      }
      acc = deep_wave_sum(acc);
      finish_node(hi - 1 - i, fl, su, acc);
    }
    hi -= nc;
  }
}

// --------------------------------------------------------------------------
// forward sweep, "narrow lattice" specialisation (log semiring).
//
// CTC-like products are DEEP and NARROW: T ~ 1000-2000 dependency levels of only
// ~180 nodes / ~450 arcs.  The generic kernel above pays ~4 dependent global
// round trips per level; with T levels that latency, not bandwidth, bounds it
// (measured 10% of HBM peak).  Here the per-level critical path touches LDS only:
//   * scores of the active frontier live in an LDS ring indexed by position;
//   * the CSR rows are streamed through LDS in CHUNKS of consecutive levels
//     (<= kCA arcs, <= kCN nodes), double buffered: while chunk c is reduced
//     out of LDS, chunk c+1's (src position, weight) pairs, row offsets and
//     node flags are in flight from HBM into registers (coalesced, issued a
//     whole chunk ahead) and land in the other LDS buffer at the chunk switch;
//   * inside a chunk there is NO global memory instruction at all, so the only
//     wait per level is lgkmcnt + one s_barrier; finished scores are flushed to
//     HBM once per chunk (for the backward pass), coalesced.
// One lane reduces one node (rows are short: mean in-degree 2.5).
// Eligibility (host): per-level arcs <= kCA, nodes <= kCN, reach <= kRing.
// HBM traffic is exactly the algorithmic 8A + 8N bytes.
// --------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// vmcnt (gfx950 counts loads and stores on one counter), which would stall every
// level on the staged loads of the NEXT chunk.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kRing = 4096;  // floats; power of two
constexpr int kCA = 2048;    // arcs per chunk (and per level)
constexpr int kCN = 1024;    // nodes per chunk (and per level)
constexpr int kTab = 256;    // levels per offset-table refill
constexpr int kJA = kCA / kBlock, kJN = kCN / kBlock;

// TROP: the tropical semiring (viterbiScore) on the same machinery -- max instead of
// log-sum-exp, the arg-max in-arc written per node (ties to the smallest arc id, the
// reference's in-list order for src-sorted products); needs HAS_INW and the arc ids of
// the in-row slots, staged alongside.
// PATH (with TROP): viterbiPath's relaxation instead -- the start node's virtual 0.0 is
// considered FIRST (strict '>', shortest.cpp:200-223), the back-pointer is the in-row
// SLOT (what path_chase_kernel follows), unreachable nodes keep -inf.
template <bool HAS_INW, bool TROP = false, bool PATH = false>
__global__ __launch_bounds__(kBlock) void sd_forward_narrow_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  DSched s = a.s;
  if (a.dyn_out) {  // sizes straight from the compose that built the lattice
    s.P = a.dyn_out->N;
    s.L = a.dyn_out->L;
    s.n_accept = a.dyn_counts[1];
  }
  const int tid = threadIdx.x;
  __shared__ float ring[kRing];
  // arc_sp holds BYTE offsets into `ring` ((position & (kRing-1)) * 4, applied once at
  // staging); both arc arrays are padded so a lane may read its 4 row slots unclamped
  __shared__ __attribute__((aligned(16))) int arc_sp[2][kCA + 4];
  __shared__ __attribute__((aligned(16))) float arc_w[2][kCA + 4];
  __shared__ __attribute__((aligned(16))) int arc_id[TROP ? 2 : 1][TROP ? kCA + 4 : 4];
  __shared__ __attribute__((aligned(16))) int node_off[2][kCN + kBlock];
  __shared__ __attribute__((aligned(16))) uint8_t node_fl[2][kCN];
  __shared__ int tab_node[kTab + 2];
  __shared__ int tab_arc[kTab + 2];
  auto ring_at = [&](int byte_off) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ring) + byte_off); };
  float* __restrict__ scores = a.scores;
  const int* __restrict__ in_srcpos = s.in_srcpos;
  const int* __restrict__ row_off = s.row_off;
  const uint8_t* __restrict__ pflags = s.pflags;
  const int last_node = s.P > 0 ? s.P - 1 : 0;

  // staging registers of the chunk in flight
  int st_sp[kJA];
  float st_w[kJA];
  int st_off[kJN + 1];
  int st_fl[kJN];

  // chunk = levels [c, e) of the current table window; ranges from the tables
  // a fixed number of levels per chunk, sized by the host from the widest level
  // (no per-switch scan of the offset tables)
  const int KL = max(a.chunk_levels, 1);
  auto chunk_end = [&](int c, int nl) { return min(c + KL, nl); };
  // Staging loads are unconditional and straight-line.  Row-ordered products
  // (HAS_INW: the arrays compose emits, padded by 16 B) are staged with 16-byte
  // vector loads -- 7 memory instructions per chunk instead of 25 -- and land in
  // LDS with ds_write_b128; host-built schedules use clamped scalar loads.
  gtnx_i4 v_sp[kJA / 4];
  gtnx_f4 v_w[kJA / 4];
  gtnx_i4 v_id[TROP ? kJA / 4 : 1];
  gtnx_i4 v_off;
  int v_off_last = 0;
  unsigned v_fl = 0;
  auto stage_load = [&](int a0, int a1, int n0) {
    if (HAS_INW) {
#pragma unroll
      for (int j = 0; j < kJA / 4; ++j) {
        const int k = a0 + 4 * tid + j * 4 * kBlock;
        v_sp[j] = *reinterpret_cast<const GTNX_G gtnx_i4*>(s.in_srcpos + k);
        v_w[j] = *reinterpret_cast<const GTNX_G gtnx_f4*>(s.in_w + k);
        if (TROP) v_id[j] = *reinterpret_cast<const GTNX_G gtnx_i4*>(s.in_arc + k);
      }
      const int nb = min(n0 + 4 * tid, last_node + 1);
      v_off = *reinterpret_cast<const GTNX_G gtnx_i4*>(s.row_off + nb);
      v_off_last = s.row_off[min(n0 + kCN, last_node + 1)];
      v_fl = *reinterpret_cast<const GTNX_G unsigned*>(s.pflags + min(n0 + 4 * tid, last_node));
      return;
    }
    const int ahi = max(a1 - 1, 0);
#pragma unroll
    for (int j = 0; j < kJA; ++j) {
      const int k = min(a0 + tid + j * kBlock, ahi);
      st_sp[j] = in_srcpos[k];
      st_w[j] = a.w[s.in_arc[k]];
    }
#pragma unroll
    for (int j = 0; j < kJN + 1; ++j) st_off[j] = row_off[min(n0 + tid + j * kBlock, last_node + 1)];
#pragma unroll
    for (int j = 0; j < kJN; ++j) st_fl[j] = pflags[min(n0 + tid + j * kBlock, last_node)];
  };
  auto stage_write = [&](int b) {
    if (HAS_INW) {
#pragma unroll
      for (int j = 0; j < kJA / 4; ++j) {
        *reinterpret_cast<gtnx_i4*>(&arc_sp[b][4 * tid + j * 4 * kBlock]) = (v_sp[j] & (kRing - 1)) << 2;
        *reinterpret_cast<gtnx_f4*>(&arc_w[b][4 * tid + j * 4 * kBlock]) = v_w[j];
        if (TROP) *reinterpret_cast<gtnx_i4*>(&arc_id[b][4 * tid + j * 4 * kBlock]) = v_id[j];
      }
      *reinterpret_cast<gtnx_i4*>(&node_off[b][4 * tid]) = v_off;
      if (tid == 0) node_off[b][kCN] = v_off_last;
      *reinterpret_cast<unsigned*>(&node_fl[b][4 * tid]) = v_fl;
      return;
    }
#pragma unroll
    for (int j = 0; j < kJA; ++j) {
      arc_sp[b][tid + j * kBlock] = (st_sp[j] & (kRing - 1)) << 2;
      arc_w[b][tid + j * kBlock] = st_w[j];
    }
#pragma unroll
    for (int j = 0; j < kJN + 1; ++j) node_off[b][tid + j * kBlock] = st_off[j];
#pragma unroll
    for (int j = 0; j < kJN; ++j) node_fl[b][tid + j * kBlock] = uint8_t(st_fl[j]);
  };

  for (int l0 = 0; l0 < s.L; l0 += kTab) {
    const int nl = min(kTab, s.L - l0);
    __syncthreads();
    for (int i = tid; i <= nl; i += kBlock) {
      const int n = s.level_off[l0 + i];
      tab_node[i] = n;
      tab_arc[i] = row_off[n];
    }
    __syncthreads();
    // prologue of the window: chunk 0 -> LDS buffer 0, chunk 1 -> registers
    int c = 0, e = chunk_end(0, nl), b = 0;
    stage_load(tab_arc[c], tab_arc[e], tab_node[c]);
    stage_write(0);
    int e2 = e < nl ? chunk_end(e, nl) : e;
    if (e < nl) stage_load(tab_arc[e], tab_arc[e2], tab_node[e]);
    __syncthreads();
    while (c < nl) {
      const int a0 = tab_arc[c], n0 = tab_node[c];
      // ---- reduce the chunk's levels out of LDS (no global memory traffic here).
      // Everything that does not depend on the scores (row bounds, flags, source
      // positions, weights) of level i+1 is read from LDS into registers BEFORE
      // level i's barrier, so a level's critical path is: ring gather -> max /
      // exp / log -> ring write -> barrier.
      int q_p = 0, q_r0 = 0, q_deg = 0, q_fl = 0, q_nhi = 0;
      int q_sp[4], q_id[4];
      float q_w[4];
      const int n_end = tab_node[e];  // chunk end (uniform)
      int next_lo = n0;               // level i's first node == level i-1's end: no table read on the chain
      auto preload = [&](int i) {
        q_nhi = tab_node[i + 1];      // only needed for the bounds test, off the address chain
        q_p = next_lo + tid;
        next_lo = q_nhi;
        const int pc = min(q_p, n_end - 1) - n0;
        q_r0 = node_off[b][pc] - a0;
        q_deg = node_off[b][pc + 1] - a0 - q_r0;
        q_fl = node_fl[b][pc];
        const int* spb = &arc_sp[b][q_r0];
        const float* wb = &arc_w[b][q_r0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q_sp[j] = spb[j];
          q_w[j] = wb[j];
          q_id[j] = TROP ? arc_id[TROP ? b : 0][q_r0 + j] : 0;
        }
      };
      auto node_update = [&](int p, int r0, int deg, int fl, const int* sp4, const float* w4, const int* id4) {
        if (TROP) {
          // max over the in-arcs, first by score then by smallest arc id; the start
          // node's virtual 0.0 comes last (shortest.cpp:118-135)
          float mx = NEG_INF;
          int best = -1, best_rank = INT_MAX;
          // `slot`: global in-row slot of the candidate (PATH's back-pointer)
          auto take = [&](float x, int id, int slot) {
            if (x > mx || (x == mx && x > NEG_INF && id < best_rank)) {
              mx = x;
              best_rank = id;
              best = PATH ? slot : id;
            }
          };
          const int slot0 = a0 + r0;  // global slot of the row's first in-arc
          if (deg <= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < deg) take(ring_at(sp4[j]) + w4[j], id4[j], slot0 + j);
          } else {
            for (int k = r0; k < r0 + deg; ++k)
              take(ring_at(arc_sp[b][k]) + arc_w[b][k], arc_id[TROP ? b : 0][k], a0 + k);
          }
          if (!(mx > NEG_INF)) best = -1;
          const bool is_start = (fl & NF_START) != 0;
          if (is_start) {
            if (PATH) {
              if (!(mx > 0.0f)) { mx = 0.0f; best = -1; }
            } else if (0.0f > mx) {
              mx = 0.0f;
              best = -1;
            }
          }
          float out = (deg + (is_start ? 1 : 0) == 0) ? NEG_INF : mx;
          if (!PATH && (fl & NF_ORPHAN)) out = 0.0f;
          ring[p & (kRing - 1)] = out;
          a.argmax[p] = best;
          return;
        }
        // the slot being overwritten belongs to position p - kRing, which no
        // later level reads (reach <= kRing); same-level lanes read other slots
        if (fl == 0 && deg <= 4) {
          // the common node (no start / orphan flag, at most 4 in-arcs), branch-free.
          // max + log(sum of exp(. - max)): sum >= 1, so v_log_f32 needs no denormal
          // scaling and is as accurate in absolute terms as the reference's
          // log1p(sum - 1); a node without in-arcs (or with a +-inf max) keeps max
          float v[4];
          float mx = NEG_INF;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = ring_at(sp4[j]) + w4[j];
            v[j] = j < deg ? x : NEG_INF;
            mx = fmaxf(mx, v[j]);
          }
          const bool fin = fabsf(mx) != POS_INF;
          const float m2 = fin ? mx : 0.0f;
          float sum = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) sum += __expf(v[j] - m2);
          ring[p & (kRing - 1)] = fin ? mx + 0.69314718f * __builtin_amdgcn_logf(sum) : mx;
          return;
        }
        const bool is_start = (fl & NF_START) != 0;
        float mx = NEG_INF, sum = 0.0f;
        const int r1 = r0 + deg;
        for (int k = r0; k < r1; ++k) mx = fmaxf(mx, ring_at(arc_sp[b][k]) + arc_w[b][k]);
        if (is_start && 0.0f > mx) mx = 0.0f;
        if (mx != POS_INF && mx != NEG_INF) {
          for (int k = r0; k < r1; ++k) sum += __expf(ring_at(arc_sp[b][k]) + arc_w[b][k] - mx);
          if (is_start) sum += __expf(0.0f - mx);
        }
        const int cnt = deg + (is_start ? 1 : 0);
        float out = (cnt == 0) ? NEG_INF : ((mx == POS_INF || mx == NEG_INF) ? mx : mx + __logf(sum));
        if (fl & NF_ORPHAN) out = 0.0f;
        ring[p & (kRing - 1)] = out;
      };
      preload(c);
      for (int i = c; i < e; ++i) {
        const int nhi = q_nhi;
        if (q_p < nhi) node_update(q_p, q_r0, q_deg, q_fl, q_sp, q_w, q_id);
        for (int p = q_p + kBlock; p < nhi; p += kBlock) {  // levels wider than the workgroup
          const int r0 = node_off[b][p - n0] - a0, deg = node_off[b][p - n0 + 1] - a0 - r0;
          int sp4[4], id4[4];
          float w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sp4[j] = arc_sp[b][r0 + j];
            w4[j] = arc_w[b][r0 + j];
            id4[j] = TROP ? arc_id[TROP ? b : 0][r0 + j] : 0;
          }
          node_update(p, r0, deg, node_fl[b][p - n0], sp4, w4, id4);
        }
        if (i + 1 < e) preload(i + 1);
        lds_barrier();
      }
      // ---- chunk switch: land the staged chunk, refill the registers, THEN flush
      // the finished scores -- the vmcnt(0) in front of stage_write() must only
      // cover loads issued a whole chunk ago, not stores issued just now
      const int n1 = tab_node[e];
      const int pc = c, pe = e;
      c = e;
      e = e2;
      b ^= 1;
      if (c < nl) {
        stage_write(b);
        e2 = e < nl ? chunk_end(e, nl) : e;
        if (e < nl) stage_load(tab_arc[e], tab_arc[e2], tab_node[e]);
      }
      (void)pc; (void)pe;
      for (int p = n0 + tid; p < n1; p += kBlock) scores[p] = ring[p & (kRing - 1)];
      lds_barrier();
    }
  }
  __syncthreads();  // scores[] stores visible before the accept reduction reads them

  // ---- accept reduction (identical to the generic kernel)
  __shared__ float sh_v[kBlock];
  __shared__ int sh_k[kBlock];
  float mx = NEG_INF;
  int bestk = INT_MAX;
  for (int k = tid; k < s.n_accept; k += kBlock) {
    const float v = scores[s.acc_pos[k]];
    if (v > mx) { mx = v; bestk = k; }
  }
  sh_v[tid] = mx;
  sh_k[tid] = bestk;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) {
      const float v2 = sh_v[tid + o];
      const int k2 = sh_k[tid + o];
      if (v2 > sh_v[tid] || (v2 == sh_v[tid] && k2 < sh_k[tid])) { sh_v[tid] = v2; sh_k[tid] = k2; }
    }
    __syncthreads();
  }
  mx = sh_v[0];
  bestk = sh_k[0];
  __syncthreads();
  float sum = 0.0f;
  if (!TROP && s.n_accept > 0 && mx != POS_INF && mx != NEG_INF)
    for (int k = tid; k < s.n_accept; k += kBlock) sum += expf(scores[s.acc_pos[k]] - mx);
  sh_v[tid] = sum;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) sh_v[tid] += sh_v[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float out = TROP ? (s.n_accept == 0 ? NEG_INF : mx) : finish_lse(mx, sh_v[0], s.n_accept);
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
// backward sweep, narrow-lattice specialisation (log semiring, identity out rows:
// the layered products compose emits, where arc k is the k-th out entry).
// Mirror image of sd_forward_narrow_kernel: levels are walked last to first, the
// out rows (dst position, weight), row offsets, flags and the chunk's forward
// scores are staged through LDS in double-buffered chunks, node gradients and
// scores of the active frontier live in two LDS rings, and inside a chunk the only
// global instructions are the fire-and-forget arc-gradient stores.
// HBM traffic: 12A + 12N (dst, weight, arc-grad per arc; offset, score, flag per node).
//
// FUSE: the lattice is a layered product of an explicit graph with an implicit
// linear chain (compose.cpp:496-518 would scatter its arc gradients to the two
// inputs in a second pass).  Here the chunk's arc gradients are instead summed in
// two LDS windows as they are flushed -- one over the explicit input's arcs (kept
// for the whole lattice), one over the chain rows of the chunk's levels (arcs that
// leave level l use chain arcs [l*C, (l+1)*C) only) -- and each chain row is stored
// exactly once, so the scatter needs no global atomics and no extra kernel.  The
// gradInfo columns ride in registers alongside the staged chunk (+8A bytes read).
// --------------------------------------------------------------------------
// wave64 sum by DPP row shifts / row broadcasts (VALU speed; no LDS permutes);
// every lane must be active, the total is returned to all lanes
#define GTNX_DPP_ADD(x, ctrl, rmask) \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float x) {
  GTNX_DPP_ADD(x, 0x111, 0xf);  // row_shr:1
  GTNX_DPP_ADD(x, 0x112, 0xf);  // row_shr:2
  GTNX_DPP_ADD(x, 0x114, 0xf);  // row_shr:4
  GTNX_DPP_ADD(x, 0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row sum
  GTNX_DPP_ADD(x, 0x142, 0xa);  // row_bcast:15 into rows 1, 3
  GTNX_DPP_ADD(x, 0x143, 0xc);  // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// win[key] += val for the active lanes of a wave, called from uniform control flow.
// A histogram-like scatter with a few very hot keys (CTC: the blank emission takes
// ~45% of a level's arcs) serialises in the LDS atomic unit -- one replay per lane
// on the same address.  Up to three candidate keys (the first still-unserved
// lane's) that at least 8 lanes share are therefore summed across the wave and
// added once; the rest go through plain ds_add_f32.
__device__ __forceinline__ void lds_add_hot(float* win, int key, float val, bool act) {
  unsigned long long rem = __ballot(act);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (!rem) break;
    const int sl = __ffsll((long long)rem) - 1;
    const int hk = __builtin_amdgcn_readlane(key, sl);
    const bool mine = act && key == hk;
    const unsigned long long m = __ballot(mine);
    if (__popcll(m) >= 8) {
      const float s = wave_sum_dpp(mine ? val : 0.0f);
      if (int(threadIdx.x & 63) == sl) atomicAdd(&win[hk], s);
      act = act && !mine;
    }
    rem &= ~m;
  }
  if (act) atomicAdd(&win[key], val);
}

constexpr int kRingB = 2048;
constexpr int kWinF = 1024;  // explicit-input arcs (floats) -- fused scatter
constexpr int kWinC = 4096;  // most chain arcs of one chunk (chunk_levels * C); the window is dynamic LDS

template <bool FUSE>
__global__ __launch_bounds__(kBlock) void sd_backward_narrow_kernel(const SdArgs* __restrict__ args) {
  const SdArgs a = args[blockIdx.x];
  DSched s = a.s;
  if (a.dyn_out) {
    s.P = a.dyn_out->N;
    s.L = a.dyn_out->L;
    s.n_accept = a.dyn_counts[1];
  }
  const int tid = threadIdx.x;
  __shared__ float sc_ring[kRingB];
  __shared__ float ng_ring[kRingB];
  __shared__ __attribute__((aligned(16))) unsigned short arc_dst[2][kCA];  // ring slots of the dst positions
  __shared__ __attribute__((aligned(16))) float arc_w[2][kCA];
  __shared__ __attribute__((aligned(16))) int node_off[2][kCN + kBlock];
  __shared__ __attribute__((aligned(16))) uint8_t node_fl[2][kCN];
  __shared__ int tab_node[kTab + 2];
  __shared__ int tab_arc[kTab + 2];
  __shared__ __attribute__((aligned(16))) float g_buf[kCA];  // arc gradients of the chunk, flushed at the switch
  __shared__ float win_f[FUSE ? kWinF : 1];
  extern __shared__ float win_c[];  // FUSE: chunk_levels * chain_C floats (sized by the host)
  if (FUSE) {
    for (int x = tid; x < kWinF; x += kBlock) win_f[x] = 0.0f;
    for (int x = tid; x < max(a.chunk_levels, 1) * a.chain_C; x += kBlock) win_c[x] = 0.0f;
  }
  const int CC = a.chain_C;
  gtnx_i4 gf_cur[kJA / 4], gc_cur[kJA / 4], gf_nxt[kJA / 4], gc_nxt[kJA / 4];
  // Explicit-input sums are carried in registers: while the lattice is stationary
  // (levels are shifted copies, compose.hip) slot (j, q) of a chunk refers to the
  // SAME input arc chunk after chunk, so its sum only goes to LDS when the key changes.
  int f_key[kJA];
  float f_acc[kJA];
#pragma unroll
  for (int x = 0; x < kJA; ++x) { f_key[x] = -1; f_acc[x] = 0.0f; }
  const GTNX_G int* __restrict__ out_off = s.out_off;
  const int last_node = s.P > 0 ? s.P - 1 : 0;
  const SdResult res = *a.result;
  const float delta = *a.delta;
  const float denom = expf(res.score - res.max_final);


  // chunk = levels [c, e) of the current window, chosen downwards from e
  const int KL = max(a.chunk_levels, 1);
  auto chunk_begin = [&](int e) { return max(e - KL, 0); };
  // 16-byte vector staging (arrays are compose-emitted and padded)
  gtnx_i4 v_dst[kJA / 4];
  gtnx_f4 v_w[kJA / 4];
  gtnx_i4 v_off;
  gtnx_f4 v_sc;
  int v_off_last = 0;
  unsigned v_fl = 0;
  auto stage_load = [&](int a0, int a1, int n0, int n1) {
#pragma unroll
    for (int j = 0; j < kJA / 4; ++j) {
      const int k = a0 + 4 * tid + j * 4 * kBlock;
      v_dst[j] = *reinterpret_cast<const GTNX_G gtnx_i4*>(s.out_dstpos + k);
      v_w[j] = *reinterpret_cast<const GTNX_G gtnx_f4*>(a.w + k);
      if (FUSE) {
        gf_nxt[j] = *reinterpret_cast<const GTNX_G gtnx_i4*>(a.gi_fixed + k);
        gc_nxt[j] = *reinterpret_cast<const GTNX_G gtnx_i4*>(a.gi_chain + k);
      }
    }
    const int nb = min(n0 + 4 * tid, last_node + 1);
    v_off = *reinterpret_cast<const GTNX_G gtnx_i4*>(s.out_off + nb);
    v_off_last = s.out_off[min(n0 + kCN, last_node + 1)];
    const int pb = min(n0 + 4 * tid, last_node);
    v_fl = *reinterpret_cast<const GTNX_G unsigned*>(s.pflags + pb);
    v_sc = *reinterpret_cast<const GTNX_G gtnx_f4*>(a.scores + pb);
  };
  auto stage_write = [&](int b, int n0, int n1) {
#pragma unroll
    for (int j = 0; j < kJA / 4; ++j) {
      uint2 pk;
      pk.x = unsigned(v_dst[j].x & (kRingB - 1)) | (unsigned(v_dst[j].y & (kRingB - 1)) << 16);
      pk.y = unsigned(v_dst[j].z & (kRingB - 1)) | (unsigned(v_dst[j].w & (kRingB - 1)) << 16);
      *reinterpret_cast<uint2*>(&arc_dst[b][4 * tid + j * 4 * kBlock]) = pk;
      *reinterpret_cast<gtnx_f4*>(&arc_w[b][4 * tid + j * 4 * kBlock]) = v_w[j];
    }
    *reinterpret_cast<gtnx_i4*>(&node_off[b][4 * tid]) = v_off;
    if (tid == 0) node_off[b][kCN] = v_off_last;
    *reinterpret_cast<unsigned*>(&node_fl[b][4 * tid]) = v_fl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = n0 + 4 * tid + i;
      if (p < n1) sc_ring[p & (kRingB - 1)] = v_sc[i];
    }
  };

  const int nwin = (s.L + kTab - 1) / kTab;
  for (int wi = nwin - 1; wi >= 0; --wi) {
    const int l0 = wi * kTab;
    const int nl = min(kTab, s.L - l0);
    __syncthreads();
    for (int i = tid; i <= nl; i += kBlock) {
      const int n = s.level_off[l0 + i];
      tab_node[i] = n;
      tab_arc[i] = out_off[n];
    }
    __syncthreads();
    int e = nl, c = chunk_begin(nl), b = 0;
    stage_load(tab_arc[c], tab_arc[e], tab_node[c], tab_node[e]);
    stage_write(0, tab_node[c], tab_node[e]);
    if (FUSE) {
#pragma unroll
      for (int j = 0; j < kJA / 4; ++j) { gf_cur[j] = gf_nxt[j]; gc_cur[j] = gc_nxt[j]; }
    }
    int c2 = c > 0 ? chunk_begin(c) : 0;
    if (c > 0) stage_load(tab_arc[c2], tab_arc[c], tab_node[c2], tab_node[c]);
    __syncthreads();
    while (e > 0) {
      const int a0 = tab_arc[c], n0 = tab_node[c];
      for (int i = e - 1; i >= c; --i) {
        const int nlo = tab_node[i], nhi = tab_node[i + 1];
        for (int p = nlo + tid; p < nhi; p += kBlock) {
          const int r0 = node_off[b][p - n0] - a0;
          const int deg = node_off[b][p - n0 + 1] - a0 - r0;
          const int fl = node_fl[b][p - n0];
          const float su = sc_ring[p & (kRingB - 1)];
          float acc = 0.0f;
          if (deg <= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int k = min(r0 + j, kCA - 1);
              const int v = arc_dst[b][k];
              const float g = ng_ring[v] * __expf(su + arc_w[b][k] - sc_ring[v]);
              if (j < deg) {
                g_buf[k] = g * delta;
                acc += g;
              }
            }
          } else {
            for (int k = r0; k < r0 + deg; ++k) {
              const int v = arc_dst[b][k];
              const float g = ng_ring[v] * __expf(su + arc_w[b][k] - sc_ring[v]);
              g_buf[k] = g * delta;
              acc += g;
            }
          }
          if (fl & NF_ACCEPT) acc += expf(su - res.max_final) / denom;  // shortest.cpp:49-60
          ng_ring[p & (kRingB - 1)] = acc;
        }
        lds_barrier();
      }
      // ---- chunk switch (downwards): land the staged chunk and refill first, then
      // flush this chunk's arc gradients (stores stay behind the staged loads)
      const int a1 = tab_arc[e];
      const int cbase = (l0 + c) * CC, crows = (e - c) * CC;  // chain rows of the finished chunk
      e = c;
      c = c2;
      b ^= 1;
      if (e > 0) stage_write(b, tab_node[c], tab_node[e]);
      if (FUSE) {
        // the finished chunk's arc gradients into the two LDS windows (same lane ->
        // arc mapping as the staging loads, so the gradInfo is already in registers)
#pragma unroll
        for (int j = 0; j < kJA / 4; ++j) {
          const int kl = 4 * tid + j * 4 * kBlock;
          const gtnx_f4 gv = *reinterpret_cast<const gtnx_f4*>(&g_buf[kl]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool in = kl + q < a1 - a0;
            const int fi = gf_cur[j][q], ci = gc_cur[j][q] - cbase;
            if (a.grad_fixed) {
              const int key = (in && fi >= 0) ? fi : -1;
              const int x = j * 4 + q;
              if (key == f_key[x]) {
                f_acc[x] += gv[q];
              } else {
                if (f_key[x] >= 0) atomicAdd(&win_f[f_key[x]], f_acc[x]);
                f_key[x] = key;
                f_acc[x] = gv[q];
              }
            }
            if (a.grad_chain) lds_add_hot(win_c, ci, gv[q], in && ci >= 0);
          }
        }
#pragma unroll
        for (int j = 0; j < kJA / 4; ++j) { gf_cur[j] = gf_nxt[j]; gc_cur[j] = gc_nxt[j]; }
      }
      if (e > 0) {
        c2 = c > 0 ? chunk_begin(c) : 0;
        if (c > 0) stage_load(tab_arc[c2], tab_arc[c], tab_node[c2], tab_node[c]);
      }
      for (int k = a0 + tid; k < a1; k += kBlock) a.arc_grad[k] = g_buf[k - a0];
      lds_barrier();
      if (FUSE && a.grad_chain) {
        // (the last level has no out-arcs and no chain row)
        for (int x = tid; x < crows && cbase + x < a.chain_A; x += kBlock) {
          a.grad_chain[cbase + x] = a.chain_accumulate ? a.grad_chain[cbase + x] + win_c[x] : win_c[x];
          win_c[x] = 0.0f;
        }
        lds_barrier();
      }
    }
  }
  if (FUSE && a.grad_fixed) {
#pragma unroll
    for (int x = 0; x < kJA; ++x)
      if (f_key[x] >= 0) atomicAdd(&win_f[f_key[x]], f_acc[x]);
    lds_barrier();
    for (int x = tid; x < a.fixed_A; x += kBlock) a.grad_fixed[x] = win_f[x];
  }
}

// --------------------------------------------------------------------------
// viterbiPath pointer chase (shortest.cpp:239-260).
// Following back-pointers through HBM is one dependent trip to memory per step (two, through the in-row slot:
// argmax[c] -> in_srcpos[k]) -- 1000 steps of a CTC lattice were 5.7 ms for ANY batch (0.8 % of HBM, round 4's
// review).  The chase itself is serial, but what it reads is not: (1) path_pred_kernel turns the back-pointers into
// predecessor POSITIONS for every node at once (a gather, fully parallel); (2) positions are a topological order
// (the schedule's: predecessors come first), so the walk only ever moves DOWN -- a workgroup stages the window of
// `kChaseWin` positions below the current node into LDS with coalesced loads and one lane walks it there (an LDS
// round trip per step instead of an HBM one), window after window; (3) the arcs' ids, labels and weights are
// gathered by all lanes once the visited positions are known.  Layered lattices (compose products: ~180 positions
// per level) cross ~45 levels per window; a graph whose arcs jump further than a window degrades to one window per
// step, the old cost.
// --------------------------------------------------------------------------
constexpr int kChaseWin = 8192;  // positions per window (32 KB of LDS)
constexpr int kChaseBlock = 256;
__global__ void path_pred_kernel(const PathArgs* __restrict__ args) {
  const PathArgs a = args[blockIdx.y];
  const int P = a.s.P;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < P; c += gridDim.x * blockDim.x) {
    const int k = a.argmax[c];
    a.pred[c] = k >= 0 ? a.s.in_srcpos[k] : -1;
  }
}
__global__ __launch_bounds__(kChaseBlock) void path_chase_kernel(const PathArgs* __restrict__ args, int n) {
  if (int(blockIdx.x) >= n) return;
  const PathArgs a = args[blockIdx.x];
  __shared__ int win[kChaseWin];
  __shared__ int sh_cur, sh_len, sh_done;
  const int tid = threadIdx.x;
  const int first = a.result->argmax_final;
  if (tid == 0) {
    a.path_len[2] = 0;  // path_tie_kernel only ever sets it
    sh_cur = first;
    sh_len = 0;
    sh_done = first == -1 ? 1 : 0;
  }
  __syncthreads();
  while (!sh_done) {
    const int c0 = sh_cur, hi = c0 + 1, lo = max(0, hi - kChaseWin);
    // (every lane's loads of the window are in flight together: unconditional, clamped)
    int v[kChaseWin / kChaseBlock];
#pragma unroll
    for (int j = 0; j < kChaseWin / kChaseBlock; ++j) v[j] = a.pred[min(lo + tid + j * kChaseBlock, c0)];
#pragma unroll
    for (int j = 0; j < kChaseWin / kChaseBlock; ++j) win[tid + j * kChaseBlock] = v[j];
    __syncthreads();
    if (tid == 0) {
      int c = c0, len = sh_len, done = 0;
      for (;;) {
        if (len >= a.cap) { done = 1; break; }
        const int p = win[c - lo];
        if (p == -1) { done = 1; break; }  // (argmax[c] == -1: the path starts here)
        a.tmp[a.cap - 1 - len] = c;
        ++len;
        c = p;
        if (c < lo || c > c0) break;  // below the window: stage the next one
      }
      sh_cur = c;
      sh_len = len;
      sh_done = done;
    }
    __syncthreads();
  }
  const int len = sh_len;
  // the path, first arc first: ids, labels, weights (gathers, all lanes)
  for (int i = tid; i < len; i += kChaseBlock) {
    const int c = a.tmp[a.cap - len + i];
    const int k = a.argmax[c];
    const int arc = a.s.in_arc[k];
    a.path_arcs[i] = arc;
    if (a.g.kind == KIND_LINEAR) {
      a.path_il[i] = arc % a.g.C;
      a.path_ol[i] = arc % a.g.C;
    } else {
      a.path_il[i] = a.g.il[arc];
      a.path_ol[i] = a.g.ol[arc];
    }
    a.path_w[i] = a.g.w[arc];
    a.path_pos[i] = c;
  }
  if (tid == 0) {
    a.path_len[0] = len;
    a.path_len[1] = first != -1;
  }
}

// Exact ties on the path.  The reference's shortestPath relaxes a node's in-arcs in the order their sources
// leave its queue (shortest.cpp:212-227) and keeps the FIRST maximum; the sweeps above break ties by the
// schedule's rank, which for a product numbered by compose (positions = node ids) is not that order.  Scores
// are unaffected; the chosen arc can differ only where two finite candidates of a visited node are EQUAL.
// One lane per path arc recounts its node's candidates; a flagged graph is rerun on a schedule that replays
// the queue (ops.cpp: op_viterbi_path).
__global__ void path_tie_kernel(const PathArgs* __restrict__ args) {
  const PathArgs a = args[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.path_len[0]) return;
  const int c = a.path_pos[i];
  const float target = a.scores[c];
  if (!(target > NEG_INF)) return;
  int equal = 0;
  for (int k = a.s.row_off[c]; k < a.s.row_off[c + 1]; ++k) {
    const float wt = a.s.in_w ? a.s.in_w[k] : a.w[a.s.in_arc[k]];
    equal += (a.scores[a.s.in_srcpos[k]] + wt == target) ? 1 : 0;
  }
  if (equal > 1) a.path_len[2] = 1;
}

template <int MODE>
void launch_fwd_mode(const SdArgs* d, int n, int g, hipStream_t st) {
  if (g >= 256)
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 256>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 64)
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 64>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 8)
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 8>), dim3(n), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL((sd_forward_kernel<MODE, 1>), dim3(n), dim3(kBlock), 0, st, d);
}
template <int MODE>
void launch_bwd_mode(const SdArgs* d, int n, int g, hipStream_t st) {
  if (g >= 256)
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 256>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 64)
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 64>), dim3(n), dim3(kBlock), 0, st, d);
  else if (g >= 8)
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 8>), dim3(n), dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL((sd_backward_kernel<MODE, 1>), dim3(n), dim3(kBlock), 0, st, d);
}

int pick_group(int avg_deg_x16) {
  if (avg_deg_x16 >= 192 * 16) return 256;
  if (avg_deg_x16 >= 24 * 16) return 64;
  if (avg_deg_x16 >= 4 * 16) return 8;
  return 1;
}

} // namespace

int sd_narrow_ring() { return kRing; }
int sd_narrow_tmp_cap() { return kCA; }
int sd_narrow_node_cap() { return kCN; }

int sd_deep_node_cap() { return kDeepP; }
namespace {
size_t deep_lds_bytes(int maxP, bool backward) {
  return 4 * (size_t((maxP + 3) & ~3) + size_t(backward ? 4 : 2) * kDeepStage);
}
}  // namespace
void launch_sd_forward_deep(const SdArgs* d_args, int n, int maxP, hipStream_t st) {
  if (n <= 0) return;
  static std::atomic<uint64_t> done{0};
  if (gtnx_first_on_device first{done})
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sd_forward_deep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              int(deep_lds_bytes(kDeepP, false)));
  hipLaunchKernelGGL(sd_forward_deep_kernel, dim3(n), dim3(64), deep_lds_bytes(maxP, false), st, d_args);
}
void launch_sd_backward_deep(const SdArgs* d_args, int n, int maxP, hipStream_t st) {
  if (n <= 0) return;
  static std::atomic<uint64_t> done{0};
  if (gtnx_first_on_device first{done})
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sd_backward_deep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              int(deep_lds_bytes(kDeepP, true)));
  hipLaunchKernelGGL(sd_backward_deep_kernel, dim3(n), dim3(64), deep_lds_bytes(maxP, true), st, d_args);
}

void launch_sd_forward(const SdArgs* d_args, int n, int mode, int narrow,
                       int avg_in_degree_x16, hipStream_t st) {
  if (n <= 0) return;
  if (narrow && mode == SD_LOG) {
    if (narrow == 2)
      hipLaunchKernelGGL(sd_forward_narrow_kernel<true>, dim3(n), dim3(kBlock), 0, st, d_args);
    else
      hipLaunchKernelGGL(sd_forward_narrow_kernel<false>, dim3(n), dim3(kBlock), 0, st, d_args);
    return;
  }
  if (narrow == 2 && mode == SD_TROPICAL) {  // row-ordered weights + arc ids of the in-row slots
    hipLaunchKernelGGL((sd_forward_narrow_kernel<true, true>), dim3(n), dim3(kBlock), 0, st, d_args);
    return;
  }
  if (narrow == 2 && mode == SD_PATH) {
    hipLaunchKernelGGL((sd_forward_narrow_kernel<true, true, true>), dim3(n), dim3(kBlock), 0, st, d_args);
    return;
  }
  const int g = pick_group(avg_in_degree_x16);
  if (mode == SD_LOG)
    launch_fwd_mode<SD_LOG>(d_args, n, g, st);
  else if (mode == SD_TROPICAL)
    launch_fwd_mode<SD_TROPICAL>(d_args, n, g, st);
  else
    launch_fwd_mode<SD_PATH>(d_args, n, g, st);
}

int sd_narrow_ring_backward() { return kRingB; }
void sd_narrow_fuse_caps(int* cap_fixed, int* cap_chain) {
  *cap_fixed = kWinF;
  *cap_chain = kWinC;
}

void launch_sd_backward(const SdArgs* d_args, int n, int mode, int narrow, int avg_out_degree_x16, hipStream_t st,
                        int fuse_lds_bytes) {
  if (n <= 0) return;
  if (narrow && mode == SD_LOG) {
    if (narrow == 2) hipLaunchKernelGGL(sd_backward_narrow_kernel<true>, dim3(n), dim3(kBlock), size_t(fuse_lds_bytes), st, d_args);
    else hipLaunchKernelGGL(sd_backward_narrow_kernel<false>, dim3(n), dim3(kBlock), 0, st, d_args);
    return;
  }
  const int g = pick_group(avg_out_degree_x16);
  if (mode == SD_LOG)
    launch_bwd_mode<SD_LOG>(d_args, n, g, st);
  else
    launch_bwd_mode<SD_TROPICAL>(d_args, n, g, st);
}

void launch_path_chase(const PathArgs* d_args, int n, int max_cap, int max_P, hipStream_t st) {
  if (n <= 0) return;
  const unsigned pb = unsigned(std::min(std::max((max_P + 1023) / 1024, 1), 256));  // (4 positions per lane, grid-stride beyond)
  hipLaunchKernelGGL(path_pred_kernel, dim3(pb, unsigned(n)), dim3(256), 0, st, d_args);
  hipLaunchKernelGGL(path_chase_kernel, dim3(unsigned(n)), dim3(kChaseBlock), 0, st, d_args, n);
  hipLaunchKernelGGL(path_tie_kernel, dim3(unsigned((std::max(max_cap, 1) + 255) / 256), unsigned(n)), dim3(256), 0, st, d_args);
}

} // namespace gtnx
