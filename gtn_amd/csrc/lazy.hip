// lazy.hip -- shortest distance / path / gradients over a composition that is NEVER
// materialised: the product of an implicit linear chain (emissions, T x C) with an
// explicit epsilon-free graph G (N nodes, A arcs).
//
// Replaces, for products too large to build (dense ASG
// transitions, C^2 (T-1) + C = 262 M arcs = 7 GB per utterance), the sequence
//   compose (compose.cpp:377-522)  ->  shortestDistance / shortestPath
//   (shortest.cpp:86-272)  ->  their gradFuncs (shortest.cpp:33-82, compose.cpp:496-518).
// Every state of the product is a pair (t, n); an arc a: s -> d of G with matched
// label l < C yields (t, s) -> (t+1, d) with weight w[a] + em[t][l] for every t, so
//   alpha[t+1][d] = (+)_{a: s->d} alpha[t][s] + w[a] + em[t][l(a)]      (forward)
//   beta [t][s]   = (+)_{a: s->d} w[a] + em[t][l(a)] + beta[t+1][d]     (backward)
// with alpha[0] = 0 on G's start nodes, beta[T] = 0 on its accept nodes.  The trimmed
// product the reference would build has the same total score (dead states carry no
// weight to an accept state), the same best path, and the same gradients.
//
// Execution model: one launch per time step over the whole BATCH of utterances that
// share G.  A workgroup owns a tile of 16 nodes x 16 utterances; lanes 0..15 of a
// 16-lane DPP row are 16 utterances of ONE node, so G's arc records are fetched once
// per row (same address across the row) and reused 16 times, while the previous
// step's scores and the emission rows of the tile's utterances sit in LDS.  This is
// VALU-bound work (T*C^2 adds/max or exp per utterance), not an HBM stream.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr int kBlock = 256;   // element-wise / per-utterance kernels
constexpr int kTile = 512;    // tile kernels: DT rows of BT lanes
constexpr int DT = 32;  // nodes per tile
constexpr int BT = 16;  // utterances per tile (one DPP row)
constexpr int RC = 16;  // records of a row staged per chunk (LDS: two 512-thread workgroups per CU at C4)
constexpr float NEG_INF = -__builtin_inff();

__device__ __forceinline__ int rec_label(const gtnx_i4& r, int chain_first) { return chain_first ? r.x : r.y; }

__global__ void lazy_pack_kernel(LazyGroup g, gtnx_i4* lrec_in, gtnx_i4* lrec_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.g.A) return;
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const gtnx_i4 r = dir ? g.g.out_rec[k] : g.g.in_rec[k];
    const int lab = rec_label(r, g.chain_first);
    gtnx_i4 o;
    o.x = r.z;
    o.y = (lab >= 0 && lab < g.C) ? lab : -1;
    o.z = __float_as_int(g.g.w[r.w]);
    o.w = r.w;
    (dir ? lrec_out : lrec_in)[k] = o;
  }
}

__global__ void lazy_init_kernel(LazyGroup g, int which) {
  // which 0: alpha[0] from start flags; 1: beta[T] from accept flags
  const int64_t tot = int64_t(g.nb) * g.N;
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= tot) return;
  const int n = int(i % g.N);
  const uint8_t f = g.g.nflags[n];
  if (which == 0)
    g.alpha[i] = (f & NF_START) ? 0.0f : NEG_INF;
  else
    g.beta[int64_t(g.T) * tot + i] = (f & NF_ACCEPT) ? 0.0f : NEG_INF;
}

// one time step.  FWD: alpha[t] -> alpha[t+1] over in-rows; BWD: beta[t+1] -> beta[t]
// over out-rows.  MODE SD_LOG: streaming log-sum-exp; SD_TROPICAL (forward only):
// max with the FIRST maximal in-arc recorded as back-pointer.
template <int MODE, bool BWD>
__global__ __launch_bounds__(kTile) void lazy_step_kernel(LazyGroup g, int t) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int N = g.N, C = g.C, Np = g.Npad, Cp = g.Cpad;
  float* prev = lds;             // [BT][Np]
  float* emr = lds + BT * Np;    // [BT][Cp]
  __shared__ __attribute__((aligned(16))) gtnx_i4 rowbuf[DT][RC + 1];  // the tile's rows, RC records at a time
  const int b0 = blockIdx.y * BT;
  const int64_t plane = int64_t(g.nb) * N;
  const float* src_plane = BWD ? g.beta + int64_t(t + 1) * plane : g.alpha + int64_t(t) * plane;
  {
    // one wave-row of the block per utterance row: unit-stride copies, no divisions
    const int rb = tid >> 5, rl = tid & 31;  // 16 rows x 32 lanes
    const bool on = b0 + rb < g.nb;
    const float* sp = src_plane + int64_t(b0 + rb) * N;
    for (int n = rl; n < N; n += 32) prev[rb * Np + n] = on ? sp[n] : NEG_INF;
    const float* ep = on ? g.em[b0 + rb] + int64_t(t) * C : nullptr;
    for (int c = rl; c < C; c += 32) emr[rb * Cp + c] = on ? ep[c] : 0.0f;
  }
  __syncthreads();
  const int bl = tid & (BT - 1), dl = tid >> 4;
  const int node = blockIdx.x * DT + dl;
  const int b = b0 + bl;
  const bool live = node < N && b < g.nb;
  const GTNX_G int* off = BWD ? g.g.out_off : g.g.in_off;
  const gtnx_i4* __restrict__ rec = BWD ? g.lrec_out : g.lrec_in;
  const int nd = min(node, N - 1);
  const int k0 = node < N ? off[nd] : 0, k1 = node < N ? off[nd + 1] : 0;
  const float* pb = prev + bl * Np;
  const float* eb = emr + bl * Cp;
  float m = NEG_INF, s = 0.0f;
  int arg = -1;
  // The 16 lanes of a DPP row serve one node: they fetch the node's next RC records
  // TOGETHER (RC/16 each, coalesced), park them in LDS and then all walk them from
  // there (same-address LDS reads broadcast).  The fetch of chunk c+1 is issued
  // before chunk c is reduced, so its latency hides behind RC arcs of VALU work; a
  // row lives inside one wave, so only the wave's own LDS counter orders this.
  constexpr int PL = RC / BT;  // records per lane per chunk
  gtnx_i4 pre[PL];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int x = 0; x < PL; ++x) pre[x] = rec[min(kk + bl * PL + x, max(k1 - 1, 0))];
  };
  if (k0 < k1) fetch(k0);
  for (int k = k0; k < k1; k += RC) {
#pragma unroll
    for (int x = 0; x < PL; ++x) rowbuf[dl][bl * PL + x] = pre[x];
    if (k + RC < k1) fetch(k + RC);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int cnt = min(RC, k1 - k);
    constexpr int UR = 2;  // arcs whose LDS gathers are in flight together
    for (int u0 = 0; u0 < cnt; u0 += UR) {
      gtnx_i4 r[UR];
      float x[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) r[u] = rowbuf[dl][min(u0 + u, RC - 1)];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const bool ok = u0 + u < cnt && r[u].y >= 0;
        const int sx = ok ? r[u].x : 0, lx = ok ? r[u].y : 0;
        x[u] = pb[sx] + __int_as_float(r[u].z) + eb[lx];
        if (!ok) x[u] = NEG_INF;
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        if (MODE == SD_LOG) {
          // streaming log-sum-exp: one exp per arc; -inf terms leave (m, s) untouched
          if (x[u] > m) {
            s = (m == NEG_INF) ? 1.0f : s * __expf(m - x[u]) + 1.0f;
            m = x[u];
          } else if (x[u] != NEG_INF) {
            s += __expf(x[u] - m);
          }
        } else {
          if (x[u] > m) {
            m = x[u];
            arg = r[u].w;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next chunk lands
  }
  if (!live) return;
  float out;
  if (MODE == SD_LOG)
    out = (m == NEG_INF || m == -NEG_INF) ? m : m + __logf(s);
  else
    out = m;
  if (BWD) {
    g.beta[int64_t(t) * plane + int64_t(b) * N + node] = out;
  } else {
    g.alpha[int64_t(t + 1) * plane + int64_t(b) * N + node] = out;
    if (MODE == SD_TROPICAL) g.bp[int64_t(t + 1) * plane + int64_t(b) * N + node] = arg;
  }
}

// accept reduction (shortest.cpp:148-159): one workgroup per utterance
template <int MODE>
__global__ __launch_bounds__(kBlock) void lazy_final_kernel(LazyGroup g) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* last = g.alpha + int64_t(g.T) * g.nb * g.N + int64_t(b) * g.N;
  __shared__ float sh_v[kBlock];
  __shared__ int sh_k[kBlock];
  float mx = NEG_INF;
  int bestk = INT_MAX;
  for (int k = tid; k < g.g.n_accept; k += kBlock) {
    const float v = last[g.g.accept_list[k]];
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
  if (MODE == SD_TROPICAL) {
    if (tid == 0) {
      g.score[b] = mx;
      g.best[b] = (bestk == INT_MAX || mx == NEG_INF) ? -1 : g.g.accept_list[bestk];
    }
    return;
  }
  float sum = 0.0f;
  if (mx != NEG_INF && mx != -NEG_INF)
    for (int k = tid; k < g.g.n_accept; k += kBlock) sum += expf(last[g.g.accept_list[k]] - mx);
  sh_v[tid] = sum;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) sh_v[tid] += sh_v[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    g.score[b] = (g.g.n_accept == 0 || mx == NEG_INF || mx == -NEG_INF) ? mx : mx + logf(sh_v[0]);
    g.best[b] = -1;
  }
}

// back-pointer chase (shortest.cpp:239-260): one lane per utterance; writes the path
// first-arc-first as (fixed arc id, composed ilabel, composed olabel, weight)
__global__ void lazy_path_kernel(LazyGroup g, int* path_arc, int* path_il, int* path_ol, float* path_w,
                                 int* path_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= g.nb) return;
  const int64_t plane = int64_t(g.nb) * g.N;
  int node = g.best[b];
  if (node < 0) {  // no accepting path: the trimmed product is the empty graph
    path_len[b] = -1;
    return;
  }
  for (int t = g.T; t >= 1; --t) {
    const int arc = g.bp[int64_t(t) * plane + int64_t(b) * g.N + node];
    const int lab = g.chain_first ? g.g.il[arc] : g.g.ol[arc];
    const int64_t o = int64_t(b) * g.T + (t - 1);
    path_arc[o] = arc;
    path_il[o] = g.chain_first ? lab : g.g.il[arc];
    path_ol[o] = g.chain_first ? g.g.ol[arc] : lab;
    path_w[o] = g.g.w[arc] + g.em[b][int64_t(t - 1) * g.C + lab];
    node = g.g.src[arc];
  }
  path_len[b] = g.T;
}

// ---- gradients of the log-semiring score -------------------------------------------
// posterior of product arc (t, a):  p = exp(alpha[t][s] + w[a] + em[t][l] + beta[t+1][d] - Z)

// chain gradient when every node's in-arcs share one matched label (`node_label`,
// -1 for nodes without in-arcs): sum_a p = exp(alpha[t+1][d] + beta[t+1][d] - Z), no arc
// loop.  One workgroup per (t, utterance); labels shared by several nodes (a CTC
// blank) are summed in an LDS row, then the row is stored once.
__global__ __launch_bounds__(kBlock) void lazy_chain_grad_nodes_kernel(LazyGroup g, const int* __restrict__ node_label) {
  extern __shared__ float row[];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* out = g.grad_em[b];
  if (!out) return;
  for (int c = tid; c < g.C; c += kBlock) row[c] = 0.0f;
  __syncthreads();
  const int64_t o = int64_t(t + 1) * g.nb * g.N + int64_t(b) * g.N;
  // an utterance without any accepting path (score -inf) has an empty product: no gradient
  const float z0 = g.zt ? g.zt[int64_t(t) * g.nb + b] : g.score[b];
  const bool zfin = z0 != NEG_INF && z0 != -NEG_INF;
  const float z = zfin ? z0 : 0.0f, dl = zfin ? *g.delta[b] : 0.0f;
  for (int n = tid; n < g.N; n += kBlock) {
    const int lab = node_label[n];
    if (lab < 0) continue;
    const float x = g.alpha[o + n] + g.beta[o + n] - z;
    if (x != NEG_INF) atomicAdd(&row[lab], expf(x) * dl);
  }
  __syncthreads();
  for (int c = tid; c < g.C; c += kBlock) out[int64_t(t) * g.C + c] = row[c];
}

// The two kernels above and lazy_local_z_kernel in one pass (N <= 1024, so a wave holds a (t, utterance) pair's
// alpha + beta row in registers): the step's own normaliser z = log sum_n exp(alpha + beta), written for the arc
// gradients, then the posteriors binned by label in the wave's LDS row and the row stored once.  One wave per
// pair, four pairs per wave: alpha and beta are read once instead of twice, by 16 x fewer workgroups.
constexpr int ZG_PAIRS = 4;
// UNIQ: no two nodes share a label (ASG transitions): plain LDS stores -- float atomics go through the LDS
// one lane at a time (measured: 1.2 of this kernel's 1.56 ms at C4 were its 16 ds_add_f32 per pair)
template <bool UNIQ>
__global__ __launch_bounds__(kBlock) void lazy_z_chain_grad_kernel(LazyGroup g, const int* __restrict__ node_label, float* zt) {
  extern __shared__ float rows[];  // [4 waves][C]
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* row = rows + wv * g.C;
  const int64_t npairs = int64_t(g.T) * g.nb;
  int lab[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = lane + 64 * i;
    const int l0 = node_label[n < g.N ? n : g.N - 1];
    lab[i] = n < g.N ? l0 : -1;
  }
  for (int j = 0; j < ZG_PAIRS; ++j) {
    const int64_t p = (int64_t(blockIdx.x) * 4 + wv) * ZG_PAIRS + j;
    if (p >= npairs) return;  // (per wave: nothing below synchronises across waves)
    const int t = int(p / g.nb), b = int(p % g.nb);
    const int64_t o = int64_t(t + 1) * g.nb * g.N + int64_t(b) * g.N;
    float x[16];
    float m = NEG_INF;
    // (clamped, unconditional loads: a load under a lane mask is followed by a wait for everything, and the
    // row would arrive one trip to memory at a time)
    float av[16], bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = lane + 64 * i;
      const int nc = n < g.N ? n : g.N - 1;
      av[i] = g.alpha[o + nc];
      bv[i] = g.beta[o + nc];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = lane + 64 * i;
      x[i] = n < g.N ? av[i] + bv[i] : NEG_INF;
      m = fmaxf(m, x[i]);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    float sum = 0.0f;
    const bool mfin = m != NEG_INF && m != -NEG_INF;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (mfin && x[i] != NEG_INF) ? __expf(x[i] - m) : 0.0f;
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) sum += __shfl_xor(sum, k, 64);
    const float z0 = (m == NEG_INF || m == -NEG_INF) ? m : m + logf(sum);
    if (lane == 0) zt[p] = z0;
    float* out = g.grad_em[b];
    if (!out) continue;
    for (int c = lane; c < g.C; c += 64) row[c] = 0.0f;
    // an utterance without any accepting path (score -inf) has an empty product: no gradient
    const bool zfin = z0 != NEG_INF && z0 != -NEG_INF;
    const float z = zfin ? z0 : 0.0f, dl = zfin ? *g.delta[b] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (lab[i] < 0) continue;
      const float xv = x[i] - z;
      if (xv != NEG_INF) {
        if (UNIQ) row[lab[i]] = __expf(xv) * dl;
        else atomicAdd(&row[lab[i]], __expf(xv) * dl);
      }
    }
    // (one wave: its LDS operations complete in order, no barrier)
    for (int c = lane; c < g.C; c += 64) out[int64_t(t) * g.C + c] = row[c];
  }
}

// general chain gradient: per (t, utterance) loop over all arcs of G
__global__ __launch_bounds__(kBlock) void lazy_chain_grad_arcs_kernel(LazyGroup g) {
  extern __shared__ float row[];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* out = g.grad_em[b];
  if (!out) return;
  for (int c = tid; c < g.C; c += kBlock) row[c] = 0.0f;
  __syncthreads();
  const int64_t plane = int64_t(g.nb) * g.N;
  const float* al = g.alpha + int64_t(t) * plane + int64_t(b) * g.N;
  const float* be = g.beta + int64_t(t + 1) * plane + int64_t(b) * g.N;
  const float* em = g.em[b] + int64_t(t) * g.C;
  // an utterance without any accepting path (score -inf) has an empty product: no gradient
  const float z0 = g.zt ? g.zt[int64_t(t) * g.nb + b] : g.score[b];
  const bool zfin = z0 != NEG_INF && z0 != -NEG_INF;
  const float z = zfin ? z0 : 0.0f, dl = zfin ? *g.delta[b] : 0.0f;
  for (int a = tid; a < g.g.A; a += kBlock) {
    const int lab = g.chain_first ? g.g.il[a] : g.g.ol[a];
    if (lab < 0 || lab >= g.C) continue;
    const float x = al[g.g.src[a]] + g.g.w[a] + em[lab] + be[g.g.dst[a]] - z;
    if (x != NEG_INF) atomicAdd(&row[lab], expf(x) * dl);
  }
  __syncthreads();
  for (int c = tid; c < g.C; c += kBlock) out[int64_t(t) * g.C + c] = row[c];
}

// gradient of G's arcs: grad[a] = sum over (t, utterance) of p.  Tile = 16 nodes x 16
// utterances x a range of time steps; per step the 16 utterances of a node are summed
// with DPP row shifts and lane 15 of the row adds the sum into an LDS accumulator
// owned by (node, in-row slot); the accumulators go out with one global atomic each
// at the end of the tile's time range.
__device__ __forceinline__ float row16_sum(float x) {
#define GTNX_ROW_ADD(ctrl) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))
  GTNX_ROW_ADD(0x111);
  GTNX_ROW_ADD(0x112);
  GTNX_ROW_ADD(0x114);
  GTNX_ROW_ADD(0x118);
#undef GTNX_ROW_ADD
  return x;  // valid in lane 15 of each 16-lane row
}

__global__ __launch_bounds__(kTile) void lazy_fixed_grad_kernel(LazyGroup g, int t_per_block, int max_in_deg) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int N = g.N, C = g.C, Np = g.Npad, Cp = g.Cpad;
  float* al = lds;                  // [BT][Np]  alpha[t]
  float* emr = al + BT * Np;        // [BT][Cp]
  float* acc = emr + BT * Cp;       // [DT][max_in_deg]
  __shared__ __attribute__((aligned(16))) gtnx_i4 rowbuf[DT][RC + 1];
  const int b0 = blockIdx.y * BT;
  const int t0 = blockIdx.z * t_per_block, t1 = min(g.T, t0 + t_per_block);
  const int64_t plane = int64_t(g.nb) * N;
  const int bl = tid & (BT - 1), dl = tid >> 4;
  const int node = blockIdx.x * DT + dl;
  const int b = b0 + bl;
  const bool live = node < N && b < g.nb;
  for (int i = tid; i < DT * max_in_deg; i += kTile) acc[i] = 0.0f;
  const int nd = min(node, N - 1);
  const int k0 = g.g.in_off[nd], deg = node < N ? g.g.in_off[nd + 1] - k0 : 0;
  // loop bounds must be uniform across the wave for the DPP sums
  __shared__ int sh_deg;
  if (tid == 0) sh_deg = 0;
  __syncthreads();
  atomicMax(&sh_deg, deg);
  __syncthreads();
  const int loop_deg = sh_deg;
  const float dl0 = live ? *g.delta[b] : 0.0f;
  for (int t = t0; t < t1; ++t) {
    const float z0 = live ? (g.zt ? g.zt[int64_t(t) * g.nb + b] : g.score[b]) : 0.0f;
    const bool zfin = z0 != NEG_INF && z0 != -NEG_INF;
    const float z = zfin ? z0 : 0.0f;
    const float dlt = (live && zfin) ? dl0 : 0.0f;
    __syncthreads();
    const float* src_plane = g.alpha + int64_t(t) * plane;
    {
      const int rb = tid >> 5, rl = tid & 31;
      const bool on = b0 + rb < g.nb;
      const float* sp = src_plane + int64_t(b0 + rb) * N;
      for (int n = rl; n < N; n += 32) al[rb * Np + n] = on ? sp[n] : NEG_INF;
      const float* ep = on ? g.em[b0 + rb] + int64_t(t) * C : nullptr;
      for (int c = rl; c < C; c += 32) emr[rb * Cp + c] = on ? ep[c] : 0.0f;
    }
    __syncthreads();
    const float bd = live ? g.beta[int64_t(t + 1) * plane + int64_t(b) * N + node] : NEG_INF;
    constexpr int PL = RC / BT;
    gtnx_i4 pre[PL];
    auto fetch = [&](int jj) {
#pragma unroll
      for (int x = 0; x < PL; ++x) pre[x] = g.lrec_in[k0 + min(jj + bl * PL + x, max(deg - 1, 0))];
    };
    fetch(0);
    for (int j0 = 0; j0 < loop_deg; j0 += RC) {
#pragma unroll
      for (int x = 0; x < PL; ++x) rowbuf[dl][bl * PL + x] = pre[x];
      if (j0 + RC < loop_deg) fetch(j0 + RC);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int cnt = min(RC, loop_deg - j0);  // uniform: the DPP sums need every lane
      for (int u = 0; u < cnt; ++u) {
        const int j = j0 + u;
        const gtnx_i4 r = rowbuf[dl][u];
        float p = 0.0f;
        if (live && j < deg && r.y >= 0) {
          const float x = al[bl * Np + r.x] + __int_as_float(r.z) + emr[bl * Cp + r.y] + bd - z;
          if (x != NEG_INF) p = __expf(x) * dlt;
        }
        p = row16_sum(p);
        if (bl == BT - 1 && j < deg) acc[dl * max_in_deg + j] += p;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (node < N && bl == BT - 1) {
    for (int j = 0; j < deg; ++j) {
      const float v = acc[dl * max_in_deg + j];
      if (v != 0.0f) atomicAdd(g.grad_fixed + g.g.in_rec[k0 + j].w, v);
    }
  }
}

// gradient of the best path (shortest.cpp:262-269 + compose.cpp:496-518): delta of path
// arc t goes to chain arc (t, label) and to G's arc
__global__ void lazy_path_grad_kernel(LazyPathGrad a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.len) return;
  // the reference indexes its saved arcs last-arc-first against deltas first-arc-first
  // (SURVEY.md section 8 A9): delta[i] lands on the mirrored path position
  const int pos = a.len - 1 - i;
  const float d = a.delta[a.delta_stride ? i : 0];
  const int arc = a.path_arc[pos];
  const int lab = a.chain_first ? a.il[pos] : a.ol[pos];
  if (a.grad_chain) atomicAdd(a.grad_chain + int64_t(pos) * a.C + lab, d);
  if (a.grad_fixed) atomicAdd(a.grad_fixed + arc, d);
}


// =====================================================================================
// dense regime (log semiring).  When every node's in-arcs carry one label (so the
// emission term factors out of the sum) and G is nearly complete, a time step is
//   alpha[t+1][b][d] = em[t][b][lab d] + cmax[d] + m_b + log( sum_s A[b][s] * E[s][d] ),
//   A[b][s] = exp(alpha[t][b][s] - m_b),  m_b = max_s alpha[t][b][s],  E = exp(w - cmax[d]),
// i.e. ONE multiply-add per arc instead of a streaming log-sum-exp (~20 instructions):
// a [nb x N] * [N x N] product per step in fp32 FMAs, LDS-tiled (32 x 32 outputs per
// workgroup, 2 x 2 per lane).  Row maxima keep A in [0, 1]; S >= E[s*][d] > 0 whenever d
// is reachable from the row's best state, and a complete G makes every d reachable, which
// is why the regime asks for a nearly complete G.  Same for beta with E transposed, and
// the gradient of G's arcs is exp(w[a]) * R[src][dst] with R = sum over (t, b) of an
// outer product, again FMAs.  The tropical semiring does not factor; it keeps the
// record-walking kernel above.
// =====================================================================================
constexpr int TT = 32;      // tile edge (outputs) and K chunk
constexpr int kDense = 256;

__global__ void lazy_dense_cmax_kernel(LazyGroup g, float* cmax) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= g.N) return;
  float m = NEG_INF;
  for (int k = g.g.in_off[d]; k < g.g.in_off[d + 1]; ++k) {
    const gtnx_i4 r = g.lrec_in[k];
    if (r.y >= 0) m = fmaxf(m, __int_as_float(r.z));
  }
  cmax[d] = m;
}
__global__ void lazy_dense_fill_kernel(LazyGroup g, float* E, const float* cmax) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.g.A) return;
  const gtnx_i4 r = g.lrec_in[k];  // {src, label, weight, arc}; destination from the arc
  if (r.y < 0) return;
  const int d = g.g.dst[r.w];
  const float w = __int_as_float(r.z), c = cmax[d];
  // w - c is NaN when both are the same infinity: -inf/-inf is a column without any live
  // arc (stays 0), +inf/+inf is the arc that makes the column's score +inf (weight 1)
  float e;
  if (c == NEG_INF) return;
  else if (c == -NEG_INF) e = (w == -NEG_INF) ? 1.0f : 0.0f;
  else e = expf(w - c);
  if (e != 0.0f) atomicAdd(&E[int64_t(r.x) * g.N + d], e);
}

// one time step; BWD: contraction over destinations with E transposed
// BWD: `vin` (when non-null) holds the contraction input of this step, beta[t+1] + em[t] +
// cmax per destination, written by the previous launch's epilogue into `vout`
template <bool BWD>
__global__ __launch_bounds__(kDense) void lazy_dense_step_kernel(LazyGroup g, int t, const float* vin, float* vout) {
  __shared__ float As[TT][TT + 2];   // [k][b]
  __shared__ float Es[TT][TT + 2];   // [k][out]
  __shared__ float mrow[TT];
  const int tid = threadIdx.x;
  const int N = g.N, C = g.C;
  const int b0 = blockIdx.y * TT, o0 = blockIdx.x * TT;  // utterance tile, output-node tile
  const int64_t plane = int64_t(g.nb) * N;
  const float* prev = BWD ? g.beta + int64_t(t + 1) * plane : g.alpha + int64_t(t) * plane;
  // value entering the contraction for (utterance b, inner node k)
  auto inner = [&](int b, int k) -> float {
    if (b >= g.nb || k >= N) return NEG_INF;
    if (BWD && vin) return vin[int64_t(b) * N + k];
    const float v = prev[int64_t(b) * N + k];
    if (!BWD) return v;
    const int lab = g.nlab[k];
    if (lab < 0) return NEG_INF;
    return v + g.em[b][int64_t(t) * C + lab] + g.cmax[k];
  };
  // ---- row maxima of the tile's 32 utterances (8 lanes per row)
  {
    const int rb = tid >> 3, rl = tid & 7;
    float m = NEG_INF;
    for (int k = rl; k < N; k += 8) m = fmaxf(m, inner(b0 + rb, k));
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 8));
    if (rl == 0) {
      mrow[rb] = m;
      if (blockIdx.x == 0 && b0 + rb < g.nb) (BWD ? g.bmax : g.amax)[int64_t(t) * g.nb + b0 + rb] = m;
    }
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;  // outputs (2*ty .. +1 utterances) x (2*tx .. +1 nodes)
  float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  for (int k0 = 0; k0 < N; k0 += TT) {
    // stage: As[k][b] = exp(inner - m_b);  Es[k][o] = E[k][o] (FWD) | E[o][k] (BWD)
    for (int i = tid; i < TT * TT; i += kDense) {
      const int bb = i / TT, kk = i % TT;   // consecutive lanes walk k: unit stride in `prev`
      const float m = mrow[bb];
      const float v = inner(b0 + bb, k0 + kk);
      As[kk][bb] = (v == NEG_INF || m == NEG_INF) ? 0.0f : __expf(v - m);
      const int r = i / TT, c = i % TT;
      float e = 0.0f;
      if (!BWD) {  // row k0 + r, columns o0 + c
        if (k0 + r < N && o0 + c < N) e = g.E[int64_t(k0 + r) * N + o0 + c];
        Es[r][c] = e;
      } else {     // row o0 + r (output = source), columns k0 + c (inner = destination)
        if (o0 + r < N && k0 + c < N) e = g.E[int64_t(o0 + r) * N + k0 + c];
        Es[c][r] = e;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < TT; ++k) {
      const float a0 = As[k][2 * ty], a1 = As[k][2 * ty + 1];
      const float e0 = Es[k][2 * tx], e1 = Es[k][2 * tx + 1];
      acc[0][0] += a0 * e0;
      acc[0][1] += a0 * e1;
      acc[1][0] += a1 * e0;
      acc[1][1] += a1 * e1;
    }
    __syncthreads();
  }
  float* out_plane = BWD ? g.beta + int64_t(t) * plane : g.alpha + int64_t(t + 1) * plane;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int b = b0 + 2 * ty + i;
    if (b >= g.nb) continue;
    const float m = mrow[2 * ty + i];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = o0 + 2 * tx + j;
      if (o >= N) continue;
      float v = NEG_INF;
      if (acc[i][j] > 0.0f && m != NEG_INF) {
        v = __logf(acc[i][j]) + m;
        if (!BWD) {
          const int lab = g.nlab[o];
          v = lab < 0 ? NEG_INF : v + g.cmax[o] + g.em[b][int64_t(t) * C + lab];
        }
      }
      out_plane[int64_t(b) * N + o] = v;
      if (BWD && vout && t >= 1) {  // what step t-1 contracts over
        const int lab = g.nlab[o];
        vout[int64_t(b) * N + o] =
            (lab < 0 || v == NEG_INF) ? NEG_INF : v + g.em[b][int64_t(t - 1) * C + lab] + g.cmax[o];
      }
    }
  }
}


// =====================================================================================
// dense regime on the matrix cores.  The step above is a [nb x N] . [N x N] float32 product
// per time step; this form runs it as v_mfma_f32_32x32x2_f32 tiles (exact float32: the same
// fmaf chain as the VALU loop) with the exp / log epilogue fused:
//   * the contraction input arrives ALREADY exponentiated and TRANSPOSED, X[k][b] = exp(x[b][k] -
//     ref[b]) -- written by the previous step's epilogue -- so both MFMA operands are 128-byte
//     coalesced rows (A: X[k][b0 ..], B: E[k][o0 ..]; E zero-padded, and transposed once for beta);
//   * ref[b] is the row maximum ONE STEP BACK (complete when the launch starts), not of the row
//     itself: the largest operand is then exp(one step's growth), not 1 -- harmless in float32 as
//     long as a step moves a row's best score by less than ~80 nats, and no pass over the row is
//     needed before exponentiating;
//   * the score planes are stored RELATIVE to that reference (LazyGroup::rel): row t + 1 of alpha holds
//     alpha[t+1] - RA[t+1] with RA[t+1] = RA[t] + m[t], m[t] = the (relative) maximum of row t -- so a stored
//     value is one step's growth, O(10), not the running score (8.7 per step at C4: 8700 at T = 1000, where one
//     float32 ulp is 1e-3 and every posterior exp(alpha + beta - Z) carried that much relative error: measured
//     3.4e-4 on the emission gradients against float64, tests/test_lazy_gpu.py::test_c4_alphabet_pinned_to_the_
//     reference[1000]).  The next operand is then simply exp(stored value); beta likewise (row t relative to
//     RB[t] = RB[t+1] + mq[t], mq[t] = the maximum of q[t] = beta[t+1] + em[t] + cmax).  What the consumers
//     need of the references cancels or is one row maximum: posteriors are normalised per time step
//     (zt, itself relative), an arc posterior's constant is amax[t] (= RA[t+1] - RA[t]), and the total score is
//     the last row's log-sum-exp + the sum of the T row maxima, taken in float64 (lazy_mfma_score_kernel);
//   * the true row maxima (the gradient kernels balance their factors around them) leave the epilogues as
//     one partial per (row, column tile) -- a plain store -- and every consumer takes the maximum of a
//     row's partials itself (an atomic max per row and tile cost 2 of a step's 10 microseconds).
// One wave per 32 x 32 output tile; 17 x 16 tiles at C4 keep 272 of the chip's 1024 SIMDs'
// matrix cores busy for 257 MFMAs each.
// =====================================================================================
typedef float gtnx_f16v __attribute__((ext_vector_type(16)));
typedef float mf_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fkey(float x) {
  const int b = __float_as_int(x);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float funkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// maximum over each aligned group of 32 lanes, in every lane of the group
__device__ __forceinline__ float half32_max(float x) {
#define GTNX_ROR_MAX(ctrl) x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false)))
  GTNX_ROR_MAX(0x128);  // row_ror:8
  GTNX_ROR_MAX(0x124);  // row_ror:4
  GTNX_ROR_MAX(0x122);  // row_ror:2
  GTNX_ROR_MAX(0x121);  // row_ror:1
#undef GTNX_ROR_MAX
  return fmaxf(x, __shfl_xor(x, 16, 64));
}

// Slots.  Leading nodes without a matched in-arc (the start node of an ASG transitions graph) are dead
// from step 1 on in both sweeps: alpha is -inf there, and beta only ever meets that alpha.  The operand
// planes index nodes ROTATED by their number (g.rot): live nodes first, so that C4's 513 nodes are 16
// column tiles of 32, one per CU, not 17 (the odd tile would double the time of the CUs that get two).
__device__ __forceinline__ int mf_node(const LazyGroup& g, int slot) { return slot + g.rot < g.N ? slot + g.rot : slot + g.rot - g.N; }
__device__ __forceinline__ int mf_slot(const LazyGroup& g, int node) { return node >= g.rot ? node - g.rot : node + g.N - g.rot; }

// Layout of both MFMA operands: [k / 4][column][k % 4] -- a lane's 8-byte load brings two of the four
// consecutive k of ITS column (two MFMAs' worth), 32 lanes cover 512 contiguous bytes.
// X_0 = exp(alpha[0]) (ref 0), amax key of row 0; or, backward, the input of step T-1:
// q = beta[T] + em[T-1] + cmax, X = exp(q) (ref 0), bmax key of row T-1
template <bool BWD>
__global__ __launch_bounds__(256) void lazy_mfma_init_kernel(LazyGroup g) {
  const int b = blockIdx.x, N = g.N;
  float* X = g.xt[BWD ? (g.T & 1) : 0];
  float m = NEG_INF;
  for (int k = threadIdx.x; k < g.Kpad; k += blockDim.x) {
    float v = NEG_INF;
    if (k < N && b < g.nb) {
      const uint8_t f = g.g.nflags[k];
      if (!BWD) {
        v = (f & NF_START) ? 0.0f : NEG_INF;
      } else if ((f & NF_ACCEPT) && g.T > 0) {
        const int lab = g.nlab[k];
        v = lab < 0 ? NEG_INF : g.em[b][int64_t(g.T - 1) * g.C + lab] + g.cmax[k];
      }
    }
    const int j = k < N ? mf_slot(g, k) : k;
    X[(int64_t(j >> 2) * g.nbpad + b) * 4 + (j & 3)] = v == NEG_INF ? 0.0f : __expf(v);
    m = fmaxf(m, v);
  }
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0 && b < g.nb) {  // partial 0 of the row; the others stay -inf
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float* mp = BWD ? g.bmaxp : g.amaxp;
    if (!BWD) mp[int64_t(b) * g.ntp] = m;
    else if (g.T > 0) mp[(int64_t(g.T - 1) * g.nb + b) * g.ntp] = m;
  }
}

// one workgroup of EIGHT waves per 32 x 32 output tile: each wave runs an eighth of the contraction, the
// partial tiles are summed through LDS and each wave finishes four of the tile's rows.  (With the planes rotated
// past the dead start node there are exactly as many tiles as CUs at C4, so a workgroup may take its CU's whole
// register file: 9.85 -> 9.1 microseconds per step against four waves.)  A step is a chain of
// latencies, not of throughput (288 MFMAs per tile against several trips to memory that the kernel
// boundary has just flushed out of L2): the operands of a wave's k groups are requested in two
// alternating batches, the next batch before the current one is multiplied, and the epilogue's scalars
// before either.
constexpr int MF_WAVES = 8;
constexpr int MF_ROWS = 16 / MF_WAVES;  // accumulator registers (tile rows per half-wave) a wave finishes
constexpr int MF_BATCH = 3;  // k groups (two MFMAs each) per operand batch, two alternating: six batches are a wave's eighth of C4's 576 padded sources (batches of 9: 9.1 us per step, of 3: 8.3, of 1: 9.1)
template <bool BWD>
__global__ __launch_bounds__(MF_WAVES * 64) void lazy_mfma_step_kernel(LazyGroup g, int t) {
  __shared__ float part[MF_WAVES][16][64];
  __shared__ float tr[32][36];
  __shared__ float refp[32][8];  // row maxima, partially reduced (ntp <= 32: eight 16-byte pieces)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, lo = l & 31, hi = l >> 5;
  // Workgroups go to the 8 XCDs in turn, and every kernel boundary empties their L2s: XCD x takes a
  // contiguous run of tiles in row-major order (all column tiles of its two row tiles at C4), so it
  // re-fetches only its own rows of the input plane, not all of it
  const int Nl = g.N - g.rot;  // live slots
  const int ncol = (Nl + 31) >> 5, total = gridDim.x;
  const int xcd = blockIdx.x & 7, per = total >> 3, rem = total & 7;
  const int tile = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int o0 = (tile % ncol) * 32, b0 = (tile / ncol) * 32;
  const int nbp = g.nbpad, Np = g.Npad2;
  const int N = g.N, C = g.C, nb = g.nb;
  // FWD: step t turns alpha[t] (X in plane t & 1) into alpha[t+1] (plane (t+1) & 1)
  // BWD: step t turns q[t] (plane (t+1) & 1) into beta[t] and q[t-1] (plane t & 1)
  const gtnx_f4* X = reinterpret_cast<const gtnx_f4*>(g.xt[BWD ? ((t + 1) & 1) : (t & 1)]);
  float* Xn = g.xt[BWD ? (t & 1) : ((t + 1) & 1)];
  const gtnx_f4* Em = reinterpret_cast<const gtnx_f4*>(BWD ? g.ETp : g.Ep);
  float* Mp = BWD ? g.bmaxp : g.amaxp;  // [T + 1][nb][ntp] row maxima, one partial per column tile
  const int ntp = g.ntp;
  const int64_t plane = int64_t(nb) * N;
  const int o = o0 + lo;      // slot
  const bool ocol = o < Nl;
  const int on_ = mf_node(g, ocol ? o : 0);  // its node: labels, column maxima and the score planes are by node
  const int te = BWD ? t - 1 : t;
  // ---- this wave's rows after the reduction: registers MF_ROWS wv .. of the summed tile
  int rowi[MF_ROWS];
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {
    const int reg = MF_ROWS * wv + v;
    rowi[v] = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
  }
  // the epilogue's per-row scalars are requested first: pointers and row maxima now, the emission once its row
  // pointer is there (by then the first operand batch is in flight behind it)
  const int lab_ = g.nlab[on_];
  const float cm_ = g.cmax[on_];
  const int lab = ocol ? lab_ : -1;
  const float cm = ocol ? cm_ : NEG_INF;
  const float* erow[MF_ROWS];
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {
    const int b = b0 + rowi[v];
    erow[v] = g.em[b < nb ? b : 0];
  }
  // Row maxima of the tile's 32 rows ONE STEP BACK of the output (row t of alpha / of q: complete when this
  // launch starts) -- the reference the output is stored against and, with it, the one the next input is
  // exponentiated against: ntp partials per row, fetched 16 bytes per thread and combined through LDS on the way
  // to the barrier the partial tiles meet at anyway.
  // (requested here, parked in LDS only after the product: the request must not be waited for up front)
  const int Q = ntp >> 2;  // 16-byte pieces per row
  const int npieces = 32 * Q;  // <= 256: at most one per thread
  gtnx_f4 rv;
  int ri;
  {
    const int i0 = int(threadIdx.x);
    const int i = i0 < npieces ? i0 : i0 % npieces;
    ri = i0 < npieces ? i : -1;
    const int r = i / Q, q = i % Q;
    const int b = b0 + r;
    const int64_t row = int64_t(t) * nb + (b < nb ? b : 0);
    rv = reinterpret_cast<const gtnx_f4*>(Mp + row * ntp)[q];
  }
  // ---- this wave's share of the k groups (4 k each = two MFMAs)
  const int groups = g.Kpad >> 2;
  const int g_lo = (groups * wv) / MF_WAVES, g_hi = (groups * (wv + 1)) / MF_WAVES;
  // Operand layout [k / 4][column][k % 4]: the half-wave hi reads the 8 bytes holding k slots 2 hi, 2 hi + 1
  // of ITS column -- the first MFMA of a group contracts slots {0, 2}, the second {1, 3} (same split in
  // both operands, so the four slots are summed once each) -- no lane loads a value it does not multiply.
  // Two accumulators: consecutive MFMAs do not wait for each other's result.
  gtnx_f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  gtnx_f16v acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const mf_f2* ap = reinterpret_cast<const mf_f2*>(X) + (b0 + lo) * 2 + hi;
  const mf_f2* bp = reinterpret_cast<const mf_f2*>(Em) + (o0 + lo) * 2 + hi;
  mf_f2 a0[MF_BATCH], e0[MF_BATCH], a1[MF_BATCH], e1[MF_BATCH];
  // Everything below is straight-line on purpose: a conditional request (or a conditional load of the
  // emission) would leave the wait before the next multiply at a branch join, where the counter has to be
  // assumed zero.  Kpad is a multiple of 4 * MF_WAVES * 2 MF_BATCH (zero rows in both operands), so a wave
  // owns a whole, even number of batches.
  auto request = [&](mf_f2(&a)[MF_BATCH], mf_f2(&e)[MF_BATCH], int q0) {
#pragma unroll
    for (int u = 0; u < MF_BATCH; ++u) {
      a[u] = ap[int64_t(q0 + u) * nbp * 2];
      e[u] = bp[int64_t(q0 + u) * Np * 2];
    }
  };
  auto multiply = [&](const mf_f2(&a)[MF_BATCH], const mf_f2(&e)[MF_BATCH]) {
#pragma unroll
    for (int u = 0; u < MF_BATCH; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, e[u].x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, e[u].y, acc2, 0, 0, 0);
    }
  };
  request(a0, e0, g_lo);
  __builtin_amdgcn_sched_barrier(0);
  float emv[MF_ROWS];
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {  // always a valid address; the value is dropped where it does not apply
    // (the row pointer came out of memory: typed global here, or it becomes a flat load, which any later
    // wait on the vector counter would have to treat as zero)
    const float x = ((const GTNX_G float*)erow[v])[int64_t(te >= 0 ? te : 0) * C + (lab >= 0 ? lab : 0)];
    emv[v] = (lab >= 0 && te >= 0) ? x : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
  for (int q = g_lo;;) {  // an even number of batches per wave: the last multiply is outside
    request(a1, e1, q + MF_BATCH);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a0, e0);
    __builtin_amdgcn_sched_barrier(0);
    q += 2 * MF_BATCH;
    if (q >= g_hi) break;
    request(a0, e0, q);
    __builtin_amdgcn_sched_barrier(0);
    multiply(a1, e1);
    __builtin_amdgcn_sched_barrier(0);
  }
  multiply(a1, e1);
  acc += acc2;
#pragma unroll
  for (int v = 0; v < 16; ++v) part[wv][v][l] = acc[v];
  if (ri >= 0) refp[ri / Q][ri % Q] = fmaxf(fmaxf(rv.x, rv.y), fmaxf(rv.z, rv.w));
  __syncthreads();
  // ---- epilogue: registers MF_ROWS wv .. of the summed tile, column lo
  float* outp = BWD ? g.beta + int64_t(t) * plane : g.alpha + int64_t(t + 1) * plane;
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {
    const int reg = MF_ROWS * wv + v;
    const int i = rowi[v];
    const int b = b0 + i;
    const bool on = b < nb;
    float a = 0.0f;
#pragma unroll
    for (int p = 0; p < MF_WAVES; p += 2) a += part[p][reg][l] + part[p + 1][reg][l];
    float m_out = refp[i][0];
    for (int q = 1; q < (ntp >> 2); ++q) m_out = fmaxf(m_out, refp[i][q]);
    // alpha[t+1] - RA[t+1] / beta[t] - RB[t] (relative: see the head of this section); the next step's contraction
    // input is exp(nxt) as it stands
    float val = NEG_INF, nxt = NEG_INF;
    if (on && ocol && a > 0.0f && m_out != NEG_INF) {
      val = __logf(a) - m_out;
      if (!BWD) {
        val = lab < 0 ? NEG_INF : val + cm + emv[v];
        nxt = val;
      } else if (t >= 1 && lab >= 0) {
        nxt = val + emv[v] + cm;
      }
    }
    if (on && ocol) outp[int64_t(b) * N + on_] = val;
    // row maximum of the next input over this tile's 32 columns -> its key
    // (over the half-wave's 32 columns: rotations inside the two 16-lane DPP rows, then ONE cross-lane read --
    // five ds_bpermute per row were 2 of a step's 10 microseconds)
    const float rm = half32_max(nxt);
    if (lo == 0 && on && (!BWD || t >= 1)) Mp[(int64_t(BWD ? t - 1 : t + 1) * nb + b) * ntp + (o0 >> 5)] = rm;
    tr[i][lo] = nxt == NEG_INF ? 0.0f : __expf(nxt);
  }
  __syncthreads();
  // the next input in operand layout: k group (o0 / 4 + kg), column b0 + bl: one 16-byte store per thread
  if ((!BWD || t >= 1) && threadIdx.x < 256) {
    const int kg = threadIdx.x >> 5, bl = threadIdx.x & 31;  // 8 k groups x 32 columns
    const int k = o0 + 4 * kg;
    if (k < g.Kpad) {
      gtnx_f4 w4;
      w4.x = (k + 0 < Nl) ? tr[bl][4 * kg + 0] : 0.0f;
      w4.y = (k + 1 < Nl) ? tr[bl][4 * kg + 1] : 0.0f;
      w4.z = (k + 2 < Nl) ? tr[bl][4 * kg + 2] : 0.0f;
      w4.w = (k + 3 < Nl) ? tr[bl][4 * kg + 3] : 0.0f;
      reinterpret_cast<gtnx_f4*>(Xn)[int64_t(k >> 2) * nbp + b0 + bl] = w4;
    }
  }
  // the dead leading nodes' scores, by the first column tile of every row tile
  if (o0 == 0 && threadIdx.x < 32 && b0 + int(threadIdx.x) < nb)
    for (int n = 0; n < g.rot; ++n) outp[int64_t(b0 + threadIdx.x) * N + n] = NEG_INF;
}

// ------------------------------------------------------------------------------------------------------------
// The same T steps as ONE launch.  Utterance b at step t + 1 depends on utterance b at step t only, so no
// grid-wide barrier is needed: the 32 rows of a row tile are produced by the `ncol` workgroups of that row tile
// (one per column tile) and consumed by the same workgroups one step later -- they meet at a counter per row
// tile.  What a step hands over (the next operand tile, 4 KB, and its row maxima) is written with 8-byte
// agent-scope stores (write-through) and read with agent-scope loads (past the reader's L1), the valid hand-off
// form without fences (MI355X_MICROARCH.md, "inter-workgroup visibility": correct for any workgroup placement;
// same-XCD placement, which the tile order arranges when blocks go to the XCDs in turn, only makes it faster).
// E -- the operand every step re-fetched after its kernel boundary -- lives in registers for the whole pass
// (GPW k groups per wave: 36 registers at C4), and there is no launch per step.  Requires every workgroup to be
// resident (one per CU): launched cooperatively, which refuses a grid that is not; the waits are bounded anyway.
__device__ __forceinline__ mf_f2 ld_agent(const mf_f2* p) {
  const unsigned long long v =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mf_f2 r;
  r.x = __uint_as_float(unsigned(v));
  r.y = __uint_as_float(unsigned(v >> 32));
  return r;
}
__device__ __forceinline__ void st_agent(float* p, float x, float y) {
  const unsigned long long v = (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool BWD, int GPW>
__global__ __launch_bounds__(MF_WAVES * 64) void lazy_mfma_chain_kernel(LazyGroup g, int* sync) {
  __shared__ float part[MF_WAVES][16][64];
  __shared__ float tr[32][36];
  __shared__ float refp[32][2][8];
  __shared__ int sh_abort;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, lo = l & 31, hi = l >> 5;
  const int Nl = g.N - g.rot;
  const int ncol = (Nl + 31) >> 5, total = gridDim.x;
  const int xcd = blockIdx.x & 7, per = total >> 3, rem = total & 7;
  const int tile = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int ct = tile % ncol, rt = tile / ncol;
  const int o0 = ct * 32, b0 = rt * 32;
  const int nbp = g.nbpad, Np = g.Npad2;
  const int N = g.N, C = g.C, nb = g.nb, T = g.T;
  int* arrive = sync + rt;                 // arrivals of this row tile's workgroups, all steps
  int* abortp = sync + (total / ncol);     // set by a workgroup whose wait ran out
  const gtnx_f4* Em = reinterpret_cast<const gtnx_f4*>(BWD ? g.ETp : g.Ep);
  float* Mp = BWD ? g.bmaxp : g.amaxp;
  const int ntp = g.ntp;
  const int64_t plane = int64_t(nb) * N;
  const int o = o0 + lo;
  const bool ocol = o < Nl;
  const int on_ = mf_node(g, ocol ? o : 0);
  int rowi[MF_ROWS];
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {
    const int reg = MF_ROWS * wv + v;
    rowi[v] = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
  }
  const int lab_ = g.nlab[on_];
  const float cm_ = g.cmax[on_];
  const int lab = ocol ? lab_ : -1;
  const float cm = ocol ? cm_ : NEG_INF;
  const GTNX_G float* erow[MF_ROWS];
#pragma unroll
  for (int v = 0; v < MF_ROWS; ++v) {
    const int b = b0 + rowi[v];
    erow[v] = (const GTNX_G float*)g.em[b < nb ? b : 0];
  }
  const int Q = ntp >> 2;
  const int npieces = 32 * 2 * Q;
  // this wave's k groups and its share of E, kept for the whole pass
  const int g_lo = GPW * wv;
  const mf_f2* bp = reinterpret_cast<const mf_f2*>(Em) + (o0 + lo) * 2 + hi;
  mf_f2 ereg[GPW];
#pragma unroll
  for (int u = 0; u < GPW; ++u) ereg[u] = bp[int64_t(g_lo + u) * Np * 2];
  if (threadIdx.x == 0) sh_abort = 0;
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = BWD ? T - 1 - s : s;
    const gtnx_f4* X = reinterpret_cast<const gtnx_f4*>(g.xt[BWD ? ((t + 1) & 1) : (t & 1)]);
    float* Xn = g.xt[BWD ? (t & 1) : ((t + 1) & 1)];
    const int te = BWD ? t - 1 : t;
    // ---- nothing another workgroup wrote: this step's emissions (their trip overlaps the wait below)
    float emv[MF_ROWS];
#pragma unroll
    for (int v = 0; v < MF_ROWS; ++v) {
      const float x = erow[v][int64_t(te >= 0 ? te : 0) * C + (lab >= 0 ? lab : 0)];
      emv[v] = (lab >= 0 && te >= 0) ? x : 0.0f;
    }
    // ---- the row tile's workgroups have published step s - 1
    if (s > 0) {
      if (threadIdx.x == 0) {
        const int want = ncol * s;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              wall_clock64() - t0 > 300000000ll) {  // 3 s of the 100 MHz counter: somebody is not resident
            __hip_atomic_store(abortp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_abort = 1;
            break;
          }
        }
      }
      __syncthreads();
      if (sh_abort) return;
    }
    const int t_in = !BWD ? (t > 0 ? t - 1 : 0) : (t < T - 1 ? t + 1 : t);
    // row maxima (partials per column tile) of this step's own row (the other half of the fetch is unused since
    // the planes went relative)
    mf_f2 rv[2][2];
    int ri[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i0 = int(threadIdx.x) + u * MF_WAVES * 64;
      const int i = i0 < npieces ? i0 : int(threadIdx.x) % npieces;
      ri[u] = i0 < npieces ? i : -1;
      const int r = i / (2 * Q), j = (i / Q) & 1, q = i % Q;
      const int b = b0 + r;
      const int64_t row = int64_t(j ? t : t_in) * nb + (b < nb ? b : 0);
      const mf_f2* src = reinterpret_cast<const mf_f2*>(Mp + row * ntp) + 2 * q;
      rv[u][0] = ld_agent(src);
      rv[u][1] = ld_agent(src + 1);
    }
    // ---- the contraction: every operand of the wave requested at once, multiplied as they arrive
    const mf_f2* ap = reinterpret_cast<const mf_f2*>(X) + (b0 + lo) * 2 + hi;
    mf_f2 a[GPW];
#pragma unroll
    for (int u = 0; u < GPW; ++u) a[u] = ld_agent(ap + int64_t(g_lo + u) * nbp * 2);
    gtnx_f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    gtnx_f16v acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < GPW; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, ereg[u].x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, ereg[u].y, acc2, 0, 0, 0);
    }
    acc += acc2;
#pragma unroll
    for (int v = 0; v < 16; ++v) part[wv][v][l] = acc[v];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (ri[u] >= 0) {
        const int i = ri[u];
        refp[i / (2 * Q)][(i / Q) & 1][i % Q] = fmaxf(fmaxf(rv[u][0].x, rv[u][0].y), fmaxf(rv[u][1].x, rv[u][1].y));
      }
    }
    __syncthreads();
    // ---- epilogue (as lazy_mfma_step_kernel)
    float* outp = BWD ? g.beta + int64_t(t) * plane : g.alpha + int64_t(t + 1) * plane;
#pragma unroll
    for (int v = 0; v < MF_ROWS; ++v) {
      const int reg = MF_ROWS * wv + v;
      const int i = rowi[v];
      const int b = b0 + i;
      const bool on = b < nb;
      float av = 0.0f;
#pragma unroll
      for (int p = 0; p < MF_WAVES; p += 2) av += part[p][reg][l] + part[p + 1][reg][l];
      float m_out = refp[i][1][0];
      for (int q = 1; q < (ntp >> 2); ++q) m_out = fmaxf(m_out, refp[i][1][q]);
      float val = NEG_INF, nxt = NEG_INF;  // (relative planes, as in lazy_mfma_step_kernel)
      if (on && ocol && av > 0.0f && m_out != NEG_INF) {
        val = __logf(av) - m_out;
        if (!BWD) {
          val = lab < 0 ? NEG_INF : val + cm + emv[v];
          nxt = val;
        } else if (t >= 1 && lab >= 0) {
          nxt = val + emv[v] + cm;
        }
      }
      if (on && ocol) outp[int64_t(b) * N + on_] = val;
      const float rm = half32_max(nxt);
      if (lo == 0 && on && (!BWD || t >= 1))
        __hip_atomic_store(Mp + (int64_t(BWD ? t - 1 : t + 1) * nb + b) * ntp + (o0 >> 5), rm, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      tr[i][lo] = nxt == NEG_INF ? 0.0f : __expf(nxt);
    }
    __syncthreads();
    if ((!BWD || t >= 1) && threadIdx.x < 256) {
      const int kg = threadIdx.x >> 5, bl = threadIdx.x & 31;
      const int k = o0 + 4 * kg;
      if (k < g.Kpad) {
        const float w0 = (k + 0 < Nl) ? tr[bl][4 * kg + 0] : 0.0f, w1 = (k + 1 < Nl) ? tr[bl][4 * kg + 1] : 0.0f;
        const float w2 = (k + 2 < Nl) ? tr[bl][4 * kg + 2] : 0.0f, w3 = (k + 3 < Nl) ? tr[bl][4 * kg + 3] : 0.0f;
        float* dst = Xn + (int64_t(k >> 2) * nbp + b0 + bl) * 4;
        st_agent(dst, w0, w1);
        st_agent(dst + 2, w2, w3);
      }
    }
    // after the first forward step the slots past the last column tile (the dead start node's) must read 0 in the
    // plane step 0 took its input from: it is this step's output plane at s == 1 (lazy_mfma_dead_rows_kernel)
    if (!BWD && s == 1 && ct == ncol - 1) {
      const int k0 = ncol * 32;
      for (int i = threadIdx.x; i < ((g.Kpad - k0) >> 2) * 32; i += MF_WAVES * 64) {
        const int kg = i >> 5, bl = i & 31;
        float* dst = Xn + (int64_t((k0 >> 2) + kg) * nbp + b0 + bl) * 4;
        st_agent(dst, 0.0f, 0.0f);
        st_agent(dst + 2, 0.0f, 0.0f);
      }
    }
    if (o0 == 0 && threadIdx.x < 32 && b0 + int(threadIdx.x) < nb)
      for (int n = 0; n < g.rot; ++n) outp[int64_t(b0 + threadIdx.x) * N + n] = NEG_INF;
    // ---- publish: every store of this workgroup has left (stores count in vmcnt), then one arrival
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// after the first forward step: the dead slots of plane 0 still hold exp(alpha[0]) (a start node's 1)
__global__ void lazy_mfma_dead_rows_kernel(LazyGroup g, float* X, int j_lo) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n = int64_t(g.Kpad - j_lo) * g.nbpad;
  if (i >= n) return;
  const int j = j_lo + int(i / g.nbpad), b = int(i % g.nbpad);
  X[(int64_t(j >> 2) * g.nbpad + b) * 4 + (j & 3)] = 0.0f;
}

// keys -> floats (maxplus.hip builds its W through integer keys)
__global__ void lazy_mfma_keys_kernel(float* p, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = funkey(__float_as_int(p[i]));
}
// the row maxima of a pass, once it is through: the maximum of every row's partials
__global__ void lazy_mfma_rowmax_kernel(const float* __restrict__ mp, float* __restrict__ out, int64_t rows, int ntp) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  float m = NEG_INF;
  for (int q = 0; q < ntp; ++q) m = fmaxf(m, mp[i * ntp + q]);
  out[i] = m;
}
// relative planes: the total score = log-sum-exp over the accept nodes of the LAST row (lazy_final_kernel, relative
// to RA[T]) + RA[T], the sum of the T row maxima -- in float64 (a float32 running score of 8700 carries 1e-3)
__global__ __launch_bounds__(64) void lazy_mfma_score_kernel(LazyGroup g) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  bool dead = false;
  for (int t = lane; t < g.T; t += 64) {
    const float m = g.amax[int64_t(t) * g.nb + b];
    dead |= m == NEG_INF;
    s += double(m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    dead |= __shfl_xor(int(dead), o, 64) != 0;
  }
  if (lane == 0) {
    const float rel = g.score[b];
    g.score[b] = (dead || rel == NEG_INF || rel == -NEG_INF) ? (dead ? NEG_INF : rel) : float(double(rel) + s);
  }
}
// E zero-padded in operand layout [k / 4][column][k % 4], and its transpose
__global__ void lazy_mfma_pad_kernel(LazyGroup g, float* Ep, float* ETp) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(g.Kpad) * g.Npad2) return;
  const int k = int(i / g.Npad2), c = int(i % g.Npad2);
  const bool in = k < g.N && c < g.N;
  const int64_t at = (int64_t(k >> 2) * g.Npad2 + c) * 4 + (k & 3);
  const int kn = in ? mf_node(g, k) : 0, cn = in ? mf_node(g, c) : 0;  // slots -> nodes
  Ep[at] = in ? g.E[int64_t(kn) * g.N + cn] : 0.0f;
  ETp[at] = in ? g.E[int64_t(cn) * g.N + kn] : 0.0f;
}

// R[s][d] += sum over a slice of (t, b) pairs of A'[s] * Q'[d], with the pair's two
// factors balanced around c = amax + bmax - Z so that neither side over- or underflows:
//   A' = exp(alpha[t][b][s] - amax + c/2),  Q' = exp(em + beta[t+1][b][d] - bmax + c/2) * delta
__global__ __launch_bounds__(kDense) void lazy_dense_fixed_grad_kernel(LazyGroup g, int pairs_per_block) {
  __shared__ float As[TT][TT + 2];   // [k][s]
  __shared__ float Qs[TT][TT + 2];   // [k][d]
  const int tid = threadIdx.x;
  const int N = g.N, C = g.C;
  const int s0 = blockIdx.x * TT, d0 = blockIdx.y * TT;
  const int64_t plane = int64_t(g.nb) * N;
  const int64_t npairs = int64_t(g.T) * g.nb;
  const int64_t p0 = int64_t(blockIdx.z) * pairs_per_block, p1 = min(npairs, p0 + pairs_per_block);
  const int tx = tid & 15, ty = tid >> 4;
  float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  for (int64_t q0 = p0; q0 < p1; q0 += TT) {
    for (int i = tid; i < TT * TT; i += kDense) {
      const int kk = i / TT, c = i % TT;  // pair kk of the chunk, column c of the tile
      const int64_t p = q0 + kk;
      float av = 0.0f, qv = 0.0f;
      if (p < p1) {
        const int t = int(p / g.nb), b = int(p % g.nb);
        const float z = g.zt ? g.zt[int64_t(t) * g.nb + b] : g.score[b];
        const float am = g.amax[int64_t(t) * g.nb + b], bm = g.bmax[int64_t(t) * g.nb + b];
        if (z != NEG_INF && z != -NEG_INF && am != NEG_INF && bm != NEG_INF) {
          const float half = 0.5f * (am + bm - z);
          if (s0 + c < N) {
            const float al = g.alpha[int64_t(t) * plane + int64_t(b) * N + s0 + c];
            if (al != NEG_INF) av = __expf(al - am + half);
          }
          if (d0 + c < N) {
            const int lab = g.nlab[d0 + c];
            const float be = g.beta[int64_t(t + 1) * plane + int64_t(b) * N + d0 + c];
            if (lab >= 0 && be != NEG_INF)
              qv = __expf(g.em[b][int64_t(t) * C + lab] + be - bm + half) * (*g.delta[b]);
          }
        }
      }
      As[kk][c] = av;
      Qs[kk][c] = qv;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < TT; ++k) {
      const float a0 = As[k][2 * ty], a1 = As[k][2 * ty + 1];
      const float e0 = Qs[k][2 * tx], e1 = Qs[k][2 * tx + 1];
      acc[0][0] += a0 * e0;
      acc[0][1] += a0 * e1;
      acc[1][0] += a1 * e0;
      acc[1][1] += a1 * e1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = s0 + 2 * ty + i, d = d0 + 2 * tx + j;
      if (s < N && d < N && acc[i][j] != 0.0f) atomicAdd(&g.R[int64_t(s) * N + d], acc[i][j]);
    }
}

// ---- gradient of G's arcs on the matrix cores: R[s][d] += sum over (t, utterance) pairs of
// A'[pair][s] * Q'[pair][d] is a product with the PAIRS as the contraction index, and both factors
// lie in HBM pair-major (alpha[t][b][.], beta[t+1][b][.]): the MFMA operand rows are the planes' own
// rows, exponentiated on the way in (two v_exp_f32 per MFMA, under its 64 cycles).
// per-pair constants once: {half - amax, half - bmax, delta, valid}, half = (amax + bmax - Z) / 2
// (relative planes: half = (bmax - zt) / 2 -- see below)
__global__ void lazy_mfma_pairs_kernel(LazyGroup g, gtnx_f4* pc) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= int64_t(g.T) * g.nb) return;
  const int b = int(p % g.nb);
  const float z = g.zt ? g.zt[p] : g.score[b];
  const float am = g.amax[p], bm = g.bmax[p];
  gtnx_f4 c = {0.0f, 0.0f, 0.0f, 0.0f};
  if (z != NEG_INF && z != -NEG_INF && am != NEG_INF && bm != NEG_INF) {
    // relative planes (g.rel): alpha[t] is stored against RA[t], beta[t+1] against RB[t+1], z = zt[t] against
    // RA[t+1] + RB[t+1], and RA[t+1] - RA[t] = amax[t]: the pair's constant is -(z + am), balanced as before
    const float half = 0.5f * (g.rel ? bm - z : am + bm - z);
    c = gtnx_f4{half - am, half - bm, *g.delta[b], 1.0f};
  }
  pc[p] = c;
}

// One wave owns a 64 x 64 block of R (2 x 2 MFMA tiles: every exponentiated operand feeds two MFMAs, so a
// pair costs 6 loads + 4 v_exp_f32 per FOUR MFMAs) and an interleaved quarter of the workgroup's pairs; the
// four waves' partial blocks meet in LDS.  Blocks are taken in SLOT space (mf_slot: live nodes first):
// a destination block of dead nodes has nothing to add, a source block of dead nodes (an ASG start node)
// only the pairs of step 0 -- at C4 that leaves 8 x 8 full blocks instead of 9 x 9 ragged ones.
template <bool STRIDED>  // the emission rows are slices of one tensor: no row-pointer load in the loop
__global__ __launch_bounds__(256) void lazy_mfma_fixed_grad_kernel(LazyGroup g, const gtnx_f4* __restrict__ pc,
                                                                  int pairs_per_block) {
  __shared__ float part[4][32][64];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, lo = l & 31, hi = l >> 5;
  const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
  const int N = g.N, C = g.C, nb = g.nb, Nl = g.N - g.rot;
  const int64_t plane = int64_t(nb) * N;
  int64_t npairs = int64_t(g.T) * nb;
  if (d0 >= Nl) return;                 // destinations without a matched in-arc: no arc, no gradient
  if (s0 >= Nl) npairs = nb;            // sources that are dead from step 1 on: alpha is -inf beyond step 0
  const int64_t p0 = int64_t(blockIdx.z) * pairs_per_block, p1 = min(npairs, p0 + pairs_per_block);
  if (p0 >= p1) return;
  // this lane's two source nodes and two destination nodes (slots s0 + lo, s0 + 32 + lo; likewise d)
  int sn[2], dn[2], lab[2];
  bool sok[2], dok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ss = s0 + 32 * h + lo, dd = d0 + 32 * h + lo;
    sok[h] = ss < N;
    dok[h] = dd < N;
    sn[h] = mf_node(g, sok[h] ? ss : 0);
    dn[h] = mf_node(g, dok[h] ? dd : 0);
    const int lb = g.nlab[dn[h]];
    lab[h] = dok[h] ? lb : -1;
  }
  const int labc[2] = {lab[0] >= 0 ? lab[0] : 0, lab[1] >= 0 ? lab[1] : 0};
  gtnx_f16v acc00 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
  // wave wv takes the pairs p0 + 2 wv + hi, + 8, + 16, ...: one MFMA contracts two pairs.  (t, b) of a
  // lane's pair advance incrementally (no division in the loop); loads are unconditional (clamped) so
  // that two rounds' operands are in flight together
  int pp = int(p0) + 2 * wv + hi;
  int t = pp / nb, b = pp - t * nb;
  const int pend = int(p1), plast = int(p1) - 1;
  constexpr int UR = 2;
  const int step_t = 8 / nb, step_b = 8 % nb;  // eight pairs on
  struct Round {
    float al[UR][2], be[UR][2], ev[UR][2];
    gtnx_f4 c[UR];
  };
  // the operands of the NEXT round are requested before the current one is exponentiated and multiplied
  // (rounds past the end re-read the last pair with weight 0: no branch around a load)
  auto request = [&](Round& r) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const bool in = pp < pend;
      const int tc = in ? t : 0, bc = in ? b : 0;
      r.c[u] = pc[in ? pp : plast];
      if (!in) r.c[u].w = 0.0f;
      const float* ar = g.alpha + int64_t(tc) * plane + int64_t(bc) * N;
      const float* br = g.beta + int64_t(tc + 1) * plane + int64_t(bc) * N;
      const GTNX_G float* er =
          (STRIDED ? (const GTNX_G float*)g.em_base + int64_t(bc) * g.em_stride : (const GTNX_G float*)g.em[bc]) + int64_t(tc) * C;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        r.al[u][h] = ar[sn[h]];
        r.be[u][h] = br[dn[h]];
        r.ev[u][h] = er[labc[h]];
      }
      pp += 8;
      b += step_b;  // (selects, not a loop: a branch between requests would zero the counted waits)
      t += step_t;
      const bool wrap = b >= nb;
      b -= wrap ? nb : 0;
      t += wrap ? 1 : 0;
    }
  };
  auto multiply = [&](const Round& r) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const bool on = r.c[u].w != 0.0f;
      float a[2], q[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a[h] = (on && sok[h] && r.al[u][h] != NEG_INF) ? __expf(r.al[u][h] + r.c[u].x) : 0.0f;
        q[h] = (on && lab[h] >= 0 && r.be[u][h] != NEG_INF) ? __expf(r.ev[u][h] + r.be[u][h] + r.c[u].y) * r.c[u].z : 0.0f;
      }
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[1], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[0], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc11, 0, 0, 0);
    }
  };
  // rounds of this wave (uniform): its pairs start at p0 + 2 wv and advance by 8 UR per round
  const int first = int(p0) + 2 * wv;
  const int rounds = first < pend ? (pend - first + 8 * UR - 1) / (8 * UR) : 0;
  Round ra, rb;
  request(ra);
  for (int it = 0; it < rounds; it += 2) {
    request(rb);
    __builtin_amdgcn_sched_barrier(0);
    multiply(ra);
    __builtin_amdgcn_sched_barrier(0);
    request(ra);
    __builtin_amdgcn_sched_barrier(0);
    multiply(rb);  // (an odd count's last round is all weight 0)
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- the four waves' partial blocks, two tiles (32 registers) at a time through LDS; wave wv then adds
  // registers 8 wv .. 8 wv + 7 of the pair of tiles into R
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // half: source rows s0 + 32 half ..
    if (half) __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      part[wv][v][l] = half ? acc10[v] : acc00[v];
      part[wv][16 + v][l] = half ? acc11[v] : acc01[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int r = 8 * wv + v;            // 0..15: destination tile 0, 16..31: tile 1
      const int reg = r & 15, dh = r >> 4;
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hi;  // row of the tile
      const float a = (part[0][r][l] + part[1][r][l]) + (part[2][r][l] + part[3][r][l]);
      const int ss = s0 + 32 * half + i;
      if (ss < N && dok[dh] && a != 0.0f) atomicAdd(&g.R[int64_t(mf_node(g, ss)) * N + dn[dh]], a);
    }
  }
}
// The same contraction with WIDE operand loads, for graphs whose live slots come in whole groups of four (C4:
// 512).  A lane holds four consecutive source slots (one 16-byte load of alpha) and two consecutive destination
// slots (one 8-byte load of beta): row tile j of a wave's 128 x 64 block takes source slots s0 + 4 lo + j, column tile
// c destination slots d0 + 2 lo + c -- which rows of R a tile holds is free as long as the write-out agrees.  Five
// load instructions per EIGHT MFMAs instead of fourteen: the 4-byte version was bound by the number of its loads.
typedef float mf_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float mf_f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int FG_DEPTH = 6;
template <bool STRIDED>
__global__ __launch_bounds__(256) void lazy_mfma_fixed_grad4_kernel(LazyGroup g, const gtnx_f4* __restrict__ pc,
                                                                   int pairs_per_block, int sb_first) {
  __shared__ float part[4][32][64];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, lo = l & 31, hi = l >> 5;
  const int sb = blockIdx.x + sb_first, db = blockIdx.y, zb = blockIdx.z;  // (sb_first: the source blocks of a launch start there)
  const int s0 = sb * 128, d0 = db * 64;
  const int N = g.N, C = g.C, nb = g.nb, Nl = g.N - g.rot;
  const int64_t plane = int64_t(nb) * N;
  int64_t npairs = int64_t(g.T) * nb;
  if (d0 >= Nl) return;       // destinations without a matched in-arc: no arc, no gradient
  if (s0 >= Nl) npairs = nb;  // sources that are dead from step 1 on: alpha is -inf beyond step 0
  const int64_t p0 = int64_t(zb) * pairs_per_block, p1 = min(npairs, p0 + pairs_per_block);
  if (p0 >= p1) return;
  // this lane's four source slots s0 + 4 lo + j and two destination slots d0 + 2 lo + c.  Live groups are
  // whole (Nl is a multiple of 4) and map to consecutive nodes; a group at or past Nl holds the dead nodes
  // (slots Nl .. N-1 = nodes 0 ..), again consecutive, the rest of it past the end of the graph
  const int sg = s0 + 4 * lo, dg = d0 + 2 * lo;
  bool sok[4], dok[2];
  int lab[2], labc[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) sok[j] = sg + j < N;
  const int sn0 = mf_node(g, sok[0] ? sg : 0);  // node of the group's first slot: the four are sn0 .. sn0 + 3
  const int sn0c = min(sn0, N - 4);             // (the load stays inside the row; components are shifted back below)
  const int dn0 = mf_node(g, dg < N ? dg : 0), dn0c = min(dn0, N - 2);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    dok[c] = dg + c < N;
    const int lb = g.nlab[dok[c] ? dn0 + c : 0];
    lab[c] = dok[c] ? lb : -1;
    labc[c] = lab[c] >= 0 ? lab[c] : 0;
  }
  const int sshift = sn0 - sn0c, dshift = dn0 - dn0c;  // 0 except in the graph's last, partial group
  gtnx_f16v acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[j][c] = gtnx_f16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int pp = int(p0) + 2 * wv + hi;
  int t = pp / nb, b = pp - t * nb;
  const int pend = int(p1), plast = int(p1) - 1;
  const int step_t = 8 / nb, step_b = 8 % nb;  // eight pairs on
  struct Round {
    mf_f4u al;
    mf_f2u be;
    float ev[2];
    gtnx_f4 c;
  };
  auto request = [&](Round& r) {
    const bool in = pp < pend;
    const int tc = in ? t : 0, bc = in ? b : 0;
    r.c = pc[in ? pp : plast];
    if (!in) r.c.w = 0.0f;
    const float* ar = g.alpha + int64_t(tc) * plane + int64_t(bc) * N;
    const float* br = g.beta + int64_t(tc + 1) * plane + int64_t(bc) * N;
    const GTNX_G float* er =
        (STRIDED ? (const GTNX_G float*)g.em_base + int64_t(bc) * g.em_stride : (const GTNX_G float*)g.em[bc]) + int64_t(tc) * C;
    r.al = *reinterpret_cast<const mf_f4u*>(ar + sn0c);
    r.be = *reinterpret_cast<const mf_f2u*>(br + dn0c);
    r.ev[0] = er[labc[0]];
    r.ev[1] = er[labc[1]];
    pp += 8;
    b += step_b;  // (selects, not a loop: a branch between requests would zero the counted waits)
    t += step_t;
    const bool wrap = b >= nb;
    b -= wrap ? nb : 0;
    t += wrap ? 1 : 0;
  };
  auto multiply = [&](const Round& r) {
    const bool on = r.c.w != 0.0f;
    const float alv[4] = {r.al.x, r.al.y, r.al.z, r.al.w}, bev[2] = {r.be.x, r.be.y};
    float a[4], q[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = alv[(j + sshift) & 3];  // (sshift != 0 only where sok[j + ...] fails anyway, except the shifted ones)
      a[j] = (on && sok[j] && x != NEG_INF) ? __expf(x + r.c.x) : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float x = bev[(c + dshift) & 1];
      q[c] = (on && lab[c] >= 0 && x != NEG_INF) ? __expf(r.ev[c] + x + r.c.y) * r.c.z : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[j][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], q[c], acc[j][c], 0, 0, 0);
  };
  // A round's eight MFMAs are 512 cycles, a trip to memory ten times that, and the accumulators leave room for
  // two waves per SIMD only: FG_DEPTH - 1 rounds are in flight behind the one being multiplied (rounds past
  // the end are weight-0 re-reads of the last pair)
  const int first = int(p0) + 2 * wv;
  const int rounds = first < pend ? (pend - first + 7) / 8 : 0;
  Round rr[FG_DEPTH];
#pragma unroll
  for (int k = 0; k < FG_DEPTH - 1; ++k) request(rr[k]);
  for (int it = 0; it < rounds; it += FG_DEPTH) {
#pragma unroll
    for (int k = 0; k < FG_DEPTH; ++k) {
      request(rr[(k + FG_DEPTH - 1) % FG_DEPTH]);
      __builtin_amdgcn_sched_barrier(0);
      multiply(rr[k]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the four waves' partial blocks, one row tile (two tiles = 32 registers) at a time through LDS; wave wv
  // then adds registers 8 wv .. 8 wv + 7 of the pair into R
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j) __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      part[wv][v][l] = acc[j][0][v];
      part[wv][16 + v][l] = acc[j][1][v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int r = 8 * wv + v;  // 0..15: column tile 0, 16..31: tile 1
      const int reg = r & 15, c = r >> 4;
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hi;  // row of the tile = lane lo' of the A operand
      const float a = (part[0][r][l] + part[1][r][l]) + (part[2][r][l] + part[3][r][l]);
      const int ss = s0 + 4 * i + j;  // the slot row i of row tile j stands for
      if (ss < N && dok[c] && a != 0.0f) atomicAdd(&g.R[int64_t(mf_node(g, ss)) * N + dn0 + c], a);
    }
  }
}
// ---- The same contraction, one WORKGROUP per 256 x 256 block of R (round 6).  The per-wave kernels above read
// the planes once per 64 (or 128) destination / source slots: at C4 (512 x 512 live slots) alpha is fetched 8 times and
// beta 4 times -- 19.5 GB of HBM traffic for 3.1 GB of planes, and that, not the matrix cores, bounded them (4.6 ms).
// Here sixteen waves (4 x 4, each a 64 x 64 sub-block = 2 x 2 MFMA tiles, 64 accumulator registers) share the
// operands of a block through LDS: a pair's 256 source values and 256 destination values are loaded and
// exponentiated ONCE per workgroup (one 16-byte load of alpha and of beta per lane and 16 pairs), double-buffered
// against the 128 MFMAs a SIMD issues per 16 pairs; every plane is read N / 256 times (twice at C4).  The pairs are
// split over gridDim.z workgroups per block (split-K); partial blocks meet in R by atomics as before.
// Floor at C4: 2 T B N^2 = 0.27 TFLOP on v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense) = 1.71 ms.
// Takes graphs whose live slots come in whole blocks of 256 (Nl % 256 == 0); dead SOURCE slots (an ASG start node:
// alive at step 0 only) are left to lazy_mfma_fixed_grad4_kernel, launched for those blocks alone.
constexpr int FGW_KB = 16;  // pairs per staged block
template <bool STRIDED>
__global__ __launch_bounds__(1024) void lazy_mfma_fixed_grad_wg_kernel(LazyGroup g, const gtnx_f4* __restrict__ pc,
                                                                      int pairs_per_block, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float As[2][FGW_KB][256];
  __shared__ __attribute__((aligned(16))) float Qs[2][FGW_KB][256];
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, lo = l & 31, hi = l >> 5;
  const int wi = wv >> 2, wj = wv & 3;
  const int N = g.N, C = g.C, nb = g.nb, rot = g.rot;
  const int s0 = blockIdx.x * 256, d0 = blockIdx.y * 256;  // live slots: node = slot + rot
  const int64_t plane = int64_t(nb) * N;
  const int64_t npairs = int64_t(g.T) * nb;
  const int64_t p0 = int64_t(blockIdx.z) * pairs_per_block, p1 = min(npairs, p0 + pairs_per_block);
  if (p0 >= p1) return;
  // staging role: pair tid / 64 of the block, columns 4 (tid % 64) .. + 3 of the source AND destination tiles
  const int sp = tid >> 6, sc = (tid & 63) * 4;
  const int snode = s0 + rot + sc, dnode = d0 + rot + sc;
  int lab[4], labc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lab[k] = g.nlab[dnode + k];
    labc[k] = lab[k] >= 0 ? lab[k] : 0;
  }
  const bool lab_run = lab[0] >= 0 && lab[1] == lab[0] + 1 && lab[2] == lab[0] + 2 && lab[3] == lab[0] + 3;  // (one 16-byte load)
  int pp = int(p0) + sp;
  int t = pp / nb, b = pp - t * nb;
  const int pend = int(p1), plast = int(p1) - 1;
  const int step_t = FGW_KB / nb, step_b = FGW_KB % nb;
  struct Round {
    mf_f4u al, be, ev;
    gtnx_f4 c;
    bool in;  // (known when the request is made: nothing LOADED is looked at before the landing)
  };
  // exp(-inf) is 0: a dead node or an invalid pair is a select on the ARGUMENT, no branch
  auto land = [&](const Round& r, int buf) {
    const bool on = r.in && r.c.w != 0.0f;
    const float alv[4] = {r.al.x, r.al.y, r.al.z, r.al.w}, bev[4] = {r.be.x, r.be.y, r.be.z, r.be.w},
                evv[4] = {r.ev.x, r.ev.y, r.ev.z, r.ev.w};
    float a[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = __expf(on ? alv[k] + r.c.x : NEG_INF);
      q[k] = __expf((on && lab[k] >= 0) ? evv[k] + bev[k] + r.c.y : NEG_INF) * r.c.z;
    }
    *reinterpret_cast<gtnx_f4*>(&As[buf][sp][sc]) = gtnx_f4{a[0], a[1], a[2], a[3]};
    *reinterpret_cast<gtnx_f4*>(&Qs[buf][sp][sc]) = gtnx_f4{q[0], q[1], q[2], q[3]};
  };
  gtnx_f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = gtnx_f16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nblk = int((p1 - p0 + FGW_KB - 1) / FGW_KB);
  auto multiply = [&](int buf) {
    const float* Ab = &As[buf][0][64 * wi + lo];
    const float* Qb = &Qs[buf][0][64 * wj + lo];
#pragma unroll
    for (int kk = 0; kk < FGW_KB / 2; ++kk) {
      const float a0 = Ab[(2 * kk + hi) * 256], a1 = Ab[(2 * kk + hi) * 256 + 32];
      const float q0 = Qb[(2 * kk + hi) * 256], q1 = Qb[(2 * kk + hi) * 256 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q1, acc[1][1], 0, 0, 0);
    }
  };
  // One round in flight: block blk + 1 is requested, block blk multiplied out of LDS (128 MFMAs per SIMD, 3.4 us:
  // the loads arrive under them), block blk + 1 exponentiated into the other buffer -- the only wait for memory is
  // at that landing, where nothing younger is outstanding.  (A round past the end is a weight-0 re-read of the last
  // pair: no branch around a request.)  LABRUN: this lane's four labels are consecutive, one 16-byte load.
  auto sweep = [&](auto labrun_tag) {
    constexpr bool LABRUN = decltype(labrun_tag)::value;
    auto req = [&](Round& r) {
      const bool in = pp < pend;
      const int tc = in ? t : 0, bc = in ? b : 0;
      r.in = in;
      r.c = pc[in ? pp : plast];
      const float* ar = g.alpha + int64_t(tc) * plane + int64_t(bc) * N;
      const float* br = g.beta + int64_t(tc + 1) * plane + int64_t(bc) * N;
      const GTNX_G float* er =
          (STRIDED ? (const GTNX_G float*)g.em_base + int64_t(bc) * g.em_stride : (const GTNX_G float*)g.em[bc]) + int64_t(tc) * C;
      r.al = *reinterpret_cast<const mf_f4u*>(ar + snode);
      r.be = *reinterpret_cast<const mf_f4u*>(br + dnode);
      if constexpr (LABRUN) {
        r.ev = *reinterpret_cast<const GTNX_G mf_f4u*>(er + labc[0]);
      } else {
        r.ev.x = er[labc[0]];
        r.ev.y = er[labc[1]];
        r.ev.z = er[labc[2]];
        r.ev.w = er[labc[3]];
      }
      pp += FGW_KB;
      b += step_b;
      t += step_t;
      const bool wrap = b >= nb;
      b -= wrap ? nb : 0;
      t += wrap ? 1 : 0;
    };
    Round r;
    req(r);
    land(r, 0);
    __syncthreads();
    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (blk + u < nblk) {  // uniform
          req(r);
          __builtin_amdgcn_sched_barrier(0);
          multiply(u);
          __builtin_amdgcn_sched_barrier(0);
          land(r, u ^ 1);  // (that buffer was last read a block ago: the barrier below closed that)
          __syncthreads();
        }
      }
    }
  };
  // (every lane of the workgroup has to agree on the form: the loads of a round are counted)
  __shared__ int s_norun;
  if (tid == 0) s_norun = 0;
  __syncthreads();
  if (!lab_run) s_norun = 1;
  __syncthreads();
  if (s_norun == 0) sweep(std::true_type{});
  else sweep(std::false_type{});
  // ---- the partial block: tile (i, j) of this wave holds rows 64 wi + 32 i + r(v, hi), columns 64 wj + 32 j + lo.
  // With `partials` it is STORED (slice-major [z][tile][256][256], 128-byte runs per half wave) and
  // lazy_mfma_fixed_grad_reduce_kernel sums the slices: 64 slices of atomics into the same megabyte of R were a
  // third of this kernel's time.  Without: atomics into R as the per-wave kernels do.
  float* mine = partials ? partials + (int64_t(blockIdx.z) * gridDim.x * gridDim.y + int64_t(blockIdx.x) * gridDim.y + blockIdx.y) * 65536 : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = 64 * wi + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * hi, col = 64 * wj + 32 * j + lo;
        const float x = acc[i][j][v];
        if (mine) mine[row * 256 + col] = x;
        else if (x != 0.0f) atomicAdd(&g.R[int64_t(s0 + rot + row) * N + d0 + rot + col], x);
      }
}
// R[live block] = sum over the slices of their partial blocks (R was zero-filled; nothing else writes these elements)
__global__ __launch_bounds__(256) void lazy_mfma_fixed_grad_reduce_kernel(LazyGroup g, const float* __restrict__ partials, int nt, int nz) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // element of tile blockIdx.y
  const int tile = blockIdx.y, ti = tile / nt, tj = tile - ti * nt;
  const float* p = partials + int64_t(tile) * 65536 + e;
  const int64_t stride = int64_t(nt) * nt * 65536;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int z = 0;
  for (; z + 4 <= nz; z += 4) {
    s0 += p[int64_t(z) * stride];
    s1 += p[int64_t(z + 1) * stride];
    s2 += p[int64_t(z + 2) * stride];
    s3 += p[int64_t(z + 3) * stride];
  }
  for (; z < nz; ++z) s0 += p[int64_t(z) * stride];
  const int row = e >> 8, col = e & 255;
  g.R[int64_t(ti * 256 + g.rot + row) * g.N + tj * 256 + g.rot + col] = (s0 + s1) + (s2 + s3);
}
// grad[a] += exp(w[a]) * R[src][dst]  (the balancing shifts of A' and Q' cancel exactly:
// A' * Q' = exp(alpha + em + beta - Z) * delta)
__global__ void lazy_dense_arc_grad_kernel(LazyGroup g) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.g.A) return;
  const gtnx_i4 r = g.lrec_in[k];
  if (r.y < 0) return;
  const int d = g.g.dst[r.w];
  const float v = g.R[int64_t(r.x) * g.N + d];
  if (v != 0.0f) atomicAdd(g.grad_fixed + r.w, v * __expf(__int_as_float(r.z)));
}

// zt[t][b] = log sum_n exp(alpha[t+1][b][n] + beta[t+1][b][n]); one wave per (t, b)
__global__ void lazy_local_z_kernel(LazyGroup g, float* zt) {
  const int64_t row = int64_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  if (row >= int64_t(g.T) * g.nb) return;
  const int t = int(row / g.nb), b = int(row % g.nb);
  const int64_t o = int64_t(t + 1) * g.nb * g.N + int64_t(b) * g.N;
  float m = NEG_INF;
  for (int n = lane; n < g.N; n += 64) m = fmaxf(m, g.alpha[o + n] + g.beta[o + n]);
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
  float sum = 0.0f;
  if (m != NEG_INF && m != -NEG_INF)
    for (int n = lane; n < g.N; n += 64) {
      const float x = g.alpha[o + n] + g.beta[o + n];
      if (x != NEG_INF) sum += expf(x - m);
    }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) sum += __shfl_xor(sum, k, 64);
  if (lane == 0) zt[row] = (m == NEG_INF || m == -NEG_INF) ? m : m + logf(sum);
}

} // namespace

size_t lazy_step_lds_bytes(const LazyGroup& g) { return sizeof(float) * size_t(BT) * size_t(g.Npad + g.Cpad); }
int lazy_tile_nodes() { return DT; }
int lazy_tile_batch() { return BT; }

void launch_lazy_pack(const LazyGroup& g, gtnx_i4* lrec_in, gtnx_i4* lrec_out, hipStream_t st) {
  if (g.g.A <= 0) return;
  hipLaunchKernelGGL(lazy_pack_kernel, dim3((g.g.A + kBlock - 1) / kBlock), dim3(kBlock), 0, st, g, lrec_in, lrec_out);
}

void launch_lazy_init(const LazyGroup& g, int which, hipStream_t st) {
  const int64_t tot = int64_t(g.nb) * g.N;
  if (tot <= 0) return;
  hipLaunchKernelGGL(lazy_init_kernel, dim3(unsigned((tot + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, g, which);
}

void launch_lazy_step(const LazyGroup& g, int t, int mode, int backward, hipStream_t st) {
  const dim3 grid((g.N + DT - 1) / DT, (g.nb + BT - 1) / BT);
  const size_t lds = lazy_step_lds_bytes(g);
  static std::atomic<uint64_t> attr_done{0};
  if (gtnx_first_on_device first{attr_done}) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lazy_step_kernel<SD_LOG, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lazy_step_kernel<SD_LOG, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lazy_step_kernel<SD_TROPICAL, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  }
  if (backward)
    hipLaunchKernelGGL((lazy_step_kernel<SD_LOG, true>), grid, dim3(kTile), lds, st, g, t);
  else if (mode == SD_LOG)
    hipLaunchKernelGGL((lazy_step_kernel<SD_LOG, false>), grid, dim3(kTile), lds, st, g, t);
  else
    hipLaunchKernelGGL((lazy_step_kernel<SD_TROPICAL, false>), grid, dim3(kTile), lds, st, g, t);
}

void launch_lazy_final(const LazyGroup& g, int mode, hipStream_t st) {
  if (g.nb <= 0) return;
  if (mode == SD_LOG)
    hipLaunchKernelGGL(lazy_final_kernel<SD_LOG>, dim3(g.nb), dim3(kBlock), 0, st, g);
  else
    hipLaunchKernelGGL(lazy_final_kernel<SD_TROPICAL>, dim3(g.nb), dim3(kBlock), 0, st, g);
}

void launch_lazy_path(const LazyGroup& g, int* path_arc, int* path_il, int* path_ol, float* path_w, int* path_len,
                      hipStream_t st) {
  if (g.nb <= 0) return;
  if (!g.bp) {  // max-plus regime: the path is re-derived from alpha
    launch_maxplus_path(g, path_arc, path_il, path_ol, path_w, path_len, st);
    return;
  }
  hipLaunchKernelGGL(lazy_path_kernel, dim3((g.nb + 63) / 64), dim3(64), 0, st, g, path_arc, path_il, path_ol, path_w,
                     path_len);
}

void launch_lazy_local_z(const LazyGroup& g, float* zt, hipStream_t st) {
  const int64_t rows = int64_t(g.T) * g.nb;
  if (rows <= 0) return;
  hipLaunchKernelGGL(lazy_local_z_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, st, g, zt);
}

bool lazy_z_chain_grad_ok(const LazyGroup& g) { return g.N <= 1024 && g.C <= 4096; }
void launch_lazy_z_chain_grad(const LazyGroup& g, const int* node_label, float* zt, hipStream_t st) {
  const int64_t npairs = int64_t(g.T) * g.nb;
  if (npairs <= 0) return;
  const int64_t per_wg = 4 * ZG_PAIRS;
  const dim3 grid(unsigned((npairs + per_wg - 1) / per_wg));
  const size_t lds = sizeof(float) * 4 * size_t(g.C);
  if (g.lab_unique) hipLaunchKernelGGL(lazy_z_chain_grad_kernel<true>, grid, dim3(kBlock), lds, st, g, node_label, zt);
  else hipLaunchKernelGGL(lazy_z_chain_grad_kernel<false>, grid, dim3(kBlock), lds, st, g, node_label, zt);
}

void launch_lazy_chain_grad(const LazyGroup& g, const int* node_label, hipStream_t st) {
  if (g.nb <= 0 || g.T <= 0) return;
  const dim3 grid(g.T, g.nb);
  const size_t lds = sizeof(float) * size_t(g.C);
  if (node_label)
    hipLaunchKernelGGL(lazy_chain_grad_nodes_kernel, grid, dim3(kBlock), lds, st, g, node_label);
  else
    hipLaunchKernelGGL(lazy_chain_grad_arcs_kernel, grid, dim3(kBlock), lds, st, g);
}

void launch_lazy_fixed_grad(const LazyGroup& g, int max_in_deg, hipStream_t st) {
  if (g.nb <= 0 || g.T <= 0 || max_in_deg <= 0) return;
  const int t_per_block = 32;
  const dim3 grid((g.N + DT - 1) / DT, (g.nb + BT - 1) / BT, (g.T + t_per_block - 1) / t_per_block);
  const size_t lds = lazy_step_lds_bytes(g) + sizeof(float) * size_t(DT) * size_t(max_in_deg);
  static std::atomic<uint64_t> attr_done{0};
  if (gtnx_first_on_device first{attr_done})
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lazy_fixed_grad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  hipLaunchKernelGGL(lazy_fixed_grad_kernel, grid, dim3(kTile), lds, st, g, t_per_block, max_in_deg);
}

void launch_lazy_path_grad(const LazyPathGrad& a, hipStream_t st) {
  if (a.len <= 0) return;
  hipLaunchKernelGGL(lazy_path_grad_kernel, dim3((a.len + 255) / 256), dim3(256), 0, st, a);
}

} // namespace gtnx

namespace gtnx {
void launch_lazy_dense_prep(const LazyGroup& g, float* E, float* cmax, hipStream_t st) {
  if (g.N <= 0) return;
  (void)hipMemsetAsync(E, 0, sizeof(float) * size_t(g.N) * size_t(g.N), st);
  hipLaunchKernelGGL(lazy_dense_cmax_kernel, dim3((g.N + 255) / 256), dim3(256), 0, st, g, cmax);
  if (g.g.A > 0)
    hipLaunchKernelGGL(lazy_dense_fill_kernel, dim3((g.g.A + 255) / 256), dim3(256), 0, st, g, E, (const float*)cmax);
}
void launch_lazy_dense_step(const LazyGroup& g, int t, int backward, hipStream_t st, const float* vin, float* vout) {
  const dim3 grid((g.N + TT - 1) / TT, (g.nb + TT - 1) / TT);
  if (backward)
    hipLaunchKernelGGL(lazy_dense_step_kernel<true>, grid, dim3(kDense), 0, st, g, t, vin, vout);
  else
    hipLaunchKernelGGL(lazy_dense_step_kernel<false>, grid, dim3(kDense), 0, st, g, t, (const float*)nullptr,
                       (float*)nullptr);
}

void launch_lazy_mfma_prep(const LazyGroup& g, hipStream_t st) {
  const int64_t n = int64_t(g.Kpad) * g.Npad2;
  hipLaunchKernelGGL(lazy_mfma_pad_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, g, const_cast<float*>(g.Ep),
                     const_cast<float*>(g.ETp));
}
// which 0: before the forward steps (all keys "-inf", X_0); 1: before the backward steps
void launch_lazy_mfma_rowmax(const LazyGroup& g, int which, hipStream_t st) {
  const int64_t rows = int64_t(g.T + 1) * g.nb;
  if (rows > 0)
    hipLaunchKernelGGL(lazy_mfma_rowmax_kernel, dim3(unsigned((rows + 255) / 256)), dim3(256), 0, st,
                       (const float*)(which ? g.bmaxp : g.amaxp), which ? g.bmax : g.amax, rows, g.ntp);
}
void launch_lazy_mfma_score(const LazyGroup& g, hipStream_t st) {
  if (g.nb > 0) hipLaunchKernelGGL(lazy_mfma_score_kernel, dim3(unsigned(g.nb)), dim3(64), 0, st, g);
}
void launch_lazy_mfma_init(const LazyGroup& g, int which, hipStream_t st) {
  const int64_t nk = int64_t(g.T + 1) * g.nb * g.ntp;
  launch_fill_i32(reinterpret_cast<int*>(which ? g.bmaxp : g.amaxp), int(0xff800000), size_t(nk), st);  // -inf
  (void)hipMemsetAsync(g.xt[0], 0, sizeof(float) * size_t(g.Kpad) * size_t(g.nbpad), st);
  (void)hipMemsetAsync(g.xt[1], 0, sizeof(float) * size_t(g.Kpad) * size_t(g.nbpad), st);
  if (which) hipLaunchKernelGGL(lazy_mfma_init_kernel<true>, dim3(g.nbpad), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(lazy_mfma_init_kernel<false>, dim3(g.nbpad), dim3(256), 0, st, g);
}
void launch_lazy_mfma_step(const LazyGroup& g, int t, int backward, hipStream_t st) {
  const int Nl = g.N - g.rot;
  const dim3 grid(unsigned((Nl + 31) / 32) * unsigned(g.nbpad / 32));
  if (backward) hipLaunchKernelGGL(lazy_mfma_step_kernel<true>, grid, dim3(MF_WAVES * 64), 0, st, g, t);
  else hipLaunchKernelGGL(lazy_mfma_step_kernel<false>, grid, dim3(MF_WAVES * 64), 0, st, g, t);
  if (!backward && t == 0 && g.rot > 0) {
    const int64_t n = int64_t(g.Kpad - Nl) * g.nbpad;
    hipLaunchKernelGGL(lazy_mfma_dead_rows_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, g, g.xt[0], Nl);
  }
}
namespace {
template <bool BWD, int GPW>
bool launch_chain(const LazyGroup& g, int* sync, dim3 grid, hipStream_t st) {
  LazyGroup gg = g;
  void* args[] = {&gg, &sync};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lazy_mfma_chain_kernel<BWD, GPW>), grid,
                                                  dim3(MF_WAVES * 64), args, 0, st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}
}  // namespace

size_t lazy_mfma_chain_sync_ints(const LazyGroup& g) { return size_t(g.nbpad / 32) + 8; }
// all T steps of a pass in one cooperative launch (lazy_mfma_chain_kernel); false: not applicable here (shape,
// residency, GTNX_NO_CHAIN) -- the caller launches the steps one by one.  `sync`: zeroed ints (see above)
bool launch_lazy_mfma_chain(const LazyGroup& g, int backward, int* sync, int cus, hipStream_t st) {
  // OFF unless GTNX_CHAIN=1: measured on the MI355X at C4 (B = 512, T = 1000, C = 512) the pass takes 10.97 ms
  // this way against 9.66 ms as T launches (profiles/r03_c4_chain.txt) -- a step's hand-off is four dependent
  // trips (write-through acknowledged -> arrival atomic -> poll -> operand loads past L1), which costs more than
  // the 2.9 us of launch + kernel boundary it removes.  Kept as the measured alternative, not as the default.
  static const bool on = getenv("GTNX_CHAIN") != nullptr && getenv("GTNX_CHAIN")[0] == '1';
  if (!on || g.T <= 0) return false;
  const int Nl = g.N - g.rot;
  const int ncol = (Nl + 31) / 32, nrow = g.nbpad / 32;
  const int grid = ncol * nrow;
  if (grid > cus || (g.Kpad & 31) != 0 || g.ntp > 32) return false;  // one workgroup per CU, all resident
  const int gpw = (g.Kpad >> 2) / MF_WAVES;
  if ((g.Kpad >> 2) % MF_WAVES != 0) return false;
#define GTNX_CHAIN(N_)                                                                      \
  if (gpw == N_)                                                                            \
    return backward ? launch_chain<true, N_>(g, sync, dim3(unsigned(grid)), st)             \
                    : launch_chain<false, N_>(g, sync, dim3(unsigned(grid)), st);
  GTNX_CHAIN(6)
  GTNX_CHAIN(12)
  GTNX_CHAIN(18)
  GTNX_CHAIN(24)
#undef GTNX_CHAIN
  return false;
}
void launch_lazy_mfma_keys(float* keys, int64_t n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(lazy_mfma_keys_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, keys, n);
}
// bytes of the partial-block scratch launch_lazy_mfma_fixed_grad can use (0: none wanted for this graph)
size_t lazy_mfma_fixed_grad_scratch_bytes(const LazyGroup& g) {
  const int Nl = g.N - g.rot;
  if (Nl < 256 || (Nl & 255) != 0 || ((g.N - g.rot) & 3) != 0 || getenv("GTNX_FIXED_GRAD_PER_WAVE") || getenv("GTNX_FIXED_GRAD_ATOMICS")) return 0;
  const int tiles = (Nl / 256) * (Nl / 256);
  return size_t(std::max(1, 512 / tiles)) * tiles * 65536 * 4;  // (up to 512 workgroups' blocks)
}
void launch_lazy_mfma_fixed_grad(const LazyGroup& g, void* pair_consts, hipStream_t st, void* partials, size_t partials_bytes) {
  if (g.N <= 0 || g.T <= 0 || g.nb <= 0 || !g.grad_fixed) return;
  const int64_t npairs = int64_t(g.T) * g.nb;
  gtnx_f4* pc = static_cast<gtnx_f4*>(pair_consts);
  hipLaunchKernelGGL(lazy_mfma_pairs_kernel, dim3(unsigned((npairs + 255) / 256)), dim3(256), 0, st, g, pc);
  static const int ppb_env = getenv("GTNX_FG_PPB") ? atoi(getenv("GTNX_FG_PPB")) : 0;
  const int pairs_per_block = ppb_env > 0 ? ppb_env : 2048;  // (the 4-byte version: 5.2 ms at 16384, 4.95 ms at 2048)
  const dim3 grid(unsigned((g.N + 63) / 64), unsigned((g.N + 63) / 64), unsigned((npairs + pairs_per_block - 1) / pairs_per_block));
  if (((g.N - g.rot) & 3) == 0 && g.N >= 8 && !getenv("GTNX_FIXED_GRAD_NARROW")) {  // whole groups of four live slots: wide loads
    // (slices of 4096 pairs: the 40 blocks of a slice run together and re-read the same rows of the planes while
    // they are still cached -- 5.7 ms at 16384, 4.5 ms at 4096; ordering the blocks by XCD on top changed nothing)
    const int ppb = ppb_env > 0 ? ppb_env : 4096;
    const int Nl = g.N - g.rot;
    static const bool no_wg = getenv("GTNX_FIXED_GRAD_PER_WAVE") != nullptr;
    if (!no_wg && Nl >= 256 && (Nl & 255) == 0) {
      // live slots in whole blocks of 256: one workgroup per 256 x 256 block of R and slice of the pairs, as many slices
      // as fill the chip once (lazy_mfma_fixed_grad_wg_kernel); the dead source slots' blocks per wave as before
      const int tiles = (Nl / 256) * (Nl / 256);
      static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        (void)hipGetLastError();
        return n > 0 ? n : 256;
      }();
      int z = std::max(1, cus / tiles);
      int64_t per = (npairs + z - 1) / z;
      per = (per + FGW_KB - 1) / FGW_KB * FGW_KB;
      z = int((npairs + per - 1) / per);
      const dim3 gw(unsigned(Nl / 256), unsigned(Nl / 256), unsigned(z));
      float* parts = (partials && z > 1 && partials_bytes >= size_t(z) * tiles * 65536 * 4) ? static_cast<float*>(partials) : nullptr;
      if (g.em_base) hipLaunchKernelGGL(lazy_mfma_fixed_grad_wg_kernel<true>, gw, dim3(1024), 0, st, g, (const gtnx_f4*)pc, int(per), parts);
      else hipLaunchKernelGGL(lazy_mfma_fixed_grad_wg_kernel<false>, gw, dim3(1024), 0, st, g, (const gtnx_f4*)pc, int(per), parts);
      if (parts) hipLaunchKernelGGL(lazy_mfma_fixed_grad_reduce_kernel, dim3(256, unsigned(tiles)), dim3(256), 0, st, g, parts, Nl / 256, z);
      if (g.rot > 0) {
        const int sb_first = Nl / 128;
        const dim3 gd(unsigned((g.N + 127) / 128 - sb_first), unsigned((g.N + 63) / 64), 1u);  // (their pairs: step 0 only)
        if (g.em_base) hipLaunchKernelGGL(lazy_mfma_fixed_grad4_kernel<true>, gd, dim3(256), 0, st, g, (const gtnx_f4*)pc, std::max(ppb, g.nb), sb_first);
        else hipLaunchKernelGGL(lazy_mfma_fixed_grad4_kernel<false>, gd, dim3(256), 0, st, g, (const gtnx_f4*)pc, std::max(ppb, g.nb), sb_first);
      }
    } else {
      const dim3 grid4(unsigned((g.N + 127) / 128), unsigned((g.N + 63) / 64), unsigned((npairs + ppb - 1) / ppb));
      if (g.em_base) hipLaunchKernelGGL(lazy_mfma_fixed_grad4_kernel<true>, grid4, dim3(256), 0, st, g, (const gtnx_f4*)pc, ppb, 0);
      else hipLaunchKernelGGL(lazy_mfma_fixed_grad4_kernel<false>, grid4, dim3(256), 0, st, g, (const gtnx_f4*)pc, ppb, 0);
    }
  } else if (g.em_base) hipLaunchKernelGGL(lazy_mfma_fixed_grad_kernel<true>, grid, dim3(256), 0, st, g, (const gtnx_f4*)pc, pairs_per_block);
  else hipLaunchKernelGGL(lazy_mfma_fixed_grad_kernel<false>, grid, dim3(256), 0, st, g, (const gtnx_f4*)pc, pairs_per_block);
  if (g.g.A > 0) hipLaunchKernelGGL(lazy_dense_arc_grad_kernel, dim3((g.g.A + 255) / 256), dim3(256), 0, st, g);
}
void launch_lazy_dense_fixed_grad(const LazyGroup& g, hipStream_t st) {
  if (g.N <= 0 || g.T <= 0 || g.nb <= 0 || !g.grad_fixed) return;
  const int64_t npairs = int64_t(g.T) * g.nb;
  const int pairs_per_block = 16384;
  const dim3 grid((g.N + TT - 1) / TT, (g.N + TT - 1) / TT, unsigned((npairs + pairs_per_block - 1) / pairs_per_block));
  hipLaunchKernelGGL(lazy_dense_fixed_grad_kernel, grid, dim3(kDense), 0, st, g, pairs_per_block);
  if (g.g.A > 0) hipLaunchKernelGGL(lazy_dense_arc_grad_kernel, dim3((g.g.A + 255) / 256), dim3(256), 0, st, g);
}
} // namespace gtnx
