// compose_wide.hip -- chain products whose explicit partner has WIDE nodes (tens to hundreds of arcs per node:
// n-gram / ASG transition graphs), gtn/functions/compose.cpp:377-522 for `intersect(emissions, transitions)`
// (benchmarks/ctc.cpp:118-122, examples/asg.cpp) with the reference's node and arc numbering.
//
// compose.hip gives a frontier node to ONE lane, which caches at most four candidate arcs in registers; a node
// with more falls to its general path -- a lane walking hundreds of arcs one dependent HBM access at a time
// (91 ms for T = 200 frames x a 30-node bigram graph).  Here the unit of work is the partner's ARC:
//
//  * the product of an implicit linear chain (gtn/creations.cpp:20-33) and an epsilon-free partner is layered in
//    time: pair (n, t) only steps to (n', t + 1), and the chain offers every label below C at every t.  So a
//    pair's arcs are the partner arcs of n whose matching label is below C, IN THE PARTNER'S LIST ORDER (the
//    chain is label-sorted with one arc per label: whichever side the matcher queries, compose.cpp:211-374, the
//    matches come out in the order of the partner's list -- as long as a partner that claims to be sorted is
//    sorted on the label being matched, which the host checks), filtered by co-reachability of (n', t + 1);
//  * co-reachability is a set of partner nodes per time, B[t] = pre(B[t + 1]), B[T] = accept: bit rows in HBM,
//    computed backwards until two consecutive rows agree (every earlier one is then the same);
//  * a BFS level is five passes of one 1024-lane workgroup over the level's candidate arcs, a WAVE per
//    frontier node: count the valid arcs, prefix sum (arc slots = the reference's arc ids), emit the arc
//    fields + atomicMin of the arc rank per destination (first arc to reach a node numbers it,
//    compose.cpp:425-440), rank the owners (node ids), patch the destinations;
//  * once the frontier comes back unchanged under a constant filter, every later level is this one moved in
//    time: ids + k W, arc ids + k Aw, chain arc + k C.  The plan kernel only leaves a hole and a record; the
//    REPLICATION kernel fills the hole with as many workgroups as the output deserves (a 100-frame x 262 k-arc
//    product is a gigabyte of arcs: one workgroup cannot write that, 256 CUs can).
//
// The in-arc rows and the ordered start / accept lists come from compose.hip's transpose passes afterwards
// (ComposeOut::csr_built = 0), which are grid-parallel already.
#include <hip/hip_runtime.h>

#include <climits>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr int WB = 1024;          // lanes of the plan workgroup
constexpr int WW = WB / 64;       // its waves
constexpr int NO_CAP = 2048;      // partner nodes (LDS tables are indexed by partner node)

__device__ __forceinline__ int wave_incl_scan(int x) {
#define GTNX_SCAN_STEP(ctrl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)
  GTNX_SCAN_STEP(0x111, 0xf);
  GTNX_SCAN_STEP(0x112, 0xf);
  GTNX_SCAN_STEP(0x114, 0xf);
  GTNX_SCAN_STEP(0x118, 0xf);
  GTNX_SCAN_STEP(0x142, 0xa);
  GTNX_SCAN_STEP(0x143, 0xc);
#undef GTNX_SCAN_STEP
  return x;
}
// exclusive prefix sum over the workgroup's lanes (sh: WW ints)
__device__ __forceinline__ int block_scan(int v, int* sh, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WW; ++w) {
    const int s = sh[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + x - v;
}
__device__ __forceinline__ unsigned long long lanes_below() {
  const int lane = threadIdx.x & 63;
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// L2: the chain is the second graph (partner = g1), else the first (partner = g2)
template <bool L2>
__global__ __launch_bounds__(WB) void compose_wide_plan_kernel(const ComposeArgs* __restrict__ args) {
  const ComposeArgs a = args[blockIdx.x];
  const DGraph& ch = L2 ? a.g2 : a.g1;
  const DGraph& pg = L2 ? a.g1 : a.g2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N1 = a.g1.N;
  const int No = pg.N, NoW = (No + 31) >> 5;
  const int TM = ch.M, C = ch.C;
  const GTNX_G gtnx_i4* __restrict__ orec = pg.out_rec;
  const GTNX_G int* __restrict__ ooff = pg.out_off;
  __shared__ int front[2][NO_CAP];
  __shared__ int first[NO_CAP];   // per partner node: smallest arc rank reaching it this level; then its id
  __shared__ int cnt[NO_CAP];     // per frontier position: valid arcs -> first arc slot
  __shared__ unsigned bcur[NO_CAP / 32], bprev[NO_CAP / 32];
  __shared__ int sh_scan[WW];
  __shared__ int sh_w[WW];
  __shared__ int sh_flag[2];
  auto fail = [&](int code) {  // workgroup-uniform
    if (tid == 0) {
      ComposeOut o{};
      o.overflow = code;
      *a.out = o;
      a.counts[0] = a.counts[1] = 0;
    }
  };
  if (No > NO_CAP || No < 1 || TM < 1) {
    fail(2);
    return;
  }
  const long long tk0 = wall_clock64();
  auto pair_id = [&](int n, int t) { return L2 ? n + N1 * t : t + N1 * n; };
  // the label the chain is matched on, functions.cpp:225-251 (g1's olabel against g2's ilabel)
  auto mlabel = [&](const gtnx_i4& r) { return L2 ? r.y : r.x; };

  // ------------------------------------------------------------------ phase B (compose.cpp:64-104)
  // rows of NoW words, one per time, in the scratch the general kernel uses as its BFS queue
  unsigned* rows = reinterpret_cast<unsigned*>(a.queue);
  for (int x = tid; x < NoW; x += WB) bprev[x] = 0u;
  __syncthreads();
  for (int n = tid; n < No; n += WB)
    if (pg.nflags[n] & NF_ACCEPT) atomicOr(&bprev[n >> 5], 1u << (n & 31));
  __syncthreads();
  for (int x = tid; x < NoW; x += WB) rows[size_t(TM) * NoW + x] = bprev[x];
  int tB = -1;  // rows of times <= tB equal row tB
  for (int t = TM - 1; t >= 0; --t) {
    for (int x = tid; x < NoW; x += WB) bcur[x] = 0u;
    if (tid == 0) sh_flag[0] = 0;
    __syncthreads();
    for (int n = wave; n < No; n += WW) {
      const int b0 = ooff[n], deg = ooff[n + 1] - b0;
      bool any = false;
      for (int k = lane; k < deg && !any; k += 64) {
        const gtnx_i4 r = orec[b0 + k];
        const int l = mlabel(r);
        any = l >= 0 && l < C && ((bprev[r.z >> 5] >> (r.z & 31)) & 1u);
      }
      if (__ballot(any) && lane == 0) atomicOr(&bcur[n >> 5], 1u << (n & 31));
    }
    __syncthreads();
    for (int x = tid; x < NoW; x += WB) {
      rows[size_t(t) * NoW + x] = bcur[x];
      if (bcur[x] != bprev[x]) sh_flag[0] = 1;
    }
    __syncthreads();
    const bool same = sh_flag[0] == 0;
    __syncthreads();
    if (same) {  // B[t] == B[t+1]: every earlier time has this set
      tB = t + 1;
      break;
    }
    for (int x = tid; x < NoW; x += WB) bprev[x] = bcur[x];
    __syncthreads();
  }
  auto load_row = [&](int t) {  // -> bcur
    const unsigned* r = rows + size_t(t <= tB ? tB : t) * NoW;
    for (int x = tid; x < NoW; x += WB) bcur[x] = r[x];
  };
  __syncthreads();  // the rows written above are read back below (same workgroup)

  // ------------------------------------------------------------------ phase F (compose.cpp:392-493)
  const long long tk1 = wall_clock64();
  int nn = 0, na = 0;
  // start pairs: the chain has one start node, so (s1 outer, s2 inner) is the partner's list order
  load_row(0);
  __syncthreads();
  {
    const int ns = pg.n_start;
    for (int s0 = 0; s0 < ns; s0 += WB) {
      const int s = s0 + tid;
      int n = 0, ok = 0;
      if (s < ns) {
        n = pg.start_list[s];
        ok = (bcur[n >> 5] >> (n & 31)) & 1u;
      }
      int tot;
      const int off = block_scan(ok, sh_scan, tot);
      if (ok) {
        const int id = nn + off;
        if (id < a.Ncap && id < NO_CAP) {
          front[0][id] = n;
          a.pair_of[id] = pair_id(n, 0);
          a.nflags[id] = uint8_t(NF_START);  // TM >= 1: time 0 is not the chain's accept node
        }
      }
      nn += tot;
    }
    if (nn > a.Ncap) {
      fail(1);
      return;
    }
  }
  __syncthreads();
  int lo = 0, hi = nn, L = 0, cur = 0;
  int max_width = 0, max_level_arcs = 0;
  int rp_L = 0, rp_K = 0, rp_lo = 0, rp_W = 0, rp_na = 0, rp_Aw = 0;
  const GTNX_G float* __restrict__ w1 = a.g1.w;
  const GTNX_G float* __restrict__ w2 = a.g2.w;
  bool bad = false;
  while (lo < hi) {
    const int W = hi - lo, t = L;
    if (tid == 0) a.level_off[L] = lo;
    max_width = max(max_width, W);
    if (t >= TM) {  // the chain's accept time: no arcs leave
      for (int p = tid; p < W; p += WB) a.out_off[lo + p] = na;
      ++L;
      lo = hi;
      break;
    }
    load_row(t + 1);
    for (int x = tid; x < No; x += WB) first[x] = INT_MAX;
    __syncthreads();
    // ---- pass A: valid arcs per frontier node
    for (int p = wave; p < W; p += WW) {
      const int n = front[cur][p];
      const int b0 = ooff[n], deg = ooff[n + 1] - b0;
      int c = 0;
      for (int k0 = 0; k0 < deg; k0 += 64) {
        bool v = false;
        if (k0 + lane < deg) {
          const gtnx_i4 r = orec[b0 + k0 + lane];
          const int l = mlabel(r);
          v = l >= 0 && l < C && ((bcur[r.z >> 5] >> (r.z & 31)) & 1u);
        }
        c += __popcll(__ballot(v));
      }
      if (lane == 0) cnt[p] = c;
    }
    __syncthreads();
    int Aw;
    {
      const int p0 = 2 * tid, p1 = 2 * tid + 1;
      const int v0 = p0 < W ? cnt[p0] : 0, v1 = p1 < W ? cnt[p1] : 0;
      const int base = block_scan(v0 + v1, sh_scan, Aw);
      if (p0 < W) {
        cnt[p0] = base;
        a.out_off[lo + p0] = na + base;
      }
      if (p1 < W) {
        cnt[p1] = base + v0;
        a.out_off[lo + p1] = na + base + v0;
      }
    }
    if (na + (long long)Aw > a.Acap) {
      bad = true;
      break;
    }
    __syncthreads();
    // ---- pass B: emit (dst provisionally the partner node), claim destinations by arc rank
    for (int p = wave; p < W; p += WW) {
      const int n = front[cur][p];
      const int b0 = ooff[n], deg = ooff[n + 1] - b0;
      int run = cnt[p];
      for (int k0 = 0; k0 < deg; k0 += 64) {
        bool v = false;
        gtnx_i4 r{};
        int l = 0;
        if (k0 + lane < deg) {
          r = orec[b0 + k0 + lane];
          l = mlabel(r);
          v = l >= 0 && l < C && ((bcur[r.z >> 5] >> (r.z & 31)) & 1u);
        }
        const unsigned long long m = __ballot(v);
        if (v) {
          const int rk = run + __popcll(m & lanes_below());
          const int ai = na + rk;
          const int ca = t * C + l;  // the chain's arc
          a.src[ai] = lo + p;
          a.dst[ai] = r.z;
          a.il[ai] = L2 ? r.x : l;
          a.ol[ai] = L2 ? l : r.y;
          a.w[ai] = L2 ? w1[r.w] + w2[ca] : w1[ca] + w2[r.w];
          a.gi1[ai] = L2 ? r.w : ca;
          a.gi2[ai] = L2 ? ca : r.w;
          atomicMin(&first[r.z], rk);
        }
        run += __popcll(m);
      }
    }
    __syncthreads();
    // ---- pass C: owners (the first arc to reach a node) in arc order, a contiguous span of arcs per wave
    const int span = ((Aw + WW * 64 - 1) / (WW * 64)) * 64;
    const int r_lo = min(wave * span, Aw), r_hi = min(r_lo + span, Aw);
    {
      int c = 0;
      for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
        const int r = r0 + lane;
        const bool own = r < r_hi && first[a.dst[na + r]] == r;
        c += __popcll(__ballot(own));
      }
      if (lane == 0) sh_w[wave] = c;
    }
    __syncthreads();
    int newn = 0, obase = 0;
#pragma unroll
    for (int w = 0; w < WW; ++w) {
      const int s = sh_w[w];
      if (w < wave) obase += s;
      newn += s;
    }
    if (nn + (long long)newn > a.Ncap || newn > NO_CAP) {
      bad = true;
      break;
    }
    __syncthreads();  // every ownership test is done before `first` turns into the id table
    {
      int run = obase;
      for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
        const int r = r0 + lane;
        int m = 0;
        bool own = false;
        if (r < r_hi) {
          m = a.dst[na + r];
          own = first[m] == r;
        }
        const unsigned long long mk = __ballot(own);
        if (own) {
          const int x = run + __popcll(mk & lanes_below());
          const int id = nn + x;
          front[cur ^ 1][x] = m;
          a.pair_of[id] = pair_id(m, t + 1);
          a.nflags[id] = uint8_t((t + 1 == TM && (pg.nflags[m] & NF_ACCEPT)) ? NF_ACCEPT : 0);
        }
        run += __popcll(mk);
      }
    }
    __syncthreads();
    // owners read first[m] == r above; now first[m] <- id for every discovered node
    for (int x = tid; x < newn; x += WB) first[front[cur ^ 1][x]] = nn + x;
    if (tid == 0) sh_flag[1] = 0;
    __syncthreads();
    // ---- pass E: destinations
    for (int r = tid; r < Aw; r += WB) a.dst[na + r] = first[a.dst[na + r]];
    // is the new frontier the old one (same order)?
    bool moved = newn != W;
    if (!moved)
      for (int x = tid; x < W; x += WB)
        if (front[cur ^ 1][x] != front[cur][x]) sh_flag[1] = 1;
    __syncthreads();
    moved = moved || sh_flag[1] != 0;
    max_level_arcs = max(max_level_arcs, Aw);
    const int na_level = na;
    na += Aw;
    nn += newn;
    // ---- stationary levels: level L expanded time L under the filter B[L+1]; levels L+1 .. L+K are the same
    // under B[L+k+1] == B[L+1] (L+k+1 <= tB) with their new nodes before the chain's accept time
    const int K = min(tB - L - 1, TM - L - 2);
    if (!moved && rp_K == 0 && K >= 1 && nn + (long long)K * W <= a.Ncap && na + (long long)K * Aw <= a.Acap) {
      rp_L = L; rp_K = K; rp_lo = lo; rp_W = W; rp_na = na_level; rp_Aw = Aw;
      nn += K * W;
      na += K * Aw;
      lo += K * W;
      hi += K * W;
      L += K;
    }
    lo = hi;
    hi = nn;
    cur ^= 1;
    ++L;
    __syncthreads();
  }
  if (bad) {
    fail(1);
    return;
  }
  if (tid == 0) {
    a.level_off[L] = nn;
    a.out_off[nn] = na;
    ComposeOut o{};
    o.N = nn;
    o.A = na;
    o.L = L;
    o.layered = 1;
    o.overflow = 0;
    o.max_width = max_width;
    o.max_level_arcs = max_level_arcs;
    o.csr_built = 0;
    o.rep_levels = rp_K;
    o.skipped = 0;
    o.t_b = int(tk1 - tk0);
    o.t_f = int(wall_clock64() - tk1);
    o.wr_L = rp_L; o.wr_K = rp_K; o.wr_lo = rp_lo; o.wr_W = rp_W; o.wr_na = rp_na; o.wr_Aw = rp_Aw;
    *a.out = o;
  }
}

// fills the hole the plan kernel left: levels wr_L + 1 .. wr_L + wr_K, copies of level wr_L moved in time
constexpr int RB = 256;
constexpr int RK = 8;  // time steps per tile
template <bool L2>
__global__ __launch_bounds__(RB) void compose_wide_replicate_kernel(const ComposeArgs* __restrict__ args) {
  const ComposeArgs a = args[blockIdx.y];
  const ComposeOut o = *a.out;
  const int K = o.wr_K;
  if (K <= 0 || o.overflow) return;
  const int L0 = o.wr_L, lo = o.wr_lo, W = o.wr_W, na0 = o.wr_na, Aw = o.wr_Aw;
  const int hi = lo + W;
  const int C = L2 ? a.g2.C : a.g1.C;
  const int tshift = L2 ? a.g1.N : 1;
  const GTNX_G float* __restrict__ cw = L2 ? a.g2.w : a.g1.w;
  const GTNX_G float* __restrict__ fw = L2 ? a.g1.w : a.g2.w;
  const int tid = threadIdx.x;
  const int etiles = (Aw + RB - 1) / RB, ktiles = (K + RK - 1) / RK;
  for (int tile = blockIdx.x; tile < etiles * ktiles; tile += gridDim.x) {
    const int e = (tile % etiles) * RB + tid;
    const int k0 = (tile / etiles) * RK + 1, k1 = min(k0 + RK, K + 1);
    if (e < Aw) {
      const int s = a.src[na0 + e], d = a.dst[na0 + e], il = a.il[na0 + e], ol = a.ol[na0 + e];
      const int g1 = a.gi1[na0 + e], g2 = a.gi2[na0 + e];
      const int ca = L2 ? g2 : g1, fa = L2 ? g1 : g2;
      const float wf = fw[fa];
      float wc[RK];
#pragma unroll
      for (int u = 0; u < RK; ++u) wc[u] = k0 + u < k1 ? cw[ca + (k0 + u) * C] : 0.0f;
#pragma unroll
      for (int u = 0; u < RK; ++u) {
        const int k = k0 + u;
        if (k >= k1) break;
        const int ai = na0 + k * Aw + e;
        a.src[ai] = s + k * W;
        a.dst[ai] = d + k * W;
        a.il[ai] = il;
        a.ol[ai] = ol;
        a.w[ai] = L2 ? wf + wc[u] : wc[u] + wf;
        a.gi1[ai] = L2 ? fa : ca + k * C;
        a.gi2[ai] = L2 ? ca + k * C : fa;
      }
    }
  }
  // nodes: the frontier of level L0 + k is ids lo + k W .. (their out offsets); its new nodes hi + k W ..
  const int ntiles = (W + RB - 1) / RB;
  for (int tile = blockIdx.x; tile < ntiles * ktiles; tile += gridDim.x) {
    const int x = (tile % ntiles) * RB + tid;
    const int k0 = (tile / ntiles) * RK + 1, k1 = min(k0 + RK, K + 1);
    if (x < W) {
      const int oo = a.out_off[lo + x], pr = a.pair_of[hi + x];
      for (int k = k0; k < k1; ++k) {
        a.out_off[lo + k * W + x] = oo + k * Aw;
        a.pair_of[hi + k * W + x] = pr + k * tshift;
        a.nflags[hi + k * W + x] = 0;  // neither start (t > 0) nor accept (t < T)
      }
    }
  }
  if (blockIdx.x == 0)
    for (int k = 1 + tid; k <= K; k += RB) a.level_off[L0 + k] = lo + k * W;
}

} // namespace

int compose_wide_node_cap() { return NO_CAP; }

void launch_compose_wide(const ComposeArgs* d_args, int n, int lin2, int max_acap, hipStream_t st) {
  if (n <= 0) return;
  int g = (max_acap + RB * RK - 1) / (RB * RK);
  g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
  // a batch of n pairs fills the chip by itself: fewer replication workgroups per pair
  if (n > 1) g = g > (8192 / n + 1) ? (8192 / n + 1) : g;
  if (lin2) {
    hipLaunchKernelGGL(compose_wide_plan_kernel<true>, dim3(n), dim3(WB), 0, st, d_args);
    hipLaunchKernelGGL(compose_wide_replicate_kernel<true>, dim3(g, n), dim3(RB), 0, st, d_args);
  } else {
    hipLaunchKernelGGL(compose_wide_plan_kernel<false>, dim3(n), dim3(WB), 0, st, d_args);
    hipLaunchKernelGGL(compose_wide_replicate_kernel<false>, dim3(g, n), dim3(RB), 0, st, d_args);
  }
}

} // namespace gtnx
