// compose_wide.hip -- compose / intersect where compose.hip's lane-per-node kernels lose: wide nodes and big outputs.
//   1. compose_wide_plan_kernel + compose_wide_replicate_kernel: chain product, partner with wide nodes (below);
//   2. compose_replicate_kernel: the stationary levels of compose.hip's FAST chain products, written by a grid;
//   3. compose_pairs_kernel (+ sorted_view_kernel): two explicit graphs, a wave per frontier pair (further down).
// All keep the reference's node ids, arc ids and arc order (gtn/functions/compose.cpp:377-522).
//
// 1. Chain products whose explicit partner has WIDE nodes (tens to hundreds of arcs per node:
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
// exclusive prefix sum over the workgroup's lanes (sh: NW ints, NW = waves of the workgroup)
template <int NW = WW>
__device__ __forceinline__ int block_scan(int v, int* sh, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
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


// ================================================================================================================
// General products (both graphs explicit) with wide nodes: a WAVE per frontier pair.
//
// compose.hip's general variant walks a pair's matches with one lane: "for q in the query list: binary search in
// the search list, walk the equal-label run" (compose.cpp:211-374), each step a dependent L2 access -- 0.6 ms for
// a 21-node CTC target against a 30-node bigram graph, 1.2 s for benchmarks/functions.cpp's compose of two
// chains with 20 and 1000 arcs per node.  Here the 64 lanes of a wave take 64 query arcs at once (one binary
// search each, in parallel), and the matches of the block -- equal-label runs of different lengths -- are
// flattened by a prefix sum over the run lengths, so that every lane of the next step holds ONE candidate arc
// pair, in the reference's order (query-major, then the search list's order).  A lane finds its (query, offset)
// by a 6-step search over the prefix held in the wave's registers (ds_bpermute).
//
// The search side's list must be sorted on the label being matched.  Graphs that are not (UnsortedMatcher,
// compose.cpp:211-236: for i in out(n1): for j in out(n2)) get a stable label-sorted VIEW of their lists, built
// once per graph by sorted_view_kernel: q-major order over a stable view is exactly the double loop's order.
//
// Everything else follows compose.hip's general variant: pair table `state` in HBM (unreached / co-reachable /
// claimed by arc rank / node id), level-synchronous BFS, arc slots by prefix sum, first-arc ownership by
// atomicMax of the claim, epsilon moves after the matches (compose.cpp:449-490).
// ================================================================================================================
constexpr int ST_UNREACH = INT_MIN;
constexpr int ST_REACH = INT_MIN + 1;
constexpr int ST_FWD = INT_MIN + 2;  // reached from a start pair, not (yet) known to reach an accept pair
constexpr int EPS = -1;
__device__ __forceinline__ int claim_of(int r) { return -2 - r; }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// calls f(has, idx, i, j, il, ol) wave-uniformly, once per chunk of up to 64 candidate arc pairs of (n1, n2) in
// the reference's order: dst pair id, arc of g1 (-1: none), arc of g2, labels of the composed arc
// LOOSE (with IN = false): the out-lists walked with the BACKWARD pass's rules -- eps:eps label matches count and
// both kinds of epsilon moves are always taken (compose.cpp:64-104 has no epsilon filter) -- for the forward
// marking pass of `trim_fwd_first`, whose set must be closed under every step the backward pass can undo
template <bool IN, int MATCH, bool LOOSE = false, class F>
__device__ __forceinline__ void wave_candidates(const ComposeArgs& a, int n1, int n2, bool eps1_ok, bool eps2_ok, F&& f) {
  const int lane = threadIdx.x & 63;
  const int N1 = a.g1.N;
  const GTNX_G int* off1 = IN ? a.g1.in_off : a.g1.out_off;
  const GTNX_G int* off2 = IN ? a.g2.in_off : a.g2.out_off;
  const int b1 = off1[n1], d1 = off1[n1 + 1] - b1;
  const int b2 = off2[n2], d2 = off2[n2 + 1] - b2;
  // roles, functions.cpp:225-251 and compose.cpp:319
  const bool sg1 = MATCH == MATCH_SINGLY_G1 || (MATCH == MATCH_DOUBLY && d1 > d2);
  const GTNX_G gtnx_i4* qrec = sg1 ? (IN ? a.g2.in_rec : a.g2.out_rec) + b2 : (IN ? a.g1.in_rec : a.g1.out_rec) + b1;
  const GTNX_G gtnx_i4* srec = sg1 ? (IN ? a.s1_in : a.s1_out) + b1 : (IN ? a.s2_in : a.s2_out) + b2;
  const int dq = sg1 ? d2 : d1, ds = sg1 ? d1 : d2;
  if (ds > 0)
    for (int q0 = 0; q0 < dq; q0 += 64) {
      gtnx_i4 qr{};
      int lb = 0, c = 0;
      if (q0 + lane < dq) {
        qr = qrec[q0 + lane];
        const int ql = sg1 ? qr.x : qr.y;  // g2's ilabel : g1's olabel
        if (IN || LOOSE || ql != EPS) {    // direct eps:eps matches are skipped going forward (compose.cpp:425-428)
          int lo = 0, hi = ds;
          while (lo < hi) {  // std::lower_bound
            const int mid = (lo + hi) >> 1;
            const gtnx_i4 m = srec[mid];
            if ((sg1 ? m.y : m.x) < ql) lo = mid + 1; else hi = mid;
          }
          lb = lo;
          hi = ds;
          while (lo < hi) {  // std::upper_bound
            const int mid = (lo + hi) >> 1;
            const gtnx_i4 m = srec[mid];
            if ((sg1 ? m.y : m.x) <= ql) lo = mid + 1; else hi = mid;
          }
          c = lo - lb;
        }
      }
      const int incl = wave_incl_scan(c);
      const int P = incl - c;
      const int tot = __shfl(incl, 63);
      for (int k0 = 0; k0 < tot; k0 += 64) {
        const int k = k0 + lane;
        const bool has = k < tot;
        int qi = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1) {
          const int cand = qi + step;  // < 64
          if (__shfl(P, cand) <= k) qi = cand;
        }
        const int s = __shfl(lb, qi) + k - __shfl(P, qi);
        gtnx_i4 q4;
        q4.x = __shfl(qr.x, qi); q4.y = __shfl(qr.y, qi); q4.z = __shfl(qr.z, qi); q4.w = __shfl(qr.w, qi);
        gtnx_i4 s4{};
        if (has) s4 = srec[s];
        const gtnx_i4 r1 = sg1 ? s4 : q4, r2 = sg1 ? q4 : s4;
        f(has, r1.z + N1 * r2.z, r1.w, r2.w, r1.x, r2.y);
      }
    }
  // epsilon moves: g1's arcs with olabel eps, then g2's arcs with ilabel eps (lists sorted on that label have
  // them first, compose.cpp:36-41)
  if (!(a.g1.flags & GF_EPS_FREE) && (IN || LOOSE || eps1_ok)) {
    const GTNX_G gtnx_i4* rec = (IN ? a.g1.in_rec : a.g1.out_rec) + b1;
    for (int k0 = 0; k0 < d1; k0 += 64) {
      gtnx_i4 r{};
      bool e = false;
      if (k0 + lane < d1) {
        r = rec[k0 + lane];
        e = r.y == EPS;
      }
      const unsigned long long m = __ballot(e);
      if (m) f(e, r.z + N1 * n2, r.w, -1, r.x, EPS);
      if ((a.g1.flags & 2) && m != ~0ull) break;
    }
  }
  if (!(a.g2.flags & GF_EPS_FREE) && (IN || LOOSE || eps2_ok)) {
    const GTNX_G gtnx_i4* rec = (IN ? a.g2.in_rec : a.g2.out_rec) + b2;
    for (int k0 = 0; k0 < d2; k0 += 64) {
      gtnx_i4 r{};
      bool e = false;
      if (k0 + lane < d2) {
        r = rec[k0 + lane];
        e = r.x == EPS;
      }
      const unsigned long long m = __ballot(e);
      if (m) f(e, n1 + N1 * r.z, -1, r.w, EPS, r.y);
      if ((a.g2.flags & 1) && m != ~0ull) break;
    }
  }
}
// does the out-list of `n` hold an arc whose matched label is epsilon? (wave-uniform result)
__device__ __forceinline__ bool wave_has_eps(const DGraph& g, int n, bool second) {
  if (g.flags & GF_EPS_FREE) return false;
  const int lane = threadIdx.x & 63;
  const int b = g.out_off[n], d = g.out_off[n + 1] - b;
  bool any = false;
  for (int k0 = 0; k0 < d && !any; k0 += 64) {
    bool e = false;
    if (k0 + lane < d) {
      const gtnx_i4 r = g.out_rec[b + k0 + lane];
      e = (second ? r.x : r.y) == EPS;
    }
    any = __ballot(e) != 0ull;
  }
  return any;
}

// WBX lanes per workgroup: 1024 for a product on its own (a level's passes spread over 16 waves); 256 for a
// batch of products, whose levels are short -- a 1024-lane workgroup has a CU to itself (16 of its 28 wave
// slots at this kernel's register count), 256-lane ones share it seven at a time
template <int MATCH, int WBX>
__global__ __launch_bounds__(WBX) void compose_pairs_kernel(const ComposeArgs* __restrict__ args) {
  constexpr int WB = WBX, WW = WBX / 64;
  const ComposeArgs a = args[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N1 = a.g1.N, N2 = a.g2.N;
  __shared__ int sh_scan[WW];
  __shared__ int sh_w[WW];
  __shared__ int sh_tail;
  __shared__ int sh_flag[2];  // [0] layered, [1] overflow
  if (tid == 0) {
    sh_tail = 0;
    sh_flag[0] = 1;
    sh_flag[1] = 0;
  }
  __syncthreads();
  const long long tk0 = wall_clock64();
  if (N1 == 0 || N2 == 0) {
    if (tid == 0) {
      ComposeOut o{};
      o.layered = 1;
      o.csr_built = 1;
      *a.out = o;
      a.out_off[0] = 0;
      a.level_off[0] = 0;
      a.in_off[0] = 0;
      a.counts[0] = a.counts[1] = 0;
    }
    return;
  }
  // ------------------------------------------------------------------ phase B (compose.cpp:64-104)
  // `state` arrives filled with ST_UNREACH; the queue of pair ids is the reference's toExplore.
  // trim_fwd_first: the product of a narrow graph and a (nearly) complete one -- a target against a transition
  // model -- has a co-reachable set of about every pair (51 k for a 100-label target against 512 labels) of
  // which the start pairs reach a few hundred.  The pairs that end up in the product are those reached from a
  // start pair AND reaching an accept pair; marking the first set first and walking backwards only inside it
  // gives the same set (a path from a marked pair stays inside the marked set) for a fraction of the work.
  const bool fwd_first = a.trim_fwd_first != 0;
  if (fwd_first) {
    const int ns1 = a.g1.n_start, ns2 = a.g2.n_start;
    const int seeds = ns1 * ns2;
    for (int t = tid; t < seeds; t += WB) {
      const int idx = a.g1.start_list[t / ns2] + N1 * a.g2.start_list[t % ns2];
      st_agent(a.state + idx, ST_FWD);
      st_agent(a.queue + t, idx);
    }
    if (tid == 0) sh_tail = seeds;
    __syncthreads();
    int lo = 0, hi = seeds;
    while (lo < hi) {
      for (int f = lo + wave; f < hi; f += WW) {
        const int idx = ld_agent(a.queue + f);
        wave_candidates<false, MATCH, true>(a, idx % N1, idx / N1, true, true, [&](bool has, int pidx, int, int, int, int) {
          const bool fresh = has && atomicCAS(a.state + pidx, ST_UNREACH, ST_FWD) == ST_UNREACH;
          const unsigned long long m = __ballot(fresh);
          if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_tail, __popcll(m));
            base = __shfl(base, 0);
            if (fresh) st_agent(a.queue + base + __popcll(m & lanes_below()), pidx);
          }
        });
      }
      __syncthreads();
      lo = hi;
      hi = sh_tail;
      __syncthreads();
    }
    __syncthreads();
  }
  {
    const int unseen = fwd_first ? ST_FWD : ST_UNREACH;  // what the backward pass may still mark
    const int na1 = a.g1.n_accept, na2 = a.g2.n_accept;
    const int seeds = na1 * na2;
    if (tid == 0) sh_tail = 0;
    __syncthreads();
    for (int t0 = 0; t0 < seeds; t0 += WB) {
      const int t = t0 + tid;
      bool ok = false;
      int idx = 0;
      if (t < seeds) {
        idx = a.g1.accept_list[t / na2] + N1 * a.g2.accept_list[t % na2];
        ok = atomicCAS(a.state + idx, unseen, ST_REACH) == unseen;
      }
      const unsigned long long m = __ballot(ok);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&sh_tail, __popcll(m));
        base = __shfl(base, 0);
        if (ok) st_agent(a.queue + base + __popcll(m & lanes_below()), idx);
      }
    }
    __syncthreads();
    int lo = 0, hi = sh_tail;
    __syncthreads();
    while (lo < hi) {
      for (int f = lo + wave; f < hi; f += WW) {
        const int idx = ld_agent(a.queue + f);
        wave_candidates<true, MATCH>(a, idx % N1, idx / N1, true, true, [&](bool has, int pidx, int, int, int, int) {
          const bool fresh = has && atomicCAS(a.state + pidx, unseen, ST_REACH) == unseen;
          const unsigned long long m = __ballot(fresh);
          if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_tail, __popcll(m));
            base = __shfl(base, 0);
            if (fresh) st_agent(a.queue + base + __popcll(m & lanes_below()), pidx);
          }
        });
      }
      __syncthreads();
      lo = hi;
      hi = sh_tail;
      __syncthreads();
    }
  }
  // ------------------------------------------------------------------ phase F (compose.cpp:392-493)
  const long long tk1 = wall_clock64();
  int nn = 0, na = 0;
  {
    const int ns1 = a.g1.n_start, ns2 = a.g2.n_start;
    const int seeds = ns1 * ns2;  // (s1 outer, s2 inner), compose.cpp:392-401
    for (int t0 = 0; t0 < seeds; t0 += WB) {
      const int t = t0 + tid;
      int idx = 0, ok = 0, s1 = 0, s2 = 0;
      if (t < seeds) {
        s1 = a.g1.start_list[t / ns2];
        s2 = a.g2.start_list[t % ns2];
        idx = s1 + N1 * s2;
        ok = ld_agent(a.state + idx) == ST_REACH;
      }
      int tot;
      const int off = block_scan<WW>(ok, sh_scan, tot);
      if (ok) {
        const int id = nn + off;
        if (id < a.Ncap) {
          a.pair_of[id] = idx;
          a.nflags[id] = uint8_t(NF_START | (((a.g1.nflags[s1] & NF_ACCEPT) && (a.g2.nflags[s2] & NF_ACCEPT)) ? NF_ACCEPT : 0));
        }
      }
      __syncthreads();  // every test of this chunk read ST_REACH before ids go in (a start pair is listed once)
      if (ok && nn + off < a.Ncap) st_agent(a.state + idx, nn + off);
      nn += tot;
    }
    if (nn > a.Ncap) sh_flag[1] = 1;
  }
  __syncthreads();
  int lo = 0, hi = nn, L = 0, max_width = 0, max_level_arcs = 0;
  while (lo < hi && !sh_flag[1]) {
    if (tid == 0) a.level_off[L] = lo;
    max_width = max(max_width, hi - lo);
    // ---- pass A: valid candidates per frontier node (count kept in out_off until the scan)
    for (int node = lo + wave; node < hi; node += WW) {
      const int pr = a.pair_of[node];
      const int n1 = pr % N1, n2 = pr / N1;
      const bool em = wave_has_eps(a.g1, n1, false) && wave_has_eps(a.g2, n2, true);
      const bool acc1 = (a.g1.nflags[n1] & NF_ACCEPT) != 0, acc2 = (a.g2.nflags[n2] & NF_ACCEPT) != 0;
      const bool e1 = !em || acc2 || !acc1, e2 = !em || acc1;  // compose.cpp:461, :476
      int c = 0;
      wave_candidates<false, MATCH>(a, n1, n2, e1, e2, [&](bool has, int idx, int, int, int, int) {
        int st = ST_UNREACH;
        if (has) st = ld_agent(a.state + idx);
        const bool v = st != ST_UNREACH && st != ST_FWD;
        c += __popcll(__ballot(v));
      });
      if (lane == 0) st_agent(a.out_off + node, c);
    }
    __syncthreads();
    int total = 0;
    for (int c0 = lo; c0 < hi; c0 += 2 * WB) {
      const int p0 = c0 + 2 * tid, p1 = p0 + 1;
      const int v0 = p0 < hi ? ld_agent(a.out_off + p0) : 0, v1 = p1 < hi ? ld_agent(a.out_off + p1) : 0;
      int tot;
      const int base = block_scan<WW>(v0 + v1, sh_scan, tot);
      if (p0 < hi) st_agent(a.out_off + p0, na + total + base);
      if (p1 < hi) st_agent(a.out_off + p1, na + total + base + v0);
      total += tot;
    }
    if (na + (long long)total > a.Acap) {
      if (tid == 0) sh_flag[1] = 1;
      __syncthreads();
      break;
    }
    __syncthreads();
    // ---- pass B: emit (dst provisionally the pair id), claim undiscovered destinations by arc rank
    for (int node = lo + wave; node < hi; node += WW) {
      const int pr = a.pair_of[node];
      const int n1 = pr % N1, n2 = pr / N1;
      const bool em = wave_has_eps(a.g1, n1, false) && wave_has_eps(a.g2, n2, true);
      const bool acc1 = (a.g1.nflags[n1] & NF_ACCEPT) != 0, acc2 = (a.g2.nflags[n2] & NF_ACCEPT) != 0;
      const bool e1 = !em || acc2 || !acc1, e2 = !em || acc1;
      int run = ld_agent(a.out_off + node);
      wave_candidates<false, MATCH>(a, n1, n2, e1, e2, [&](bool has, int idx, int i, int j, int il, int ol) {
        int cur = ST_UNREACH;
        if (has) cur = ld_agent(a.state + idx);
        const bool v = cur != ST_UNREACH && cur != ST_FWD;
        const unsigned long long m = __ballot(v);
        if (v) {
          const int ai = run + __popcll(m & lanes_below());
          a.src[ai] = node;
          a.dst[ai] = idx;
          a.il[ai] = il;
          a.ol[ai] = ol;
          a.w[ai] = (i >= 0 ? a.g1.w[i] : 0.0f) + (j >= 0 ? a.g2.w[j] : 0.0f);
          a.gi1[ai] = i;
          a.gi2[ai] = j;
          if (cur < 0) atomicMax(a.state + idx, claim_of(ai - na));
        }
        run += __popcll(m);
      });
    }
    __syncthreads();
    // ---- pass C: owners in arc order, a contiguous span of the level's arcs per wave
    const int span = ((total + WW * 64 - 1) / (WW * 64)) * 64;
    const int r_lo = min(wave * span, total), r_hi = min(r_lo + span, total);
    {
      int c = 0;
      for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
        const int r = r0 + lane;
        const bool own = r < r_hi && ld_agent(a.state + a.dst[na + r]) == claim_of(r);
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
    if (nn + (long long)newn > a.Ncap) {
      if (tid == 0) sh_flag[1] = 1;
      __syncthreads();
      break;
    }
    {
      int run = obase;
      for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
        const int r = r0 + lane;
        int idx = 0;
        bool own = false;
        if (r < r_hi) {
          idx = a.dst[na + r];
          own = ld_agent(a.state + idx) == claim_of(r);
        }
        const unsigned long long mk = __ballot(own);
        int id = -1;
        if (own) {
          id = nn + run + __popcll(mk & lanes_below());
          const int d1 = idx % N1, d2 = idx / N1;
          a.pair_of[id] = idx;
          a.nflags[id] = uint8_t((((a.g1.nflags[d1] & NF_START) && (a.g2.nflags[d2] & NF_START)) ? NF_START : 0) |
                                 (((a.g1.nflags[d1] & NF_ACCEPT) && (a.g2.nflags[d2] & NF_ACCEPT)) ? NF_ACCEPT : 0));
        }
        if (r < r_hi) a.in_list[na + r] = id;  // scratch: in_list is built by the transpose passes
        run += __popcll(mk);
      }
    }
    __syncthreads();  // every ownership test has read its claim
    for (int r = tid; r < total; r += WB) {
      const int id = a.in_list[na + r];
      if (id >= 0) st_agent(a.state + a.dst[na + r], id);
    }
    __syncthreads();
    bool lay = true;
    for (int r = tid; r < total; r += WB) {
      const int id = ld_agent(a.state + a.dst[na + r]);
      a.dst[na + r] = id;
      lay = lay && id >= hi;
    }
    if (!lay) sh_flag[0] = 0;
    max_level_arcs = max(max_level_arcs, total);
    na += total;
    nn += newn;
    lo = hi;
    hi = nn;
    ++L;
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    a.level_off[L] = nn;
    a.out_off[nn < a.Ncap + 1 ? nn : a.Ncap] = na;
    ComposeOut o{};
    o.N = nn;
    o.A = na;
    o.L = L;
    o.layered = sh_flag[0];
    o.overflow = sh_flag[1];
    o.max_width = max_width;
    o.max_level_arcs = max_level_arcs;
    o.csr_built = 0;
    o.t_b = int(tk1 - tk0);
    o.t_f = int(wall_clock64() - tk1);
    *a.out = o;
    if (sh_flag[1]) a.counts[0] = a.counts[1] = 0;
  }
}

// stable label-sorted copy of every node's out- and in-records (key: olabel when the graph is searched as g1,
// ilabel as g2): a workgroup per node, rank of an element = elements before it in (key, position) order
__global__ __launch_bounds__(256) void sorted_view_kernel(DGraph g, int key_ol, gtnx_i4* __restrict__ out_view,
                                                          gtnx_i4* __restrict__ in_view) {
  const int n = blockIdx.x;
  for (int side = 0; side < 2; ++side) {
    const int b = (side ? g.in_off : g.out_off)[n], d = (side ? g.in_off : g.out_off)[n + 1] - b;
    const GTNX_G gtnx_i4* rec = (side ? g.in_rec : g.out_rec) + b;
    gtnx_i4* view = (side ? in_view : out_view) + b;
    for (int e = threadIdx.x; e < d; e += blockDim.x) {
      const gtnx_i4 r = rec[e];
      const int key = key_ol ? r.y : r.x;
      int rank = 0;
      for (int x = 0; x < d; ++x) {
        const gtnx_i4 o = rec[x];
        const int k2 = key_ol ? o.y : o.x;
        rank += (k2 < key) || (k2 == key && x < e);
      }
      view[rank] = r;
    }
  }
}


// The same for compose.hip's FAST variant (ComposeArgs::rep_grid): its products also carry the in-arc rows
// (in_off / in_src / in_w, and in_list unless `skip`), and in `skip` mode leave src / il / ol out.  The in-row
// slot p of the template level holds arc in_list[p]: its copy k levels later is that arc's copy.
template <bool L2>
__global__ __launch_bounds__(RB) void compose_replicate_kernel(const ComposeArgs* __restrict__ args) {
  const ComposeArgs a = args[blockIdx.y];
  const ComposeOut o = *a.out;
  const int K = o.wr_K;
  if (K <= 0 || o.overflow) return;
  const int L0 = o.wr_L, lo = o.wr_lo, W = o.wr_W, na0 = o.wr_na, Aw = o.wr_Aw;
  const int hi = lo + W;
  const bool skip = o.skipped != 0;
  const int C = L2 ? a.g2.C : a.g1.C;
  const GTNX_G float* __restrict__ cw = L2 ? a.g2.w : a.g1.w;
  const GTNX_G float* __restrict__ fw = L2 ? a.g1.w : a.g2.w;
  const int tid = threadIdx.x;
  const int etiles = (Aw + RB - 1) / RB, ktiles = (K + RK - 1) / RK;
  for (int tile = blockIdx.x; tile < etiles * ktiles; tile += gridDim.x) {
    const int e = (tile % etiles) * RB + tid;
    const int k0 = (tile / etiles) * RK + 1, k1 = min(k0 + RK, K + 1);
    if (e < Aw) {
      const int t = na0 + e;
      const int d = a.dst[t], g1 = a.gi1[t], g2 = a.gi2[t];
      const int ca = L2 ? g2 : g1, fa = L2 ? g1 : g2;
      const float wf = fw[fa];
      // the in-row slot with this index
      const int is = a.in_src[t], ia = a.in_list[t];
      const int ica = L2 ? a.gi2[ia] : a.gi1[ia];
      const float iwf = fw[L2 ? a.gi1[ia] : a.gi2[ia]];
      int s = 0, il = 0, ol = 0;
      if (!skip) {
        s = a.src[t];
        il = a.il[t];
        ol = a.ol[t];
      }
      float wc[RK], iwc[RK];
#pragma unroll
      for (int u = 0; u < RK; ++u) {
        wc[u] = k0 + u < k1 ? cw[ca + (k0 + u) * C] : 0.0f;
        iwc[u] = k0 + u < k1 ? cw[ica + (k0 + u) * C] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < RK; ++u) {
        const int k = k0 + u;
        if (k >= k1) break;
        const int ai = t + k * Aw;
        a.dst[ai] = d + k * W;
        a.w[ai] = wf + wc[u];
        a.gi1[ai] = L2 ? fa : ca + k * C;
        a.gi2[ai] = L2 ? ca + k * C : fa;
        a.in_src[ai] = is + k * W;
        a.in_w[ai] = iwf + iwc[u];
        if (!skip) {
          a.src[ai] = s + k * W;
          a.il[ai] = il;
          a.ol[ai] = ol;
          a.in_list[ai] = ia + k * Aw;
        }
      }
    }
  }
  const int ntiles = (W + RB - 1) / RB;
  for (int tile = blockIdx.x; tile < ntiles * ktiles; tile += gridDim.x) {
    const int x = (tile % ntiles) * RB + tid;
    const int k0 = (tile / ntiles) * RK + 1, k1 = min(k0 + RK, K + 1);
    if (x < W) {
      const int oo = a.out_off[lo + x], io = a.in_off[hi + x];
      for (int k = k0; k < k1; ++k) {
        a.out_off[lo + k * W + x] = oo + k * Aw;
        a.in_off[hi + k * W + x] = io + k * Aw;
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

void launch_compose_replicate(const ComposeArgs* d_args, int n, int lin2, int max_acap, hipStream_t st) {
  if (n <= 0) return;
  int g = (max_acap + RB * RK - 1) / (RB * RK);
  g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
  if (n > 1) g = g > (8192 / n + 1) ? (8192 / n + 1) : g;
  if (lin2) hipLaunchKernelGGL(compose_replicate_kernel<true>, dim3(g, n), dim3(RB), 0, st, d_args);
  else hipLaunchKernelGGL(compose_replicate_kernel<false>, dim3(g, n), dim3(RB), 0, st, d_args);
}

namespace {
template <int WBX>
void launch_pairs_w(const ComposeArgs* d_args, int n, int matcher, hipStream_t st) {
  switch (matcher) {
    case MATCH_UNSORTED: hipLaunchKernelGGL((compose_pairs_kernel<MATCH_UNSORTED, WBX>), dim3(n), dim3(WBX), 0, st, d_args); break;
    case MATCH_SINGLY_G1: hipLaunchKernelGGL((compose_pairs_kernel<MATCH_SINGLY_G1, WBX>), dim3(n), dim3(WBX), 0, st, d_args); break;
    case MATCH_SINGLY_G2: hipLaunchKernelGGL((compose_pairs_kernel<MATCH_SINGLY_G2, WBX>), dim3(n), dim3(WBX), 0, st, d_args); break;
    default: hipLaunchKernelGGL((compose_pairs_kernel<MATCH_DOUBLY, WBX>), dim3(n), dim3(WBX), 0, st, d_args); break;
  }
}
}  // namespace
void launch_compose_pairs(const ComposeArgs* d_args, int n, int matcher, hipStream_t st) {
  if (n <= 0) return;
  if (n >= 64) launch_pairs_w<256>(d_args, n, matcher, st);
  else launch_pairs_w<1024>(d_args, n, matcher, st);
}

void launch_sorted_view(const DGraph& g, int key_olabel, void* out_view, void* in_view, hipStream_t st) {
  if (g.N <= 0 || g.A <= 0) return;
  hipLaunchKernelGGL(sorted_view_kernel, dim3(g.N), dim3(256), 0, st, g, key_olabel, static_cast<gtnx_i4*>(out_view),
                     static_cast<gtnx_i4*>(in_view));
}

} // namespace gtnx
