// maxplus.hip -- viterbiScore / viterbiPath over a never-built product chain o G when G is dense
// (ASG transitions, SURVEY.md section 8 config C4: N = 513 nodes, 262 k arcs; the product would hold
// 262 M arcs per utterance).  Replaces compose (compose.cpp:377-522) + shortestPath
// (shortest.cpp:190-272) for such products with
//
//   alpha[t+1][b][d] = ( max_s  alpha[t][b][s] + W[s][d] )  +  em[b][t][label(d)]
//
// one launch per time step over the whole batch: a max-plus "matrix product" [nb x N] (x) [N x N] on
// the vector ALUs, then ONE back-trace wave per utterance that re-derives the winning arc of each
// visited state from alpha (no back-pointer planes: they would be another (T+1) nb N words that the
// sweep has to write and nobody but T of them per utterance is ever read).
//
// Step kernel.  512 x 513 outputs per step is a small problem for 1024 SIMDs, so the tile is chosen for
// operand traffic, not for size: a wave owns 64 utterances (one per lane) x 16 destinations and a sixteenth
// of the sources (16 waves per workgroup: four per SIMD cover one another's scalar waits).  alpha[t] arrives TRANSPOSED ([s / 4][b][4], written by the previous step's epilogue):
// one 16-byte load per lane brings four sources.  W arrives through the SCALAR path: its 16 values per
// source are wave-uniform, so they are s_load'ed into SGPRs (pairs of sources interleaved, so that
// v_pk_add_f32 takes {x[s], x[s+1]} + {W[s][d], W[s+1][d]} with an SGPR pair as operand) and never touch
// LDS or VGPRs.  Per pair of product arcs: one v_pk_add_f32 + one v_max3_f32.  The sixteen partial tiles
// meet in LDS; each wave finishes one destination (emission added once per destination: all in-arcs
// of a node share their label in this regime).
//
// Ties.  The generic kernel (lazy.hip: lazy_step_kernel<SD_TROPICAL>) keeps the FIRST maximal in-arc in
// in-row order, comparing fl(fl(alpha + w) + em).  Rounding is monotone, so the sweep above produces the
// same alpha bit for bit, and the back-trace evaluates exactly that comparison over the visited node's
// real in-row (parallel arcs and missing arcs included) -- same arc, same path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr float NEG_INF = -__builtin_inff();
constexpr int MP_WAVES = 16;  // source split inside a workgroup
constexpr int MP_FIN = 1;     // destinations a wave finishes (MP_COLS / MP_WAVES)
constexpr int MP_COLS = 16;   // destinations per workgroup
static_assert(MP_FIN * MP_WAVES == MP_COLS, "every destination of the tile is finished by one wave");

typedef float mp_f2 __attribute__((ext_vector_type(2)));
typedef float mp_f16 __attribute__((ext_vector_type(16)));
// constant address space: uniform addresses become s_load
typedef const __attribute__((address_space(4))) mp_f2* mp_kc_f2;

__device__ __forceinline__ int mp_key(float x) {
  const int b = __float_as_int(x);
  return b >= 0 ? b : b ^ 0x7fffffff;
}

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Scalar loads return out of order, so every wait on them is lgkmcnt(0).  arrived(): the pair requested one
// compute block ago is there -- waited for BEFORE the next request is issued (the fences keep the
// scheduler from moving the request above the wait, or the reductions above the request).
__device__ __forceinline__ void arrived() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }

// ---- W in scalar-operand layout: [dblock][source pair][16 destinations][2 sources], keys first
__global__ void maxplus_wfill_kernel(int* Wq, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) Wq[i] = mp_key(NEG_INF);
}
__global__ void maxplus_wmax_kernel(LazyGroup g, int* Wq) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.g.A) return;
  const gtnx_i4 r = g.lrec_in[k];  // {source, matched label or -1, weight bits, arc}
  if (r.y < 0) return;
  const int j = g.mp_colidx[g.g.dst[r.w]];
  if (j < 0) return;
  const int64_t at = ((int64_t(j >> 4) * (g.Kpad >> 1) + (r.x >> 1)) * MP_COLS + (j & 15)) * 2 + (r.x & 1);
  atomicMax(&Wq[at], mp_key(__int_as_float(r.z)));  // parallel arcs: the larger weight
}

// alpha[0] in operand layout (plane 0), -inf everywhere in plane 1 (dead nodes and padding stay -inf)
__global__ void maxplus_init_kernel(LazyGroup g) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(g.Kpad) * g.nbpad) return;
  const int k = int(i / g.nbpad), b = int(i % g.nbpad);
  const int64_t at = (int64_t(k >> 2) * g.nbpad + b) * 4 + (k & 3);
  g.xt[0][at] = (k < g.N && (g.g.nflags[k] & NF_START)) ? 0.0f : NEG_INF;
  g.xt[1][at] = NEG_INF;
}

__global__ __launch_bounds__(MP_WAVES * 64) void maxplus_step_kernel(LazyGroup g, int t) {
  __shared__ float part[MP_WAVES][MP_COLS][64];
  const int l = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nslab = g.nbpad >> 6;
  // consecutive workgroups (one per XCD in turn) take consecutive utterance slabs: the workgroups of one
  // slab, which all stream the same 64 columns of the input plane, share an L2 when there are 8 slabs
  const int slab = blockIdx.x % nslab, dblk = blockIdx.x / nslab;
  const int b = slab * 64 + l;
  const bool on = b < g.nb;
  const int N = g.N, C = g.C, nbp = g.nbpad;
  // ---- the two destinations this wave finishes: their emission terms are requested now
  int dn[MP_FIN];
  float ev[MP_FIN];
  {
    const float* erow = g.em[on ? b : 0];
#pragma unroll
    for (int v = 0; v < MP_FIN; ++v) {
      const int j = dblk * MP_COLS + MP_FIN * wv + v;
      dn[v] = j < g.mp_ncol ? g.mp_colnode[j] : -1;
      const int lab = dn[v] >= 0 ? g.nlab[dn[v]] : -1;
      ev[v] = (on && lab >= 0) ? erow[int64_t(t) * C + lab] : 0.0f;
    }
  }
  // ---- this wave's eighth of the source groups (4 sources = 2 pairs each)
  const int G = g.Kpad >> 2;
  const int g_lo = (G * wv) / MP_WAVES, g_hi = (G * (wv + 1)) / MP_WAVES;
  const gtnx_f4* X = reinterpret_cast<const gtnx_f4*>(g.xt[t & 1]) + b;
  mp_kc_f2 W = (mp_kc_f2)(g.mp_Wq) + (int64_t(dblk) * (2 * G) + 2 * g_lo) * MP_COLS;
  float acc[MP_COLS];
#pragma unroll
  for (int r = 0; r < MP_COLS; ++r) acc[r] = NEG_INF;
  if (g_lo < g_hi) {
    gtnx_f4 x = X[int64_t(g_lo) * nbp];
    mp_f2 wa[MP_COLS], wb[MP_COLS];
#pragma unroll
    for (int r = 0; r < MP_COLS; ++r) wa[r] = W[r];
    for (int q = g_lo; q < g_hi; ++q) {
      // a pair's 32 scalars are awaited before the next pair is requested; each request then has one pair's 32
      // VALU instructions (and the SIMD's other wave) to land behind.  The W block carries one pair of
      // padding at its end, so the last request stays inside it.
      gtnx_f4 xn = X[int64_t(q + 1 < g_hi ? q + 1 : q) * nbp];
      arrived();
#pragma unroll
      for (int r = 0; r < MP_COLS; ++r) wb[r] = W[MP_COLS + r];
      fence();
      {
        const mp_f2 xv = {x.x, x.y};
        mp_f2 s[MP_COLS];  // all sums first: a packed add's result is not read by the very next instruction
#pragma unroll
        for (int r = 0; r < MP_COLS; ++r) s[r] = xv + wa[r];
#pragma unroll
        for (int r = 0; r < MP_COLS; ++r) acc[r] = max3(acc[r], s[r].x, s[r].y);
      }
      fence();
      arrived();
#pragma unroll
      for (int r = 0; r < MP_COLS; ++r) wa[r] = W[2 * MP_COLS + r];
      fence();
      {
        const mp_f2 xv = {x.z, x.w};
        mp_f2 s[MP_COLS];  // all sums first: a packed add's result is not read by the very next instruction
#pragma unroll
        for (int r = 0; r < MP_COLS; ++r) s[r] = xv + wb[r];
#pragma unroll
        for (int r = 0; r < MP_COLS; ++r) acc[r] = max3(acc[r], s[r].x, s[r].y);
      }
      fence();
      W += 2 * MP_COLS;
      // the next group's input was requested a whole group ago; naming it here keeps the request there
      asm volatile("" : "+v"(xn.x), "+v"(xn.y), "+v"(xn.z), "+v"(xn.w));
      x = xn;
    }
  }
#pragma unroll
  for (int r = 0; r < MP_COLS; ++r) part[wv][r][l] = acc[r];
  __syncthreads();
  const int64_t plane = int64_t(g.nb) * N;
  float* outp = g.alpha + int64_t(t + 1) * plane + int64_t(b) * N;
  float* Xn = g.xt[(t + 1) & 1];
#pragma unroll
  for (int v = 0; v < MP_FIN; ++v) {
    const int r = MP_FIN * wv + v;
    float m = part[0][r][l];
#pragma unroll
    for (int p = 1; p < MP_WAVES; ++p) m = __builtin_fmaxf(m, part[p][r][l]);
    if (on && dn[v] >= 0) {
      const float out = m + ev[v];
      outp[dn[v]] = out;
      Xn[(int64_t(dn[v] >> 2) * nbp + b) * 4 + (dn[v] & 3)] = out;
    }
  }
  // nodes without a matched in-arc (the start node of an ASG transitions graph): -inf from step 1 on
  if (dblk == 0 && on) {
    for (int i = wv; i < g.mp_ndead; i += MP_WAVES) {
      const int d = g.mp_dead[i];
      outp[d] = NEG_INF;
      Xn[(int64_t(d >> 2) * nbp + b) * 4 + (d & 3)] = NEG_INF;
    }
  }
}

// ---- back-trace (shortest.cpp:239-260): one wave per utterance.  A step's chain is kept to ONE trip to
// memory (the visited node's in-records): row offsets and labels of G sit in LDS, the alpha row and the
// emission row of the step after next are requested while the current step is reduced and parked in LDS,
// and the winner's source / arc come from the winning lane by cross-lane read instead of a second load.
// wave-wide reductions by DPP (result in every lane)
#define MP_DPP6(op)                                            \
  "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"  \
  op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
  op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
  op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
  op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"             \
  op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
__device__ __forceinline__ float wave_max63(float x) {
  asm volatile(MP_DPP6("v_max_f32_dpp") : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ int wave_min63(int x) {
  asm volatile(MP_DPP6("v_min_i32_dpp") : "+v"(x));
  return __builtin_amdgcn_readlane(x, 63);
}
constexpr int MP_EMROW = 1024;  // emission rows up to this many labels are staged (else one load per step)
template <int RMAX>  // 64-lane slices of the staged rows: ceil(max(N, staged C) / 64), rounded up to 4 / 8 / 12 / 16
__global__ __launch_bounds__(64) void maxplus_path_kernel(LazyGroup g, int* path_arc, int* path_il, int* path_ol,
                                                         float* path_w, int* path_len) {
  extern __shared__ float lds[];
  const int b = blockIdx.x, l = threadIdx.x;
  const int N = g.N, C = g.C, T = g.T;
  const bool stage_em = C <= MP_EMROW;
  float* rows = lds;                                   // [2][N] alpha[t-1] / alpha[t-2]
  float* erows = rows + 2 * N;                         // [2][C] (stage_em)
  int* ioff = reinterpret_cast<int*>(erows + (stage_em ? 2 * C : 0));  // [N + 1]
  int* nlab = ioff + N + 1;                            // [N]
  const int64_t plane = int64_t(g.nb) * N;
  int node = g.best[b];
  if (node < 0 || T < 1) {  // no accepting path: the trimmed product is the empty graph
    if (l == 0) path_len[b] = node < 0 ? -1 : 0;
    return;
  }
  for (int n = l; n <= N; n += 64) ioff[n] = g.g.in_off[n];
  for (int n = l; n < N; n += 64) nlab[n] = g.nlab[n];
  const GTNX_G float* em = (const GTNX_G float*)g.em[b];
  const float* arow = g.alpha + int64_t(b) * N;
  constexpr int RB = 9;  // records per lane per batch: 576 cover the 513-arc in-rows of C4 in ONE trip to memory
  struct Rows {
    float a[RMAX], e[RMAX];
  };
  // (always valid addresses, no branches: a conditional request would turn the counted wait for the
  // records, which are requested BEFORE these rows and so return before them, into a wait for everything)
  auto fetch = [&](Rows& r, int t) {  // alpha[t] and the emission row of step t
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int n = l + 64 * i;
      r.a[i] = arow[int64_t(t) * plane + (n < N ? n : N - 1)];
      r.e[i] = stage_em ? em[int64_t(t) * C + (n < C ? n : C - 1)] : 0.0f;
    }
  };
  auto park = [&](const Rows& r, int buf) {
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      const int n = l + 64 * i;
      if (n < N) rows[buf * N + n] = r.a[i];
      if (stage_em && n < C) erows[buf * C + n] = r.e[i];
    }
  };
  // Rows are requested TWO steps before they are parked (a step's own chain is shorter than a trip to HBM),
  // into two register sets that alternate: step t requests the rows of step t - 3 and parks those of t - 2.
  Rows r0, r1;
  fetch(r0, T - 1);
  park(r0, (T - 1) & 1);
  fetch(r1, T >= 2 ? T - 2 : 0);
  __syncthreads();
  bool failed = false;
  int tied = 0;
  {  // two accept states with the best score: which one the reference takes depends on the built product's accept
     // order (shortest.cpp:226-237) -- a tie like any other
    const float top = g.score[b];
    const float* last = g.alpha + int64_t(T) * plane + int64_t(b) * N;
    int hits = 0;
    for (int k = l; k < g.g.n_accept; k += 64) hits += (last[g.g.accept_list[k]] == top && top > NEG_INF) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
    if (hits > 1) tied = 1;
  }
  int keep_arc = 0, keep_lab = 0;
  float keep_e = 0.0f;
  const bool by_node = g.tie_by_node != 0;  // (uniform)
  auto step = [&](int t, Rows& rq, const Rows& rp) {  // rq: set to request into, rp: set to park
    const float* prev = rows + ((t - 1) & 1) * N;
    const int lab = nlab[node];  // every matched in-arc of `node` carries this label
    const float e = lab < 0 ? 0.0f : (stage_em ? erows[((t - 1) & 1) * C + lab] : em[int64_t(t - 1) * C + lab]);
    const int k0 = ioff[node], k1 = ioff[node + 1];
    float m = NEG_INF;
    int arg = INT_MAX, bsrc = 0, barc = 0, same = 0;  // same: records of this lane that hold its maximum
    // the in-row, nine records per lane at a time, all requested before the first is looked at (one
    // exposed trip to memory per 576 records, not one per record); the rows of the step after next queue
    // up behind the first batch
    for (int kb = k0; kb == k0 || kb < k1; kb += 64 * RB) {
      gtnx_i4 r[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int k = kb + l + 64 * i;
        r[i] = g.lrec_in[k < k1 ? k : (k1 > 0 ? k1 - 1 : 0)];
      }
      if (kb == k0) {
        __builtin_amdgcn_sched_barrier(0);
        fetch(rq, t >= 3 ? t - 3 : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int k = kb + l + 64 * i;
        if (k < k1 && r[i].y >= 0) {
          const float x = prev[r[i].x] + __int_as_float(r[i].z) + e;
          same = x > m ? 1 : (x == m ? same + 1 : same);
          // of equal maxima the one from the SMALLEST source node (then the earliest record): for a transitions
          // graph whose every node reaches every node, lists in node order, that is the order the reference's queue
          // visits the sources in every layer (ops_lazy.cpp: dense_ties_by_node_order) -- its own tie-break
          if (x > m || (by_node && x == m && r[i].x < bsrc)) {
            m = x;
            arg = k;
            bsrc = r[i].x;
            barc = r[i].w;
          }
        }
      }
    }
    // first maximum in in-row order: the wave's maximum, then the smallest record index holding it
    const float mx = wave_max63(m);
    // two equal finite candidates at a state of the best path: the reference's choice depends on the order its
    // queue reached the sources (shortest.cpp:215-218), which a product that is never built does not have --
    // reported (path_len[nb + b]); the walk keeps the first maximum in in-row order
    {
      const bool at = m == mx && mx > NEG_INF;
      const unsigned long long holders = __builtin_amdgcn_ballot_w64(at);
      const unsigned long long many = __builtin_amdgcn_ballot_w64(at && same > 1);
      if (__builtin_popcountll(holders) > 1 || many != 0ull) tied = 1;
    }
    // by node: (source node, lane) -- the lane's own best is already its smallest source; 64 lanes, sources below
    // 2^24.  Else: the smallest record index holding the maximum (record k sits with lane (k - k0) mod 64)
    arg = wave_min63((m == mx && arg != INT_MAX) ? (by_node ? ((bsrc << 6) | l) : arg) : INT_MAX);
    if (arg == INT_MAX) {  // cannot happen below a finite best score
      failed = true;
      return;
    }
    const int wl = by_node ? (arg & 63) : ((arg - k0) & 63);
    bsrc = __builtin_amdgcn_readlane(bsrc, wl);
    barc = __builtin_amdgcn_readlane(barc, wl);
    // the step's arc goes to the lane that owns path index t - 1 (selects: a lane-0 store here would put a
    // branch join between the requests above and the counted wait of park() below); every 64 steps the lanes
    // write their entries out together
    const bool mine = ((t - 1) & 63) == l;
    keep_arc = mine ? barc : keep_arc;
    keep_lab = mine ? lab : keep_lab;
    keep_e = mine ? e : keep_e;
    node = bsrc;
    park(rp, t & 1);  // step t-2's rows -> the buffers step t's sat in ((t - 2) & 1 == t & 1)
    __syncthreads();
    if (((t - 1) & 63) == 0) {
      const int idx = (t - 1) + l;
      if (idx < T) {
        const int64_t o = int64_t(b) * T + idx;
        path_arc[o] = keep_arc;
        path_il[o] = g.chain_first ? keep_lab : g.g.il[keep_arc];
        path_ol[o] = g.chain_first ? g.g.ol[keep_arc] : keep_lab;
        path_w[o] = g.g.w[keep_arc] + keep_e;
      }
    }
  };
  for (int t = T; t >= 1 && !failed; t -= 2) {
    step(t, r0, r1);
    if (t >= 2 && !failed) step(t - 1, r1, r0);
  }
  if (failed) {
    if (l == 0) path_len[b] = -1;
    return;
  }
  if (l == 0) {
    path_len[b] = T;
    path_len[g.nb + b] = tied;
  }
}

}  // namespace

size_t maxplus_w_floats(const LazyGroup& g) {
  const size_t dblocks = size_t((g.mp_ncol + MP_COLS - 1) / MP_COLS);
  return (dblocks * size_t(g.Kpad >> 1) + 1) * MP_COLS * 2;  // + one pair of padding (prefetch)
}

void launch_maxplus_prep(const LazyGroup& g, hipStream_t st) {
  const int64_t n = int64_t(maxplus_w_floats(g));
  int* keys = reinterpret_cast<int*>(const_cast<float*>(g.mp_Wq));
  hipLaunchKernelGGL(maxplus_wfill_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, keys, n);
  if (g.g.A > 0) hipLaunchKernelGGL(maxplus_wmax_kernel, dim3((g.g.A + 255) / 256), dim3(256), 0, st, g, keys);
  launch_lazy_mfma_keys(const_cast<float*>(g.mp_Wq), n, st);  // keys -> floats
  const int64_t m = int64_t(g.Kpad) * g.nbpad;
  hipLaunchKernelGGL(maxplus_init_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, st, g);
}

void launch_maxplus_step(const LazyGroup& g, int t, hipStream_t st) {
  const int dblocks = (g.mp_ncol + MP_COLS - 1) / MP_COLS, nslab = g.nbpad >> 6;
  if (dblocks <= 0 || nslab <= 0) return;
  hipLaunchKernelGGL(maxplus_step_kernel, dim3(dblocks * nslab), dim3(MP_WAVES * 64), 0, st, g, t);
}

void launch_maxplus_path(const LazyGroup& g, int* path_arc, int* path_il, int* path_ol, float* path_w, int* path_len,
                         hipStream_t st) {
  if (g.nb <= 0) return;
  const size_t lds = sizeof(float) * (size_t(4) * g.N + 1 + (g.C <= MP_EMROW ? size_t(2) * g.C : 0));
  // (every staged slice is a load per lane per step: 513 nodes and 512 labels need 9, not 16)
  const int slices = (std::max(g.N, g.C <= MP_EMROW ? g.C : 0) + 63) / 64;
  if (slices <= 4) hipLaunchKernelGGL(maxplus_path_kernel<4>, dim3(g.nb), dim3(64), lds, st, g, path_arc, path_il, path_ol, path_w, path_len);
  else if (slices <= 8) hipLaunchKernelGGL(maxplus_path_kernel<8>, dim3(g.nb), dim3(64), lds, st, g, path_arc, path_il, path_ol, path_w, path_len);
  else if (slices <= 10) hipLaunchKernelGGL(maxplus_path_kernel<10>, dim3(g.nb), dim3(64), lds, st, g, path_arc, path_il, path_ol, path_w, path_len);
  else if (slices <= 12) hipLaunchKernelGGL(maxplus_path_kernel<12>, dim3(g.nb), dim3(64), lds, st, g, path_arc, path_il, path_ol, path_w, path_len);
  else hipLaunchKernelGGL(maxplus_path_kernel<16>, dim3(g.nb), dim3(64), lds, st, g, path_arc, path_il, path_ol, path_w, path_len);
}

}  // namespace gtnx
