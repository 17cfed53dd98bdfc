// band.hip -- forwardScore of chain o G (never built) and its gradient when G is BANDED:
// every arc of G goes from node n to n, n+1 or n+2, at most one arc per (n, step), and all
// in-arcs of a node carry one matched label.  That is the CTC target acceptor
// (benchmarks/ctc.cpp:40-58, examples/ctc.cpp:21-41) and the force-alignment acceptor of
// examples/asg.cpp:50-57; any other G takes lazy_pair.hip.  What the kernels compute is
// shortest.cpp:86-170 over the product compose.cpp:377-522 would build, and
// shortest.cpp:33-62 + compose.cpp:496-518 for the gradient; with `norm` set the same
// launch also computes forwardScore of the chain itself (functions.cpp:320-322 over
// creations.cpp:20-33) and the backward launch adds its softmax gradient, so every
// emission is read once per sweep and every gradient row is written exactly once.
//
// One workgroup per utterance, its waves split into ROLES.  Forward: 8 waves -- 4 SWEEPERS run the recursion (wave
// w owns nodes [64w NPL, 64(w+1) NPL), NPL = 1 or 2 nodes per lane) and 4 HELPERS stage emission rows HBM -> registers
// -> LDS, compute the rows' log-sum-exps (the chain's own forwardScore) and, on a region's first sweep, copy the
// emissions to where the backward sweep reads them.  Backward: 12 waves -- 4 sweepers, 4 STAGERS (waves 4, 5: emission
// chunks; 6, 7: alpha rows and the rows' scalars) and 4 DRAINERS (gather the node posteriors by label, add the
// normaliser's softmax term, store each gradient row once, coalesced).  Two workgroups per CU (<= 78 KB of LDS each,
// <= 80 registers per lane in the backward kernel).
// The recursion over time is a BLOCK-SKEWED PIPELINE: a node needs values of lower-numbered nodes only (higher-
// numbered in the backward sweep), so sweeper w runs one block of K time steps (K = 8 forward, 4 backward at C3)
// behind sweeper w-1 and finds the boundary values of a whole block in LDS, written a tick earlier.  Inside a tick a
// sweeper runs its K steps with no synchronisation at all (neighbours move by DPP wave shifts; the block's emission
// gathers, boundary values and alpha rows are fetched from LDS up front); one LDS-only barrier per tick keeps the
// roles in step.  Nothing on the critical path waits for LDS or HBM latency:
//   * emission rows live in an LDS ring (NBE = 5 blocks forward, NBGE = 8 backward), landed by the helper / staging
//     waves from 16-byte loads requested two ticks ahead; the sweepers never issue a global load;
//   * scores are in log2 units (bare v_exp_f32 / v_log_f32), "minus infinity" is -1e30 so no step needs an inf / NaN
//     guard, and every RN rows each sweeper shifts ITS nodes by their maximum (per-wave shifts summed in fp64;
//     boundary values are re-based when they cross waves): magnitudes stay O(10) and float32 keeps ~1e-6 relative
//     accuracy on every posterior for any T;
//   * the backward recursion works on POSTERIORS: 2^(alpha[t][n] + y_k - score) is the posterior of arc k leaving
//     (t, n), the three of a node sum to the node's posterior S, and beta[t][n] = log2 S - (alpha[t][n] - score) --
//     three exponentials and a logarithm per node and step, no shift by a maximum (a posterior is at most 1; one
//     below 2^-126 is dropped, and with it at most that much of the total mass);
//   * the backward sweepers write node posteriors into an LDS ring with plain stores; when the last sweeper has left a
//     block its rows are summed (each row's posteriors are rescaled to their exact total: at four rows per block by
//     ONE wave, sixteen lanes per row -- the emission-staging wave that has no chunk to land that tick) and the
//     drainers write the gradient rows; G's arc gradients are register accumulators of the sweepers.
// The sweeps are priced in INSTRUCTIONS: a wave alone issues one vector instruction per ~8.5 cycles, a
// v_cndmask_b32 costs a SIMD 2.6 plain ones, packed float32 arithmetic is free width -- hence no select on a
// wave-uniform condition in a steady tick, steady ticks compiled without range tests in every role, packed adds and
// multiplies wherever two values share an instruction.
// HBM traffic per utterance: forward 4TC + 4(T+1)NS, backward 8TC + 4(T+1)NS (+ G).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

#include "kernels.h"

#ifdef GTNX_BAND_TIMING
__device__ long long g_band_timing[256];
#endif

namespace gtnx {
namespace {

constexpr float NEGF = -1.0e30f;   // log-domain zero
constexpr float DEADF = -1.0e29f;  // anything below is "no path"
constexpr float LOG2E = 1.44269504088896340736f;
constexpr double LN2 = 0.693147180559945309417;
constexpr int BW = 256;  // lanes of one role: 4 sweeper waves + 4 helper waves per workgroup
constexpr int WG = 512;
constexpr int WGB = 768;  // the backward sweep: 4 sweeper + 4 staging + 4 draining waves
constexpr int NBE = 5;   // blocks in the emission / alpha rings: one landing, four in use (lead .. last wave)
constexpr int NBG = 7;   // blocks the backward sweep keeps: one landing, four in use, one being summed, one draining
constexpr int NBGE = 8;  // ... of emissions: they land a tick before the rest of their chunk
#ifndef GTNX_BWD_SPLIT
#define GTNX_BWD_SPLIT 0  // (1: measured slower, see band_backward_kernel)
#endif
#ifndef GTNX_FWD_COPY_DEPTH
#define GTNX_FWD_COPY_DEPTH 2
#endif
#ifndef GTNX_FWD_COPY_ROTATE
#define GTNX_FWD_COPY_ROTATE 1
#endif
#ifndef GTNX_FWD_DEPTH
#define GTNX_FWD_DEPTH 2
#endif

#ifdef GTNX_BAND_TIMING
// diagnostic build (tools/ubench/band_bench.hip): cycles per phase of a tick, wave 0 of workgroup 0
#define GTNX_TM(slot)                 \
  do {                                \
    const long long now_ = clock64(); \
    tm_acc[slot] += now_ - tm_last;   \
    tm_last = now_;                   \
  } while (0)
#define GTNX_TM_INIT(base)                       \
  const int tm_base = base;                      \
  long long tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  long long tm_n = 0;                            \
  long long tm_last = clock64()
#define GTNX_TM_TICK() tm_n += 1
#define GTNX_TM_DUMP()                                                      \
  do {                                                                      \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                       \
      for (int i_ = 0; i_ < 7; ++i_) g_band_timing[tm_base + (threadIdx.x >> 6) * 7 + i_] += tm_acc[i_]; \
      if (threadIdx.x == 0) g_band_timing[tm_base + 127] += tm_n;           \
    }                                                                       \
  } while (0)
#else
#define GTNX_TM(slot) do { } while (0)
#define GTNX_TM_INIT(base) do { } while (0)
#define GTNX_TM_TICK() do { } while (0)
#define GTNX_TM_DUMP() do { } while (0)
#endif

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef GTNX_EXP_FAKE_MATH  // (tools/ubench experiment, wrong results: what the sweeps cost with full-rate stand-ins for the transcendentals)
__device__ __forceinline__ float ex2(float x) { return __builtin_fmaf(x, 0.001f, 1.0f); }
__device__ __forceinline__ float lg2(float x) { return __builtin_fmaf(x, 0.001f, -0.001f); }
#else
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
#endif

// A value that came from a global load and is first used inside the main loop would make the
// compiler put its `s_waitcnt vmcnt` THERE -- where it also waits for every global store issued
// since (stores count in vmcnt on gfx9): one store round trip per gradient row.  settle() is a
// use in front of the loop; uniform() does the same and moves a wave-uniform value to an SGPR.
__device__ __forceinline__ void settle(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void settle(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float uniform(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double uniform(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(b)), hi = __builtin_amdgcn_readfirstlane(unsigned(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// lane i <- lane i-1 (lane 0 takes `fill`) / lane i <- lane i+1 (lane 63 takes `fill`)
__device__ __forceinline__ float wave_shr1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
// wave64 reductions: six DPP ops (a lane without a source keeps its value), total in lane 63
#define GTNX_DPP6(op)                                                         \
  "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  \
  "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"  \
  "s_nop 1"
__device__ __forceinline__ float wave_max(float x) {  // uniform result
  asm volatile(GTNX_DPP6("v_max_f32_dpp") : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_sum63(float x) {  // lane 63 holds the total
  asm volatile(GTNX_DPP6("v_add_f32_dpp") : "+v"(x));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum63(x)), 63));
}
// two wave64 sums at once (the two chains fill each other's DPP wait states); totals in lane 63
#define GTNX_DPP2(ctl)                                                  \
  "v_add_f32_dpp %0, %0, %0 " ctl "\n\tv_add_f32_dpp %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
__device__ __forceinline__ void wave_sum63x2(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               GTNX_DPP2("row_shr:1 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:2 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:4 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:8 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_bcast:15 row_mask:0xa bank_mask:0xf")
               GTNX_DPP2("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b));
}
// the same inside every aligned group of 16 lanes (one DPP row): totals in lanes 15, 31, 47, 63
__device__ __forceinline__ void row_sum15x2(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               GTNX_DPP2("row_shr:1 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:2 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:4 row_mask:0xf bank_mask:0xf")
               GTNX_DPP2("row_shr:8 row_mask:0xf bank_mask:0xf")
               : "+v"(a), "+v"(b));
}
// all-reduce inside every aligned group of 16 lanes (one DPP row) by rotations
#define GTNX_ROR4(op)                                                        \
  "s_nop 1\n\t" op " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"    \
  "s_nop 1\n\t" op " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"    \
  "s_nop 1\n\t" op " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"    \
  "s_nop 1\n\t" op " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"    \
  "s_nop 1"
__device__ __forceinline__ float row16_max(float x) {
  asm volatile(GTNX_ROR4("v_max_f32_dpp") : "+v"(x));
  return x;
}
__device__ __forceinline__ float row16_sum(float x) {
  asm volatile(GTNX_ROR4("v_add_f32_dpp") : "+v"(x));
  return x;
}
// log2(2^x0 + 2^x1 + 2^x2): the largest term is exactly 1, so two v_exp_f32 and one v_log_f32
__device__ __forceinline__ float lse3(float x0, float x1, float x2) {
  const float mx = fmaxf(fmaxf(x0, x1), x2);
  const float md = __builtin_amdgcn_fmed3f(x0, x1, x2);
  const float mn = fminf(fminf(x0, x1), x2);
  return mx + lg2(1.0f + ex2(md - mx) + ex2(mn - mx));
}

// what a lane knows about its NPL nodes
template <int NPL>
struct NodeRegs {
  int lab[NPL];      // matched label (0 when the node has no in-arc; its weights are NEGF then)
  float wi[3][NPL];  // in-arc from n-k, log2 units (NEGF: no such arc)
  float wo[3][NPL];  // out-arc to n+k
  int ao[3][NPL];    // arc ids of the out-arcs (-1: none)
  bool has_in[NPL];
  bool start[NPL], accept[NPL];
};
template <int NPL, bool WANT_OUT>
__device__ __forceinline__ void load_nodes(const BandPair& P, int m0, NodeRegs<NPL>& g) {
  const GTNX_G gtnx_i4* nodes = reinterpret_cast<const GTNX_G gtnx_i4*>(P.nodes);
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int m = m0 + j;
    g.lab[j] = 0;
    g.start[j] = g.accept[j] = g.has_in[j] = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g.wi[k][j] = g.wo[k][j] = NEGF;
      g.ao[k][j] = -1;
    }
    if (m < P.N) {
      const gtnx_i4 q = nodes[m];  // {label, arc from m, arc from m-1, arc from m-2}
      g.lab[j] = q.x >= 0 ? q.x : 0;
      const uint8_t f = P.nflags[m];
      g.start[j] = (f & NF_START) != 0;
      g.accept[j] = (f & NF_ACCEPT) != 0;
      const int a[3] = {q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (a[k] >= 0) {
          g.has_in[j] = true;
          g.wi[k][j] = P.w ? fmaxf(P.w[a[k]] * LOG2E, NEGF) : 0.0f;
        }
    }
    if (WANT_OUT) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (m + k < P.N) {
          const gtnx_i4 q = nodes[m + k];
          const int a = k == 0 ? q.y : (k == 1 ? q.z : q.w);
          if (a >= 0) {
            g.ao[k][j] = a;
            g.wo[k][j] = P.w ? fmaxf(P.w[a] * LOG2E, NEGF) : 0.0f;
          }
        }
      }
    }
  }
}

// emissions land in log2 units with -inf clamped to the finite log-zero
__device__ __forceinline__ float em2(float x) { return fmaxf(x * LOG2E, NEGF); }

// rows of a chunk are contiguous in HBM: element e (0 .. rows * W) -> row e / W; W <= 4096, e <= 4096
__device__ __forceinline__ int row_of(int e, int W, float invW) {
  int r = int(float(e) * invW);
  if (r * W > e) --r;
  if ((r + 1) * W <= e) ++r;
  return r;
}

// A chunk of `rows` contiguous HBM rows of W floats each, staged through NS floats per lane of
// a group of LN lanes: issue() starts the loads, each() hands them back with their coordinates.
template <int NS, int LN = BW>
struct Stage {
  float v[NS];
  // loads are unconditional per lane (lanes past the end re-read the last element): a load under
  // a lane mask would have to be merged with its default, i.e. waited for, right where it is
  // issued.  Whole rounds past the end are skipped (uniform).  VEC is a compile-time choice: with
  // both forms in one kernel the second would wait for the first (they write the same registers)
  template <bool VEC>
  __device__ __forceinline__ void issue(const GTNX_G float* src, int cnt, int tid) {
    if (cnt <= 0) return;  // uniform
    // Both modes fetch the chunk -- K rows, contiguous in HBM -- with 16-byte loads (global_load_dwordx4 needs
    // dword alignment only).  VEC: rows are a multiple of 4 floats and the source is 16-byte aligned, so a load
    // never straddles two rows and lands with one 16-byte LDS store at a precomputed offset.  !VEC (any
    // alphabet, any alignment): the same loads, landed element by element with the row found per element.
    // (C = 255 used to take one 4-byte load per element: 2.6x the backward sweep's time at C = 256.)
    // (cnt >= 4: the band path takes alphabets of at least 4 labels -- band_min_labels())
#pragma unroll
    for (int i = 0; i < NS / 4; ++i) {
      if (i > 0 && 4 * i * LN >= cnt) break;  // uniform
      const int e = min(4 * (i * LN + tid), cnt - 4);
      // (e >= 0: a 32-bit BYTE offset beside the uniform base -- no 64-bit lane arithmetic, no register pair per slot)
      const gtnx_f4 q = *reinterpret_cast<const GTNX_G gtnx_f4*>(reinterpret_cast<const GTNX_G char*>(src) + unsigned(4 * e));
      v[4 * i] = q.x;
      v[4 * i + 1] = q.y;
      v[4 * i + 2] = q.z;
      v[4 * i + 3] = q.w;
    }
  }
  // VEC mode: where slot i of a FULL chunk of K rows lands in a ring block whose rows are `stride`
  // floats apart and in reverse order (chunk row r -> block row K-1-r), computed once: a chunk of
  // fewer rows sits (K - rows) block rows lower
  // (reverse = false: the forward sweep's ring, chunk row r -> block row r)
  int off[NS / 4];
  __device__ __forceinline__ void init_offsets(int W, int stride, int K, int tid, bool reverse = true) {
    const float invW = 1.0f / float(W);
#pragma unroll
    for (int i = 0; i < NS / 4; ++i) {
      const int e = 4 * (i * LN + tid);
      const int r = row_of(min(e, 4096), W, invW);
      off[i] = (reverse ? K - 1 - r : r) * stride + (e - r * W);
    }
  }
  // f(i, ring offset, values) for the slots of a chunk of `rows` rows of W floats
  template <class F>
  __device__ __forceinline__ void each_vec(int cnt, int shift, int tid, F&& f) const {
#pragma unroll
    for (int i = 0; i < NS / 4; ++i) {
      if (i > 0 && 4 * i * LN >= cnt) break;  // uniform
      if (4 * (i * LN + tid) < cnt) f(i, off[i] - shift, gtnx_f4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]});
    }
  }
  // a use of every staging register: all of this wave's loads are waited for HERE, and the
  // compiler knows that none is pending when the next issue() reuses the registers
  __device__ __forceinline__ void settle_all() {
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("" : "+v"(v[i]));
  }
  // VEC: f(i, e, r, c, q) per 16-byte slot -- slot index, element, chunk row, column, the four values.
  // !VEC: f(i, e, r, c, {x, 0, 0, 0}) per ELEMENT (a slot may straddle rows; the last slot of a chunk whose size
  // is not a multiple of 4 was fetched from cnt - 4 and lands there: the overlap rewrites the same values)
  template <bool VEC, class F>
  __device__ __forceinline__ void each(int cnt, int W, int tid, F&& f) const {
    const float invW = 1.0f / float(W);
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < NS / 4; ++i) {
        if (i > 0 && 4 * i * LN >= cnt) break;  // uniform
        const int e = 4 * (i * LN + tid);
        if (e < cnt) {
          const int r = row_of(e, W, invW);
          f(i, e, r, e - r * W, gtnx_f4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]});
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS / 4; ++i) {
        if (i > 0 && 4 * i * LN >= cnt) break;  // uniform
        if (4 * (i * LN + tid) < cnt) {
          const int e = min(4 * (i * LN + tid), cnt - 4);
          const int r0 = row_of(e, W, invW), c0 = e - r0 * W;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // (W >= 4: an element is in row r0 or the next one.  No loop in here: anything the compiler cannot
            //  unroll turns v[] into a runtime-indexed array in scratch memory)
            int r = r0, col = c0 + k;
            if (col >= W) {
              col -= W;
              ++r;
            }
            f(i, e + k, r, col, gtnx_f4{v[4 * i + k], 0.0f, 0.0f, 0.0f});
          }
        }
      }
    }
  }
};

// LDS layout shared by host (size) and device (carving); all counts in floats
struct BandLds {
  int CS;  // ring row stride of emission rows
  int o_ering, o_aring, o_oring, o_snode, o_misc, total;
};
__host__ __device__ inline BandLds band_lds(int C, int K, int NSmax, bool backward) {
  BandLds L;
  L.CS = C + 4;
  int o = 0;
  L.o_ering = o;  // the backward sweep keeps a block's emissions until its gradient rows are out
  o += (backward ? NBGE : NBE) * K * L.CS;
  o = (o + 3) & ~3;
  L.o_aring = o;  // alpha rows
  o += backward ? NBE * K * NSmax : 0;
  o = (o + 3) & ~3;
  L.o_oring = o;  // node posteriors (>= 2C ints: the prologue sorts labels here)
  o += backward ? max(NBG * K * NSmax, 2 * C) : 0;
  o = (o + 3) & ~3;
  L.o_snode = o;  // nodes sorted by (label, node)
  o += backward ? 512 : 0;
  // doubles: offr[5][16], red[16], find[4], aofr[NBE][K][4] | floats: bndr[5][4K][2], fin[8], lsep[2][16][8][2] / lser[NBG][K]
  // (region 0 of offr / bndr is constant: what the first wave of the pipeline reads as its neighbour)
  L.o_misc = o;
  o += 160 + 32 + 8 + 2 * NBE * K * 4 + 40 * K + 8 + 512;
  L.total = o;
  return L;
}
struct BandMisc {
  double *offr, *red, *find, *aofr;
  float *bndr, *fin, *lsep;
};
template <int K>
__device__ __forceinline__ BandMisc band_misc(float* lds, const BandLds& L) {
  BandMisc m;
  m.offr = reinterpret_cast<double*>(lds + L.o_misc);
  m.red = m.offr + 80;
  m.find = m.red + 16;
  m.aofr = m.find + 4;
  m.bndr = lds + L.o_misc + 200 + 2 * NBE * K * 4;
  m.fin = m.bndr + 40 * K;
  m.lsep = m.fin + 8;
  return m;
}

// shift of a wave's running row by its maximum; the reduction is independent of the next
// steps' chain, so the scheduler may interleave it with them
__device__ __forceinline__ float wave_max_of(const float* v, int n) {
  float mx = v[0];
  for (int j = 1; j < n; ++j) mx = fmaxf(mx, v[j]);
  return wave_max(mx);
}

// ==========================================================================================
// forward: alpha[t+1][m] = em[t][lab m] + log sum_k exp(alpha[t][m-k] + w_k(m))
// ==========================================================================================
template <int NPL, bool UNIT, int K, bool VEC>
__global__ __launch_bounds__(WG) void band_forward_kernel(const BandPair* __restrict__ pairs, int NSmax) {
#define GTNX_BAND_PAIR pairs[blockIdx.x]
#include "band_forward_body.inc"
#undef GTNX_BAND_PAIR
}
// the same sweep with THE pair as the kernel's own argument: a single utterance through the per-graph functions is a
// chain of dependent operations, and a 160-byte table copied to the device is one more of them
template <int NPL, bool UNIT, int K, bool VEC>
__global__ __launch_bounds__(WG) void band_forward_one_kernel(BandPair one, int NSmax) {
#define GTNX_BAND_PAIR one
#include "band_forward_body.inc"
#undef GTNX_BAND_PAIR
}

// ==========================================================================================
// backward: beta[t][n] = log sum_k exp(w_k + em[t][lab(n+k)] + beta[t+1][n+k]);
//   d score / d em[t][l]   = sum over nodes m with label l of exp(alpha[t+1][m] + beta[t+1][m] - score)
//   d score / d w(n->n+k)  = sum_t exp(alpha[t][n] + w + em[t][lab(n+k)] + beta[t+1][n+k] - score)
// Virtual row v = T-1-t ascends with the ticks; wave 3 leads.  LDS float atomics retire about
// one lane per 3 clocks per CU (tools/ubench/lat.hip), so nothing is scattered: every node
// writes its posterior to an LDS ring with a plain store, and when the last wave has left a
// block, helper lane c GATHERS the nodes that carry label c (lists sorted by label come with
// the pair) and stores the finished gradient element -- coalesced, once.
// ==========================================================================================
// the pair of a backward launch over the forward launch's table (kernels.h: BandPatch)
__device__ __forceinline__ BandPair band_patched(BandPair P, const BandPatch& pt) {
  if (pt.on) {  // (uniform)
    const int b = P.bidx;
    P.delta = pt.delta + b;
    P.delta_norm = pt.delta_norm ? pt.delta_norm + b : nullptr;
    P.rowlse = pt.rowlse ? pt.rowlse + int64_t(b) * pt.M : nullptr;
    P.norm = nullptr;
    P.em_copy = nullptr;
    P.grad_em = pt.grad_em ? pt.grad_em + int64_t(b) * pt.A : nullptr;
    P.grad_fixed = pt.grad_fixed ? pt.grad_fixed + P.goff : nullptr;
  }
  return P;
}
// BIG: a block of K rows holds more than 1024 emissions (K = 2, C > 512): 32 staging registers per
// helper lane and stream instead of 16 -- such shapes run one workgroup per CU (LDS), so the
// register budget is 256 there and 128 otherwise
template <int NPL, bool UNIT, bool GRADG, int K, bool VEC, bool BIG>
__global__ __launch_bounds__(WGB, BIG ? 3 : 6) void band_backward_kernel(const BandPair* __restrict__ pairs, int NSmax, BandPatch patch) {
#define GTNX_BAND_PAIR band_patched(pairs[blockIdx.x], patch)
#include "band_backward_body.inc"
#undef GTNX_BAND_PAIR
}
template <int NPL, bool UNIT, bool GRADG, int K, bool VEC, bool BIG>
__global__ __launch_bounds__(WGB, BIG ? 3 : 6) void band_backward_one_kernel(BandPair one, int NSmax) {
#define GTNX_BAND_PAIR one
#include "band_backward_body.inc"
#undef GTNX_BAND_PAIR
}


// ==========================================================================================
// CTC target acceptors built where they are used: the graph of benchmarks/ctc.cpp:40-58 /
// examples/ctc.cpp:21-41 (2U+1 nodes, blank at even nodes, self-loop + arc from the previous
// node everywhere, a skip arc between different labels; arc ids in the reference's addArc
// order) as the band records the sweeps read -- one workgroup per label sequence, nothing but
// the labels crosses PCIe.  Lists sorted by (label, node) as band_info() makes them on the host.
// ==========================================================================================
__global__ __launch_bounds__(512) void ctc_targets_kernel(const CtcTargetArgs* __restrict__ args, int blank) {
  const CtcTargetArgs a = args[blockIdx.x];
  // the label sequence and its skip flags (label i differs from label i - 1), padded to whole 16-byte reads
  __shared__ __attribute__((aligned(16))) int tl[256 + 4];
  __shared__ __attribute__((aligned(16))) int tk[256 + 4];
  const int N = a.N, U = (N - 1) >> 1, m = threadIdx.x;
  if (m < 260) {
    const int v = m < U ? a.labels[m] : 0x7fffffff;  // (padding: above every real label, no skip arc)
    tl[m] = v;
    tk[m] = (m > 0 && m < U && v != a.labels[m - 1]) ? 1 : 0;
  }
  __syncthreads();
  if (m >= N) return;
  const bool odd = (m & 1) != 0;
  const int my = odd ? tl[(m - 1) >> 1] : blank;
  const int skip = odd ? tk[(m - 1) >> 1] : 0;
  // Rank among the (label, node) pairs and first arc id.  Node 2i + 1 carries label i, every even node the blank:
  // the nodes in front of (my, m) are the label nodes with a smaller label (or mine, earlier) -- one pass over the
  // U labels, four per LDS read, where rounds 2-5 compared against all N nodes -- plus a closed form for the blanks.
  // The arcs in front of node m: one self loop per node, one step arc per node but the first, the skips counted
  // in the same pass.
  const int mi = (m - 1) >> 1;  // (odd m: my own label index; even m: labels 0 .. m/2 - 1 lie in front of me)
  const int before = odd ? mi : (m >> 1);  // label indices < before belong to nodes in front of m
  int rank = 0, skips = 0;
  for (int o = 0; o < U; o += 4) {
    const gtnx_i4 l4 = *reinterpret_cast<const gtnx_i4*>(tl + o), k4 = *reinterpret_cast<const gtnx_i4*>(tk + o);
    const int lo[4] = {l4.x, l4.y, l4.z, l4.w}, ko[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool front = o + k < before;
      rank += (lo[k] < my || (lo[k] == my && front)) ? 1 : 0;
      skips += front ? ko[k] : 0;
    }
  }
  // the blank nodes (even, U + 1 of them): all in front of a larger label, the earlier ones in front of an equal one
  rank += blank < my ? U + 1 : (blank == my ? (m + 1) >> 1 : 0);
  const int base = (m > 0 ? 2 * m - 1 : 0) + skips;
  GTNX_G BandNode* nd = const_cast<GTNX_G BandNode*>(a.nodes);
  nd[m].lab = my;
  nd[m].aid[0] = base;
  nd[m].aid[1] = m > 0 ? base + 1 : -1;
  nd[m].aid[2] = skip ? base + 2 : -1;
  a.nflags[m] = uint8_t((m == 0 ? NF_START : 0) | (m + 2 >= N ? NF_ACCEPT : 0));
  a.snode[rank] = m;
  a.slab[rank] = my;
  if (m == N - 1 && a.n_arcs) a.n_arcs[0] = base + 1 + (m > 0 ? 1 : 0) + skip;
}

// ==========================================================================================
// viterbiScore / viterbiPath of chain o G (never built), G banded: the tropical recursion
// shortest.cpp:86-170 / :190-272 would run over the product compose.cpp:377-522 builds, with
// one back-pointer (which of the <= 3 in-arcs) per (time, node), the pointer chase and the
// path's arcs in one launch.  Arithmetic as the reference's on the built lattice: a composed
// arc weighs w_G + em (compose.cpp:431-434), a candidate is score(src) + weight, maxima are
// exact -- so scores are bit-identical to the built path's.  Which of two EQUAL candidates wins
// depends on the built lattice's node numbering (in-list order for viterbiScore, queue order
// for viterbiPath); an exact tie between finite candidates only raises `tie`, and the host runs
// that utterance through the built lattice instead (ops.cpp: band_viterbi).
// One workgroup of 512 lanes per utterance (lane = node), emission rows staged through LDS.
// ==========================================================================================
__global__ __launch_bounds__(512) void band_viterbi_kernel(const BandDecode* __restrict__ pairs) {
  const BandDecode P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, N = P.N, NS = P.NS;
  extern __shared__ float lds[];
  const float NINF = -__builtin_inff();
  float* a0 = lds;               // [2 + 512] two pads in front: nodes -1 and -2
  float* a1 = lds + 516;
  float* stage = lds + 1032;     // emission rows / back-pointer rows
  const int m = threadIdx.x;
  const GTNX_G gtnx_i4* nodes = reinterpret_cast<const GTNX_G gtnx_i4*>(P.nodes);
  int lab = 0;
  float wi[3] = {NINF, NINF, NINF};
  bool accept = false, start = false;
  if (m < N) {
    const gtnx_i4 q = nodes[m];
    lab = q.x >= 0 ? q.x : 0;
    const int a[3] = {q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (a[k] >= 0) wi[k] = P.w ? P.w[a[k]] : 0.0f;
    const uint8_t f = P.nflags[m];
    start = (f & NF_START) != 0;
    accept = (f & NF_ACCEPT) != 0;
  }
  if (m < 2) a0[m] = a1[m] = NINF;
  a0[2 + m] = start ? 0.0f : NINF;  // shortest.cpp:201-207 (paths begin at start nodes, time 0)
  a1[2 + m] = NINF;
  int tie = 0;
  const int R = max(1, min(64, P.stage_floats / C));  // emission rows per chunk
  float* cur = a0;
  float* nxt = a1;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += R) {
    const int rows = min(R, T - t0);
    for (int e = threadIdx.x; e < rows * C; e += blockDim.x) stage[e] = P.em[int64_t(t0) * C + e];
    __syncthreads();
    for (int i = 0; i < rows; ++i) {
      const float e = stage[i * C + lab];
      const float c0 = cur[2 + m] + (wi[0] + e), c1 = cur[1 + m] + (wi[1] + e), c2 = cur[m] + (wi[2] + e);
      float best = c0;
      int k = 0;
      if (c1 > best) best = c1, k = 1;
      if (c2 > best) best = c2, k = 2;
      if (best > NINF && ((k != 0 && c0 == best) || (k != 1 && c1 == best) || (k != 2 && c2 == best))) tie = 1;
      if (m < NS) P.bp[int64_t(t0 + i) * NS + m] = uint8_t(best > NINF ? k : 3);
      nxt[2 + m] = m < N ? best : NINF;
      __syncthreads();
      float* sw = cur;
      cur = nxt;
      nxt = sw;
    }
  }
  // the best accept node (shortest.cpp:153-167 / :233-244): block maximum, smallest node among equals
  __shared__ float red_v[8];
  __shared__ int red_m[8], s_best, s_tie;
  float v = accept ? cur[2 + m] : NINF;
  int bm = (m < N && accept && v > NINF) ? m : 1 << 30;
  const float wmx = wave_max(v);
  if (!(v == wmx)) bm = 1 << 30;
  int cnt = (bm != (1 << 30)) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) {
    bm = min(bm, __shfl_xor(bm, o));
    cnt += __shfl_xor(cnt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red_v[threadIdx.x >> 6] = wmx;
    red_m[threadIdx.x >> 6] = bm;
  }
  if (cnt > 1) tie = 1;
  if (threadIdx.x == 0) s_tie = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    float bv = NINF;
    int bn = -1;
    for (int w = 0; w < 8; ++w) {
      if (red_m[w] == (1 << 30)) continue;
      if (red_v[w] > bv) bv = red_v[w], bn = red_m[w];
      else if (red_v[w] == bv && bv > NINF) s_tie = 1;
    }
    s_best = bn;
    P.score[0] = bv;
  }
  if (tie) atomicOr(&s_tie, 1);
  __syncthreads();
  const int best = s_best;
  if (threadIdx.x == 0) {
    P.tie[0] = s_tie;
    P.path_len[0] = best >= 0 ? T : -1;
  }
  if (best < 0 || T == 0) return;
  // ---- chase the pointers, back-pointer rows staged through LDS (coalesced), one lane walking
  uint8_t* bst = reinterpret_cast<uint8_t*>(stage);
  const int RB = max(1, min(T, P.stage_floats * 4 / NS));
  __shared__ int s_node;
  if (threadIdx.x == 0) {
    s_node = best;
    P.pnode[T] = best;
  }
  for (int hi = T; hi > 0; hi -= RB) {
    const int lo = max(0, hi - RB);
    __syncthreads();
    for (int e = threadIdx.x; e < (hi - lo) * NS; e += blockDim.x) bst[e] = P.bp[int64_t(lo) * NS + e];
    __syncthreads();
    if (threadIdx.x == 0) {
      int node = s_node;
      for (int t = hi - 1; t >= lo; --t) {
        node -= bst[(t - lo) * NS + node] & 3;
        P.pnode[t] = node;
      }
      s_node = node;
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- the path's arcs: arc t enters node pnode[t+1] from pnode[t]
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const int m1 = P.pnode[t + 1], k = m1 - P.pnode[t];
    const gtnx_i4 q = nodes[m1];
    const int arc = k == 0 ? q.y : (k == 1 ? q.z : q.w);
    P.path_arc[t] = arc;
    P.path_lab[t] = q.x;
    P.path_w[t] = (P.w ? P.w[arc] : 0.0f) + P.em[int64_t(t) * C + q.x];
  }
}

// ==========================================================================================
// The same recursion with ONE WAVE per utterance and no barrier at all (the kernel above spends a workgroup
// barrier and a byte store per time step).  A lane owns NPL consecutive nodes and keeps their scores in
// registers; the two neighbours a node looks at are a register of the same lane or one wave-wide DPP shift
// away.  Emission rows reach LDS by DMA (global_load_lds_dwordx4: a kilobyte per wave instruction, no staging
// registers) in blocks of 2048 floats, three blocks in the ring: while one is swept, the next has landed or
// is landing and the one after it is in flight -- the wait at a block's top is a COUNTED s_waitcnt (vmcnt(16):
// everything older than the sixteen newest vector-memory operations is done; a block is eight of them, always
// -- lanes past the end of the tensor re-read its last 16 bytes -- and the back-pointer stores in between only
// make the block older).  Back-pointers are 2 bits per (time, node), packed into one 32-bit word per lane for
// 16 / NPL steps and stored coalesced: T N / 4 bytes instead of T N.  Code 3 says "two finite candidates were
// equal here": the pointer chase reports a tie only when the BEST PATH runs through such a node (the host then
// lets the built lattice decide, ops_band.cpp) -- ties elsewhere do not change the path.  The chase is a scalar
// walk: the word row of a group of steps is loaded by all lanes (the next one already in flight) and the
// walker takes its lane's word with v_readlane -- no dependent memory access on the chain.  Same arithmetic,
// same outputs as band_viterbi_kernel.
// (Built and measured, not in the tree: TWO waves per utterance, the upper half of the nodes a block of rows behind
//  the lower half -- boundary scores through an LDS ring, a progress counter per wave, no barrier.  Correct, and
//  slower: 0.47 against 0.37 ms at C3, 1.29 against 0.82 at B = 2048 -- of the ~95 instructions a step costs a
//  wave, ~55 do not depend on how many nodes a lane owns: addresses, the two DPP shifts, packing, the loop.)
// Algorithmic bytes per utterance: 4 T C (emissions, once) + T N / 2 (back-pointers out and in) + 20 T (the path).
// ==========================================================================================
constexpr int VBLK = 2048;  // floats per staged block (C <= VBLK, C % 4 == 0, 16-byte aligned tensor)

// RANKED (a second launch over the utterances whose first pass met an exact tie on the best path, when their target
// is CTC-shaped: ops_band.cpp tie_ranks): equal candidates are decided by the rank of their source node -- the
// position the reference's queue (viterbiPath) or its compose (viterbiScore's in-lists, the accept list) gives that
// node in every layer of the lattice -- so the answer is the reference's without building the lattice.
template <int NPL, bool RANKED>
__global__ __launch_bounds__(64) void band_viterbi_wave_kernel(const BandDecode* __restrict__ pairs) {
  const BandDecode P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, N = P.N;
  const int lane = threadIdx.x;
  constexpr int SPW = 16 / NPL;  // steps per back-pointer word
  constexpr int NLD = VBLK / 256;
  __shared__ __attribute__((aligned(16))) float ring[3 * VBLK];
  const float NINF = -__builtin_inff();
  const GTNX_G gtnx_i4* nodes = reinterpret_cast<const GTNX_G gtnx_i4*>(P.nodes);
  int loff[NPL];  // byte offset of the node's label inside an emission row
  float w0[NPL], w1[NPL], w2[NPL], alpha[NPL];
  bool acc[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int m = lane * NPL + j;
    loff[j] = 0;
    w0[j] = w1[j] = w2[j] = NINF;  // (nodes past N keep -inf: every candidate of theirs is -inf)
    alpha[j] = NINF;
    acc[j] = false;
    if (m < N) {
      const gtnx_i4 q = nodes[m];
      loff[j] = 4 * (q.x >= 0 ? q.x : 0);
      if (q.y >= 0) w0[j] = P.w ? P.w[q.y] : 0.0f;
      if (q.z >= 0) w1[j] = P.w ? P.w[q.z] : 0.0f;
      if (q.w >= 0) w2[j] = P.w ? P.w[q.w] : 0.0f;
      const uint8_t f = P.nflags[m];
      if (f & NF_START) alpha[j] = 0.0f;  // shortest.cpp:201-207 (paths begin at start nodes, time 0)
      acc[j] = (f & NF_ACCEPT) != 0;
    }
  }
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    settle(w0[j]);
    settle(w1[j]);
    settle(w2[j]);
    settle(loff[j]);
  }
  int r0[NPL], r1[NPL], r2[NPL];  // RANKED: ranks of the three source nodes of every node of this lane
  if (RANKED) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int m = lane * NPL + j;
      r0[j] = m < N ? P.rank_in[m] : 0x7ffffffe;
      r1[j] = (m >= 1 && m - 1 < N) ? P.rank_in[m - 1] : 0x7ffffffe;
      r2[j] = (m >= 2 && m - 2 < N) ? P.rank_in[m - 2] : 0x7ffffffe;
    }
  }
  const int R = max(1, VBLK / C);  // rows per block
  const int NB = (T + R - 1) / R;
  const int64_t total = int64_t(T) * C;
  auto issue = [&](int b) {
    const int64_t f0 = int64_t(b) * R * C;
    float* dst = ring + (b % 3) * VBLK;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int64_t f = f0 + i * 256 + lane * 4;
      f = f + 4 <= total ? f : total - 4;  // (past the end: the tensor's last 16 bytes again, never read from LDS)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.em + f),
                                       (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
    }
  };
  GTNX_G unsigned* bp32 = reinterpret_cast<GTNX_G unsigned*>(P.bp);
  unsigned word = 0;
  if (NB > 0) issue(0);
  if (NB > 1) issue(1);
#ifdef GTNX_VIT_NO_SWEEP  // (tools/ubench/viterbi_bench.hip: everything but the recursion -- the emissions still stream)
#define GTNX_VIT_ROWS 0
#else
#define GTNX_VIT_ROWS rows
#endif
  for (int b = 0; b < NB; ++b) {
    if (b + 2 < NB) {
      issue(b + 2);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (b + 1 < NB) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* buf = reinterpret_cast<const char*>(ring + (b % 3) * VBLK);
    const int t0 = b * R, rows = min(R, T - t0);
    float e[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) e[j] = *reinterpret_cast<const float*>(buf + loff[j]);
    for (int i = 0; i < GTNX_VIT_ROWS; ++i) {
      const int t = t0 + i;
      // the next row's emissions while this row is reduced (the last row of the block reads its own again)
      const char* nrow = buf + (i + 1 < rows ? i + 1 : i) * (C * 4);
      float en[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) en[j] = *reinterpret_cast<const float*>(nrow + loff[j]);
      float s1, s2;
      if (NPL >= 2) {
        s1 = wave_shr1(alpha[NPL - 1], NINF);
        s2 = wave_shr1(alpha[NPL >= 2 ? NPL - 2 : 0], NINF);
      } else {
        s1 = wave_shr1(alpha[0], NINF);
        s2 = wave_shr1(s1, NINF);
      }
      float na[NPL];
      const int sh = (t % SPW) * 2 * NPL;
      unsigned codes = 0;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const float p1 = j > 0 ? alpha[j > 0 ? j - 1 : 0] : s1;
        const float p2 = j > 1 ? alpha[j > 1 ? j - 2 : 0] : (j == 1 ? s1 : s2);
        const float c0 = alpha[j] + (w0[j] + e[j]), c1 = p1 + (w1[j] + e[j]), c2 = p2 + (w2[j] + e[j]);
        const float best = fmaxf(fmaxf(c0, c1), c2);
        const float med = __builtin_amdgcn_fmed3f(c0, c1, c2);
        unsigned k;
        if (RANKED) {  // of the candidates that reach the maximum, the one from the source with the smallest rank
          const int q0 = c0 == best ? r0[j] : 0x7fffffff, q1 = c1 == best ? r1[j] : 0x7fffffff,
                    q2 = c2 == best ? r2[j] : 0x7fffffff;
          k = (q0 <= q1 && q0 <= q2) ? 0u : (q1 <= q2 ? 1u : 2u);
          k = best == NINF ? 3u : k;
          (void)med;
        } else {
          // first maximum in the order own node, node - 1, node - 2 (as the kernel above); two equal maxima: code 3
          k = c0 == best ? 0u : (c1 == best ? 1u : 2u);
          k = med == best ? 3u : k;  // (all -inf: 3 as well -- the best path never comes through a dead node)
        }
        codes |= k << (2 * j);
        na[j] = best;
      }
      word |= codes << sh;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        alpha[j] = na[j];
        e[j] = en[j];
      }
      if ((t % SPW) == SPW - 1 || t == T - 1) {
        bp32[int64_t(t / SPW) * 64 + lane] = word;
        word = 0;
      }
    }
  }
  // the best accept node (shortest.cpp:153-167 / :233-244): maximum, smallest node among equals
  float lv = NINF;
#pragma unroll
  for (int j = 0; j < NPL; ++j) lv = fmaxf(lv, acc[j] ? alpha[j] : NINF);
  const float wmx = wave_max(lv);
  int bm = 1 << 30, cnt = 0;
#pragma unroll
  for (int j = NPL - 1; j >= 0; --j)
    if (acc[j] && alpha[j] == wmx && wmx > NINF) {
      const int m = lane * NPL + j;
      // RANKED: (rank in the accept list, node) -- N <= 512 and the ranks are below N
      bm = RANKED ? min(bm, (P.rank_acc[m] << 16) | m) : m;
      ++cnt;
    }
  for (int o = 32; o > 0; o >>= 1) {
    bm = min(bm, __shfl_xor(bm, o));
    cnt += __shfl_xor(cnt, o);
  }
  int tie = (cnt > 1 && !RANKED) ? 1 : 0;  // (uniform)
  const int best = bm == (1 << 30) ? -1 : (RANKED ? (bm & 0xffff) : bm);
  if (lane == 0) {
    P.score[0] = best >= 0 ? wmx : NINF;
    P.path_len[0] = best >= 0 ? T : -1;
  }
  if (best < 0 || T == 0) {
    if (lane == 0) P.tie[0] = tie;
    return;
  }
#ifdef GTNX_VIT_NO_CHASE  // (tools/ubench/viterbi_bench.hip: the sweep alone)
  if (lane == 0) P.tie[0] = tie;
  return;
#endif
  // (the back-pointer words this wave stored are what it loads next: its own stores, waited for)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- chase the pointers: a scalar walk over the word rows -- sixteen rows (16 SPW steps) per batch, loaded by
  // all lanes before the walk, the next batch in flight meanwhile; the nodes of the path are kept in LDS for the
  // last phase (the emission ring is free by now) and stored 64 at a time
  unsigned node = unsigned(__builtin_amdgcn_readfirstlane(best));
  int* lpn = reinterpret_cast<int*>(ring);  // [min(T + 1, 3 VBLK)]
  constexpr int LPN = 3 * VBLK;
  if (lane == 0) {
    P.pnode[T] = int(node);
    if (T < LPN) lpn[T] = int(node);
  }
  const int NW = (T + SPW - 1) / SPW;
  constexpr int RB = 16;  // word rows per batch
  unsigned cur[RB], nxt[RB];
  const int nbatch = (NW + RB - 1) / RB;
  auto fetch = [&](unsigned (&dst)[RB], int q) {  // rows q RB .. q RB + RB - 1 (those past the end: the last row again)
#pragma unroll
    for (int r = 0; r < RB; ++r) dst[r] = bp32[int64_t(min(q * RB + r, NW - 1)) * 64 + lane];
  };
  fetch(cur, nbatch - 1);
  int pn = 0;
  for (int q = nbatch - 1; q >= 0; --q) {
    if (q > 0) fetch(nxt, q - 1);
#pragma unroll
    for (int r = RB - 1; r >= 0; --r) {
      const int wi = q * RB + r;
      if (wi < NW) {
#pragma unroll
        for (int u = SPW - 1; u >= 0; --u) {
          const int t = wi * SPW + u;
          if (t < T) {
            const unsigned wv = unsigned(__builtin_amdgcn_readlane(int(cur[r]), int(node / NPL)));
            unsigned k = (wv >> (u * 2 * NPL + 2 * (node % NPL))) & 3u;
            if (k == 3u) {  // two equal candidates ON the path: the built lattice decides; the walk goes on (own node)
              tie = 1;
              k = 0;
            }
            node -= min(k, node);
            if (lane == (t & 63)) pn = int(node);
            if ((t & 63) == 0) {  // the nodes of steps t .. t + 63, one per lane
              if (t + lane < T) {
                P.pnode[t + lane] = pn;
                if (t + lane < LPN) lpn[t + lane] = pn;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) cur[r] = nxt[r];
  }
  if (lane == 0) P.tie[0] = tie;
  if (T >= LPN) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- the path's arcs: arc t enters node pnode[t+1] from pnode[t]
  for (int t = lane; t < T; t += 64) {
    const int m1 = T < LPN ? lpn[t + 1] : P.pnode[t + 1], k = m1 - (T < LPN ? lpn[t] : P.pnode[t]);
    const gtnx_i4 q = nodes[m1];
    const int arc = k == 0 ? q.y : (k == 1 ? q.z : q.w);
    P.path_arc[t] = arc;
    P.path_lab[t] = q.x;
    P.path_w[t] = (arc >= 0 && P.w ? P.w[arc] : 0.0f) + P.em[int64_t(t) * C + (q.x >= 0 ? q.x : 0)];
  }
}

// Force-alignment acceptors composed with an ASG transitions graph, as band records: what
// compose(forceAlign(target), transitions) of examples/asg.cpp:50-68 builds -- a chain of U + 1
// nodes, node m carrying label l_m, arc 2m-2 the step m-1 -> m and arc 2m-1 the self-loop at m (the
// reference's compose numbers them in this order) -- with the arcs' weights GATHERED from the
// transitions' weights (asgTransitions layout: arc i = <s> -> label i, arc N + i N + j = label j ->
// label i) and the map arc -> transitions arc kept for the gradient's way back.
__global__ __launch_bounds__(512) void asg_fal_targets_kernel(const AsgFalArgs* __restrict__ args, int NL) {
  const AsgFalArgs a = args[blockIdx.x];
  __shared__ int lab[512];
  const int U = a.U, m = threadIdx.x;  // nodes 0 .. U
  int my = -1;
  if (m >= 1 && m <= U) my = a.labels[m - 1];
  lab[m] = my;
  __syncthreads();
  if (m > U) return;
  GTNX_G BandNode* nd = const_cast<GTNX_G BandNode*>(a.nodes);
  nd[m].lab = my;
  nd[m].aid[0] = m >= 1 ? 2 * m - 1 : -1;
  nd[m].aid[1] = m >= 1 ? 2 * m - 2 : -1;
  nd[m].aid[2] = -1;
  a.nflags[m] = uint8_t((m == 0 ? NF_START : 0) | (m == U ? NF_ACCEPT : 0));
  if (m >= 1) {
    int rank = 0;
    for (int o = 1; o <= U; ++o) {
      const int lo = lab[o];
      rank += (lo < my || (lo == my && o < m)) ? 1 : 0;
    }
    a.snode[rank] = m;
    a.slab[rank] = my;
    const int step = m == 1 ? my : NL + my * NL + lab[m - 1];
    const int self = NL + my * NL + my;
    a.arc_map[2 * m - 2] = step;
    a.arc_map[2 * m - 1] = self;
    a.w[2 * m - 2] = a.trans_w[step];
    a.w[2 * m - 1] = a.trans_w[self];
  }
}
// d loss / d transitions: every force-alignment arc hands its gradient to the transitions arc it came from
// (blockIdx.y: the sequence; tab: {gradient offset, arc-map offset, arcs} per sequence)
__global__ void asg_fal_scatter_kernel(const float* __restrict__ g, const int* __restrict__ maps,
                                       const int64_t* __restrict__ tab, float* __restrict__ trans_grad) {
  const int64_t* e = tab + int64_t(blockIdx.y) * 3;
  const int64_t n = e[2];
  const float* gb = g + e[0];
  const int* mb = maps + e[1];
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    if (mb[i] >= 0 && gb[i] != 0.0f) atomicAdd(trans_grad + mb[i], gb[i]);
}

template <class K>
void big_lds(K kern) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
}

// `one` (host memory, n == 1): the pair goes as the kernel's argument instead of through the table (instantiated for
// NPL = 1, 16-byte-aligned emissions, not BIG: band_one_ok)
template <int NPL, int K, bool VEC>
void launch_fwd2(const BandPair* d, const BandPair* one, int n, int ns, size_t lds, bool unit, hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  if (gtnx_first_on_device first{done}) {
    big_lds(band_forward_kernel<NPL, true, K, VEC>);
    big_lds(band_forward_kernel<NPL, false, K, VEC>);
    if constexpr (NPL == 1 && VEC) {
      big_lds(band_forward_one_kernel<NPL, true, K, VEC>);
      big_lds(band_forward_one_kernel<NPL, false, K, VEC>);
    }
  }
  if constexpr (NPL == 1 && VEC) {
    if (one) {
      if (unit) hipLaunchKernelGGL((band_forward_one_kernel<NPL, true, K, VEC>), dim3(1), dim3(WG), lds, st, *one, ns);
      else hipLaunchKernelGGL((band_forward_one_kernel<NPL, false, K, VEC>), dim3(1), dim3(WG), lds, st, *one, ns);
      return;
    }
  }
  if (unit) hipLaunchKernelGGL((band_forward_kernel<NPL, true, K, VEC>), dim3(n), dim3(WG), lds, st, d, ns);
  else hipLaunchKernelGGL((band_forward_kernel<NPL, false, K, VEC>), dim3(n), dim3(WG), lds, st, d, ns);
}
template <int NPL, int K>
void launch_fwd(const BandPair* d, const BandPair* one, int n, int ns, size_t lds, bool unit, bool vec, hipStream_t st) {
  if (vec) launch_fwd2<NPL, K, true>(d, one, n, ns, lds, unit, st);
  else launch_fwd2<NPL, K, false>(d, one, n, ns, lds, unit, st);
}
template <int NPL, int K, bool VEC, bool BIG>
void launch_bwd3(const BandPair* d, const BandPair* one, int n, int ns, size_t lds, bool unit, bool gradg, hipStream_t st,
                 const BandPatch& pt) {
  constexpr bool ONE = NPL == 1 && VEC && !BIG;
  static std::atomic<uint64_t> done{0};
  if (gtnx_first_on_device first{done}) {
    big_lds(band_backward_kernel<NPL, true, true, K, VEC, BIG>);
    big_lds(band_backward_kernel<NPL, true, false, K, VEC, BIG>);
    big_lds(band_backward_kernel<NPL, false, true, K, VEC, BIG>);
    big_lds(band_backward_kernel<NPL, false, false, K, VEC, BIG>);
    if constexpr (ONE) {
      big_lds(band_backward_one_kernel<NPL, true, true, K, VEC, BIG>);
      big_lds(band_backward_one_kernel<NPL, true, false, K, VEC, BIG>);
      big_lds(band_backward_one_kernel<NPL, false, true, K, VEC, BIG>);
      big_lds(band_backward_one_kernel<NPL, false, false, K, VEC, BIG>);
    }
  }
  if constexpr (ONE) {
    if (one) {
      if (unit) {
        if (gradg) hipLaunchKernelGGL((band_backward_one_kernel<NPL, true, true, K, VEC, BIG>), dim3(1), dim3(WGB), lds, st, *one, ns);
        else hipLaunchKernelGGL((band_backward_one_kernel<NPL, true, false, K, VEC, BIG>), dim3(1), dim3(WGB), lds, st, *one, ns);
      } else {
        if (gradg) hipLaunchKernelGGL((band_backward_one_kernel<NPL, false, true, K, VEC, BIG>), dim3(1), dim3(WGB), lds, st, *one, ns);
        else hipLaunchKernelGGL((band_backward_one_kernel<NPL, false, false, K, VEC, BIG>), dim3(1), dim3(WGB), lds, st, *one, ns);
      }
      return;
    }
  }
  if (unit) {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, true, true, K, VEC, BIG>), dim3(n), dim3(WGB), lds, st, d, ns, pt);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, true, false, K, VEC, BIG>), dim3(n), dim3(WGB), lds, st, d, ns, pt);
  } else {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, false, true, K, VEC, BIG>), dim3(n), dim3(WGB), lds, st, d, ns, pt);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, false, false, K, VEC, BIG>), dim3(n), dim3(WGB), lds, st, d, ns, pt);
  }
}
template <int NPL, int K>
void launch_bwd(const BandPair* d, const BandPair* one, int n, int ns, size_t lds, bool unit, bool gradg, bool vec, bool big,
                hipStream_t st, const BandPatch& pt) {
  if constexpr (K == 2) {
    if (big) {
      if (vec) launch_bwd3<NPL, K, true, true>(d, one, n, ns, lds, unit, gradg, st, pt);
      else launch_bwd3<NPL, K, false, true>(d, one, n, ns, lds, unit, gradg, st, pt);
      return;
    }
  }
  if (vec) launch_bwd3<NPL, K, true, false>(d, one, n, ns, lds, unit, gradg, st, pt);
  else launch_bwd3<NPL, K, false, false>(d, one, n, ns, lds, unit, gradg, st, pt);
}

constexpr size_t LDS_TWO = 78 * 1024;   // two workgroups per CU
constexpr size_t LDS_ONE = 156 * 1024;  // one

} // namespace

int band_block_rows(int C, int max_NS, bool backward);
// The backward sweep's LDS row stride of alpha / posterior rows: with four rows per block one wave sums a block's four
// posterior rows sixteen lanes per row, sixteen nodes per lane (band_backward_body.inc: ROWSUM4) -- a lane's nodes are
// inside the row or past it
static int band_backward_lds_stride(int K, int max_NS) { return K == 4 ? (max_NS + 15) / 16 * 16 : max_NS; }
int band_max_nodes() { return 512; }
int band_max_labels() { return 1024; }
int band_min_labels() { return 4; }  // a chunk of emission rows is fetched with 16-byte loads (Stage::issue)
int band_npl(int max_nodes) { return max_nodes <= 256 ? 1 : 2; }
int band_row_stride(int N, int) { return (N + 3) / 4 * 4; }
#ifndef GTNX_FWD_SHIFT_ROWS
#define GTNX_FWD_SHIFT_ROWS 4
#endif
// (log2 of RNk in band_forward_body.inc: the rows between two shifts of a wave's running row)
int band_forward_lgrn(int C) {
  const int K = band_block_rows(C, 0, false);
  return K >= GTNX_FWD_SHIFT_ROWS ? (GTNX_FWD_SHIFT_ROWS == 8 ? 3 : 2) : (K >= 4 ? 2 : 1);
}
// rows per tick: the largest block that keeps two workgroups on a CU, else the smallest (one per CU)
int band_block_rows(int C, int max_NS, bool backward) {
  for (int k = backward ? 4 : 8; k >= 2; k /= 2) {
    if (k * C > 2048 || k * max_NS > 2048) continue;
    if (backward && k > 2 && (k * C > 1024 || k * max_NS > 1024)) continue;  // 16 staging registers per stream
    const int ns_lds = backward ? band_backward_lds_stride(k, max_NS) : max_NS;
    if (4 * size_t(band_lds(C, k, ns_lds, backward).total) + 64 <= LDS_TWO) return k;
  }
  if (4 * size_t(band_lds(C, 2, max_NS, backward).total) + 64 <= LDS_ONE) return 2;
  return 0;
}

// a launch of ONE pair whose record may travel as the kernel's argument (host pointer `one` of launch_band_*)
bool band_one_ok(int npl, int C, int max_NS, bool vec, bool backward) {
  if (npl != 1 || !vec) return false;
  if (!backward) return true;
  const int K = band_block_rows(C, max_NS, true);
  return !(K * C > 1024 || K * max_NS > 1024);
}
void launch_band_forward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, bool vec, hipStream_t st,
                         const BandPair* one) {
  if (n <= 0) return;
  if (one && (n != 1 || !band_one_ok(npl, C, max_NS, vec, false))) throw std::logic_error("band.hip: not a single-pair launch");
  const int K = band_block_rows(C, 0, false);  // the forward sweep stages no alpha rows
  const size_t lds = 4 * size_t(band_lds(C, K, max_NS, false).total) + 64;
  if (npl == 1) {
    if (K == 8) launch_fwd<1, 8>(d_pairs, one, n, max_NS, lds, unit, vec, st);
    else if (K == 4) launch_fwd<1, 4>(d_pairs, one, n, max_NS, lds, unit, vec, st);
    else launch_fwd<1, 2>(d_pairs, one, n, max_NS, lds, unit, vec, st);
  } else {
    if (K == 8) launch_fwd<2, 8>(d_pairs, one, n, max_NS, lds, unit, vec, st);
    else if (K == 4) launch_fwd<2, 4>(d_pairs, one, n, max_NS, lds, unit, vec, st);
    else launch_fwd<2, 2>(d_pairs, one, n, max_NS, lds, unit, vec, st);
  }
}

// every pair of the launch shares C; max_NS: largest alpha row stride of the launch
void launch_band_backward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, bool gradg, bool vec,
                          hipStream_t st, const BandPair* one, const BandPatch* patch) {
  if (n <= 0) return;
  if (one && patch) throw std::logic_error("band.hip: a patched launch reads a device table");
  BandPatch pt{};
  if (patch) pt = *patch;
  if (one && (n != 1 || !band_one_ok(npl, C, max_NS, vec, true))) throw std::logic_error("band.hip: not a single-pair launch");
  const int K = band_block_rows(C, max_NS, true);
  const bool big = K * C > 1024 || K * max_NS > 1024;
  const int ns_lds = band_backward_lds_stride(K, max_NS);  // (the kernels' NSmax: the stride of rows in LDS, not in HBM)
  const size_t lds = 4 * size_t(band_lds(C, K, ns_lds, true).total) + 64;
  if (npl == 1) {
    if (K == 4) launch_bwd<1, 4>(d_pairs, one, n, ns_lds, lds, unit, gradg, vec, big, st, pt);
    else launch_bwd<1, 2>(d_pairs, one, n, ns_lds, lds, unit, gradg, vec, big, st, pt);
  } else {
    // (four rows per block need K max_NS <= 1024, i.e. at most 256 nodes: one node per lane)
    if (K == 4) throw std::logic_error("band.hip: four rows per block with two nodes per lane");
    launch_bwd<2, 2>(d_pairs, one, n, ns_lds, lds, unit, gradg, vec, big, st, pt);
  }
}

// max_nodes / max_labels: the widest partner and alphabet of the batch; vec: every emission tensor is 16-byte aligned
// and every alphabet a multiple of 4 (and T C >= 4).  The one-wave kernel takes such batches with <= 512 nodes and
// <= 2048 labels
// (GTNX_VITERBI_WG=1: the workgroup-per-utterance kernel, kept for wider shapes and for comparison).
template <int NPL>
static void launch_viterbi_wave(const BandDecode* d_pairs, int n, int ranked, hipStream_t st) {
  if (ranked) hipLaunchKernelGGL((band_viterbi_wave_kernel<NPL, true>), dim3(n), dim3(64), 0, st, d_pairs);
  else hipLaunchKernelGGL((band_viterbi_wave_kernel<NPL, false>), dim3(n), dim3(64), 0, st, d_pairs);
}
bool band_viterbi_wave_ok(int max_nodes, int max_labels, int vec) {
  static const bool force_wg = std::getenv("GTNX_VITERBI_WG") != nullptr;
  return !force_wg && vec && max_nodes <= 512 && max_labels <= VBLK;
}
void launch_band_viterbi(const BandDecode* d_pairs, int n, int stage_floats, int max_nodes, int max_labels, int vec,
                         hipStream_t st, int ranked) {
  if (n <= 0) return;
  if (band_viterbi_wave_ok(max_nodes, max_labels, vec)) {
    if (max_nodes <= 64) launch_viterbi_wave<1>(d_pairs, n, ranked, st);
    else if (max_nodes <= 128) launch_viterbi_wave<2>(d_pairs, n, ranked, st);
    else if (max_nodes <= 256) launch_viterbi_wave<4>(d_pairs, n, ranked, st);
    else launch_viterbi_wave<8>(d_pairs, n, ranked, st);
    return;
  }
  if (ranked) return;  // (callers ask band_viterbi_wave_ok first)
  static std::atomic<uint64_t> done{0};
  if (gtnx_first_on_device first{done}) big_lds(band_viterbi_kernel);
  hipLaunchKernelGGL(band_viterbi_kernel, dim3(n), dim3(512), 4 * size_t(1032 + stage_floats) + 64, st, d_pairs);
}

void launch_asg_fal_targets(const AsgFalArgs* d_args, int n, int n_labels, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(asg_fal_targets_kernel, dim3(n), dim3(512), 0, st, d_args, n_labels);
}
void launch_asg_fal_scatter(const float* g, const int* maps, const int64_t* tab, int n_seq, int64_t longest, float* trans_grad,
                            hipStream_t st) {
  if (n_seq > 0 && longest > 0)
    hipLaunchKernelGGL(asg_fal_scatter_kernel, dim3(unsigned(std::min<int64_t>((longest + 255) / 256, 64)), unsigned(n_seq)),
                       dim3(256), 0, st, g, maps, tab, trans_grad);
}
void launch_ctc_targets(const CtcTargetArgs* d_args, int n, int blank, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(ctc_targets_kernel, dim3(n), dim3(512), 0, st, d_args, blank);
}

} // namespace gtnx
