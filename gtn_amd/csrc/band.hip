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
// One workgroup of 4 waves per utterance; wave w owns nodes [64w NPL, 64(w+1) NPL), NPL = 1
// or 2 nodes per lane.  The recursion over time is a SKEWED PIPELINE: a node needs values of
// lower-numbered nodes only (higher-numbered in the backward sweep), so wave w runs one time
// step behind wave w-1 and finds the two boundary values it needs in LDS, written a whole
// step earlier.  Inside a wave neighbours move by DPP wave shifts.  One LDS-only barrier per
// tick keeps the waves in step; nothing on the critical path waits for LDS or HBM latency:
//   * emission rows live in an LDS ring, landed a chunk ahead from 16-byte loads issued two
//     chunks ahead (every wave stages its share; there are no helper waves);
//   * scores are in log2 units (bare v_exp_f32 / v_log_f32), "minus infinity" is -1e30 so no
//     step needs an inf / NaN guard, and every RN rows the row is shifted by the maximum of
//     the row RN rows back (the shifts are summed in fp64): magnitudes stay O(10) and float32
//     keeps ~1e-5 relative accuracy on every posterior for any T;
//   * the backward sweep adds node posteriors into an LDS ring of gradient rows (pre-filled
//     with the normaliser's softmax term) which is drained with coalesced stores once the
//     last wave is through with a chunk; G's arc gradients are register accumulators.
// HBM traffic per utterance: forward 4TC + 4(T+1)NS, backward 8TC + 4(T+1)NS (+ G).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr float NEGF = -1.0e30f;   // log-domain zero
constexpr float DEADF = -1.0e29f;  // anything below is "no path"
constexpr float LOG2E = 1.44269504088896340736f;
constexpr double LN2 = 0.693147180559945309417;
constexpr int BW = 256;  // lanes per workgroup (4 waves)
constexpr int RN = 4;    // rows between shifts of the running row; also the lag of the shift
constexpr int LAGW = 3;  // ticks between the leading and the last wave

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }

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
// LDS float add without return value (the compiler's atomic optimiser would wrap a plain
// atomicAdd of a uniform address in a lane loop)
__device__ __forceinline__ void lds_add(float* p, float v) {
  asm volatile("ds_add_f32 %0, %1" ::"v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))), "v"(v) : "memory");
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

// A chunk of `rows` contiguous HBM rows of W floats each, staged through NS floats per lane:
// issue() starts the loads, land() writes them to an LDS ring whose row `slot_of(r)` holds chunk row r.
template <int NS>
struct Stage {
  float v[NS];
  __device__ __forceinline__ void issue(const GTNX_G float* src, int cnt, bool vec, int tid) {
    if (vec) {
#pragma unroll
      for (int i = 0; i < NS / 4; ++i) {
        const int e = 4 * (i * BW + tid);
        gtnx_f4 q = {0.0f, 0.0f, 0.0f, 0.0f};
        if (e < cnt) q = *reinterpret_cast<const GTNX_G gtnx_f4*>(src + e);
        v[4 * i] = q.x;
        v[4 * i + 1] = q.y;
        v[4 * i + 2] = q.z;
        v[4 * i + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = i * BW + tid;
        v[i] = e < cnt ? src[e] : 0.0f;
      }
    }
  }
  // f(i, e, r, c, q): slot index, element, chunk row, column, values (4 in vec mode, else q.x)
  template <class F>
  __device__ __forceinline__ void each(int cnt, int W, bool vec, int tid, F&& f) const {
    const float invW = 1.0f / float(W);
    if (vec) {
#pragma unroll
      for (int i = 0; i < NS / 4; ++i) {
        const int e = 4 * (i * BW + tid);
        if (e < cnt) {
          const int r = row_of(e, W, invW);
          f(i, e, r, e - r * W, gtnx_f4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]});
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = i * BW + tid;
        if (e < cnt) {
          const int r = row_of(e, W, invW);
          f(i, e, r, e - r * W, gtnx_f4{v[i], 0.0f, 0.0f, 0.0f});
        }
      }
    }
  }
};

// shift applied to row `row` (a multiple of RN, >= RN): the maximum of row - RN over all waves
__device__ __forceinline__ float shift_of(const float* mxr, int row) {
  const float* p = mxr + (((row - RN) / RN) & 7) * 4;
  const float s = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
  return s > DEADF ? s : 0.0f;
}

// LDS layout shared by host (size) and device (carving); all counts in floats
struct BandLds {
  int CS;               // ring row stride of emission / gradient rows
  int NER, NAR, NGR;    // ring depths: emission rows, alpha rows, gradient rows
  int o_ering, o_aring, o_obuf, o_gring, o_scratch, o_misc, total;
};
__host__ __device__ inline BandLds band_lds(int C, int R, int NSmax, bool backward) {
  BandLds L;
  L.CS = C + 4;
  L.NER = 2 * R + LAGW + 1;
  L.NAR = backward ? L.NER : 0;
  L.NGR = backward ? (R >= 4 ? 3 * R : 4 * R) : 0;
  int o = 0;
  L.o_ering = o;
  o += L.NER * L.CS;
  o = (o + 3) & ~3;
  L.o_aring = o;
  o += L.NAR * NSmax;
  o = (o + 3) & ~3;
  L.o_obuf = o;  // doubles
  o += 2 * L.NAR;
  o = (o + 3) & ~3;
  L.o_gring = o;
  o += L.NGR * L.CS;
  L.o_scratch = o;
  o += backward ? BW : 0;
  o = (o + 3) & ~3;
  L.o_misc = o;  // red[16] doubles, bnd[4][4][2], mxr[8][4], fin[8], lse pairs [2][16][8][2]
  o += 32 + 32 + 32 + 8 + 512;
  L.total = o;
  return L;
}

// ==========================================================================================
// forward: alpha[t+1][m] = em[t][lab m] + log sum_k exp(alpha[t][m-k] + w_k(m))
// ==========================================================================================
template <int NPL, bool UNIT>
__global__ __launch_bounds__(BW) void band_forward_kernel(const BandPair* __restrict__ pairs, int R, int NSmax) {
  const BandPair P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, NS = P.NS;
  extern __shared__ float lds[];
  const BandLds L = band_lds(C, R, NSmax, false);
  float* ering = lds + L.o_ering;
  double* red = reinterpret_cast<double*>(lds + L.o_misc);  // [16]
  float* bnd = lds + L.o_misc + 32;    // [4 waves][4 slots][2]
  float* mxr = bnd + 32;               // [8 slots][4 waves]
  float* fin = mxr + 32;               // [4 waves][2]
  float* lsep = fin + 8;               // [2 chunk parities][16 rows][8 groups][2]
  const int CS = L.CS, NER = L.NER;
  const int Rm = R - 1, lgR = 31 - __builtin_clz(R);
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int m0 = tid * NPL;
  const int nchunks = (T + R - 1) / R;
  const bool vec = __builtin_amdgcn_readfirstlane(int(C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.em) & 15) == 0));
  const bool want_lse = P.norm != nullptr || P.rowlse != nullptr;

  NodeRegs<NPL> g;
  load_nodes<NPL, false>(P, m0, g);
  const bool writer = m0 < NS;
  float a[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) a[j] = g.start[j] ? 0.0f : NEGF;
  double off = 0.0;
  GTNX_G float* arow = P.alpha + m0;
  if (writer) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) arow[j] = a[j];
  }
  if (tid == 0) P.aoff[1] = 0.0;

  // ---- staging of the emission ring
  Stage<16> st;
  auto rows_of = [&](int c) { return c < nchunks ? min(R, T - c * R) : 0; };
  auto issue = [&](int c) { st.issue(P.em + int64_t(c) * R * C, rows_of(c) * C, vec, tid); };
  auto land = [&](int c) {
    const int base = (c * R) % NER;
    st.each(rows_of(c) * C, C, vec, tid, [&](int, int, int r, int col, gtnx_f4 q) {
      int s = base + r;
      if (s >= NER) s -= NER;
      float* d = ering + s * CS + col;
      if (vec) *reinterpret_cast<gtnx_f4*>(d) = gtnx_f4{em2(q.x), em2(q.y), em2(q.z), em2(q.w)};
      else d[0] = em2(q.x);
    });
  };
  // ---- row-wise log2-sum-exp2 of a landed chunk (normaliser): 256 / R lanes per row, pairs
  // (max, sum) per 16-lane group in phase A, merged per row by one lane in phase B
  const int LPR = BW / R, G16 = LPR / 16;
  const int EPL = (C + LPR - 1) / LPR;  // <= 16
  double normacc = 0.0;
  auto lse_a = [&](int c) {
    const int rows = rows_of(c), rr = tid / LPR, sub = tid - rr * LPR;
    if (rows <= 0) return;
    int s = (c * R) % NER + min(rr, rows - 1);
    if (s >= NER) s -= NER;
    const float* e = ering + s * CS + sub * EPL;
    float x[16];
    float m = NEGF;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] = (i < EPL && sub * EPL + i < C) ? e[i] : NEGF;
      m = fmaxf(m, x[i]);
    }
    m = row16_max(m);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (i < EPL) ? ex2(x[i] - m) : 0.0f;
    sum = row16_sum(sum);
    if ((tid & 15) == 0 && rr < rows) {
      float* p = lsep + (c & 1) * 256 + (rr * 8 + (sub >> 4)) * 2;
      p[0] = m;
      p[1] = sum;
    }
  };
  auto lse_b = [&](int c) {
    const int rows = rows_of(c);
    if (tid < rows) {
      const float* p = lsep + (c & 1) * 256 + tid * 16;
      float M = p[0];
      for (int k = 1; k < G16; ++k) M = fmaxf(M, p[2 * k]);
      float S = 0.0f;
      for (int k = 0; k < G16; ++k) S += p[2 * k + 1] * ex2(p[2 * k] - M);
      const float l2 = M + lg2(S);
      if (P.rowlse) P.rowlse[c * R + tid] = l2;
      normacc += double(l2);
    }
  };

  // ---- prologue
  issue(0);
  land(0);
  issue(1);
  if (NPL == 1) {
    if (l >= 62) bnd[(w * 4 + 0) * 2 + (63 - l)] = a[0];
  } else if (l == 63) {
    bnd[(w * 4 + 0) * 2 + 0] = a[NPL - 1];
    bnd[(w * 4 + 0) * 2 + 1] = a[0];
  }
  {
    float mx = a[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, a[j]);
    mx = wave_max(mx);
    if (l == 0) mxr[0 * 4 + w] = mx;
  }
  lds_barrier();

  float e[NPL], b1 = NEGF, b2 = NEGF;
  int ps = 0;  // ring slot of the next row to prefetch (rows come in order from 0)
  auto prefetch = [&](int t) {  // inputs of the tick that consumes emission row t
    if (t >= 0 && t < T) {
      const float* er = ering + ps * CS;
      ps = ps + 1 == NER ? 0 : ps + 1;
#pragma unroll
      for (int j = 0; j < NPL; ++j) e[j] = er[g.lab[j]];
      if (w > 0) {
        const float* bp = bnd + ((w - 1) * 4 + (t & 3)) * 2;
        b1 = bp[0];
        b2 = bp[1];
      }
    }
  };
  prefetch(0 - w);

  const int nticks = T + LAGW;
  int pa = -1, pb = -1;  // chunk whose lse phase A / B runs this tick
  for (int tau = 0; tau < nticks; ++tau) {
    if ((tau & Rm) == 0) {  // chunk boundary of the leading wave: chunk tau / R + 1 lands, + 2 is requested
      const int c = tau >> lgR;
      land(c + 1);
      issue(c + 2);
    }
    if (want_lse) {
      if (pb >= 0) lse_b(pb);
      pb = pa;
      pa = -1;
      if (tau == 0) pa = 0;
      else if (((tau - 1) & Rm) == 0) pa = ((tau - 1) >> lgR) + 1;
      if (pa >= nchunks) pa = -1;
      if (pa >= 0) lse_a(pa);
    }
    const int t = tau - w;
    if (t >= 0 && t < T) {
      const float p1 = wave_shr1(a[NPL - 1], b1);
      const float p2 = NPL == 2 ? wave_shr1(a[0], b2) : wave_shr1(p1, b2);
      float nw[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const float s1 = j == 0 ? p1 : a[0];
        const float s2 = j == 0 ? p2 : p1;
        float x0, x1, x2;
        if (UNIT) {  // self-loop and previous-node arc everywhere, weight 0
          x0 = a[j];
          x1 = s1;
          x2 = s2 + g.wi[2][j];
        } else {
          x0 = a[j] + g.wi[0][j];
          x1 = s1 + g.wi[1][j];
          x2 = s2 + g.wi[2][j];
        }
        nw[j] = lse3(x0, x1, x2) + e[j];
      }
      const int row = t + 1;
      if (row % RN == 0) {  // uniform
        const float sft = shift_of(mxr, row);
        float mx = NEGF;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          nw[j] -= sft;
          mx = fmaxf(mx, nw[j]);
        }
        off += double(sft);
        mx = wave_max(mx);
        if (l == 0) mxr[((row / RN) & 7) * 4 + w] = mx;
        if (tid == 0) P.aoff[1 + row / RN] = off;
      }
      arow += NS;
#pragma unroll
      for (int j = 0; j < NPL; ++j) a[j] = nw[j];
      if (writer) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) arow[j] = a[j];
      }
      if (NPL == 1) {
        if (l >= 62) bnd[(w * 4 + (row & 3)) * 2 + (63 - l)] = a[0];
      } else if (l == 63) {
        bnd[(w * 4 + (row & 3)) * 2 + 0] = a[NPL - 1];
        bnd[(w * 4 + (row & 3)) * 2 + 1] = a[0];
      }
    }
    prefetch(t + 1);
    lds_barrier();
  }
  if (want_lse && pb >= 0) lse_b(pb);
  // score = log sum over accept nodes of alpha[T]  (shortest.cpp:153-167)
  {
    float f = NEGF;
#pragma unroll
    for (int j = 0; j < NPL; ++j) f = fmaxf(f, g.accept[j] ? a[j] : NEGF);
    const float mx = wave_max(f);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) s += g.accept[j] ? ex2(a[j] - mx) : 0.0f;
    s = wave_sum(s);
    if (l == 0) {
      fin[2 * w] = mx;
      fin[2 * w + 1] = s;
    }
    if (tid < 16) red[tid] = normacc;
  }
  lds_barrier();
  if (tid == 0) {
    const float M = fmaxf(fmaxf(fin[0], fin[2]), fmaxf(fin[4], fin[6]));
    const bool dead = !(M > DEADF);
    float S = 0.0f;
    for (int k = 0; k < 4; ++k) S += fin[2 * k + 1] * ex2(fin[2 * k] - M);
    const double z2 = dead ? double(NEGF) : off + double(M) + double(lg2(S));
    P.aoff[0] = z2;
    P.score[0] = dead ? -__builtin_inff() : float(z2 * LN2);
    if (P.norm) {
      double n2 = 0.0;
      for (int k = 0; k < 16; ++k) n2 += red[k];
      P.norm[0] = n2 < double(DEADF) ? -__builtin_inff() : float(n2 * LN2);
    }
  }
}

// ==========================================================================================
// backward: beta[t][n] = log sum_k exp(w_k + em[t][lab(n+k)] + beta[t+1][n+k]);
//   d score / d em[t][l]   = sum over nodes m with label l of exp(alpha[t+1][m] + beta[t+1][m] - score)
//   d score / d w(n->n+k)  = sum_t exp(alpha[t][n] + w + em[t][lab(n+k)] + beta[t+1][n+k] - score)
// Virtual row v = T-1-t ascends with the ticks; wave 3 leads.
// ==========================================================================================
template <int NPL, bool UNIT, bool GRADG>
__global__ __launch_bounds__(BW) void band_backward_kernel(const BandPair* __restrict__ pairs, int R, int NSmax) {
  const BandPair P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, NS = P.NS;
  extern __shared__ float lds[];
  const BandLds L = band_lds(C, R, NSmax, true);
  float* ering = lds + L.o_ering;
  float* aring = lds + L.o_aring;
  double* obuf = reinterpret_cast<double*>(lds + L.o_obuf);
  float* gring = lds + L.o_gring;
  float* scratch = lds + L.o_scratch;
  float* bnd = lds + L.o_misc + 32;  // [4 waves][4 slots][2]
  float* mxr = bnd + 32;             // [8 slots][4 waves]
  const int CS = L.CS, NER = L.NER, NGR = L.NGR;
  const int Rm = R - 1, lgR = 31 - __builtin_clz(R);
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lag = 3 - w;
  const int l = tid & 63;
  const int m0 = tid * NPL;
  const int nchunks = (T + R - 1) / R;
  const double z2 = P.aoff[0];
  const bool dead = !(z2 > double(DEADF));
  const float ds = P.delta[0];
  const bool want_em = P.grad_em != nullptr;
  const bool vec = __builtin_amdgcn_readfirstlane(int(C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.em) & 15) == 0 &&
                                                      (!want_em || (reinterpret_cast<uintptr_t>(P.grad_em) & 15) == 0)));
  const float dn = P.delta_norm ? P.delta_norm[0] : 0.0f;
  const bool soft = P.delta_norm != nullptr && P.rowlse != nullptr;

  NodeRegs<NPL> g;
  load_nodes<NPL, true>(P, m0, g);
  bool hotn[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) hotn[j] = g.has_in[j] && g.lab[j] == P.hot;
  float b[NPL], ahi[NPL], acc[3][NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    b[j] = g.accept[j] ? 0.0f : NEGF;
    ahi[j] = m0 < NS ? P.alpha[int64_t(T) * NS + m0 + j] : NEGF;
    acc[0][j] = acc[1][j] = acc[2][j] = 0.0f;
  }
  double Ahi = P.aoff[1 + T / RN];
  double bz = -z2;  // (sum of beta's shifts) - score

  // ---- staging: chunk c covers virtual rows [cR, cR + rows), i.e. t from T-1-cR down; HBM rows t_lo ..
  Stage<8> se;   // emissions
  Stage<16> sa;  // alpha rows
  float lse_v[8];
  double off_v = 0.0;
  auto rows_of = [&](int c) { return c < nchunks ? min(R, T - c * R) : 0; };
  auto tlo_of = [&](int c) { return T - c * R - rows_of(c); };
  auto issue = [&](int c) {
    const int rows = rows_of(c), tlo = tlo_of(c);
    se.issue(P.em + int64_t(tlo) * C, rows * C, vec, tid);
    sa.issue(P.alpha + int64_t(tlo) * NS, rows * NS, true, tid);
    if (tid < rows) off_v = P.aoff[1 + (tlo + tid) / RN];
    if (soft && want_em)
      se.each(rows * C, C, vec, tid, [&](int i, int, int r, int, gtnx_f4) { lse_v[i] = P.rowlse[tlo + r]; });
  };
  // chunk row r (HBM order, ascending t) is virtual row vhi - r
  auto land = [&](int c) {
    const int rows = rows_of(c);
    if (rows <= 0) return;
    const int vhi = c * R + rows - 1;
    const int eb = vhi % NER, gb = vhi % NGR;
    se.each(rows * C, C, vec, tid, [&](int i, int, int r, int col, gtnx_f4 q) {
      int s = eb - r;
      if (s < 0) s += NER;
      float* d = ering + s * CS + col;
      const gtnx_f4 q2 = {em2(q.x), em2(q.y), em2(q.z), em2(q.w)};
      if (vec) *reinterpret_cast<gtnx_f4*>(d) = q2;
      else d[0] = q2.x;
      if (want_em) {  // the gradient row starts as the normaliser's term dn * softmax(em[t]) (or zero)
        int sg = gb - r;
        if (sg < 0) sg += NGR;
        float* dg = gring + sg * CS + col;
        gtnx_f4 p = {0.0f, 0.0f, 0.0f, 0.0f};
        if (soft) {
          const float lv = lse_v[i];
          p = gtnx_f4{dn * ex2(q2.x - lv), dn * ex2(q2.y - lv), dn * ex2(q2.z - lv), dn * ex2(q2.w - lv)};
        }
        if (vec) *reinterpret_cast<gtnx_f4*>(dg) = p;
        else dg[0] = p.x;
      }
    });
    sa.each(rows * NS, NS, true, tid, [&](int, int, int r, int col, gtnx_f4 q) {
      int s = eb - r;
      if (s < 0) s += NER;
      *reinterpret_cast<gtnx_f4*>(aring + s * NSmax + col) = q;
    });
    if (tid < rows) {
      int s = eb - tid;
      if (s < 0) s += NER;
      obuf[s] = off_v;
    }
  };
  // finished gradient rows of chunk c: LDS -> HBM
  auto drain = [&](int c) {
    const int rows = rows_of(c);
    if (!want_em || rows <= 0) return;
    const int vhi = c * R + rows - 1, gb = vhi % NGR, cnt = rows * C;
    GTNX_G float* dst = P.grad_em + int64_t(tlo_of(c)) * C;
    const float invC = 1.0f / float(C);
    if (vec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = 4 * (i * BW + tid);
        if (e < cnt) {
          const int r = row_of(e, C, invC);
          int sg = gb - r;
          if (sg < 0) sg += NGR;
          *reinterpret_cast<GTNX_G gtnx_f4*>(dst + e) = *reinterpret_cast<const gtnx_f4*>(gring + sg * CS + (e - r * C));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = i * BW + tid;
        if (e < cnt) {
          const int r = row_of(e, C, invC);
          int sg = gb - r;
          if (sg < 0) sg += NGR;
          dst[e] = gring[sg * CS + (e - r * C)];
        }
      }
    }
  };

  // ---- prologue
  issue(0);
  land(0);
  issue(1);
  if (NPL == 1) {
    // q of row v is published per tick; nothing to publish for the initial beta
  }
  {
    float mx = b[0];
#pragma unroll
    for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, b[j]);
    mx = wave_max(mx);
    if (l == 0) mxr[0 * 4 + w] = mx;
  }
  lds_barrier();

  float e[NPL], alo[NPL];
  double Alo = 0.0;
  int ps = 0, gs = 0;  // ring slots (emission / alpha, gradient) of the next row: rows come in order from 0
  auto prefetch = [&](int v) {  // inputs of the tick that consumes virtual row v
    if (v >= 0 && v < T) {
      const int s = ps;
      ps = ps + 1 == NER ? 0 : ps + 1;
      const float* er = ering + s * CS;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        e[j] = er[g.lab[j]];
        alo[j] = aring[s * NSmax + min(m0 + j, NSmax - 1)];
      }
      Alo = obuf[s];
    }
  };
  prefetch(0 - lag);

  const int nticks = T + LAGW;
  int cd = 0;  // next chunk to drain
  for (int tau = 0; tau < nticks; ++tau) {
    if (cd < nchunks && tau == (cd + 1) * R + LAGW) {  // the last wave left chunk cd a tick ago
      drain(cd);
      ++cd;
    }
    if ((tau & Rm) == 0) {
      const int c = tau >> lgR;
      land(c + 1);
      issue(c + 2);
    }
    const int v = tau - lag;
    if (v >= 0 && v < T && !dead) {
      float n1b = NEGF, n2b = NEGF;
      if (w < 3) {  // q of the next wave's first nodes for this row, published a tick ago
        const float* bp = bnd + ((w + 1) * 4 + (v & 3)) * 2;
        n1b = bp[0];
        n2b = bp[1];
      }
      float q[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) q[j] = e[j] + b[j];
      if (NPL == 1) {
        if (l < 2) bnd[(w * 4 + (v & 3)) * 2 + l] = q[0];
      } else if (l == 0) {
        bnd[(w * 4 + (v & 3)) * 2 + 0] = q[0];
        bnd[(w * 4 + (v & 3)) * 2 + 1] = q[NPL - 1];
      }
      // node posteriors at time t+1 -> gradient row t
      if (want_em) {
        const float dh = float(Ahi + bz);
        float* grow = gring + gs * CS;
        float hv = 0.0f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const float occ = ex2(ahi[j] + b[j] + dh) * ds;
          const bool direct = g.has_in[j] && !hotn[j];
          if (hotn[j]) hv += occ;
          lds_add(direct ? grow + g.lab[j] : scratch + tid, occ);
        }
        if (P.hot >= 0) {
          hv = wave_sum63(hv);
          lds_add(l == 63 ? grow + P.hot : scratch + tid, hv);
        }
      }
      const float n1 = wave_shl1(q[0], n1b);
      const float n2 = NPL == 2 ? wave_shl1(q[NPL - 1], n2b) : wave_shl1(n1, n2b);
      const float dl = GRADG ? float(Alo + bz) : 0.0f;
      float nb[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const float s1 = j + 1 < NPL ? q[NPL - 1] : n1;
        const float s2 = j + 1 < NPL ? n1 : (NPL == 2 ? n2 : n2);
        float y0, y1, y2;
        if (UNIT) {
          y0 = q[j];
          y1 = s1;
          y2 = s2 + g.wo[2][j];
        } else {
          y0 = q[j] + g.wo[0][j];
          y1 = s1 + g.wo[1][j];
          y2 = s2 + g.wo[2][j];
        }
        if (GRADG) {
          // the exponentials of the log-sum-exp are the arc posteriors up to one factor per node
          const float mx = fmaxf(fmaxf(y0, y1), y2);
          const float e0 = ex2(y0 - mx), e1 = ex2(y1 - mx), e2 = ex2(y2 - mx);
          nb[j] = mx + lg2(e0 + e1 + e2);
          const float f = ex2(alo[j] + mx + dl);
          acc[0][j] += e0 * f;
          acc[1][j] += e1 * f;
          acc[2][j] += e2 * f;
        } else {
          nb[j] = lse3(y0, y1, y2);
        }
      }
      const int row = v + 1;
      if (row % RN == 0) {
        const float sft = shift_of(mxr, row);
        float mx = NEGF;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          nb[j] -= sft;
          mx = fmaxf(mx, nb[j]);
        }
        bz += double(sft);
        mx = wave_max(mx);
        if (l == 0) mxr[((row / RN) & 7) * 4 + w] = mx;
      }
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        b[j] = nb[j];
        ahi[j] = alo[j];
      }
      Ahi = Alo;
      gs = gs + 1 == NGR ? 0 : gs + 1;
    }
    prefetch(v + 1);
    lds_barrier();
  }
  for (; cd < nchunks; ++cd) drain(cd);
  if (GRADG && P.grad_fixed && !dead) {
#pragma unroll
    for (int j = 0; j < NPL; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (g.ao[k][j] >= 0) P.grad_fixed[g.ao[k][j]] = acc[k][j] * ds;
  }
}

template <int NPL>
void launch_fwd_npl(const BandPair* d, int n, int R, int ns, size_t lds, bool unit, hipStream_t st) {
  if (unit) hipLaunchKernelGGL((band_forward_kernel<NPL, true>), dim3(n), dim3(BW), lds, st, d, R, ns);
  else hipLaunchKernelGGL((band_forward_kernel<NPL, false>), dim3(n), dim3(BW), lds, st, d, R, ns);
}
template <int NPL>
void launch_bwd_npl(const BandPair* d, int n, int R, int ns, size_t lds, bool unit, bool gradg, hipStream_t st) {
  if (unit) {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, true, true>), dim3(n), dim3(BW), lds, st, d, R, ns);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, true, false>), dim3(n), dim3(BW), lds, st, d, R, ns);
  } else {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, false, true>), dim3(n), dim3(BW), lds, st, d, R, ns);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, false, false>), dim3(n), dim3(BW), lds, st, d, R, ns);
  }
}

template <class K>
void big_lds(K kern) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
}
void band_attrs() {
  static bool done = false;
  if (done) return;
  done = true;
#define GTNX_BAND_ATTR(NPL)                           \
  big_lds(band_backward_kernel<NPL, true, true>);     \
  big_lds(band_backward_kernel<NPL, true, false>);    \
  big_lds(band_backward_kernel<NPL, false, true>);    \
  big_lds(band_backward_kernel<NPL, false, false>);   \
  big_lds(band_forward_kernel<NPL, true>);            \
  big_lds(band_forward_kernel<NPL, false>);
  GTNX_BAND_ATTR(1)
  GTNX_BAND_ATTR(2)
#undef GTNX_BAND_ATTR
}

} // namespace

int band_max_nodes() { return 512; }
int band_max_labels() { return 1024; }
int band_npl(int max_nodes) { return max_nodes <= 256 ? 1 : 2; }
int band_row_stride(int N, int) { return (N + 3) / 4 * 4; }
int band_rows_per_chunk(int C, bool backward) {
  const int cap = backward ? 2048 : 4096;
  for (int r = backward ? 8 : 16; r > 2; r /= 2)
    if (r * C <= cap) return r;
  return 2;
}

void launch_band_forward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, hipStream_t st) {
  if (n <= 0) return;
  band_attrs();
  const int R = band_rows_per_chunk(C, false);
  const size_t lds = 4 * size_t(band_lds(C, R, max_NS, false).total) + 64;
  if (npl == 1) launch_fwd_npl<1>(d_pairs, n, R, max_NS, lds, unit, st);
  else launch_fwd_npl<2>(d_pairs, n, R, max_NS, lds, unit, st);
}

// every pair of the launch shares C; max_NS: largest alpha row stride of the launch
void launch_band_backward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, bool gradg, hipStream_t st) {
  if (n <= 0) return;
  band_attrs();
  const int R = band_rows_per_chunk(C, true);
  const size_t lds = 4 * size_t(band_lds(C, R, max_NS, true).total) + 64;
  if (npl == 1) launch_bwd_npl<1>(d_pairs, n, R, max_NS, lds, unit, gradg, st);
  else launch_bwd_npl<2>(d_pairs, n, R, max_NS, lds, unit, gradg, st);
}

} // namespace gtnx
