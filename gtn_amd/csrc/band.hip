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
// One workgroup per utterance, 4 waves with different jobs:
//   wave 0  (the sweeper) owns ALL nodes of G, NPL consecutive nodes per lane.  The T
//           dependent steps never leave the wave: no barrier, no LDS round trip for the
//           recursion -- neighbours inside a lane are registers, the two values that
//           cross a lane boundary move with one DPP wave shift each.  Scores are in log2
//           units (bare v_exp_f32 / v_log_f32), "minus infinity" is -1e30 so no step needs
//           an inf / NaN guard, and every RN steps the row is shifted by its maximum (the
//           shifts are summed in fp64) so the magnitudes stay O(10) and float32 keeps
//           ~1e-6 relative accuracy on every posterior for any T.
//   waves 1-3 (the stagers) do everything that is not the recursion: HBM -> registers ->
//           LDS for the emission chunk after next (16-byte loads a whole chunk ahead),
//           the per-row log-sum-exp of the normaliser, and in the backward kernel the
//           alpha rows, the softmax term of the gradient rows and their single coalesced
//           store.  They meet the sweeper at ONE LDS-only barrier per chunk of rows.
// HBM traffic per utterance: forward 4TC + 4(T+1)NS, backward 8TC + 4(T+1)NS (+ G).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr float NEGF = -1.0e30f;        // log-domain zero
constexpr float DEADF = -1.0e29f;       // anything below is "no path"
constexpr float LOG2E = 1.44269504088896340736f;
constexpr double LN2 = 0.693147180559945309417;
constexpr int BW = 256;    // lanes per workgroup
constexpr int NSTG = 192;  // stager lanes (waves 1..3)
constexpr int SL4 = 6;     // 16-byte slots per stager lane and chunk (4096 floats / 4 / 192)
constexpr int SL1 = 22;    // 4-byte slots (chunk base not 16-byte aligned)
constexpr int RN = 4;      // steps between renormalisations of the running row

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }

// lane i <- lane i-1 (lane 0 keeps `fill`) / lane i <- lane i+1 (lane 63 keeps `fill`)
__device__ __forceinline__ float wave_shr1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
#define GTNX_BAND_DPP(op, x, ctrl, rmask, idv) \
  x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(idv), __float_as_int(x), ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float fadd(float a, float b) { return a + b; }
// wave64 reductions by DPP row shifts / broadcasts (all 64 lanes active); result uniform
__device__ __forceinline__ float wave_max(float x) {
  GTNX_BAND_DPP(fmaxf, x, 0x111, 0xf, x);
  GTNX_BAND_DPP(fmaxf, x, 0x112, 0xf, x);
  GTNX_BAND_DPP(fmaxf, x, 0x114, 0xf, x);
  GTNX_BAND_DPP(fmaxf, x, 0x118, 0xf, x);
  GTNX_BAND_DPP(fmaxf, x, 0x142, 0xa, x);
  GTNX_BAND_DPP(fmaxf, x, 0x143, 0xc, x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_sum63(float x) {  // lane 63 holds the total
  GTNX_BAND_DPP(fadd, x, 0x111, 0xf, 0.0f);
  GTNX_BAND_DPP(fadd, x, 0x112, 0xf, 0.0f);
  GTNX_BAND_DPP(fadd, x, 0x114, 0xf, 0.0f);
  GTNX_BAND_DPP(fadd, x, 0x118, 0xf, 0.0f);
  GTNX_BAND_DPP(fadd, x, 0x142, 0xa, 0.0f);
  GTNX_BAND_DPP(fadd, x, 0x143, 0xc, 0.0f);
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum63(x)), 63));
}

// log2(2^x0 + 2^x1 + 2^x2): the largest term is exactly 1, so two v_exp_f32 and one v_log_f32
__device__ __forceinline__ float lse3(float x0, float x1, float x2) {
  const float mx = fmaxf(fmaxf(x0, x1), x2);
  const float md = __builtin_amdgcn_fmed3f(x0, x1, x2);
  const float mn = fminf(fminf(x0, x1), x2);
  return mx + lg2(1.0f + ex2(md - mx) + ex2(mn - mx));
}

// what a sweeper lane knows about its NPL nodes
template <int NPL>
struct NodeRegs {
  int lab[NPL];     // matched label (0 when the node has no in-arc; its weights are NEGF then)
  float wi[3][NPL]; // in-arc from n-k, log2 units (NEGF: no such arc)
  float wo[3][NPL]; // out-arc to n+k
  int ai[3][NPL];   // arc ids of the in-arcs (-1: none)
  int ao[3][NPL];   // arc ids of the out-arcs
  bool start[NPL], accept[NPL];
};
template <int NPL, bool WANT_OUT>
__device__ __forceinline__ void load_nodes(const BandPair& P, int lane, NodeRegs<NPL>& g) {
  const GTNX_G gtnx_i4* nodes = reinterpret_cast<const GTNX_G gtnx_i4*>(P.nodes);
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int m = lane * NPL + j;
    g.lab[j] = 0;
    g.start[j] = g.accept[j] = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g.wi[k][j] = g.wo[k][j] = NEGF;
      g.ai[k][j] = g.ao[k][j] = -1;
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
          g.ai[k][j] = a[k];
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

template <int NPL>
__device__ __forceinline__ void store_row(GTNX_G float* p, const float (&a)[NPL]) {
  if constexpr (NPL == 1) {
    p[0] = a[0];
  } else if constexpr (NPL == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<GTNX_G f2*>(p) = f2{a[0], a[1]};
  } else {
#pragma unroll
    for (int j = 0; j < NPL; j += 4) *reinterpret_cast<GTNX_G gtnx_f4*>(p + j) = gtnx_f4{a[j], a[j + 1], a[j + 2], a[j + 3]};
  }
}
template <int NPL>
__device__ __forceinline__ void load_row_lds(const float* p, float (&a)[NPL]) {
  if constexpr (NPL == 1) {
    a[0] = p[0];
  } else if constexpr (NPL == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = *reinterpret_cast<const f2*>(p);
    a[0] = v.x;
    a[1] = v.y;
  } else {
#pragma unroll
    for (int j = 0; j < NPL; j += 4) {
      const gtnx_f4 v = *reinterpret_cast<const gtnx_f4*>(p + j);
      a[j] = v.x;
      a[j + 1] = v.y;
      a[j + 2] = v.z;
      a[j + 3] = v.w;
    }
  }
}

// ---- stager side: a contiguous range of `cnt` floats, HBM -> registers -> LDS --------------
struct Stage {
  float v[4 * SL4];  // 24 >= SL1
};
__device__ __forceinline__ void stage_issue(Stage& s, const GTNX_G float* src, int cnt, bool vec, int sl) {
  if (vec) {
#pragma unroll
    for (int i = 0; i < SL4; ++i) {
      const int e = 4 * (i * NSTG + sl);
      gtnx_f4 q = {0.0f, 0.0f, 0.0f, 0.0f};
      if (e < cnt) q = *reinterpret_cast<const GTNX_G gtnx_f4*>(src + e);
      s.v[4 * i] = q.x;
      s.v[4 * i + 1] = q.y;
      s.v[4 * i + 2] = q.z;
      s.v[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SL1; ++i) {
      const int e = i * NSTG + sl;
      s.v[i] = e < cnt ? src[e] : 0.0f;
    }
  }
}
// emissions land in log2 units with -inf clamped to the finite log-zero
__device__ __forceinline__ float em2(float x) { return fmaxf(x * LOG2E, NEGF); }
template <bool SCALE>
__device__ __forceinline__ void stage_land(const Stage& s, float* dst, int cnt, bool vec, int sl) {
  if (vec) {
#pragma unroll
    for (int i = 0; i < SL4; ++i) {
      const int e = 4 * (i * NSTG + sl);
      if (e < cnt) {
        gtnx_f4 q = {s.v[4 * i], s.v[4 * i + 1], s.v[4 * i + 2], s.v[4 * i + 3]};
        if (SCALE) q = gtnx_f4{em2(q.x), em2(q.y), em2(q.z), em2(q.w)};
        *reinterpret_cast<gtnx_f4*>(dst + e) = q;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < SL1; ++i) {
      const int e = i * NSTG + sl;
      if (e < cnt) dst[e] = SCALE ? em2(s.v[i]) : s.v[i];
    }
  }
}

// row-wise log2-sum-exp2 of `rows` staged emission rows, one wave per row; wave `w` of `nw`
__device__ __forceinline__ void row_lse(const float* e0, int rows, int C, int t0, int w, int nw, int lane,
                                        GTNX_G float* rowlse, double& acc) {
  for (int r = w; r < rows; r += nw) {
    const float* e = e0 + r * C;
    float mx = NEGF;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, e[c]);
    mx = wave_max(mx);
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += ex2(e[c] - mx);
    s = wave_sum(s);
    const float l = mx + lg2(s);
    acc += double(l);
    if (rowlse && lane == 0) rowlse[t0 + r] = l;
  }
}

// ==========================================================================================
// forward: alpha[t+1][m] = em[t][lab m] + log sum_k exp(alpha[t][m-k] + w_k(m))
// ==========================================================================================
template <int NPL, bool UNIT>
__global__ __launch_bounds__(BW) void band_forward_kernel(const BandPair* __restrict__ pairs, int R) {
  const BandPair P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, NS = P.NS;
  extern __shared__ float lds[];
  const int RC = R * C;
  float* ebuf = lds;  // [2][RC]
  __shared__ double red[4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nchunks = (T + R - 1) / R;

  if (wave == 0) {
    // ------------------------------------------------------------------ sweeper
    NodeRegs<NPL> g;
    load_nodes<NPL, false>(P, lane, g);
    const bool writer = lane * NPL < NS;
    float a[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) a[j] = g.start[j] ? 0.0f : NEGF;
    double off = 0.0;
    GTNX_G float* arow = P.alpha + lane * NPL;
    if (writer) store_row<NPL>(arow, a);
    if (lane == 0) P.aoff[0] = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      lds_barrier();  // chunk k is in ebuf[k & 1]
      const float* e0 = ebuf + (k & 1) * RC;
      const int t0 = k * R, rows = min(R, T - t0);
      float e[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) e[j] = e0[g.lab[j]];
      for (int r = 0; r < rows; ++r) {
        const int t = t0 + r;
        float en[NPL];  // next row's emissions, in flight during this step
        const float* e1 = e0 + min(r + 1, rows - 1) * C;
#pragma unroll
        for (int j = 0; j < NPL; ++j) en[j] = e1[g.lab[j]];
        const float p1 = wave_shr1(a[NPL - 1], NEGF);
        const float p2 = NPL >= 2 ? wave_shr1(a[NPL >= 2 ? NPL - 2 : 0], NEGF) : wave_shr1(p1, NEGF);
        float nw[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const float s1 = j >= 1 ? a[j >= 1 ? j - 1 : 0] : p1;
          const float s2 = j >= 2 ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
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
        if ((t + 1) % RN == 0) {  // uniform
          float mx = nw[0];
#pragma unroll
          for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, nw[j]);
          mx = wave_max(mx);
          if (mx > DEADF) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) nw[j] = fmaxf(nw[j] - mx, NEGF);
            off += double(mx);
          }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          a[j] = nw[j];
          e[j] = en[j];
        }
        arow += NS;
        if (writer) store_row<NPL>(arow, a);
        if (lane == 0) P.aoff[t + 1] = off;
      }
    }
    // score = log sum over accept nodes of alpha[T]  (shortest.cpp:153-167)
    float f = NEGF;
#pragma unroll
    for (int j = 0; j < NPL; ++j) f = fmaxf(f, g.accept[j] ? a[j] : NEGF);
    const float mx = wave_max(f);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) s += g.accept[j] ? ex2(a[j] - mx) : 0.0f;
    s = wave_sum(s);
    const bool dead = !(mx > DEADF);
    const double z2 = dead ? double(NEGF) : off + double(mx) + double(lg2(s));
    if (lane == 0) {
      P.aoff[T + 1] = z2;
      P.score[0] = dead ? -__builtin_inff() : float(z2 * LN2);
    }
    lds_barrier();  // the stagers' partial sums of the normaliser
    if (P.norm && lane == 0) {
      const double n2 = red[1] + red[2] + red[3];
      P.norm[0] = n2 < double(DEADF) ? -__builtin_inff() : float(n2 * LN2);
    }
  } else {
    // ------------------------------------------------------------------ stagers
    const int sl = threadIdx.x - 64;
    const bool vec = __builtin_amdgcn_readfirstlane(int(C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.em) & 15) == 0));
    const bool want_lse = P.norm != nullptr || P.rowlse != nullptr;
    Stage st;
    double acc = 0.0;
    auto cnt_of = [&](int k) { return k < nchunks ? min(R, T - k * R) * C : 0; };
    stage_issue(st, P.em, cnt_of(0), vec, sl);
    stage_land<true>(st, ebuf, cnt_of(0), vec, sl);
    stage_issue(st, P.em + int64_t(RC), cnt_of(1), vec, sl);
    for (int k = 0; k < nchunks; ++k) {
      lds_barrier();  // chunk k readable; the sweeper is done with chunk k-1, i.e. with ebuf[(k+1) & 1]
      stage_land<true>(st, ebuf + ((k + 1) & 1) * RC, cnt_of(k + 1), vec, sl);
      stage_issue(st, P.em + int64_t(k + 2) * RC, cnt_of(k + 2), vec, sl);
      if (want_lse) row_lse(ebuf + (k & 1) * RC, min(R, T - k * R), C, k * R, wave - 1, 3, lane, P.rowlse, acc);
    }
    if (lane == 0) red[wave] = acc;
    lds_barrier();
  }
}

// ==========================================================================================
// backward: beta[t][n] = log sum_k exp(w_k + em[t][lab(n+k)] + beta[t+1][n+k]);
//   d score / d em[t][l]   = sum over nodes m with label l of exp(alpha[t+1][m] + beta[t+1][m] - score)
//   d score / d w(n->n+k)  = sum_t exp(alpha[t][n] + w + em[t][lab(n+k)] + beta[t+1][n+k] - score)
// ==========================================================================================
template <int NPL, bool UNIT, bool GRADG>
__global__ __launch_bounds__(BW) void band_backward_kernel(const BandPair* __restrict__ pairs, int R) {
  const BandPair P = pairs[blockIdx.x];
  const int T = P.T, C = P.C, NS = P.NS;
  const int Cp = C + 64;  // gradient rows carry 64 scratch columns (one per lane) for redirected adds
  extern __shared__ float lds[];
  const int RC = R * C, RG = R * Cp, RA = R * NS;
  float* abuf = lds;                 // [2][RA]  alpha rows t0 .. t0+rows-1 (16-byte rows)
  double* obuf = reinterpret_cast<double*>(abuf + 2 * RA);  // [2][R] their offsets
  float* ebuf = abuf + 2 * RA + 4 * R;  // [2][RC]  emissions, log2 units
  float* grow = ebuf + 2 * RC;       // [2][RG]  gradient rows of the chunk
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nchunks = (T + R - 1) / R;
  const double z2 = P.aoff[T + 1];
  const bool dead = !(z2 > double(DEADF));
  const float ds = P.delta[0];
  const bool want_em = P.grad_em != nullptr;

  if (wave == 0) {
    // ------------------------------------------------------------------ sweeper
    NodeRegs<NPL> g;
    load_nodes<NPL, true>(P, lane, g);
    int gcol[NPL];
    bool hotn[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int m = lane * NPL + j;
      const bool has_in = m < P.N && (g.ai[0][j] >= 0 || g.ai[1][j] >= 0 || g.ai[2][j] >= 0);
      hotn[j] = has_in && g.lab[j] == P.hot;
      gcol[j] = (has_in && !hotn[j]) ? g.lab[j] : C + lane;
    }
    float b[NPL], ahi[NPL];
    float acc[3][NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      b[j] = g.accept[j] ? 0.0f : NEGF;
      acc[0][j] = acc[1][j] = acc[2][j] = 0.0f;
    }
    {
      const GTNX_G float* pa = P.alpha + int64_t(T) * NS + lane * NPL;
#pragma unroll
      for (int j = 0; j < NPL; ++j) ahi[j] = lane * NPL < NS ? pa[j] : NEGF;
    }
    double Ahi = P.aoff[T];
    double bz = -z2;  // (sum of beta's shifts) - score
    for (int i = 0; i < nchunks; ++i) {
      lds_barrier();  // chunk i (counted from the end) is staged
      const int k = nchunks - 1 - i, t0 = k * R, rows = min(R, T - t0);
      const float* eb = ebuf + (i & 1) * RC;
      const float* ab = abuf + (i & 1) * RA + lane * NPL;
      float* gb = grow + (i & 1) * RG;
      const double* ob = obuf + (i & 1) * R;
      if (dead) continue;  // no accepting path: only the normaliser's term reaches the gradient rows
      for (int r = rows - 1; r >= 0; --r) {
        const int t = t0 + r;
        float e[NPL], alo[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) e[j] = eb[r * C + g.lab[j]];
        load_row_lds<NPL>(ab + r * NS, alo);
        const double Alo = ob[r];
        // node posteriors at time t+1 -> gradient row t
        if (want_em) {
          const float dh = float(Ahi + bz);
          float hv = 0.0f;
#pragma unroll
          for (int j = 0; j < NPL; ++j) {
            const float occ = ex2(ahi[j] + b[j] + dh) * ds;
            if (hotn[j]) hv += occ;
            atomicAdd(gb + r * Cp + gcol[j], hotn[j] ? 0.0f : occ);
          }
          if (P.hot >= 0) {
            hv = wave_sum63(hv);
            if (lane == 63) atomicAdd(gb + r * Cp + P.hot, hv);
          }
        }
        float q[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) q[j] = e[j] + b[j];
        const float n1 = wave_shl1(q[0], NEGF);
        const float n2 = NPL >= 2 ? wave_shl1(q[NPL >= 2 ? 1 : 0], NEGF) : wave_shl1(n1, NEGF);
        const float dl = GRADG ? float(Alo + bz) : 0.0f;
        float nb[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          const float s1 = j + 1 < NPL ? q[j + 1 < NPL ? j + 1 : 0] : n1;
          const float s2 = j + 2 < NPL ? q[j + 2 < NPL ? j + 2 : 0] : (j + 2 == NPL ? n1 : n2);
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
        if (t % RN == 0) {
          float mx = nb[0];
#pragma unroll
          for (int j = 1; j < NPL; ++j) mx = fmaxf(mx, nb[j]);
          mx = wave_max(mx);
          if (mx > DEADF) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) nb[j] = fmaxf(nb[j] - mx, NEGF);
            bz += double(mx);
          }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          b[j] = nb[j];
          ahi[j] = alo[j];
        }
        Ahi = Alo;
      }
    }
    lds_barrier();
    if (GRADG && P.grad_fixed && !dead) {
#pragma unroll
      for (int j = 0; j < NPL; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (g.ao[k][j] >= 0) P.grad_fixed[g.ao[k][j]] = acc[k][j] * ds;
    }
  } else {
    // ------------------------------------------------------------------ stagers
    const int sl = threadIdx.x - 64;
    const bool vec = __builtin_amdgcn_readfirstlane(int(C % 4 == 0 && (reinterpret_cast<uintptr_t>(P.em) & 15) == 0 &&
                                                        (!want_em || (reinterpret_cast<uintptr_t>(P.grad_em) & 15) == 0)));
    const float dn = P.delta_norm ? P.delta_norm[0] : 0.0f;
    const bool soft = P.delta_norm != nullptr && P.rowlse != nullptr;
    Stage se, sa;
    float lse_v[SL1];  // row log-sum-exp of the element(s) of each slot (softmax term)
    double off_v = 0.0;
    auto rows_of = [&](int i) { return i < nchunks ? min(R, T - (nchunks - 1 - i) * R) : 0; };
    auto t0_of = [&](int i) { return (nchunks - 1 - i) * R; };
    // element e of a chunk -> (row, column)
    auto row_of = [&](int e) {
      int r = int(float(e) * (1.0f / float(C)));
      if (r * C > e) --r;
      if ((r + 1) * C <= e) ++r;
      return r;
    };
    auto issue = [&](int i) {
      const int rows = rows_of(i), t0 = t0_of(i);
      stage_issue(se, P.em + int64_t(t0) * C, rows * C, vec, sl);
      stage_issue(sa, P.alpha + int64_t(t0) * NS, rows * NS, true, sl);
      if (sl < rows) off_v = P.aoff[t0 + sl];
      if (soft && want_em) {
        const int ns = vec ? SL4 : SL1;
#pragma unroll
        for (int s = 0; s < SL1; ++s) {
          if (s < ns) {
            const int e = vec ? 4 * (s * NSTG + sl) : s * NSTG + sl;
            lse_v[s] = e < rows * C ? P.rowlse[t0 + row_of(e)] : 0.0f;
          }
        }
      }
    };
    // gradient rows of chunk i start as the normaliser's term dn * softmax(em[t]) (or zero)
    auto prefill = [&](int i) {
      if (!want_em) return;
      const int rows = rows_of(i), cnt = rows * C;
      float* gb = grow + (i & 1) * RG;
      if (vec) {
#pragma unroll
        for (int s = 0; s < SL4; ++s) {
          const int e = 4 * (s * NSTG + sl);
          if (e < cnt) {
            const int r = row_of(e), c = e - r * C;
            gtnx_f4 q = {0.0f, 0.0f, 0.0f, 0.0f};
            if (soft) {
              const float l = lse_v[s];
              q = gtnx_f4{dn * ex2(em2(se.v[4 * s]) - l), dn * ex2(em2(se.v[4 * s + 1]) - l),
                          dn * ex2(em2(se.v[4 * s + 2]) - l), dn * ex2(em2(se.v[4 * s + 3]) - l)};
            }
            *reinterpret_cast<gtnx_f4*>(gb + r * Cp + c) = q;
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < SL1; ++s) {
          const int e = s * NSTG + sl;
          if (e < cnt) {
            const int r = row_of(e), c = e - r * C;
            gb[r * Cp + c] = soft ? dn * ex2(em2(se.v[s]) - lse_v[s]) : 0.0f;
          }
        }
      }
    };
    // finished gradient rows of chunk i: LDS -> HBM, same element -> lane map as prefill
    auto drain = [&](int i) {
      if (!want_em) return;
      const int rows = rows_of(i), cnt = rows * C;
      const float* gb = grow + (i & 1) * RG;
      GTNX_G float* dst = P.grad_em + int64_t(t0_of(i)) * C;
      if (vec) {
#pragma unroll
        for (int s = 0; s < SL4; ++s) {
          const int e = 4 * (s * NSTG + sl);
          if (e < cnt) {
            const int r = row_of(e), c = e - r * C;
            *reinterpret_cast<GTNX_G gtnx_f4*>(dst + e) = *reinterpret_cast<const gtnx_f4*>(gb + r * Cp + c);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < SL1; ++s) {
          const int e = s * NSTG + sl;
          if (e < cnt) {
            const int r = row_of(e), c = e - r * C;
            dst[e] = gb[r * Cp + c];
          }
        }
      }
    };
    auto land = [&](int i) {
      const int rows = rows_of(i);
      stage_land<true>(se, ebuf + (i & 1) * RC, rows * C, vec, sl);
      stage_land<false>(sa, abuf + (i & 1) * RA, rows * NS, true, sl);
      if (sl < rows) obuf[(i & 1) * R + sl] = off_v;
      prefill(i);
    };
    issue(0);
    land(0);
    issue(1);
    for (int i = 0; i < nchunks; ++i) {
      lds_barrier();  // chunk i is the sweeper's; it is done with chunk i-1
      if (i >= 1) drain(i - 1);
      land(i + 1);  // into the buffers drain(i-1) just emptied, by the same lanes
      issue(i + 2);
    }
    lds_barrier();
    if (nchunks >= 1) drain(nchunks - 1);
  }
}

size_t band_lds_bytes(int C, int R, int NS, bool backward) {
  size_t fl = 2 * size_t(R) * C;
  if (backward) fl += 2 * size_t(R) * (C + 64) + 2 * size_t(R) * NS + 4 * size_t(R);
  return 4 * fl + 64;
}

template <int NPL>
void launch_fwd_npl(const BandPair* d, int n, int R, size_t lds, bool unit, hipStream_t st) {
  if (unit) hipLaunchKernelGGL((band_forward_kernel<NPL, true>), dim3(n), dim3(BW), lds, st, d, R);
  else hipLaunchKernelGGL((band_forward_kernel<NPL, false>), dim3(n), dim3(BW), lds, st, d, R);
}
template <int NPL>
void launch_bwd_npl(const BandPair* d, int n, int R, size_t lds, bool unit, bool gradg, hipStream_t st) {
  if (unit) {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, true, true>), dim3(n), dim3(BW), lds, st, d, R);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, true, false>), dim3(n), dim3(BW), lds, st, d, R);
  } else {
    if (gradg) hipLaunchKernelGGL((band_backward_kernel<NPL, false, true>), dim3(n), dim3(BW), lds, st, d, R);
    else hipLaunchKernelGGL((band_backward_kernel<NPL, false, false>), dim3(n), dim3(BW), lds, st, d, R);
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
  big_lds(band_backward_kernel<NPL, false, false>);
  GTNX_BAND_ATTR(1)
  GTNX_BAND_ATTR(2)
  GTNX_BAND_ATTR(4)
  GTNX_BAND_ATTR(8)
#undef GTNX_BAND_ATTR
}

} // namespace

int band_max_nodes() { return 512; }
int band_max_labels() { return 4096; }
int band_npl(int max_nodes) { return max_nodes <= 64 ? 1 : (max_nodes <= 128 ? 2 : (max_nodes <= 256 ? 4 : 8)); }
int band_row_stride(int N, int npl) {
  const int q = npl < 4 ? 4 : npl;
  return (N + q - 1) / q * q;
}
int band_rows_per_chunk(int C, bool backward) {
  const int cap = backward ? 2048 : 4096, top = backward ? 8 : 16;
  return std::max(1, std::min(top, cap / std::max(C, 1)));
}

void launch_band_forward(const BandPair* d_pairs, int n, int npl, int C, bool unit, hipStream_t st) {
  if (n <= 0) return;
  const int R = band_rows_per_chunk(C, false);
  const size_t lds = band_lds_bytes(C, R, 0, false);
  switch (npl) {
    case 1: launch_fwd_npl<1>(d_pairs, n, R, lds, unit, st); break;
    case 2: launch_fwd_npl<2>(d_pairs, n, R, lds, unit, st); break;
    case 4: launch_fwd_npl<4>(d_pairs, n, R, lds, unit, st); break;
    default: launch_fwd_npl<8>(d_pairs, n, R, lds, unit, st); break;
  }
}

// every pair of the launch shares C and the row stride NS
void launch_band_backward(const BandPair* d_pairs, int n, int npl, int C, int NS, bool unit, bool gradg, hipStream_t st) {
  if (n <= 0) return;
  band_attrs();
  const int R = band_rows_per_chunk(C, true);
  const size_t lds = band_lds_bytes(C, R, NS, true);
  switch (npl) {
    case 1: launch_bwd_npl<1>(d_pairs, n, R, lds, unit, gradg, st); break;
    case 2: launch_bwd_npl<2>(d_pairs, n, R, lds, unit, gradg, st); break;
    case 4: launch_bwd_npl<4>(d_pairs, n, R, lds, unit, gradg, st); break;
    default: launch_bwd_npl<8>(d_pairs, n, R, lds, unit, gradg, st); break;
  }
}

} // namespace gtnx
