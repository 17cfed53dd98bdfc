// compose.hip -- WFST composition / intersection on the GPU.
//
// Replaces gtn/functions/compose.cpp:377-522 (detail::compose), :64-104
// (findReachable), :108-208 (node/arc emission incl. epsilon handling), the three
// matchers (:211-374) and the gradient scatter (:496-518).
//
// Contract kept bit-exact with the reference: the composed graph's NODE IDS are
// the reference's BFS discovery order and its ARC IDS the reference's emission
// order (grouped by source node in id order; within a node: matcher order, then
// first-graph epsilon arcs, then second-graph epsilon arcs).  That holds because
//   * the reference's FIFO pops nodes in id order, so arcs are src-sorted, and
//   * a node's id is the rank of the first arc (in arc order) that reaches it,
// both of which a level-synchronous BFS reproduces with prefix sums.
//
// Execution model: ONE persistent workgroup per (g1, g2) pair; a batch is one
// launch of B workgroups.  Phase B marks co-reachable state pairs by a backward
// frontier BFS over the dense N1*N2 `state` table in HBM (atomic test-and-set,
// frontier queue appended through an LDS cursor).  Phase F expands the forward
// frontier level by level: each lane owns one frontier node, counts its valid
// matches, a workgroup prefix sum assigns arc slots, lanes emit the SoA arc
// fields (coalesced per field), first-touch ownership of new state pairs is
// resolved with atomicMax "claims" + a second prefix sum, then destinations are
// patched.  A second, fully parallel pass builds the in-arc CSR (with src ids
// and weights permuted into row order for the forward-score kernel).
#include <hip/hip_runtime.h>

#include <climits>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr int kBlock = 256;
constexpr int ST_UNREACH = INT_MIN;
constexpr int ST_REACH = INT_MIN + 1;
constexpr int EPS = -1;

__device__ __forceinline__ int claim_of(int r) { return -2 - r; }

// `state` is touched by L2 atomics; read/write it L1-bypassing so a lane never
// sees a line cached before another lane's atomic (MI355X: sc1 loads/stores).
__device__ __forceinline__ int ld_state(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- graph accessors (explicit SoA/CSR or implicit linear chain) -------------
// An adjacency entry is fetched as ONE record {ilabel, olabel, other node, arc id}:
// host-built graphs carry packed 16-byte records in list order (out_rec / in_rec),
// so walking a node's arcs is "offsets, then independent 16 B loads" -- a
// dependent-load chain of depth 2 instead of list -> label -> node (depth 4),
// which is what bounds a BFS level (every hop is an L1/L2 round trip).
struct Rec {
  int il, ol, node, arc;
};
struct Adj {
  int n;
  int base;
  int node;
};
// LDS-resident copy of a small explicit graph's adjacency for the current phase
// (in-lists during phase B, out-lists during phase F): same member names as
// DGraph so the accessors below are generic over both.
#define GTNX_L __attribute__((address_space(3)))
struct LGraph {
  int N, A, M, C, flags;
  const GTNX_L int* in_off;
  const GTNX_L int* out_off;
  const GTNX_L gtnx_i4* in_rec;
  const GTNX_L gtnx_i4* out_rec;
  const GTNX_L uint8_t* nflags;
};
template <bool LIN, class G>
__device__ __forceinline__ Adj out_adj(const G& g, int node) {
  Adj a;
  a.node = node;
  if (LIN) {
    a.n = node < g.M ? g.C : 0;
    a.base = node * g.C;
  } else {
    a.base = g.out_off[node];
    a.n = g.out_off[node + 1] - a.base;
  }
  return a;
}
template <bool LIN, class G>
__device__ __forceinline__ Adj in_adj(const G& g, int node) {
  Adj a;
  a.node = node;
  if (LIN) {
    a.n = node > 0 ? g.C : 0;
    a.base = (node - 1) * g.C;
  } else {
    a.base = g.in_off[node];
    a.n = g.in_off[node + 1] - a.base;
  }
  return a;
}
template <bool IN, bool LIN, class G>
__device__ __forceinline__ Rec adj_rec(const G& g, const Adj& a, int k) {
  Rec r;
  if (LIN) {
    r.il = r.ol = k;
    r.arc = a.base + k;
    r.node = IN ? a.node - 1 : a.node + 1;
    return r;
  }
  // every explicit graph entering compose carries records (host-built: made at
  // upload; device-built: build_records_kernel below), so this is one 16 B load
  const gtnx_i4 v = (IN ? g.in_rec : g.out_rec)[a.base + k];
  r.il = v.x; r.ol = v.y; r.node = v.z; r.arc = v.w;
  return r;
}
template <bool LIN, class G>
__device__ __forceinline__ bool g_start(const G& g, int n) {
  return LIN ? n == 0 : (g.nflags[n] & NF_START) != 0;
}
template <bool LIN, class G>
__device__ __forceinline__ bool g_accept(const G& g, int n) {
  return LIN ? (g.M > 0 && n == g.M) : (g.nflags[n] & NF_ACCEPT) != 0;
}
template <bool LIN>
__device__ __forceinline__ int g_start_at(const DGraph& g, int k) { return LIN ? 0 : g.start_list[k]; }
template <bool LIN>
__device__ __forceinline__ int g_accept_at(const DGraph& g, int k) { return LIN ? g.M : g.accept_list[k]; }

// ---- matcher: calls f(r1, r2) for every pair of arcs (r1 of g1, r2 of g2) leaving
// (IN: entering) the node pair with olabel1 == ilabel2, in the reference's order:
// "for q in query list: for s in the equal-label run of the search list"
// (compose.cpp:211-374; roles per matcher as in functions.cpp:225-251).
template <bool IN, int MATCH, bool LQ, bool LS, bool SG1, class GQ, class GS, class F>
__device__ __forceinline__ void enum_matches_role(const GQ& gq, const GS& gs, const Adj& q, const Adj& s, F&& f) {
  // SG1: the search side is g1 (query = g2); all roles resolved at compile time
  constexpr bool sorted = MATCH != MATCH_UNSORTED;
  auto one_query = [&](const Rec& qr) {
    const int ql = SG1 ? qr.il : qr.ol;
    if (!IN && ql == EPS) return;  // direct eps:eps matches are skipped (compose.cpp:425-428)
    if (LS) {
      if (ql >= 0 && ql < gs.C && s.n > 0) {
        const Rec sr = adj_rec<IN, true>(gs, s, ql);
        if (SG1) f(sr, qr); else f(qr, sr);
      }
    } else if (sorted) {
      int lo = 0, hi = s.n;
      while (lo < hi) {  // std::lower_bound
        const int mid = (lo + hi) >> 1;
        const Rec m = adj_rec<IN, false>(gs, s, mid);
        if ((SG1 ? m.ol : m.il) < ql) lo = mid + 1; else hi = mid;
      }
      for (int k = lo; k < s.n; ++k) {
        const Rec sr = adj_rec<IN, false>(gs, s, k);
        if ((SG1 ? sr.ol : sr.il) != ql) break;
        if (SG1) f(sr, qr); else f(qr, sr);
      }
    } else {
      for (int k = 0; k < s.n; ++k) {
        const Rec sr = adj_rec<IN, false>(gs, s, k);
        if ((SG1 ? sr.ol : sr.il) == ql) {
          if (SG1) f(sr, qr); else f(qr, sr);
        }
      }
    }
  };
  // the first four query records are fetched together (independent loads)
  Rec q4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < q.n) q4[k] = adj_rec<IN, LQ>(gq, q, k);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < q.n) one_query(q4[k]);
  for (int k = 4; k < q.n; ++k) one_query(adj_rec<IN, LQ>(gq, q, k));
}

template <bool IN, int MATCH, bool L1, bool L2, class G1, class G2, class F>
__device__ __forceinline__ void enum_matches(const G1& g1, const G2& g2, int n1, int n2, F&& f) {
  const Adj l1 = IN ? in_adj<L1>(g1, n1) : out_adj<L1>(g1, n1);
  const Adj l2 = IN ? in_adj<L2>(g2, n2) : out_adj<L2>(g2, n2);
  if (MATCH == MATCH_SINGLY_G1) {
    enum_matches_role<IN, MATCH, L2, L1, true>(g2, g1, l2, l1, f);
  } else if (MATCH == MATCH_DOUBLY) {
    if (l1.n > l2.n)  // compose.cpp:319
      enum_matches_role<IN, MATCH, L2, L1, true>(g2, g1, l2, l1, f);
    else
      enum_matches_role<IN, MATCH, L1, L2, false>(g1, g2, l1, l2, f);
  } else {
    enum_matches_role<IN, MATCH, L1, L2, false>(g1, g2, l1, l2, f);
  }
}

// epsilon arcs of one side's list: g1 arcs with olabel eps / g2 arcs with ilabel eps
template <bool IN, bool LIN, class G, class F>
__device__ __forceinline__ void enum_eps(const G& g, const Adj& l, bool second, F&& f) {
  if (LIN) return;
  const bool sorted = second ? (g.flags & 1) : (g.flags & 2);
  for (int k = 0; k < l.n; ++k) {
    const Rec r = adj_rec<IN, false>(g, l, k);
    if ((second ? r.il : r.ol) != EPS) {
      if (sorted) break;  // eps sorts first (compose.cpp:36-41, 169-176)
      continue;
    }
    f(r);
  }
}
template <bool LIN, class G>
__device__ __forceinline__ bool has_eps(const G& g, const Adj& l, bool second) {
  if (LIN || l.n == 0) return false;
  const bool sorted = second ? (g.flags & 1) : (g.flags & 2);
  if (sorted) {  // eps (-1) sorts first: one record decides
    const Rec r = adj_rec<false, false>(g, l, 0);
    return (second ? r.il : r.ol) == EPS;
  }
  bool r = false;
  enum_eps<false, false>(g, l, second, [&](const Rec&) { r = true; });
  return r;
}

// ---- workgroup exclusive scan (wave64 shuffles + one LDS hop) -------------------
// Workgroup barrier ordering LDS traffic only (see shortest.hip): __syncthreads()
// also drains vmcnt, i.e. waits for every global store issued so far -- a full
// HBM round trip per barrier in a kernel that stores at every BFS level.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wg_barrier(bool lds_only) {
  if (lds_only) lds_barrier(); else __syncthreads();
}

// wave64 inclusive prefix sum by DPP: row_shr 1/2/4/8 scan each 16-lane row (lanes without a
// source add 0), row_bcast:15 carries row 0 -> 1 and 2 -> 3, row_bcast:31 carries lanes
// 0..31 -> rows 2, 3.  Six VALU ops instead of six dependent LDS permutes (__shfl_up).
__device__ __forceinline__ int wave_incl_scan_dpp(int x) {
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

template <int BLK = kBlock>
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*>= 8 ints*/, int& total, bool lds_only = false) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wave_incl_scan_dpp(v);
  wg_barrier(lds_only);  // protect sh from a previous use
  if (lane == 63) sh[wave] = x;
  wg_barrier(lds_only);
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < BLK / 64; ++w) {
    const int s = sh[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + x - v;
}

// ================================================================================
// the composition kernel
// ================================================================================
// LDS working set of the fast paths (ints): a per-chunk claim hash (keys / first
// arc rank / assigned id), the next frontier's pair ids, and per-level in-degree
// counters + cursors for the fused in-arc CSR.
// (sizes scale with the workgroup size BLK of the instantiation, see compose_kernel:
//  HC = 4*BLK claim-hash slots per chunk -- at most 3/4 of them arcs; FC = BLK frontier
//  pairs kept in LDS per level; WC = 2*BLK new nodes per level whose in-rows are built
//  in LDS; BQ = HC backward-BFS frontier pairs kept in LDS per level)
constexpr int KC = 4;      // candidates cached per lane (registers)
// When the pair table is small (2 * N1*N2 bits fit the dynamic LDS request) the
// co-reachability bitmap and a "discovered" bitmap live in LDS for the whole
// kernel: phase B then runs without any HBM traffic on its critical path and
// phase F consults HBM `state` only for pairs discovered in an earlier level.
constexpr int kMaxBitmapBytes = 2 * 26624;  // both bitmaps; keeps 2 workgroups per CU

struct Cand {
  int idx[KC], i[KC], j[KC], il[KC], ol[KC];
  int n;
  __device__ __forceinline__ void push(int id, int ai, int aj, int l_in = 0, int l_out = 0) {
    // select chain instead of dynamic indexing keeps everything in registers
#pragma unroll
    for (int m = 0; m < KC; ++m)
      if (n == m) { idx[m] = id; i[m] = ai; j[m] = aj; il[m] = l_in; ol[m] = l_out; }
    ++n;
  }
};

// FAST: the compact variant -- LDS bitmaps required, every cold path (nodes with
// more than KC candidates, chunks too large for the claim hash) compiled OUT; on
// meeting one it sets ComposeOut::overflow = 2 and leaves, and the host re-runs
// that pair with the general variant.  Keeping the cold code out of the hot
// kernel matters: the BFS inner loop is instruction-fetch sensitive.
template <bool C>
struct SelView {
  __device__ static __forceinline__ const LGraph& get(const LGraph& l, const DGraph&) { return l; }
};
template <>
struct SelView<false> {
  __device__ static __forceinline__ const DGraph& get(const LGraph&, const DGraph& d) { return d; }
};

// C1: g1 (explicit, small) has its adjacency of the current phase cached in LDS
// BLK: workgroup size.  256 by default; the FAST variant also exists at 512 lanes with
// doubled LDS working set, so that partners of up to 512 nodes keep every BFS level in
// ONE chunk (which the fused in-row build, the skipped arrays and the stationary-level
// replication all require).
template <int MATCH, bool L1, bool L2, bool FAST, bool C1, int BLK>
__global__ __launch_bounds__(BLK) void compose_kernel(const ComposeArgs* __restrict__ args) {
  constexpr int kBlock = BLK;
  constexpr int HC = 4 * BLK;
  constexpr int HC_LOG2 = BLK == 256 ? 10 : 11;
  constexpr int FC = BLK;
  constexpr int WC = 2 * BLK;
  constexpr int BQ = HC;
  static_assert(BLK == 256 || BLK == 512, "claim-hash width is tied to the block size");
  const ComposeArgs a = args[blockIdx.x];
  const int tid = threadIdx.x;
  const int N1 = a.g1.N, N2 = a.g2.N;
  __shared__ int sh_scan[8];
  __shared__ int sh_tail;
  __shared__ int sh_flag[4];
  __shared__ int hkeys[HC];
  __shared__ int hvals[HC];
  __shared__ int hids[HC];
  __shared__ int front[2][FC];   // also the backward queue (2 x BQ/2 ... see below)
  __shared__ int incnt[WC];
  __shared__ int incur[WC];
  __shared__ int sh_lst;     // this chunk created start / accept nodes
  __shared__ int sh_rep[2];  // stationarity mismatch flags: [0] phase B sets, [1] phase F frontiers
  extern __shared__ __attribute__((aligned(16))) unsigned dyn_bits[];
  const int nwords = (N1 * N2 + 31) >> 5;
  const bool lds_state = FAST ? true : (a.lds_state != 0);
  // Two layouts of the co-reachability / discovered bitmaps in LDS:
  //  classic : 2 x (N1*N2) bits, indexed by pair id;
  //  chain   : (a.chain_bits > 0; chain products with an epsilon-free partner only) a
  //            WINDOW of the last `Wmax` time slices of No bits each (slice s = time
  //            TM - s), one slice for the stationary set every earlier time shares, and
  //            two per-level "discovered" slices -- independent of the chain length, so
  //            T*U products far beyond the classic budget stay on the fast path.
  constexpr bool CHP = FAST && (L1 != L2);
  const bool CH = CHP && a.chain_bits > 0;
  const int No = L2 ? N1 : N2;               // nodes of the explicit partner
  const int NoW = (No + 31) >> 5;
  const int Wmax = a.chain_bits;
  const int bm_words = CH ? NoW * (Wmax + 3) : 2 * nwords;
  unsigned* reach_bits = dyn_bits;           // classic [nwords] | chain window [Wmax][NoW]
  unsigned* disc_bits = dyn_bits + nwords;   // classic
  unsigned* stat_bits = dyn_bits + NoW * (CH ? Wmax : 0);  // chain: stationary set (zero until known)
  unsigned* disc2 = stat_bits + NoW;         // chain: [2][NoW], by level parity
  // ---- optional LDS cache of g1: offsets, records (16 B aligned), node flags
  const int A1 = a.g1.A;
  int* c_off = reinterpret_cast<int*>(dyn_bits + ((bm_words + 3) & ~3));
  gtnx_i4* c_rec = reinterpret_cast<gtnx_i4*>(c_off + ((N1 + 1 + 3) & ~3));
  uint8_t* c_fl = reinterpret_cast<uint8_t*>(c_rec + A1);
  LGraph lg;
  lg.N = N1; lg.A = A1; lg.M = a.g1.M; lg.C = a.g1.C; lg.flags = a.g1.flags;
  lg.in_off = lg.out_off = (const GTNX_L int*)c_off;
  lg.in_rec = lg.out_rec = (const GTNX_L gtnx_i4*)c_rec;
  lg.nflags = (const GTNX_L uint8_t*)c_fl;
  const auto& g1v = SelView<C1>::get(lg, a.g1);
  auto cache_g1 = [&](bool in_lists) {
    if (!C1) return;
    const GTNX_G int* off = in_lists ? a.g1.in_off : a.g1.out_off;
    const GTNX_G gtnx_i4* rec = in_lists ? a.g1.in_rec : a.g1.out_rec;
    for (int x = tid; x <= N1; x += kBlock) c_off[x] = off[x];
    for (int x = tid; x < A1; x += kBlock) c_rec[x] = rec[x];
    for (int x = tid; x < N1; x += kBlock) c_fl[x] = a.g1.nflags[x];
    __syncthreads();
  };

  // ---- stationary-level replication (product with ONE implicit linear chain whose
  // partner has no epsilon labels): the pair (n, t) only ever steps to (n', t+1) and
  // the chain offers every label at every t, so both the co-reachable set B[t] and
  // the ordered forward frontier S[t] are images of a time-invariant map.  Once
  // B[t-1] == B[t] every earlier B equals it; once S[t+1] == S[t] (same order) under
  // a constant filter, every later level is the previous one shifted by one time
  // step -- ids + W, arc ids + Aw, chain arc + C -- and is emitted from registers
  // for all remaining stationary t at once instead of one BFS level at a time.
  const bool skip = FAST && a.skip != 0;  // derivable arrays are left out (see ComposeArgs::skip)
  // the stationary levels are written by compose_wide.hip's replication kernel, launched behind this one: this
  // workgroup only records which level to copy and leaves the hole (ComposeOut::wr_*)
  // (the record goes to memory at once: six more values alive across the level loop cost the FAST variant its
  //  second wave per SIMD -- 254 -> 256 registers, one workgroup per CU instead of two, 2.95 -> 4.75 ms at C3)
  const bool rep_grid = FAST && a.rep_grid != 0;
  bool grid_done = false;
  // pair -> node id table in HBM: only ever read back for pairs discovered in an earlier
  // chunk of the same level or across levels (non-layered arcs); with single-chunk levels
  // (skip) in the time-windowed layout neither can happen, so the writes are dropped too
  const bool no_state = skip && (FAST && (L1 != L2)) && a.chain_bits > 0;
  constexpr bool REP = FAST && (L1 != L2);
  const bool rep_ok = REP && ((L2 ? a.g1.flags : a.g2.flags) & GF_EPS_FREE);
  const int tshift = L2 ? N1 : 1;           // pair-id step per time step
  const int TM = L2 ? a.g2.M : a.g1.M;      // chain length
  const int CL = L2 ? a.g2.C : a.g1.C;      // chain arcs per time step
  int tB = -1;                              // B[t] is the stationary set for every t <= tB
  // chain layout helpers; the caller knows the time of the pair it asks about
  auto other_of = [&](int idx, int t) { return L2 ? idx - N1 * t : (idx - t) / N1; };
  auto ch_word = [&](int t, int n) -> unsigned* {
    const int sl = TM - t;
    return ((tB >= 0 && t <= tB) || sl >= Wmax || sl < 0) ? &stat_bits[n >> 5] : &reach_bits[sl * NoW + (n >> 5)];
  };
  if (tid == 0) {
    sh_tail = 0;
    sh_rep[0] = sh_rep[1] = 0;
    sh_lst = 0;
    sh_flag[0] = 1;  // layered
    sh_flag[1] = 0;  // overflow
    sh_flag[2] = 1;  // in-CSR built in-kernel is valid
    sh_flag[3] = 0;  // per-chunk: bit0 wide node, bit1 discovered-pair lookup
  }
  __syncthreads();
  const long long tk0 = wall_clock64();
  long long tk_rep = 0;
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

  // ------------------------------------------------------------------ phase B
  // compose.cpp:64-104 -- `state` was pre-filled with ST_UNREACH.  The frontier
  // queue of a level lives in LDS (hkeys/hvals double as the two queue buffers)
  // when it fits, else in HBM; a lane gathers its (<= KC) predecessor pairs
  // first, then issues their state loads together and CASes the unreached ones.
  {
    auto bq = [&](int b) -> int* { return b ? hvals : hkeys; };
    const int na1 = a.g1.n_accept, na2 = a.g2.n_accept;
    const int seeds = na1 * na2;
    if (lds_state) {
      for (int x = tid; x < bm_words; x += kBlock) dyn_bits[x] = 0u;
      __syncthreads();
    }
    cache_g1(true);
    for (int t = tid; t < seeds; t += kBlock) {
      const int f = g_accept_at<L1>(a.g1, t / na2), s = g_accept_at<L2>(a.g2, t % na2);
      const int idx = f + N1 * s;
      if (CH) {
        const int n = other_of(idx, TM);
        atomicOr(&reach_bits[n >> 5], 1u << (n & 31));  // slice 0 = time TM
      } else if (lds_state) atomicOr(&reach_bits[idx >> 5], 1u << (idx & 31));
      else st_state(a.state + idx, ST_REACH);
      if (t < BQ) bq(0)[t] = idx; else a.queue[t] = idx;
    }
    if (tid == 0) sh_tail = seeds;
    __syncthreads();
    int lo = 0, hi = seeds, cur = 0;
    int tau = TM;  // time of the frontier being expanded (rep_ok: one time per level)
    auto mark = [&](int idx, int nxt) {
      bool fresh;
      int chn = 0, chs = 0;
      if (CH) {
        chs = TM - (tau - 1);  // slice of the pairs being marked
        if (chs >= Wmax) {     // not stationary within the window: the general variant takes over
          sh_flag[1] = 2;
          return;
        }
        chn = other_of(idx, tau - 1);
        const unsigned bit = 1u << (chn & 31);
        fresh = !(atomicOr(&reach_bits[chs * NoW + (chn >> 5)], bit) & bit);
      } else {
        fresh = lds_state ? !(atomicOr(&reach_bits[idx >> 5], 1u << (idx & 31)) & (1u << (idx & 31)))
                          : atomicCAS(a.state + idx, ST_UNREACH, ST_REACH) == ST_UNREACH;
      }
      if (fresh) {
        const int pos = atomicAdd(&sh_tail, 1);
        if (pos - hi < BQ) bq(nxt)[pos - hi] = idx; else a.queue[pos] = idx;
        if (REP && rep_ok) {  // same partner node one time step later must be in the set
          if (CH) {
            if (!((reach_bits[(chs - 1) * NoW + (chn >> 5)] >> (chn & 31)) & 1u)) sh_rep[0] = 1;
          } else {
            const int q = idx + tshift;
            if (!((reach_bits[q >> 5] >> (q & 31)) & 1u)) sh_rep[0] = 1;
          }
        }
      }
    };
    while (lo < hi) {
      for (int f = lo + tid; f < hi; f += kBlock) {
        const int idx = (f - lo) < BQ ? bq(cur)[f - lo] : a.queue[f];
        const int n1 = idx % N1, n2 = idx / N1;
        Cand c;
        c.n = 0;
        enum_matches<true, MATCH, L1, L2>(g1v, a.g2, n1, n2, [&](const Rec& r1, const Rec& r2) { c.push(r1.node + N1 * r2.node, 0, 0); });
        enum_eps<true, L1>(g1v, in_adj<L1>(g1v, n1), false, [&](const Rec& r) { c.push(r.node + N1 * n2, 0, 0); });
        enum_eps<true, L2>(a.g2, in_adj<L2>(a.g2, n2), true, [&](const Rec& r) { c.push(n1 + N1 * r.node, 0, 0); });
        if (c.n <= KC) {
          int st[KC];
#pragma unroll
          for (int m = 0; m < KC; ++m) {
            if (m >= c.n) st[m] = 0;
            else if (CH) {
              const int n = other_of(c.idx[m], tau - 1);
              st[m] = ((*ch_word(tau - 1, n) >> (n & 31)) & 1u) ? 0 : ST_UNREACH;
            } else if (lds_state) st[m] = ((reach_bits[c.idx[m] >> 5] >> (c.idx[m] & 31)) & 1u) ? 0 : ST_UNREACH;
            else st[m] = ld_state(a.state + c.idx[m]);
          }
#pragma unroll
          for (int m = 0; m < KC; ++m)
            if (m < c.n && st[m] == ST_UNREACH) mark(c.idx[m], cur ^ 1);
        } else if (FAST) {
          sh_flag[1] = 2;  // hand this pair to the general kernel
        } else {
          auto slow = [&](int id) {
            const bool un = lds_state ? !((reach_bits[id >> 5] >> (id & 31)) & 1u)
                                      : ld_state(a.state + id) == ST_UNREACH;
            if (un) mark(id, cur ^ 1);
          };
          enum_matches<true, MATCH, L1, L2>(g1v, a.g2, n1, n2, [&](const Rec& r1, const Rec& r2) { slow(r1.node + N1 * r2.node); });
          enum_eps<true, L1>(g1v, in_adj<L1>(g1v, n1), false, [&](const Rec& r) { slow(r.node + N1 * n2); });
          enum_eps<true, L2>(a.g2, in_adj<L2>(a.g2, n2), true, [&](const Rec& r) { slow(n1 + N1 * r.node); });
        }
      }
      // LDS-only barrier when nothing this level communicated through HBM
      wg_barrier(lds_state);
      const int prev_w = hi - lo;
      const bool same = REP && rep_ok && sh_rep[0] == 0;
      lo = hi;
      hi = sh_tail;
      cur ^= 1;
      if (REP && rep_ok && tid == 0) sh_rep[0] = 0;
      wg_barrier(lds_state && (hi - lo) <= BQ);
      if (REP && rep_ok) {
        if (CH && same && hi - lo == prev_w && prev_w > 0 && tau >= 1) {
          // B[tau-1] == B[tau]: that set serves every earlier time; no fill in this layout
          tB = tau;
          for (int x = tid; x < NoW; x += kBlock) stat_bits[x] = reach_bits[(TM - tau) * NoW + x];
          lds_barrier();
          break;
        }
        if (!CH && same && hi - lo == prev_w && prev_w > 0 && tau >= 2) {
          // B[tau-1] == B[tau]: fill every earlier time with the same partner set
          tB = tau;
          const int total_bits = N1 * N2;
          for (int wd = tid; wd < nwords; wd += kBlock) {
            int p = wd << 5;
            int tm = L2 ? p / N1 : p % N1, ot = L2 ? p % N1 : p / N1;
            unsigned v = 0;
            for (int b = 0; b < 32 && p < total_bits; ++b, ++p) {
              if (tm <= tau - 2) {
                const int q = L2 ? ot + N1 * tau : tau + N1 * ot;
                v |= ((reach_bits[q >> 5] >> (q & 31)) & 1u) << b;
              }
              if (L2) { if (++ot == N1) { ot = 0; ++tm; } }
              else    { if (++tm == N1) { tm = 0; ++ot; } }
            }
            // bits of times >= tau-1 in this word were set by the BFS; nobody writes now
            if (v) reach_bits[wd] |= v;
          }
          lds_barrier();
          break;
        }
        --tau;
      }
    }
  }

  if (FAST && sh_flag[1]) {
    if (tid == 0) { ComposeOut o{}; o.overflow = 2; *a.out = o; a.counts[0] = a.counts[1] = 0; }
    return;
  }
  if (lds_state && !CH) {
    // publish the co-reachability table for the general (HBM) code paths of phase F
    for (int x = tid; x < N1 * N2; x += kBlock)
      a.state[x] = ((reach_bits[x >> 5] >> (x & 31)) & 1u) ? ST_REACH : ST_UNREACH;
    __syncthreads();
  }
  // ------------------------------------------------------------------ phase F
  const long long tk1 = wall_clock64();
  cache_g1(false);
  int nn = 0, na = 0;
  // ordered start / accept lists are produced here as nodes are numbered (ids rise
  // with discovery, so appending per level in id order keeps them sorted); only the
  // general chunk path leaves them to the transpose kernels
  int ns_tot = 0, na_tot = 0;
  bool lists_ok = true;
  {
    // start pairs in (s1 outer, s2 inner) order (compose.cpp:392-401)
    const int ns1 = a.g1.n_start, ns2 = a.g2.n_start;
    const int seeds = ns1 * ns2;
    for (int t0 = 0; t0 < seeds; t0 += kBlock) {
      const int t = t0 + tid;
      int idx = 0, ok = 0, s1 = 0, s2 = 0;
      if (t < seeds) {
        s1 = g_start_at<L1>(a.g1, t / ns2);
        s2 = g_start_at<L2>(a.g2, t % ns2);
        idx = s1 + N1 * s2;
        if (CH) {
          const int n = other_of(idx, 0);
          ok = (*ch_word(0, n) >> (n & 31)) & 1u;
        } else {
          ok = ld_state(a.state + idx) == ST_REACH;
        }
      }
      int tot, tota;
      const int off = block_excl_scan<BLK>(ok, sh_scan, tot);
      const int acc0 = ok && g_accept<L1>(g1v, s1) && g_accept<L2>(a.g2, s2);
      const int offa = block_excl_scan<BLK>(acc0, sh_scan, tota);
      if (ok) {
        const int id = nn + off;
        if (id < a.Ncap) {
          a.pair_of[id] = idx;
          a.nflags[id] = uint8_t(NF_START | (acc0 ? NF_ACCEPT : 0));
          a.start_list[id] = id;
          if (acc0) a.accept_list[na_tot + offa] = id;
          a.in_off[id] = 0;  // level 0 has no in-arcs in a layered product
          if (id < FC) front[0][id] = idx;
          if (lds_state && !CH) atomicOr(&disc_bits[idx >> 5], 1u << (idx & 31));
          if (!no_state) st_state(a.state + idx, id);
        } else {
          sh_flag[1] = 1;
        }
      }
      nn += tot;
      na_tot += tota;
    }
    ns_tot = nn;
    for (int x = tid; x < WC; x += kBlock) incnt[x] = 0;
    __syncthreads();
  }

  int lo = 0, hi = nn, L = 0, fcur = 0;
  int max_width = 0, max_level_arcs = 0, rep_levels = 0;
  while (lo < hi && !sh_flag[1]) {
    if (tid == 0) {
      a.level_off[L] = lo;
      if (REP) sh_rep[1] = 0;
    }
    if (CH)  // level L flags pairs of time L+1 in slice (L+1)&1; the other one is recycled for level L+1
      for (int x = tid; x < NoW; x += kBlock) disc2[(L & 1) * NoW + x] = 0u;
    const int na_level = na;
    max_width = max(max_width, hi - lo);
    const bool front_in_lds = (hi - lo) <= FC;
    const bool single_chunk = (hi - lo) <= kBlock;
    // registers of this lane's emitted arcs (fast path), kept for the fused in-CSR
    int my_dst[KC], my_ai[KC];
    float my_w[KC];
    // ... and, for the stationary-level replication, the rest of the arc template
    int my_i[KC], my_j[KC], my_il[KC], my_ol[KC], my_pos[KC], my_own[KC];
    int my_out = 0, newn_level = 0;
    bool fast_level = true;
    for (int c0 = lo; c0 < hi; c0 += kBlock) {
      const int node = c0 + tid;
      const bool live = node < hi;
      int n1 = 0, n2 = 0;
      bool eps1_ok = false, eps2_ok = false;
      Adj o1{}, o2{};
      Cand c;
      c.n = 0;
      // clear the claim table (ordered before its use by the scan's barriers).  In the
      // time-windowed layout every candidate of this level lives at time L+1, so the partner
      // node (< No <= HC) indexes the table directly: no keys, no probing
      if (CH) {
        for (int x = tid; x < No; x += kBlock) hvals[x] = INT_MAX;
      } else {
        for (int x = tid; x < HC; x += kBlock) {
          hkeys[x] = -1;
          hvals[x] = INT_MAX;
        }
      }
      if (live) {
        const int pr = front_in_lds ? front[fcur][node - lo] : a.pair_of[node];
        n1 = pr % N1;
        n2 = pr / N1;
        o1 = out_adj<L1>(g1v, n1);
        o2 = out_adj<L2>(a.g2, n2);
        // epsilon_matched <=> some (i, j) with olabel1(i) == ilabel2(j) == eps
        const bool em = has_eps<L1>(g1v, o1, false) && has_eps<L2>(a.g2, o2, true);
        const bool acc1 = g_accept<L1>(g1v, n1), acc2 = g_accept<L2>(a.g2, n2);
        eps1_ok = !em || acc2 || !acc1;  // compose.cpp:461
        eps2_ok = !em || acc1;           // compose.cpp:476
        enum_matches<false, MATCH, L1, L2>(g1v, a.g2, n1, n2, [&](const Rec& r1, const Rec& r2) {
          c.push(r1.node + N1 * r2.node, r1.arc, r2.arc, r1.il, r2.ol);
        });
        if (eps1_ok) enum_eps<false, L1>(g1v, o1, false, [&](const Rec& r) { c.push(r.node + N1 * n2, r.arc, -1, r.il, EPS); });
        if (eps2_ok) enum_eps<false, L2>(a.g2, o2, true, [&](const Rec& r) { c.push(n1 + N1 * r.node, -1, r.arc, EPS, r.ol); });
      }
      // state of the cached candidates.  With the bitmaps in LDS no HBM access is
      // needed unless a candidate pair was already discovered (non-layered arcs or
      // an earlier chunk of this level): then -- and only then -- the workgroup
      // drains its stores (full barrier) and reads the id from the HBM table.
      int st[KC];
      bool hit = false;
#pragma unroll
      for (int m = 0; m < KC; ++m) {
        if (m >= c.n) {
          st[m] = ST_UNREACH;
        } else if (CH) {
          const int n = other_of(c.idx[m], L + 1);
          const unsigned bit = 1u << (n & 31);
          st[m] = !(*ch_word(L + 1, n) & bit) ? ST_UNREACH
                                               : ((disc2[((L + 1) & 1) * NoW + (n >> 5)] & bit) ? 0 : ST_REACH);
          hit = hit || st[m] == 0;
        } else if (lds_state) {
          const unsigned bit = 1u << (c.idx[m] & 31);
          const int wd = c.idx[m] >> 5;
          st[m] = !(reach_bits[wd] & bit) ? ST_UNREACH : ((disc_bits[wd] & bit) ? 0 : ST_REACH);
          hit = hit || st[m] == 0;
        } else {
          st[m] = ld_state(a.state + c.idx[m]);
        }
      }
      // Arc weights are fetched NOW, ahead of the scan and before any store of this
      // level: vmcnt retires in order, so a load issued after a store cannot be
      // waited on without also waiting for that store's HBM round trip.
      float wpre[KC];
#pragma unroll
      for (int m = 0; m < KC; ++m) {
        wpre[m] = 0.0f;
        if (m < c.n) wpre[m] = (c.i[m] >= 0 ? a.g1.w[c.i[m]] : 0.0f) + (c.j[m] >= 0 ? a.g2.w[c.j[m]] : 0.0f);
      }
      if (c.n > KC) atomicOr(&sh_flag[3], 1);
      if (hit) atomicOr(&sh_flag[3], 2);
      wg_barrier(lds_state);
      const int chunk_flags = sh_flag[3];
      if (chunk_flags & 2) {
        __syncthreads();  // every earlier st_state() of this workgroup has landed
#pragma unroll
        for (int m = 0; m < KC; ++m)
          if (m < c.n && st[m] == 0) st[m] = ld_state(a.state + c.idx[m]);
      }
      int cnt = 0;
      if (c.n <= KC) {
#pragma unroll
        for (int m = 0; m < KC; ++m) cnt += st[m] != ST_UNREACH;
      } else if (!FAST) {
        // wide node (more than KC candidates): count by re-enumeration (HBM table)
        enum_matches<false, MATCH, L1, L2>(g1v, a.g2, n1, n2, [&](const Rec& r1, const Rec& r2) {
          cnt += ld_state(a.state + r1.node + N1 * r2.node) != ST_UNREACH;
        });
        if (eps1_ok)
          enum_eps<false, L1>(g1v, o1, false, [&](const Rec& r) { cnt += ld_state(a.state + r.node + N1 * n2) != ST_UNREACH; });
        if (eps2_ok)
          enum_eps<false, L2>(a.g2, o2, true, [&](const Rec& r) { cnt += ld_state(a.state + n1 + N1 * r.node) != ST_UNREACH; });
      }
      int total;
      const int off = block_excl_scan<BLK>(cnt, sh_scan, total, lds_state);
      const bool fast = !(chunk_flags & 1) && total <= (HC * 3) / 4;
      if (FAST && !fast) {
        if (tid == 0) sh_flag[1] = 2;
        lds_barrier();
        break;
      }
      if (!fast) __syncthreads();  // the general path below goes through HBM
      if (na + total > a.Acap) {
        if (tid == 0) sh_flag[1] = 1;
        __syncthreads();
        break;
      }
      if (live) a.out_off[node] = na + off;
      if (REP) my_out = na + off;
      int newn = 0;
      if (fast) {
        // ---------------- fast chunk: claims / ids through the LDS hash
        int slot[KC], rr[KC];
        int k = 0;
#pragma unroll
        for (int m = 0; m < KC; ++m) {
          slot[m] = -1;
          rr[m] = -1;
          if (m < c.n && st[m] != ST_UNREACH) {
            const int r = off + k++;
            const int ai = na + r;
            const int i = c.i[m], j = c.j[m];
            const int il = c.il[m], ol = c.ol[m];
            const float w = wpre[m];
            if (!skip) {
              a.src[ai] = node;
              a.il[ai] = il;
              a.ol[ai] = ol;
            }
            a.w[ai] = w;
            a.gi1[ai] = i;
            a.gi2[ai] = j;
            rr[m] = r;
            my_w[m] = w;
            if (REP) { my_i[m] = i; my_j[m] = j; my_il[m] = il; my_ol[m] = ol; }
            if (st[m] < 0) {  // co-reachable, not discovered yet: claim by smallest arc rank
              unsigned h;
              if (CH) {
                h = unsigned(other_of(c.idx[m], L + 1));
              } else {
                h = (unsigned(c.idx[m]) * 2654435761u) >> (32 - HC_LOG2);
                while (true) {
                  const int old = atomicCAS(&hkeys[h], -1, c.idx[m]);
                  if (old == -1 || old == c.idx[m]) break;
                  h = (h + 1) & (HC - 1);
                }
              }
              atomicMin(&hvals[h], r);
              slot[m] = int(h);
            }
          }
        }
        wg_barrier(lds_state);
        int nown = 0;
        bool own[KC];
#pragma unroll
        for (int m = 0; m < KC; ++m) {
          own[m] = slot[m] >= 0 && hvals[slot[m]] == rr[m];
          nown += own[m];
        }
        int t2;
        int rank = block_excl_scan<BLK>(nown, sh_scan, t2, lds_state);
        newn = t2;
        int own_id[KC], own_fl[KC], n_s = 0, n_a = 0;
#pragma unroll
        for (int m = 0; m < KC; ++m) {
          if (REP) my_own[m] = -1;
          own_id[m] = -1;
          own_fl[m] = 0;
          if (own[m]) {
            const int id = nn + rank++;
            if (REP) my_own[m] = id;
            if (id < a.Ncap) {
              const int idx = c.idx[m];
              const int d1 = idx % N1, d2 = idx / N1;
              const int fl = ((g_start<L1>(g1v, d1) && g_start<L2>(a.g2, d2)) ? NF_START : 0) |
                             ((g_accept<L1>(g1v, d1) && g_accept<L2>(a.g2, d2)) ? NF_ACCEPT : 0);
              own_id[m] = id;
              own_fl[m] = fl;
              n_s += (fl & NF_START) != 0;
              n_a += (fl & NF_ACCEPT) != 0;
              if (fl) sh_lst = 1;
              hids[slot[m]] = id;
              a.pair_of[id] = idx;
              a.nflags[id] = uint8_t(fl);
              if (id - hi < FC) front[fcur ^ 1][id - hi] = idx;
              if (CH) {
                const int n = other_of(idx, L + 1);
                atomicOr(&disc2[((L + 1) & 1) * NoW + (n >> 5)], 1u << (n & 31));
              } else if (lds_state) atomicOr(&disc_bits[idx >> 5], 1u << (idx & 31));
              if (!no_state) st_state(a.state + idx, id);
            } else {
              hids[slot[m]] = 0;
              sh_flag[1] = 1;
            }
          }
        }
        wg_barrier(lds_state);
        if (sh_lst) {  // workgroup-uniform: some lane numbered a start / accept node
          int ts, ta;
          int os = block_excl_scan<BLK>(n_s, sh_scan, ts, lds_state);
          int oa = block_excl_scan<BLK>(n_a, sh_scan, ta, lds_state);
#pragma unroll
          for (int m = 0; m < KC; ++m) {
            if (own_fl[m] & NF_START) a.start_list[ns_tot + os++] = own_id[m];
            if (own_fl[m] & NF_ACCEPT) a.accept_list[na_tot + oa++] = own_id[m];
          }
          ns_tot += ts;
          na_tot += ta;
          if (tid == 0) sh_lst = 0;  // every lane read it before the scans' barriers
        }
        if (REP && rep_ok) {
          // is the new frontier the old one moved one time step (same order)?
          if (newn != hi - lo) { if (tid == 0) sh_rep[1] = 1; }
          else if (tid < newn && tid < FC && front[fcur ^ 1][tid] != front[fcur][tid] + tshift) sh_rep[1] = 1;
          newn_level = newn;
        }
        int lay = 1, csr_ok = 1;
#pragma unroll
        for (int m = 0; m < KC; ++m) {
          my_dst[m] = -1;
          my_ai[m] = -1;
          if (rr[m] >= 0) {
            const int id = slot[m] >= 0 ? hids[slot[m]] : st[m];
            a.dst[na + rr[m]] = id;
            if (id < hi) lay = 0;
            else if (id - hi < WC) atomicAdd(&incnt[id - hi], 1);
            else csr_ok = 0;
            my_dst[m] = id;
            my_ai[m] = na + rr[m];
          }
        }
        if (!lay) sh_flag[0] = 0;
        if (!lay || !csr_ok) sh_flag[2] = 0;
      } else if (!FAST) {
        // ---------------- general chunk: claims through the global state table
        fast_level = false;
        lists_ok = false;
        if (live) {
          int r = off;
          auto emit = [&](int idx, int il, int ol, float w, int i, int j) {
            const int cur = ld_state(a.state + idx);
            if (cur == ST_UNREACH) return;
            const int ai = na + r;
            a.src[ai] = node;
            a.dst[ai] = idx;  // patched to the node id below
            a.il[ai] = il;
            a.ol[ai] = ol;
            a.w[ai] = w;
            a.gi1[ai] = i;
            a.gi2[ai] = j;
            if (cur < 0) atomicMax(a.state + idx, claim_of(r));
            ++r;
          };
          enum_matches<false, MATCH, L1, L2>(g1v, a.g2, n1, n2, [&](const Rec& r1, const Rec& r2) {
            emit(r1.node + N1 * r2.node, r1.il, r2.ol, a.g1.w[r1.arc] + a.g2.w[r2.arc], r1.arc, r2.arc);
          });
          if (eps1_ok)
            enum_eps<false, L1>(g1v, o1, false, [&](const Rec& r) { emit(r.node + N1 * n2, r.il, EPS, a.g1.w[r.arc], r.arc, -1); });
          if (eps2_ok)
            enum_eps<false, L2>(a.g2, o2, true, [&](const Rec& r) { emit(n1 + N1 * r.node, EPS, r.ol, a.g2.w[r.arc], -1, r.arc); });
        }
        __syncthreads();
        for (int r0 = 0; r0 < total; r0 += kBlock) {
          const int r = r0 + tid;
          int own = 0, idx = 0;
          if (r < total) {
            idx = a.dst[na + r];
            own = ld_state(a.state + idx) == claim_of(r);
          }
          int t2;
          const int rank = block_excl_scan<BLK>(own, sh_scan, t2);
          int id = -1;
          if (own) {
            id = nn + newn + rank;
            if (id < a.Ncap) {
              const int d1 = idx % N1, d2 = idx / N1;
              a.pair_of[id] = idx;
              a.nflags[id] = uint8_t(((g_start<L1>(g1v, d1) && g_start<L2>(a.g2, d2)) ? NF_START : 0) |
                                     ((g_accept<L1>(g1v, d1) && g_accept<L2>(a.g2, d2)) ? NF_ACCEPT : 0));
              if (id - hi < FC) front[fcur ^ 1][id - hi] = idx;
              if (lds_state) atomicOr(&disc_bits[idx >> 5], 1u << (idx & 31));
            } else {
              sh_flag[1] = 1;
              id = -1;
            }
          }
          if (r < total) a.in_list[na + r] = id;  // scratch: in_list is built later
          newn += t2;
        }
        __syncthreads();  // every ownership test has read its claim
        for (int r = tid; r < total; r += kBlock) {
          const int id = a.in_list[na + r];
          if (id >= 0) st_state(a.state + a.dst[na + r], id);
        }
        __syncthreads();
        int lay = 1, csr_ok = 1;
        for (int r = tid; r < total; r += kBlock) {
          const int id = ld_state(a.state + a.dst[na + r]);
          a.dst[na + r] = id;
          if (id < hi) lay = 0;
          else if (id - hi < WC) atomicAdd(&incnt[id - hi], 1);
          else csr_ok = 0;
        }
        if (!lay) sh_flag[0] = 0;
        if (!lay || !csr_ok) sh_flag[2] = 0;
      }
      na += total;
      nn += newn;
      if (tid == 0) sh_flag[3] = 0;
      // a later chunk of the SAME level may look up ids this chunk stored in HBM
      wg_barrier(lds_state && single_chunk && fast);
    }
    if (sh_flag[1]) break;
    // ---- fused in-arc CSR of the next level: rows of nodes [hi, nn) are exactly
    // the arcs [na_level, na) when the product is layered.  Counts were taken in
    // LDS above; one scan gives the row offsets, then arcs are placed (from
    // registers when the level was a single fast chunk, else re-read from HBM).
    {
      const int W = nn - hi;
      max_level_arcs = max(max_level_arcs, na - na_level);
      constexpr int PER = WC / kBlock;
      int my_inoff[PER];
      const bool csr_level = sh_flag[2] && W <= WC;
      bool from_regs = false;
      if (csr_level) {
        int loc[PER];
        int sum = 0;
#pragma unroll
        for (int x = 0; x < PER; ++x) {
          loc[x] = incnt[tid * PER + x];
          sum += loc[x];
        }
        int tot;
        int run = block_excl_scan<BLK>(sum, sh_scan, tot, lds_state);
#pragma unroll
        for (int x = 0; x < PER; ++x) {
          const int nidx = tid * PER + x;
          if (nidx < W) a.in_off[hi + nidx] = na_level + run;
          my_inoff[x] = na_level + run;
          incur[nidx] = na_level + run;
          run += loc[x];
          incnt[nidx] = 0;
        }
        from_regs = single_chunk && fast_level;
        if (skip && !from_regs) sh_flag[1] = 2;  // the re-read path needs src[]: general variant
        wg_barrier(lds_state && from_regs);  // the re-read path needs this level's stores
        if (from_regs) {
#pragma unroll
          for (int m = 0; m < KC; ++m) {
            if (my_ai[m] >= 0) {
              const int pos = atomicAdd(&incur[my_dst[m] - hi], 1);
              if (!skip || rep_grid) a.in_list[pos] = my_ai[m];  // (the replication kernel reads its template level's)
              a.in_src[pos] = lo + tid;
              a.in_w[pos] = my_w[m];
              if (REP) my_pos[m] = pos;
            }
          }
        } else {
          for (int k = na_level + tid; k < na; k += kBlock) {
            const int pos = atomicAdd(&incur[a.dst[k] - hi], 1);
            a.in_list[pos] = k;
            a.in_src[pos] = a.src[k];
            a.in_w[pos] = a.w[k];
          }
        }
      } else {
        if (tid == 0) sh_flag[2] = 0;
        for (int x = tid; x < WC; x += kBlock) incnt[x] = 0;
      }
      // ---- stationary-level replication (see the top of the kernel).  Level L
      // expanded the frontier of time L under the filter B[L+1]; if the frontier
      // came back unchanged and B is constant up to tB, levels L+1 .. L+K with
      // K = tB - L - 1 are this level shifted by k time steps.  Every condition
      // below is workgroup-uniform (shared flags read behind the scan barriers).
      if (REP && rep_ok) {
        const int Aw = na - na_level;
        const int K = min(tB - L - 1, TM - L - 2);  // new nodes stay before the chain's accept time
        if (csr_level && from_regs && sh_flag[0] && !sh_flag[1] && sh_rep[1] == 0 && W == hi - lo &&
            newn_level == W && W > 0 && K >= 2 && nn + (long long)K * W <= a.Ncap &&
            na + (long long)K * Aw <= a.Acap) {
          const long long tr0 = wall_clock64();
          if (rep_grid && !grid_done) {
            grid_done = true;
            if (tid == 0) {
              a.out->wr_L = L; a.out->wr_K = K; a.out->wr_lo = lo; a.out->wr_W = W; a.out->wr_na = na_level; a.out->wr_Aw = Aw;
            }
          } else {
          const GTNX_G float* cw = L2 ? a.g2.w : a.g1.w;  // chain weights, one row per time step
          float wfix[KC];
          int carc[KC];
#pragma unroll
          for (int m = 0; m < KC; ++m) {
            wfix[m] = 0.0f;
            carc[m] = 0;
            if (my_ai[m] >= 0) {
              wfix[m] = L2 ? a.g1.w[my_i[m]] : a.g2.w[my_j[m]];
              carc[m] = L2 ? my_j[m] : my_i[m];
            }
          }
          const bool live_src = tid < hi - lo;
          constexpr int U = 4;
          constexpr int TF = HC / 2;  // arcs per level the flat path can hold
          if (Aw <= TF) {
            // ---- flat path: the level's arc template is transposed through LDS (the
            // claim hash and the in-row cursors are idle here) from "lane = source
            // node" to "lane = arc slot", so that every store instruction of a wave
            // covers 64 CONSECUTIVE elements of its output array (the lane-per-node
            // layout strides by the node's out-degree and touches ~3x the lines).
            int* t_sd = hkeys;            // src rank | dst rank << 16
            int* t_il = hkeys + TF;
            int* t_ol = hvals;
            int* t_ca = hvals + TF;       // chain arc at the template's time step
            int* t_wf = hids;             // fixed-side weight (float bits)
            int* t_gf = hids + TF;        // fixed-side arc
            int* t_inl = incur;           // in-row slot -> arc rank
#pragma unroll
            for (int m = 0; m < KC; ++m) {
              if (my_ai[m] >= 0) {
                const int r = my_ai[m] - na_level;
                t_sd[r] = tid | ((my_dst[m] - hi) << 16);
                t_il[r] = my_il[m];
                t_ol[r] = my_ol[m];
                t_ca[r] = carc[m];
                t_wf[r] = __float_as_int(wfix[m]);
                t_gf[r] = L2 ? my_i[m] : my_j[m];
                t_inl[my_pos[m] - na_level] = r;
              }
            }
            lds_barrier();
            constexpr int PE = TF / kBlock;  // flat slots per lane
            int e_sd[PE], e_il[PE], e_ol[PE], e_ca[PE], e_gf[PE], p_r[PE], p_sd[PE], p_ca[PE];
            float e_wf[PE], p_wf[PE];
            bool e_on[PE];
#pragma unroll
            for (int x = 0; x < PE; ++x) {
              const int e = tid + x * kBlock;
              e_on[x] = e < Aw;
              const int ee = e_on[x] ? e : 0;
              e_sd[x] = t_sd[ee]; e_il[x] = t_il[ee]; e_ol[x] = t_ol[ee]; e_ca[x] = t_ca[ee];
              e_wf[x] = __int_as_float(t_wf[ee]); e_gf[x] = t_gf[ee];
              p_r[x] = t_inl[ee];
              p_sd[x] = t_sd[p_r[x]]; p_ca[x] = t_ca[p_r[x]]; p_wf[x] = __int_as_float(t_wf[p_r[x]]);
            }
            for (int k0 = 1; k0 <= K; k0 += U) {
              float we[U][PE], wp[U][PE];
#pragma unroll
              for (int u = 0; u < U; ++u)
#pragma unroll
                for (int x = 0; x < PE; ++x) {
                  const bool on = e_on[x] && k0 + u <= K;
                  we[u][x] = on ? cw[e_ca[x] + (k0 + u) * CL] : 0.0f;
                  wp[u][x] = on ? cw[p_ca[x] + (k0 + u) * CL] : 0.0f;
                }
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int k = k0 + u;
                if (k > K) break;
                const int dn = k * W, da = k * Aw, dc = k * CL;
#pragma unroll
                for (int x = 0; x < PE; ++x) {
                  if (e_on[x]) {
                    const int ai = na_level + da + tid + x * kBlock;
                    if (!skip) {
                      a.src[ai] = lo + (e_sd[x] & 0xffff) + dn;
                      a.il[ai] = e_il[x];
                      a.ol[ai] = e_ol[x];
                      a.in_list[ai] = na_level + da + p_r[x];
                    }
                    a.dst[ai] = hi + (e_sd[x] >> 16) + dn;
                    a.w[ai] = e_wf[x] + we[u][x];
                    a.gi1[ai] = L2 ? e_gf[x] : e_ca[x] + dc;
                    a.gi2[ai] = L2 ? e_ca[x] + dc : e_gf[x];
                    a.in_src[ai] = lo + (p_sd[x] & 0xffff) + dn;
                    a.in_w[ai] = p_wf[x] + wp[u][x];
                  }
                }
#pragma unroll
                for (int m = 0; m < KC; ++m)
                  if (my_own[m] >= 0) a.nflags[my_own[m] + dn] = 0;  // neither start (t > 0) nor accept (t < TM)
                if (live_src) a.out_off[lo + tid + dn] = my_out + da;
#pragma unroll
                for (int x = 0; x < PER; ++x)
                  if (tid * PER + x < W) a.in_off[hi + tid * PER + x + dn] = my_inoff[x] + da;
              }
            }
            lds_barrier();  // the template arrays go back to their owners
          } else
          for (int k0 = 1; k0 <= K; k0 += U) {
            float wk[U][KC];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
              for (int m = 0; m < KC; ++m)
                wk[u][m] = (my_ai[m] >= 0 && k0 + u <= K) ? cw[carc[m] + (k0 + u) * CL] : 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int k = k0 + u;
              if (k > K) break;
              const int dn = k * W, da = k * Aw, dc = k * CL;
#pragma unroll
              for (int m = 0; m < KC; ++m) {
                if (my_ai[m] >= 0) {
                  const int ai = my_ai[m] + da, pos = my_pos[m] + da;
                  const float w = wfix[m] + wk[u][m];
                  if (!skip) {
                    a.src[ai] = lo + tid + dn;
                    a.il[ai] = my_il[m];
                    a.ol[ai] = my_ol[m];
                    a.in_list[pos] = ai;
                  }
                  a.dst[ai] = my_dst[m] + dn;
                  a.w[ai] = w;
                  a.gi1[ai] = L2 ? my_i[m] : my_i[m] + dc;
                  a.gi2[ai] = L2 ? my_j[m] + dc : my_j[m];
                  a.in_src[pos] = lo + tid + dn;
                  a.in_w[pos] = w;
                }
                if (my_own[m] >= 0) a.nflags[my_own[m] + dn] = 0;  // neither start (t > 0) nor accept (t < TM)
              }
              if (live_src) a.out_off[lo + tid + dn] = my_out + da;
#pragma unroll
              for (int x = 0; x < PER; ++x)
                if (tid * PER + x < W) a.in_off[hi + tid * PER + x + dn] = my_inoff[x] + da;
            }
          }
          for (int k = 1 + tid; k <= K; k += kBlock) a.level_off[L + k] = lo + k * W;
          }  // (inline replication)
          // the last replicated level's new nodes are the next frontier
          if (tid < W) front[fcur ^ 1][tid] += K * tshift;
          nn += K * W;
          na += K * Aw;
          lo += K * W;
          hi += K * W;
          L += K;
          rep_levels += K;
          if (CH)  // the level parity jumped: both per-level slices start clean
            for (int x = tid; x < 2 * NoW; x += kBlock) disc2[x] = 0u;
          tk_rep += wall_clock64() - tr0;
        }
      }
    }
    lo = hi;
    hi = nn;
    fcur ^= 1;
    ++L;
    // next level reads its frontier from LDS when it fits, else pair_of in HBM
    wg_barrier(lds_state && (hi - lo) <= FC);
  }
  __syncthreads();
  // ordered start / accept lists when the in-kernel CSR is valid: done by the
  // (cheap, parallel) list kernel afterwards -- see tr_lists_kernel.
  if (tid == 0) {
    a.level_off[L] = nn;
    a.out_off[nn < a.Ncap + 1 ? nn : a.Ncap] = na;
    a.in_off[nn < a.Ncap + 1 ? nn : a.Ncap] = na;
    ComposeOut o{};
    o.N = nn;
    o.A = na;
    o.L = L;
    o.layered = sh_flag[0];
    o.overflow = sh_flag[1];
    o.max_width = max_width;
    o.max_level_arcs = max_level_arcs;
    o.rep_levels = rep_levels;
    o.skipped = skip ? 1 : 0;
    o.t_b = int(tk1 - tk0);
    o.t_f = int(wall_clock64() - tk1);
    o.t_rep = int(tk_rep);
    if (grid_done) {  // written by this lane when the hole was left
      o.wr_L = a.out->wr_L; o.wr_K = a.out->wr_K; o.wr_lo = a.out->wr_lo; o.wr_W = a.out->wr_W; o.wr_na = a.out->wr_na; o.wr_Aw = a.out->wr_Aw;
    }
    o.csr_built = sh_flag[2] && sh_flag[0] && lists_ok;
    if (lists_ok) {
      a.counts[0] = ns_tot;
      a.counts[1] = na_tot;
    }
    *a.out = o;
  }
}

// ================================================================================
// in-arc CSR (transpose) + ordered start/accept lists: three fully parallel passes
// ================================================================================
constexpr int kChunk = 2048;  // nodes per scan chunk

__global__ void tr_count_kernel(const ComposeArgs* __restrict__ args) {
  const ComposeArgs a = args[blockIdx.y];
  if (a.out->csr_built) return;  // rows were built inside compose_kernel
  const int A = a.out->A;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < A; k += gridDim.x * blockDim.x)
    atomicAdd(a.in_cursor + a.dst[k], 1);
}

// per chunk: sums of (in-degree, start flag, accept flag) -> in_off scratch tail
__global__ __launch_bounds__(kBlock) void tr_chunk_sum_kernel(const ComposeArgs* __restrict__ args, int3* __restrict__ sums,
                                                              int chunks_per_graph) {
  const ComposeArgs a = args[blockIdx.y];
  const int N = a.out->N;
  const int c = blockIdx.x;
  if (c * kChunk >= N && c > 0) return;
  int d = 0, s = 0, ac = 0;
  for (int n = c * kChunk + threadIdx.x; n < min(N, (c + 1) * kChunk); n += kBlock) {
    d += a.in_cursor[n];
    const uint8_t f = a.nflags[n];
    s += (f & NF_START) != 0;
    ac += (f & NF_ACCEPT) != 0;
  }
  __shared__ int sh[3][kBlock];
  sh[0][threadIdx.x] = d;
  sh[1][threadIdx.x] = s;
  sh[2][threadIdx.x] = ac;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[(size_t)blockIdx.y * chunks_per_graph + c] = make_int3(sh[0][0], sh[1][0], sh[2][0]);
}

// one lane per graph turns chunk sums into chunk offsets (chunks are few)
__global__ void tr_chunk_scan_kernel(const ComposeArgs* __restrict__ args, int3* __restrict__ sums, int n, int chunks_per_graph) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const ComposeArgs a = args[g];
  const int N = a.out->N;
  const int nc = (N + kChunk - 1) / kChunk;
  int3 run = make_int3(0, 0, 0);
  for (int c = 0; c < nc; ++c) {
    const int3 v = sums[(size_t)g * chunks_per_graph + c];
    sums[(size_t)g * chunks_per_graph + c] = run;
    run.x += v.x;
    run.y += v.y;
    run.z += v.z;
  }
  if (!a.out->csr_built) a.in_off[N] = run.x;
  a.counts[0] = run.y;
  a.counts[1] = run.z;
}

__global__ __launch_bounds__(kBlock) void tr_offsets_kernel(const ComposeArgs* __restrict__ args, const int3* __restrict__ sums,
                                                            int chunks_per_graph) {
  const ComposeArgs a = args[blockIdx.y];
  const int N = a.out->N;
  const int c = blockIdx.x;
  if (c * kChunk >= N) return;
  __shared__ int sh_scan[8];
  int3 run = sums[(size_t)blockIdx.y * chunks_per_graph + c];
  for (int n0 = c * kChunk; n0 < min(N, (c + 1) * kChunk); n0 += kBlock) {
    const int n = n0 + threadIdx.x;
    int d = 0, s = 0, ac = 0;
    if (n < N) {
      d = a.in_cursor[n];
      const uint8_t f = a.nflags[n];
      s = (f & NF_START) != 0;
      ac = (f & NF_ACCEPT) != 0;
    }
    int td, ts, ta;
    const int od = block_excl_scan(d, sh_scan, td);
    const int os = block_excl_scan(s, sh_scan, ts);
    const int oa = block_excl_scan(ac, sh_scan, ta);
    if (n < N) {
      if (!a.out->csr_built) {
        a.in_off[n] = run.x + od;
        a.in_cursor[n] = run.x + od;
      }
      if (s) a.start_list[run.y + os] = n;
      if (ac) a.accept_list[run.z + oa] = n;
    }
    run.x += td;
    run.y += ts;
    run.z += ta;
  }
}

__global__ void tr_scatter_kernel(const ComposeArgs* __restrict__ args) {
  const ComposeArgs a = args[blockIdx.y];
  if (a.out->csr_built) return;
  const int A = a.out->A;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < A; k += gridDim.x * blockDim.x) {
    const int pos = atomicAdd(a.in_cursor + a.dst[k], 1);
    a.in_list[pos] = k;
    a.in_src[pos] = a.src[k];
    a.in_w[pos] = a.w[k];
  }
}

// ================================================================================
// gradient scatter (compose.cpp:496-518)
// ================================================================================
// Each workgroup owns a contiguous chunk of composed arcs.  Composed arcs are
// emitted in BFS order, so the input arcs a chunk refers to are clustered (for
// CTC: one or two target-graph nodes' arcs and a handful of emission frames).
// When the chunk's index range fits, contributions are first summed in an LDS
// window (ds_add_f32) and flushed with ONE global atomic per touched input arc;
// otherwise it falls back to direct global atomics.  This removes the heavy
// same-address contention (hundreds of composed arcs per blank-label emission).
constexpr int kGradChunk = 4096;
constexpr int kGradWin = 4096;

__global__ __launch_bounds__(kBlock) void compose_grad_kernel(const ComposeGradArgs* __restrict__ args) {
  const ComposeGradArgs a = args[blockIdx.y];
  const int k0 = blockIdx.x * kGradChunk;
  if (k0 >= a.A) return;
  const int k1 = min(a.A, k0 + kGradChunk);
  const int tid = threadIdx.x;
  __shared__ float win1[kGradWin];
  __shared__ float win2[kGradWin];
  __shared__ int red[4][kBlock];
  int mn1 = INT_MAX, mx1 = -1, mn2 = INT_MAX, mx2 = -1;
  for (int k = k0 + tid; k < k1; k += kBlock) {
    const int i = a.gi1[k], j = a.gi2[k];
    if (i >= 0) { mn1 = min(mn1, i); mx1 = max(mx1, i); }
    if (j >= 0) { mn2 = min(mn2, j); mx2 = max(mx2, j); }
  }
  red[0][tid] = mn1; red[1][tid] = mx1; red[2][tid] = mn2; red[3][tid] = mx2;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] = min(red[0][tid], red[0][tid + o]);
      red[1][tid] = max(red[1][tid], red[1][tid + o]);
      red[2][tid] = min(red[2][tid], red[2][tid + o]);
      red[3][tid] = max(red[3][tid], red[3][tid + o]);
    }
    __syncthreads();
  }
  mn1 = red[0][0]; mx1 = red[1][0]; mn2 = red[2][0]; mx2 = red[3][0];
  const bool lds1 = a.grad1 && mx1 >= 0 && (mx1 - mn1) < kGradWin;
  const bool lds2 = a.grad2 && mx2 >= 0 && (mx2 - mn2) < kGradWin;
  if (lds1) for (int x = tid; x <= mx1 - mn1; x += kBlock) win1[x] = 0.0f;
  if (lds2) for (int x = tid; x <= mx2 - mn2; x += kBlock) win2[x] = 0.0f;
  __syncthreads();
  for (int k = k0 + tid; k < k1; k += kBlock) {
    const float d = a.delta[k];
    const int i = a.gi1[k], j = a.gi2[k];
    if (a.grad1 && i >= 0) {
      if (lds1) atomicAdd(&win1[i - mn1], d); else atomicAdd(a.grad1 + i, d);
    }
    if (a.grad2 && j >= 0) {
      if (lds2) atomicAdd(&win2[j - mn2], d); else atomicAdd(a.grad2 + j, d);
    }
  }
  __syncthreads();
  if (lds1)
    for (int x = tid; x <= mx1 - mn1; x += kBlock) {
      const float v = win1[x];
      if (v != 0.0f) atomicAdd(a.grad1 + mn1 + x, v);
    }
  if (lds2)
    for (int x = tid; x <= mx2 - mn2; x += kBlock) {
      const float v = win2[x];
      if (v != 0.0f) atomicAdd(a.grad2 + mn2 + x, v);
    }
}

// packed adjacency records for a device-built graph that is used as a compose input
__global__ void build_records_kernel(DGraph g, gtnx_i4* out_rec, gtnx_i4* in_rec) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < g.A; k += gridDim.x * blockDim.x) {
    const int ao = g.out_list ? g.out_list[k] : k;
    out_rec[k] = gtnx_i4{g.il[ao], g.ol[ao], g.dst[ao], ao};
    const int ai = g.in_list[k];
    in_rec[k] = gtnx_i4{g.il[ai], g.ol[ai], g.src[ai], ai};
  }
}

int grid_x(int n, int cap) {
  int g = (n + kBlock - 1) / kBlock;
  return g < 1 ? 1 : (g > cap ? cap : g);
}

} // namespace

int compose_max_bitmap_bytes() { return kMaxBitmapBytes; }
size_t compose_chain_bitmap_bytes(int No, int slices) { return 4 * size_t((No + 31) / 32) * size_t(slices + 3); }

namespace {
template <int MATCH, bool L1, bool L2, bool FAST, bool C1, int BLK>
void launch_compose_b(const ComposeArgs* d_args, int n, int dyn, hipStream_t st) {
  static std::atomic<int> max_set[64];  // per device (kernels.h: gtnx_first_on_device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dyn > max_set[dev & 63].load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(compose_kernel<MATCH, L1, L2, FAST, C1, BLK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    max_set[dev & 63].store(dyn, std::memory_order_release);
  }
  hipLaunchKernelGGL((compose_kernel<MATCH, L1, L2, FAST, C1, BLK>), dim3(n), dim3(BLK), dyn, st, d_args);
}
thread_local int g_compose_wide = 0;  // set by launch_compose for the duration of one dispatch (ops may run on several host threads)
template <int MATCH, bool L1, bool L2, bool FAST, bool C1>
void launch_compose_t(const ComposeArgs* d_args, int n, int dyn, hipStream_t st) {
  // the 512-lane form only for chain products (one side linear) on the FAST variant
  if (FAST && (L1 != L2) && g_compose_wide) launch_compose_b<MATCH, L1, L2, FAST, C1, (FAST && (L1 != L2)) ? 512 : 256>(d_args, n, dyn, st);
  else launch_compose_b<MATCH, L1, L2, FAST, C1, 256>(d_args, n, dyn, st);
}
template <int MATCH, bool FAST>
void launch_compose_f(const ComposeArgs* d, int n, int lin1, int lin2, int dyn, int cache1, hipStream_t st) {
  if (lin1 && lin2) launch_compose_t<MATCH, true, true, FAST, false>(d, n, dyn, st);
  else if (lin1) launch_compose_t<MATCH, true, false, FAST, false>(d, n, dyn, st);
  else if (lin2) {
    if (FAST && cache1) launch_compose_t<MATCH, false, true, FAST, FAST>(d, n, dyn, st);
    else launch_compose_t<MATCH, false, true, FAST, false>(d, n, dyn, st);
  } else {
    if (FAST && cache1) launch_compose_t<MATCH, false, false, FAST, FAST>(d, n, dyn, st);
    else launch_compose_t<MATCH, false, false, FAST, false>(d, n, dyn, st);
  }
}
template <int MATCH>
void launch_compose_m(const ComposeArgs* d, int n, int lin1, int lin2, int dyn, int fast, int cache1, hipStream_t st) {
  if (fast) launch_compose_f<MATCH, true>(d, n, lin1, lin2, dyn, cache1, st);
  else launch_compose_f<MATCH, false>(d, n, lin1, lin2, dyn, 0, st);
}
} // namespace

// One instantiation per (matcher, g1 linear?, g2 linear?): every role / kind
// branch of the matcher folds at compile time, which keeps the per-level code
// path a few hundred instructions (the all-in-one kernel was ~40k lines of ISA
// and instruction-cache bound).  All graphs of a launch share the triple.
size_t compose_g1_cache_bytes(int N1, int A1) {
  return 16 + 4 * size_t((N1 + 1 + 3) & ~3) + 16 * size_t(A1) + size_t((N1 + 15) & ~15);
}
// dynamic bytes that keep 2 workgroups per CU next to the static working set
int compose_lds_budget(int wide) { return 80 * 1024 - (wide ? 38 : 19) * 1024; }

void launch_compose(const ComposeArgs* d_args, int n, int matcher, int lin1, int lin2, int dyn_lds_bytes,
                    int fast, int cache1, int wide, hipStream_t st) {
  if (n <= 0) return;
  g_compose_wide = wide;
  switch (matcher) {
    case MATCH_UNSORTED: launch_compose_m<MATCH_UNSORTED>(d_args, n, lin1, lin2, dyn_lds_bytes, fast, cache1, st); break;
    case MATCH_SINGLY_G1: launch_compose_m<MATCH_SINGLY_G1>(d_args, n, lin1, lin2, dyn_lds_bytes, fast, cache1, st); break;
    case MATCH_SINGLY_G2: launch_compose_m<MATCH_SINGLY_G2>(d_args, n, lin1, lin2, dyn_lds_bytes, fast, cache1, st); break;
    default: launch_compose_m<MATCH_DOUBLY>(d_args, n, lin1, lin2, dyn_lds_bytes, fast, cache1, st); break;
  }
}

size_t compose_transpose_scratch_bytes(int n, int maxNcap) {
  const size_t chunks = size_t(maxNcap + kChunk - 1) / kChunk + 1;
  return sizeof(int3) * chunks * size_t(n > 0 ? n : 1);
}

void launch_compose_transpose(const ComposeArgs* d_args, int n, int maxAcap, int maxNcap, void* scratch,
                              hipStream_t st) {
  if (n <= 0) return;
  const int chunks = (maxNcap + kChunk - 1) / kChunk + 1;
  int3* g_sums = static_cast<int3*>(scratch);
  hipLaunchKernelGGL(tr_count_kernel, dim3(grid_x(maxAcap, 2048), n), dim3(kBlock), 0, st, d_args);
  hipLaunchKernelGGL(tr_chunk_sum_kernel, dim3(chunks, n), dim3(kBlock), 0, st, d_args, g_sums, chunks);
  hipLaunchKernelGGL(tr_chunk_scan_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_args, g_sums, n, chunks);
  hipLaunchKernelGGL(tr_offsets_kernel, dim3(chunks, n), dim3(kBlock), 0, st, d_args, (const int3*)g_sums, chunks);
  hipLaunchKernelGGL(tr_scatter_kernel, dim3(grid_x(maxAcap, 2048), n), dim3(kBlock), 0, st, d_args);
}

namespace {
// arrays compose_kernel left out (ComposeArgs::skip), derived from what it did write
__global__ void compose_fill_src_kernel(ComposeFillArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  for (int k = a.out_off[n]; k < a.out_off[n + 1]; ++k) a.src[k] = n;
}
__global__ void compose_fill_arcs_kernel(ComposeFillArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.A) return;
  const int i = a.gi1[k], j = a.gi2[k];
  a.il[k] = i < 0 ? EPS : (a.lab1 ? a.lab1[i] : i % a.C1);  // compose.cpp:443-446
  a.ol[k] = j < 0 ? EPS : (a.lab2 ? a.lab2[j] : j % a.C2);
  // in-rows again, this time with the arc id of every slot (tr_scatter_kernel's job)
  const int pos = atomicAdd(a.in_cursor + a.dst[k], 1);
  a.in_list[pos] = k;
  a.in_src[pos] = a.src[k];
  a.in_w[pos] = a.w[k];
}
} // namespace

void launch_compose_fill(const ComposeFillArgs& a, hipStream_t st) {
  if (a.N > 0) hipLaunchKernelGGL(compose_fill_src_kernel, dim3((a.N + 255) / 256), dim3(256), 0, st, a);
  if (a.A > 0) hipLaunchKernelGGL(compose_fill_arcs_kernel, dim3((a.A + 255) / 256), dim3(256), 0, st, a);
}

void launch_build_records(const DGraph& g, void* out_rec, void* in_rec, hipStream_t st) {
  if (g.A <= 0) return;
  hipLaunchKernelGGL(build_records_kernel, dim3(grid_x(g.A, 1024)), dim3(kBlock), 0, st, g,
                     static_cast<gtnx_i4*>(out_rec), static_cast<gtnx_i4*>(in_rec));
}

void launch_compose_grad(const ComposeGradArgs* d_args, int n, int maxA, hipStream_t st) {
  if (n <= 0 || maxA <= 0) return;
  hipLaunchKernelGGL(compose_grad_kernel, dim3((maxA + kGradChunk - 1) / kGradChunk, n), dim3(kBlock), 0, st, d_args);
}

} // namespace gtnx
