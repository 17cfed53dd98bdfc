// kernels.h -- device-side views (PODs laid out in HBM) and the launch entry
// points of the HIP kernels.  Everything here is plain data + free functions so
// the host runtime (graph.cpp / ops.cpp) never sees kernel internals.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// kernel attributes (hipFuncSetAttribute: the dynamic LDS limit) belong to a (function, DEVICE) pair: a launcher sets
// them the first time it runs on each device, not once per process (runtime.h: one context per device)
#include <atomic>
#include <cstdint>
// `if (auto first = gtnx_first_on_device(done)) { ...hipFuncSetAttribute... }`: the device's bit is published when
// `first` goes out of scope, i.e. AFTER the attribute calls -- a second thread launching on the same device meanwhile
// (the side-stream thread) sees it clear and sets the attributes itself (twice is harmless) instead of launching
// under the default limit
struct gtnx_first_on_device {
  std::atomic<uint64_t>& done;
  uint64_t bit;
  bool first;
  explicit gtnx_first_on_device(std::atomic<uint64_t>& d) : done(d) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = uint64_t(1) << (dev & 63);
    first = !(done.load(std::memory_order_acquire) & bit);
  }
  gtnx_first_on_device(const gtnx_first_on_device&) = delete;
  ~gtnx_first_on_device() {
    if (first) done.fetch_or(bit, std::memory_order_acq_rel);
  }
  explicit operator bool() const { return first; }
};

namespace gtnx {

// Pointers inside the argument structs are loaded from memory, so the compiler
// would treat them as generic ("flat") addresses: flat_load/flat_store, which
// are slower and -- worse -- tick lgkmcnt as well as vmcnt, so an LDS-only
// barrier (s_waitcnt lgkmcnt(0); s_barrier) would still wait for every HBM
// access in flight.  In device code the members are therefore typed as GLOBAL
// (address space 1) pointers; the host sees ordinary pointers of the same size.
#if defined(__HIP_DEVICE_COMPILE__)
#define GTNX_G __attribute__((address_space(1)))
#else
#define GTNX_G
#endif

// 16-byte adjacency record as a builtin vector (loadable from any address space)
typedef int gtnx_i4 __attribute__((ext_vector_type(4)));
typedef float gtnx_f4 __attribute__((ext_vector_type(4)));

enum : int { KIND_EXPLICIT = 0, KIND_LINEAR = 1 };
enum : int { NF_START = 1, NF_ACCEPT = 2, NF_ORPHAN = 4 /* unqueued accept node: score 0.0 */ };

// ---------------------------------------------------------------------------
// Structure-of-arrays view of one graph in HBM (replaces gtn/graph.h:58-73's
// AoS Arc + per-node std::vector<int> in/out).  Arc arrays are in arc-id (API)
// order.  Adjacency is CSR over arc ids: out_list[out_off[n] .. out_off[n+1])
// is Graph::out(n) in the reference's list order, likewise in_*.
// KIND_LINEAR (gtn/creations.cpp:20-33) stores nothing but (M, C): arc a is
// (a / C) -> (a / C + 1) with label a % C; only the weight tensor is material.
// ---------------------------------------------------------------------------
struct DGraph {
  int kind;
  int N, A;
  int M, C;
  int n_start, n_accept;
  int flags;  // bit0 ilabelSorted, bit1 olabelSorted, bit2 GF_EPS_FREE (no epsilon label on any arc), bit3 GF_ACCEPTOR
  const GTNX_G int* src;
  const GTNX_G int* dst;
  const GTNX_G int* il;
  const GTNX_G int* ol;
  const GTNX_G uint8_t* nflags;
  const GTNX_G int* start_list;
  const GTNX_G int* accept_list;
  const GTNX_G int* out_off;
  const GTNX_G int* out_list;  // nullptr => identity (arcs already grouped by src in id order)
  const GTNX_G int* in_off;
  const GTNX_G int* in_list;
  // packed adjacency records in list order: {ilabel, olabel, dst (out) / src (in), arc id};
  // present for host-built graphs, nullptr for device-built ones
  const GTNX_G gtnx_i4* out_rec;
  const GTNX_G gtnx_i4* in_rec;
  const GTNX_G float* w;  // weights, arc-id order (filled per op; not part of the structure)
};

// ---------------------------------------------------------------------------
// Level schedule of a DAG for shortest distance, in *position space*: nodes
// renumbered by their place in the reference's Kahn FIFO order
// (gtn/functions/shortest.cpp:96-145), grouped into dependency levels.
//   rows   : row_off[p] .. row_off[p+1] index in_srcpos/in_arc  (in-arcs of p)
//   in_w   : optional weights permuted into row order (compose emits it)
//   out_*  : the transposed rows for the backward sweep
// Device-built layered products have position == node id and out_arc == nullptr
// (arc k is the k-th out entry).
// ---------------------------------------------------------------------------
struct DSched {
  int P;  // scheduled nodes
  int L;  // levels
  int n_accept;
  int flags;  // bit0: tie-break by arc id (rows unordered); bit1: out rows are identity
  const GTNX_G int* level_off;  // [L+1]
  const GTNX_G int* row_off;    // [P+1]
  const GTNX_G int* in_srcpos;  // [Ain]
  const GTNX_G int* in_arc;     // [Ain] arc ids
  const GTNX_G int* in_rank;    // [Ain] push rank for viterbiPath ties (nullptr => arc id)
  const GTNX_G float* in_w;     // [Ain] or nullptr
  const GTNX_G uint8_t* pflags; // [P] NF_START | NF_ACCEPT by position
  const GTNX_G int* acc_pos;    // [n_accept] position of accept()[k]
  const GTNX_G int* out_off;    // [P+1]
  const GTNX_G int* out_dstpos; // [Aout]
  const GTNX_G int* out_arc;    // [Aout] or nullptr (identity)
};
enum : int { SCHED_TIE_BY_ARC = 1, SCHED_OUT_IDENTITY = 2 };

// the same schedule built on the device from the graph's CSR arrays (levelize.hip); positions inside a level
// are in node-id order (no queue replay: not for viterbiPath's tie-break)
struct LevelizeOut {
  int* level_off;    // [N + 2]
  int* row_off;      // [N + 1]
  int* in_srcpos;    // [A]
  int* in_arc;       // [A]
  uint8_t* pflags;   // [N]
  int* acc_pos;      // [n_accept]
  int* out_off;      // [N + 1]
  int* out_dstpos;   // [A]
  int* out_arc;      // [A]
};
enum : int { LV_P = 0, LV_L, LV_N_IN, LV_N_OUT, LV_ERROR, LV_MAX_WIDTH, LV_MAX_LEVEL_ARCS, LV_MAX_REACH, LV_INFO_INTS };
size_t levelize_scratch_bytes(int N, int A);
void device_levelize(const DGraph& g, const LevelizeOut& out, void* scratch, int* info_host /* LV_INFO_INTS */, hipStream_t st);

// per-graph result of the forward sweep (kept for the backward sweep)
struct SdResult {
  float score;      // shortest.cpp:159
  float max_final;  // maxScoresCache.back()
  int argmax_final; // position of the arg-max accept node (tropical), -1 if none
  int pad;
};

// one shortest-distance problem
struct SdArgs {
  DSched s;
  const GTNX_G float* w;   // arc-id order weights
  GTNX_G float* scores;    // [P]
  GTNX_G int* argmax;      // [P] tropical only (arc id, -1 = the start node's virtual 0)
  GTNX_G SdResult* result; // [1]
  GTNX_G float* out_score; // [1] the scalar graph's weight
  // backward
  const GTNX_G float* delta;  // [1] upstream gradient of the scalar
  GTNX_G float* node_grad;    // [P]
  GTNX_G float* arc_grad;     // [A] arc-id order
  int chunk_levels;           // narrow kernels: levels per LDS chunk (host: caps / widest level)
  // narrow kernels: when set, P / L / accept count are read from the producing compose's
  // result block on the device (the host may not know them yet -- graph.h DeferredSizes)
  const GTNX_G struct ComposeOut* dyn_out;
  const GTNX_G int* dyn_counts;
  // fused compose-gradient scatter (narrow backward only; see sd_narrow_fuse_caps):
  // arc k of the lattice came from arc gi_fixed[k] of the explicit compose input and
  // arc gi_chain[k] of the linear chain, whose arcs of level l lie in [l*C, (l+1)*C)
  const GTNX_G int* gi_fixed;
  const GTNX_G int* gi_chain;
  GTNX_G float* grad_fixed;   // [fixed_A] zero-filled by the host; may be null
  GTNX_G float* grad_chain;   // [T*C]     zero-filled by the host; may be null
  int chain_C, fixed_A, chain_A;
  int chain_accumulate;  // grad_chain already holds a gradient: rows are added to, not stored
};

enum : int { SD_LOG = 0, SD_TROPICAL = 1, SD_PATH = 2 };

// narrow != 0 (2: every graph carries row-ordered weights in_w) selects the LDS-ring kernel for deep, narrow lattices (every graph
// of the batch must satisfy: level arcs <= sd_narrow_tmp_cap(), per-level reach
// <= sd_narrow_ring()); only the log semiring has it so far.
int sd_narrow_ring();
int sd_narrow_tmp_cap();
int sd_narrow_node_cap();
void launch_sd_forward(const SdArgs* d_args, int n, int mode, int narrow,
                       int avg_in_degree_x16, hipStream_t st);
// deep, thin DAGs in the log semiring (thousands of levels of one or two nodes): one wave per graph, scores /
// node gradients in LDS, rows staged a chunk ahead (shortest.hip).  Every graph needs P <= sd_deep_node_cap().
int sd_deep_node_cap();
void launch_sd_forward_deep(const SdArgs* d_args, int n, int maxP, hipStream_t st);
void launch_sd_backward_deep(const SdArgs* d_args, int n, int maxP, hipStream_t st);
int sd_narrow_ring_backward();
// narrow != 0: LDS-ring kernel (log semiring, identity out rows, eligibility as forward
// with reach <= sd_narrow_ring_backward())
// narrow == 2: as 1, plus the fused gradient scatter into the compose inputs (every
// graph must satisfy fixed_A <= cap_fixed and chunk_levels * chain_C <= cap_chain)
// fuse_lds_bytes (narrow == 2): dynamic LDS of the chain window, max over the batch of
// chunk_levels * chain_C * 4
void launch_sd_backward(const SdArgs* d_args, int n, int mode, int narrow, int avg_out_degree_x16, hipStream_t st,
                        int fuse_lds_bytes = 0);
void sd_narrow_fuse_caps(int* cap_fixed, int* cap_chain);

// viterbiPath pointer chase (shortest.cpp:239-245): writes path arc ids first-arc-first
struct PathArgs {
  DSched s;
  DGraph g;
  const GTNX_G int* argmax;       // back-pointers by position (arc id / -1)
  const GTNX_G SdResult* result;
  GTNX_G int* path_arcs;          // [cap]
  GTNX_G int* path_il;            // [cap]
  GTNX_G int* path_ol;            // [cap]
  GTNX_G float* path_w;           // [cap]
  GTNX_G int* path_len;           // [4]: length, has_node, exact tie on the path, -
  int cap;
  // exact-tie check of the visited nodes (path_tie_kernel): node scores by position, weights by arc id
  const GTNX_G float* scores;
  const GTNX_G float* w;
  GTNX_G int* path_pos;           // [cap] position of the node each path arc enters
  GTNX_G int* pred;               // [P] scratch: position of the node each position's best in-arc leaves (-1: none)
  GTNX_G int* tmp;                // [cap] scratch: the visited positions, last first
};
void launch_path_chase(const PathArgs* d_args, int n, int max_cap, int max_P, hipStream_t st);

// ---------------------------------------------------------------------------
// linear-chain emissions graphs: forwardScore / viterbiScore and their grads
// are row reductions over the [M][C] weight tensor (no graph traversal).
// ---------------------------------------------------------------------------
struct LinArgs {
  const GTNX_G float* w;   // [M][C]
  int M, C;
  GTNX_G float* out_score; // [1]
  GTNX_G float* partial;   // [splits]
  const GTNX_G float* delta;
  GTNX_G float* grad;      // [M][C]
  int accumulate;          // backward: grad += (the graph already holds a gradient) instead of grad =
};
// vec_rows != 0 (log semiring only): every member has C % 4 == 0, C <= 1024 and 16-byte
// aligned weight / gradient rows -- rows are reduced out of registers (linear_rows_kernel)
void launch_linear_forward(const LinArgs* d_args, int n, int tropical, int vec_rows, hipStream_t st);
void launch_linear_backward(const LinArgs* d_args, int n, int tropical, int vec_rows, hipStream_t st);

// ---------------------------------------------------------------------------
// composition (gtn/functions/compose.cpp:377-522)
// ---------------------------------------------------------------------------
enum : int { MATCH_UNSORTED = 0, MATCH_SINGLY_G1 = 1, MATCH_SINGLY_G2 = 2, MATCH_DOUBLY = 3 };

struct ComposeOut {
  int N, A, L;
  int layered;   // every arc goes from BFS level k to k+1
  int overflow;  // capacity exceeded (host bound was wrong) -- never expected
  int max_width;       // widest BFS level (nodes)
  int max_level_arcs;  // most arcs emitted by one level
  int csr_built;       // in_off / in_list / in_src / in_w were built inside compose_kernel
  int rep_levels;      // BFS levels emitted by stationary-level replication (not expanded one by one)
  int skipped;         // src / il / ol / in_list were NOT written (ComposeArgs::skip honoured); see compose_fill
  int t_b, t_f, t_rep; // 100 MHz ticks spent in phase B / phase F (total) / replication (diagnostics)
  // compose_wide.hip: the hole its plan kernel left for the replication kernel -- levels wr_L + 1 .. wr_L + wr_K
  // are level wr_L (frontier ids wr_lo .. wr_lo + wr_W, arcs wr_na .. wr_na + wr_Aw) moved in time
  int wr_L, wr_K, wr_lo, wr_W, wr_na, wr_Aw;
};

constexpr int GF_EPS_FREE = 4;
constexpr int GF_ACCEPTOR = 8;  // ilabel == olabel on every arc (known at upload; products of acceptors inherit it)

struct ComposeArgs {
  DGraph g1, g2;
  int matcher;
  int lds_state;  // co-reachability / discovered bitmaps live in LDS
  // FAST variant only: leave out the arrays that are derivable from the others (src from
  // out_off; il / ol from gradInfo and the inputs' labels; in_list by re-scattering the
  // in-rows) -- 16 of the 40 bytes a composed arc costs.  launch_compose_fill() writes
  // them when somebody needs them (inspection, another compose, viterbi, generic kernels).
  int skip;
  // FAST variant, chain product with an epsilon-free partner: > 0 selects the time-windowed
  // bitmap layout with this many time slices (compose.hip); 0 = classic pair-indexed bitmaps
  int chain_bits;
  // FAST variant, chain products: != 0 leaves the stationary levels to launch_compose_replicate(), which must
  // follow on the same stream (ComposeOut::wr_* is the hand-over)
  int rep_grid;
  int Ncap, Acap;
  GTNX_G int* state;     // [N1*N2] pair -> INT_MIN unreachable / R / claim / node id
  GTNX_G int* queue;     // [N1*N2] backward-BFS queue of pair ids
  // outputs (SoA, arc-id order)
  GTNX_G int* src;
  GTNX_G int* dst;
  GTNX_G int* il;
  GTNX_G int* ol;
  GTNX_G float* w;
  GTNX_G int* gi1;       // gradInfo.first  (compose.cpp:443-446)
  GTNX_G int* gi2;       // gradInfo.second
  GTNX_G uint8_t* nflags;
  GTNX_G int* pair_of;   // [Ncap] composed node -> pair id
  GTNX_G int* out_off;   // [Ncap+1]
  GTNX_G int* level_off; // [Ncap+2]
  // in-CSR (filled by the transpose kernels)
  GTNX_G int* in_off;    // [Ncap+1]
  GTNX_G int* in_cursor; // [Ncap]
  GTNX_G int* in_list;   // [Acap]
  GTNX_G int* in_src;    // [Acap]
  GTNX_G float* in_w;    // [Acap]
  GTNX_G int* start_list;  // [Ncap]
  GTNX_G int* accept_list; // [Ncap]
  GTNX_G int* counts;      // [2] n_start, n_accept
  GTNX_G ComposeOut* out;
  // compose_pairs_kernel only: the lists it binary-searches, sorted on the matched label -- g1's out / in
  // records by olabel, g2's by ilabel (the graph's own records when it is sorted on that label, else a stable
  // sorted view made by launch_sorted_view); null for a side the matcher never searches
  int trim_fwd_first;  // compose_pairs_kernel: mark the pairs reached from the start pairs first (see the kernel)
  const GTNX_G gtnx_i4* s1_out;
  const GTNX_G gtnx_i4* s1_in;
  const GTNX_G gtnx_i4* s2_out;
  const GTNX_G gtnx_i4* s2_in;
};
// dyn_lds_bytes: 2 bitmaps of N1*N2 bits for the largest pair table of the batch
// when every graph has lds_state set, else 0
int compose_max_bitmap_bytes();
// bytes of the chain layout for a partner with `No` nodes and `slices` time slices
size_t compose_chain_bitmap_bytes(int No, int slices);
// all n problems share (matcher, g1 is linear, g2 is linear)
// fast != 0: the compact LDS-only variant (needs dyn_lds_bytes > 0); pairs it cannot
// handle come back with ComposeOut::overflow == 2 and must be re-run with fast = 0
// cache1 != 0 (fast only): g1's adjacency is cached in LDS; dyn_lds_bytes then
// covers the bitmaps plus compose_g1_cache_bytes() of the largest g1
size_t compose_g1_cache_bytes(int N1, int A1);
int compose_lds_budget(int wide);
// wide != 0 (fast chain products only): 512-lane workgroups, partners of up to 512 nodes
void launch_compose(const ComposeArgs* d_args, int n, int matcher, int lin1, int lin2, int dyn_lds_bytes,
                    int fast, int cache1, int wide, hipStream_t st);
// compose_wide.hip: chain products (exactly one side KIND_LINEAR, chain length >= 1) with an epsilon-free partner
// of at most compose_wide_node_cap() nodes whose arcs match in the partner's list order (see the file's header);
// any out-degree.  All n pairs share `lin2` (the chain is the second graph).  Leaves ComposeOut::csr_built = 0:
// launch_compose_transpose() follows.
// the stationary levels a FAST chain-product launch left out (ComposeArgs::rep_grid): arcs, in-rows, node
// arrays of levels wr_L + 1 .. wr_L + wr_K from level wr_L, with as many workgroups as the output deserves
void launch_compose_replicate(const ComposeArgs* d_args, int n, int lin2, int max_acap, hipStream_t st);
int compose_wide_node_cap();
void launch_compose_wide(const ComposeArgs* d_args, int n, int lin2, int max_acap, hipStream_t st);
// compose_wide.hip, general products (both graphs explicit, any degrees, epsilons allowed): a wave per frontier
// pair.  `state` must arrive filled with INT_MIN; s1_* / s2_* set for the side(s) `matcher` searches (UNSORTED
// searches g2 through its sorted view).  Leaves csr_built = 0.
void launch_compose_pairs(const ComposeArgs* d_args, int n, int matcher, hipStream_t st);
void launch_sorted_view(const DGraph& g, int key_olabel, void* out_view /* [A] 16-byte records */, void* in_view, hipStream_t st);
struct ComposeFillArgs {
  int N, A;
  const GTNX_G int* out_off;
  const GTNX_G int* dst;
  const GTNX_G float* w;
  const GTNX_G int* gi1;
  const GTNX_G int* gi2;
  const GTNX_G int* lab1;  // ilabels of input 1 (null: linear chain with C1 labels per step)
  const GTNX_G int* lab2;  // olabels of input 2
  int C1, C2;
  GTNX_G int* src;
  GTNX_G int* il;
  GTNX_G int* ol;
  GTNX_G int* in_cursor;   // [N] scratch, preset to in_off
  GTNX_G int* in_list;
  GTNX_G int* in_src;
  GTNX_G float* in_w;
};
void launch_compose_fill(const ComposeFillArgs& a, hipStream_t st);
size_t compose_transpose_scratch_bytes(int n, int maxNcap);
void launch_compose_transpose(const ComposeArgs* d_args, int n, int maxAcap, int maxNcap, void* scratch,
                              hipStream_t st);

// out_rec / in_rec (16 B * A each) for a device-built explicit graph
void launch_build_records(const DGraph& g, void* out_rec, void* in_rec, hipStream_t st);

struct ComposeGradArgs {
  const GTNX_G float* delta; // [A] grads of the composed arcs
  const GTNX_G int* gi1;
  const GTNX_G int* gi2;
  int A;
  int A1, A2;
  GTNX_G float* grad1; // nullptr when input 0 has calcGrad = false
  GTNX_G float* grad2;
};
void launch_compose_grad(const ComposeGradArgs* d_args, int n, int maxA, hipStream_t st);

// ---------------------------------------------------------------------------
// lazy chain products (lazy.hip): shortest distance / path / gradients over
// chain (T x C emissions) o G without building the product
// ---------------------------------------------------------------------------
struct LazyGroup {          // utterances that share one explicit graph G
  DGraph g;                 // G: CSR + packed records + weights
  int chain_first;          // compose(chain, G): match on G's ilabel; else on its olabel
  int T, C, N, nb;          // chain length / labels per step, |G| nodes, utterances
  int Npad, Cpad;           // LDS row strides (odd: 16 utterance rows hit 16 banks)
  // G's rows repacked for the time-step kernels: {other node, matched label or -1,
  // weight bits, arc id} in in-row / out-row order -- one 16-byte load per arc
  const gtnx_i4* lrec_in;
  const gtnx_i4* lrec_out;
  const float* const* em;   // [nb] chain weights, T*C each
  const float* em_base;     // ... or, when the chains are slices of one tensor at a constant stride (floats),
  int64_t em_stride;        //     em[b] == em_base + b * em_stride (null / 0: use the table)
  float* alpha;             // [T+1][nb][N]
  float* beta;              // [T+1][nb][N] (gradients only)
  int* bp;                  // [T+1][nb][N] back-pointers (arc of G), tropical only
  float* score;             // [nb]
  int* best;                // [nb] best accept node of G (-1: no path), tropical only
  const float* const* delta;  // [nb] upstream gradient of each score
  float* const* grad_em;      // [nb] chain gradient buffers (T*C) or null entries
  float* grad_fixed;          // [A] gradient of G's arcs (zero-filled) or null
  // dense regime (log semiring; every node's in-arcs share one matched label and G is
  // nearly complete): the recursion is a product with E = exp(w - cmax) in the
  // probability domain -- see lazy.hip "dense regime"
  const float* E;             // [N][N] row = source, col = destination (0: no arc)
  const float* cmax;          // [N] largest in-arc weight per destination (-inf: none)
  const int* nlab;            // [N] matched label of the node's in-arcs (-1: none)
  int lab_unique;             // no two nodes share a matched label (gradient rows need no atomics)
  int tie_by_node;            // maxplus_path_kernel: of equal maxima the one from the smallest SOURCE NODE (viterbiPath of
                              // a transitions graph whose layers the reference's queue visits in node order); 0: the
                              // first one in in-row order (the reference's in-list order: viterbiScore's gradient)
  float* amax;                // [T+1][nb] row max of alpha[t]   (written by the forward steps)
  float* bmax;                // [T][nb]   row max of em + beta[t+1] + cmax (backward steps)
  float* amaxp;               // [T+1][nb][ntp] the same, one partial per column tile (matrix-core form: reduced into
  float* bmaxp;               //                amax / bmax once a pass is through, launch_lazy_mfma_rowmax)
  int ntp;                    // partials per row: column tiles rounded up to 4
  float* R;                   // [N][N] sum over (t, utterance) of the arc posteriors / exp(w)
  // gradients: per (t, utterance) normaliser log sum_n exp(alpha[t+1][n] + beta[t+1][n]).  It
  // equals the total score exactly in exact arithmetic; in float32 the two sweeps drift apart
  // by a few ulps of |score| over T steps, and normalising each time step by its own sum keeps
  // every step's posterior mass at 1 (the standard alpha-beta remedy).  Null: use `score`.
  const float* zt;
  // matrix-core form of the dense regime (lazy.hip: lazy_mfma_*)
  float* xt[2];               // [Kpad / 4][nbpad][4] the contraction input, exponentiated, in operand layout (two planes)
  const float* Ep;            // [Kpad / 4][Npad2][4] E zero-padded
  const float* ETp;           // [Kpad / 4][Npad2][4] its transpose
  int Kpad, Npad2, nbpad;     // N rounded up to operand batches / to 32; nb rounded up to 32 (max-plus form: to 64)
  int rot;                    // leading nodes without a matched in-arc: the planes index nodes rotated by this many
  int rel;                    // alpha / beta / amax / bmax / zt are stored RELATIVE to per-row references (matrix-core form,
                              // lazy.hip "dense regime on the matrix cores"): alpha[t] against RA[t] = sum of amax[0 .. t-1]
  // max-plus form of the dense regime (tropical semiring; maxplus.hip): xt holds alpha itself
  const float* mp_Wq;         // [dblock][Kpad / 2][16][2] largest weight per (source, destination column), -inf: no arc
  const int* mp_colidx;       // [N] node -> destination column, -1 for nodes without a matched in-arc
  const int* mp_colnode;      // [mp_ncol] column -> node
  const int* mp_dead;         // [mp_ndead] the nodes without a matched in-arc
  int mp_ncol, mp_ndead;
};
size_t lazy_step_lds_bytes(const LazyGroup& g);
int lazy_tile_nodes();
int lazy_tile_batch();
void launch_lazy_pack(const LazyGroup& g, gtnx_i4* lrec_in, gtnx_i4* lrec_out, hipStream_t st);
void launch_lazy_init(const LazyGroup& g, int which, hipStream_t st);
void launch_lazy_step(const LazyGroup& g, int t, int mode, int backward, hipStream_t st);
void launch_lazy_final(const LazyGroup& g, int mode, hipStream_t st);
void launch_lazy_path(const LazyGroup& g, int* path_arc, int* path_il, int* path_ol, float* path_w, int* path_len,
                      hipStream_t st);
// node_label != null: every node's in-arcs share one matched label (no arc loop)
void launch_lazy_local_z(const LazyGroup& g, float* zt, hipStream_t st);  // [T][nb]
void launch_lazy_chain_grad(const LazyGroup& g, const int* node_label, hipStream_t st);
// local z + chain gradient in one pass (shared in-arc labels, N <= 1024): writes zt like launch_lazy_local_z
bool lazy_z_chain_grad_ok(const LazyGroup& g);
void launch_lazy_z_chain_grad(const LazyGroup& g, const int* node_label, float* zt, hipStream_t st);
void launch_lazy_fixed_grad(const LazyGroup& g, int max_in_deg, hipStream_t st);
struct LazyPathGrad {
  const float* delta;   // [len] (stride 1) or one scalar (stride 0)
  int delta_stride;
  const int* path_arc;  // [len] arcs of G, first-arc-first
  const int* il;
  const int* ol;
  int len, C, chain_first;
  float* grad_chain;    // [T*C] or null
  float* grad_fixed;    // [A] or null
};
void launch_lazy_path_grad(const LazyPathGrad& a, hipStream_t st);
// one workgroup per (chain, small G) pair: lazy_pair.hip
struct LazyPair {
  DGraph g;                    // G with packed records and weights
  const GTNX_G float* em;      // [T][C] chain weights
  GTNX_G float* alpha;         // [T+1][N]
  GTNX_G float* score;         // [1]
  const GTNX_G float* delta;   // [1] upstream gradient of the score   (backward)
  GTNX_G float* grad_em;       // [T][C] written completely, or null   (backward)
  GTNX_G float* grad_fixed;    // [A] zero-filled by the host, or null (backward)
  int T, C, chain_first, pad;
};
int lazy_pair_max_nodes();
int lazy_pair_max_degree();
int lazy_pair_block(int max_nodes);   // lanes per workgroup for a batch whose largest G has max_nodes
int lazy_pair_max_labels(int block);
// all pairs of one launch share C; cus: compute units of the device (workgroups are spread evenly)
void launch_lazy_pair_forward(const LazyPair* d_pairs, int n, int block, int C, int cus, hipStream_t st);
void launch_lazy_pair_backward(const LazyPair* d_pairs, int n, int block, int C, int cus, hipStream_t st);
// one workgroup per (chain, BANDED G) pair: band.hip.  G's arcs go from n to n, n+1 or n+2,
// at most one per (n, step), and all in-arcs of a node carry one matched label.
struct BandNode {
  int lab;     // matched label of the node's in-arcs (-1: no in-arc)
  int aid[3];  // arc id of the in-arc from n, n-1, n-2 (-1: none)
};
struct BandPair {
  const GTNX_G BandNode* nodes;    // [N]
  const GTNX_G uint8_t* nflags;    // [N] NF_START | NF_ACCEPT
  const GTNX_G int* snode;         // [n_lab] nodes with an in-arc, sorted by (label, node)
  const GTNX_G int* slab;          // [n_lab] their labels
  const GTNX_G float* w;           // G's weights, arc-id order; null: all zero
  const GTNX_G float* em;          // [T][C] chain weights
  GTNX_G float* em_copy;           // forward only, or null: every emission the sweep stages is also stored here (the
                                   // copy a region owes its emission graphs, region.cpp PendingCopy: one pass
                                   // over the caller's tensor instead of two)
  GTNX_G float* alpha;             // [T+1][NS] log2 units, shifted rows
  GTNX_G double* aoff;             // [0]: the score in log2 units; [1 + (r >> lgrn) * 4 + w]: shift of alpha row r, wave w
  GTNX_G float* score;             // [1]
  GTNX_G float* norm;              // [1] forwardScore of the chain itself, or null
  GTNX_G float* rowlse;            // [T] log2-sum-exp2 of every emission row, or null
  const GTNX_G float* delta;       // [1] upstream gradient of the score         (backward)
  const GTNX_G float* delta_norm;  // [1] upstream gradient of norm, or null     (backward)
  GTNX_G float* grad_em;           // [T][C] written completely, or null         (backward)
  GTNX_G float* grad_fixed;        // [A] zero-filled by the host, or null       (backward)
  int N, T, C, NS;
  int hot;                         // label shared by >= 8 nodes of G (CTC: blank), or -1
  int lgrn;                        // log2 of the rows per shift period of the forward launch
  int n_lab;
  int bidx;                        // the pair's index in its batch record (BandPatch)
  int64_t goff;                    // offset of G's arc gradients inside the batch's gradient block (BandPatch)
};
// A backward launch over the table the FORWARD launch left on the device: what differs between the two launches of a
// batch record -- where the upstream gradients are read and the gradients written -- travels by value with the launch,
// as bases the kernel offsets by the pair's index (one table upload, a 5 us launch of its own, less per step).
struct BandPatch {
  int on;                          // 0: the table is complete (every launch but a batch record's backward)
  int M;                           // rows of a chain (stride of rowlse)
  const GTNX_G float* delta;       // [n]
  const GTNX_G float* delta_norm;  // [n] or null
  GTNX_G float* rowlse;            // [n][M] or null
  GTNX_G float* grad_em;           // [n][A] or null
  GTNX_G float* grad_fixed;        // + goff, or null
  int64_t A;                       // arcs of a chain (stride of grad_em)
};
// viterbiScore / viterbiPath of chain o (banded G): one workgroup per pair (band.hip)
struct BandDecode {
  const GTNX_G BandNode* nodes;  // [N]
  const GTNX_G uint8_t* nflags;  // [N]
  const GTNX_G float* w;         // G's weights (natural log), arc-id order; null: all zero
  const GTNX_G float* em;        // [T][C]
  GTNX_G uint8_t* bp;            // which in-arc won (0: from n, 1: n-1, 2: n-2; 3: none): [T][NS] bytes, or 2 bits per
                                 // (time, node) packed per lane (band_viterbi_wave_kernel); T * NS + 512 bytes either way
  GTNX_G int* pnode;             // [T + 1] nodes of the best path
  GTNX_G int* path_arc;          // [T] arcs of G along it, first-arc-first
  GTNX_G int* path_lab;          // [T] matched labels
  GTNX_G float* path_w;          // [T] weights of the product's arcs
  GTNX_G int* path_len;          // [1] T, or -1 when no accepting path exists
  GTNX_G float* score;           // [1]
  GTNX_G int* tie;               // [1] an exact tie between finite candidates was seen
  // ranked launches only (exact ties of CTC-shaped targets decided without the lattice, ops_band.cpp tie_ranks):
  // of two equal candidates the one whose SOURCE node has the smaller rank_in wins, of two equal accept nodes the
  // one with the smaller rank_acc
  const GTNX_G int* rank_in;     // [N]
  const GTNX_G int* rank_acc;    // [N]
  int N, T, C, NS;
  int stage_floats, pad;         // LDS staging area of the launch (floats)
};
// ranked != 0: band_viterbi_wave_kernel's RANKED variant (needs the shapes the wave kernel takes: vec, C <= 2048)
void launch_band_viterbi(const BandDecode* d_pairs, int n, int stage_floats, int max_nodes, int max_labels, int vec,
                         hipStream_t st, int ranked = 0);
bool band_viterbi_wave_ok(int max_nodes, int max_labels, int vec);
// ---------------------------------------------------------------------------
// rational.hip: clone / concat / closure / union_ (functions.cpp:66-223) built on the device
// ---------------------------------------------------------------------------
struct RationalSeg {      // one input graph's place in the output
  DGraph g;               // its device view, weights included (implicit chains: n_start = n_accept = 1)
  int node_off, arc_off;  // where its nodes / arcs start
  int conn_off;           // where the epsilon connectors INTO it (concat) / around it (closure) start
  int keep_start, keep_accept;
  int pad;
};
struct RationalOut {
  int N, A;
  GTNX_G int* src;
  GTNX_G int* dst;
  GTNX_G int* il;
  GTNX_G int* ol;
  GTNX_G float* w;
  GTNX_G uint8_t* nflags;
  GTNX_G int* start_list;
  GTNX_G int* accept_list;
  GTNX_G int* out_off;
  GTNX_G int* out_list;
  GTNX_G int* in_off;
  GTNX_G int* in_list;
};
size_t rational_csr_temp_bytes(int N, int A);
// projection: 0 none, 1 input, 2 output (functions.h Projection); closure != 0: node 0 is the new start / accept node
void launch_rational_build(const RationalSeg* d_segs, int nseg, int max_A, int max_N, int max_conn, const RationalOut& out,
                           int projection, int closure, void* temp, hipStream_t st);
void launch_rational_adjacency(const RationalOut& out, void* temp, hipStream_t st);
// remove (functions.cpp:253-318): kept nodes = start nodes and nodes with an in-arc that does not carry the removed
// label pair; per kept node the reference's breadth-first walk over the removed arcs (queue order, out-list order),
// one lane per kept node, in batches of `rows` nodes that share the stamp / queue scratch
struct RemoveArgs {
  DGraph g;                 // explicit, adjacency resident
  int ilabel, olabel;
  const GTNX_G int* new_id;  // [N] kept node -> its id in the result (exclusive scan of the keep flags)
  const GTNX_G int* roots;   // [K] result node -> node of g
  int K, root0, rows;
  GTNX_G int* stamp;         // [rows][N], zero before the first batch
  GTNX_G int* queue;         // [rows][N]
  GTNX_G int* arc_cnt;       // [K] (count pass)
  const GTNX_G int* arc_off; // [K + 1] (emit pass)
  RationalOut out;
};
void launch_remove_keep(const DGraph& g, int ilabel, int olabel, int* keep, hipStream_t st);
void launch_remove_roots(const int* keep, const int* new_id, int N, int* roots, hipStream_t st);
void launch_remove_walk(const RemoveArgs& a, bool emit, hipStream_t st);
size_t scan_temp_bytes(int n);
void launch_exclusive_scan(const int* in, int* out, int n, void* temp, size_t temp_bytes, hipStream_t st);
// the binary graph format's arc table ({src, dst, ilabel, olabel} x A, 16-byte aligned), weights and node flags, all
// on the device, into a structure's arrays + adjacency (the device side of gtnx_graph_load_buffer)
void launch_rational_load(const void* rows, const float* w, const uint8_t* flags, const RationalOut& out, void* temp,
                          hipStream_t st);
// one CTC target acceptor per label sequence, written as band records on the device (band.hip)
struct CtcTargetArgs {
  const GTNX_G int* labels;   // [U]
  GTNX_G BandNode* nodes;     // [N = 2U + 1]
  GTNX_G uint8_t* nflags;     // [N]
  GTNX_G int* snode;          // [N]
  GTNX_G int* slab;           // [N]
  GTNX_G int* n_arcs;         // [1] or null
  int N, pad;
};
void launch_ctc_targets(const CtcTargetArgs* d_args, int n, int blank, hipStream_t st);
// one force-alignment acceptor o transitions per label sequence (examples/asg.cpp:50-68), as band records
struct AsgFalArgs {
  const GTNX_G int* labels;      // [U]
  const GTNX_G float* trans_w;   // weights of the transitions graph, asgTransitions arc layout
  GTNX_G BandNode* nodes;        // [U + 1]
  GTNX_G uint8_t* nflags;        // [U + 1]
  GTNX_G int* snode;             // [U]
  GTNX_G int* slab;              // [U]
  GTNX_G float* w;               // [2U] arc weights (gathered)
  GTNX_G int* arc_map;           // [2U] transitions arc of every arc
  int U, pad;
};
void launch_asg_fal_targets(const AsgFalArgs* d_args, int n, int n_labels, hipStream_t st);
void launch_asg_fal_scatter(const float* g, const int* maps, const int64_t* tab /* 3 per sequence */, int n_seq, int64_t longest,
                            float* trans_grad, hipStream_t st);
int band_max_nodes();
int band_max_labels();
int band_min_labels();
int band_npl(int max_nodes);               // nodes per lane: 1 or 2
int band_row_stride(int N, int npl);       // NS
int band_forward_lgrn(int C);
// all pairs of one launch share C and npl; unit: self-loop + previous-node arc at every node, all weights 0;
// vec: C % 4 == 0 and every pair's emissions are 16-byte aligned (16-byte staging loads)
// `one` (HOST memory; a launch of one pair for which band_one_ok holds): the record travels as the kernel's own argument
// and d_pairs is not read
bool band_one_ok(int npl, int C, int max_NS, bool vec, bool backward);
void launch_band_forward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, bool vec, hipStream_t st,
                         const BandPair* one = nullptr);
// patch (optional; not with `one`): d_pairs is the table a FORWARD launch used -- see BandPatch
void launch_band_backward(const BandPair* d_pairs, int n, int npl, int C, int max_NS, bool unit, bool gradg, bool vec,
                          hipStream_t st, const BandPair* one = nullptr, const BandPatch* patch = nullptr);
// dense regime
void launch_lazy_dense_prep(const LazyGroup& g, float* E, float* cmax, hipStream_t st);  // nlab must be set
// backward: vin / vout = the two halves of a [2][nb][N] scratch (vin null on the first step)
void launch_lazy_dense_step(const LazyGroup& g, int t, int backward, hipStream_t st, const float* vin = nullptr,
                            float* vout = nullptr);
void launch_lazy_dense_fixed_grad(const LazyGroup& g, hipStream_t st);                   // R zero-filled
size_t maxplus_w_floats(const LazyGroup& g);                                   // mp_ncol / Kpad set
void launch_maxplus_prep(const LazyGroup& g, hipStream_t st);                  // mp_Wq, both input planes
void launch_maxplus_step(const LazyGroup& g, int t, hipStream_t st);           // alpha[t] -> alpha[t+1]
void launch_maxplus_path(const LazyGroup& g, int* path_arc, int* path_il, int* path_ol, float* path_w, int* path_len,
                         hipStream_t st);
void launch_lazy_mfma_prep(const LazyGroup& g, hipStream_t st);                // Ep / ETp from E
void launch_lazy_mfma_init(const LazyGroup& g, int which, hipStream_t st);     // keys, first input (0 forward, 1 backward)
void launch_lazy_mfma_step(const LazyGroup& g, int t, int backward, hipStream_t st);
// the whole pass (steps 0 .. T-1, or T-1 .. 0) as ONE cooperative launch: workgroups of a row tile hand their
// step's output to each other through agent-scope stores / loads and a counter; false when not applicable
size_t lazy_mfma_chain_sync_ints(const LazyGroup& g);
bool launch_lazy_mfma_chain(const LazyGroup& g, int backward, int* zeroed_sync, int cus, hipStream_t st);
void launch_lazy_mfma_keys(float* keys, int64_t n, hipStream_t st);            // order-preserving integer keys -> floats
void launch_lazy_mfma_rowmax(const LazyGroup& g, int which, hipStream_t st);   // amaxp -> amax (0) / bmaxp -> bmax (1)
void launch_lazy_mfma_score(const LazyGroup& g, hipStream_t st);               // score (relative, lazy_final) += sum of amax, float64
// R zero-filled.  partials: scratch of lazy_mfma_fixed_grad_scratch_bytes(g) bytes or null (then the blocks meet in R by atomics)
size_t lazy_mfma_fixed_grad_scratch_bytes(const LazyGroup& g);
void launch_lazy_mfma_fixed_grad(const LazyGroup& g, void* pair_consts /* 16 B x T x nb */, hipStream_t st, void* partials = nullptr,
                                 size_t partials_bytes = 0);

// ---------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------
void launch_fill_i32(int* p, int v, size_t n, hipStream_t st);
void launch_fill_f32(float* p, float v, size_t n, hipStream_t st);
void launch_scalar_seed(float* root, float* g0, float s0, int acc0, float* g1, float s1, int acc1, size_t n, hipStream_t st);
// out[i] = sa * a[i] + sb * b[i]  over n scalars held at arbitrary addresses
struct ScalarArgs {
  const GTNX_G float* a;
  const GTNX_G float* b; // may be nullptr
  GTNX_G float* out;
  GTNX_G float* mirror;  // pinned host memory that gets the value too, or nullptr (runtime.h: mirror_slot)
};
void launch_scalar_combine(const ScalarArgs* d_args, int n, float sa, float sb, hipStream_t st);
void launch_copy_small(void* dst, const void* src, size_t bytes, hipStream_t st);  // src: device or pinned host memory
void launch_scalar_combine_one(const ScalarArgs& a, float sa, float sb, hipStream_t st);  // the record as a kernel argument
// o0[i] = s0 * d[i], o1[i] = s1 * d[i] (o1 may be null); seed != null: d is 1 and is written to *seed
struct ScalarFanArgs {
  const GTNX_G float* d;
  GTNX_G float* seed;
  GTNX_G float* o0;
  GTNX_G float* o1;
};
void launch_scalar_fan(const ScalarFanArgs* d_args, int n, float s0, float s1, hipStream_t st);
void launch_scalar_fan_one(const ScalarFanArgs& a, float s0, float s1, hipStream_t st);
// dst[i] += src[i] for a batch of vectors (atomic when dst's may repeat)
struct AxpyArgs {
  GTNX_G float* dst;
  const GTNX_G float* src;
  int64_t n;
  float scale;
};
void launch_axpy_batch(const AxpyArgs* d_args, int n, int64_t maxn, int atomic, hipStream_t st);
// dst[0 .. bytes) = src[0 .. bytes) for a batch of segments (16-byte accesses where both ends allow it)
struct CopySeg {
  void* dst;
  const void* src;
  int64_t bytes;  // a multiple of 4
};
void launch_copy_segments(const CopySeg* d_segs, int n, int64_t max_bytes, hipStream_t st);
// gather n scalars at arbitrary addresses into a dense array
void launch_gather_scalars(const float* const* d_ptrs, float* out, int n, hipStream_t st);
// out[i] = (accumulate ? out[i] : 0) + sa * a[i] + sb * b[i]  (b may be null) -- dense vectors of a batch record
void launch_vec_axpby(float* out, const float* a, const float* b, size_t n, float sa, float sb, int accumulate, hipStream_t st);
// viterbiPath grad: grad[arcs[a]] += delta[a]
struct ScatterArgs {
  const GTNX_G int* idx;
  const GTNX_G float* delta;
  GTNX_G float* grad;
  int n;
};
void launch_scatter_add(const ScatterArgs* d_args, int n, int maxn, hipStream_t st);
// materialise a KIND_LINEAR graph's arc arrays
void launch_linear_materialize(int M, int C, int* src, int* dst, int* il, int* ol, hipStream_t st);

} // namespace gtnx
