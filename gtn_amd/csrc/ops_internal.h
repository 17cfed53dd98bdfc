// ops_internal.h -- what the translation units of the op layer share with each other and with nobody else:
//   ops.cpp           creation, scalar ops, GradSink, the autograd tape (backward), items / gradients out, realize
//   ops_built.cpp     forwardScore / viterbiScore / viterbiPath of BUILT graphs (shortest.hip, linear chains)
//   ops_compose.cpp   compose / intersect: capacities, the FAST / general / wide-node kernels, symbolic products
//   ops_symbolic.cpp  which kernels score a symbolic chain product: THE ROUTE TABLE (symbolic_route)
//   ops_band.cpp      banded partners (CTC targets, force-alignment acceptors): band.hip
//   ops_lazy.cpp      record-walking / dense (MFMA) / max-plus / per-pair kernels: lazy.hip, maxplus.hip, lazy_pair.hip
//   ops_rational.cpp  rational operations, remove, the binary format as device builders: rational.hip
#pragma once

#include "ops.h"

#include "gtn/parallel.h"  // header-only worker pool (no engine dependency)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

namespace gtnx {

// ---- ops.cpp: records, result graphs, shared helpers
// Records are ordered by creation; sequence numbers are 2^20 apart so that a record made later
// to stand in for an older one (realize()) can be filed right behind it.
constexpr uint64_t kSeqStride = uint64_t(1) << 20;
extern std::atomic<uint64_t> g_seq_ctr;
extern std::atomic<uint64_t> g_seq_sub;
inline uint64_t next_seq() { return g_seq_ctr.fetch_add(1) * kSeqStride; }

void init_scalar_structure(Graph& g);
void init_scalar_result(Graph& g);
Graph make_output(const std::shared_ptr<OpRecord>& op, int idx, std::vector<Graph> inputs);
template <class T>
const T& bcast(const std::vector<T>& v, size_t n, size_t i) {
  if (v.size() == n) return v[i];
  if (v.size() == 1) return v[0];
  // parallel_map.h:85-88
  throw_runtime("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
}
void set_dev_weights(Graph& g, const DevMemP& owner, float* ptr, int64_t n);
void ensure_records(Structure& st);
const gtnx_i4* sorted_view(Structure& st, bool key_ol, bool in_lists);
float* grad_dev_ptr(Graph& out);
// The delta an op uses when it pushes its gradient PAST its input into that input's inputs (the input is a symbolic
// product: it has no gradient of its own).  In the reference the product's gradient accumulates over backward passes
// (backward(g, retainGraph = true) twice: the root holds 1, then 2; the product 1 P, then 3 P; its inputs get 4 x the
// single-pass gradient), so from the second pass on the multiplier is the SUM of the output's gradients over the
// passes, not the current one.  Single pass (every criterion step): grad_dev_ptr(out), nothing else happens.
float* through_delta(Graph& out);
extern thread_local bool t_backward_retain;  // the backward pass being run keeps the graph (op_backward)
// Set by a parallelMap region whose recorded calls end in a function that WAITS for the GPU (viterbiScore /
// viterbiPath read their results back): compose's reclamation point is skipped and what the caller let go of is
// taken apart while the host waits for the launch instead (Runtime::d2h_sync -> drain_while_busy) -- in the decode
// loop of the reference's API that was 1.2 ms per batch in front of the launch.
extern thread_local bool t_reclaim_at_wait;

// ---- gradient launches of one backward() over the same emission chains, gathered before they go:
// forwardScore(emissions) contributes dn * softmax(row), forwardScore(target o emissions) the node
// posteriors; registered here by their records and launched together (flush_chain_plan, ops_band.cpp) the
// band backward kernel writes every gradient row once, softmax term included.
struct ChainGradPlan {
  struct Lin {
    Member m;             // output of forwardScore(chain)
    Graph chain;          // (the tape forgets the inputs as soon as the record's backward returns)
    const float* delta;   // d / d norm
    const float* rowlse;  // per-row log2-sum-exp2 of the chain (NormCache)
    DevMemP keep;
    bool fused = false;
  };
  struct Band {
    int C, npl, unit, gradg, vec;
    BandPair p;
    Weights* chain_w;
  };
  std::unordered_map<Weights*, Lin> lin;  // by chain weights
  std::vector<Band> band;
  std::vector<DevMemP> keep;
  GradSink sink;
  double bytes = 0;
  bool empty() const { return lin.empty() && band.empty(); }
};
extern thread_local ChainGradPlan* t_chain_plan;
void flush_chain_plan();
// ops_built.cpp: the gradient of forwardScore(linear chains) launched now (members the plan did not fuse)
void linear_sd_backward_now(std::vector<Member>& ms, std::vector<Graph>& ins);

// ---- ops_built.cpp
void fill_path_graph(Graph& out, int len, bool has_node, const int* il, const int* ol, const float* w);

// ---- ops_compose.cpp
std::vector<Graph> op_compose_impl(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect, bool allow_lazy);

// ---- symbolic chain products (ops_symbolic.cpp is the table; the routes live in ops_band.cpp / ops_lazy.cpp)
enum SymbolicRoute : int {
  ROUTE_BAND = 0,      // banded partner: band.hip sweeps (forward / backward / Viterbi)
  ROUTE_PAIR,          // small partner, one workgroup per pair: lazy_pair.hip (log semiring only)
  ROUTE_DENSE_MFMA,    // dense partner in the probability domain on the matrix cores: lazy.hip
  ROUTE_DENSE,         // dense partner, VALU form
  ROUTE_MAXPLUS,       // tropical semiring over a dense partner: maxplus.hip
  ROUTE_WALK,          // record-walking sweeps: lazy.hip
  ROUTE_COUNT
};
const char* symbolic_route_name(int r);
// the route forwardScore (tropical: viterbiScore / viterbiPath) takes for this product
SymbolicRoute symbolic_route(const LazyProduct& lp, bool tropical);
std::vector<Graph> lazy_shortest_distance(std::vector<Graph>& gs, bool tropical);

std::shared_ptr<OpRecord> make_lazy_compose_op();
bool lazy_shape_ok(const Structure& chain, const Structure& fixed);
bool lazy_pair_shape_ok(const Structure& chain, Structure& fixed);
bool lazy_pair_ok(const LazyProduct& lp);
std::vector<Graph> lazy_pair_forward_score(std::vector<Graph>& gs);
std::vector<Graph> lazy_group_shortest_distance(std::vector<Graph>& gs, bool tropical);
std::vector<Graph> lazy_viterbi_path(std::vector<Graph>& gs);
// exact ties on a best path reported by the max-plus walk / of those, left in in-row order (product too large to build)
extern std::atomic<int64_t> g_viterbi_ties_seen, g_viterbi_ties_unresolved;
// which dense form lazy_group_shortest_distance would pick for this product (ROUTE_DENSE_MFMA / ROUTE_DENSE /
// ROUTE_MAXPLUS), or ROUTE_WALK
SymbolicRoute lazy_group_route(const LazyProduct& lp, bool tropical);
bool lazy_dense_ok(Structure& fs, bool chain_first, int C, std::shared_ptr<Structure::DenseInfo>* di_out);
struct LazyPathOp : OpRecord {
  struct Saved {
    std::vector<int> arcs, il, ol;  // first-arc-first
    int C = 0, chain_first = 0;
  };
  std::vector<Saved> saved;
  void backward(std::vector<Member>& ms) override;
};

bool band_shape_ok(const Structure& chain, Structure& fixed, bool chain_first);
void band_prepare(const std::vector<Graph*>& fixed, const std::vector<uint8_t>& chain_first, bool want_ranks = false);
bool band_ok(const LazyProduct& lp, std::shared_ptr<BandInfo>* out = nullptr);
std::vector<Graph> band_forward_score(std::vector<Graph>& gs);
std::vector<Graph> band_viterbi(std::vector<Graph>& gs, bool want_path);

} // namespace gtnx
