// ops.h -- batched graph functions and the batch-level autograd tape.
//
// Every function takes vectors of graphs and runs ONE launch per kernel family
// for the whole vector (a single graph is a batch of one).  This replaces the
// reference's per-utterance parallelMap over CPU threads
// (gtn/parallel/parallel_map.h:153-188) with batch-of-graphs kernels.
//
// Autograd (gtn/autograd.cpp:17-67): each batched call creates one OpRecord with
// a global sequence number; outputs remember (record, index).  backward() walks
// the inputs() DAG from the roots, groups reachable outputs by record and runs
// the records in decreasing sequence order -- creation order is a topological
// order, so this is the reference's reverse tape sweep with one batched kernel
// per record instead of one gradFunc call per graph.
#pragma once

#include <cstring>
#include <tuple>
#include <vector>

#include "graph.h"

namespace gtnx {

struct Member {
  int idx;    // index inside the producing batch
  Graph out;  // the output graph (its g->grad is the incoming delta)
};

struct OpRecord {
  uint64_t seq = 0;
  virtual void backward(std::vector<Member>& members) = 0;
  // true: this record's backward may only REGISTER its launch with the running backward's chain-gradient
  // plan (ops.cpp: ChainGradPlan), so that the normaliser's softmax term and the sweep's posteriors
  // of the same emissions leave in one kernel
  virtual bool joins_chain_plan() const { return false; }
  virtual ~OpRecord() {}
};

// accumulates addGrad calls of one backward step and flushes them with at most
// one batched accumulate launch (first gradient of a graph is adopted in place)
struct GradSink {
  struct Item {
    Graph g;
    DevMemP owner;
    float* ptr;
  };
  std::vector<Item> items;
  void add(const Graph& g, const DevMemP& owner, float* ptr) {
    if (g.calc_grad()) items.push_back({g, owner, ptr});
  }
  void flush();
};

// band.hip launches are grouped by these (one kernel instantiation per group)
struct BandLaunchKey {
  int C, npl, unit, gradg, vec;
  bool operator<(const BandLaunchKey& o) const {
    return std::tie(C, npl, unit, gradg, vec) < std::tie(o.C, o.npl, o.unit, o.gradg, o.vec);
  }
  bool operator==(const BandLaunchKey& o) const { return !(*this < o) && !(o < *this); }
};
int band_vec_ok(const BandPair& p);
// prof_name: a profiled span (Runtime::Scope) around the kernel launches only.  table_out (forward launches of a batch
// record): the device table of a launch that was ONE group, kept for the record's backward (kernels.h: BandPatch)
void band_launch(std::vector<std::pair<BandLaunchKey, BandPair>>& tab, bool backward, const char* prof_name = nullptr,
                 double prof_bytes = 0.0, DevMemP* table_out = nullptr);
// the backward launch of a batch record over its forward launch's table: no upload
void band_launch_patched(const DevMemP& table, int n, const BandLaunchKey& key, int max_ns, const BandPatch& patch,
                         const char* prof_name, double prof_bytes);

enum ScalarKind { SK_NEGATE = 0, SK_ADD = 1, SK_SUBTRACT = 2 };

std::vector<Graph> op_scalar(ScalarKind k, std::vector<Graph>& a, std::vector<Graph>& b);
std::vector<Graph> op_shortest_distance(std::vector<Graph>& gs, bool tropical);
std::vector<Graph> op_viterbi_path(std::vector<Graph>& gs);
std::vector<Graph> op_compose(std::vector<Graph>& a, std::vector<Graph>& b, bool intersect);
// rational operations built on the device (rational.hip): clone / projections, concat, closure, union_
enum RationalKind { RAT_CLONE = 0, RAT_CONCAT = 1, RAT_CLOSURE = 2, RAT_UNION = 3 };
Graph op_rational(int kind, std::vector<Graph>& inputs, int projection);
// remove (functions.cpp:253-318) built on the device (rational.hip): node and arc ids, arc order and the start / accept
// lists as the reference's breadth-first construction gives them; weights dropped; no gradient
Graph op_remove(Graph& g, int ilabel, int olabel);
// a graph in the binary format of utils.cpp:152-225 (the whole file image) built on the device: one staging copy,
// the arc table split and the adjacency lists built by kernels (rational.hip)
Graph op_load_buffer(const void* data, size_t bytes);
void op_backward(std::vector<Graph>& roots, Graph* grad, bool retain, bool seed = true);
// throws what op_backward would throw before changing anything (a tape that is gone: autograd.cpp:42-45)
void backward_validate(Graph& root);
// per-thread hint of gtnx_compose_mode (include/gtn_amd.h); returns the previous value
int compose_mode_hint(int mode);
// build a symbolic (lazy) chain product for real, in place; no-op otherwise
void realize(Graph& g);

Graph make_scalar_graph(float v, bool calc_grad);
Graph make_linear_graph(int M, int N, bool calc_grad);
std::vector<Graph> make_linear_graphs_device(int B, int M, int N, bool calc_grad, const void* dev, bool borrow = false);
Graph make_user_op(std::vector<Graph>& inputs, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*));
void set_user_grad_fn(Graph& g, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*));

void items_host(std::vector<Graph>& gs, float* out);
void items_device(std::vector<Graph>& gs, void* dev_out);
void grads_device(std::vector<Graph>& gs, void* dev_out, const int64_t* offsets);
void grads_bind_device(std::vector<Graph>& gs, void* dev_out, const int64_t* offsets);

// A kernel's argument table (one record per graph of the launch) on the device.  GTNX_ARGS_IN_PLACE=<bytes>: a table
// of at most that many bytes is not copied -- the kernel reads it in place from the pinned host block (mapped into
// the device's address space; each workgroup reads its record once, through the scalar cache).  A single-utterance
// loss makes five such copies; measured on BASELINE config C1 the in-place form is worth nothing (0.157 against 0.159
// ms per loss: the read over the host link costs what the copy did), so it is OFF by default.
// The block goes back to the pinned pool behind an event on the stream, i.e. after the launch that reads it.
inline bool upload_in_place(size_t bytes) {
  static const size_t lim = [] {
    const char* e = std::getenv("GTNX_ARGS_IN_PLACE");
    return e ? size_t(std::atol(e)) : size_t(0);
  }();
  return bytes <= lim;
}
template <class T>
DevMemP upload_vec(const std::vector<T>& v) {
  Runtime& rt = Runtime::get();
  size_t bytes = sizeof(T) * v.size();
  if (bytes && upload_in_place(bytes)) {
    PinnedMemP p = rt.alloc_pinned(bytes);
    std::memcpy(p->ptr, v.data(), bytes);
    DevMemP d = std::make_shared<DevMem>();
    d->ptr = p->ptr;
    d->bytes = bytes;
    d->borrowed = true;
    d->keep = p;
    return d;
  }
  DevMemP d = rt.alloc(bytes ? bytes : 1);
  if (bytes) {
    PinnedMemP p = rt.alloc_pinned(bytes);
    std::memcpy(p->ptr, v.data(), bytes);
    rt.h2d_pinned(d->ptr, p->ptr, bytes);
  }
  return d;
}

} // namespace gtnx
