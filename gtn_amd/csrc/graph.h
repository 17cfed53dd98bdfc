// graph.h -- host-side graph value handle of the engine.
//
// Mirrors the three shared pieces of the reference's gtn::Graph
// (gtn/graph.h:439-464: SharedGraph / weights / SharedGrad) so copies alias and
// deepCopy detaches exactly as there, but each piece is dual-resident:
//   Structure : host SoA mirror (for graphs built with addNode/addArc or when a
//               device-built graph is inspected) + SoA/CSR buffers in HBM
//   Weights   : host vector + device buffer (possibly a slice of a batch arena)
//   GradState : autograd tape node (producing op, inputs, grad graph)
#pragma once

#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "runtime.h"

namespace gtnx {

struct Graph;
struct OpRecord;

// cached level schedule for shortest distance (see kernels.h: DSched)
struct Schedule {
  bool error = false;  // an accept node is never reached: cycle / self-loop / disconnected
  DevMemP mem;
  DSched view{};
  int max_level_width = 0;
  int max_level_arcs = 0;  // most in-arcs entering one level
  int max_reach = 0;       // max over levels of (last position + 1 - min source position)
  int64_t n_in = 0, n_out = 0;
  bool all_written = false;  // every arc gets a gradient (no memset needed)
  bool has_rank = false;
  // view.in_w (weights permuted into row order, emitted by compose) is only
  // valid for this weights object at this version
  const void* in_w_of = nullptr;
  uint64_t in_w_version = 0;
  const float* in_w = nullptr;
  // Set by compose for a layered product with ONE implicit linear chain (level ==
  // chain time step): lets forwardScore's backward push the arc gradients straight
  // into the two compose inputs (see SdOp::backward / sd_backward_narrow_kernel).
  uint64_t producer_seq = 0;         // seq of the compose record that emitted the structure
  int chain_side = 0;                // 0: not eligible, 1 / 2: which compose input is the chain
  int chain_C = 0;                   // chain arcs per time step
  int64_t fixed_A = 0;               // arcs of the other input
  const int* gi_fixed = nullptr;     // gradInfo column of the explicit input
  const int* gi_chain = nullptr;     // gradInfo column of the chain
  // the producing compose's result block on the device (sizes for the narrow kernels)
  const struct ComposeOut* dyn_out = nullptr;
  const int* dyn_counts = nullptr;
};

struct Structure {
  // Non-null: the RESULT of a graph function called inside a parallelMap region that has not run yet, or
  // has run as part of a batch record (region.cpp).  Nothing else in this structure is valid; every access
  // through the C ABI goes to region_value() first.
  struct Pending* pending = nullptr;  // (lives in the same allocation: region.cpp PlaceholderStructure)
  std::atomic<int> pending_uses{0};  // as Weights::pending_uses
  // the deferred-reclamation list of the thread that made the graph: a handle that is destroyed elsewhere is sent
  // there to be taken apart (runtime.h: every thread frees what it allocated)
  Runtime::InboxP home;
  // the device the graph was made on (the calling thread's at that moment, runtime.h); -1: none (placeholders)
  int device = -1;
  // The graph is exactly the CTC target acceptor of benchmarks/ctc.cpp:40-58 over these labels (checked at
  // arcSort, O(A)): a batch of such graphs takes the device-built band records (batch.cpp: CTC_TARGETS)
  std::shared_ptr<std::vector<int>> ctc_labels;
  std::weak_ptr<struct Batch> leaf_batch;  // the CTC_TARGETS record this graph is an element of (as Weights::leaf_batch)
  int ctc_blank = 0;
  bool ctc_checked = false;  // detect_ctc_shape ran on the current arcs
  int kind = KIND_EXPLICIT;
  int64_t N = 0, A = 0;
  int M = 0, C = 0;  // KIND_LINEAR
  bool ilabel_sorted = false, olabel_sorted = false;

  // ---- host mirror (arc-id order SoA + lazily built CSR in reference list order)
  bool host_valid = true;
  std::vector<int> src, dst, il, ol;
  std::vector<uint8_t> nflags;
  std::vector<int> start, accept;
  bool csr_valid = false;
  int sort_pending = 0;  // arcSort asked for before the lists existed: 1 by ilabel, 2 by olabel (applied by ensure_csr)
  std::vector<int> in_off, in_list, out_off, out_list;

  // ---- device mirror
  bool dev_valid = false;
  DevMemP dev_mem;  // owner (may be shared by a whole batch)
  DGraph dview{};   // pointers into dev_mem (w unset)
  DevMemP rec_mem;  // packed adjacency records of a device-built structure (made on demand)
  // stable label-sorted copies of the adjacency records ([0] by ilabel, [1] by olabel; out records then in
  // records), made on demand for compose_pairs_kernel when this graph is searched without being sorted
  DevMemP sview_mem[2];
  const void* sview_of[2] = {nullptr, nullptr};  // the out_rec array the view was made from

  std::shared_ptr<Schedule> sched;  // valid while the structure is unchanged
  std::mutex grad_lock;             // graph.h:450

  // Non-null: a composition whose sizes (N, A, levels, start / accept counts) are still
  // on the device -- the compose that produced it did not wait for them (DeferredSizes).
  // N and A read -1 meanwhile; capN / capA bound them.  resolve_sizes() waits and fills in.
  std::shared_ptr<struct DeferredSizes> deferred;
  int deferred_idx = 0;
  int64_t capN = 0, capA = 0;
  void resolve_sizes();
  int64_t bound_nodes() const { return deferred ? capN : N; }
  int64_t bound_arcs() const { return deferred ? capA : A; }

  // Non-null: a device-built composition whose derivable arrays (src, il, ol, in_list)
  // have not been written yet (kernels.h: ComposeArgs::skip).  ensure_full() writes them;
  // every path that hands the arrays to a kernel or to the host goes through it.
  std::shared_ptr<struct PartialInfo> partial;
  void ensure_full();

  // Non-null: this is a composition that has NOT been built (ops.cpp: lazy chain
  // products).  Nothing else in the structure is valid until realize() fills it in.
  std::shared_ptr<struct LazyProduct> lazy;

  // Cached band records (kernels.h: BandNode) per matched-label side (0: ilabel, 1: olabel);
  // made by band_info() on first use as the fixed side of a symbolic chain product
  std::shared_ptr<struct BandInfo> band[2];

  // Cached facts for the dense regime of the never-built products (ops.cpp: lazy_forward), per matched-label
  // side (0: ilabel, 1: olabel) and alphabet size: the shared in-arc label of every node (empty: they differ),
  // the count of matchable arcs, and the device tables of maxplus.hip
  struct DenseInfo {
    int C = -1;
    std::vector<int> lab;
    int max_in_deg = 0;
    int64_t valid = 0;
    int ncol = 0, ndead = 0;
    bool uniq = false;  // no two nodes share a label
    // every node's out-list is exactly the nodes that have in-arcs, in increasing node order, and the accept list
    // increases too (an ASG transitions graph as examples/asg.cpp builds it): under exact ties the reference's
    // queue then visits the sources of every product node in node order (ops_lazy.cpp: dense_ties_by_node_order)
    bool ties_by_node_order = false;
    DevMemP tables;  // [N] labels | [N] node -> column | [ncol] column -> node | [ndead] dead nodes
  };
  std::shared_ptr<DenseInfo> dense[2];

  int max_deg = -1;     // widest in- or out-row (cached; touch() forgets it)
  int max_degree();

  void materialize();  // LINEAR -> EXPLICIT host arrays
  void ensure_host();  // download a device-built structure
  void ensure_csr();
  void touch();        // structural mutation: drop device mirror, schedule, sort flags
  int num_in(int n);
  int num_out(int n);
};

// forwardScore of a linear chain over these weights, left behind by a sweep that read every
// emission anyway (band.hip): the scalar and the per-row log2-sum-exp2 (softmax term of the gradient)
struct NormCache {
  uint64_t version = 0;  // of the weights it was computed from
  DevMemP mem;
  float* norm = nullptr;
  float* rowlse = nullptr;
};

// Weights handed over inside a parallelMap region (region.cpp): the values sit in a pinned staging
// chunk (host source, copied at the call like graph.cpp:179-181) or are still the caller's device buffer
// (device source: read at the region's join); the join moves the whole region's weights with one copy.
// The device image of the device-source weights ONE thread staged inside a region (region.cpp: a slice): the
// join allocates one arena for all slices, copies every segment with one launch and publishes each slice's
// base address here; the weights themselves are not touched by the join -- they settle (Weights::settle_staged)
// when somebody first needs them.
// The copy a region's join owes the emission graphs whose weights were set from DEVICE memory inside the region
// (graph.cpp:179-181: setWeights copies).  It is not launched at once: when the region's first consumer of those
// weights is the band forward sweep -- the criterion step -- the sweep stores every emission it stages (BandPair::
// em_copy) and the tensor is read once instead of twice; anything else that wants the values (Weights::settle_staged,
// the other consumers of a LINEAR record, the end of the join at the latest: the caller's buffer is only promised
// until parallelMap returns) launches the copy kernel first.
struct PendingCopy {
  std::vector<CopySeg> segs;  // destination ascending
  int64_t max_bytes = 0;
  int device = 0;
  std::atomic<bool> done{false};
  std::mutex mu;
  void settle();                              // launch copy_segments unless the copy has been made
  const void* src_of(const void* dst) const;  // the caller's address whose copy goes to dst (null: none)
};
struct StageBlock {
  DevMemP mem;
  std::atomic<float*> base{nullptr};
  std::shared_ptr<PendingCopy> pend;  // set with `base`
};
struct StagedWeights {
  PinnedMemP chunk;            // keeps the staging chunk alive (host source)
  const float* src = nullptr;  // pinned host address or the caller's device address
  bool on_device = false;
  std::shared_ptr<StageBlock> blk;  // device source: where the join's copy lands ...
  size_t off = 0;                   // ... at this byte offset
};

struct Weights {
  int64_t n = 0;
  std::vector<float> host;
  bool host_valid = true;
  // n zeros that nobody has stored yet (linearGraph before setWeights, creations.cpp:20-33): neither copy is
  // valid, ensure_host() / ensure_weights_device_batch() write them out on first use
  bool zero = false;
  std::shared_ptr<StagedWeights> staged;  // set: neither copy is valid yet (see StagedWeights)
  // queued calls of a parallelMap region that will read these weights (region.cpp): a mutation flushes them first
  std::atomic<int> pending_uses{0};
  // the LINEAR batch record these weights are an element of (batch.cpp: batch_linear_from_graphs), at `leaf_version`:
  // a later call over the same graphs finds the record -- and what a sweep left behind in it -- again
  std::weak_ptr<struct Batch> leaf_batch;
  uint64_t leaf_version = 0;
  bool host_escaped = false;  // a mutable host pointer was handed out (Graph::weights())
  uint64_t version = 0;       // bumped on every mutation
  uint64_t zero_version = ~uint64_t(0);  // version all_zero was taken at
  bool all_zero = false;
  bool is_all_zero();         // host-valid weights only; cached per version
  // is_all_zero() if that is known without computing anything (what a thread may ask of weights it shares)
  bool known_all_zero() const { return zero || (host_valid && !host_escaped && zero_version == version && all_zero); }
  // staged from a device source and the region's join has made the copy: adopt it (true: dev is valid now)
  bool settle_staged();
  std::shared_ptr<NormCache> norm_cache;
  // (not while a mutable host pointer is out and the host copy is the live one: writes through it do not bump
  //  `version` -- the cache is of the last upload, and so may be stale)
  const NormCache* valid_norm_cache() const {
    if (host_escaped && host_valid) return nullptr;
    return norm_cache && norm_cache->version == version && dev_valid ? norm_cache.get() : nullptr;
  }
  DevMemP dev_mem;
  float* dev = nullptr;
  bool dev_valid = false;
  // the kernel that produced this ONE value wrote it to pinned host memory as well (runtime.h: mirror_slot), at
  // `mirror_version`: ensure_host() waits for the stream and reads it there instead of copying
  Runtime::MirrorSlot mirror;
  uint64_t mirror_version = 0;
  void ensure_host();
};

// Is G banded (arcs n -> n, n+1, n+2, one per step, one matched label per node)?  Then the
// sweeps of band.hip apply and this is all of G the device ever sees (16 bytes per node).
struct BandInfo {
  bool ok = false;
  bool unit_shape = false;  // self-loop at every node and an arc from the previous node at every node but 0
  int hot = -1;             // label carried by >= 8 nodes (CTC: blank)
  int max_label = -1;
  std::vector<BandNode> nodes;
  std::vector<int> snode, slab;  // nodes with an in-arc sorted by (label, node), and their labels
  DevMemP dev_mem;          // one arena per uploaded batch
  const BandNode* dev = nullptr;
  const uint8_t* dev_flags = nullptr;
  const int* dev_snode = nullptr;
  const int* dev_slab = nullptr;
  // ops_band.cpp tie_ranks (CTC-shaped graphs): 0 not computed, 1 there, -1 does not apply
  int rank_state = 0;
  std::vector<int> rank_kahn, rank_create;
};
std::shared_ptr<BandInfo> band_info(Structure& s, bool use_ilabel);   // host part, cached
void detect_ctc_shape(Structure& s);                                  // fills Structure::ctc_labels; cached
// ops_band.cpp: the reference's queue order and creation order of a CTC-shaped target's nodes (how exact ties of its
// products are decided without the lattice); false: does not apply to this graph.  Cached in the graph's BandInfo.
bool ctc_tie_ranks(Structure& s, bool use_ilabel, const std::vector<int>** kahn, const std::vector<int>** create);
void ensure_band_device_batch(const std::vector<BandInfo*>& bs, const std::vector<Structure*>& ss);

struct GradState;

// The three pieces of a graph (structure 824 bytes, weights, gradient state) for MANY graphs out of ONE buffer:
// while a scope is alive on the calling thread, Graph::make_result -- the RESULTS of an op, the elements of a batch
// record -- carves its pieces (and their reference counts) out of the scope's buffer instead of three heap blocks of
// its own.  A vector function's n results are 3 n blocks of ~1 KB otherwise -- beyond the allocator's per-thread
// caches, so each is a trip through its bins both ways, and their memory is cold by the time it comes round again
// (tools/nullhip/small_step c2).  Every piece keeps a lifetime of its own (it is destroyed when its last reference
// goes); the buffer goes back to a per-thread cache when the last piece carved out of it has died.  A scope that
// runs out falls back to the heap; scopes nest (each has its own buffer).  The price: ONE result kept alive keeps
// its whole buffer (n x 1.3 KB).  n < 8: no buffer.
struct GraphSlabScope {
  explicit GraphSlabScope(size_t n);
  ~GraphSlabScope();
  GraphSlabScope(const GraphSlabScope&) = delete;
  GraphSlabScope& operator=(const GraphSlabScope&) = delete;

 private:
  bool active_ = false;
  void* prev_ = nullptr;  // the enclosing scope's buffer, shelved
};

struct Graph {
  std::shared_ptr<Structure> s;
  std::shared_ptr<Weights> w;
  std::shared_ptr<GradState> g;

  explicit Graph(bool calc_grad = true);
  struct Empty {};
  explicit Graph(Empty) {}  // no pieces at all (region.cpp: placeholders carry a structure only)
  Graph(bool calc_grad, std::shared_ptr<Structure> shared);  // fresh weights / grad state over an existing structure
  static Graph make_result(bool calc_grad);  // fresh pieces, for op outputs

  // graph.cpp:33-67
  int add_node(bool start, bool accept);
  int add_arc(int src, int dst, int il, int ol, float w);
  // bulk forms (one reservation, one invalidation): flags NF_START | NF_ACCEPT per node; w may be null (zeros)
  void add_nodes(int n, const uint8_t* start, const uint8_t* accept);
  void add_arcs(int n, const int* src, const int* dst, const int* il, const int* ol, const float* w);
  int64_t num_nodes() const { return s->N; }
  int64_t num_arcs() const { return s->A; }
  int64_t num_start();
  int64_t num_accept();
  float item();
  void arc_sort(bool olabel);
  static Graph deep_copy(const Graph& src);
  const float* weights_host(bool mut);
  void set_weights_host(const float* p);
  void set_weights_device(const void* p);
  inline bool calc_grad() const;
  inline bool is_grad_available() const;
  Graph& grad();
  void set_calc_grad(bool c);
  inline void zero_grad();
  uintptr_t id() const { return reinterpret_cast<uintptr_t>(g.get()); }

  // addGrad (graph.cpp:91-129).  `owner`/`dev` is a device vector of numArcs
  // floats; when `adopt` the buffer becomes the grad without a copy.
  void add_grad_host(const float* v, int64_t n);
  void add_grad_device(const DevMemP& owner, float* dev, bool adopt);
  // A FIRST gradient adopted in place: only the buffer is noted (GradState::lazy_*); the gradient graph is
  // built when somebody asks for it (grad(), an accumulation).  A criterion step hands out two gradients per
  // utterance that the next step usually throws away unread.
  void materialize_grad();
};

struct GradState {
  bool calc_grad = true;
  std::shared_ptr<OpRecord> op;  // producing op; nullptr for leaves
  int op_idx = 0;
  bool has_grad_fn = false;      // mirrors `gradFunc != nullptr`
  std::vector<Graph> inputs;
  std::unique_ptr<Graph> grad;
  DevMemP lazy_owner;            // a first gradient that has no graph yet (Graph::materialize_grad)
  float* lazy_ptr = nullptr;
  std::atomic<int> n_consumers{0};  // op outputs that list this graph as an input (two threads may reclaim at once)
  bool grad_propagated = false;  // the consumer already pushed this grad into our inputs
  // ops.cpp through_delta: the sum of this graph's gradient over the backward passes made with retainGraph -- what the
  // gradient of a SYMBOLIC product feeding it would have accumulated to (autograd.cpp:40-52 adds to it in every pass)
  DevMemP through_acc;
  // gtnx_grads_bind_device_n: caller-owned device memory the FIRST gradient of this graph is to be
  // written to (the emission-gradient tensor of a criterion); kernels that can store there directly do
  DevMemP grad_dest_mem;
  float* grad_dest = nullptr;
  ~GradState();                  // gives the consumer counts of `inputs` back
};
inline bool Graph::calc_grad() const { return g->calc_grad; }
inline bool Graph::is_grad_available() const { return g->grad != nullptr || g->lazy_ptr != nullptr; }
inline void Graph::zero_grad() {
  g->grad.reset();
  g->lazy_owner.reset();
  g->lazy_ptr = nullptr;
}

// Sizes of one compose batch left on the device.  For a chain product whose partner is
// epsilon-free, no wider than a workgroup, with at most KC out-arcs per node, every
// capacity and every fast-path condition of compose_kernel is provable on the host, so
// nothing the host does next depends on the actual sizes: the CTC / ASG loss path
// (forwardScore, its backward) sizes its buffers by the bounds and the kernels read N, L
// and the accept count from the ComposeOut block on the device.  The header is copied
// to pinned memory behind the kernel; whoever needs real numbers first (inspection,
// another op) waits on the event -- see resolve().
struct DeferredSizes {
  hipEvent_t ev = nullptr;
  PinnedMemP host;
  size_t hdr_out = 0, hdr_cnt = 0;
  std::atomic<bool> done{false};
  std::mutex mu;  // resolve(): the thread that finds `done` set must also find the sizes applied
  struct Member {
    std::weak_ptr<Structure> s;
    std::weak_ptr<Weights> w;
  };
  std::vector<Member> members;
  struct GradW {
    std::weak_ptr<Weights> w;
    int member;
  };
  std::vector<GradW> grads;        // gradient graphs adopted meanwhile: their n is A
  struct ProfFix {
    std::string name;
    double per_arc, per_node;
    int member;
  };
  std::vector<ProfFix> prof;       // algorithmic bytes owed to the profiler
  void resolve();
  ~DeferredSizes();
};
// compose batches still unresolved, oldest first; the run-ahead of the host is capped
void deferred_register(const std::shared_ptr<DeferredSizes>& d);
void deferred_resolve_all();
void deferred_limit(size_t keep);
void apply_compose_sizes(Structure& s, Weights* w, const ComposeOut& co, int n_start, int n_accept);

struct PartialInfo {
  ComposeFillArgs args{};                // pointers into the product's own arena
  std::shared_ptr<Structure> in1, in2;   // keep the inputs' label arrays alive
  DevMemP keep1, keep2;
};

// compose(chain, fixed) / compose(fixed, chain) kept symbolic (see Structure::lazy)
struct LazyProduct {
  Graph chain, fixed;
  int chain_side;  // 1: chain is the first compose argument, 2: the second
  bool intersect;
};

// ---- batched residency helpers (ONE staging copy for a whole batch)
void ensure_device_batch(const std::vector<Structure*>& ss);
void ensure_weights_device_batch(const std::vector<Weights*>& ws);
void ensure_schedule_batch(const std::vector<Structure*>& ss, bool need_rank);
DGraph device_view(Graph& g);  // structure view + weight pointer (after the ensure_* calls)

bool graphs_equal(Graph& a, Graph& b);       // gtn/utils.cpp:45-77
bool graphs_isomorphic(Graph& a, Graph& b);  // gtn/utils.cpp:79-150

} // namespace gtnx
