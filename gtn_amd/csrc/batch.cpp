// batch.cpp -- see batch.h.  Reference semantics per element: functions.cpp:18-64 (scalar ops),
// :225-251 (compose / intersect), :320-326 (forwardScore) over creations.cpp:20-33 chains and the
// target acceptor of benchmarks/ctc.cpp:40-58; autograd.cpp:17-67 for backward.
#include "batch.h"

#include <algorithm>
#include <atomic>
#include <unordered_map>
#include <unordered_set>

namespace gtnx {

namespace {

std::atomic<uint64_t> g_batch_seq{1};

struct CompMode {  // compositions made while materialising stay symbolic until somebody looks inside
  int old;
  CompMode() : old(compose_mode_hint(2)) {}
  ~CompMode() { compose_mode_hint(old); }
};

BatchP make_batch(Batch::Kind k, int n, bool cg) {
  auto b = std::make_shared<Batch>();
  b->kind = k;
  b->n = n;
  b->calc_grad = cg;
  return b;
}

BatchP result(Batch::Kind k, int n, std::shared_ptr<BatchOp> op) {
  bool cg = false;
  for (auto& i : op->inputs) cg |= i->calc_grad;
  BatchP r = make_batch(k, n, cg);
  op->seq = g_batch_seq.fetch_add(1);
  r->op = std::move(op);  // kept without calc_grad too: it says how to rebuild the elements as graphs
  return r;
}

bool native(const Batch& b, Batch::Kind k) { return b.kind == k; }

// ---- the gradient array of a batch: bound destination, else a fresh block
void alloc_grad(Batch& b, bool zero) {
  Runtime& rt = Runtime::get();
  b.g_off.resize(size_t(b.n) + 1);
  b.g_off[0] = 0;
  for (int i = 0; i < b.n; ++i) b.g_off[size_t(i) + 1] = b.g_off[size_t(i)] + b.elem_size(i);
  const size_t bytes = sizeof(float) * size_t(b.g_off[size_t(b.n)]);
  if (b.dest) {
    b.g_mem = b.dest_mem;
    b.g_dev = b.dest;
    b.dest = nullptr;
    if (zero && bytes) HIP_CHECK(hipMemsetAsync(b.g_dev, 0, bytes, rt.stream()));
  } else {
    b.g_mem = zero ? rt.alloc_zero(bytes ? bytes : 4) : rt.alloc(bytes ? bytes : 4);
    b.g_dev = b.g_mem->as<float>();
  }
}

// a block the kernels may overwrite: the batch's own gradient when it has none yet, else a scratch
// block that add_scratch() folds in afterwards (addGrad semantics, graph.cpp:108-129)
struct GradTarget {
  float* ptr = nullptr;
  DevMemP scratch;
};
GradTarget grad_target(Batch& b, bool zero) {
  GradTarget t;
  if (!b.g_dev) {
    alloc_grad(b, zero);
    t.ptr = b.g_dev;
    return t;
  }
  Runtime& rt = Runtime::get();
  const size_t bytes = sizeof(float) * size_t(b.g_off[size_t(b.n)]);
  t.scratch = zero ? rt.alloc_zero(bytes ? bytes : 4) : rt.alloc(bytes ? bytes : 4);
  t.ptr = t.scratch->as<float>();
  return t;
}
void add_scratch(Batch& b, const GradTarget& t) {
  if (!t.scratch) return;
  launch_vec_axpby(b.g_dev, t.ptr, nullptr, size_t(b.g_off[size_t(b.n)]), 1.0f, 0.0f, 1, Runtime::get().stream());
}

// ---- ops -------------------------------------------------------------------------------------
struct BScalarOp : BatchOp {
  ScalarKind kind;
  void backward(Batch& out) override {
    Runtime& rt = Runtime::get();
    for (size_t i = 0; i < inputs.size(); ++i) {
      Batch& in = *inputs[i];
      if (!in.calc_grad) continue;  // subtract only feeds input 1 when it wants a gradient (functions.cpp:55-57)
      const float s = (kind == SK_NEGATE || (kind == SK_SUBTRACT && i == 1)) ? -1.0f : 1.0f;
      const bool have = in.g_dev != nullptr;
      if (!have) alloc_grad(in, false);
      launch_vec_axpby(in.g_dev, out.g_dev, nullptr, size_t(in.n), s, 0.0f, have ? 1 : 0, rt.stream());
    }
  }
};

// launches registered by one backward() over the same chains, gathered so that the normaliser's
// softmax term and the sweep's posteriors leave in one kernel (ops.cpp: ChainGradPlan, per batch)
struct BackwardPlan {
  std::unordered_map<Batch*, Batch*> lin;  // chain batch -> output of forwardScore(chain) waiting for a sweep
  std::unordered_set<Batch*> fused;
};
thread_local BackwardPlan* t_plan = nullptr;
thread_local bool t_batch_retain = false;  // the batch backward being run keeps the tape (batch_backward)

struct BFsLinearOp : BatchOp {
  void backward(Batch& out) override {
    Batch& e = *inputs[0];
    if (!e.calc_grad) return;
    if (t_plan && e.nc_rowlse && !t_plan->lin.count(&e)) {
      t_plan->lin[&e] = &out;  // rides with the sweep over the same chains, if one comes
      return;
    }
    run(out);
  }
  void run(Batch& out) {
    Batch& e = *inputs[0];
    Runtime& rt = Runtime::get();
    if (e.w_pend) e.w_pend->settle();  // (the values are read here: graph.h PendingCopy)
    const bool have = e.g_dev != nullptr;
    if (!have) alloc_grad(e, false);
    std::vector<LinArgs> args;
    args.resize(size_t(e.n));
    const size_t A = size_t(e.M) * size_t(e.C);
    for (int b = 0; b < e.n; ++b) {
      LinArgs& a = args[size_t(b)];
      a.w = e.w_dev + size_t(b) * A;
      a.M = e.M;
      a.C = e.C;
      a.out_score = nullptr;
      a.partial = nullptr;
      a.delta = out.g_dev + b;
      a.grad = e.g_dev + size_t(b) * A;
      a.accumulate = have ? 1 : 0;
    }
    DevMemP d = upload_vec(args);
    const bool vec_rows = e.C % 4 == 0 && e.C <= 1024 && (reinterpret_cast<uintptr_t>(e.w_dev) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(e.g_dev) & 15) == 0;
    GTNX_PROF("linear_forward_grad", (have ? 12.0 : 8.0) * double(A) * e.n);
    launch_linear_backward(d->as<LinArgs>(), e.n, 0, vec_rows ? 1 : 0, rt.stream());
  }
};

struct BFsBandOp : BatchOp {  // inputs[0]: the PRODUCT
  std::vector<BandPair> pairs;
  DevMemP arena;
  DevMemP fwd_table;  // the forward launch's device table (one launch group, read in place): the backward launch reads
                      // it again, the pointers that differ travel with the launch (kernels.h: BandPatch)
  // The gradient this op pushes PAST the symbolic product into its inputs.  In the reference the product is a graph
  // with a gradient of its own, which accumulates over backward passes and is re-scattered whole by compose's gradient
  // function in every pass (autograd.cpp:40-52, compose.cpp:496-518): over a retained tape the inputs receive, in
  // pass k, the SUM over passes 1..k of this output's (accumulated) gradient times the posteriors -- 1, 1+3, 1+3+6 ...
  // when the root is backward()ed three times.  Single pass (every criterion step): the output's gradient itself, and
  // nothing is allocated.  (ops.cpp: through_delta is the per-graph form; until round 6 the batch form used the
  // current gradient -- 1, 3, 6 -- found by tools/double_backward_fit.py against the unmodified reference.)
  DevMemP through_acc;
  void backward(Batch& out) override {
    Batch& prod = *inputs[0];
    Batch& fx = *prod.fixed;
    Batch& ch = *prod.chain;
    Runtime& rt = Runtime::get();
    const float* delta = out.g_dev;
    if (through_acc || t_batch_retain) {
      const size_t n = size_t(prod.n);
      if (!through_acc) through_acc = rt.alloc_zero(sizeof(float) * (n ? n : 1));
      launch_vec_axpby(through_acc->as<float>(), out.g_dev, nullptr, n, 1.0f, 0.0f, /*accumulate=*/1, rt.stream());
      delta = through_acc->as<float>();
    }
    GradTarget ge, gf;
    if (ch.calc_grad) ge = grad_target(ch, false);  // every row is written by the kernel
    // (device-built CTC targets: every arc lies in the band and the sweep writes them all, zeros included -- no fill;
    //  force-alignment acceptors keep theirs)
    if (fx.calc_grad) gf = grad_target(fx, fx.fal);
    Batch* lin_out = nullptr;
    if (t_plan && ch.calc_grad && !ge.scratch) {
      auto it = t_plan->lin.find(&ch);
      if (it != t_plan->lin.end() && !t_plan->fused.count(&ch)) {
        lin_out = it->second;
        t_plan->fused.insert(&ch);
      }
    }
    std::vector<std::pair<BandLaunchKey, BandPair>> tab;
    tab.reserve(pairs.size());
    double bytes = 0;
    const size_t A = size_t(ch.M) * size_t(ch.C);
    bool one_key = true;
    int max_ns = 0;
    for (int b = 0; b < prod.n; ++b) {
      BandPair p = pairs[size_t(b)];
      p.delta = delta + b;
      p.delta_norm = lin_out ? lin_out->g_dev + b : nullptr;
      p.rowlse = lin_out ? ch.nc_rowlse + size_t(b) * size_t(ch.M) : nullptr;
      p.norm = nullptr;
      p.grad_em = ge.ptr ? ge.ptr + size_t(b) * A : nullptr;
      p.grad_fixed = gf.ptr ? gf.ptr + fx.g_off[size_t(b)] : nullptr;
      tab.push_back({BandLaunchKey{p.C, band_npl(p.N), fx.fal ? 0 : 1, p.grad_fixed ? 1 : 0, band_vec_ok(p)}, p});
      one_key = one_key && tab.back().first == tab.front().first;
      max_ns = std::max(max_ns, p.NS);
      bytes += 4.0 * p.T * p.C * (p.grad_em ? 2 : 1) + 4.0 * double(p.T + 1) * p.NS +
               (p.grad_fixed ? 4.0 * double(fx.elem_size(b)) : 0.0);
    }
    static const bool no_patch = std::getenv("GTNX_NO_BAND_PATCH") != nullptr;
    if (fwd_table && one_key && prod.n > 1 && !no_patch) {
      // every field the loop above changed is a base + the pair's index (or the offset the table carries): no upload
      BandPatch pt{};
      pt.on = 1;
      pt.M = ch.M;
      pt.delta = delta;
      pt.delta_norm = lin_out ? lin_out->g_dev : nullptr;
      pt.rowlse = lin_out ? ch.nc_rowlse : nullptr;
      pt.grad_em = ge.ptr;
      pt.grad_fixed = gf.ptr;
      pt.A = int64_t(A);
      band_launch_patched(fwd_table, prod.n, tab.front().first, max_ns, pt, "band_forward_score_grad", bytes);
    } else {
      band_launch(tab, true, "band_forward_score_grad", bytes);
    }
    if (ch.calc_grad) add_scratch(ch, ge);
    if (fx.calc_grad) add_scratch(fx, gf);
    (void)rt;
  }
};


// the force-alignment acceptors were cut out of the transitions graph: their arc gradients go back into it
struct BFalOp : BatchOp {
  void backward(Batch& out) override {
    Graph& tr = out.trans;
    if (!tr.calc_grad() || !out.g_dev) return;
    Runtime& rt = Runtime::get();
    const int64_t A = tr.num_arcs();
    DevMemP gm = rt.alloc_zero(sizeof(float) * size_t(A ? A : 1));
    // one launch for the whole batch: {gradient offset, arc-map offset (ints), arcs} per sequence
    std::vector<int64_t> tab(size_t(out.n) * 3);
    int64_t longest = 0;
    for (int b = 0; b < out.n; ++b) {
      const int64_t len = out.g_off[size_t(b) + 1] - out.g_off[size_t(b)];
      tab[size_t(b) * 3 + 0] = out.g_off[size_t(b)];
      tab[size_t(b) * 3 + 1] = int64_t(out.map_off[size_t(b)] / sizeof(int));
      tab[size_t(b) * 3 + 2] = len;
      longest = std::max(longest, len);
    }
    DevMemP dt = upload_vec(tab);
    launch_asg_fal_scatter(out.g_dev, out.rec_mem->as<int>(), dt->as<int64_t>(), out.n, longest, gm->as<float>(), rt.stream());
    tr.add_grad_device(gm, gm->as<float>(), /*adopt=*/true);  // (accumulates when the graph holds a gradient already)
  }
};

// scalar graphs (results of the per-graph functions) taken into a batch expression: forward gathers their
// values, backward hands every graph its own delta and runs the per-graph tape from there.  The wrapper is an
// identity the reference's tape does not have, so it must be transparent to the reference's accumulation rule
// (every node passes its ACCUMULATED gradient on, autograd.cpp:40-52): over a retained tape the wrapper's own
// gradient grows with every backward, and what the graphs are handed is only what they have not seen yet
// (`passed`) -- in a fresh block, never the wrapper's own buffer (an adopted gradient aliases its source).
struct BFromGraphsOp : BatchOp {
  DevMemP passed;  // the wrapper's gradient as of the last backward
  void backward(Batch& out) override {
    Batch& src = *inputs[0];
    Runtime& rt = Runtime::get();
    const size_t n = size_t(src.n);
    DevMemP delta = rt.alloc(sizeof(float) * (n ? n : 1));
    if (passed)
      launch_vec_axpby(delta->as<float>(), out.g_dev, passed->as<float>(), n, 1.0f, -1.0f, 0, rt.stream());
    else
      rt.d2d(delta->ptr, out.g_dev, sizeof(float) * n);
    if (retain) {
      if (!passed) passed = rt.alloc(sizeof(float) * (n ? n : 1));
      rt.d2d(passed->ptr, out.g_dev, sizeof(float) * n);
    }
    std::vector<Graph> roots;
    for (int b = 0; b < src.n; ++b) {
      Graph& r = src.graphs[size_t(b)];
      if (!r.calc_grad()) continue;
      r.add_grad_device(delta, delta->as<float>() + b, /*adopt=*/true);
      roots.push_back(r);
    }
    if (!roots.empty()) op_backward(roots, nullptr, retain, /*seed=*/false);
  }
  bool retain = false;
};

}  // namespace

Batch::~Batch() {
  if (host_ev) (void)hipEventDestroy(static_cast<hipEvent_t>(host_ev));
  if (give_back && !origins.empty()) {
    for (const Origin& o : origins) {
      auto* part = new std::vector<Graph>();
      part->reserve(o.end - o.begin);
      for (size_t i = o.begin; i < o.end && i < graphs.size(); ++i) part->push_back(std::move(graphs[i]));
      give_back(o.home, part);
    }
    graphs.clear();
    return;
  }
  // taken apart off the caller's critical path, a few dozen graphs per entry so that the threads of the next
  // parallelMap region share them (runtime.cpp: drain_some)
  if (graphs.size() < 64 || !Runtime::initialized()) return;
  Runtime& rt = Runtime::get();
  for (size_t i = 0; i < graphs.size(); i += 32) {
    auto* part = new std::vector<Graph>();
    const size_t e = std::min(graphs.size(), i + 32);
    part->reserve(e - i);
    for (size_t k = i; k < e; ++k) part->push_back(std::move(graphs[k]));
    rt.defer_delete(part, [](void* q) { delete static_cast<std::vector<Graph>*>(q); }, e - i);
  }
  graphs.clear();
}

int64_t Batch::elem_size(int b) const {
  switch (kind) {
    case SCALAR: return 1;
    case LINEAR: return int64_t(M) * C;
    case CTC_TARGETS:
      if (fal) return 2 * int64_t(lab_off[size_t(b) + 1] - lab_off[size_t(b)]);
      return 3 * int64_t(2 * (lab_off[size_t(b) + 1] - lab_off[size_t(b)]) + 1);
    default: return 0;
  }
}

// ---- creation --------------------------------------------------------------------------------
BatchP batch_from_graphs(std::vector<Graph> gs) {
  bool cg = false;
  for (auto& g : gs) cg |= g.calc_grad();
  BatchP b = make_batch(Batch::GRAPHS, int(gs.size()), cg);
  b->graphs = std::move(gs);
  b->materialised = true;
  return b;
}

namespace {
Graph ctc_target_graph_host(const int* t, int U, int blank, bool cg) {
  // benchmarks/ctc.cpp:40-58, same node and arc order
  const int L = 2 * U + 1;
  std::vector<uint8_t> st(size_t(L), 0), ac(size_t(L), 0);
  std::vector<int> src, dst, lab;
  src.reserve(size_t(3 * L));
  dst.reserve(size_t(3 * L));
  lab.reserve(size_t(3 * L));
  for (int l = 0; l < L; ++l) {
    const int idx = (l - 1) / 2;
    st[size_t(l)] = l == 0;
    ac[size_t(l)] = l == L - 1 || l + 2 == L;
    const int label = l % 2 ? t[idx] : blank;
    src.push_back(l), dst.push_back(l), lab.push_back(label);
    if (l > 0) src.push_back(l - 1), dst.push_back(l), lab.push_back(label);
    if (l % 2 && l > 1 && label != t[idx - 1]) src.push_back(l - 2), dst.push_back(l), lab.push_back(label);
  }
  Graph g(cg);
  g.add_nodes(L, st.data(), ac.data());
  g.add_arcs(int(src.size()), src.data(), dst.data(), lab.data(), lab.data(), nullptr);
  g.arc_sort(false);
  return g;
}
}  // namespace

BatchP batch_ctc_targets(const int* labels, const int* lengths, int n, int blank, bool calc_grad) {
  GTNX_HOST_T("batch.ctc_targets");
  if (n < 0) throw_invalid("[gtnx_batch_ctc_targets] negative batch size");
  BatchP b = make_batch(Batch::CTC_TARGETS, n, calc_grad);
  b->blank = blank;
  b->lab_off.resize(size_t(n) + 1);
  b->lab_off[0] = 0;
  for (int i = 0; i < n; ++i) {
    if (lengths[i] < 0) throw_invalid("[gtnx_batch_ctc_targets] negative length");
    b->lab_off[size_t(i) + 1] = b->lab_off[size_t(i)] + lengths[i];
    b->max_nodes = std::max(b->max_nodes, 2 * lengths[i] + 1);
  }
  const size_t total = size_t(b->lab_off[size_t(n)]);
  b->labels.assign(labels, labels + total);
  int mn = blank;
  b->max_label = blank;
  for (size_t i = 0; i < total; ++i) {
    b->max_label = std::max(b->max_label, labels[i]);
    mn = std::min(mn, labels[i]);
  }
  if (mn < 0 || b->max_nodes > band_max_nodes() || n == 0) {
    // epsilon / negative labels or targets wider than a workgroup: ordinary graphs
    std::vector<Graph> gs;
    gs.reserve(size_t(n));
    for (int i = 0; i < n; ++i)
      gs.push_back(ctc_target_graph_host(labels + b->lab_off[size_t(i)], lengths[i], blank, calc_grad));
    return batch_from_graphs(std::move(gs));
  }
  // records on the device: labels | the kernel's argument table | per element {nodes, flags, snode, slab, n_arcs}.
  // Labels and table cross the host link in ONE copy (they were two: a copy is a dependent operation of ~5 us at
  // the head of every step)
  Runtime& rt = Runtime::get();
  const size_t lab_bytes = align_up(sizeof(int) * (total ? total : 1), 256);
  const size_t arg_bytes = align_up(sizeof(CtcTargetArgs) * size_t(n), 256);
  size_t bytes = lab_bytes + arg_bytes;
  b->rec_off.resize(size_t(n));
  for (int i = 0; i < n; ++i) {
    const size_t N = size_t(2 * lengths[i] + 1);
    b->rec_off[size_t(i)] = bytes;
    bytes += align_up(sizeof(BandNode) * N, 64) + align_up(N, 64) + 2 * align_up(4 * N, 64) + 64;
  }
  b->rec_mem = rt.alloc(bytes);
  PinnedMemP pin = rt.alloc_pinned(lab_bytes + arg_bytes);
  std::memcpy(pin->ptr, labels, sizeof(int) * total);
  CtcTargetArgs* args = reinterpret_cast<CtcTargetArgs*>(static_cast<char*>(pin->ptr) + lab_bytes);
  for (int i = 0; i < n; ++i) {
    const size_t N = size_t(2 * lengths[i] + 1);
    char* base = b->rec_mem->as<char>(b->rec_off[size_t(i)]);
    CtcTargetArgs& a = args[size_t(i)];
    a.labels = b->rec_mem->as<int>() + b->lab_off[size_t(i)];
    a.nodes = reinterpret_cast<BandNode*>(base);
    base += align_up(sizeof(BandNode) * N, 64);
    a.nflags = reinterpret_cast<uint8_t*>(base);
    base += align_up(N, 64);
    a.snode = reinterpret_cast<int*>(base);
    base += align_up(4 * N, 64);
    a.slab = reinterpret_cast<int*>(base);
    base += align_up(4 * N, 64);
    a.n_arcs = reinterpret_cast<int*>(base);
    a.N = int(N);
    a.pad = 0;
  }
  rt.h2d_pinned(b->rec_mem->ptr, pin->ptr, lab_bytes + arg_bytes);
  launch_ctc_targets(reinterpret_cast<const CtcTargetArgs*>(b->rec_mem->as<char>(lab_bytes)), n, blank, rt.stream());
  return b;
}


BatchP batch_asg_force_align(const int* labels, const int* lengths, int n, Graph& transitions, int n_labels) {
  if (n < 0) throw_invalid("[gtnx_batch_asg_force_align] negative batch size");
  // the arc layout the gather relies on (gtn_amd/criteria/asg_criterion.h: asgTransitions; examples/asg.cpp:36-47)
  if (transitions.num_nodes() != int64_t(n_labels) + 1 || transitions.num_arcs() != int64_t(n_labels) * (n_labels + 1))
    throw_invalid("[gtnx_batch_asg_force_align] the transitions graph must have the asgTransitions(N) layout");
  BatchP b = make_batch(Batch::CTC_TARGETS, n, transitions.calc_grad());
  b->fal = true;
  b->trans = transitions;
  b->trans_labels = n_labels;
  b->lab_off.resize(size_t(n) + 1);
  b->lab_off[0] = 0;
  int max_u = 0;
  for (int i = 0; i < n; ++i) {
    if (lengths[i] < 0) throw_invalid("[gtnx_batch_asg_force_align] negative length");
    b->lab_off[size_t(i) + 1] = b->lab_off[size_t(i)] + lengths[i];
    max_u = std::max(max_u, lengths[i]);
  }
  b->max_nodes = max_u + 1;
  const size_t total = size_t(b->lab_off[size_t(n)]);
  b->labels.assign(labels, labels + total);
  int mn = 0;
  for (size_t i = 0; i < total; ++i) {
    b->max_label = std::max(b->max_label, labels[i]);
    mn = std::min(mn, labels[i]);
  }
  if (mn < 0 || b->max_label >= n_labels || b->max_nodes > band_max_nodes() || n == 0) {
    b->kind = Batch::GRAPHS;  // labels outside the transitions / too long for a workgroup: the per-graph way
    batch_materialise(*b);
    return b;
  }
  Runtime& rt = Runtime::get();
  std::vector<Weights*> ws{transitions.w.get()};
  ensure_weights_device_batch(ws);
  size_t bytes = align_up(sizeof(int) * (total ? total : 1), 256);
  b->rec_off.resize(size_t(n));
  b->w_off.resize(size_t(n));
  b->map_off.resize(size_t(n));
  for (int i = 0; i < n; ++i) {
    const size_t N = size_t(lengths[i]) + 1, A = 2 * size_t(lengths[i]);
    b->rec_off[size_t(i)] = bytes;
    bytes += align_up(sizeof(BandNode) * N, 64) + align_up(N, 64) + 2 * align_up(4 * N, 64);
    b->w_off[size_t(i)] = bytes;
    bytes += align_up(4 * (A ? A : 1), 64);
    b->map_off[size_t(i)] = bytes;
    bytes += align_up(4 * (A ? A : 1), 64);
  }
  b->rec_mem = rt.alloc(bytes);
  {
    PinnedMemP pin = rt.alloc_pinned(sizeof(int) * (total ? total : 1));
    std::memcpy(pin->ptr, labels, sizeof(int) * total);
    rt.h2d_pinned(b->rec_mem->ptr, pin->ptr, sizeof(int) * total);
  }
  std::vector<AsgFalArgs> args;
  args.resize(size_t(n));
  for (int i = 0; i < n; ++i) {
    const size_t N = size_t(lengths[i]) + 1;
    char* base = b->rec_mem->as<char>(b->rec_off[size_t(i)]);
    AsgFalArgs& a = args[size_t(i)];
    a.labels = b->rec_mem->as<int>() + b->lab_off[size_t(i)];
    a.trans_w = transitions.w->dev;
    a.nodes = reinterpret_cast<BandNode*>(base);
    base += align_up(sizeof(BandNode) * N, 64);
    a.nflags = reinterpret_cast<uint8_t*>(base);
    base += align_up(N, 64);
    a.snode = reinterpret_cast<int*>(base);
    base += align_up(4 * N, 64);
    a.slab = reinterpret_cast<int*>(base);
    a.w = b->rec_mem->as<float>(b->w_off[size_t(i)]);
    a.arc_map = b->rec_mem->as<int>(b->map_off[size_t(i)]);
    a.U = lengths[i];
    a.pad = 0;
  }
  DevMemP d = upload_vec(args);
  launch_asg_fal_targets(d->as<AsgFalArgs>(), n, n_labels, rt.stream());
  auto op = std::make_shared<BFalOp>();
  op->seq = g_batch_seq.fetch_add(1);
  b->op = op;  // (its one input, the transitions graph, is an ordinary graph: kept in `trans`)
  return b;
}

BatchP batch_linear(int n, int M, int C, bool calc_grad, const void* dev, bool borrow) {
  if (n < 0 || M < 0 || C < 0) throw_invalid("[gtnx_batch_linear] negative size");
  Runtime& rt = Runtime::get();
  BatchP b = make_batch(Batch::LINEAR, n, calc_grad);
  b->M = M;
  b->C = C;
  const size_t bytes = sizeof(float) * size_t(n) * size_t(M) * size_t(C);
  if (borrow && dev) {
    b->w_mem = std::make_shared<DevMem>();
    b->w_mem->ptr = const_cast<void*>(dev);
    b->w_mem->bytes = bytes;
    b->w_mem->borrowed = true;
  } else {
    b->w_mem = dev ? rt.alloc(bytes ? bytes : 4) : rt.alloc_zero(bytes ? bytes : 4);
    if (dev && bytes) rt.d2d(b->w_mem->ptr, dev, bytes);
  }
  b->w_dev = b->w_mem->as<float>();
  return b;
}

// ---- the caller's graphs as native leaves --------------------------------------------------------
BatchP batch_ctc_targets_from_graphs(const std::vector<Graph>& gs) {
  GTNX_HOST_T("batch.ctc_targets_from_graphs");
  const int n = int(gs.size());
  if (n == 0) return nullptr;
  detect_ctc_shape(*gs[0].s);
  const Structure& s0 = *gs[0].s;
  if (!s0.ctc_labels) return nullptr;
  const bool cg = gs[0].calc_grad();
  std::vector<int> flat, len;
  len.reserve(size_t(n));
  flat.reserve(size_t(n) * s0.ctc_labels->size());
  for (auto& g : gs) {
    detect_ctc_shape(*g.s);
    const Structure& s = *g.s;
    // (arcSort by input label is what benchmarks/ctc.cpp:56 asks for; a CTC acceptor's lists are the same either way)
    if (!s.ctc_labels || s.ctc_blank != s0.ctc_blank || g.calc_grad() != cg || s.N > band_max_nodes() ||
        !g.w->is_all_zero())
      return nullptr;
    flat.insert(flat.end(), s.ctc_labels->begin(), s.ctc_labels->end());
    len.push_back(int(s.ctc_labels->size()));
  }
  BatchP b = batch_ctc_targets(flat.data(), len.data(), n, s0.ctc_blank, cg);
  if (b->kind != Batch::CTC_TARGETS) return nullptr;  // (labels the records cannot hold: the per-graph way)
  b->graphs = gs;
  b->leaf = true;
  for (auto& g : gs) g.s->leaf_batch = b;
  return b;
}

BatchP batch_linear_from_graphs(const std::vector<Graph>& gs) {
  GTNX_HOST_T("batch.linear_from_graphs");
  const int n = int(gs.size());
  if (n == 0) return nullptr;
  const Structure& s0 = *gs[0].s;
  if (s0.kind != KIND_LINEAR || s0.M < 1 || s0.C < 1) return nullptr;
  const bool cg = gs[0].calc_grad();
  std::vector<Weights*> ws;
  ws.reserve(size_t(n));
  std::unordered_set<Weights*> distinct;
  for (auto& g : gs) {
    const Structure& s = *g.s;
    if (s.kind != KIND_LINEAR || s.M != s0.M || s.C != s0.C || g.calc_grad() != cg || g.w->host_escaped ||
        !distinct.insert(g.w.get()).second)
      return nullptr;
    ws.push_back(g.w.get());
  }
  ensure_weights_device_batch(ws);
  Runtime& rt = Runtime::get();
  const size_t A = size_t(s0.M) * size_t(s0.C);
  BatchP b = make_batch(Batch::LINEAR, n, cg);
  b->M = s0.M;
  b->C = s0.C;
  bool contiguous = true;
  for (int i = 0; i < n && contiguous; ++i)
    contiguous = ws[size_t(i)]->dev == ws[0]->dev + size_t(i) * A && ws[size_t(i)]->dev_mem == ws[0]->dev_mem;
  if (contiguous) {
    b->w_mem = ws[0]->dev_mem;
    b->w_dev = ws[0]->dev;
  } else {
    // one [n][M][C] tensor: gather the graphs' weights and let the graphs read it from now on (same values)
    b->w_mem = rt.alloc(sizeof(float) * A * size_t(n));
    b->w_dev = b->w_mem->as<float>();
    std::vector<CopySeg> segs;
    segs.resize(size_t(n));
    for (int i = 0; i < n; ++i) segs[size_t(i)] = {b->w_dev + size_t(i) * A, ws[size_t(i)]->dev, int64_t(4 * A)};
    DevMemP d = upload_vec(segs);
    launch_copy_segments(d->as<CopySeg>(), n, int64_t(4 * A), rt.stream());
    for (int i = 0; i < n; ++i) {
      ws[size_t(i)]->dev_mem = b->w_mem;
      ws[size_t(i)]->dev = b->w_dev + size_t(i) * A;
    }
  }
  b->graphs = gs;
  b->leaf = true;
  for (auto& g : gs) {
    g.w->leaf_batch = b;
    g.w->leaf_version = g.w->version;
  }
  // gtnx_grads_bind_device_n on the graphs: their first gradients go straight to the caller's tensor when
  // that is one block in element order
  if (cg && gs[0].g->grad_dest && !gs[0].is_grad_available()) {
    bool block = true;
    for (int i = 0; i < n && block; ++i)
      block = gs[size_t(i)].g->grad_dest == gs[0].g->grad_dest + size_t(i) * A && !gs[size_t(i)].is_grad_available();
    if (block) {
      b->dest_mem = gs[0].g->grad_dest_mem;
      b->dest = gs[0].g->grad_dest;
    }
  }
  return b;
}

// ---- elements as graphs ------------------------------------------------------------------------
namespace {
// what the batch-level backward produced moves into the element graphs
void push_grads_to_graphs(Batch& x) {
  if (!x.g_dev) return;
  GTNX_HOST_T("batch.push_grads_to_graphs");
  GradSink sink;  // first gradients are adopted in place, the others folded in by ONE launch
  const size_t n = size_t(x.n);
  for (size_t i = 0; i < n; ++i) {
    // (the graphs were built by the region's threads: each is a miss in this core's caches)
    if (i + 6 < n) __builtin_prefetch(x.graphs[i + 6].g.get(), 1);
    if (i + 3 < n) __builtin_prefetch(x.graphs[i + 3].s.get(), 1);
    Graph& g = x.graphs[i];
    if (!g.calc_grad()) continue;
    float* ptr = x.g_dev + x.g_off[i];
    if (!g.s->deferred) {  // the common case -- a first gradient: only the block is noted (Graph::add_grad_device)
      std::lock_guard<std::mutex> lk(g.s->grad_lock);
      if (!g.is_grad_available()) {
        g.g->lazy_owner = x.g_mem;
        g.g->lazy_ptr = ptr;
        continue;
      }
    }
    sink.add(g, x.g_mem, ptr);
  }
  sink.flush();
  x.g_dev = nullptr;  // (the graphs hold the block now)
  x.g_mem.reset();
}
}  // namespace

void batch_materialise(Batch& x) {
  if (x.materialised) return;
  if (x.leaf) {  // the elements have been graphs all along
    x.materialised = true;
    push_grads_to_graphs(x);
    return;
  }
  GraphSlabScope slab_scope(size_t(x.n > 0 ? x.n : 0));  // the elements' pieces out of one allocation (graph.h)
  std::vector<Graph> gs;
  switch (x.fal && x.kind == Batch::GRAPHS ? Batch::CTC_TARGETS : x.kind) {
    case Batch::GRAPHS: break;
    case Batch::CTC_TARGETS:
      gs.reserve(size_t(x.n));
      if (x.fal) {  // compose(forceAlign(target), transitions), examples/asg.cpp:50-68
        std::vector<Graph> fals;
        for (int i = 0; i < x.n; ++i) {
          const int* t = x.labels.data() + x.lab_off[size_t(i)];
          const int U = x.lab_off[size_t(i) + 1] - x.lab_off[size_t(i)];
          Graph f(false);
          f.add_node(true, U == 0);
          for (int l = 1; l <= U; ++l) {
            f.add_node(false, l == U);
            f.add_arc(l - 1, l, t[l - 1], t[l - 1], 0.0f);
            f.add_arc(l, l, t[l - 1], t[l - 1], 0.0f);
          }
          fals.push_back(std::move(f));
        }
        std::vector<Graph> tr{x.trans};
        gs = op_compose(fals, tr, false);
        break;
      }
      for (int i = 0; i < x.n; ++i)
        gs.push_back(ctc_target_graph_host(x.labels.data() + x.lab_off[size_t(i)],
                                           x.lab_off[size_t(i) + 1] - x.lab_off[size_t(i)], x.blank, x.calc_grad));
      break;
    case Batch::LINEAR: {
      gs.reserve(size_t(x.n));
      const int64_t A = int64_t(x.M) * x.C;
      for (int i = 0; i < x.n; ++i) {
        Graph g = Graph::make_result(x.calc_grad);  // (out of the scope's slab: an element refers to nothing)
        Structure& s = *g.s;
        s.kind = KIND_LINEAR;
        s.M = x.M;
        s.C = x.C;
        s.N = int64_t(x.M) + 1;
        s.A = A;
        s.ilabel_sorted = s.olabel_sorted = true;
        Weights& w = *g.w;
        w.n = A;
        if (x.w_pend) x.w_pend->settle();  // (the element graphs' weights are looked at by whoever gets them)
        w.dev_mem = x.w_mem;
        w.dev = x.w_dev + size_t(i) * size_t(A);
        w.dev_valid = true;
        w.host_valid = false;
        w.version++;
        gs.push_back(std::move(g));
      }
      break;
    }
    case Batch::PRODUCT: {
      batch_materialise(*x.fixed);
      batch_materialise(*x.chain);
      CompMode symbolic;
      gs = x.chain_first ? op_compose(x.chain->graphs, x.fixed->graphs, x.intersect)
                         : op_compose(x.fixed->graphs, x.chain->graphs, x.intersect);
      break;
    }
    case Batch::SCALAR: {
      if (!x.v_mem && x.v_dev) {  // values in the caller's memory (batch_scalar's items_dev): element graphs own theirs
        Runtime& rt = Runtime::get();
        x.v_mem = rt.alloc(sizeof(float) * size_t(x.n ? x.n : 1));
        rt.d2d(x.v_mem->ptr, x.v_dev, sizeof(float) * size_t(x.n));
        x.v_dev = x.v_mem->as<float>();
      }
      if (!x.op) {  // the tape is gone (backward without retain): plain values
        gs.reserve(size_t(x.n));
        for (int i = 0; i < x.n; ++i) {
          Graph g(false);
          Structure& s = *g.s;
          s.kind = KIND_LINEAR;
          s.M = s.C = 1;
          s.N = 2;
          s.A = 1;
          Weights& w = *g.w;
          w.n = 1;
          w.dev_mem = x.v_mem;
          w.dev = x.v_dev + i;
          w.dev_valid = true;
          w.host_valid = false;
          w.version++;
          gs.push_back(std::move(g));
        }
        break;
      }
      for (auto& in : x.op->inputs) batch_materialise(*in);
      if (auto* so = dynamic_cast<BScalarOp*>(x.op.get())) {
        std::vector<Graph> none;
        gs = op_scalar(so->kind, so->inputs[0]->graphs, so->inputs.size() > 1 ? so->inputs[1]->graphs : none);
      } else {
        gs = op_shortest_distance(x.op->inputs[0]->graphs, false);
      }
      break;
    }
  }
  if (x.kind != Batch::GRAPHS || x.fal) x.graphs = std::move(gs);
  x.materialised = true;
  // what the batch-level backward already produced moves into the element graphs
  if (x.g_dev && ((x.kind == Batch::CTC_TARGETS && !x.fal) || x.kind == Batch::LINEAR)) {
    for (int i = 0; i < x.n; ++i) x.graphs[size_t(i)].add_grad_device(x.g_mem, x.g_dev + x.g_off[size_t(i)], true);
    x.g_dev = nullptr;  // (the graphs hold the block now)
    x.g_mem.reset();
  }
}

Graph batch_get(const BatchP& x, int i) {
  if (i < 0 || i >= x->n) throw_range("[gtnx_batch_get] element index out of range");
  batch_materialise(*x);
  return x->graphs[size_t(i)];
}

// ---- functions -------------------------------------------------------------------------------
BatchP batch_compose(const BatchP& a, const BatchP& b, bool intersect) {
  const Batch *fx = nullptr, *ch = nullptr;
  bool chain_first = false;
  if (native(*a, Batch::CTC_TARGETS) && native(*b, Batch::LINEAR)) fx = a.get(), ch = b.get();
  if (native(*a, Batch::LINEAR) && native(*b, Batch::CTC_TARGETS)) fx = b.get(), ch = a.get(), chain_first = true;
  if (fx && a->n == b->n && ch->C >= band_min_labels() && ch->C <= band_max_labels() && fx->max_label < ch->C && ch->M <= (1 << 20)) {
    struct Op : BatchOp {
      void backward(Batch&) override {}  // a symbolic product has no gradient of its own
    };
    auto op = std::make_shared<Op>();
    op->inputs = {a, b};
    BatchP r = result(Batch::PRODUCT, a->n, op);
    r->fixed = chain_first ? b : a;
    r->chain = chain_first ? a : b;
    r->chain_first = chain_first;
    r->intersect = intersect;
    return r;
  }
  batch_materialise(*a);
  batch_materialise(*b);
  return batch_from_graphs(op_compose(a->graphs, b->graphs, intersect));
}

BatchP batch_shortest_distance(const BatchP& x, bool tropical) {
  GTNX_HOST_T("batch.shortest_distance");
  Runtime& rt = Runtime::get();
  if (!tropical && native(*x, Batch::PRODUCT) && !x->materialised) {
    Batch& fx = *x->fixed;
    Batch& ch = *x->chain;
    const int n = x->n, T = ch.M, C = ch.C;
    auto op = std::make_shared<BFsBandOp>();
    op->inputs = {x};
    const bool want_norm = ch.nc_norm == nullptr;
    // scores [n] | norm [n] | rowlse [n][T] | per element: shifts, alpha plane
    size_t bytes = align_up(4 * size_t(n), 256);
    const size_t o_norm = bytes;
    bytes = align_up(bytes + 4 * size_t(n), 256);
    const size_t o_lse = bytes;
    bytes = align_up(bytes + 4 * size_t(n) * size_t(T > 0 ? T : 1), 256);
    std::vector<size_t> oo, ao;
    oo.resize(size_t(n));
    ao.resize(size_t(n));
    for (int b = 0; b < n; ++b) {
      const int Ub = fx.lab_off[size_t(b) + 1] - fx.lab_off[size_t(b)];
      const int N = fx.fal ? Ub + 1 : 2 * Ub + 1;
      const int ns = band_row_stride(N, band_npl(N));
      oo[size_t(b)] = bytes;
      bytes = align_up(bytes + 8 * (4 * size_t(T) + 16), 256);
      ao[size_t(b)] = bytes;
      bytes = align_up(bytes + 4 * size_t(T + 1) * size_t(ns), 256);
    }
    op->arena = rt.alloc(bytes);
    op->pairs.resize(size_t(n));
    std::vector<std::pair<BandLaunchKey, BandPair>> tab;
    tab.reserve(size_t(n));
    double abytes = 0;
    // the copy of the chain's values a region still owes (graph.h PendingCopy): made by this sweep when every
    // element's source is known; the lock keeps another thread's settle() from making it at the same time
    std::shared_ptr<PendingCopy> pend = ch.w_pend;
    std::unique_lock<std::mutex> pend_lock;
    bool fuse_copy = false;
    if (pend && !pend->done.load(std::memory_order_acquire)) {
      pend_lock = std::unique_lock<std::mutex>(pend->mu);
      fuse_copy = !pend->done.load(std::memory_order_relaxed) && pend->device == rt.device();
      const size_t A = size_t(T) * size_t(C);
      for (int b = 0; b < n && fuse_copy; ++b) fuse_copy = pend->src_of(ch.w_dev + size_t(b) * A) != nullptr;
      // (a copy that also serves graphs outside this record is made whole: the sweep would leave the rest undone)
      fuse_copy = fuse_copy && pend->segs.size() == size_t(n);
      if (!fuse_copy) {
        pend_lock.unlock();
        pend->settle();
      }
    }
    int64_t goff_run = 0;
    for (int b = 0; b < n; ++b) {
      BandPair& p = op->pairs[size_t(b)];
      p = BandPair{};
      const int U = fx.lab_off[size_t(b) + 1] - fx.lab_off[size_t(b)];
      const size_t N = fx.fal ? size_t(U) + 1 : size_t(2 * U + 1);
      char* base = fx.rec_mem->as<char>(fx.rec_off[size_t(b)]);
      p.nodes = reinterpret_cast<BandNode*>(base);
      base += align_up(sizeof(BandNode) * N, 64);
      p.nflags = reinterpret_cast<uint8_t*>(base);
      base += align_up(N, 64);
      p.snode = reinterpret_cast<int*>(base);
      base += align_up(4 * N, 64);
      p.slab = reinterpret_cast<int*>(base);
      p.n_lab = fx.fal ? U : int(N);
      p.w = fx.fal ? fx.rec_mem->as<float>(fx.w_off[size_t(b)]) : nullptr;  // CTC targets: all-zero weights
      p.em = ch.w_dev + size_t(b) * size_t(T) * size_t(C);
      p.N = int(N);
      p.T = T;
      p.C = C;
      p.NS = band_row_stride(p.N, band_npl(p.N));
      p.alpha = op->arena->as<float>(ao[size_t(b)]);
      p.aoff = op->arena->as<double>(oo[size_t(b)]);
      p.score = op->arena->as<float>(4 * size_t(b));
      if (want_norm) {
        p.norm = op->arena->as<float>(o_norm + 4 * size_t(b));
        p.rowlse = op->arena->as<float>(o_lse + 4 * size_t(b) * size_t(T));
      }
      p.hot = (!fx.fal && U + 1 >= 8) ? fx.blank : -1;
      p.lgrn = band_forward_lgrn(C);
      p.bidx = b;
      p.goff = goff_run;  // (the prefix sums alloc_grad makes at backward time: elements back to back)
      goff_run += int64_t(fx.elem_size(b));
      // The chain's values may still be in the caller's buffer (staged by the region this sweep belongs to): the
      // sweep reads them THERE and stores them at p.em on its way -- the copy the region owes -- so that the
      // backward sweep, and anybody else later, reads the graph's own copy.  (p, as kept for backward: the copy.)
      BandPair q = p;
      if (fuse_copy) {
        q.em = static_cast<const float*>(pend->src_of(p.em));
        q.em_copy = const_cast<float*>(p.em);
      }
      tab.push_back({BandLaunchKey{C, band_npl(q.N), fx.fal ? 0 : 1, 0, band_vec_ok(q)}, q});
      abytes += 4.0 * T * C + 4.0 * double(T + 1) * p.NS + (fuse_copy ? 4.0 * T * C : 0.0);
    }
    // (a launch that also makes the region's copy reads the caller's buffer: its table is not the backward sweep's)
    band_launch(tab, false, "band_forward_score", abytes, fuse_copy ? nullptr : &op->fwd_table);
    if (fuse_copy) pend->done.store(true, std::memory_order_release);  // (under pend->mu, taken above)
    if (want_norm) {
      ch.nc_mem = op->arena;
      ch.nc_norm = op->arena->as<float>(o_norm);
      ch.nc_rowlse = op->arena->as<float>(o_lse);
    }
    DevMemP arena = op->arena;
    BatchP r = result(Batch::SCALAR, n, op);
    r->v_mem = arena;
    r->v_dev = arena->as<float>();
    return r;
  }
  if (!tropical && native(*x, Batch::LINEAR) && !x->materialised && x->nc_norm) {
    // left behind by the sweep over the same chains: nothing to launch
    auto op = std::make_shared<BFsLinearOp>();
    op->inputs = {x};
    BatchP r = result(Batch::SCALAR, x->n, op);
    r->v_mem = x->nc_mem;
    r->v_dev = x->nc_norm;
    return r;
  }
  if (!tropical && native(*x, Batch::LINEAR) && !x->materialised && x->n > 0 && x->M > 0) {
    // forwardScore of B chains over one [B][M][C] tensor (BASELINE config C2) as ONE record: row log-sum-exps summed
    // per chain (misc.hip: linear_rows_kernel / linear_forward_kernel, shortest.cpp:86-170 on a chain), no element
    // graphs; the gradient is the softmax of every row (BFsLinearOp, as when a sweep left the scores behind)
    if (x->w_pend) x->w_pend->settle();  // (the values are read here: graph.h PendingCopy)
    auto op = std::make_shared<BFsLinearOp>();
    op->inputs = {x};
    const int n = x->n;
    DevMemP res = rt.alloc(sizeof(float) * size_t(n) * 9);
    float* scal = res->as<float>();
    float* partial = scal + n;
    std::vector<LinArgs> args;
    args.resize(size_t(n));
    const size_t A = size_t(x->M) * size_t(x->C);
    for (int b = 0; b < n; ++b) {
      LinArgs& a = args[size_t(b)];
      a.w = x->w_dev + size_t(b) * A;
      a.M = x->M;
      a.C = x->C;
      a.out_score = scal + b;
      a.partial = partial + size_t(b) * 8;
      a.delta = nullptr;
      a.grad = nullptr;
      a.accumulate = 0;
    }
    DevMemP d = upload_vec(args);
    const bool vec_rows = x->C % 4 == 0 && x->C <= 1024 && (reinterpret_cast<uintptr_t>(x->w_dev) & 15) == 0 && (A % 4) == 0;
    {
      GTNX_PROF("linear_forward", 4.0 * double(A) * n);
      launch_linear_forward(d->as<LinArgs>(), n, 0, vec_rows ? 1 : 0, rt.stream());
    }
    BatchP r = result(Batch::SCALAR, n, op);
    r->v_mem = res;
    r->v_dev = scal;
    return r;
  }
  batch_materialise(*x);
  return batch_from_graphs(op_shortest_distance(x->graphs, tropical));
}

BatchP batch_viterbi_path(const BatchP& x) {
  batch_materialise(*x);
  return batch_from_graphs(op_viterbi_path(x->graphs));
}


namespace {
// a GRAPHS batch of one-arc graphs as a native SCALAR batch (values gathered; backward continues on the graphs' tape)
BatchP scalars_from_graphs(const BatchP& gsb) {
  Runtime& rt = Runtime::get();
  for (auto& g : gsb->graphs)
    if (g.num_arcs() != 1) return nullptr;
  auto op = std::make_shared<BFromGraphsOp>();
  op->inputs = {gsb};
  BatchP r = result(Batch::SCALAR, gsb->n, op);
  r->v_mem = rt.alloc(sizeof(float) * size_t(gsb->n ? gsb->n : 1));
  r->v_dev = r->v_mem->as<float>();
  items_device(gsb->graphs, r->v_dev);
  return r;
}
}  // namespace

BatchP batch_scalar(ScalarKind k, const BatchP& a0, const BatchP& b0, void* items_dev) {
  BatchP a = a0, b = b0;
  const bool binary = k != SK_NEGATE;
  // one side native, the other the per-graph functions' results: take those in, do not rebuild the native side
  if (binary && a->n == b->n) {
    const bool an = a->kind == Batch::SCALAR && !a->materialised, bn = b->kind == Batch::SCALAR && !b->materialised;
    if (an && !bn && b->kind == Batch::GRAPHS) {
      if (BatchP w = scalars_from_graphs(b)) b = w;
    } else if (bn && !an && a->kind == Batch::GRAPHS) {
      if (BatchP w = scalars_from_graphs(a)) a = w;
    }
  }

  if (native(*a, Batch::SCALAR) && !a->materialised &&
      (!binary || (native(*b, Batch::SCALAR) && !b->materialised && b->n == a->n))) {
    Runtime& rt = Runtime::get();
    auto op = std::make_shared<BScalarOp>();
    op->kind = k;
    op->inputs = binary ? std::vector<BatchP>{a, b} : std::vector<BatchP>{a};
    BatchP r = result(Batch::SCALAR, a->n, op);
    if (items_dev) {  // the values go where the caller wants them (no block of the engine's, no copy afterwards)
      r->v_dev = static_cast<float*>(items_dev);
    } else {
      r->v_mem = rt.alloc(sizeof(float) * size_t(a->n ? a->n : 1));
      r->v_dev = r->v_mem->as<float>();
    }
    launch_vec_axpby(r->v_dev, a->v_dev, binary ? b->v_dev : nullptr, size_t(a->n), k == SK_NEGATE ? -1.0f : 1.0f,
                     k == SK_SUBTRACT ? -1.0f : 1.0f, 0, rt.stream());
    return r;
  }
  batch_materialise(*a);
  std::vector<Graph> none;
  if (binary) batch_materialise(*b);
  BatchP r = batch_from_graphs(op_scalar(k, a->graphs, binary ? b->graphs : none));
  if (items_dev) batch_items_device(r, items_dev);
  return r;
}

// ---- autograd --------------------------------------------------------------------------------
void batch_backward(const BatchP& root, bool retain) {
  GTNX_HOST_T("batch.backward");
  if (root->tape_cleared)
    throw_invalid("[autograd::backward] Cannot Backward twice without retaining the graph.");  // autograd.cpp:44-47
  if (root->materialised || !root->op || root->kind != Batch::SCALAR) {
    batch_materialise(*root);
    op_backward(root->graphs, nullptr, retain);
    return;
  }
  Runtime& rt = Runtime::get();
  // reachable producers, newest first (creation order is a topological order: autograd.cpp:17-67)
  std::vector<Batch*> order;
  std::unordered_set<Batch*> seen;
  std::vector<BatchP> hold;  // (dropping a producer at the end must not take its inputs away under us)
  std::vector<BatchP> stack{root};
  while (!stack.empty()) {
    BatchP b = stack.back();
    stack.pop_back();
    if (!b->calc_grad || !seen.insert(b.get()).second) continue;
    hold.push_back(b);
    if (b->op) {
      order.push_back(b.get());
      for (auto& in : b->op->inputs) stack.push_back(in);
    }
  }
  std::sort(order.begin(), order.end(), [](Batch* a, Batch* b) { return a->op->seq > b->op->seq; });
  // seed: addGrad(ones) (autograd.cpp:57-62) -- onto whatever an earlier, retained backward left there
  bool root_done = false;  // the root's own gradient function ran with the seed (one launch instead of three)
  if (!root->g_dev) {
    alloc_grad(*root, false);
    auto* sop = dynamic_cast<BScalarOp*>(root->op.get());
    bool plain = sop != nullptr && !sop->inputs.empty() && sop->inputs.size() <= 2;
    if (plain)
      for (auto& in : sop->inputs) plain = plain && in->kind == Batch::SCALAR && !in->materialised && in->n == root->n;
    if (plain && sop->inputs.size() == 2 && sop->inputs[0].get() == sop->inputs[1].get()) plain = false;  // add(x, x)
    if (plain) {
      float* g[2] = {nullptr, nullptr};
      float sc[2] = {0.0f, 0.0f};
      int acc[2] = {0, 0};
      for (size_t i = 0; i < sop->inputs.size(); ++i) {
        Batch& in = *sop->inputs[i];
        if (!in.calc_grad) continue;  // (functions.cpp:55-57)
        sc[i] = (sop->kind == SK_NEGATE || (sop->kind == SK_SUBTRACT && i == 1)) ? -1.0f : 1.0f;
        acc[i] = in.g_dev != nullptr;
        if (!in.g_dev) alloc_grad(in, false);
        g[i] = in.g_dev;
      }
      launch_scalar_seed(root->g_dev, g[0], sc[0], acc[0], g[1], sc[1], acc[1], size_t(root->n), rt.stream());
      root_done = true;
    } else {
      launch_fill_f32(root->g_dev, 1.0f, size_t(root->n), rt.stream());
    }
  } else {
    DevMemP ones = rt.alloc(sizeof(float) * size_t(root->n ? root->n : 1));
    launch_fill_f32(ones->as<float>(), 1.0f, size_t(root->n), rt.stream());
    launch_vec_axpby(root->g_dev, ones->as<float>(), nullptr, size_t(root->n), 1.0f, 0.0f, 1, rt.stream());
  }
  // Scalars taken in from the per-graph functions (BFromGraphsOp: their backward runs the PER-GRAPH tape) go last, after
  // the leaves' gradients have moved into the element graphs: nothing on the batch tape waits for them (their inputs are
  // graphs), and the per-graph kernels then find the emission graphs' gradient already in place -- the normaliser's
  // rows, written straight into the caller's tensor -- and accumulate into it (ops_built.cpp: the fused backward's
  // in-place form).  The other order cost the built-lattice step two passes over the [B][T][C] gradient (an
  // accumulate and a copy, 0.64 ms of 7.6 at C3) and a zero fill.
  std::vector<Batch*> from_graphs;
  BackwardPlan plan;
  t_plan = &plan;
  struct RetainScope {
    bool prev;
    explicit RetainScope(bool r) : prev(t_batch_retain) { t_batch_retain = r; }
    ~RetainScope() { t_batch_retain = prev; }
  } retain_scope(retain);
  try {
    for (Batch* b : order) {
      if (!b->g_dev) continue;  // no gradient reached it (a symbolic product never holds one)
      if (root_done && b == root.get()) continue;
      if (auto* f = dynamic_cast<BFromGraphsOp*>(b->op.get())) {
        f->retain = retain;
        from_graphs.push_back(b);
        continue;
      }
      b->op->backward(*b);
    }
    // normalisers that found no sweep to ride with
    for (auto& kv : plan.lin)
      if (!plan.fused.count(kv.first)) static_cast<BFsLinearOp*>(kv.second->op.get())->run(*kv.second);
  } catch (...) {
    t_plan = nullptr;
    throw;
  }
  t_plan = nullptr;
  // leaves whose elements were taken out as graphs: those carry the gradient
  for (Batch* b : seen)
    if ((b->materialised || b->leaf) && b->g_dev && !b->op) push_grads_to_graphs(*b);
  for (Batch* b : from_graphs) b->op->backward(*b);
  if (!retain)
    for (Batch* b : order) {
      if (b != root.get()) {
        b->g_dev = nullptr;  // gradients of intermediate results go with the tape (autograd.cpp:58-64)
        b->g_mem.reset();
      }
      b->op.reset();
      b->tape_cleared = true;
    }
}

// ---- gathers ---------------------------------------------------------------------------------
void batch_items_device(const BatchP& x, void* dev_out) {
  if (x->kind == Batch::SCALAR && !x->materialised) {
    if (static_cast<const void*>(x->v_dev) == dev_out) return;  // (written there in the first place)
    Runtime::get().d2d(dev_out, x->v_dev, sizeof(float) * size_t(x->n));
    return;
  }
  batch_materialise(*x);
  items_device(x->graphs, dev_out);
}
float batch_item_host(const BatchP& x, int i) {
  if (i < 0 || i >= x->n) throw_range("[gtnx_batch_get] element index out of range");
  if (!x->host_vals_valid) {
    x->host_vals.resize(size_t(x->n));
    if (x->host_ev) {  // on its way since the values were launched (batch_prefetch_items)
      Runtime::get().drain_until(x->host_ev);  // (reclaims while the sweep is still running)
      std::memcpy(x->host_vals.data(), x->host_pin->ptr, sizeof(float) * size_t(x->n));
      (void)hipEventDestroy(static_cast<hipEvent_t>(x->host_ev));
      x->host_ev = nullptr;
      x->host_pin.reset();
    } else {
      Runtime::get().d2h_sync(x->host_vals.data(), x->v_dev, sizeof(float) * size_t(x->n));
    }
    x->host_vals_valid = true;
  }
  return x->host_vals[size_t(i)];
}
void batch_prefetch_items(const BatchP& x) {
  if (!x || x->kind != Batch::SCALAR || x->materialised || x->host_vals_valid || x->host_ev || x->n <= 0 || !x->v_dev) return;
  Runtime& rt = Runtime::get();
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  x->host_pin = rt.alloc_pinned(sizeof(float) * size_t(x->n));
  rt.d2h_pinned_async(x->host_pin->ptr, x->v_dev, sizeof(float) * size_t(x->n));
  HIP_CHECK(hipEventRecord(ev, rt.stream()));
  x->host_ev = ev;
}
void batch_items_host(const BatchP& x, float* out) {
  if (x->kind == Batch::SCALAR && !x->materialised) {
    Runtime::get().d2h_sync(out, x->v_dev, sizeof(float) * size_t(x->n));
    return;
  }
  batch_materialise(*x);
  items_host(x->graphs, out);
}
void batch_grads_bind(const BatchP& x, void* dev_out, const int64_t* offsets) {
  if (x->materialised) {
    grads_bind_device(x->graphs, dev_out, offsets);
    return;
  }
  if (x->kind != Batch::LINEAR && x->kind != Batch::SCALAR) return;  // a hint
  int64_t o = 0;
  for (int i = 0; i < x->n; ++i) {
    if (offsets[i] != offsets[0] + o) return;  // not back to back: gathered afterwards
    o += x->elem_size(i);
  }
  if (x->n == 0) return;
  x->dest_mem = std::make_shared<DevMem>();
  x->dest_mem->ptr = dev_out;
  x->dest_mem->borrowed = true;
  x->dest = static_cast<float*>(dev_out) + offsets[0];
}
void batch_grads_device(const BatchP& x, void* dev_out, const int64_t* offsets) {
  if (x->kind == Batch::CTC_TARGETS) batch_materialise(*x);  // exact arc counts live with the element graphs
  if (x->materialised) {
    grads_device(x->graphs, dev_out, offsets);
    return;
  }
  if (!x->g_dev) throw_logic("[gtn::Graph::grad] Gradient not calculated yet.");  // graph.cpp:131-140
  Runtime& rt = Runtime::get();
  std::vector<AxpyArgs> ax;
  int64_t maxn = 0;
  for (int i = 0; i < x->n; ++i) {
    float* dst = static_cast<float*>(dev_out) + offsets[i];
    float* src = x->g_dev + x->g_off[size_t(i)];
    if (dst == src) continue;  // written in place
    const int64_t len = x->g_off[size_t(i) + 1] - x->g_off[size_t(i)];
    ax.push_back({dst, src, len, 1.0f});
    maxn = std::max(maxn, len);
  }
  if (ax.empty()) return;
  DevMemP d = upload_vec(ax);
  launch_axpy_batch(d->as<AxpyArgs>(), int(ax.size()), maxn, /*copy*/ 2, rt.stream());
}

} // namespace gtnx
