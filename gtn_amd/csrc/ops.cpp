// ops.cpp -- batched graph functions + batch-level autograd (see ops.h)
#include "ops.h"

#include "gtn/parallel.h"  // header-only worker pool (no engine dependency)

#include <chrono>
#include <tuple>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <unordered_set>

#include "ops_internal.h"

namespace gtnx {

std::atomic<uint64_t> g_seq_ctr{1};
std::atomic<uint64_t> g_seq_sub{0};


// ---- the constant structure of a scalar result (functions.cpp:26-28, shortest.cpp:183-186)
void init_scalar_structure(Graph& g) {
  Structure& s = *g.s;
  s.N = 2;
  s.A = 1;
  s.nflags = {NF_START, NF_ACCEPT};
  s.start = {0};
  s.accept = {1};
  s.src = {0};
  s.dst = {1};
  s.il = {0};
  s.ol = {0};
  s.host_valid = true;
}

// The same graph as an OP RESULT.  A scalar result is the linear graph with one step and
// one label (node 0 -> node 1, label 0), so it uses the implicit KIND_LINEAR form: no host
// arrays at all -- a training step makes three scalars per utterance, and their seven
// one-element vectors each were a sixth of the step's allocations.
void init_scalar_result(Graph& g) {
  Structure& s = *g.s;
  s.kind = KIND_LINEAR;
  s.M = 1;
  s.C = 1;
  s.N = 2;
  s.A = 1;
  s.host_valid = true;  // implicit
}

// result graph on the tape: calcGrad = any(inputs) (graph.cpp:16-27)
Graph make_output(const std::shared_ptr<OpRecord>& op, int idx, std::vector<Graph> inputs) {
  bool cg = false;
  for (auto& i : inputs) cg |= i.calc_grad();
  Graph out = Graph::make_result(cg);
  if (cg) {
    out.g->op = op;
    out.g->op_idx = idx;
    out.g->has_grad_fn = true;
    out.g->inputs = std::move(inputs);
    for (auto& i : out.g->inputs) i.g->n_consumers++;
  }
  return out;
}


void set_dev_weights(Graph& g, const DevMemP& owner, float* ptr, int64_t n) {
  Weights& w = *g.w;
  w.n = n;
  w.dev_mem = owner;
  w.dev = ptr;
  w.dev_valid = true;
  w.host_valid = false;
  w.zero = false;
  w.staged.reset();
  w.version++;
}

// packed adjacency records of a device-built structure (results of an earlier compose),
// made on first use as an op input; host-built structures get theirs at upload
void ensure_records(Structure& st) {
  if (st.kind != KIND_EXPLICIT || st.dview.out_rec || st.A == 0) return;
  Runtime& rt = Runtime::get();
  st.ensure_full();
  st.rec_mem = rt.alloc(32 * size_t(st.A));
  gtnx_i4* orec = st.rec_mem->as<gtnx_i4>();
  gtnx_i4* irec = orec + st.A;
  launch_build_records(st.dview, orec, irec, rt.stream());
  st.dview.out_rec = orec;
  st.dview.in_rec = irec;
}

// label-sorted view of an explicit structure's records (see Structure::sview_mem); needs ensure_records()
const gtnx_i4* sorted_view(Structure& st, bool key_ol, bool in_lists) {
  if (st.A == 0) return nullptr;
  const int k = key_ol ? 1 : 0;
  if (!st.sview_mem[k] || st.sview_of[k] != st.dview.out_rec) {
    Runtime& rt = Runtime::get();
    st.sview_mem[k] = rt.alloc(32 * size_t(st.A));
    gtnx_i4* ov = st.sview_mem[k]->as<gtnx_i4>();
    launch_sorted_view(st.dview, k, ov, ov + st.A, rt.stream());
    st.sview_of[k] = st.dview.out_rec;
  }
  return st.sview_mem[k]->as<gtnx_i4>() + (in_lists ? st.A : 0);
}

float* grad_dev_ptr(Graph& out) {
  // the incoming delta of an output graph, resident on the device
  Graph& gr = out.grad();
  if (!gr.w->dev_valid || gr.w->host_escaped) {
    std::vector<Weights*> v{gr.w.get()};
    ensure_weights_device_batch(v);
  }
  return gr.w->dev;
}

thread_local bool t_backward_retain = false;
thread_local bool t_reclaim_at_wait = false;
float* through_delta(Graph& out) {
  float* d = grad_dev_ptr(out);
  GradState& gs = *out.g;
  if (!gs.through_acc && !t_backward_retain) return d;  // no earlier pass, no later one: this pass's gradient is all there is
  Runtime& rt = Runtime::get();
  const size_t n = size_t(out.num_arcs());
  if (!gs.through_acc) gs.through_acc = rt.alloc_zero(sizeof(float) * (n ? n : 1));
  float* acc = gs.through_acc->as<float>();
  launch_vec_axpby(acc, d, nullptr, n, 1.0f, 0.0f, /*accumulate=*/1, rt.stream());
  return acc;
}

// ======================================================================
// GradSink
// ======================================================================
void GradSink::flush() {
  if (items.empty()) return;
  GTNX_HOST_T("gradsink.flush");
  Runtime& rt = Runtime::get();
  std::vector<AxpyArgs> ax;
  std::unordered_set<float*> seen;
  bool dup = false;
  int64_t maxn = 0;
  for (auto& it : items) {
    Graph& g = it.g;
    if (!g.is_grad_available()) {
      g.add_grad_device(it.owner, it.ptr, /*adopt=*/true);
      continue;
    }
    g.s->resolve_sizes();  // accumulating into an existing gradient needs the arc count
    Weights& gw = *g.grad().w;
    if (!gw.dev_valid || gw.host_escaped) {
      std::vector<Weights*> v{&gw};
      ensure_weights_device_batch(v);
    }
    gw.host_valid = false;
    gw.host_escaped = false;  // the device copy is the live one now
    gw.version++;
    if (!seen.insert(gw.dev).second) dup = true;
    ax.push_back({gw.dev, it.ptr, g.num_arcs(), 1.0f});
    maxn = std::max<int64_t>(maxn, g.num_arcs());
  }
  if (!ax.empty()) {
    DevMemP d = upload_vec(ax);
    launch_axpy_batch(d->as<AxpyArgs>(), int(ax.size()), maxn, dup ? 1 : 0, rt.stream());
  }
  items.clear();
}

// ======================================================================
// creations
// ======================================================================
Graph make_scalar_graph(float v, bool calc_grad) {
  // creations.cpp:12-18 (labels are epsilon here, unlike op results)
  Graph g(calc_grad);
  init_scalar_structure(g);
  g.s->il = {GTNX_EPSILON};
  g.s->ol = {GTNX_EPSILON};
  g.w->host = {v};
  g.w->n = 1;
  return g;
}

Graph make_linear_graph(int M, int N, bool calc_grad) {
  if (M < 0 || N < 0) throw_invalid("[gtn::linearGraph] negative size");
  Graph g(calc_grad);
  Structure& s = *g.s;
  s.kind = KIND_LINEAR;
  s.M = M;
  s.C = N;
  s.N = int64_t(M) + 1;
  s.A = int64_t(M) * N;
  s.ilabel_sorted = s.olabel_sorted = true;  // creations.cpp:30-31
  s.host_valid = true;                       // implicit
  // zero weights (creations.cpp:24-28), not stored until somebody reads them: an emissions graph gets its
  // weights from setWeights right away, and T*C zeros per utterance were a page-fault storm on the host
  g.w->host_valid = false;
  g.w->zero = true;
  g.w->n = s.A;
  return g;
}

std::vector<Graph> make_linear_graphs_device(int B, int M, int N, bool calc_grad, const void* dev, bool borrow) {
  GTNX_HOST_T("linear_graphs_device.total");
  Runtime& rt = Runtime::get();
  std::vector<Graph> out;
  out.reserve(B);
  const int64_t A = int64_t(M) * N;
  DevMemP arena;
  if (borrow && dev) {  // the weights ARE the caller's tensor (it promises to keep it alive and unchanged)
    arena = std::make_shared<DevMem>();
    arena->ptr = const_cast<void*>(dev);
    arena->bytes = sizeof(float) * size_t(A) * size_t(B);
    arena->borrowed = true;
  } else {
    {
      GTNX_HOST_T("linear_graphs_device.alloc");
      arena = rt.alloc(sizeof(float) * size_t(A) * size_t(B > 0 ? B : 1));
    }
    GTNX_HOST_T("linear_graphs_device.d2d");
    if (dev && A && B) rt.d2d(arena->ptr, dev, sizeof(float) * size_t(A) * size_t(B));
  }
  // the three pieces of all B graphs out of ONE buffer (a graph is a structure, weights and a gradient state: 1536
  // heap blocks per batch of 512, made and later freed one by one on the thread every step waits for); the buffer
  // comes from and goes back to the thread's cache (graph.h: GraphSlabScope -- a fresh 577 KB block per step was 140
  // page faults)
  GraphSlabScope slab_scope(size_t(B > 0 ? B : 0));
  const Runtime::InboxP home = Runtime::home();
  GTNX_HOST_T("linear_graphs_device.graphs");
  for (int b = 0; b < B; ++b) {
    Graph g = Graph::make_result(calc_grad);
    Structure& s = *g.s;
    s.home = home;
    s.device = Runtime::current_device();
    s.kind = KIND_LINEAR;
    s.M = M;
    s.C = N;
    s.N = int64_t(M) + 1;
    s.A = A;
    s.ilabel_sorted = s.olabel_sorted = true;
    set_dev_weights(g, arena, arena->as<float>() + size_t(b) * size_t(A), A);
    out.push_back(std::move(g));
  }
  return out;
}

// ======================================================================
// scalar ops (functions.cpp:18-64)
// ======================================================================
struct ScalarOp : OpRecord {
  ScalarKind kind;
  float scale(int input) const { return kind == SK_NEGATE || (kind == SK_SUBTRACT && input == 1) ? -1.0f : 1.0f; }
  // the gradient function of member `out`: b0 = s0 d, b1 = s1 d into `buf`, handed to the inputs.  seed != null: d is
  // the seed of a backward pass from `out` (one launch for seed and function)
  void backward(std::vector<Member>& ms) override { run(ms, nullptr, nullptr); }
  void run(std::vector<Member>& ms, const DevMemP& seed_mem, float* seed) {
    Runtime& rt = Runtime::get();
    const int n = int(ms.size());
    DevMemP buf = seed ? seed_mem : rt.alloc(sizeof(float) * 2 * size_t(n));
    float* b0 = seed ? seed + 1 : buf->as<float>();
    float* b1 = b0 + n;
    std::vector<ScalarFanArgs> fan(static_cast<size_t>(n));
    GradSink sink;
    for (int m = 0; m < n; ++m) {
      Graph& out = ms[m].out;
      fan[size_t(m)] = {seed ? nullptr : grad_dev_ptr(out), seed, b0 + m, nullptr};
      sink.add(out.g->inputs[0], buf, b0 + m);
      if (kind != SK_NEGATE) {
        // subtract only feeds input 1 when it wants a gradient (functions.cpp:55-57)
        fan[size_t(m)].o1 = b1 + m;
        sink.add(out.g->inputs[1], buf, b1 + m);
      }
    }
    if (n == 1) {
      launch_scalar_fan_one(fan[0], scale(0), scale(1), rt.stream());
    } else {
      DevMemP d = upload_vec(fan);
      launch_scalar_fan(d->as<ScalarFanArgs>(), n, scale(0), scale(1), rt.stream());
    }
    sink.flush();
  }
};

std::vector<Graph> op_scalar(ScalarKind k, std::vector<Graph>& a, std::vector<Graph>& b) {
  GraphSlabScope slab_scope(std::max(a.size(), b.size()));  // the results' pieces out of one allocation (graph.h)
  static const char* msg1[] = {"[gtn::negate] input must have only one arc",
                               "[gtn::add] inputs must have only one arc",
                               "[gtn::subtract] inputs must have only one arc"};
  const bool binary = k != SK_NEGATE;
  const size_t n = binary ? std::max(a.size(), b.size()) : a.size();
  std::vector<Graph> outs;
  if (n == 0) return outs;
  Runtime& rt = Runtime::get();
  std::vector<Weights*> ws;
  for (size_t i = 0; i < n; ++i) {
    const Graph& x = bcast(a, n, i);
    if (x.num_arcs() != 1) throw_logic(msg1[k]);
    ws.push_back(x.w.get());
    if (binary) {
      const Graph& y = bcast(b, n, i);
      if (y.num_arcs() != 1) throw_logic(msg1[k]);
      ws.push_back(y.w.get());
    }
  }
  ensure_weights_device_batch(ws);
  auto op = std::make_shared<ScalarOp>();
  op->kind = k;
  op->seq = next_seq();
  DevMemP res = rt.alloc(sizeof(float) * n);
  std::vector<ScalarArgs> args(n);
  for (size_t i = 0; i < n; ++i) {
    const Graph& x = bcast(a, n, i);
    std::vector<Graph> ins{x};
    args[i].a = x.w->dev;
    args[i].b = nullptr;
    if (binary) {
      const Graph& y = bcast(b, n, i);
      ins.push_back(y);
      args[i].b = y.w->dev;
    }
    args[i].out = res->as<float>() + i;
    Graph out = make_output(op, int(i), std::move(ins));
    init_scalar_result(out);
    set_dev_weights(out, res, res->as<float>() + i, 1);
    outs.push_back(std::move(out));
  }
  const float sa = k == SK_NEGATE ? -1.0f : 1.0f, sb = k == SK_SUBTRACT ? -1.0f : 1.0f;
  if (n == 1) {  // one utterance through the per-graph functions: the record travels with the launch, and the value
                 // (a loss, as a rule) goes to pinned host memory too: item() will not need a copy
    Weights& w = *outs[0].w;
    w.mirror = rt.mirror_slot();
    w.mirror_version = w.version;
    args[0].mirror = w.mirror.ptr;
    launch_scalar_combine_one(args[0], sa, sb, rt.stream());
  } else {
    DevMemP d = upload_vec(args);
    launch_scalar_combine(d->as<ScalarArgs>(), int(n), sa, sb, rt.stream());
  }
  return outs;
}

// backward() from the result of a scalar op that holds no gradient yet (the loss of a criterion written with the
// per-graph functions: subtract(forwardScore(emissions), forwardScore(intersect(...)))): the seed (autograd.cpp:57-62)
// and the op's own gradient function in one launch instead of a fill, two tables and two launches
static std::shared_ptr<ScalarOp> seed_scalar_root(Graph& root) {
  if (!root.calc_grad() || root.is_grad_available() || root.num_arcs() != 1 || !root.g->has_grad_fn) return nullptr;
  auto op = std::dynamic_pointer_cast<ScalarOp>(root.g->op);
  if (!op || root.g->inputs.empty() || root.g->inputs.size() != (op->kind == SK_NEGATE ? 1u : 2u)) return nullptr;
  Runtime& rt = Runtime::get();
  DevMemP buf = rt.alloc(sizeof(float) * 3);  // [seed, input 0's share, input 1's share]
  float* seed = buf->as<float>();
  root.add_grad_device(buf, seed, /*adopt=*/true);
  std::vector<Member> ms{{root.g->op_idx, root}};
  op->run(ms, buf, seed);
  return op;  // (op_backward skips this record's turn for `root`: no mark on the record -- other outputs of a vector
              //  op may be differentiated by other threads at the same time)
}


// turn a symbolic product into the ordinary materialised one, in place
void realize(Graph& g) {
  if (!g.s || !g.s->lazy) return;
  LazyProduct lp = *g.s->lazy;
  g.s->lazy.reset();
  std::vector<Graph> av{lp.chain_side == 1 ? lp.chain : lp.fixed}, bv{lp.chain_side == 1 ? lp.fixed : lp.chain};
  std::vector<Graph> r = op_compose_impl(av, bv, lp.intersect, false);
  Graph& real = r[0];
  real.s->resolve_sizes();  // the pieces move into `g` below: they must be final
  Structure& d = *g.s;
  Structure& o = *real.s;
  d.kind = o.kind;
  d.N = o.N;
  d.A = o.A;
  d.M = o.M;
  d.C = o.C;
  d.ilabel_sorted = o.ilabel_sorted;
  d.olabel_sorted = o.olabel_sorted;
  d.host_valid = o.host_valid;
  d.src = std::move(o.src);
  d.dst = std::move(o.dst);
  d.il = std::move(o.il);
  d.ol = std::move(o.ol);
  d.nflags = std::move(o.nflags);
  d.start = std::move(o.start);
  d.accept = std::move(o.accept);
  d.csr_valid = false;
  d.dev_valid = o.dev_valid;
  d.dev_mem = o.dev_mem;
  d.dview = o.dview;
  d.rec_mem = o.rec_mem;
  d.sched = o.sched;
  d.partial = o.partial;
  Weights& dw = *g.w;
  Weights& ow = *real.w;
  dw.n = ow.n;
  dw.host = std::move(ow.host);
  dw.host_valid = ow.host_valid;
  dw.host_escaped = false;
  dw.version++;
  dw.dev_mem = ow.dev_mem;
  dw.dev = ow.dev;
  dw.dev_valid = ow.dev_valid;
  if (d.sched && d.sched->in_w_of == real.w.get()) {
    d.sched->in_w_of = g.w.get();
    d.sched->in_w_version = dw.version;
  }
  if (g.g->op) {
    // The tape now runs through the real compose record -- filed where the symbolic one was:
    // consumers already recorded against this product (forwardScore of it, say) must still run
    // first in the reverse sweep.
    const uint64_t old = g.g->op->seq;
    g.g->op = real.g->op;
    g.g->op_idx = real.g->op_idx;
    g.g->op->seq = old + 1 + g_seq_sub.fetch_add(1) % (kSeqStride - 2);
    if (d.sched && d.sched->producer_seq) d.sched->producer_seq = g.g->op->seq;
  }
}

// ======================================================================
// user-defined ops (Graph(GradFunc, inputs), graph.h:76-78)
// ======================================================================
struct UserOp : OpRecord {
  gtnx_grad_fn fn;
  void* ctx;
  void (*ctx_free)(void*);
  ~UserOp() override {
    if (ctx_free) ctx_free(ctx);
  }
  void backward(std::vector<Member>& ms) override {
    for (auto& m : ms) {
      std::vector<gtnx_graph_t> hs;
      for (auto& in : m.out.g->inputs) hs.push_back(reinterpret_cast<gtnx_graph_t>(&in));
      gtnx_status_t st = fn(ctx, hs.data(), int(hs.size()), reinterpret_cast<gtnx_graph_t>(&m.out.grad()));
      if (st != GTNX_OK) throw Error(st, std::string("user gradFunc failed: ") + gtnx_last_error());
    }
  }
};

Graph make_user_op(std::vector<Graph>& inputs, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*)) {
  auto op = std::make_shared<UserOp>();
  op->fn = fn;
  op->ctx = ctx;
  op->ctx_free = ctx_free;
  op->seq = next_seq();
  Graph out = make_output(op, 0, inputs);
  if (!fn) out.g->has_grad_fn = false;
  return out;
}

void set_user_grad_fn(Graph& g, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*)) {
  // Graph::setGradFunc (graph.h:293-297): a no-op unless the graph wants gradients
  if (!g.calc_grad()) {
    if (ctx_free) ctx_free(ctx);
    return;
  }
  auto op = std::make_shared<UserOp>();
  op->fn = fn;
  op->ctx = ctx;
  op->ctx_free = ctx_free;
  op->seq = next_seq();
  g.g->op = op;
  g.g->op_idx = 0;
  g.g->has_grad_fn = fn != nullptr;
}

// ======================================================================
// backward (autograd.cpp:17-67)
// ======================================================================
using Tape = std::map<uint64_t, std::pair<std::shared_ptr<OpRecord>, std::vector<Member>>, std::greater<uint64_t>>;
// reachable graphs grouped by producing record; throws (before anything has been changed) when part of the
// tape is gone already -- autograd.cpp:42-45
void collect_tape(std::vector<Graph>& roots, Tape& tape) {
  std::unordered_set<GradState*> seen;
  std::vector<Graph> stack(roots.begin(), roots.end());
  while (!stack.empty()) {
    Graph g = stack.back();
    stack.pop_back();
    if (!seen.insert(g.g.get()).second) continue;
    for (auto& in : g.g->inputs) stack.push_back(in);
    if (g.g->has_grad_fn) {
      if (!g.g->op || g.g->inputs.empty())  // autograd.cpp:42-45
        throw_invalid("[autograd::backward] Cannot Backward twice without retaining the graph.");
      auto& slot = tape[g.g->op->seq];
      slot.first = g.g->op;
      slot.second.push_back({g.g->op_idx, g});
    }
  }
}

void backward_validate(Graph& root) {
  std::vector<Graph> roots{root};
  Tape tape;
  collect_tape(roots, tape);
}

void op_backward(std::vector<Graph>& roots, Graph* grad, bool retain, bool seed) {
  GTNX_HOST_T("backward.total");
  Runtime& rt = Runtime::get();
  for (auto& r : roots) realize(r);
  std::shared_ptr<ScalarOp> seeded_op;  // the root's record, when its gradient function ran together with the seed
  // ---- seed (autograd.cpp:57-67); seed == false: the roots hold their deltas already (batch.cpp)
  if (!seed) {
  } else if (grad) {
    for (auto& r : roots) {
      if (!r.calc_grad()) continue;
      if (grad->num_arcs() != r.num_arcs()) throw_logic("[Graph::addGrad] Invalid grad size.");
      std::vector<Weights*> v{grad->w.get()};
      ensure_weights_device_batch(v);
      r.add_grad_device(grad->w->dev_mem, grad->w->dev, /*adopt=*/false);
    }
  } else if (roots.size() == 1 && (seeded_op = seed_scalar_root(roots[0]))) {
  } else {
    size_t tot = 0;
    for (auto& r : roots) tot += size_t(r.num_arcs());
    DevMemP ones = rt.alloc(sizeof(float) * (tot ? tot : 1));
    launch_fill_f32(ones->as<float>(), 1.0f, tot, rt.stream());
    GradSink sink;
    size_t off = 0;
    for (auto& r : roots) {
      sink.add(r, ones, ones->as<float>() + off);
      off += size_t(r.num_arcs());
    }
    sink.flush();
  }
  // ---- collect the tape (after the seed, like autograd.cpp:57-67: a second backward without retain throws
  // with the seed added; region.cpp validates the roots of a gathered backward beforehand)
  Tape tape;
  collect_tape(roots, tape);
  struct RetainScope {
    bool prev;
    explicit RetainScope(bool r) : prev(t_backward_retain) { t_backward_retain = r; }
    ~RetainScope() { t_backward_retain = prev; }
  } retain_scope(retain);
  // ---- reverse sweep: creation order is a topological order
  ChainGradPlan plan;
  struct PlanScope {
    ChainGradPlan* prev;
    explicit PlanScope(ChainGradPlan* p) : prev(t_chain_plan) { t_chain_plan = p; }
    ~PlanScope() { t_chain_plan = prev; }
  } plan_scope(&plan);
  for (auto& kv : tape) {
    auto& members = kv.second.second;
    if (!kv.second.first->joins_chain_plan()) flush_chain_plan();  // anything else may look at the chains' gradients
    std::sort(members.begin(), members.end(), [](const Member& a, const Member& b) { return a.idx < b.idx; });
    // a consumer of a symbolic product pushes its gradient straight into the product's inputs
    // (grad_propagated); if that was the only consumer there is nothing left for this record to do
    // for that member -- also when the product has been built in the meantime
    members.erase(std::remove_if(members.begin(), members.end(),
                                 [](const Member& m) { return m.out.g->grad_propagated && !m.out.is_grad_available() && !m.out.s->lazy; }),
                  members.end());
    for (auto& m : members)
      if (!(m.out.s->lazy && m.out.g->grad_propagated))
        (void)m.out.grad();  // throws "Gradient not calculated yet." like autograd.cpp:46
    const bool ran_with_seed = seeded_op && kv.second.first.get() == seeded_op.get() && members.size() == 1 &&
                               members[0].out.g == roots[0].g;
    if (!members.empty() && !ran_with_seed) kv.second.first->backward(members);
    if (!retain) {
      // autograd.cpp:47-50: the tape (inputs, saved forward state) goes away with
      // backward; the objects themselves are reclaimed at the next sync point
      auto* dead_ops = new std::vector<std::shared_ptr<OpRecord>>();
      auto* dead_inputs = new std::vector<std::vector<Graph>>();
      dead_ops->reserve(members.size());
      dead_inputs->reserve(members.size());
      for (auto& m : members) {
        for (auto& in : m.out.g->inputs) in.g->n_consumers--;
        dead_inputs->push_back(std::move(m.out.g->inputs));
        m.out.g->inputs.clear();
        dead_ops->push_back(std::move(m.out.g->op));
        m.out.g->op.reset();
      }
      rt.defer_delete(dead_ops, [](void* q) { delete static_cast<std::vector<std::shared_ptr<OpRecord>>*>(q); });
      rt.defer_delete(dead_inputs, [](void* q) { delete static_cast<std::vector<std::vector<Graph>>*>(q); });
    }
  }
  flush_chain_plan();
}

// ======================================================================
// batched item() / grad gathering
// ======================================================================
void items_host(std::vector<Graph>& gs, float* out) {
  const size_t n = gs.size();
  if (n == 0) return;
  std::vector<const float*> ptrs;
  std::vector<size_t> which;
  for (size_t i = 0; i < n; ++i) {
    if (gs[i].num_arcs() != 1)
      throw_invalid("[Graph::item] Cannot convert Graph with more than 1 arc to a scalar.");
    Weights& w = *gs[i].w;
    if (!w.host_valid && !w.dev_valid) w.ensure_host();  // unwritten zeros / weights staged by a parallelMap region
    if (w.host_valid) {
      out[i] = w.host[0];
    } else {
      ptrs.push_back(w.dev);
      which.push_back(i);
    }
  }
  if (ptrs.empty()) return;
  Runtime& rt = Runtime::get();
  DevMemP dp = upload_vec(ptrs);
  DevMemP dense = rt.alloc(sizeof(float) * ptrs.size());
  launch_gather_scalars(dp->as<const float*>(), dense->as<float>(), int(ptrs.size()), rt.stream());
  std::vector<float> host(ptrs.size());
  rt.d2h_sync(host.data(), dense->ptr, sizeof(float) * ptrs.size());
  for (size_t k = 0; k < which.size(); ++k) {
    out[which[k]] = host[k];
    Weights& w = *gs[which[k]].w;
    w.host.assign(1, host[k]);
    w.host_valid = true;
  }
}

void items_device(std::vector<Graph>& gs, void* dev_out) {
  const size_t n = gs.size();
  if (n == 0) return;
  std::vector<Weights*> ws;
  for (auto& g : gs) {
    if (g.num_arcs() != 1)
      throw_invalid("[Graph::item] Cannot convert Graph with more than 1 arc to a scalar.");
    ws.push_back(g.w.get());
  }
  ensure_weights_device_batch(ws);
  std::vector<const float*> ptrs;
  for (auto& g : gs) ptrs.push_back(g.w->dev);
  Runtime& rt = Runtime::get();
  DevMemP dp = upload_vec(ptrs);
  launch_gather_scalars(dp->as<const float*>(), static_cast<float*>(dev_out), int(n), rt.stream());
}


void grads_device(std::vector<Graph>& gs, void* dev_out, const int64_t* offsets) {
  const size_t n = gs.size();
  if (n == 0) return;
  Runtime& rt = Runtime::get();
  std::vector<Weights*> ws;
  for (auto& g : gs) ws.push_back(g.grad().w.get());
  ensure_weights_device_batch(ws);
  std::vector<AxpyArgs> ax;
  ax.reserve(n);
  int64_t maxn = 0;
  for (size_t i = 0; i < n; ++i) {
    float* dst = static_cast<float*>(dev_out) + offsets[i];
    if (gs[i].grad().w->dev == dst) continue;  // written in place (grads_bind_device)
    ax.push_back({dst, gs[i].grad().w->dev, gs[i].num_arcs(), 1.0f});
    maxn = std::max(maxn, gs[i].num_arcs());
  }
  if (ax.empty()) return;
  DevMemP d = upload_vec(ax);
  launch_axpy_batch(d->as<AxpyArgs>(), int(ax.size()), maxn, /*copy mode*/ 2, rt.stream());
}

void grads_bind_device(std::vector<Graph>& gs, void* dev_out, const int64_t* offsets) {
  if (gs.empty()) return;
  auto mem = std::make_shared<DevMem>();
  mem->ptr = dev_out;
  mem->borrowed = true;
  for (size_t i = 0; i < gs.size(); ++i) {
    gs[i].g->grad_dest_mem = mem;
    gs[i].g->grad_dest = static_cast<float*>(dev_out) + offsets[i];
  }
}

} // namespace gtnx
