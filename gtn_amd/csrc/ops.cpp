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

namespace gtnx {

namespace {
// Records are ordered by creation; sequence numbers are 2^20 apart so that a record made later
// to stand in for an older one (realize()) can be filed right behind it.
constexpr uint64_t kSeqStride = uint64_t(1) << 20;
std::atomic<uint64_t> g_seq_ctr{1};
std::atomic<uint64_t> g_seq_sub{0};
inline uint64_t next_seq() { return g_seq_ctr.fetch_add(1) * kSeqStride; }


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
  Graph out(cg);
  if (cg) {
    out.g->op = op;
    out.g->op_idx = idx;
    out.g->has_grad_fn = true;
    out.g->inputs = std::move(inputs);
    for (auto& i : out.g->inputs) i.g->n_consumers++;
  }
  return out;
}

// lazy chain products (defined further down)
std::vector<Graph> lazy_shortest_distance(std::vector<Graph>& gs, bool tropical);
std::vector<Graph> lazy_viterbi_path(std::vector<Graph>& gs);
std::shared_ptr<OpRecord> make_lazy_compose_op();
bool lazy_shape_ok(const Structure& chain, const Structure& fixed);
bool lazy_pair_shape_ok(const Structure& chain, Structure& fixed);
bool band_shape_ok(const Structure& chain, Structure& fixed, bool chain_first);
void band_prepare(const std::vector<Graph*>& fixed, const std::vector<uint8_t>& chain_first);

template <class T>
const T& bcast(const std::vector<T>& v, size_t n, size_t i) {
  if (v.size() == n) return v[i];
  if (v.size() == 1) return v[0];
  // parallel_map.h:85-88
  throw_runtime("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
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
static const gtnx_i4* sorted_view(Structure& st, bool key_ol, bool in_lists) {
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
} // namespace

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
    arena = rt.alloc(sizeof(float) * size_t(A) * size_t(B > 0 ? B : 1));
    GTNX_HOST_T("linear_graphs_device.d2d");
    if (dev && A && B) rt.d2d(arena->ptr, dev, sizeof(float) * size_t(A) * size_t(B));
  }
  for (int b = 0; b < B; ++b) {
    Graph g(calc_grad);
    Structure& s = *g.s;
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
namespace {
struct ScalarOp : OpRecord {
  ScalarKind kind;
  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    const int n = int(ms.size());
    DevMemP buf = rt.alloc(sizeof(float) * 2 * size_t(n));
    float* b0 = buf->as<float>();
    float* b1 = b0 + n;
    std::vector<ScalarArgs> a0, a1;
    GradSink sink;
    for (int m = 0; m < n; ++m) {
      Graph& out = ms[m].out;
      float* d = grad_dev_ptr(out);
      a0.push_back({d, nullptr, b0 + m});
      sink.add(out.g->inputs[0], buf, b0 + m);
      if (kind != SK_NEGATE) {
        // subtract only feeds input 1 when it wants a gradient (functions.cpp:55-57)
        a1.push_back({d, nullptr, b1 + m});
        sink.add(out.g->inputs[1], buf, b1 + m);
      }
    }
    DevMemP d0 = upload_vec(a0);
    launch_scalar_combine(d0->as<ScalarArgs>(), n, kind == SK_NEGATE ? -1.0f : 1.0f, 0.0f, rt.stream());
    if (!a1.empty()) {
      DevMemP d1 = upload_vec(a1);
      launch_scalar_combine(d1->as<ScalarArgs>(), int(a1.size()), kind == SK_SUBTRACT ? -1.0f : 1.0f, 0.0f,
                            rt.stream());
    }
    sink.flush();
  }
};
} // namespace

std::vector<Graph> op_scalar(ScalarKind k, std::vector<Graph>& a, std::vector<Graph>& b) {
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
  DevMemP d = upload_vec(args);
  launch_scalar_combine(d->as<ScalarArgs>(), int(n), k == SK_NEGATE ? -1.0f : 1.0f,
                        k == SK_SUBTRACT ? -1.0f : 1.0f, rt.stream());
  return outs;
}

// ======================================================================
// shortest distance: forwardScore / viterbiScore (functions.cpp:320-326)
// ======================================================================
namespace {

// effective schedule view for this call: in_w only while it still matches the weights
DSched sched_view(Graph& g, bool need_full = true) {
  if (need_full) g.s->ensure_full();  // in_arc (= in_list) may not have been written yet
  Schedule& sc = *g.s->sched;
  DSched v = sc.view;
  v.in_w = (sc.in_w && sc.in_w_of == g.w.get() && sc.in_w_version == g.w->version) ? sc.in_w : nullptr;
  return v;
}

// ---- gradient launches of one backward() over the same emission chains, gathered before they go:
// forwardScore(emissions) contributes dn * softmax(row), forwardScore(target o emissions) the node
// posteriors; registered here by their records and launched together (flush_chain_plan) the
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
thread_local ChainGradPlan* t_chain_plan = nullptr;
void flush_chain_plan();

struct LinearSdOp : OpRecord {
  bool tropical;
  bool joins_chain_plan() const override { return !tropical; }
  void backward(std::vector<Member>& all) override {
    // members whose chain has its row log-sum-exps at hand wait for the sweep over the same chain
    std::vector<Member> ms;
    std::vector<Graph> ins;
    for (auto& m : all) {
      Graph& in = m.out.g->inputs[0];
      const NormCache* nc = (!tropical && t_chain_plan && in.calc_grad()) ? in.w->valid_norm_cache() : nullptr;
      if (nc && nc->rowlse && !t_chain_plan->lin.count(in.w.get())) {
        ChainGradPlan::Lin l{m, in, grad_dev_ptr(m.out), nc->rowlse, nc->mem, false};
        t_chain_plan->lin.emplace(in.w.get(), std::move(l));
      } else {
        ms.push_back(m);
        ins.push_back(in);
      }
    }
    if (!ms.empty()) run_now(ms, ins);
  }
  void run_now(std::vector<Member>& ms, std::vector<Graph>& ins) {
    Runtime& rt = Runtime::get();
    const int n = int(ms.size());
    std::vector<Weights*> ws;
    size_t total = 0;
    for (auto& in : ins) {
      ws.push_back(in.w.get());
      total += size_t(in.num_arcs());
    }
    ensure_weights_device_batch(ws);
    DevMemP grads = rt.alloc(sizeof(float) * (total ? total : 1));
    std::vector<LinArgs> args(n);
    GradSink sink;
    size_t off = 0;
    int maxM = 0;
    double bytes = 0;
    std::unordered_set<GradState*> seen_in;
    for (int i = 0; i < n; ++i) {
      Graph& in = ins[i];
      LinArgs& a = args[i];
      a.w = in.w->dev;
      a.M = in.s->M;
      a.C = in.s->C;
      a.out_score = nullptr;
      a.partial = nullptr;
      a.delta = grad_dev_ptr(ms[i].out);
      a.accumulate = 0;
      // a gradient that already lives on the device is updated in place (one pass
      // instead of write + read-modify-write); addGrad semantics, graph.cpp:108-129
      const bool first_use = seen_in.insert(in.g.get()).second;
      if (first_use && in.calc_grad() && in.is_grad_available()) {
        Weights& gw = *in.grad().w;
        if (gw.dev_valid && !(gw.host_escaped && gw.host_valid) && gw.n == in.num_arcs()) {
          a.grad = gw.dev;
          a.accumulate = 1;
          gw.host_valid = false;
          gw.version++;
          off += 0;
          maxM = std::max(maxM, a.M);
          bytes += 12.0 * double(in.num_arcs());
          continue;
        }
      }
      a.grad = grads->as<float>() + off;
      sink.add(in, grads, a.grad);
      off += size_t(in.num_arcs());
      maxM = std::max(maxM, a.M);
      bytes += 8.0 * double(in.num_arcs());
    }
    DevMemP d = upload_vec(args);
    {
      GTNX_PROF("linear_forward_grad", bytes);
      bool vec_rows = true;
      for (auto& a : args)
        vec_rows = vec_rows && a.C % 4 == 0 && a.C <= 1024 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.grad) & 15) == 0;
      (void)maxM;
      launch_linear_backward(d->as<LinArgs>(), n, tropical ? 1 : 0, vec_rows ? 1 : 0, rt.stream());
    }
    sink.flush();
  }
};

struct SdOp : OpRecord {
  int mode;
  DevMemP arena;  // scores / argmax / results of the whole batch
  struct Saved {
    std::shared_ptr<Schedule> sched;
    float* scores;
    int* argmax;
    SdResult* result;
  };
  std::vector<Saved> saved;

  void backward(std::vector<Member>& ms) override {
    GTNX_HOST_T("backward.sd_op");
    Runtime& rt = Runtime::get();
    const int n = int(ms.size());
    std::vector<Weights*> ws;
    for (auto& m : ms) ws.push_back(m.out.g->inputs[0].w.get());
    ensure_weights_device_batch(ws);
    // one arena: arc grads (A) + node grads (P) per member
    size_t bytes = 0;
    std::vector<size_t> off_a(n), off_n(n);
    bool need_zero = false;
    for (int i = 0; i < n; ++i) {
      Graph& in = ms[i].out.g->inputs[0];
      if (in.s->deferred && mode != SD_LOG) in.s->resolve_sizes();
      const Saved& sv = saved[ms[i].idx];
      off_a[i] = bytes;
      bytes = align_up(bytes + 4 * size_t(in.s->bound_arcs()), 256);
      off_n[i] = bytes;
      bytes = align_up(bytes + 4 * size_t(sv.sched->view.P), 256);
      need_zero |= !sv.sched->all_written;
    }
    DevMemP g = need_zero ? rt.alloc_zero(bytes) : rt.alloc(bytes ? bytes : 1);
    // narrow-lattice kernel eligibility (whole batch)
    bool narrow = mode == SD_LOG;
    {
      int64_t tot_levels = 0;
      for (int i = 0; i < n; ++i) {
        const Schedule& sc = *saved[ms[i].idx].sched;
        narrow = narrow && (sc.view.flags & SCHED_OUT_IDENTITY) && sc.max_level_arcs <= sd_narrow_tmp_cap() &&
                 sc.max_level_width <= sd_narrow_node_cap() && sc.max_reach <= sd_narrow_ring_backward();
        tot_levels += sc.view.L;
      }
      narrow = narrow && tot_levels >= 32 * int64_t(n);
    }
    if (!narrow)
      for (int i = 0; i < n; ++i) ms[i].out.g->inputs[0].s->ensure_full();  // (waits for deferred sizes too)
    // Fused compose-gradient scatter: when every lattice of the batch is a layered
    // product with one linear chain, produced by a compose whose ONLY consumer is
    // this forwardScore and which holds no gradient yet, the kernel sums the arc
    // gradients into the compose inputs itself and the compose record's own
    // backward (compose.cpp:496-518) is skipped for these members.  (With a
    // gradient already present -- a second backward over a retained tape -- the
    // reference re-scatters the ACCUMULATED delta, so that case stays unfused.)
    int cap_f = 0, cap_c = 0;
    sd_narrow_fuse_caps(&cap_f, &cap_c);
    bool fuse = narrow && !getenv("GTNX_NO_FUSED_SCATTER");
    for (int i = 0; i < n && fuse; ++i) {
      Graph& in = ms[i].out.g->inputs[0];
      const Schedule& sc = *saved[ms[i].idx].sched;
      fuse = sc.chain_side != 0 && in.g->op && in.g->op->seq == sc.producer_seq && in.g->inputs.size() == 2 &&
             in.g->n_consumers == 1 && in.calc_grad() && !in.is_grad_available() && sc.fixed_A <= cap_f &&
             sc.chain_C <= cap_c && in.s->sched.get() == &sc;
    }
    DevMemP fg;
    std::vector<size_t> off_f(n), off_c(n);
    if (fuse) {
      size_t fb = 0;
      for (int i = 0; i < n; ++i) {
        auto& cin = ms[i].out.g->inputs[0].g->inputs;
        const Schedule& sc = *saved[ms[i].idx].sched;
        Graph& fixed = cin[sc.chain_side == 1 ? 1 : 0];
        Graph& chain = cin[sc.chain_side == 1 ? 0 : 1];
        off_f[i] = fb;
        if (fixed.calc_grad()) fb = align_up(fb + 4 * size_t(fixed.num_arcs()), 256);
        off_c[i] = fb;
        if (chain.calc_grad()) fb = align_up(fb + 4 * size_t(chain.num_arcs()), 256);
      }
      fg = rt.alloc_zero(fb ? fb : 1);
    }
    std::vector<SdArgs> args(n);
    GradSink sink;
    std::unordered_set<GradState*> fused_chain_seen;
    int64_t tot_out = 0, tot_p = 0;
    double alg = 0;
    for (int i = 0; i < n; ++i) {
      Graph& in = ms[i].out.g->inputs[0];
      const Saved& sv = saved[ms[i].idx];
      SdArgs& a = args[i];
      std::memset(&a, 0, sizeof(a));
      a.s = sv.sched->view;
      a.s.in_w = nullptr;
      a.w = in.w->dev;
      a.scores = sv.scores;
      a.argmax = sv.argmax;
      a.result = sv.result;
      a.out_score = nullptr;
      a.delta = grad_dev_ptr(ms[i].out);
      a.arc_grad = g->as<float>(off_a[i]);
      a.node_grad = g->as<float>(off_n[i]);
      a.chunk_levels = std::max(1, std::min(sd_narrow_tmp_cap() / std::max(sv.sched->max_level_arcs, 1),
                                            sd_narrow_node_cap() / std::max(sv.sched->max_level_width, 1)));
      sink.add(in, g, a.arc_grad);
      if (fuse) {
        const Schedule& sc = *sv.sched;
        auto& cin = in.g->inputs;
        Graph& fixed = cin[sc.chain_side == 1 ? 1 : 0];
        Graph& chain = cin[sc.chain_side == 1 ? 0 : 1];
        a.gi_fixed = sc.gi_fixed;
        a.gi_chain = sc.gi_chain;
        a.chain_C = sc.chain_C;
        a.fixed_A = int(sc.fixed_A);
        a.chain_A = int(chain.num_arcs());
        a.grad_fixed = fixed.calc_grad() ? fg->as<float>(off_f[i]) : nullptr;
        a.grad_chain = chain.calc_grad() ? fg->as<float>(off_c[i]) : nullptr;
        a.chunk_levels = std::max(1, std::min(a.chunk_levels, cap_c / std::max(sc.chain_C, 1)));
        if (a.grad_fixed) sink.add(fixed, fg, a.grad_fixed);
        // a chain that already holds a device gradient (e.g. from forwardScore(emissions),
        // run earlier in the sweep) is accumulated into in place: one pass, no axpy
        bool in_place = false;
        if (a.grad_chain && chain.is_grad_available() && fused_chain_seen.insert(chain.g.get()).second) {
          Weights& gw = *chain.grad().w;
          if (gw.dev_valid && !(gw.host_escaped && gw.host_valid) && gw.n == chain.num_arcs()) {
            a.grad_chain = gw.dev;
            a.chain_accumulate = 1;
            gw.host_valid = false;
            gw.version++;
            in_place = true;
          }
        }
        if (a.grad_chain && !in_place) sink.add(chain, fg, a.grad_chain);
        in.g->grad_propagated = true;
        alg += 4.0 * double(fixed.num_arcs() + chain.num_arcs());
      }
      tot_out += sv.sched->n_out;
      tot_p += sv.sched->view.P;
      const char* pname = mode == SD_LOG ? "forward_score_grad" : "viterbi_score_grad";
      if (in.s->deferred) {
        if (rt.prof_on()) in.s->deferred->prof.push_back({pname, fuse ? 20.0 : 12.0, 12.0, in.s->deferred_idx});
      } else {
        alg += (fuse ? 20.0 : 12.0) * double(in.num_arcs()) + 12.0 * double(sv.sched->view.P);
      }
      if (narrow) {
        a.dyn_out = sv.sched->dyn_out;
        a.dyn_counts = sv.sched->dyn_counts;
      }
    }
    DevMemP d = upload_vec(args);
    {
      GTNX_PROF(mode == SD_LOG ? "forward_score_grad" : "viterbi_score_grad", alg);
      int fuse_lds = 0;
      if (fuse)
        for (auto& a : args) fuse_lds = std::max(fuse_lds, 4 * std::max(a.chunk_levels, 1) * a.chain_C);
      launch_sd_backward(d->as<SdArgs>(), n, mode, narrow ? (fuse ? 2 : 1) : 0,
                         int(tot_p ? (tot_out * 16) / tot_p : 0), rt.stream(), fuse_lds);
    }
    sink.flush();
  }
};

} // namespace

std::vector<Graph> op_shortest_distance(std::vector<Graph>& gs, bool tropical) {
  GTNX_HOST_T("shortest_distance.total");
  const size_t n = gs.size();
  std::vector<Graph> outs(n, Graph(false));
  if (n == 0) return outs;
  {
    // symbolic chain products take the time-synchronous kernels; the rest go on below
    std::vector<Graph> lz, rest;
    std::vector<size_t> lz_i, rest_i;
    for (size_t i = 0; i < n; ++i) {
      if (gs[i].s->lazy) { lz.push_back(gs[i]); lz_i.push_back(i); }
      else { rest.push_back(gs[i]); rest_i.push_back(i); }
    }
    if (!lz.empty()) {
      std::vector<Graph> lo = lazy_shortest_distance(lz, tropical);
      for (size_t k = 0; k < lz.size(); ++k) outs[lz_i[k]] = lo[k];
      if (!rest.empty()) {
        std::vector<Graph> ro = op_shortest_distance(rest, tropical);
        for (size_t k = 0; k < rest.size(); ++k) outs[rest_i[k]] = ro[k];
      }
      return outs;
    }
  }
  Runtime& rt = Runtime::get();
  std::vector<int> lin, exp;
  for (size_t i = 0; i < n; ++i) (gs[i].s->kind == KIND_LINEAR ? lin : exp).push_back(int(i));
  std::vector<Weights*> ws;
  for (auto& g : gs) ws.push_back(g.w.get());
  ensure_weights_device_batch(ws);

  // ---- linear-chain members: streaming row reductions
  if (!lin.empty()) {
    const int m = int(lin.size());
    auto op = std::make_shared<LinearSdOp>();
    op->tropical = tropical;
    op->seq = next_seq();
    DevMemP res = rt.alloc(sizeof(float) * size_t(m) * 9);
    float* scal = res->as<float>();
    float* partial = scal + m;
    std::vector<LinArgs> args(m);
    int maxM = 0;
    double bytes = 0;
    std::vector<LinArgs> todo;
    todo.reserve(m);
    for (int k = 0; k < m; ++k) {
      Graph& g = gs[lin[k]];
      Graph out = make_output(op, k, {g});
      init_scalar_result(out);
      // a sweep over target o emissions has read every emission of this chain already and left
      // forwardScore(emissions) behind (band.hip): nothing to launch
      const NormCache* nc = tropical ? nullptr : g.w->valid_norm_cache();
      if (nc && nc->norm) {
        set_dev_weights(out, nc->mem, nc->norm, 1);
        outs[lin[k]] = std::move(out);
        continue;
      }
      LinArgs& a = args[k];
      a.w = g.w->dev;
      a.M = g.s->M;
      a.C = g.s->C;
      a.out_score = scal + k;
      a.partial = partial + size_t(k) * 8;
      a.delta = nullptr;
      a.grad = nullptr;
      a.accumulate = 0;
      maxM = std::max(maxM, a.M);
      bytes += 4.0 * double(g.num_arcs());
      todo.push_back(a);
      set_dev_weights(out, res, scal + k, 1);
      outs[lin[k]] = std::move(out);
    }
    if (!todo.empty()) {
      DevMemP d = upload_vec(todo);
      GTNX_PROF("linear_forward", bytes);
      bool vec_rows = true;
      for (auto& a : todo) vec_rows = vec_rows && a.C % 4 == 0 && a.C <= 1024 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
      (void)maxM;
      launch_linear_forward(d->as<LinArgs>(), int(todo.size()), tropical ? 1 : 0, vec_rows ? 1 : 0, rt.stream());
    }
  }

  // ---- general DAGs: level-scheduled persistent kernel
  if (!exp.empty()) {
    const int m = int(exp.size());
    // lattices whose sizes are still on the device stay that way only for the
    // log-semiring narrow kernel (bounds suffice on the host); everything else waits
    for (int i : exp)
      if (gs[i].s->deferred && (tropical || !gs[i].s->sched)) gs[i].s->resolve_sizes();
    std::vector<Structure*> ss;
    for (int i : exp) ss.push_back(gs[i].s.get());
    ensure_schedule_batch(ss, false);
    for (int i : exp)
      if (gs[i].s->sched->error) throw_invalid(kCycleMsg);  // shortest.cpp:149-152
    auto op = std::make_shared<SdOp>();
    op->mode = tropical ? SD_TROPICAL : SD_LOG;
    op->seq = next_seq();
    size_t bytes = 0;
    std::vector<size_t> off_s(m), off_a(m), off_r(m);
    for (int k = 0; k < m; ++k) {
      const int P = gs[exp[k]].s->sched->view.P;
      off_s[k] = bytes;
      bytes = align_up(bytes + 4 * size_t(P) + 16, 256);  // +16: vector staging may read past the end
      off_a[k] = bytes;
      if (tropical) bytes = align_up(bytes + 4 * size_t(P), 256);
      off_r[k] = bytes;
      bytes += 256;
    }
    size_t off_out = bytes;
    bytes += 4 * size_t(m);
    DevMemP arena = rt.alloc(bytes);
    op->arena = arena;
    op->saved.resize(m);
    std::vector<SdArgs> args(m);
    int64_t tot_in = 0, tot_p = 0;
    int maxw = 0;
    double alg = 0;
    // deep & narrow lattices take the LDS-ring kernel (whole batch must qualify; the
    // tropical form additionally needs the row-ordered weights compose emits)
    bool narrow = true;
    int64_t tot_levels = 0;
    for (int k = 0; k < m; ++k) {
      Schedule& sc = *gs[exp[k]].s->sched;
      narrow = narrow && sc.max_level_arcs <= sd_narrow_tmp_cap() && sc.max_level_width <= sd_narrow_node_cap() &&
               sc.max_reach <= sd_narrow_ring();
      tot_levels += sc.view.L;
    }
    narrow = narrow && tot_levels >= 32 * int64_t(m);
    if (!narrow)
      for (int k = 0; k < m; ++k) gs[exp[k]].s->resolve_sizes();  // generic kernels take sizes from the host
    for (int k = 0; k < m; ++k) {
      Graph& g = gs[exp[k]];
      Schedule& sc = *g.s->sched;
      SdArgs& a = args[k];
      std::memset(&a, 0, sizeof(a));
      // the log narrow kernel reads in_src / in_w / row offsets only; everything else also arc ids
      a.s = sched_view(g, /*need_full=*/!narrow || tropical);
      a.w = g.w->dev;
      a.scores = arena->as<float>(off_s[k]);
      a.argmax = tropical ? arena->as<int>(off_a[k]) : nullptr;
      a.result = arena->as<SdResult>(off_r[k]);
      a.out_score = arena->as<float>(off_out) + k;
      a.delta = nullptr;
      a.node_grad = nullptr;
      a.arc_grad = nullptr;
      a.chunk_levels = std::max(1, std::min(sd_narrow_tmp_cap() / std::max(sc.max_level_arcs, 1),
                                            sd_narrow_node_cap() / std::max(sc.max_level_width, 1)));
      if (narrow) {
        a.dyn_out = sc.dyn_out;
        a.dyn_counts = sc.dyn_counts;
      }
      op->saved[k] = {g.s->sched, a.scores, a.argmax, a.result};
      tot_in += sc.n_in;
      tot_p += sc.view.P;
      maxw = std::max(maxw, sc.max_level_width);
      if (g.s->deferred) {
        if (rt.prof_on())
          g.s->deferred->prof.push_back({tropical ? "viterbi_score" : "forward_score", 8.0, 8.0, g.s->deferred_idx});
      } else {
        alg += 8.0 * double(g.num_arcs()) + 8.0 * double(g.num_nodes());
      }
      Graph out = make_output(op, k, {g});
      init_scalar_result(out);
      set_dev_weights(out, arena, a.out_score, 1);
      outs[exp[k]] = std::move(out);
    }
    DevMemP d = upload_vec(args);
    GTNX_PROF(tropical ? "viterbi_score" : "forward_score", alg);
    bool all_inw = true;
    for (auto& a : args) all_inw = all_inw && a.s.in_w != nullptr;
    if (!all_inw)
      for (int k = 0; k < m; ++k) gs[exp[k]].s->ensure_full();  // weights by arc id need in_arc
    launch_sd_forward(d->as<SdArgs>(), m, op->mode, narrow ? (all_inw ? 2 : 1) : 0,
                      int(tot_p ? (tot_in * 16) / tot_p : 0), rt.stream());
  }
  return outs;
}

// ======================================================================
// viterbiPath (functions.cpp:328-330, shortest.cpp:190-272)
// ======================================================================
namespace {
// the chain graph of a best path (shortest.cpp:248-260), written straight into the
// host mirror: `len` arcs i -> i+1; len < 0: the empty graph; len == 0: one node
void fill_path_graph(Graph& out, int len, bool has_node, const int* il, const int* ol, const float* w) {
  Structure& s = *out.s;
  if (len < 0 || (!has_node && len == 0)) return;
  const int N = len + 1;
  s.N = N;
  s.A = len;
  s.nflags.assign(size_t(N), 0);
  s.nflags[0] |= NF_START;
  s.nflags[size_t(N) - 1] |= NF_ACCEPT;
  s.start = {0};
  s.accept = {N - 1};
  s.src.resize(size_t(len));
  s.dst.resize(size_t(len));
  for (int i = 0; i < len; ++i) {
    s.src[size_t(i)] = i;
    s.dst[size_t(i)] = i + 1;
  }
  s.il.assign(il, il + len);
  s.ol.assign(ol, ol + len);
  s.host_valid = true;
  s.csr_valid = false;
  s.dev_valid = false;
  out.w->host.assign(w, w + len);
  out.w->n = len;
  out.w->host_valid = true;
  out.w->dev_valid = false;
}
struct PathOp : OpRecord {
  // per member: the path's arc ids in the order the reference's gradFunc indexes
  // them (last-arc-first, shortest.cpp:240-245 & 262-268)
  std::vector<std::vector<int>> arcs_rev;
  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    const int n = int(ms.size());
    size_t tot_idx = 0, tot_grad = 0;
    for (auto& m : ms) {
      tot_idx += arcs_rev[m.idx].size();
      tot_grad += size_t(m.out.g->inputs[0].num_arcs());
    }
    std::vector<int> idx_host;
    idx_host.reserve(tot_idx);
    DevMemP grads = rt.alloc_zero(sizeof(float) * (tot_grad ? tot_grad : 1));
    std::vector<ScatterArgs> args(n);
    std::vector<size_t> ioff(n);
    for (int i = 0; i < n; ++i) {
      ioff[i] = idx_host.size();
      const auto& v = arcs_rev[ms[i].idx];
      idx_host.insert(idx_host.end(), v.begin(), v.end());
    }
    DevMemP didx = upload_vec(idx_host);
    GradSink sink;
    size_t goff = 0;
    int maxn = 0;
    for (int i = 0; i < n; ++i) {
      Graph& in = ms[i].out.g->inputs[0];
      ScatterArgs& a = args[i];
      a.idx = didx->as<int>() + ioff[i];
      a.n = int(arcs_rev[ms[i].idx].size());
      a.delta = a.n ? grad_dev_ptr(ms[i].out) : nullptr;
      a.grad = grads->as<float>() + goff;
      sink.add(in, grads, a.grad);
      goff += size_t(in.num_arcs());
      maxn = std::max(maxn, a.n);
    }
    DevMemP d = upload_vec(args);
    launch_scatter_add(d->as<ScatterArgs>(), n, maxn, rt.stream());
    sink.flush();
  }
};
} // namespace

std::vector<Graph> op_viterbi_path(std::vector<Graph>& gs) {
  const size_t n = gs.size();
  std::vector<Graph> outs;
  if (n == 0) return outs;
  {
    std::vector<Graph> lz, rest;
    std::vector<size_t> lz_i, rest_i;
    for (size_t i = 0; i < n; ++i) {
      if (gs[i].s->lazy) { lz.push_back(gs[i]); lz_i.push_back(i); }
      else { rest.push_back(gs[i]); rest_i.push_back(i); }
    }
    if (!lz.empty()) {
      outs.assign(n, Graph(false));
      std::vector<Graph> lo = lazy_viterbi_path(lz);
      for (size_t k = 0; k < lz.size(); ++k) outs[lz_i[k]] = lo[k];
      if (!rest.empty()) {
        std::vector<Graph> ro = op_viterbi_path(rest);
        for (size_t k = 0; k < rest.size(); ++k) outs[rest_i[k]] = ro[k];
      }
      return outs;
    }
  }
  Runtime& rt = Runtime::get();
  for (auto& g : gs) g.s->resolve_sizes();  // path extraction sizes its buffers from the real counts
  for (auto& g : gs) g.s->materialize();  // TODO(linear fast path): row arg-max needs no graph
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (auto& g : gs) {
    ss.push_back(g.s.get());
    ws.push_back(g.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  ensure_schedule_batch(ss, true);
  for (auto& g : gs)
    if (g.s->sched->error) throw_invalid(kCycleMsg);  // shortest.cpp:229-232
  const int m = int(n);
  size_t bytes = 0;
  std::vector<size_t> off_s(m), off_a(m), off_r(m), off_p(m);
  std::vector<int> cap(m);
  for (int k = 0; k < m; ++k) {
    const DSched& v = gs[k].s->sched->view;
    cap[k] = std::max(v.L, 1);
    off_s[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(v.P), 256);
    off_a[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(v.P), 256);
    off_r[k] = bytes;
    bytes += 256;
    off_p[k] = bytes;
    bytes = align_up(bytes + 20 * size_t(cap[k]) + 16, 256);
  }
  DevMemP arena = rt.alloc(bytes);
  std::vector<SdArgs> args(m);
  std::vector<PathArgs> pargs(m);
  int max_cap = 1;
  int64_t tot_in = 0, tot_p = 0;
  for (int k = 0; k < m; ++k) {
    Graph& g = gs[k];
    SdArgs& a = args[k];
    a.s = sched_view(g);
    a.w = g.w->dev;
    a.scores = arena->as<float>(off_s[k]);
    a.argmax = arena->as<int>(off_a[k]);
    a.result = arena->as<SdResult>(off_r[k]);
    a.out_score = nullptr;
    a.delta = nullptr;
    a.node_grad = nullptr;
    a.arc_grad = nullptr;
    {
      const Schedule& sc = *g.s->sched;
      a.chunk_levels = std::max(1, std::min(sd_narrow_tmp_cap() / std::max(sc.max_level_arcs, 1),
                                            sd_narrow_node_cap() / std::max(sc.max_level_width, 1)));
    }
    PathArgs& p = pargs[k];
    p.s = a.s;
    p.g = device_view(g);
    p.argmax = a.argmax;
    p.result = a.result;
    char* pb = arena->as<char>(off_p[k]);
    p.path_len = reinterpret_cast<int*>(pb);
    p.path_arcs = reinterpret_cast<int*>(pb + 16);
    p.path_il = p.path_arcs + cap[k];
    p.path_ol = p.path_il + cap[k];
    p.path_w = reinterpret_cast<float*>(p.path_ol + cap[k]);
    p.path_pos = reinterpret_cast<int*>(p.path_w + cap[k]);
    p.cap = cap[k];
    p.scores = a.scores;
    p.w = a.w;
    max_cap = std::max(max_cap, cap[k]);
    tot_in += g.s->sched->n_in;
    tot_p += a.s.P;
  }
  DevMemP d = upload_vec(args);
  DevMemP dp = upload_vec(pargs);
  {
    GTNX_PROF("viterbi_path", 0.0);
    // deep narrow lattices with row-ordered weights (compose products): LDS-ring kernel
    bool narrow = true;
    int64_t tot_levels = 0;
    for (int k = 0; k < m; ++k) {
      const Schedule& sc = *gs[k].s->sched;
      narrow = narrow && args[k].s.in_w != nullptr && sc.max_level_arcs <= sd_narrow_tmp_cap() &&
               sc.max_level_width <= sd_narrow_node_cap() && sc.max_reach <= sd_narrow_ring();
      tot_levels += sc.view.L;
    }
    narrow = narrow && tot_levels >= 32 * int64_t(m);
    launch_sd_forward(d->as<SdArgs>(), m, SD_PATH, narrow ? 2 : 0, int(tot_p ? (tot_in * 16) / tot_p : 0), rt.stream());
    launch_path_chase(dp->as<PathArgs>(), m, max_cap, rt.stream());
  }
  // the path is at most L arcs: bring it to the host and build the chain graph there
  std::vector<char> host(bytes);
  rt.d2h_sync(host.data(), arena->ptr, bytes);
  // Exact ties on a path through a product that carries compose's own schedule (ties by arc id, positions =
  // node ids): rerun those on the schedule that replays the reference's queue (graph.cpp:
  // build_host_schedule), whose rank IS the reference's relaxation order.  Host-built graphs have it already.
  std::vector<int> tied;
  for (int k = 0; k < m; ++k) {
    const int* pl = reinterpret_cast<const int*>(host.data() + off_p[k]);
    if (pl[2] && (gs[k].s->sched->view.flags & SCHED_TIE_BY_ARC) && !getenv("GTNX_NO_TIE_RERUN")) tied.push_back(k);
  }
  std::vector<Graph> redo;
  if (!tied.empty()) {
    std::vector<Graph> tg;
    for (int k : tied) {
      gs[k].s->resolve_sizes();
      gs[k].s->ensure_full();
      gs[k].s->ensure_host();
      gs[k].s->sched.reset();
      tg.push_back(gs[k]);
    }
    redo = op_viterbi_path(tg);
  }
  auto op = std::make_shared<PathOp>();
  op->seq = next_seq();
  op->arcs_rev.resize(m);
  size_t next_tied = 0;
  for (int k = 0; k < m; ++k) {
    if (next_tied < tied.size() && tied[next_tied] == k) {
      outs.push_back(std::move(redo[next_tied++]));
      continue;
    }
    const char* pb = host.data() + off_p[k];
    const int* pl = reinterpret_cast<const int*>(pb);
    const int len = pl[0], has_node = pl[1];
    const int* arcs = reinterpret_cast<const int*>(pb + 16);
    const int* il = arcs + cap[k];
    const int* ol = il + cap[k];
    const float* w = reinterpret_cast<const float*>(ol + cap[k]);
    Graph out = make_output(op, k, {gs[k]});
    fill_path_graph(out, len, has_node != 0, il, ol, w);
    op->arcs_rev[k].assign(arcs, arcs + len);
    std::reverse(op->arcs_rev[k].begin(), op->arcs_rev[k].end());
    outs.push_back(std::move(out));
  }
  return outs;
}

// ======================================================================
// compose / intersect (functions.cpp:225-251, compose.cpp:377-522)
// ======================================================================
namespace {

struct ComposeOp : OpRecord {
  DevMemP arena;
  struct Saved {
    const int* gi1;
    const int* gi2;
    int A;
  };
  std::vector<Saved> saved;
  std::shared_ptr<DeferredSizes> deferred;  // sizes of the batch still on the device
  void backward(std::vector<Member>& all) override {
    Runtime& rt = Runtime::get();
    // members whose consumer already scattered their gradient (SdOp::backward, fused)
    std::vector<Member> ms;
    for (auto& m : all) {
      if (m.out.g->grad_propagated) m.out.g->grad_propagated = false;
      else ms.push_back(m);
    }
    if (ms.empty()) return;
    if (deferred) {  // the separate gradient kernel needs the arc counts
      deferred->resolve();
      for (auto& m : ms) saved[m.idx].A = int(m.out.s->A);
    }
    const int n = int(ms.size());
    size_t bytes = 0;
    std::vector<size_t> o1(n), o2(n);
    for (int i = 0; i < n; ++i) {
      auto& ins = ms[i].out.g->inputs;
      o1[i] = bytes;
      if (ins[0].calc_grad()) bytes = align_up(bytes + 4 * size_t(ins[0].num_arcs()), 256);
      o2[i] = bytes;
      if (ins[1].calc_grad()) bytes = align_up(bytes + 4 * size_t(ins[1].num_arcs()), 256);
    }
    DevMemP g = rt.alloc_zero(bytes ? bytes : 1);
    std::vector<ComposeGradArgs> args(n);
    GradSink sink;
    int maxA = 0;
    double alg = 0;
    for (int i = 0; i < n; ++i) {
      auto& ins = ms[i].out.g->inputs;
      const Saved& sv = saved[ms[i].idx];
      ComposeGradArgs& a = args[i];
      a.A = sv.A;
      a.gi1 = sv.gi1;
      a.gi2 = sv.gi2;
      a.delta = sv.A ? grad_dev_ptr(ms[i].out) : nullptr;
      a.A1 = int(ins[0].num_arcs());
      a.A2 = int(ins[1].num_arcs());
      a.grad1 = ins[0].calc_grad() ? g->as<float>(o1[i]) : nullptr;
      a.grad2 = ins[1].calc_grad() ? g->as<float>(o2[i]) : nullptr;
      if (a.grad1) sink.add(ins[0], g, a.grad1);
      if (a.grad2) sink.add(ins[1], g, a.grad2);
      maxA = std::max(maxA, sv.A);
      alg += 12.0 * sv.A + 4.0 * (a.A1 + a.A2);
    }
    DevMemP d = upload_vec(args);
    {
      GTNX_PROF("compose_grad", alg);
      launch_compose_grad(d->as<ComposeGradArgs>(), n, maxA, rt.stream());
    }
    sink.flush();
  }
};

// label histogram of the labels compose matches on (olabel of g1 / ilabel of g2).
// Dense counts when the labels are small (the usual case), a hash map otherwise.
struct LabelHist {
  bool linear = false;
  int M = 0, C = 0;
  std::vector<int64_t> dense;               // dense[l] for 0 <= l < dense.size()
  std::unordered_map<int, int64_t> sparse;  // labels >= kDenseMax
  int64_t eps = 0;
  int64_t count(int l) const {
    if (l < int(dense.size())) return dense[l];
    auto it = sparse.find(l);
    return it == sparse.end() ? 0 : it->second;
  }
};
constexpr int kDenseMax = 1 << 16;
void label_hist(Structure& s, bool use_olabel, LabelHist& h) {
  if (s.kind == KIND_LINEAR) {
    h.linear = true;
    h.M = s.M;
    h.C = s.C;
    return;
  }
  s.ensure_host();
  const std::vector<int>& lab = use_olabel ? s.ol : s.il;
  int mx = -1;
  for (int l : lab) mx = std::max(mx, l);
  h.dense.assign(size_t(std::min(mx + 1, kDenseMax)), 0);
  for (int l : lab) {
    if (l == GTNX_EPSILON)
      h.eps++;
    else if (l < kDenseMax)
      h.dense[l]++;
    else
      h.sparse[l]++;
  }
}
int64_t match_bound(const LabelHist& a, const LabelHist& b) {
  // sum over non-eps labels of cnt_a[l] * cnt_b[l]
  if (a.linear && b.linear) return int64_t(std::min(a.C, b.C)) * a.M * b.M;
  const LabelHist& e = a.linear ? b : a;  // an explicit side
  const LabelHist& o = a.linear ? a : b;
  int64_t t = 0;
  for (size_t l = 0; l < e.dense.size(); ++l) {
    if (!e.dense[l]) continue;
    t += e.dense[l] * (o.linear ? ((int(l) < o.C) ? int64_t(o.M) : 0) : o.count(int(l)));
  }
  for (auto& kv : e.sparse) t += kv.second * (o.linear ? ((kv.first < o.C) ? int64_t(o.M) : 0) : o.count(kv.first));
  return t;
}
} // namespace

std::vector<Graph> op_compose_impl(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect, bool allow_lazy);
std::vector<Graph> op_compose(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect) {
  return op_compose_impl(av, bv, intersect, true);
}

namespace {
thread_local int t_compose_mode = 0;
}
int compose_mode_hint(int mode) {
  const int old = t_compose_mode;
  t_compose_mode = mode;
  return old;
}

std::vector<Graph> op_compose_impl(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect, bool allow_lazy) {
  GTNX_HOST_T("compose.total");
  const size_t n = std::max(av.size(), bv.size());
  std::vector<Graph> outs;
  if (n == 0) return outs;
  Runtime& rt = Runtime::get();
  double ht_mark = 0;
  auto ht_phase = [&](const char* name) {  // GTNX_HOST_TIMING: time since the previous mark
    if (!HostTimer::enabled()) return;
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (ht_mark != 0) { HostTimer t(name); t.t0 = ht_mark; }
    ht_mark = now;
  };
  ht_phase("");
  if (allow_lazy) {
    // The criteria's hint (mode 2) with banded partners -- CTC targets: the product stays symbolic and
    // band.hip sweeps it, so nothing of the inputs is uploaded, counted or sorted here.
    const char* env = getenv("GTNX_LAZY_COMPOSE");
    const int mode = env && env[0] >= '0' && env[0] <= '2' ? env[0] - '0' : t_compose_mode;
    if (mode == 2 && !getenv("GTNX_NO_BAND")) {
      bool ok = true;
      std::vector<Graph*> fx(n);
      std::vector<uint8_t> cf(n);
      for (size_t i = 0; i < n && ok; ++i) {
        Graph& a = const_cast<Graph&>(bcast(av, n, i));
        Graph& b = const_cast<Graph&>(bcast(bv, n, i));
        const bool l1 = a.s->kind == KIND_LINEAR && !a.s->lazy, l2 = b.s->kind == KIND_LINEAR && !b.s->lazy;
        ok = l1 != l2;
        if (!ok) break;
        fx[i] = l1 ? &b : &a;
        cf[i] = l1;
        ok = !fx[i]->s->lazy && !fx[i]->s->deferred && fx[i]->s->kind == KIND_EXPLICIT && fx[i]->s->host_valid;
      }
      if (ok) {
        ht_phase("compose.0a_checks");
        band_prepare(fx, cf);
        ht_phase("compose.0b_band_prepare");
        for (size_t i = 0; i < n && ok; ++i) {
          Graph& a = const_cast<Graph&>(bcast(av, n, i));
          Graph& b = const_cast<Graph&>(bcast(bv, n, i));
          ok = band_shape_ok(*(cf[i] ? a : b).s, *fx[i]->s, cf[i] != 0);
        }
      }
      if (ok) {
        ht_phase("compose.0c_shape_ok");
        rt.drain_deferred();  // the step's reclamation point (see below)
        ht_phase("compose.0d_drain");
        auto lop = make_lazy_compose_op();
        outs.reserve(n);
        for (size_t i = 0; i < n; ++i) {
          Graph& a = const_cast<Graph&>(bcast(av, n, i));
          Graph& b = const_cast<Graph&>(bcast(bv, n, i));
          Graph out = make_output(lop, int(i), {a, b});
          out.s->host_valid = false;
          out.s->lazy = std::make_shared<LazyProduct>(LazyProduct{cf[i] ? a : b, cf[i] ? b : a, cf[i] ? 1 : 2, intersect});
          outs.push_back(std::move(out));
        }
        ht_phase("compose.0_symbolic_band");
        return outs;
      }
    }
  }
  for (auto& g : av) realize(g);
  for (auto& g : bv) realize(g);
  for (auto& g : av) g.s->resolve_sizes();
  for (auto& g : bv) g.s->resolve_sizes();
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (size_t i = 0; i < n; ++i) {
    Graph& a = const_cast<Graph&>(bcast(av, n, i));
    Graph& b = const_cast<Graph&>(bcast(bv, n, i));
    ss.push_back(a.s.get());
    ss.push_back(b.s.get());
    ws.push_back(a.w.get());
    ws.push_back(b.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  ht_phase("compose.1_upload_inputs");
  // device-built inputs (results of an earlier compose) get their packed
  // adjacency records now; host-built ones got them at upload
  for (Structure* st : ss) ensure_records(*st);

  // ---- capacities from label histograms (exact upper bound on matches)
  std::unordered_map<Structure*, LabelHist> h1, h2;
  struct Cap {
    int64_t N1, N2, Ncap, Acap, pairs;
  };
  std::vector<Cap> caps(n);
  std::map<std::tuple<Structure*, int, bool>, std::pair<int64_t, int64_t>> direct_counts;
  for (size_t i = 0; i < n; ++i) {
    Structure& s1 = *bcast(av, n, i).s;
    Structure& s2 = *bcast(bv, n, i).s;
    if ((s1.kind == KIND_LINEAR) != (s2.kind == KIND_LINEAR)) {
      // one implicit chain: every label below C matches M chain arcs -- a single
      // pass over the explicit side's labels, no histogram
      const bool l1 = s1.kind == KIND_LINEAR;
      Structure& e = l1 ? s2 : s1;
      const Structure& ch = l1 ? s1 : s2;
      // (a partner shared by the whole batch -- ASG transitions -- is counted once)
      const auto key = std::make_tuple(&e, ch.C, l1);
      auto hit_it = direct_counts.find(key);
      if (hit_it == direct_counts.end()) {
        e.ensure_host();
        const std::vector<int>& lab = l1 ? e.il : e.ol;
        int64_t h = 0, ep = 0;
        for (int l : lab) {
          h += (l >= 0 && l < ch.C);
          ep += (l == GTNX_EPSILON);
        }
        hit_it = direct_counts.emplace(key, std::make_pair(h, ep)).first;
      }
      const int64_t hit = hit_it->second.first, eps = hit_it->second.second;
      Cap& c = caps[i];
      c.N1 = s1.N;
      c.N2 = s2.N;
      c.pairs = c.N1 * c.N2;
      c.Acap = hit * ch.M + eps * ch.N;
      const int64_t starts = l1 ? int64_t(s2.start.size()) : int64_t(s1.start.size());
      c.Ncap = std::min<int64_t>(c.pairs, c.Acap + starts);
      if (c.pairs > (int64_t(1) << 30) || c.Acap > (int64_t(1) << 30))
        throw_runtime("[gtn::compose] composed graph too large for 32-bit indices");
      continue;
    }
    if (!h1.count(&s1)) label_hist(s1, true, h1[&s1]);
    if (!h2.count(&s2)) label_hist(s2, false, h2[&s2]);
    const LabelHist& x = h1[&s1];
    const LabelHist& y = h2[&s2];
    Cap& c = caps[i];
    c.N1 = s1.N;
    c.N2 = s2.N;
    c.pairs = c.N1 * c.N2;
    c.Acap = match_bound(x, y) + x.eps * c.N2 + y.eps * c.N1;
    const int64_t starts = (s1.kind == KIND_LINEAR ? 1 : int64_t(s1.start.size())) *
                           (s2.kind == KIND_LINEAR ? 1 : int64_t(s2.start.size()));
    c.Ncap = std::min<int64_t>(c.pairs, c.Acap + starts);
    if (c.pairs > (int64_t(1) << 30) || c.Acap > (int64_t(1) << 30))
      throw_runtime("[gtn::compose] composed graph too large for 32-bit indices");
  }

  ht_phase("compose.2_caps");
  // ---- keep the product symbolic?  Only a chain product with an epsilon-free partner
  // qualifies; it is taken when building the batch would not fit (or on request).
  if (allow_lazy) {
    // mode 0: only when the batch would not fit; 1: whenever eligible; 2: when the per-pair
    // kernels of lazy_pair.hip apply.  The caller's hint (gtnx_compose_mode), overridden by
    // GTNX_LAZY_COMPOSE ("0" additionally forbids symbolic products altogether)
    const char* env = getenv("GTNX_LAZY_COMPOSE");
    const int mode = env && env[0] >= '0' && env[0] <= '2' ? env[0] - '0' : t_compose_mode;
    const bool force = mode == 1, never = env && env[0] == '0';
    const char* benv = getenv("GTNX_LAZY_BYTES");
    const double budget = benv ? atof(benv) : 128e9;
    bool eligible = !never;
    // "2": also whenever every product has the per-pair kernels of lazy_pair.hip (small G)
    bool pairs = mode == 2;
    double est = 0;
    for (size_t i = 0; i < n && eligible; ++i) {
      Graph& a = const_cast<Graph&>(bcast(av, n, i));
      Graph& b = const_cast<Graph&>(bcast(bv, n, i));
      const bool l1 = a.s->kind == KIND_LINEAR, l2 = b.s->kind == KIND_LINEAR;
      eligible = (l1 != l2) && (((l1 ? b : a).s->dview.flags & GF_EPS_FREE) != 0) &&
                 lazy_shape_ok(*(l1 ? a : b).s, *(l1 ? b : a).s);
      pairs = pairs && eligible && lazy_pair_shape_ok(*(l1 ? a : b).s, *(l1 ? b : a).s);
      est += 44.0 * double(caps[i].Acap) + 30.0 * double(caps[i].Ncap) + 8.0 * double(caps[i].pairs);
    }
    if (eligible && (force || pairs || est > budget)) {
      // nothing downstream of a symbolic product waits for the GPU, so this is the step's
      // reclamation point (objects the caller let go of since the last one; cheap while
      // their memory is still warm for the allocator -- see Runtime::defer_delete)
      rt.drain_deferred();
      auto lop = make_lazy_compose_op();
      for (size_t i = 0; i < n; ++i) {
        Graph& a = const_cast<Graph&>(bcast(av, n, i));
        Graph& b = const_cast<Graph&>(bcast(bv, n, i));
        const bool l1 = a.s->kind == KIND_LINEAR;
        Graph out = make_output(lop, int(i), {a, b});
        out.s->host_valid = false;
        out.s->lazy = std::make_shared<LazyProduct>(LazyProduct{l1 ? a : b, l1 ? b : a, l1 ? 1 : 2, intersect});
        outs.push_back(std::move(out));
      }
      return outs;
    }
  }
  // ---- arenas.  Scratch is laid out by kind (all `state` tables contiguous,
  // all in-degree cursors contiguous) so ONE fill and ONE memset initialise the
  // whole batch; result headers (sizes) are contiguous so ONE copy returns them.
  struct Off {
    size_t state, queue, pair_of, in_cursor;
    size_t src, dst, il, ol, w, gi1, gi2, nf, out_off, level_off, in_off, in_list, in_src, in_w, sl, al;
  };
  std::vector<Off> offs(n);
  size_t st_b = 0, cu_b = 0, sc_b = 0, rb = 0;
  auto add = [](size_t& tot, size_t bytes) {
    size_t o = tot;
    tot = align_up(tot + bytes, 256);
    return o;
  };
  const size_t hdr_out = add(rb, sizeof(ComposeOut) * n);
  const size_t hdr_cnt = add(rb, 8 * n);
  int64_t maxA = 0, maxN = 0;
  for (size_t i = 0; i < n; ++i) {
    const Cap& c = caps[i];
    Off& o = offs[i];
    const size_t A = size_t(c.Acap), N = size_t(c.Ncap), P = size_t(c.pairs);
    o.state = add(st_b, 4 * P);
    o.in_cursor = add(cu_b, 4 * N);
    o.queue = add(sc_b, 4 * P);
    o.pair_of = add(sc_b, 4 * N);
    o.src = add(rb, 4 * A);
    o.dst = add(rb, 4 * A + 16);
    o.il = add(rb, 4 * A);
    o.ol = add(rb, 4 * A);
    o.w = add(rb, 4 * A + 16);
    o.gi1 = add(rb, 4 * A);
    o.gi2 = add(rb, 4 * A);
    o.nf = add(rb, N + 16);
    o.out_off = add(rb, 4 * (N + 1) + 16);
    o.level_off = add(rb, 4 * (N + 2));
    o.in_off = add(rb, 4 * (N + 1) + 16);
    o.in_list = add(rb, 4 * A);
    o.in_src = add(rb, 4 * A + 16);
    o.in_w = add(rb, 4 * A + 16);
    o.sl = add(rb, 4 * N);
    o.al = add(rb, 4 * N);
    maxA = std::max(maxA, c.Acap);
    maxN = std::max(maxN, c.Ncap);
  }
  DevMemP st_mem = rt.alloc(st_b ? st_b : 1);
  DevMemP cu_mem = rt.alloc(cu_b ? cu_b : 1);
  DevMemP sc_mem = rt.alloc(sc_b ? sc_b : 1);
  DevMemP res = rt.alloc(rb ? rb : 1);
  // small pair tables keep their bitmaps in LDS (whole batch must qualify, the
  // dynamic LDS request is per launch); the HBM table is then written by the kernel
  // Two layouts (compose.hip): the classic pair-indexed bitmaps (2 * N1*N2 bits), and for
  // chain products with an epsilon-free partner a window of time slices whose size does
  // not depend on the chain length.  `fast_ok`: every pair fits one of them (FAST variant);
  // `classic_ok[i]`: the general variant may keep pair i's classic bitmaps in LDS.
  std::vector<int> chain_slices(n, 0);
  std::vector<char> classic_ok(n, 0), full_window(n, 0);
  size_t fast_bm = 0, classic_bm = 0;
  bool fast_ok = true;
  // Chain products whose partner has wide nodes (more candidate arcs per node than the lane-per-node kernel
  // caches: transition graphs) go to compose_wide.hip: a wave per frontier node, stationary levels written by a
  // grid.  Its arc order is the partner's list order, which is the reference's as long as a partner that is
  // matched as "sorted" is sorted on the label being matched (g2: ilabel, g1: olabel; functions.cpp:225-251).
  std::vector<char> wide_ok(n, 0), wide_pref(n, 0);
  if (!getenv("GTNX_NO_WIDE_COMPOSE")) {
    for (size_t i = 0; i < n; ++i) {
      const Structure& s1 = *bcast(av, n, i).s;
      const Structure& s2 = *bcast(bv, n, i).s;
      const bool l1 = s1.kind == KIND_LINEAR, l2 = s2.kind == KIND_LINEAR;
      if (l1 == l2) continue;
      const Structure& ex = l1 ? s2 : s1;
      const Structure& ch = l1 ? s1 : s2;
      const bool sorted_claim = intersect ? (ex.ilabel_sorted || ex.olabel_sorted) : (l1 ? ex.ilabel_sorted : ex.olabel_sorted);
      // (an acceptor sorted on either label is sorted on both)
      const bool sorted_on_match = (l1 ? ex.ilabel_sorted : ex.olabel_sorted) || ((ex.dview.flags & GF_ACCEPTOR) && sorted_claim);
      wide_ok[i] = (ex.dview.flags & GF_EPS_FREE) && ch.M >= 1 && ex.N >= 1 && ex.N <= compose_wide_node_cap() &&
                   (!sorted_claim || sorted_on_match);
      wide_pref[i] = wide_ok[i] && (ex.A > 4 * ex.N || getenv("GTNX_FORCE_WIDE_COMPOSE"));
    }
  }
  // Products of two explicit graphs with wide nodes take compose_wide.hip's wave-per-pair kernel.  It searches
  // sorted lists only (an unsorted second graph through a stable sorted view), so a graph that is matched as
  // "sorted" must be sorted on the label being matched.
  std::vector<char> pairs_ok(n, 0), pairs_pref(n, 0);
  auto matcher_of = [&](const Structure& s1, const Structure& s2) {
    const bool c1 = intersect ? (s1.ilabel_sorted || s1.olabel_sorted) : s1.olabel_sorted;
    const bool c2 = intersect ? (s2.ilabel_sorted || s2.olabel_sorted) : s2.ilabel_sorted;
    return (c1 && c2) ? MATCH_DOUBLY : (c1 ? MATCH_SINGLY_G1 : (c2 ? MATCH_SINGLY_G2 : MATCH_UNSORTED));
  };
  if (!getenv("GTNX_NO_WIDE_COMPOSE") && !getenv("GTNX_NO_PAIRS_COMPOSE")) {
    for (size_t i = 0; i < n; ++i) {
      const Structure& s1 = *bcast(av, n, i).s;
      const Structure& s2 = *bcast(bv, n, i).s;
      if (s1.kind != KIND_EXPLICIT || s2.kind != KIND_EXPLICIT) continue;
      const int m = matcher_of(s1, s2);
      // (an acceptor sorted on either label is sorted on both)
      const bool a1 = (s1.dview.flags & GF_ACCEPTOR) && (s1.ilabel_sorted || s1.olabel_sorted);
      const bool a2 = (s2.dview.flags & GF_ACCEPTOR) && (s2.ilabel_sorted || s2.olabel_sorted);
      const bool t1 = (m == MATCH_DOUBLY || m == MATCH_SINGLY_G1) ? (s1.olabel_sorted || a1) : true;
      const bool t2 = (m == MATCH_DOUBLY || m == MATCH_SINGLY_G2) ? (s2.ilabel_sorted || a2) : true;
      pairs_ok[i] = t1 && t2;
      pairs_pref[i] = pairs_ok[i] && (s1.A > 4 * s1.N || s2.A > 4 * s2.N || getenv("GTNX_FORCE_WIDE_COMPOSE"));
    }
  }
  // 512-lane workgroups when some chain product's partner has 257..512 nodes (and none more)
  bool wide = false;
  {
    bool any_wide = false, all_fit = true;
    for (size_t i = 0; i < n; ++i) {
      const Structure& s1 = *bcast(av, n, i).s;
      const Structure& s2 = *bcast(bv, n, i).s;
      const bool l1 = s1.kind == KIND_LINEAR, l2 = s2.kind == KIND_LINEAR;
      if (l1 == l2) continue;
      const int64_t No = (l1 ? s2 : s1).N;
      any_wide = any_wide || No > 256;
      all_fit = all_fit && No <= 512;
    }
    wide = any_wide && all_fit && !getenv("GTNX_NARROW_COMPOSE");
  }
  {
    const bool no_chain = getenv("GTNX_CLASSIC_BITMAPS") != nullptr;
    const size_t budget = std::min<size_t>(size_t(compose_max_bitmap_bytes()), size_t(compose_lds_budget(wide ? 1 : 0)));
    for (size_t i = 0; i < n; ++i) {
      const Structure& s1 = *bcast(av, n, i).s;
      const Structure& s2 = *bcast(bv, n, i).s;
      const size_t classic = 2 * 4 * ((size_t(caps[i].pairs) + 31) / 32);
      classic_ok[i] = classic <= budget;
      if (classic_ok[i]) classic_bm = std::max(classic_bm, classic);
      size_t mine = classic;
      const bool l1 = s1.kind == KIND_LINEAR, l2 = s2.kind == KIND_LINEAR;
      // (partners of up to 1024 nodes: the kernel indexes its claim table by partner node)
      if (!no_chain && l1 != l2 && ((l1 ? s2 : s1).dview.flags & GF_EPS_FREE) && (l1 ? s2 : s1).N >= 1 &&
          (l1 ? s2 : s1).N <= 1024) {
        const int No = int((l1 ? s2 : s1).N), TMc = (l1 ? s1 : s2).M;
        const int64_t room = int64_t(budget / (4 * size_t((No + 31) / 32))) - 3;
        // a window over ALL times when it fits (then the fast variant cannot run out of
        // slices); else ~No slices: stationarity arrives within that many steps, if at all
        int slices = int(std::min<int64_t>(TMc + 1, room));
        if (slices < TMc + 1) slices = int(std::min<int64_t>(No + 64, room));
        full_window[i] = slices >= TMc + 1;
        if (slices >= std::min(TMc + 1, 64)) {
          chain_slices[i] = slices;
          mine = compose_chain_bitmap_bytes(No, slices);
        }
      }
      if (wide_pref[i] || pairs_pref[i]) continue;  // never runs the FAST variant
      fast_ok = fast_ok && mine <= budget;
      fast_bm = std::max(fast_bm, mine);
    }
  }
  const int bitmap_bytes = int(fast_bm);
  const bool lds_state = fast_ok;  // the FAST variant can run
  // ... and, when it still fits, g1's adjacency records as well
  size_t g1_cache = 0;
  for (size_t i = 0; i < n; ++i) {
    const Structure& s1 = *bcast(av, n, i).s;
    if (s1.kind == KIND_EXPLICIT) g1_cache = std::max(g1_cache, compose_g1_cache_bytes(int(s1.N), int(s1.A)));
  }
  const bool cache1 = lds_state && g1_cache > 0 && bitmap_bytes + int(g1_cache) <= compose_lds_budget(wide ? 1 : 0);
  const int dyn_fast = bitmap_bytes + (cache1 ? int(g1_cache) : 0);
  bool state_filled = false;
  auto fill_state = [&] {  // the general variant's HBM pair table starts as "unreached"
    if (!state_filled) launch_fill_i32(st_mem->as<int>(), INT32_MIN, st_b / 4, rt.stream());
    state_filled = true;
  };
  if (!lds_state) fill_state();
  bool cursors_zeroed = false;
  auto zero_cursors = [&] {  // in-degree cursors of the transpose passes
    if (!cursors_zeroed) HIP_CHECK(hipMemsetAsync(cu_mem->ptr, 0, cu_b ? cu_b : 1, rt.stream()));
    cursors_zeroed = true;
  };
  std::vector<ComposeArgs> args(n);
  for (size_t i = 0; i < n; ++i) {
    const Cap& c = caps[i];
    const Off& o = offs[i];
    Graph& a = const_cast<Graph&>(bcast(av, n, i));
    Graph& b = const_cast<Graph&>(bcast(bv, n, i));
    ComposeArgs& x = args[i];
    x.g1 = device_view(a);
    x.g2 = device_view(b);
    // matcher dispatch, functions.cpp:225-251
    const bool s1 = intersect ? (a.s->ilabel_sorted || a.s->olabel_sorted) : a.s->olabel_sorted;
    const bool s2 = intersect ? (b.s->ilabel_sorted || b.s->olabel_sorted) : b.s->ilabel_sorted;
    x.matcher = (s1 && s2) ? MATCH_DOUBLY : (s1 ? MATCH_SINGLY_G1 : (s2 ? MATCH_SINGLY_G2 : MATCH_UNSORTED));
    x.lds_state = classic_ok[i] ? 1 : 0;  // read by the general variant only (FAST implies LDS)
    x.chain_bits = chain_slices[i];
    x.rep_grid = 0;
    {
      // chain product, epsilon-free partner no wider than a workgroup: every level is a
      // single fast chunk, so the FAST variant may leave the derivable arrays out
      const bool full_env = getenv("GTNX_FULL_COMPOSE") != nullptr;
      const bool l1 = a.s->kind == KIND_LINEAR, l2 = b.s->kind == KIND_LINEAR;
      const Structure& ex = l1 ? *b.s : *a.s;
      x.skip = (!full_env && lds_state && !wide_pref[i] && l1 != l2 && ((l1 ? x.g2.flags : x.g1.flags) & GF_EPS_FREE) &&
                ex.N <= (wide ? 512 : 256))
                   ? 1 : 0;
    }
    x.Ncap = int(c.Ncap);
    x.Acap = int(c.Acap);
    char* rp = res->as<char>();
    x.state = st_mem->as<int>(o.state);
    x.in_cursor = cu_mem->as<int>(o.in_cursor);
    x.queue = sc_mem->as<int>(o.queue);
    x.pair_of = sc_mem->as<int>(o.pair_of);
    x.src = reinterpret_cast<int*>(rp + o.src);
    x.dst = reinterpret_cast<int*>(rp + o.dst);
    x.il = reinterpret_cast<int*>(rp + o.il);
    x.ol = reinterpret_cast<int*>(rp + o.ol);
    x.w = reinterpret_cast<float*>(rp + o.w);
    x.gi1 = reinterpret_cast<int*>(rp + o.gi1);
    x.gi2 = reinterpret_cast<int*>(rp + o.gi2);
    x.nflags = reinterpret_cast<uint8_t*>(rp + o.nf);
    x.out_off = reinterpret_cast<int*>(rp + o.out_off);
    x.level_off = reinterpret_cast<int*>(rp + o.level_off);
    x.in_off = reinterpret_cast<int*>(rp + o.in_off);
    x.in_list = reinterpret_cast<int*>(rp + o.in_list);
    x.in_src = reinterpret_cast<int*>(rp + o.in_src);
    x.in_w = reinterpret_cast<float*>(rp + o.in_w);
    x.start_list = reinterpret_cast<int*>(rp + o.sl);
    x.accept_list = reinterpret_cast<int*>(rp + o.al);
    x.counts = reinterpret_cast<int*>(rp + hdr_cnt) + 2 * i;
    x.out = reinterpret_cast<ComposeOut*>(rp + hdr_out) + i;
  }
  // ---- may the sizes stay on the device (graph.h: DeferredSizes)?  Every pair must be a
  // chain product the FAST variant provably finishes: single-chunk levels (partner no
  // wider than the workgroup), at most KC candidates per node (out-degree), a level's
  // arcs within the claim hash, a bitmap window over all times, arrays left out.
  bool defer = lds_state && n > 0 && !getenv("GTNX_SYNC_COMPOSE");
  for (size_t i = 0; i < n && defer; ++i) defer = !wide_pref[i] && !pairs_pref[i];
  for (size_t i = 0; i < n && defer; ++i) {
    Graph& a = const_cast<Graph&>(bcast(av, n, i));
    Graph& b = const_cast<Graph&>(bcast(bv, n, i));
    const bool l1 = a.s->kind == KIND_LINEAR, l2 = b.s->kind == KIND_LINEAR;
    defer = l1 != l2 && args[i].skip && full_window[i] && chain_slices[i] > 0;
    if (!defer) break;
    Structure& ex = l1 ? *b.s : *a.s;
    ex.ensure_host();
    ex.ensure_csr();
    int max_deg = 0;  // phase B walks in-lists, phase F out-lists: both within KC candidates
    for (int64_t nn = 0; nn < ex.N; ++nn)
      max_deg = std::max(max_deg, std::max(ex.out_off[nn + 1] - ex.out_off[nn], ex.in_off[nn + 1] - ex.in_off[nn]));
    defer = max_deg <= 4 && ex.A <= (wide ? 1536 : 768) && ex.N >= 1 && (l1 ? a : b).s->M >= 1;
  }
  ht_phase("compose.3_alloc_args");
  // Launch groups share a kernel instantiation: (matcher, g1 linear, g2 linear).
  // First pass: the compact LDS-only variant when the pair tables fit; pairs it
  // hands back (overflow == 2: a node with many candidates, an oversized chunk)
  // are re-run with the general variant.
  auto key_of = [&](size_t i) {
    return ((args[i].matcher & 0xff) << 2) | ((args[i].g1.kind == KIND_LINEAR) << 1) | (args[i].g2.kind == KIND_LINEAR);
  };
  std::vector<char> hdr(hdr_cnt + 8 * n);
  const ComposeOut* res_out = reinterpret_cast<const ComposeOut*>(hdr.data() + hdr_out);
  const int* res_counts = reinterpret_cast<const int*>(hdr.data() + hdr_cnt);
  std::shared_ptr<DeferredSizes> deferred;
  // who writes a FAST chain product's stationary levels: the pair's own workgroup (inline), or the replication
  // kernel of compose_wide.hip behind it.  A batch of hundreds of pairs fills the chip with its own workgroups
  // (C3, 512 pairs: 2.95 ms inline, 5.4 ms through the grid); a single utterance has ONE workgroup writing
  // 18 MB (benchmarks/ctc.cpp ctcLoss: 1.95 ms inline, where the grid takes a fraction).
  const char* rep_env = getenv("GTNX_GRID_REPLICATION");
  auto inline_rep_for = [&](size_t pairs) {
    if (getenv("GTNX_INLINE_REPLICATION")) return true;
    if (rep_env) return rep_env[0] == '0';
    return pairs > 128;
  };
  // kind: 0 the general variant, 1 FAST, 2 compose_wide.hip (chain products), 3 compose_wide.hip (explicit pairs)
  auto run = [&](std::vector<size_t> order, int kind) {
    const bool fast = kind == 1;
    const size_t m = order.size();
    const bool inline_rep = inline_rep_for(m);
    double alg = 0;
    for (size_t i : order) alg += 36.0 * double(caps[i].Acap) + 8.0 * double(caps[i].Ncap);
    if (kind == 2)  // one launch per side the chain is on
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return (key_of(x) & 1) < (key_of(y) & 1); });
    else
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return key_of(x) < key_of(y); });
    std::vector<ComposeArgs> sorted_args(m);
    for (size_t i = 0; i < m; ++i) {
      sorted_args[i] = args[order[i]];
      if (!fast) {
        sorted_args[i].skip = 0;
        if (!sorted_args[i].lds_state) fill_state();
      }
      // FAST chain products: stationary levels by the replication kernel behind the compose launch
      sorted_args[i].rep_grid = (fast && !inline_rep && (sorted_args[i].g1.kind == KIND_LINEAR) != (sorted_args[i].g2.kind == KIND_LINEAR)) ? 1 : 0;
      if (kind == 3) {  // the lists the wave-per-pair kernel searches
        fill_state();
        ComposeArgs& x = sorted_args[i];
        Structure& s1 = *bcast(av, n, order[i]).s;
        Structure& s2 = *bcast(bv, n, order[i]).s;
        x.s1_out = x.s1_in = x.s2_out = x.s2_in = nullptr;
        if (x.matcher == MATCH_DOUBLY || x.matcher == MATCH_SINGLY_G1) {
          x.s1_out = x.g1.out_rec;
          x.s1_in = x.g1.in_rec;
        }
        if (x.matcher == MATCH_DOUBLY || x.matcher == MATCH_SINGLY_G2) {
          x.s2_out = x.g2.out_rec;
          x.s2_in = x.g2.in_rec;
        }
        if (x.matcher == MATCH_UNSORTED) {
          x.s2_out = sorted_view(s2, false, false);
          x.s2_in = sorted_view(s2, false, true);
        }
      }
    }
    DevMemP dargs = upload_vec(sorted_args);
    DevMemP tscratch = rt.alloc(compose_transpose_scratch_bytes(int(m), int(maxN)));
    {
      GTNX_PROF(intersect ? "intersect" : "compose", alg);
      for (size_t g0 = 0; g0 < m && kind == 2;) {
        size_t g1 = g0;
        int64_t acap = 0;
        while (g1 < m && (key_of(order[g1]) & 1) == (key_of(order[g0]) & 1)) acap = std::max(acap, caps[order[g1++]].Acap);
        launch_compose_wide(dargs->as<ComposeArgs>() + g0, int(g1 - g0), key_of(order[g0]) & 1, int(acap), rt.stream());
        g0 = g1;
      }
      for (size_t g0 = 0; g0 < m && kind == 3;) {
        size_t g1 = g0;
        while (g1 < m && (key_of(order[g1]) >> 2) == (key_of(order[g0]) >> 2)) ++g1;
        launch_compose_pairs(dargs->as<ComposeArgs>() + g0, int(g1 - g0), key_of(order[g0]) >> 2, rt.stream());
        g0 = g1;
      }
      for (size_t g0 = 0; g0 < m && kind < 2;) {
        size_t g1 = g0;
        while (g1 < m && key_of(order[g1]) == key_of(order[g0])) ++g1;
        const int key = key_of(order[g0]);
        launch_compose(dargs->as<ComposeArgs>() + g0, int(g1 - g0), key >> 2, (key >> 1) & 1, key & 1,
                       fast ? dyn_fast : int(classic_bm), fast ? 1 : 0, (fast && cache1) ? 1 : 0,
                       (fast && wide) ? 1 : 0, rt.stream());
        if (fast && !inline_rep && ((key >> 1) & 1) != (key & 1)) {
          int64_t acap = 0;
          for (size_t q = g0; q < g1; ++q) acap = std::max(acap, caps[order[q]].Acap);
          launch_compose_replicate(dargs->as<ComposeArgs>() + g0, int(g1 - g0), key & 1, int(acap), rt.stream());
        }
        g0 = g1;
      }
    }
    if (defer && fast) {
      // no wait: the header follows the kernel into pinned memory, an event marks it
      rt.drain_deferred();
      deferred = std::make_shared<DeferredSizes>();
      deferred->host = rt.alloc_pinned(hdr.size());
      deferred->hdr_out = hdr_out;
      deferred->hdr_cnt = hdr_cnt;
      HIP_CHECK(hipMemcpyAsync(deferred->host->ptr, res->ptr, hdr.size(), hipMemcpyDeviceToHost, rt.stream()));
      HIP_CHECK(hipEventCreateWithFlags(&deferred->ev, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(deferred->ev, rt.stream()));
      return;
    }
    // sizes back to the host: the contiguous header block, one copy, one sync
    rt.d2h_sync(hdr.data(), res->ptr, hdr.size());
    // products whose in-arc CSR / start & accept lists were not produced inside the
    // compose kernel (non-layered or very wide levels) get them from the parallel
    // transpose passes; the common layered case never launches them
    bool need_tr = false;
    for (size_t i = 0; i < m; ++i) {
      const ComposeOut& co = res_out[order[i]];
      need_tr = need_tr || (!co.csr_built && co.overflow == 0);
    }
    if (need_tr) {
      zero_cursors();
      {
        GTNX_PROF("compose_transpose", 0.0);
        launch_compose_transpose(dargs->as<ComposeArgs>(), int(m), int(maxA), int(maxN), tscratch->ptr, rt.stream());
      }
      rt.d2h_sync(hdr.data(), res->ptr, hdr.size());
    }
  };
  {
    std::vector<size_t> all, wides, pairs;
    for (size_t i = 0; i < n; ++i) (wide_pref[i] ? wides : pairs_pref[i] ? pairs : all).push_back(i);
    if (defer) deferred_limit(1);  // the host runs at most two batches ahead of the GPU
    if (!all.empty()) run(all, lds_state ? 1 : 0);
    if (!wides.empty()) run(wides, 2);
    if (!pairs.empty()) run(pairs, 3);
    // pairs the FAST variant handed back: chain products go to compose_wide.hip whatever their degrees (a node
    // with many IN-arcs stops the FAST variant's backward pass too; bit rows per time do not care), the rest to
    // the general variant
    std::vector<size_t> redo, redo_wide, redo_pairs;
    for (size_t i = 0; i < n && !deferred; ++i)
      if (res_out[i].overflow == 2) {
        const Structure& s1 = *bcast(av, n, i).s;
        const Structure& s2 = *bcast(bv, n, i).s;
        if (wide_ok[i] && !wide_pref[i]) redo_wide.push_back(i);
        else if (pairs_ok[i] && !pairs_pref[i] && (s1.A > 2 * s1.N || s2.A > 2 * s2.N)) redo_pairs.push_back(i);
        else redo.push_back(i);
      }
    if (!redo_wide.empty()) {
      run(redo_wide, 2);
      for (size_t i : redo_wide)
        if (res_out[i].overflow == 2) redo.push_back(i);
    }
    if (!redo_pairs.empty()) run(redo_pairs, 3);
    if (getenv("GTNX_COMPOSE_STATS") && !deferred)
      fprintf(stderr, "[gtnx] compose: n=%zu redo=%zu graph0: N=%d A=%d levels=%d replicated=%d  us: B=%.0f F=%.0f (rep %.0f)\n", n, redo.size(),
              res_out[0].N, res_out[0].A, res_out[0].L, res_out[0].rep_levels, res_out[0].t_b * 0.01,
              res_out[0].t_f * 0.01, res_out[0].t_rep * 0.01);
    if (!redo.empty()) run(redo, 0);
  }

  ht_phase("compose.4_launch_wait");
  auto op = std::make_shared<ComposeOp>();
  op->seq = next_seq();
  op->arena = res;
  op->saved.resize(n);
  for (size_t i = 0; i < n; ++i) {
    ComposeOut co = deferred ? ComposeOut{} : res_out[i];
    if (deferred) {  // what the proven fast path guarantees; the numbers come later
      co.layered = 1;
      co.csr_built = 1;
      co.skipped = 1;
      co.N = co.A = -1;
    }
    if (co.overflow) throw_runtime("[gtn::compose] internal capacity bound exceeded");
    const ComposeArgs& x = args[i];
    Graph& a = const_cast<Graph&>(bcast(av, n, i));
    Graph& b = const_cast<Graph&>(bcast(bv, n, i));
    Graph out = make_output(op, int(i), {a, b});
    Structure& s = *out.s;
    s.kind = KIND_EXPLICIT;
    s.N = co.N;
    s.A = co.A;
    s.host_valid = false;
    s.dev_valid = true;
    s.dev_mem = res;
    DGraph& v = s.dview;
    std::memset(&v, 0, sizeof(v));
    v.kind = KIND_EXPLICIT;
    v.N = co.N;
    v.A = co.A;
    v.n_start = deferred ? -1 : res_counts[2 * i];
    v.n_accept = deferred ? -1 : res_counts[2 * i + 1];
    // a product's labels come from its inputs' arcs (epsilon only where an input had one)
    v.flags = (x.g1.flags & x.g2.flags & (GF_EPS_FREE | GF_ACCEPTOR));
    v.src = x.src;
    v.dst = x.dst;
    v.il = x.il;
    v.ol = x.ol;
    v.nflags = x.nflags;
    v.start_list = x.start_list;
    v.accept_list = x.accept_list;
    v.out_off = x.out_off;
    v.out_list = nullptr;  // arcs are grouped by source in id order
    v.in_off = x.in_off;
    v.in_list = x.in_list;
    set_dev_weights(out, res, x.w, co.A);
    if (co.skipped) {
      auto pi = std::make_shared<PartialInfo>();
      ComposeFillArgs& f = pi->args;
      f.N = co.N;
      f.A = co.A;
      f.out_off = x.out_off;
      f.dst = x.dst;
      f.w = x.w;
      f.gi1 = x.gi1;
      f.gi2 = x.gi2;
      f.lab1 = a.s->kind == KIND_LINEAR ? nullptr : x.g1.il;
      f.lab2 = b.s->kind == KIND_LINEAR ? nullptr : x.g2.ol;
      f.C1 = a.s->kind == KIND_LINEAR ? a.s->C : 1;
      f.C2 = b.s->kind == KIND_LINEAR ? b.s->C : 1;
      f.src = x.src;
      f.il = x.il;
      f.ol = x.ol;
      f.in_list = x.in_list;
      f.in_src = x.in_src;
      f.in_w = x.in_w;
      pi->in1 = a.s;
      pi->in2 = b.s;
      pi->keep1 = a.s->dev_mem;
      pi->keep2 = b.s->dev_mem;
      s.partial = pi;
    }
    if (co.layered) {
      auto sc = std::make_shared<Schedule>();
      sc->mem = res;
      sc->n_in = co.A;
      sc->n_out = co.A;
      sc->all_written = true;
      sc->has_rank = true;  // rank == arc id for src-sorted arcs
      sc->max_level_width = co.max_width;
      sc->max_level_arcs = co.max_level_arcs;
      sc->max_reach = 2 * co.max_width;  // in-arcs come from the previous level only
      DSched& d = sc->view;
      d.P = co.N;
      d.L = co.L;
      d.n_accept = v.n_accept;
      d.flags = SCHED_TIE_BY_ARC | SCHED_OUT_IDENTITY;
      d.level_off = x.level_off;
      d.row_off = x.in_off;
      d.in_srcpos = x.in_src;
      d.in_arc = x.in_list;
      d.in_rank = nullptr;
      d.in_w = nullptr;
      d.pflags = x.nflags;
      d.acc_pos = x.accept_list;
      d.out_off = x.out_off;
      d.out_dstpos = x.dst;
      d.out_arc = nullptr;
      sc->in_w = x.in_w;
      sc->in_w_of = out.w.get();
      sc->in_w_version = out.w->version;
      sc->dyn_out = x.out;
      sc->dyn_counts = x.counts;
      // exactly one implicit chain and an epsilon-free partner: level == chain time
      const bool l1 = a.s->kind == KIND_LINEAR, l2 = b.s->kind == KIND_LINEAR;
      if (deferred) {
        // bounds in place of the numbers (the kernels read the real ones on the device)
        const Structure& ex = l1 ? *b.s : *a.s;
        const Structure& ch = l1 ? *a.s : *b.s;
        sc->n_in = sc->n_out = caps[i].Acap;
        sc->max_level_width = int(ex.N);
        sc->max_level_arcs = int(caps[i].Acap / std::max(ch.M, 1));
        sc->max_reach = 2 * int(ex.N);
        d.P = int(caps[i].Ncap);
        d.L = ch.M + 1;
        d.n_accept = 0;
      }
      if (l1 != l2 && ((l1 ? x.g2.flags : x.g1.flags) & GF_EPS_FREE) && out.calc_grad()) {
        sc->producer_seq = op->seq;
        sc->chain_side = l1 ? 1 : 2;
        sc->chain_C = l1 ? a.s->C : b.s->C;
        sc->fixed_A = l1 ? b.num_arcs() : a.num_arcs();
        sc->gi_fixed = l1 ? x.gi2 : x.gi1;
        sc->gi_chain = l1 ? x.gi1 : x.gi2;
      }
      s.sched = sc;
    }
    op->saved[i] = {x.gi1, x.gi2, co.A};
    if (deferred) {
      s.deferred = deferred;
      s.deferred_idx = int(i);
      s.capN = caps[i].Ncap;
      s.capA = caps[i].Acap;
      deferred->members.push_back({out.s, out.w});
    }
    outs.push_back(std::move(out));
  }
  if (deferred) {
    op->deferred = deferred;
    deferred_register(deferred);
  }
  ht_phase("compose.5_outputs");
  return outs;
}


// ======================================================================
// Lazy chain products (kernels: lazy.hip).  compose(chain, G) / compose(G, chain)
// with an implicit linear chain and an epsilon-free G is kept SYMBOLIC when building
// it is infeasible (or GTNX_LAZY_COMPOSE=1): forwardScore / viterbiScore /
// viterbiPath and their gradients then run as time-synchronous dynamic programs
// over (t, node of G), batched over the utterances that share G.  Any other use of
// the result (inspection, another op) realises it through the ordinary compose.
// ======================================================================
namespace {

struct LazyComposeOp : OpRecord {
  void backward(std::vector<Member>& ms) override {
    for (auto& m : ms) {
      if (!m.out.g->grad_propagated)
        throw_logic("[gtn::compose] internal: gradient reached an unrealised lazy product");
      m.out.g->grad_propagated = false;
    }
  }
};

// everything the forward pass of one group leaves behind
struct LazyGroupState {
  LazyGroup view{};                 // host copy of the kernel argument
  DevMemP arena;                    // alpha / bp / score / best / em pointer table
  DevMemP labels;                   // node_label (null when in-arc labels differ per node)
  const int* node_label = nullptr;
  int max_in_deg = 0;
  bool dense = false;               // probability-domain products (lazy.hip "dense regime")
  bool mfma = false;                // ... on the matrix cores (v_mfma_f32_32x32x2_f32)
  bool maxplus = false;             // tropical semiring over a dense G (maxplus.hip): no back-pointer planes
  bool lab_unique = false;          // no two nodes of G share a matched label
  DevMemP dense_mem;
  Graph fixed;                      // keeps G alive
  std::vector<Graph> chains;        // per member
  std::vector<int> member_of;       // output index -> member slot (filled by the caller)
};

int lazy_lds_limit() { return 150 * 1024; }

std::shared_ptr<OpRecord> make_lazy_compose_op() {
  auto op = std::make_shared<LazyComposeOp>();
  op->seq = next_seq();
  return op;
}

bool lazy_shape_ok(const Structure& chain, const Structure& fixed) {
  const int64_t np = (fixed.N | 1) + 0, cp = (int64_t(chain.C) | 1);
  // (a chain without a step has no accept node, creations.cpp:20-33: left to the ordinary compose)
  return size_t(lazy_tile_batch()) * size_t(np + cp) * 4 <= size_t(lazy_lds_limit()) && fixed.N > 0 && chain.C > 0 &&
         chain.M >= 1;
}

struct LazyKey {
  Structure* fs;
  Weights* fw;
  int T, C, side;
  bool operator<(const LazyKey& o) const {
    return std::tie(fs, fw, T, C, side) < std::tie(o.fs, o.fw, o.T, o.C, o.side);
  }
};

// forward pass (log or tropical) of every lazy product in `gs`; returns one state per
// group and, through `slot`, (group, member) of each input
std::vector<int> lazy_node_labels(Structure& fs, bool chain_first, int C, int* max_in_deg);
// host facts about a fixed partner G for the dense regime, taken once per structure (a trainer keeps its
// transitions graph; only the weights move)
bool labels_unique(const std::vector<int>& lab) {
  std::unordered_set<int> seen;
  for (int l : lab)
    if (l >= 0 && !seen.insert(l).second) return false;
  return true;
}
std::shared_ptr<Structure::DenseInfo> dense_info(Structure& fs, bool chain_first, int C) {
  std::shared_ptr<Structure::DenseInfo>& slot = fs.dense[chain_first ? 0 : 1];
  if (slot && slot->C == C) return slot;
  auto di = std::make_shared<Structure::DenseInfo>();
  di->C = C;
  fs.ensure_host();
  fs.ensure_csr();
  di->lab = lazy_node_labels(fs, chain_first, C, &di->max_in_deg);
  const std::vector<int>& ml = chain_first ? fs.il : fs.ol;
  for (int l : ml) di->valid += (l >= 0 && l < C);
  if (!di->lab.empty()) {
    const int N = int(fs.N);
    std::vector<int> tab(size_t(N), -1), colnode, dead;
    for (int n = 0; n < N; ++n) {
      if (di->lab[size_t(n)] >= 0) {
        tab[size_t(n)] = int(colnode.size());
        colnode.push_back(n);
      } else {
        dead.push_back(n);
      }
    }
    di->ncol = int(colnode.size());
    di->ndead = int(dead.size());
    di->uniq = labels_unique(di->lab);
    std::vector<int> ints(di->lab);
    ints.insert(ints.end(), tab.begin(), tab.end());
    ints.insert(ints.end(), colnode.begin(), colnode.end());
    ints.insert(ints.end(), dead.begin(), dead.end());
    di->tables = upload_vec(ints);
  }
  slot = di;
  return di;
}
std::vector<std::shared_ptr<LazyGroupState>> lazy_forward(std::vector<Graph>& gs, int mode,
                                                          std::vector<std::pair<int, int>>& slot) {
  Runtime& rt = Runtime::get();
  std::map<LazyKey, int> index;
  std::vector<std::shared_ptr<LazyGroupState>> groups;
  slot.resize(gs.size());
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (size_t i = 0; i < gs.size(); ++i) {
    LazyProduct& lp = *gs[i].s->lazy;
    LazyKey k{lp.fixed.s.get(), lp.fixed.w.get(), lp.chain.s->M, lp.chain.s->C, lp.chain_side};
    auto it = index.find(k);
    if (it == index.end()) {
      it = index.emplace(k, int(groups.size())).first;
      auto st = std::make_shared<LazyGroupState>();
      st->fixed = lp.fixed;
      st->view.chain_first = lp.chain_side == 1;
      groups.push_back(st);
      ss.push_back(lp.fixed.s.get());
      ws.push_back(lp.fixed.w.get());
    }
    LazyGroupState& st = *groups[it->second];
    slot[i] = {it->second, int(st.chains.size())};
    st.chains.push_back(lp.chain);
    ws.push_back(lp.chain.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  for (auto& gp : groups) {
    LazyGroupState& st = *gp;
    Structure& fs = *st.fixed.s;
    ensure_records(fs);
    const Structure& cs = *st.chains[0].s;
    const int nb = int(st.chains.size());
    const int T = cs.M, C = cs.C, N = int(fs.N);
    LazyGroup& v = st.view;
    v.g = device_view(st.fixed);
    v.T = T;
    v.C = C;
    v.N = N;
    v.nb = nb;
    v.Npad = N | 1;
    v.Cpad = C | 1;
    // host facts about G: shared in-arc label per node, widest in-row
    fs.ensure_host();
    fs.ensure_csr();
    const size_t plane = size_t(nb) * size_t(N);
    size_t bytes = 0;
    auto add = [&](size_t b) {
      size_t o = bytes;
      bytes = align_up(bytes + b, 256);
      return o;
    };
    // tropical semiring over a dense G whose nodes' in-arcs share one matched label: the max-plus sweeps of
    // maxplus.hip (decided here because they need no back-pointer planes)
    std::shared_ptr<Structure::DenseInfo> di;
    if (!getenv("GTNX_NO_DENSE") && N <= 1024 && N >= 8) di = dense_info(fs, st.view.chain_first != 0, C);
    const bool dense_ok = di && !di->lab.empty() && 2 * di->valid >= int64_t(N) * N;
    if (di) st.max_in_deg = di->max_in_deg;
    if (di) st.lab_unique = di->uniq;
    st.maxplus = mode == SD_TROPICAL && dense_ok && T >= 1 && di->ncol > 0;
    const size_t o_alpha = add(4 * plane * size_t(T + 1));
    const size_t o_bp = (mode == SD_LOG || st.maxplus) ? 0 : add(4 * plane * size_t(T + 1));
    const size_t o_score = add(4 * size_t(nb));
    const size_t o_best = add(4 * size_t(nb));
    const size_t o_em = add(8 * size_t(nb));
    const size_t o_lin = add(16 * size_t(fs.A));
    const size_t o_lout = add(16 * size_t(fs.A));
    st.arena = rt.alloc(bytes);
    v.lrec_in = st.arena->as<gtnx_i4>(o_lin);
    v.lrec_out = st.arena->as<gtnx_i4>(o_lout);
    v.alpha = st.arena->as<float>(o_alpha);
    v.bp = (mode == SD_LOG || st.maxplus) ? nullptr : st.arena->as<int>(o_bp);
    v.score = st.arena->as<float>(o_score);
    v.best = st.arena->as<int>(o_best);
    std::vector<const float*> em(nb);
    for (int b = 0; b < nb; ++b) em[b] = st.chains[b].w->dev;
    PinnedMemP pin = rt.alloc_pinned(8 * size_t(nb));
    std::memcpy(pin->ptr, em.data(), 8 * size_t(nb));
    rt.h2d(st.arena->as<char>(o_em), pin->ptr, 8 * size_t(nb));
    v.em = reinterpret_cast<const float* const*>(st.arena->as<char>(o_em));
    // slices of one tensor (linearGraphs over a [B][T][C] tensor, the criteria): the kernels' inner loops
    // compute the row address instead of loading it
    v.em_base = nullptr;
    v.em_stride = 0;
    if (nb >= 1 && em[0]) {
      const int64_t stride = nb > 1 ? em[1] - em[0] : int64_t(T) * C;
      bool strided = stride >= int64_t(T) * C;
      for (int b = 1; b < nb && strided; ++b) strided = em[b] - em[b - 1] == stride;
      if (strided) {
        v.em_base = em[0];
        v.em_stride = stride;
      }
    }
    if (st.maxplus) {
      // columns = nodes with a matched in-arc; the others (an ASG start node) are -inf from step 1 on
      v.mp_ncol = di->ncol;
      v.mp_ndead = di->ndead;
      v.Kpad = (N + 3) & ~3;
      v.nbpad = (nb + 63) & ~63;
      st.labels = di->tables;
      st.node_label = st.labels->as<int>();
      v.nlab = st.node_label;
      v.mp_colidx = st.node_label + N;
      v.mp_colnode = v.mp_colidx + N;
      v.mp_dead = v.mp_colnode + v.mp_ncol;
      const size_t wf = maxplus_w_floats(v), xf = size_t(v.Kpad) * size_t(v.nbpad);
      st.dense_mem = rt.alloc(4 * (align_up(wf, 64) + 2 * align_up(xf, 64)));
      float* base = st.dense_mem->as<float>();
      v.mp_Wq = base;
      v.xt[0] = base + align_up(wf, 64);
      v.xt[1] = v.xt[0] + align_up(xf, 64);
    }
  }
  // dense regime? (log semiring, one label per node's in-arcs, G nearly complete)
  for (size_t i = 0; i < gs.size(); ++i) groups[slot[i].first]->view.chain_first = gs[i].s->lazy->chain_side == 1;
  for (auto& gp : groups) {
    LazyGroupState& st = *gp;
    LazyGroup& v = st.view;
    Structure& fs = *st.fixed.s;
    if (mode != SD_LOG || getenv("GTNX_NO_DENSE") || v.N > 1024 || v.N < 8) continue;
    std::shared_ptr<Structure::DenseInfo> di = dense_info(fs, v.chain_first != 0, v.C);
    st.max_in_deg = di->max_in_deg;
    st.lab_unique = di->uniq;
    const std::vector<int>& lab = di->lab;
    if (lab.empty()) continue;
    if (2 * di->valid < int64_t(v.N) * v.N) continue;
    st.labels = di->tables;
    st.node_label = st.labels->as<int>();
    const size_t nn = size_t(v.N) * size_t(v.N);
    size_t bytes = 0;
    auto add = [&](size_t b) {
      size_t o2 = bytes;
      bytes = align_up(bytes + b, 256);
      return o2;
    };
    const size_t o_E = add(4 * nn), o_c = add(4 * size_t(v.N)), o_am = add(4 * size_t(v.T + 1) * size_t(v.nb)),
                 o_bm = add(4 * size_t(v.T + 1) * size_t(v.nb));
    // matrix-core form (lazy.hip: lazy_mfma_*): padded E and its transpose, two transposed input planes
    v.rot = 0;
    while (v.rot < v.N - 1 && lab[size_t(v.rot)] < 0) ++v.rot;
    v.Kpad = (v.N + 575) / 576 * 576;  // zero rows up to an even number of operand batches per wave (lazy.hip: 4 k x 8 waves x 2 x 9 groups)
    v.Npad2 = (v.N + 31) & ~31;
    v.nbpad = (v.nb + 31) & ~31;
    const size_t o_Ep = add(4 * size_t(v.Kpad) * size_t(v.Npad2)), o_ETp = add(4 * size_t(v.Kpad) * size_t(v.Npad2)),
                 o_x0 = add(4 * size_t(v.Kpad) * size_t(v.nbpad)), o_x1 = add(4 * size_t(v.Kpad) * size_t(v.nbpad));
    // row maxima as one partial per column tile (the step kernels store, the consumers reduce: no atomics)
    v.ntp = (((v.N - v.rot + 31) / 32) + 3) & ~3;
    const size_t o_amp = add(4 * size_t(v.T + 1) * size_t(v.nb) * size_t(v.ntp)),
                 o_bmp = add(4 * size_t(v.T + 1) * size_t(v.nb) * size_t(v.ntp));
    st.dense_mem = rt.alloc(bytes);
    v.amaxp = st.dense_mem->as<float>(o_amp);
    v.bmaxp = st.dense_mem->as<float>(o_bmp);
    v.E = st.dense_mem->as<float>(o_E);
    v.cmax = st.dense_mem->as<float>(o_c);
    v.nlab = st.node_label;
    v.amax = st.dense_mem->as<float>(o_am);
    v.bmax = st.dense_mem->as<float>(o_bm);
    v.Ep = st.dense_mem->as<float>(o_Ep);
    v.ETp = st.dense_mem->as<float>(o_ETp);
    v.xt[0] = st.dense_mem->as<float>(o_x0);
    v.xt[1] = st.dense_mem->as<float>(o_x1);
    st.dense = true;
    st.mfma = getenv("GTNX_DENSE_VALU") == nullptr;
  }
  // chain_first comes from the products themselves (same for a whole group by key)
  for (size_t i = 0; i < gs.size(); ++i) groups[slot[i].first]->view.chain_first = gs[i].s->lazy->chain_side == 1;
  for (auto& gp : groups) {
    LazyGroupState& st = *gp;
    GTNX_PROF(mode == SD_LOG ? "lazy_forward_score" : (st.maxplus ? "maxplus_viterbi" : "lazy_viterbi"), 0.0);
    launch_lazy_pack(st.view, const_cast<gtnx_i4*>(st.view.lrec_in), const_cast<gtnx_i4*>(st.view.lrec_out), rt.stream());
    launch_lazy_init(st.view, 0, rt.stream());
    if (st.dense) {
      launch_lazy_dense_prep(st.view, const_cast<float*>(st.view.E), const_cast<float*>(st.view.cmax), rt.stream());
      if (st.mfma) {
        launch_lazy_mfma_prep(st.view, rt.stream());
        launch_lazy_mfma_init(st.view, 0, rt.stream());
        {
          DevMemP sync = rt.alloc_zero(sizeof(int) * lazy_mfma_chain_sync_ints(st.view));
          if (!launch_lazy_mfma_chain(st.view, 0, sync->as<int>(), rt.cu_count(), rt.stream()))
            for (int t = 0; t < st.view.T; ++t) launch_lazy_mfma_step(st.view, t, 0, rt.stream());
        }
        launch_lazy_mfma_rowmax(st.view, 0, rt.stream());
      } else {
        for (int t = 0; t < st.view.T; ++t) launch_lazy_dense_step(st.view, t, 0, rt.stream());
      }
    } else if (st.maxplus) {
      launch_maxplus_prep(st.view, rt.stream());
      for (int t = 0; t < st.view.T; ++t) launch_maxplus_step(st.view, t, rt.stream());
    } else {
      for (int t = 0; t < st.view.T; ++t) launch_lazy_step(st.view, t, mode, 0, rt.stream());
    }
    launch_lazy_final(st.view, mode, rt.stream());
  }
  return groups;
}

// shared in-arc label of every node of G (matched side), or empty if some node's differ
std::vector<int> lazy_node_labels(Structure& fs, bool chain_first, int C, int* max_in_deg) {
  std::vector<int> lab(size_t(fs.N), -1);
  bool moore = true;
  int md = 0;
  for (int64_t n = 0; n < fs.N; ++n) {
    md = std::max(md, fs.in_off[n + 1] - fs.in_off[n]);
    for (int k = fs.in_off[n]; k < fs.in_off[n + 1]; ++k) {
      const int a = fs.in_list[k];
      const int l = chain_first ? fs.il[a] : fs.ol[a];
      if (l < 0 || l >= C) continue;
      if (lab[n] == -1) lab[n] = l;
      else if (lab[n] != l) moore = false;
    }
  }
  *max_in_deg = md;
  if (!moore) lab.clear();
  return lab;
}

struct LazySdOp : OpRecord {
  int mode;
  std::vector<std::shared_ptr<LazyGroupState>> groups;
  std::vector<std::pair<int, int>> slot;  // output index -> (group, member)

  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    GradSink sink;
    // members by group
    std::vector<std::vector<Member*>> by_group(groups.size());
    for (auto& m : ms) by_group[slot[m.idx].first].push_back(&m);
    DevMemP zero = rt.alloc_zero(256);
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      if (by_group[gi].empty()) continue;
      LazyGroupState& st = *groups[gi];
      LazyGroup v = st.view;
      const int nb = v.nb, T = v.T, C = v.C, N = v.N;
      const size_t plane = size_t(nb) * size_t(N);
      Graph& fixed = st.fixed;
      // per-member pointers: upstream delta, chain gradient buffer
      std::vector<const float*> delta(nb, zero->as<float>());
      std::vector<float*> gem(nb, nullptr);
      size_t gbytes = 0;
      std::vector<size_t> goff(nb, 0);
      std::vector<Member*> of_slot(nb, nullptr);
      for (Member* m : by_group[gi]) of_slot[slot[m->idx].second] = m;
      for (int b = 0; b < nb; ++b) {
        if (!of_slot[b]) continue;
        delta[b] = grad_dev_ptr(of_slot[b]->out);
        if (st.chains[b].calc_grad()) {
          goff[b] = gbytes;
          gbytes = align_up(gbytes + 4 * size_t(T) * size_t(C), 256);
        }
      }
      const bool want_fixed = fixed.calc_grad();
      const size_t o_fixed = gbytes;
      if (want_fixed) gbytes = align_up(gbytes + 4 * size_t(fixed.num_arcs()), 256);
      DevMemP gmem = rt.alloc_zero(gbytes ? gbytes : 1);
      for (int b = 0; b < nb; ++b)
        if (of_slot[b] && st.chains[b].calc_grad()) gem[b] = gmem->as<float>(goff[b]);
      v.grad_fixed = want_fixed ? gmem->as<float>(o_fixed) : nullptr;
      // pointer tables
      DevMemP tabs = rt.alloc(16 * size_t(nb));
      PinnedMemP pin = rt.alloc_pinned(16 * size_t(nb));
      std::memcpy(pin->as<char>(), delta.data(), 8 * size_t(nb));
      std::memcpy(pin->as<char>(8 * size_t(nb)), gem.data(), 8 * size_t(nb));
      rt.h2d(tabs->ptr, pin->ptr, 16 * size_t(nb));
      v.delta = reinterpret_cast<const float* const*>(tabs->as<char>());
      v.grad_em = reinterpret_cast<float* const*>(tabs->as<char>(8 * size_t(nb)));
      if (mode == SD_LOG) {
        DevMemP beta = rt.alloc(4 * plane * size_t(T + 1));
        v.beta = beta->as<float>();
        GTNX_PROF("lazy_forward_score_grad", 0.0);
        launch_lazy_init(v, 1, rt.stream());
        if (st.dense && st.mfma) {
          launch_lazy_mfma_init(v, 1, rt.stream());
          {
            DevMemP sync = rt.alloc_zero(sizeof(int) * lazy_mfma_chain_sync_ints(v));
            if (!launch_lazy_mfma_chain(v, 1, sync->as<int>(), rt.cu_count(), rt.stream()))
              for (int t = T - 1; t >= 0; --t) launch_lazy_mfma_step(v, t, 1, rt.stream());
          }
          launch_lazy_mfma_rowmax(v, 1, rt.stream());
        } else if (st.dense) {
          DevMemP vs = rt.alloc(8 * plane);  // two planes: input of this step / of the next
          float* vb[2] = {vs->as<float>(), vs->as<float>() + plane};
          for (int t = T - 1; t >= 0; --t)
            launch_lazy_dense_step(v, t, 1, rt.stream(), t == T - 1 ? nullptr : vb[(t + 1) & 1], vb[t & 1]);
        } else {
          for (int t = T - 1; t >= 0; --t) launch_lazy_step(v, t, SD_LOG, 1, rt.stream());
        }
        if (!st.labels && st.max_in_deg == 0) {
          fixed.s->ensure_host();
          fixed.s->ensure_csr();
          std::vector<int> lab = lazy_node_labels(*fixed.s, v.chain_first != 0, C, &st.max_in_deg);
          if (!lab.empty()) {
            st.lab_unique = labels_unique(lab);
            st.labels = upload_vec(lab);
            st.node_label = st.labels->as<int>();
          }
        }
        DevMemP ztm = rt.alloc(4 * size_t(T > 0 ? T : 1) * size_t(nb));
        if (st.node_label && lazy_z_chain_grad_ok(v)) {
          v.lab_unique = st.lab_unique ? 1 : 0;
          launch_lazy_z_chain_grad(v, st.node_label, ztm->as<float>(), rt.stream());
          v.zt = ztm->as<float>();
        } else {
          launch_lazy_local_z(v, ztm->as<float>(), rt.stream());
          v.zt = ztm->as<float>();
          launch_lazy_chain_grad(v, st.node_label, rt.stream());
        }
        if (want_fixed && st.dense) {
          DevMemP rmem = rt.alloc_zero(4 * size_t(N) * size_t(N));
          v.R = rmem->as<float>();
          if (st.mfma) {
            DevMemP pcm = rt.alloc(16 * size_t(T > 0 ? T : 1) * size_t(nb));
            launch_lazy_mfma_fixed_grad(v, pcm->ptr, rt.stream());
          } else {
            launch_lazy_dense_fixed_grad(v, rt.stream());
          }
        } else if (want_fixed) {
          const size_t lds = lazy_step_lds_bytes(v) + 4 * size_t(lazy_tile_nodes()) * size_t(st.max_in_deg);
          if (lds > size_t(lazy_lds_limit()))
            throw_runtime("[gtn::backward] lazy product: graph too wide for the arc-gradient kernel");
          launch_lazy_fixed_grad(v, st.max_in_deg, rt.stream());
        }
        (void)beta;  // released after the launches are queued (stream-ordered pool)
      } else {
        // viterbiScore (shortest.cpp:65-74, tropical): one-hot along the best path
        const size_t pbytes = size_t(nb) * size_t(T) * 16 + 4 * size_t(nb);
        DevMemP pm = rt.alloc(pbytes ? pbytes : 1);
        int* parc = pm->as<int>();
        int* pil = parc + size_t(nb) * T;
        int* pol = pil + size_t(nb) * T;
        float* pw = reinterpret_cast<float*>(pol + size_t(nb) * T);
        int* plen = reinterpret_cast<int*>(pw + size_t(nb) * T);
        launch_lazy_path(v, parc, pil, pol, pw, plen, rt.stream());
        std::vector<int> lens(nb);
        rt.d2h_sync(lens.data(), plen, 4 * size_t(nb));
        for (int b = 0; b < nb; ++b) {
          if (!of_slot[b] || lens[b] <= 0) continue;
          LazyPathGrad a{};
          a.delta = delta[b];
          a.delta_stride = 0;
          a.path_arc = parc + size_t(b) * T;
          a.il = pil + size_t(b) * T;
          a.ol = pol + size_t(b) * T;
          a.len = lens[b];
          a.C = C;
          a.chain_first = v.chain_first;
          a.grad_chain = gem[b];
          a.grad_fixed = v.grad_fixed;
          launch_lazy_path_grad(a, rt.stream());
        }
      }
      for (int b = 0; b < nb; ++b) {
        if (!of_slot[b]) continue;
        if (gem[b]) sink.add(st.chains[b], gmem, gem[b]);
        of_slot[b]->out.g->inputs[0].g->grad_propagated = true;
      }
      if (want_fixed) sink.add(fixed, gmem, v.grad_fixed);
    }
    sink.flush();
  }
};

std::vector<Graph> lazy_group_shortest_distance(std::vector<Graph>& gs, bool tropical) {
  auto op = std::make_shared<LazySdOp>();
  op->mode = tropical ? SD_TROPICAL : SD_LOG;
  op->seq = next_seq();
  op->groups = lazy_forward(gs, op->mode, op->slot);
  std::vector<Graph> outs;
  for (size_t i = 0; i < gs.size(); ++i) {
    LazyGroupState& st = *op->groups[op->slot[i].first];
    Graph out = make_output(op, int(i), {gs[i]});
    init_scalar_result(out);
    set_dev_weights(out, st.arena, st.view.score + op->slot[i].second, 1);
    outs.push_back(std::move(out));
  }
  return outs;
}

// ---- one workgroup per (chain, small G) pair: lazy_pair.hip.  The CTC shape -- every
// utterance brings its own target graph -- where the batched time-step kernels above
// (one G shared by the batch) would run one launch per utterance and time step.
bool lazy_pair_shape_ok(const Structure& cs, Structure& fs) {
  if (fs.kind != KIND_EXPLICIT || fs.N < 1 || fs.N > lazy_pair_max_nodes() || cs.C < 1 || cs.M < 1) return false;
  if (cs.C > lazy_pair_max_labels(lazy_pair_block(int(fs.N)))) return false;
  return fs.max_degree() <= lazy_pair_max_degree();
}
bool lazy_pair_ok(const LazyProduct& lp) { return lazy_pair_shape_ok(*lp.chain.s, *lp.fixed.s); }

struct LazyPairSdOp : OpRecord {
  std::vector<LazyPair> pairs;  // by output index; device pointers
  std::vector<Graph> chains, fixed;
  DevMemP arena;                // alpha planes + scores

  // launches `tab` (any order) grouped by label count, widest G of a group picks the block
  static void launch(std::vector<LazyPair>& tab, bool backward) {
    Runtime& rt = Runtime::get();
    if (tab.empty()) return;
    std::stable_sort(tab.begin(), tab.end(), [](const LazyPair& x, const LazyPair& y) { return x.C < y.C; });
    DevMemP d = upload_vec(tab);
    const LazyPair* dp = d->as<LazyPair>();
    for (size_t i0 = 0; i0 < tab.size();) {
      size_t i1 = i0;
      int maxn = 0;
      while (i1 < tab.size() && tab[i1].C == tab[i0].C) maxn = std::max(maxn, tab[i1++].g.N);
      const int blk = lazy_pair_block(maxn);
      if (backward)
        launch_lazy_pair_backward(dp + i0, int(i1 - i0), blk, tab[i0].C, rt.cu_count(), rt.stream());
      else
        launch_lazy_pair_forward(dp + i0, int(i1 - i0), blk, tab[i0].C, rt.cu_count(), rt.stream());
      i0 = i1;
    }
  }

  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    GradSink sink;
    size_t eb = 0, fb = 0;
    std::vector<size_t> eo(ms.size(), 0), fo(ms.size(), 0);
    std::vector<float*> dest(ms.size(), nullptr);
    std::vector<DevMemP> dest_mem(ms.size());
    for (size_t k = 0; k < ms.size(); ++k) {
      const int i = ms[k].idx;
      if (chains[i].calc_grad()) {
        GradState& cg = *chains[i].g;
        if (cg.grad_dest && !chains[i].is_grad_available()) {  // first gradient: straight into the caller's tensor
          dest[k] = cg.grad_dest;
          dest_mem[k] = cg.grad_dest_mem;
          cg.grad_dest = nullptr;  // (a second sweep over the same chain accumulates onto it)
        } else {
          eo[k] = eb;
          eb = align_up(eb + 4 * size_t(pairs[i].T) * size_t(pairs[i].C), 256);
        }
      }
      if (fixed[i].calc_grad()) {
        fo[k] = fb;
        fb = align_up(fb + 4 * size_t(pairs[i].g.A), 256);
      }
    }
    DevMemP gem = rt.alloc(eb ? eb : 1);       // every row is written by the kernel
    DevMemP gfx = rt.alloc_zero(fb ? fb : 1);  // arcs that never match stay 0
    std::vector<LazyPair> tab;
    tab.reserve(ms.size());
    for (size_t k = 0; k < ms.size(); ++k) {
      const int i = ms[k].idx;
      LazyPair p = pairs[i];
      p.delta = grad_dev_ptr(ms[k].out);
      p.grad_em = chains[i].calc_grad() ? (dest[k] ? dest[k] : gem->as<float>(eo[k])) : nullptr;
      p.grad_fixed = fixed[i].calc_grad() ? gfx->as<float>(fo[k]) : nullptr;
      tab.push_back(p);
      if (p.grad_em) sink.add(chains[i], dest[k] ? dest_mem[k] : gem, p.grad_em);
      if (p.grad_fixed) sink.add(fixed[i], gfx, p.grad_fixed);
      ms[k].out.g->inputs[0].g->grad_propagated = true;
    }
    {
      // algorithmic bytes: emissions in, emission gradient out, alpha back in, G's arc gradients out
      double bytes = 0;
      for (const LazyPair& p : tab)
        bytes += 4.0 * p.T * p.C * (p.grad_em ? 2 : 1) + 4.0 * double(p.T + 1) * p.g.N + (p.grad_fixed ? 4.0 * p.g.A : 0.0);
      GTNX_PROF("lazy_pair_forward_score_grad", bytes);
      launch(tab, true);
    }
    sink.flush();
  }
};

std::vector<Graph> lazy_pair_forward_score(std::vector<Graph>& gs) {
  Runtime& rt = Runtime::get();
  auto op = std::make_shared<LazyPairSdOp>();
  op->seq = next_seq();
  const size_t n = gs.size();
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (size_t i = 0; i < n; ++i) {
    LazyProduct& lp = *gs[i].s->lazy;
    op->chains.push_back(lp.chain);
    op->fixed.push_back(lp.fixed);
    ss.push_back(lp.fixed.s.get());
    ws.push_back(lp.fixed.w.get());
    ws.push_back(lp.chain.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  for (Structure* st : ss) ensure_records(*st);
  size_t bytes = align_up(4 * n, 256);
  std::vector<size_t> ao(n);
  for (size_t i = 0; i < n; ++i) {
    ao[i] = bytes;
    bytes = align_up(bytes + 4 * size_t(op->chains[i].s->M + 1) * size_t(op->fixed[i].s->N), 256);
  }
  op->arena = rt.alloc(bytes);
  op->pairs.resize(n);
  for (size_t i = 0; i < n; ++i) {
    LazyPair& p = op->pairs[i];
    p = LazyPair{};
    p.g = device_view(op->fixed[i]);
    p.em = op->chains[i].w->dev;
    p.alpha = op->arena->as<float>(ao[i]);
    p.score = op->arena->as<float>(4 * i);
    p.T = op->chains[i].s->M;
    p.C = op->chains[i].s->C;
    p.chain_first = gs[i].s->lazy->chain_side == 1;
  }
  {
    double bytes = 0;  // algorithmic: emissions in, alpha out (kept for the backward sweep)
    for (const LazyPair& p : op->pairs) bytes += 4.0 * p.T * p.C + 4.0 * double(p.T + 1) * p.g.N;
    GTNX_PROF("lazy_pair_forward_score", bytes);
    std::vector<LazyPair> tab = op->pairs;
    LazyPairSdOp::launch(tab, false);
  }
  std::vector<Graph> outs;
  outs.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    Graph out = make_output(op, int(i), {gs[i]});
    init_scalar_result(out);
    set_dev_weights(out, op->arena, op->pairs[i].score, 1);
    outs.push_back(std::move(out));
  }
  return outs;
}

} // namespace
int band_vec_ok(const BandPair& p) { return p.C % 4 == 0 && (reinterpret_cast<uintptr_t>(p.em) & 15) == 0; }
// launches `tab` grouped by (C, nodes per lane, unit, G wants a gradient, 16-byte staging)
void band_launch(std::vector<std::pair<BandLaunchKey, BandPair>>& tab, bool backward) {
  Runtime& rt = Runtime::get();
  if (tab.empty()) return;
  std::stable_sort(tab.begin(), tab.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  std::vector<BandPair> flat;
  flat.reserve(tab.size());
  for (auto& e : tab) flat.push_back(e.second);
  DevMemP d = upload_vec(flat);
  const BandPair* dp = d->as<BandPair>();
  for (size_t i0 = 0; i0 < tab.size();) {
    size_t i1 = i0;
    int max_ns = 0;
    while (i1 < tab.size() && tab[i1].first == tab[i0].first) max_ns = std::max(max_ns, tab[i1++].second.NS);
    const BandLaunchKey& k = tab[i0].first;
    if (backward)
      launch_band_backward(dp + i0, int(i1 - i0), k.npl, k.C, max_ns, k.unit != 0, k.gradg != 0, k.vec != 0, rt.stream());
    else
      launch_band_forward(dp + i0, int(i1 - i0), k.npl, k.C, max_ns, k.unit != 0, k.vec != 0, rt.stream());
    i0 = i1;
  }
}
namespace {
// ---- one workgroup per (chain, BANDED G) pair: band.hip.  CTC targets and force-alignment
// acceptors: a single wave carries the whole recursion, the other waves stage.
bool band_shape_ok(const Structure& cs, Structure& fs, bool chain_first) {
  if (cs.kind != KIND_LINEAR || cs.C < band_min_labels() || cs.C > band_max_labels() || cs.M < 0 || cs.M > (1 << 20)) return false;
  std::shared_ptr<BandInfo> b = band_info(fs, chain_first);
  return b->ok && b->max_label < cs.C;
}
bool band_ok(const LazyProduct& lp, std::shared_ptr<BandInfo>* out = nullptr) {
  if (!band_shape_ok(*lp.chain.s, *lp.fixed.s, lp.chain_side == 1)) return false;
  if (out) *out = band_info(*lp.fixed.s, lp.chain_side == 1);
  return true;
}
// band records and the all-zero test of a batch of partners, on the worker pool (a training step
// brings one fresh target graph per utterance); both are cached on the graph afterwards
void band_prepare(const std::vector<Graph*>& fixed, const std::vector<uint8_t>& chain_first) {
  std::vector<size_t> todo;
  std::unordered_set<Structure*> seen;
  for (size_t i = 0; i < fixed.size(); ++i) {
    Structure* st = fixed[i]->s.get();
    if (!st->band[chain_first[i] ? 0 : 1] && seen.insert(st).second) todo.push_back(i);
  }
  auto body = [&](size_t q) {
    Graph& g = *fixed[todo[q]];
    band_info(*g.s, chain_first[todo[q]] != 0);
    (void)g.w->is_all_zero();
  };
  if (todo.size() >= 64) gtn::detail::runIndexed(todo.size(), body, 32);
  else for (size_t q = 0; q < todo.size(); ++q) body(q);
}

struct BandSdOp : OpRecord {
  std::vector<BandPair> pairs;  // by output index; device pointers
  std::vector<Graph> chains, fixed;
  std::vector<std::shared_ptr<BandInfo>> infos;
  std::vector<uint8_t> unit;    // unit-shaped G with all-zero weights
  DevMemP arena;                // alpha planes, row shifts, scores

  using Key = BandLaunchKey;
  static int band_vec(const BandPair& p) { return band_vec_ok(p); }
  static void launch(std::vector<std::pair<Key, BandPair>>& tab, bool backward) { band_launch(tab, backward); }

  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    size_t eb = 0, fb = 0;
    std::vector<size_t> eo(ms.size(), 0), fo(ms.size(), 0);
    std::vector<float*> dest(ms.size(), nullptr);
    std::vector<DevMemP> dest_mem(ms.size());
    for (size_t k = 0; k < ms.size(); ++k) {
      const int i = ms[k].idx;
      if (chains[i].calc_grad()) {
        GradState& cg = *chains[i].g;
        if (cg.grad_dest && !chains[i].is_grad_available()) {  // first gradient: straight into the caller's tensor
          dest[k] = cg.grad_dest;
          dest_mem[k] = cg.grad_dest_mem;
          cg.grad_dest = nullptr;  // (a second sweep over the same chain accumulates onto it)
        } else {
          eo[k] = eb;
          eb = align_up(eb + 4 * size_t(pairs[i].T) * size_t(pairs[i].C), 256);
        }
      }
      if (fixed[i].calc_grad()) {
        fo[k] = fb;
        fb = align_up(fb + 4 * size_t(fixed[i].s->A), 256);
      }
    }
    DevMemP gem = rt.alloc(eb ? eb : 1);       // every row is written by the kernel
    DevMemP gfx = rt.alloc_zero(fb ? fb : 1);  // arcs that never match stay 0
    ChainGradPlan local;
    ChainGradPlan& plan = t_chain_plan ? *t_chain_plan : local;
    plan.keep.push_back(gem);
    plan.keep.push_back(gfx);
    plan.keep.push_back(arena);
    for (size_t k = 0; k < ms.size(); ++k) {  // what the deferred launch reads must outlive this record
      plan.keep.push_back(infos[ms[k].idx]->dev_mem);
      plan.keep.push_back(chains[ms[k].idx].w->dev_mem);
      plan.keep.push_back(fixed[ms[k].idx].w->dev_mem);
    }
    for (size_t k = 0; k < ms.size(); ++k) {
      const int i = ms[k].idx;
      BandPair p = pairs[i];
      p.delta = grad_dev_ptr(ms[k].out);
      p.delta_norm = nullptr;
      p.grad_em = chains[i].calc_grad() ? (dest[k] ? dest[k] : gem->as<float>(eo[k])) : nullptr;
      p.grad_fixed = fixed[i].calc_grad() ? gfx->as<float>(fo[k]) : nullptr;
      plan.band.push_back({p.C, band_npl(p.N), int(unit[i]), p.grad_fixed ? 1 : 0, band_vec(p), p, chains[i].w.get()});
      if (p.grad_em) plan.sink.add(chains[i], dest[k] ? dest_mem[k] : gem, p.grad_em);
      if (p.grad_fixed) plan.sink.add(fixed[i], gfx, p.grad_fixed);
      ms[k].out.g->inputs[0].g->grad_propagated = true;
      // algorithmic bytes: emissions in, emission gradient out, alpha back in, G's arc gradients out
      plan.bytes += 4.0 * p.T * p.C * (p.grad_em ? 2 : 1) + 4.0 * double(p.T + 1) * p.NS +
                    (p.grad_fixed ? 4.0 * double(fixed[i].s->A) : 0.0);
    }
    if (!t_chain_plan) {  // not inside backward(): launch at once
      t_chain_plan = &local;
      flush_chain_plan();
      t_chain_plan = nullptr;
    }
  }
  bool joins_chain_plan() const override { return true; }
};

// launches what the records of one backward() registered: band sweeps (with the softmax term of the
// normaliser where forwardScore(emissions) of the same chain is on the tape too), then the normalisers
// that found no sweep to ride with
void flush_chain_plan() {
  ChainGradPlan* plan = t_chain_plan;
  if (!plan || plan->empty()) return;
    std::vector<std::pair<BandSdOp::Key, BandPair>> tab;
  tab.reserve(plan->band.size());
  for (auto& b : plan->band) {
    auto it = b.p.grad_em ? plan->lin.find(b.chain_w) : plan->lin.end();
    if (it != plan->lin.end() && !it->second.fused) {
      b.p.delta_norm = it->second.delta;
      b.p.rowlse = const_cast<float*>(it->second.rowlse);
      it->second.fused = true;
    } else {
      b.p.delta_norm = nullptr;
    }
    tab.push_back({BandSdOp::Key{b.C, b.npl, b.unit, b.gradg, b.vec}, b.p});
  }
  if (!tab.empty()) {
    GTNX_PROF("band_forward_score_grad", plan->bytes);
    BandSdOp::launch(tab, true);
  }
  plan->sink.flush();
  // normalisers without a sweep: their own kernel
  std::vector<Member> rest;
  std::vector<Graph> rest_in;
  for (auto& kv : plan->lin)
    if (!kv.second.fused) {
      rest.push_back(kv.second.m);
      rest_in.push_back(kv.second.chain);
    }
  if (!rest.empty()) {
    LinearSdOp lin;
    lin.tropical = false;
    lin.run_now(rest, rest_in);
  }
  plan->band.clear();
  plan->lin.clear();
  plan->keep.clear();
  plan->bytes = 0;
}

std::vector<Graph> band_forward_score(std::vector<Graph>& gs) {
  Runtime& rt = Runtime::get();
  auto op = std::make_shared<BandSdOp>();
  op->seq = next_seq();
  const size_t n = gs.size();
  std::vector<BandInfo*> bis;
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  op->unit.resize(n);
  {
    std::vector<Graph*> fx(n);
    std::vector<uint8_t> cf(n);
    for (size_t i = 0; i < n; ++i) {
      fx[i] = &gs[i].s->lazy->fixed;
      cf[i] = gs[i].s->lazy->chain_side == 1;
    }
    band_prepare(fx, cf);
  }
  op->chains.reserve(n);
  op->fixed.reserve(n);
  op->infos.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    LazyProduct& lp = *gs[i].s->lazy;
    std::shared_ptr<BandInfo> b;
    band_ok(lp, &b);
    op->chains.push_back(lp.chain);
    op->fixed.push_back(lp.fixed);
    op->infos.push_back(b);
    bis.push_back(b.get());
    ss.push_back(lp.fixed.s.get());
    const bool zero = lp.fixed.w->is_all_zero();
    op->unit[i] = zero && b->unit_shape;
    if (!zero) ws.push_back(lp.fixed.w.get());
    ws.push_back(lp.chain.w.get());
  }
  ensure_band_device_batch(bis, ss);
  ensure_weights_device_batch(ws);
  // scores [n], then per pair: the chain's own forwardScore (a by-product: every emission is read
  // anyway) + its per-row log-sum-exps, shifts, alpha plane
  size_t bytes = align_up(8 * n, 256);
  std::vector<size_t> ao(n), oo(n), lo(n);
  for (size_t i = 0; i < n; ++i) {
    const int T = op->chains[i].s->M, N = int(op->fixed[i].s->N);
    const int ns = band_row_stride(N, band_npl(N));
    lo[i] = bytes;
    bytes = align_up(bytes + 4 * size_t(T > 0 ? T : 1), 256);
    oo[i] = bytes;
    bytes = align_up(bytes + 8 * (4 * size_t(T) + 16), 256);  // score + one shift per wave and period (>= 1 row)
    ao[i] = bytes;
    bytes = align_up(bytes + 4 * size_t(T + 1) * size_t(ns), 256);
  }
  op->arena = rt.alloc(bytes);
  op->pairs.resize(n);
  std::vector<std::pair<BandSdOp::Key, BandPair>> tab;
  tab.reserve(n);
  double abytes = 0;
  for (size_t i = 0; i < n; ++i) {
    BandPair& p = op->pairs[i];
    p = BandPair{};
    const BandInfo& b = *op->infos[i];
    p.nodes = b.dev;
    p.nflags = b.dev_flags;
    p.snode = b.dev_snode;
    p.slab = b.dev_slab;
    p.n_lab = int(b.snode.size());
    p.w = op->fixed[i].w->is_all_zero() ? nullptr : op->fixed[i].w->dev;
    p.em = op->chains[i].w->dev;
    p.N = int(op->fixed[i].s->N);
    p.T = op->chains[i].s->M;
    p.C = op->chains[i].s->C;
    p.NS = band_row_stride(p.N, band_npl(p.N));
    p.alpha = op->arena->as<float>(ao[i]);
    p.aoff = op->arena->as<double>(oo[i]);
    p.score = op->arena->as<float>(4 * i);
    if (!op->chains[i].w->valid_norm_cache()) {
      p.norm = op->arena->as<float>(4 * (n + i));
      p.rowlse = op->arena->as<float>(lo[i]);
    }
    p.hot = b.hot;
    p.lgrn = band_forward_lgrn(p.C);
    tab.push_back({BandSdOp::Key{p.C, band_npl(p.N), int(op->unit[i]), 0, BandSdOp::band_vec(p)}, p});
    abytes += 4.0 * p.T * p.C + 4.0 * double(p.T + 1) * p.NS;  // emissions in, alpha out (kept for the backward sweep)
  }
  {
    GTNX_PROF("band_forward_score", abytes);
    BandSdOp::launch(tab, false);
  }
  for (size_t i = 0; i < n; ++i) {
    const BandPair& p = op->pairs[i];
    if (!p.norm) continue;
    auto nc = std::make_shared<NormCache>();
    nc->version = op->chains[i].w->version;
    nc->mem = op->arena;
    nc->norm = p.norm;
    nc->rowlse = p.rowlse;
    op->chains[i].w->norm_cache = std::move(nc);
  }
  std::vector<Graph> outs;
  outs.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    Graph out = make_output(op, int(i), {gs[i]});
    init_scalar_result(out);
    set_dev_weights(out, op->arena, op->pairs[i].score, 1);
    outs.push_back(std::move(out));
  }
  return outs;
}

std::vector<Graph> band_viterbi(std::vector<Graph>& gs, bool want_path);
std::vector<Graph> lazy_shortest_distance(std::vector<Graph>& gs, bool tropical) {
  if (!tropical && !getenv("GTNX_NO_LAZY_PAIRS")) {
    std::vector<Graph> br, pr, gr;
    std::vector<size_t> bi, pi, gi;
    const bool band = getenv("GTNX_NO_BAND") == nullptr;  // read per call: tests flip it at run time
    for (size_t i = 0; i < gs.size(); ++i) {
      if (band && band_ok(*gs[i].s->lazy)) { br.push_back(gs[i]); bi.push_back(i); }
      else if (lazy_pair_ok(*gs[i].s->lazy)) { pr.push_back(gs[i]); pi.push_back(i); }
      else { gr.push_back(gs[i]); gi.push_back(i); }
    }
    if (br.size() == gs.size()) return band_forward_score(br);
    if (!pr.empty() || !br.empty()) {
      std::vector<Graph> outs(gs.size(), Graph(false));
      if (!br.empty()) {
        std::vector<Graph> bo = band_forward_score(br);
        for (size_t k = 0; k < bi.size(); ++k) outs[bi[k]] = std::move(bo[k]);
      }
      std::vector<Graph> po = pr.empty() ? std::vector<Graph>() : lazy_pair_forward_score(pr);
      for (size_t k = 0; k < pi.size(); ++k) outs[pi[k]] = std::move(po[k]);
      if (!gr.empty()) {
        std::vector<Graph> go = lazy_group_shortest_distance(gr, tropical);
        for (size_t k = 0; k < gi.size(); ++k) outs[gi[k]] = std::move(go[k]);
      }
      return outs;
    }
  }
  if (tropical && !getenv("GTNX_NO_BAND")) {  // banded partners: one launch per batch, back-pointers and all
    std::vector<Graph> br, gr;
    std::vector<size_t> bi, gi;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (band_ok(*gs[i].s->lazy)) { br.push_back(gs[i]); bi.push_back(i); }
      else { gr.push_back(gs[i]); gi.push_back(i); }
    }
    if (!br.empty()) {
      std::vector<Graph> outs(gs.size(), Graph(false));
      std::vector<Graph> bo = band_viterbi(br, false);
      for (size_t k = 0; k < bi.size(); ++k) outs[bi[k]] = std::move(bo[k]);
      if (!gr.empty()) {
        std::vector<Graph> go = lazy_group_shortest_distance(gr, tropical);
        for (size_t k = 0; k < gi.size(); ++k) outs[gi[k]] = std::move(go[k]);
      }
      return outs;
    }
  }
  return lazy_group_shortest_distance(gs, tropical);
}

struct LazyPathOp : OpRecord {
  struct Saved {
    std::vector<int> arcs, il, ol;  // first-arc-first
    int C = 0, chain_first = 0;
  };
  std::vector<Saved> saved;
  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    GradSink sink;
    for (auto& m : ms) {
      const Saved& sv = saved[m.idx];
      Graph& comp = m.out.g->inputs[0];
      comp.g->grad_propagated = true;
      const int len = int(sv.arcs.size());
      if (len == 0 || comp.g->inputs.size() != 2) continue;
      Graph& chain = comp.g->inputs[sv.chain_first ? 0 : 1];
      Graph& fixed = comp.g->inputs[sv.chain_first ? 1 : 0];
      std::vector<int> packed;
      packed.insert(packed.end(), sv.arcs.begin(), sv.arcs.end());
      packed.insert(packed.end(), sv.il.begin(), sv.il.end());
      packed.insert(packed.end(), sv.ol.begin(), sv.ol.end());
      DevMemP dp = upload_vec(packed);
      size_t bytes = 0;
      const size_t oc = bytes;
      if (chain.calc_grad()) bytes = align_up(bytes + 4 * size_t(chain.num_arcs()), 256);
      const size_t of = bytes;
      if (fixed.calc_grad()) bytes = align_up(bytes + 4 * size_t(fixed.num_arcs()), 256);
      DevMemP gm = rt.alloc_zero(bytes ? bytes : 1);
      LazyPathGrad a{};
      a.delta = grad_dev_ptr(m.out);
      a.delta_stride = 1;
      a.path_arc = dp->as<int>();
      a.il = a.path_arc + len;
      a.ol = a.il + len;
      a.len = len;
      a.C = sv.C;
      a.chain_first = sv.chain_first;
      a.grad_chain = chain.calc_grad() ? gm->as<float>(oc) : nullptr;
      a.grad_fixed = fixed.calc_grad() ? gm->as<float>(of) : nullptr;
      launch_lazy_path_grad(a, rt.stream());
      if (a.grad_chain) sink.add(chain, gm, a.grad_chain);
      if (a.grad_fixed) sink.add(fixed, gm, a.grad_fixed);
    }
    sink.flush();
  }
};

// ---- viterbiScore / viterbiPath of a symbolic chain o (banded G): band_viterbi_kernel
// gradient of viterbiScore: the best path's arcs, d score each (shortest.cpp:64-81 on the built lattice)
struct BandViterbiScoreOp : OpRecord {
  struct Saved {
    DevMemP mem;
    const int *arc = nullptr, *lab = nullptr;
    int len = -1, C = 0, chain_first = 0;
  };
  std::vector<Saved> saved;
  void backward(std::vector<Member>& ms) override {
    Runtime& rt = Runtime::get();
    GradSink sink;
    for (auto& m : ms) {
      const Saved& sv = saved[m.idx];
      Graph& comp = m.out.g->inputs[0];
      comp.g->grad_propagated = true;
      if (sv.len <= 0 || comp.g->inputs.size() != 2) continue;
      Graph& chain = comp.g->inputs[sv.chain_first ? 0 : 1];
      Graph& fixed = comp.g->inputs[sv.chain_first ? 1 : 0];
      size_t bytes = 0;
      const size_t oc = bytes;
      if (chain.calc_grad()) bytes = align_up(bytes + 4 * size_t(chain.num_arcs()), 256);
      const size_t of = bytes;
      if (fixed.calc_grad()) bytes = align_up(bytes + 4 * size_t(fixed.num_arcs()), 256);
      if (!bytes) continue;
      DevMemP gm = rt.alloc_zero(bytes);
      LazyPathGrad a{};
      a.delta = grad_dev_ptr(m.out);
      a.delta_stride = 0;
      a.path_arc = sv.arc;
      a.il = a.ol = sv.lab;  // the matched label either way
      a.len = sv.len;
      a.C = sv.C;
      a.chain_first = sv.chain_first;
      a.grad_chain = chain.calc_grad() ? gm->as<float>(oc) : nullptr;
      a.grad_fixed = fixed.calc_grad() ? gm->as<float>(of) : nullptr;
      launch_lazy_path_grad(a, rt.stream());
      if (a.grad_chain) sink.add(chain, gm, a.grad_chain);
      if (a.grad_fixed) sink.add(fixed, gm, a.grad_fixed);
    }
    sink.flush();
  }
};

std::vector<Graph> lazy_viterbi_path(std::vector<Graph>& gs);
std::vector<Graph> band_viterbi(std::vector<Graph>& gs, bool want_path) {
  Runtime& rt = Runtime::get();
  const size_t n = gs.size();
  std::vector<BandInfo*> bis;
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  std::vector<std::shared_ptr<BandInfo>> infos(n);
  {
    std::vector<Graph*> fx(n);
    std::vector<uint8_t> cf(n);
    for (size_t i = 0; i < n; ++i) {
      fx[i] = &gs[i].s->lazy->fixed;
      cf[i] = gs[i].s->lazy->chain_side == 1;
    }
    band_prepare(fx, cf);
  }
  for (size_t i = 0; i < n; ++i) {
    LazyProduct& lp = *gs[i].s->lazy;
    band_ok(lp, &infos[i]);
    bis.push_back(infos[i].get());
    ss.push_back(lp.fixed.s.get());
    if (!lp.fixed.w->is_all_zero()) ws.push_back(lp.fixed.w.get());
    ws.push_back(lp.chain.w.get());
  }
  ensure_band_device_batch(bis, ss);
  ensure_weights_device_batch(ws);
  // per pair: back-pointers [T][NS] bytes | pnode [T+1] | path arc, label, weight [T] each | len, score, tie
  size_t bytes = 0;
  std::vector<size_t> o_bp(n), o_pn(n), o_pa(n), o_hd(n);
  int max_c = 1;
  for (size_t i = 0; i < n; ++i) {
    const LazyProduct& lp = *gs[i].s->lazy;
    const size_t T = size_t(lp.chain.s->M), N = size_t(lp.fixed.s->N);
    const size_t ns = size_t(band_row_stride(int(N), band_npl(int(N))));
    o_bp[i] = bytes;
    bytes = align_up(bytes + T * ns + 1, 256);
    o_pn[i] = bytes;
    bytes = align_up(bytes + 4 * (T + 1), 256);
    o_pa[i] = bytes;
    bytes = align_up(bytes + 12 * (T ? T : 1), 256);
    o_hd[i] = bytes;
    bytes = align_up(bytes + 16, 256);
    max_c = std::max(max_c, lp.chain.s->C);
  }
  DevMemP arena = rt.alloc(bytes);
  const int stage_floats = std::max(4096, max_c);
  std::vector<BandDecode> tab(n);
  for (size_t i = 0; i < n; ++i) {
    const LazyProduct& lp = *gs[i].s->lazy;
    const BandInfo& b = *infos[i];
    BandDecode& p = tab[i];
    p = BandDecode{};
    p.nodes = b.dev;
    p.nflags = b.dev_flags;
    p.w = lp.fixed.w->is_all_zero() ? nullptr : lp.fixed.w->dev;
    p.em = lp.chain.w->dev;
    p.N = int(lp.fixed.s->N);
    p.T = lp.chain.s->M;
    p.C = lp.chain.s->C;
    p.NS = band_row_stride(p.N, band_npl(p.N));
    p.bp = arena->as<uint8_t>(o_bp[i]);
    p.pnode = arena->as<int>(o_pn[i]);
    p.path_arc = arena->as<int>(o_pa[i]);
    p.path_lab = p.path_arc + (p.T ? p.T : 1);
    p.path_w = reinterpret_cast<float*>(p.path_lab + (p.T ? p.T : 1));
    p.path_len = arena->as<int>(o_hd[i]);
    p.score = reinterpret_cast<float*>(p.path_len + 1);
    p.tie = p.path_len + 2;
    p.stage_floats = stage_floats;
  }
  {
    DevMemP d = upload_vec(tab);
    GTNX_PROF(want_path ? "band_viterbi_path" : "band_viterbi_score", 0.0);
    launch_band_viterbi(d->as<BandDecode>(), int(n), stage_floats, rt.stream());
  }
  // heads (length, score, tie) of every pair; the paths themselves only when they become graphs
  std::vector<char> host(bytes);
  if (want_path) {
    rt.d2h_sync(host.data(), arena->ptr, bytes);
  } else {
    DevMemP heads = rt.alloc(16 * n);
    std::vector<AxpyArgs> ax;
    for (size_t i = 0; i < n; ++i) ax.push_back({heads->as<float>(16 * i), reinterpret_cast<float*>(tab[i].path_len), 3, 1.0f});
    DevMemP d = upload_vec(ax);
    launch_axpy_batch(d->as<AxpyArgs>(), int(n), 3, /*copy*/ 2, rt.stream());
    std::vector<char> hh(16 * n);
    rt.d2h_sync(hh.data(), heads->ptr, 16 * n);
    for (size_t i = 0; i < n; ++i) std::memcpy(host.data() + o_hd[i], hh.data() + 16 * i, 12);
  }
  std::vector<Graph> outs(n, Graph(false));
  std::vector<size_t> tied;
  std::shared_ptr<LazyPathOp> pop;
  std::shared_ptr<BandViterbiScoreOp> sop;
  if (want_path) {
    pop = std::make_shared<LazyPathOp>();
    pop->seq = next_seq();
    pop->saved.resize(n);
  } else {
    sop = std::make_shared<BandViterbiScoreOp>();
    sop->seq = next_seq();
    sop->saved.resize(n);
  }
  for (size_t i = 0; i < n; ++i) {
    const int* hd = reinterpret_cast<const int*>(host.data() + o_hd[i]);
    const int len = hd[0];
    if (hd[2] && len >= 0) {  // an exact tie: the built lattice decides (its node numbering breaks it)
      tied.push_back(i);
      continue;
    }
    LazyProduct& lp = *gs[i].s->lazy;
    const int chain_first = lp.chain_side == 1;
    if (want_path) {
      Graph out = make_output(pop, int(i), {gs[i]});
      if (len >= 0) {
        const int* harc = reinterpret_cast<const int*>(host.data() + o_pa[i]);
        const int* hlab = harc + (tab[i].T ? tab[i].T : 1);
        const float* hw = reinterpret_cast<const float*>(hlab + (tab[i].T ? tab[i].T : 1));
        // labels of the product's arcs: the chain's on its side, G's arc label on the other
        lp.fixed.s->ensure_host();
        std::vector<int> il, ol;
        il.resize(size_t(len));
        ol.resize(size_t(len));
        for (int t = 0; t < len; ++t) {
          il[size_t(t)] = chain_first ? hlab[t] : lp.fixed.s->il[size_t(harc[t])];
          ol[size_t(t)] = chain_first ? lp.fixed.s->ol[size_t(harc[t])] : hlab[t];
        }
        fill_path_graph(out, len, true, il.data(), ol.data(), hw);
        LazyPathOp::Saved& sv = pop->saved[i];
        sv.arcs.assign(harc, harc + len);
        sv.il = std::move(il);
        sv.ol = std::move(ol);
      }
      pop->saved[i].C = tab[i].C;
      pop->saved[i].chain_first = chain_first;
      outs[i] = std::move(out);
    } else {
      Graph out = make_output(sop, int(i), {gs[i]});
      init_scalar_result(out);
      set_dev_weights(out, arena, tab[i].score, 1);
      BandViterbiScoreOp::Saved& sv = sop->saved[i];
      sv.mem = arena;
      sv.arc = tab[i].path_arc;
      sv.lab = tab[i].path_lab;
      sv.len = len;
      sv.C = tab[i].C;
      sv.chain_first = chain_first;
      outs[i] = std::move(out);
    }
  }
  if (!tied.empty()) {
    std::vector<Graph> tg;
    for (size_t i : tied) {
      // The lattice is built and its level schedule taken by replaying the reference's queue on it
      // (graph.cpp: build_host_schedule, as for any host-built graph) instead of the id-order schedule a
      // layered product normally gets for free: under exact ties the winner is the arc whose source left
      // the queue first (shortest.cpp:212-227), and that order is not the node-id order.
      realize(gs[i]);
      gs[i].s->resolve_sizes();
      gs[i].s->ensure_full();
      gs[i].s->ensure_host();
      gs[i].s->sched.reset();
      tg.push_back(gs[i]);
    }
    std::vector<Graph> to = want_path ? op_viterbi_path(tg) : op_shortest_distance(tg, true);
    for (size_t k = 0; k < tied.size(); ++k) outs[tied[k]] = std::move(to[k]);
  }
  return outs;
}

std::vector<Graph> lazy_viterbi_path(std::vector<Graph>& gs) {
  if (!getenv("GTNX_NO_BAND")) {
    std::vector<Graph> br, gr;
    std::vector<size_t> bi, gi;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (band_ok(*gs[i].s->lazy)) { br.push_back(gs[i]); bi.push_back(i); }
      else { gr.push_back(gs[i]); gi.push_back(i); }
    }
    if (!br.empty()) {
      std::vector<Graph> outs(gs.size(), Graph(false));
      std::vector<Graph> bo = band_viterbi(br, true);
      for (size_t k = 0; k < bi.size(); ++k) outs[bi[k]] = std::move(bo[k]);
      if (!gr.empty()) {
        std::vector<Graph> go = lazy_viterbi_path(gr);
        for (size_t k = 0; k < gi.size(); ++k) outs[gi[k]] = std::move(go[k]);
      }
      return outs;
    }
  }
  Runtime& rt = Runtime::get();
  std::vector<std::pair<int, int>> slot;
  GTNX_HOST_T("lazy_viterbi_path.total");
  std::vector<std::shared_ptr<LazyGroupState>> groups;
  {
    GTNX_HOST_T("lazy_viterbi_path.1_forward_enqueue");
    groups = lazy_forward(gs, SD_TROPICAL, slot);
  }
  auto op = std::make_shared<LazyPathOp>();
  op->seq = next_seq();
  op->saved.resize(gs.size());
  std::vector<Graph> outs(gs.size(), Graph(false));
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    LazyGroupState& st = *groups[gi];
    const LazyGroup& v = st.view;
    const size_t nT = size_t(v.nb) * size_t(v.T);
    const size_t pbytes = nT * 16 + 4 * size_t(v.nb);
    DevMemP pm = rt.alloc(pbytes ? pbytes : 1);
    int* parc = pm->as<int>();
    int* pil = parc + nT;
    int* pol = pil + nT;
    float* pw = reinterpret_cast<float*>(pol + nT);
    int* plen = reinterpret_cast<int*>(pw + nT);
    launch_lazy_path(v, parc, pil, pol, pw, plen, rt.stream());
    PinnedMemP host = rt.alloc_pinned(pbytes ? pbytes : 1);  // 16 B per path arc: pageable memory would be staged and slow
    {
      GTNX_HOST_T("lazy_viterbi_path.2_wait_download");
      rt.d2h_sync(host->ptr, pm->ptr, pbytes);
    }
    GTNX_HOST_T("lazy_viterbi_path.3_path_graphs");
    const int* harc = host->as<int>();
    const int* hil = harc + nT;
    const int* hol = hil + nT;
    const float* hw = reinterpret_cast<const float*>(hol + nT);
    const int* hlen = reinterpret_cast<const int*>(hw + nT);
    // the path graphs (8 host arrays of T entries each per utterance): every element touches only its own
    // objects, so a large batch is built by a few threads (3.5 -> <1 ms of a 17 ms decode at C4)
    auto build = [&](size_t i) {
      if (slot[i].first != int(gi)) return;
      const int b = slot[i].second;
      const int len = hlen[b];
      Graph out = make_output(op, int(i), {gs[i]});
      // shortest.cpp:248-260; no accepting path -> the empty graph
      if (len >= 0) {
        const size_t o0 = size_t(b) * size_t(v.T);
        fill_path_graph(out, len, true, hil + o0, hol + o0, hw + o0);
        LazyPathOp::Saved& sv = op->saved[i];
        sv.arcs.assign(harc + o0, harc + o0 + len);
        sv.il.assign(hil + o0, hil + o0 + len);
        sv.ol.assign(hol + o0, hol + o0 + len);
      }
      op->saved[i].C = v.C;
      op->saved[i].chain_first = v.chain_first;
      outs[i] = std::move(out);
    };
    const size_t n_out = gs.size();
    const size_t nthreads = (n_out * size_t(v.T) >= (size_t(1) << 16)) ? std::min<size_t>(8, n_out) : 1;
    if (nthreads <= 1) {
      for (size_t i = 0; i < n_out; ++i) build(i);
    } else {
      std::atomic<size_t> next{0};
      std::exception_ptr err;
      std::mutex err_mu;
      auto worker = [&] {
        try {
          for (size_t i = next.fetch_add(1); i < n_out; i = next.fetch_add(1)) build(i);
        } catch (...) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (!err) err = std::current_exception();
        }
      };
      std::vector<std::thread> pool;
      for (size_t k = 1; k < nthreads; ++k) pool.emplace_back(worker);
      worker();
      for (auto& th : pool) th.join();
      if (err) std::rethrow_exception(err);
    }
  }
  return outs;
}

} // namespace

// turn a symbolic product into the ordinary materialised one, in place
std::vector<Graph> op_compose_impl(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect, bool allow_lazy);
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
namespace {
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
} // namespace

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
namespace {
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
}  // namespace

void backward_validate(Graph& root) {
  std::vector<Graph> roots{root};
  Tape tape;
  collect_tape(roots, tape);
}

void op_backward(std::vector<Graph>& roots, Graph* grad, bool retain, bool seed) {
  GTNX_HOST_T("backward.total");
  Runtime& rt = Runtime::get();
  for (auto& r : roots) realize(r);
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
    if (!members.empty()) kv.second.first->backward(members);
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


// ======================================================================
// rational operations (functions.cpp:66-223), built on the device: rational.hip
// ======================================================================
namespace {
struct RationalOp : OpRecord {
  // per output: where each input's arcs start in the output's arc order (functions.cpp:98-110, 159-164, 191-200:
  // the gradient of an input is a slice of the deltas)
  std::vector<std::vector<int64_t>> arc_off;
  void backward(std::vector<Member>& ms) override {
    // addGrad COPIES the slice (graph.cpp:91-129): an input's gradient must not alias the output's buffer -- a
    // retained tape run twice, concat({g, g}) or a later accumulation into the input would otherwise write into
    // the output's gradient too.  One arena for the record's slices, one batched copy, then the sink adopts it.
    Runtime& rt = Runtime::get();
    GradSink sink;
    std::vector<CopySeg> segs;
    struct Item {
      Graph* in;
      size_t off;
    };
    std::vector<Item> items;
    size_t total = 0;
    int64_t longest = 0;
    for (auto& m : ms) {
      Graph& gr = m.out.grad();
      if (!gr.w->dev_valid || gr.w->host_escaped) {
        std::vector<Weights*> v{gr.w.get()};
        ensure_weights_device_batch(v);
      }
      auto& ins = m.out.g->inputs;
      for (size_t i = 0; i < ins.size(); ++i) {
        if (!ins[i].calc_grad()) continue;
        const int64_t bytes = int64_t(sizeof(float)) * ins[i].num_arcs();
        items.push_back({&ins[i], total});
        segs.push_back({nullptr, gr.w->dev + arc_off[m.idx][i], bytes});
        total += align_up(size_t(bytes), 16);
        longest = std::max(longest, bytes);
      }
    }
    if (items.empty()) return;
    DevMemP arena = rt.alloc(total ? total : 16);
    for (size_t k = 0; k < items.size(); ++k) segs[k].dst = arena->as<char>(items[k].off);
    DevMemP d = upload_vec(segs);
    launch_copy_segments(d->as<CopySeg>(), int(segs.size()), longest, rt.stream());
    for (auto& it : items) sink.add(*it.in, arena, arena->as<float>(it.off));
    sink.flush();
  }
};
}  // namespace

Graph op_rational(int kind, std::vector<Graph>& ins, int projection) {
  Runtime& rt = Runtime::get();
  const bool closure = kind == RAT_CLOSURE, concat = kind == RAT_CONCAT;
  auto op = std::make_shared<RationalOp>();
  op->seq = next_seq();
  if (ins.empty()) {  // a^0 accepts the empty string (functions.cpp:117-121); the empty union is the empty graph
    Graph out = make_output(op, 0, {});
    if (concat) out.add_node(true, true);
    op->arc_off.push_back({});
    return out;
  }
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (auto& g : ins) {
    g.s->resolve_sizes();
    if (g.s->kind != KIND_LINEAR) ss.push_back(g.s.get());
    ws.push_back(g.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  const int k = int(ins.size());
  std::vector<RationalSeg> segs;
  segs.resize(size_t(k));
  int64_t N = closure ? 1 : 0, A = 0;
  int max_A = 0, max_N = 0, max_conn = 0;
  bool eps_free = true;
  std::vector<int64_t> offs;
  for (int i = 0; i < k; ++i) {
    RationalSeg& s = segs[size_t(i)];
    s = RationalSeg{};
    s.g = device_view(ins[size_t(i)]);
    if (s.g.kind == KIND_LINEAR) {
      s.g.N = int(ins[size_t(i)].s->N);
      s.g.A = int(ins[size_t(i)].s->A);
      s.g.M = ins[size_t(i)].s->M;
      s.g.C = ins[size_t(i)].s->C;
      s.g.n_start = s.g.n_accept = 1;
    } else if (!(s.g.flags & 4)) {
      eps_free = false;
    }
    s.node_off = int(N);
    if (concat && i > 0) {  // the connectors into graph i come right after graph i's own arcs (functions.cpp:139-149)
      s.arc_off = int(A);
      s.conn_off = int(A) + s.g.A;
    } else {
      s.arc_off = int(A);
      s.conn_off = int(A) + s.g.A;
    }
    offs.push_back(A);
    int conn = 0;
    if (concat && i > 0) conn = segs[size_t(i) - 1].g.n_accept * s.g.n_start;
    if (closure) conn = s.g.n_start + s.g.n_accept;
    if (conn) eps_free = false;
    s.keep_start = closure ? 0 : (concat ? i == 0 : 1);
    s.keep_accept = closure ? 0 : (concat ? i == k - 1 : 1);
    N += s.g.N;
    A += int64_t(s.g.A) + conn;
    max_A = std::max(max_A, s.g.A);
    max_N = std::max(max_N, s.g.N);
    max_conn = std::max(max_conn, conn);
  }
  if (N > (int64_t(1) << 30) || A > (int64_t(1) << 30)) throw_runtime("[gtn] rational operation: result too large");
  // one arena: arc arrays, weights, flags, lists, adjacency
  size_t bytes = 0;
  auto add = [&](size_t b) {
    const size_t at = bytes;
    bytes = align_up(bytes + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(N);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_w = add(a4), o_fl = add(size_t(N)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(bytes);
  RationalOut ro{};
  ro.N = int(N);
  ro.A = int(A);
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_w);
  ro.nflags = arena->as<uint8_t>(o_fl);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  DevMemP dsegs = upload_vec(segs);
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(int(N), int(A)));
  launch_rational_build(dsegs->as<RationalSeg>(), k, max_A, max_N, max_conn, ro, projection, closure ? 1 : 0, temp->ptr, rt.stream());
  // counts of the output's start / accept nodes follow from the inputs'
  int n_start = 0, n_accept = 0;
  if (closure) n_start = n_accept = 1;
  else if (concat) n_start = segs.front().g.n_start, n_accept = segs.back().g.n_accept;
  else
    for (auto& s : segs) n_start += s.g.n_start, n_accept += s.g.n_accept;
  Graph out = make_output(op, 0, ins);
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = N;
  st.A = A;
  st.host_valid = false;
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = int(N);
  v.A = int(A);
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  op->arc_off.push_back(std::move(offs));
  return out;
}

// ======================================================================
// remove (functions.cpp:253-318), built on the device: rational.hip
// ======================================================================
namespace {
struct RemoveOp : OpRecord {
  void backward(std::vector<Member>&) override {
    throw_logic("[gtn::remove] gradient compuation not implemented");  // functions.cpp:271-273
  }
};
}  // namespace

Graph op_remove(Graph& gin, int ilabel, int olabel) {
  Runtime& rt = Runtime::get();
  auto op = std::make_shared<RemoveOp>();
  op->seq = next_seq();
  std::vector<Graph> ins{gin};
  Graph out = make_output(op, 0, ins);
  gin.s->resolve_sizes();
  // the structure the walk reads: explicit, with out-lists on the device (an implicit chain is written out into a
  // copy; a composition result that keeps its out-lists implicit goes through its host mirror once)
  Graph src = gin;
  if (src.s->kind == KIND_LINEAR) src = Graph::deep_copy(gin);
  std::vector<Structure*> ss{src.s.get()};
  ensure_device_batch(ss);
  DGraph g = device_view(src);
  if (g.A > 0 && (!g.out_off || !g.out_list)) {
    src = Graph::deep_copy(src);  // (host arrays, uploaded with their adjacency lists)
    ss[0] = src.s.get();
    ensure_device_batch(ss);
    g = device_view(src);
  }
  const int N = g.N;
  if (N == 0) return out;
  const size_t scan_b = scan_temp_bytes(N + 1);
  // keep flags -> ids of the kept nodes
  DevMemP ids = rt.alloc(4 * (3 * size_t(N) + 8) + scan_b);
  int* keep = ids->as<int>();
  int* new_id = keep + (N + 1);
  int* roots = new_id + (N + 1);
  void* scan_tmp = roots + N + 2;
  HIP_CHECK(hipMemsetAsync(keep, 0, 4 * size_t(N + 1), rt.stream()));
  launch_remove_keep(g, ilabel, olabel, keep, rt.stream());
  launch_exclusive_scan(keep, new_id, N + 1, scan_tmp, scan_b, rt.stream());
  int K = 0;
  rt.d2h_sync(&K, new_id + N, sizeof(int));
  if (K == 0) return out;
  launch_remove_roots(keep, new_id, N, roots, rt.stream());
  // walks: batches of `rows` kept nodes share 2 x rows x N ints of scratch (at most ~256 MB)
  const int rows = int(std::max<int64_t>(1, std::min<int64_t>(K, (int64_t(32) << 20) / std::max(N, 1))));
  DevMemP scratch = rt.alloc_zero(8 * size_t(rows) * size_t(N));
  DevMemP counts = rt.alloc(4 * (2 * size_t(K) + 4) + scan_temp_bytes(K + 1));
  RemoveArgs ra{};
  ra.g = g;
  ra.ilabel = ilabel;
  ra.olabel = olabel;
  ra.new_id = new_id;
  ra.roots = roots;
  ra.K = K;
  ra.rows = rows;
  ra.stamp = scratch->as<int>();
  ra.queue = ra.stamp + size_t(rows) * size_t(N);
  ra.arc_cnt = counts->as<int>();
  int* arc_off = ra.arc_cnt + (K + 1);
  HIP_CHECK(hipMemsetAsync(ra.arc_cnt, 0, 4 * size_t(K + 1), rt.stream()));
  for (int r0 = 0; r0 < K; r0 += rows) {
    ra.root0 = r0;
    launch_remove_walk(ra, false, rt.stream());
  }
  launch_exclusive_scan(ra.arc_cnt, arc_off, K + 1, arc_off + K + 2, scan_temp_bytes(K + 1), rt.stream());
  int A = 0;
  rt.d2h_sync(&A, arc_off + K, sizeof(int));
  // the result's arena (as op_rational lays it out)
  size_t total = 0;
  auto add = [&](size_t b) {
    const size_t at = total;
    total = align_up(total + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(K);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_w = add(a4), o_nf = add(size_t(K)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(total);
  RationalOut ro{};
  ro.N = K;
  ro.A = A;
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_w);
  ro.nflags = arena->as<uint8_t>(o_nf);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  ra.out = ro;
  ra.arc_off = arc_off;
  // (the stamps of the count pass are k + 1: the emit pass uses K + k + 1 through a shifted tag base)
  HIP_CHECK(hipMemsetAsync(ra.stamp, 0, 4 * size_t(rows) * size_t(N), rt.stream()));
  for (int r0 = 0; r0 < K; r0 += rows) {
    ra.root0 = r0;
    launch_remove_walk(ra, true, rt.stream());
  }
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(K, A));
  launch_rational_adjacency(ro, temp->ptr, rt.stream());
  // start / accept counts for the view (the ordered lists are built by the adjacency pass)
  std::vector<uint8_t> fl(static_cast<size_t>(K));
  rt.d2h_sync(fl.data(), ro.nflags, size_t(K));
  int n_start = 0, n_accept = 0;
  bool eps_free = false;  // (not known without a pass over the labels: claim nothing)
  for (uint8_t f : fl) n_start += (f & NF_START) ? 1 : 0, n_accept += (f & NF_ACCEPT) ? 1 : 0;
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = K;
  st.A = A;
  st.host_valid = false;
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = K;
  v.A = A;
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  return out;
}

// ======================================================================
// binary graph format (utils.cpp:152-225) straight into device buffers
// ======================================================================
Graph op_load_buffer(const void* data, size_t bytes) {
  // layout: int32 {N, A, n_start, n_accept} | start[n_start] | accept[n_accept] | {src, dst, ilabel, olabel} x A | float w[A]
  const char* p = static_cast<const char*>(data);
  if (bytes < 16) throw_invalid("[gtn::load] truncated graph file");
  int head[4];
  std::memcpy(head, p, 16);
  const int64_t N = head[0], A = head[1], ns = head[2], na = head[3];
  if (N < 0 || A < 0 || ns < 0 || na < 0 || ns > N || na > N) throw_invalid("[gtn::load] corrupt graph file header");
  const size_t need = 16 + 4 * size_t(ns + na) + 20 * size_t(A);
  if (bytes < need) throw_invalid("[gtn::load] truncated graph file");
  const int* start = reinterpret_cast<const int*>(p + 16);
  const int* accept = start + ns;
  const int* rows = accept + na;
  const float* w = reinterpret_cast<const float*>(rows + 4 * A);
  Graph out(true);
  if (N == 0) return out;
  // A small graph -- or a host without a GPU: building a graph is host work everywhere in this engine, only the
  // graph FUNCTIONS need the device -- is put together like addNode / addArc would, in two bulk appends; large
  // decoding graphs take the device route below.
  if (A < 4096 || Runtime::device_count() == 0) {
    std::vector<uint8_t> st(static_cast<size_t>(N), 0), ac(static_cast<size_t>(N), 0);
    for (int64_t i = 0; i < ns; ++i) {
      if (start[i] < 0 || start[i] >= N) throw_range("[gtn::load] start node out of range");
      st[size_t(start[i])] = 1;
    }
    for (int64_t i = 0; i < na; ++i) {
      if (accept[i] < 0 || accept[i] >= N) throw_range("[gtn::load] accept node out of range");
      ac[size_t(accept[i])] = 1;
    }
    out.add_nodes(int(N), st.data(), ac.data());
    std::vector<int> src(static_cast<size_t>(A)), dst(src.size()), il(src.size()), ol(src.size());
    for (int64_t a = 0; a < A; ++a) {
      src[size_t(a)] = rows[4 * a];
      dst[size_t(a)] = rows[4 * a + 1];
      il[size_t(a)] = rows[4 * a + 2];
      ol[size_t(a)] = rows[4 * a + 3];
    }
    std::vector<float> ww(w, w + A);  // (the file image need not be aligned for floats in place)
    out.add_arcs(int(A), src.data(), dst.data(), il.data(), ol.data(), ww.data());
    return out;
  }
  Runtime& rt = Runtime::get();
  // host: node flags from the two lists; the arc table is checked like addArc checks it (graph.cpp:47-66)
  std::vector<uint8_t> flags(static_cast<size_t>(N), 0);
  for (int64_t i = 0; i < ns; ++i) {
    if (start[i] < 0 || start[i] >= N) throw_range("[gtn::load] start node out of range");
    flags[size_t(start[i])] |= NF_START;
  }
  for (int64_t i = 0; i < na; ++i) {
    if (accept[i] < 0 || accept[i] >= N) throw_range("[gtn::load] accept node out of range");
    flags[size_t(accept[i])] |= NF_ACCEPT;
  }
  bool eps_free = true;
  for (int64_t a = 0; a < A; ++a) {
    const int* r = rows + 4 * a;
    if (r[0] < 0 || r[0] >= N || r[1] < 0 || r[1] >= N) throw_range("[Graph::addArc] node index out of range");
    if (r[2] < GTNX_EPSILON || r[3] < GTNX_EPSILON) throw_invalid("[Graph::addArc] labels must be >= epsilon");
    eps_free = eps_free && r[2] >= 0 && r[3] >= 0;
  }
  // one staging copy: rows (16-byte aligned) | weights | flags
  const size_t o_rows = 0, o_w = align_up(16 * size_t(A), 256), o_fl = align_up(o_w + 4 * size_t(A), 256);
  const size_t in_bytes = align_up(o_fl + size_t(N), 256);
  PinnedMemP pin = rt.alloc_pinned(in_bytes);
  std::memcpy(pin->as<char>(o_rows), rows, 16 * size_t(A));
  std::memcpy(pin->as<char>(o_w), w, 4 * size_t(A));
  std::memcpy(pin->as<char>(o_fl), flags.data(), size_t(N));
  DevMemP raw = rt.alloc(in_bytes);
  rt.h2d(raw->ptr, pin->ptr, in_bytes);
  // the structure's own arena: arc arrays, weights, flags, lists, adjacency (as op_rational lays it out)
  size_t total = 0;
  auto add = [&](size_t b) {
    const size_t at = total;
    total = align_up(total + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(N);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_ww = add(a4), o_nf = add(size_t(N)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(total);
  RationalOut ro{};
  ro.N = int(N);
  ro.A = int(A);
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_ww);
  ro.nflags = arena->as<uint8_t>(o_nf);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(int(N), int(A)));
  launch_rational_load(raw->as<char>(o_rows), raw->as<float>(o_w), raw->as<uint8_t>(o_fl), ro, temp->ptr, rt.stream());
  int n_start = 0, n_accept = 0;  // (a node listed twice counts once: addNode's lists have one entry per node)
  for (uint8_t f : flags) n_start += (f & NF_START) ? 1 : 0, n_accept += (f & NF_ACCEPT) ? 1 : 0;
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = N;
  st.A = A;
  st.host_valid = false;  // the host mirror is pulled when somebody inspects the graph
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = int(N);
  v.A = int(A);
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  return out;
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
