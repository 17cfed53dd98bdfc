// region.cpp -- see region.h.  Replaces the reference's per-utterance execution inside
// gtn::parallelMap (gtn/parallel/parallel_map.h:153-188, benchmarks/ctc.cpp:150-165) with deferred calls
// that the region's join runs as batch records (batch.h).
#include "region.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace gtnx {

namespace {

// pinned staging of the region's host-source setWeights calls: ONE block for all threads, bump-allocated, so
// that the join moves the region's weights to the device with one copy (the device arena is its image)
struct StageArena {
  PinnedMemP mem;
  std::atomic<size_t> used{0};
  size_t cap = 0;
};

thread_local int t_depth = 0;  // nesting of gtnx_parallel_enter on this thread
thread_local int t_exec = 0;   // > 0: this thread is running queued calls (its own graph functions run eagerly)
thread_local std::vector<std::shared_ptr<Pending>> t_queue;
thread_local std::vector<std::shared_ptr<Weights>> t_stage;
thread_local std::vector<Graph*> t_trash;
thread_local bool t_vector_call = false;  // recording the calls of a gtnx_*_n vector form (region_run_vector)
thread_local std::shared_ptr<StageArena> t_arena;

struct Shared {
  std::mutex mu;  // what the region's threads handed in
  std::vector<std::shared_ptr<Pending>> queue;
  std::vector<std::shared_ptr<Weights>> stage;
  std::exception_ptr first_error;  // of a run nobody was there to catch (a forced closure's other members)
  std::shared_ptr<StageArena> arena;  // the current staging block (replaced when full and at every join)
  size_t arena_hint = size_t(64) << 20;  // bytes the next block starts with (what the last region needed)
  std::atomic<int> draining{0};       // threads taking the runtime's deferred list apart right now
  std::recursive_mutex exec;  // one runner at a time (recursive: a user gradFunc may call back into the engine)
};
Shared& shared() {
  static Shared* s = new Shared();  // never destroyed: worker threads may outlive static destruction
  return *s;
}

struct ExecScope {
  ExecScope() { ++t_exec; }
  ~ExecScope() { --t_exec; }
};

inline bool is_placeholder(const Graph& g) { return g.s && g.s->pending; }

void count_use(const Graph& g, int d) {
  if (!g.s || is_placeholder(g)) return;
  g.s->pending_uses += d;
  if (g.w) g.w->pending_uses += d;
}

// a call's value as the batch functions see it
struct Val {
  Batch* batch = nullptr;     // element `idx` of this record (kept alive by its call), or
  const BatchP* batch_p = nullptr;
  int idx = -1;
  const Graph* g = nullptr;   // an ordinary graph (the call's own input, or an earlier call's result)
  std::exception_ptr err;
};

// inputs of calls that have run: let go of in one piece, off the joining thread (a placeholder whose handle
// is gone already dies with its last reference -- that is here)
thread_local std::vector<Graph>* t_released = nullptr;

void finish(Pending& p, int state) {
  count_use(p.a, -1);
  count_use(p.b, -1);
  if (t_released) {
    if (p.a.s) t_released->push_back(std::move(p.a));
    if (p.b.s) t_released->push_back(std::move(p.b));
  }
  p.a = Graph(Graph::Empty{});
  p.b = Graph(Graph::Empty{});
  p.state.store(state, std::memory_order_release);
}

void fail(Pending& p, std::exception_ptr e) {
  p.err = e;
  finish(p, 2);
}

void set_result(Pending& p, const BatchP& r, int idx) {
  if (r->kind == Batch::GRAPHS) {  // the elements are graphs already
    p.res = r->graphs[size_t(idx)];
    p.has_res.store(true, std::memory_order_release);
  } else {
    p.batch = r;
    p.idx = idx;
  }
  finish(p, 1);
}

Graph& result_graph(Pending& p) {
  if (!p.has_res.load(std::memory_order_acquire)) {
    std::lock_guard<std::recursive_mutex> lk(shared().exec);
    if (!p.has_res.load(std::memory_order_acquire)) {
      ExecScope es;
      p.res = batch_get(p.batch, p.idx);
      p.has_res.store(true, std::memory_order_release);
    }
  }
  return p.res;
}

// ---- weights handed over inside the region: one copy per staging block / one launch for device sources
void apply_stage(std::vector<std::shared_ptr<Weights>>& stage) {
  if (stage.empty()) return;
  GTNX_HOST_T("region.apply_stage");
  Runtime& rt = Runtime::get();
  // host sources: the device arena is the image of the used span of the pinned block
  struct Span {
    const char* lo;
    const char* hi;
    DevMemP dev;
  };
  std::unordered_map<PinnedMem*, Span> spans;
  std::vector<Weights*> dsrc;
  size_t dtotal = 0;
  for (auto& w : stage) {
    if (!w->staged) continue;  // read (and settled) in the meantime, or overwritten
    if (w->staged->on_device) {
      dsrc.push_back(w.get());
      dtotal += align_up(sizeof(float) * size_t(w->n), 16);
      continue;
    }
    const char* src = reinterpret_cast<const char*>(w->staged->src);
    const char* end = src + align_up(sizeof(float) * size_t(w->n), 16);
    auto it = spans.find(w->staged->chunk.get());
    if (it == spans.end())
      spans.emplace(w->staged->chunk.get(), Span{src, end, nullptr});
    else {
      it->second.lo = std::min(it->second.lo, src);
      it->second.hi = std::max(it->second.hi, end);
    }
  }
  for (auto& kv : spans) {
    Span& sp = kv.second;
    sp.dev = rt.alloc(size_t(sp.hi - sp.lo));
    rt.h2d(sp.dev->ptr, sp.lo, size_t(sp.hi - sp.lo));
  }
  DevMemP darena;
  if (!dsrc.empty()) {
    darena = rt.alloc(dtotal ? dtotal : 16);
    std::vector<CopySeg> segs;
    segs.reserve(dsrc.size());
    int64_t max_bytes = 0;
    size_t off = 0;
    for (Weights* w : dsrc) {
      const size_t bytes = sizeof(float) * size_t(w->n);
      segs.push_back({darena->as<char>(off), w->staged->src, int64_t(bytes)});
      max_bytes = std::max<int64_t>(max_bytes, int64_t(bytes));
      w->dev_mem = darena;
      w->dev = darena->as<float>(off);
      off += align_up(bytes, 16);
    }
    DevMemP d = upload_vec(segs);
    launch_copy_segments(d->as<CopySeg>(), int(segs.size()), max_bytes, rt.stream());
  }
  for (auto& w : stage) {
    if (!w->staged) continue;
    if (!w->staged->on_device) {
      const Span& sp = spans[w->staged->chunk.get()];
      w->dev_mem = sp.dev;
      w->dev = sp.dev->as<float>(size_t(reinterpret_cast<const char*>(w->staged->src) - sp.lo));
    }
    w->dev_valid = true;
    w->host_valid = false;
    w->staged.reset();  // (the pinned block goes back to the pool behind the copy: stream order)
  }
  stage.clear();
}

// ---- one run --------------------------------------------------------------------------------------------
struct Group {
  RegionOp op;
  int depth;
  bool postponed = false;  // forwardScore of plain linear chains: after the sweeps over the same chains (they
                           // leave it behind, batch.cpp: nc_norm), unless somebody needs it earlier
  bool ran = false;
  std::vector<Pending*> calls;
};

struct Run {
  std::vector<std::shared_ptr<Pending>>& calls;
  std::vector<Group> groups;
  std::unordered_map<Weights*, BatchP> linear_of;      // first element's weights -> leaf LINEAR record
  std::unordered_map<Structure*, BatchP> targets_of;   // first element's structure -> leaf CTC_TARGETS record
  std::exception_ptr first_error;

  explicit Run(std::vector<std::shared_ptr<Pending>>& c) : calls(c) {}

  void note_error(std::exception_ptr e) {
    if (!first_error) first_error = e;
  }

  Val value_of(Graph& x) {
    Val v;
    if (!is_placeholder(x)) {
      v.g = &x;
      return v;
    }
    std::shared_ptr<Pending> q(x.s, x.s->pending);
    if (q->state.load(std::memory_order_acquire) == 0) {
      if (q->group >= 0) {
        run_group(groups[size_t(q->group)]);
      } else {  // queued by a thread that has not handed its calls in: run that one now
        std::vector<std::shared_ptr<Pending>> one{q};
        Run sub(one);
        sub.run_all();
      }
    }
    if (q->state.load(std::memory_order_acquire) == 2) {
      v.err = q->err;
      return v;
    }
    if (q->has_res.load(std::memory_order_acquire)) {
      v.g = &q->res;
    } else {
      v.batch = q->batch.get();
      v.batch_p = &q->batch;
      v.idx = q->idx;
    }
    return v;
  }

  static Graph graph_of(Val& v) {
    if (v.g) return *v.g;
    return batch_get(*v.batch_p, v.idx);
  }

  // all values are the elements of ONE record, each exactly once: that record and the element of each call
  static BatchP aligned(std::vector<Val>& vs, std::vector<int>& perm) {
    if (vs.empty() || !vs[0].batch) return nullptr;
    Batch* x = vs[0].batch;
    if (size_t(x->n) != vs.size()) return nullptr;
    std::vector<uint8_t> seen(vs.size(), 0);
    perm.resize(vs.size());
    for (size_t k = 0; k < vs.size(); ++k) {
      if (vs[k].batch != x || vs[k].idx < 0 || vs[k].idx >= x->n || seen[size_t(vs[k].idx)]) return nullptr;
      seen[size_t(vs[k].idx)] = 1;
      perm[k] = vs[k].idx;
    }
    return *vs[0].batch_p;
  }

  // the inputs of a group as ONE batch record + the element each call reads
  BatchP as_batch(std::vector<Val>& vs, std::vector<int>& perm, bool want_targets) {
    if (BatchP x = aligned(vs, perm)) return x;
    perm.resize(vs.size());
    for (size_t k = 0; k < vs.size(); ++k) perm[k] = int(k);
    bool all_graphs = true;
    for (auto& v : vs) all_graphs = all_graphs && v.g;
    std::vector<Graph> gs;
    gs.reserve(vs.size());
    for (auto& v : vs) gs.push_back(graph_of(v));
    if (all_graphs && vs.size() >= 2) {
      const Structure& s0 = *gs[0].s;
      if (s0.kind == KIND_LINEAR && !s0.lazy) {
        auto it = linear_of.find(gs[0].w.get());
        if (it != linear_of.end() && same_leaves(*it->second, gs)) return it->second;
        if (BatchP old = gs[0].w->leaf_batch.lock())  // made by an earlier call over the same graphs, unchanged since
          if (old->kind == Batch::LINEAR && same_leaves(*old, gs)) {
            linear_of[gs[0].w.get()] = old;
            return old;
          }
        if (BatchP b = batch_linear_from_graphs(gs)) {
          linear_of[gs[0].w.get()] = b;
          return b;
        }
      } else if (want_targets && s0.kind == KIND_EXPLICIT && s0.host_valid && !s0.lazy && s0.N <= band_max_nodes()) {
        auto it = targets_of.find(gs[0].s.get());
        if (it != targets_of.end() && same_leaves(*it->second, gs)) return it->second;
        if (BatchP old = gs[0].s->leaf_batch.lock())
          if (old->kind == Batch::CTC_TARGETS && same_leaves(*old, gs)) {
            targets_of[gs[0].s.get()] = old;
            return old;
          }
        if (BatchP b = batch_ctc_targets_from_graphs(gs)) {
          targets_of[gs[0].s.get()] = b;
          return b;
        }
      }
    }
    return batch_from_graphs(std::move(gs));
  }

  static bool same_leaves(const Batch& b, const std::vector<Graph>& gs) {
    if (size_t(b.n) != gs.size() || b.graphs.size() != gs.size()) return false;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (b.graphs[i].s != gs[i].s || b.graphs[i].w != gs[i].w || b.graphs[i].g != gs[i].g) return false;
      if (gs[i].calc_grad() != b.calc_grad) return false;
      // (a LINEAR record holds the weights' values: they must not have changed since; a target record holds
      //  labels only -- touch() forgets it when the structure changes -- and needs all-zero weights)
      if (b.kind == Batch::LINEAR && gs[i].w->leaf_version != gs[i].w->version) return false;
      if (b.kind == Batch::CTC_TARGETS && !gs[i].w->is_all_zero()) return false;
    }
    return true;
  }

  static bool binary(RegionOp op) { return op == RO_ADD || op == RO_SUB || op == RO_COMPOSE || op == RO_INTERSECT; }

  // the batch function of a group over whole records
  BatchP apply(RegionOp op, const BatchP& a, const BatchP& b, int mode) {
    switch (op) {
      case RO_NEG: return batch_scalar(SK_NEGATE, a, nullptr);
      case RO_ADD: return batch_scalar(SK_ADD, a, b);
      case RO_SUB: return batch_scalar(SK_SUBTRACT, a, b);
      case RO_COMPOSE:
      case RO_INTERSECT: {
        // the lattices of such a loop are looked at by forwardScore only: kept symbolic where the sweep kernels
        // apply (gtnx_compose_mode 2; looking inside one still builds it)
        struct Mode {
          int old;
          explicit Mode(int m) : old(compose_mode_hint(m)) {}
          ~Mode() { compose_mode_hint(old); }
        } scope(mode);
        return batch_compose(a, b, op == RO_INTERSECT);
      }
      case RO_FS: return batch_shortest_distance(a, false);
      case RO_VS: return batch_shortest_distance(a, true);
      case RO_VP: return batch_viterbi_path(a);
      default: return nullptr;
    }
  }

  void run_calls(RegionOp op, std::vector<Pending*>& cs, bool retry_singly) {
    const size_t n = cs.size();
    std::vector<Pending*> live;
    std::vector<Val> la, lb;
    live.reserve(n);
    la.reserve(n);
    if (binary(op)) lb.reserve(n);
    {
      GTNX_HOST_T("region.run_calls.resolve");
      for (size_t k = 0; k < n; ++k) {
        // (the records were written by the pool's threads: every one of them is a miss in this core's caches --
        //  fetch the record eight calls ahead and what its inputs point to four calls ahead)
        if (k + 8 < n) __builtin_prefetch(cs[k + 8], 1);
        if (k + 4 < n) {
          __builtin_prefetch(cs[k + 4]->a.s.get());
          if (binary(op)) __builtin_prefetch(cs[k + 4]->b.s.get());
        }
        Pending& p = *cs[k];
        Val a = value_of(p.a), b;
        if (binary(op)) b = value_of(p.b);
        if (a.err || b.err) {  // an input failed: so does this call, with the same error
          fail(p, a.err ? a.err : b.err);
          continue;
        }
        live.push_back(&p);
        la.push_back(std::move(a));
        if (binary(op)) lb.push_back(std::move(b));
      }
    }
    if (live.empty()) return;
    try {
      if (op == RO_BWD || op == RO_BWD_RETAIN) {
        run_backward(live, la, op == RO_BWD_RETAIN);
        return;
      }
      std::vector<int> pa, pb;
      // compose / intersect: the mode the calls were made under (-1, the engine's own policy -- gtn_amd.h:
      // symbolic for small partners built on the host -- unless the caller set one); target records (always
      // symbolic) only when the mode allows symbolic results
      int mode = 0;
      const bool comp = op == RO_COMPOSE || op == RO_INTERSECT;
      if (comp) {
        mode = live[0]->mode;
        const char* env = std::getenv("GTNX_LAZY_COMPOSE");  // (the process-wide override, read per call: ops.cpp)
        if (env && env[0] >= '0' && env[0] <= '2') mode = env[0] - '0';
      }
      const bool records = comp && mode != 0 && !std::getenv("GTNX_NO_BAND");
      BatchP A = as_batch(la, pa, records), B;
      if (binary(op)) {
        B = as_batch(lb, pb, records);
        if (pa != pb) {  // the two sides are elements of records in different orders: line them up as graphs
          std::vector<Graph> ga, gb;
          for (auto& v : la) ga.push_back(graph_of(v));
          for (auto& v : lb) gb.push_back(graph_of(v));
          A = batch_from_graphs(std::move(ga));
          B = batch_from_graphs(std::move(gb));
          for (size_t k = 0; k < pa.size(); ++k) pa[k] = int(k);
        }
      }
      BatchP R = apply(op, A, B, mode);
      GTNX_HOST_T("region.run_calls.results");
      for (size_t k = 0; k < live.size(); ++k) {
        if (k + 8 < live.size()) __builtin_prefetch(live[k + 8], 1);
        set_result(*live[k], R, pa[k]);
      }
    } catch (...) {
      if (!retry_singly || live.size() == 1) {
        const std::exception_ptr e = std::current_exception();
        note_error(e);
        for (Pending* p : live)
          if (p->state.load(std::memory_order_acquire) == 0) fail(*p, e);
        return;
      }
      // one by one (these functions do not change their inputs): every call gets its own result or error
      for (size_t k = 0; k < live.size(); ++k) {
        if (live[k]->state.load(std::memory_order_acquire) != 0) continue;
        std::vector<Pending*> one{live[k]};
        run_calls(op, one, false);
      }
    }
  }

  void run_backward(std::vector<Pending*>& live, std::vector<Val>& roots, bool retain) {
    std::vector<int> perm;
    if (BatchP x = aligned(roots, perm)) {
      batch_backward(x, retain);  // (throws before it changes anything: batch.cpp)
      for (Pending* p : live) finish(*p, 1);
      return;
    }
    // per-graph tapes: a root that cannot run (backward twice without retain) fails alone, the others go
    // through ONE sweep -- which either happens for all of them or, if it throws midway, is reported to all
    std::vector<Graph> gs;
    std::vector<Pending*> ok;
    for (size_t k = 0; k < live.size(); ++k) {
      try {
        Graph g = graph_of(roots[k]);
        realize(g);
        backward_validate(g);
        gs.push_back(std::move(g));
        ok.push_back(live[k]);
      } catch (...) {
        note_error(std::current_exception());
        fail(*live[k], std::current_exception());
      }
    }
    if (gs.empty()) return;
    try {
      op_backward(gs, nullptr, retain);
      for (Pending* p : ok) finish(*p, 1);
    } catch (...) {
      note_error(std::current_exception());
      for (Pending* p : ok) fail(*p, std::current_exception());
    }
  }

  void run_group(Group& g) {
    if (g.ran) return;
    g.ran = true;
    static const char* names[RO_COUNT] = {"region.negate", "region.add", "region.subtract", "region.compose", "region.intersect",
                                          "region.forward_score", "region.viterbi_score", "region.viterbi_path",
                                          "region.backward", "region.backward_retain"};
    GTNX_HOST_T(names[g.op]);
    run_calls(g.op, g.calls, true);
  }

  void run_all() {
    // group by (function, depth); backward calls last, in one group per retain flag
    std::unordered_map<uint64_t, int> index;
    for (auto& sp : calls) {
      Pending& p = *sp;
      if (p.state.load(std::memory_order_acquire) != 0) continue;  // ran already (somebody looked at it)
      const bool bwd = p.op == RO_BWD || p.op == RO_BWD_RETAIN;
      const uint64_t key = (uint64_t(bwd ? 0xffffff : uint32_t(p.depth)) << 16) | (uint64_t(uint8_t(p.mode)) << 8) | uint64_t(p.op);
      auto it = index.find(key);
      if (it == index.end()) {
        it = index.emplace(key, int(groups.size())).first;
        groups.push_back(Group{p.op, bwd ? 0x7fffffff : p.depth});
      }
      p.group = it->second;
      groups[size_t(it->second)].calls.push_back(&p);
    }
    for (auto& g : groups) {
      if (g.op != RO_FS) continue;
      g.postponed = true;
      for (Pending* p : g.calls)
        if (is_placeholder(p->a) || p->a.s->kind != KIND_LINEAR) {
          g.postponed = false;
          break;
        }
    }
    std::vector<int> order(groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = int(i);
    auto prio = [](RegionOp op) { return (op == RO_COMPOSE || op == RO_INTERSECT) ? 0 : 1; };
    std::sort(order.begin(), order.end(), [&](int x, int y) {
      const Group &a = groups[size_t(x)], &b = groups[size_t(y)];
      const bool ba = a.depth == 0x7fffffff, bb = b.depth == 0x7fffffff;
      if (ba != bb) return bb;
      if (a.postponed != b.postponed) return b.postponed;
      if (a.depth != b.depth) return a.depth < b.depth;
      if (prio(a.op) != prio(b.op)) return prio(a.op) < prio(b.op);
      return int(a.op) < int(b.op);
    });
    for (int gi : order) run_group(groups[size_t(gi)]);
    for (auto& sp : calls) sp->group = -1;
  }
};

// runs `calls` (and what they depend on); returns the first error of the run
std::exception_ptr execute(std::vector<std::shared_ptr<Pending>>& calls, std::vector<std::shared_ptr<Weights>>& stage) {
  std::lock_guard<std::recursive_mutex> lk(shared().exec);
  GTNX_HOST_T("region.execute");
  ExecScope es;
  std::exception_ptr err;
  std::vector<Graph>* outer = t_released;
  auto* released = new std::vector<Graph>();
  released->reserve(2 * calls.size());
  t_released = released;
  try {
    apply_stage(stage);
    Run run(calls);
    run.run_all();
    err = run.first_error;
  } catch (...) {
    err = std::current_exception();
    for (auto& sp : calls)
      if (sp->state.load(std::memory_order_acquire) == 0) fail(*sp, err);
  }
  t_released = outer;
  // the calls themselves and what they held: taken apart by the pool's threads (gtnx_reclaim), not here
  auto* dead = new std::vector<std::shared_ptr<Pending>>();
  dead->swap(calls);
  if (Runtime::initialized()) {
    Runtime& rt = Runtime::get();
    rt.defer_delete(released, [](void* q) { delete static_cast<std::vector<Graph>*>(q); });
    rt.defer_delete(dead, [](void* q) { delete static_cast<std::vector<std::shared_ptr<Pending>>*>(q); });
  } else {
    delete released;
    delete dead;
  }
  return err;
}

void take_shared(std::vector<std::shared_ptr<Pending>>& q, std::vector<std::shared_ptr<Weights>>& st) {
  Shared& sh = shared();
  std::lock_guard<std::mutex> lk(sh.mu);
  q.insert(q.end(), sh.queue.begin(), sh.queue.end());
  st.insert(st.end(), sh.stage.begin(), sh.stage.end());
  sh.queue.clear();
  sh.stage.clear();
  if (sh.arena) {  // the next region starts a fresh block, sized by what this one used
    sh.arena_hint = std::max<size_t>(size_t(16) << 20, std::min(sh.arena_hint, 2 * sh.arena->used.load()));
    sh.arena.reset();
  }
}

}  // namespace

bool region_active() { return t_depth > 0 && t_exec == 0; }

namespace {
std::atomic<int64_t> g_first_enter_us{0};  // GTNX_HOST_TIMING: when the current region's first thread arrived
int64_t now_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

void region_enter() {
  if (HostTimer::enabled() && t_depth == 0) {
    int64_t zero = 0;
    g_first_enter_us.compare_exchange_strong(zero, now_us());
  }
  // what earlier steps let go of is taken apart by the region's threads together (a batch's graphs were built
  // by such threads too), not by the one thread that joins them
  // (a handful of them: the allocator's locks are what more threads would wait on)
  if (t_depth++ == 0 && Runtime::initialized()) {
    Shared& sh = shared();
    static const int max_drainers = [] {
      const char* e = std::getenv("GTNX_DRAIN_THREADS");
      return e ? std::atoi(e) : 0;
    }();
    if (sh.draining.fetch_add(1) < max_drainers) Runtime::get().drain_deferred();
    sh.draining.fetch_sub(1);
  }
}

void region_leave() {
  if (t_depth <= 0 || --t_depth > 0) return;
  if (!t_queue.empty() || !t_stage.empty()) {
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.queue.insert(sh.queue.end(), std::make_move_iterator(t_queue.begin()), std::make_move_iterator(t_queue.end()));
    sh.stage.insert(sh.stage.end(), std::make_move_iterator(t_stage.begin()), std::make_move_iterator(t_stage.end()));
    t_queue.clear();
    t_stage.clear();
  }
  t_arena.reset();
  if (!t_trash.empty()) {
    if (Runtime::initialized()) {
      auto* dead = new std::vector<Graph*>();
      dead->swap(t_trash);
      Runtime::get().defer_delete(dead, [](void* q) {
        auto* v = static_cast<std::vector<Graph*>*>(q);
        for (Graph* g : *v) delete g;
        delete v;
      });
    } else {
      for (Graph* g : t_trash) delete g;
      t_trash.clear();
    }
  }
}

void region_flush() {
  if (HostTimer::enabled() && t_depth == 0) {
    const int64_t t0 = g_first_enter_us.exchange(0);
    if (t0) host_timer_add("region.pool_phase(first enter -> join)", double(now_us() - t0) * 1e-3);
  }
  std::vector<std::shared_ptr<Pending>> q;
  std::vector<std::shared_ptr<Weights>> st;
  std::exception_ptr stored;
  if (t_depth == 0) {
    take_shared(q, st);
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    stored = sh.first_error;
    sh.first_error = nullptr;
  }
  // (inside an enclosing region -- a nested parallelMap -- only the caller's own calls: they are what it joins)
  q.insert(q.end(), std::make_move_iterator(t_queue.begin()), std::make_move_iterator(t_queue.end()));
  st.insert(st.end(), std::make_move_iterator(t_stage.begin()), std::make_move_iterator(t_stage.end()));
  t_queue.clear();
  t_stage.clear();
  std::exception_ptr err;
  if (!q.empty() || !st.empty()) err = execute(q, st);
  if (stored) std::rethrow_exception(stored);
  if (err) std::rethrow_exception(err);
}

namespace {
// a placeholder's structure and its call in one allocation
struct PlaceholderStructure : Structure {
  Pending call;
};
}  // namespace

namespace {
using PlaceholderSlab = std::vector<PlaceholderStructure>;

// the structure + call of one placeholder: its own allocation, or element i of a slab made for a whole vector form
// (one allocation and one release for its n results)
Graph record_into(const std::shared_ptr<PlaceholderSlab>& slab, size_t i, RegionOp op, const Graph& a, const Graph* b) {
  std::shared_ptr<PlaceholderStructure> ps =
      slab ? std::shared_ptr<PlaceholderStructure>(slab, &(*slab)[i]) : std::make_shared<PlaceholderStructure>();
  std::shared_ptr<Pending> p(ps, &ps->call);
  p->op = op;
  p->a = a;
  if (b) p->b = *b;
  int d = 0;
  auto look = [&d](const Graph& x) {
    if (is_placeholder(x)) {
      const Pending& q = *x.s->pending;
      if (q.state.load(std::memory_order_acquire) == 0) d = std::max(d, q.depth);
    } else {
      count_use(x, +1);
    }
  };
  look(a);
  if (b) look(*b);
  p->depth = d + 1;
  if (op == RO_COMPOSE || op == RO_INTERSECT) {
    const int hint = compose_mode_hint(0);
    compose_mode_hint(hint);
    // (inside a parallelMap region only "whenever eligible" is taken from the thread's hint: the lattices of such
    // a loop are looked at by forwardScore only)
    p->mode = t_vector_call ? int8_t(hint) : int8_t(hint == 1 ? 1 : -1);
  }
  t_queue.push_back(p);
  Graph ph{Graph::Empty{}};
  ps->pending = &ps->call;
  ph.s = std::move(ps);
  return ph;
}
}  // namespace

Graph region_record(RegionOp op, const Graph& a, const Graph* b) { return record_into(nullptr, 0, op, a, b); }

namespace {
// the calls of one vector form: recorded on this thread's queue (whatever was there before stays in front of
// them, in order) and joined at once
struct VectorCall {
  int depth0;
  bool exec0;
  VectorCall() : depth0(t_depth), exec0(t_vector_call) {
    ++t_depth;  // (record even when the thread is not inside a parallelMap region)
    t_vector_call = true;
  }
  ~VectorCall() {
    t_vector_call = exec0;
    t_depth = depth0;
  }
};
void join_now() {
  std::vector<std::shared_ptr<Pending>> q;
  std::vector<std::shared_ptr<Weights>> st;
  q.swap(t_queue);
  st.swap(t_stage);
  std::exception_ptr err = execute(q, st);
  if (err) std::rethrow_exception(err);
}
}  // namespace

void region_run_vector(RegionOp op, Graph* const* a, int na, Graph* const* b, int nb, Graph* out) {
  const int n = b ? std::max(na, nb) : na;
  if ((na != n && na != 1) || (b && nb != n && nb != 1))  // parallel_map.h:85-88
    throw_runtime("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
  {
    VectorCall scope;
    auto slab = std::make_shared<PlaceholderSlab>(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) out[i] = record_into(slab, size_t(i), op, *a[na == 1 ? 0 : i], b ? b[nb == 1 ? 0 : i] : nullptr);
  }
  join_now();
}

void region_run_backward_vector(Graph* const* roots, int n, bool retain) {
  {
    VectorCall scope;
    for (int i = 0; i < n; ++i) region_record_backward(*roots[i], retain);
  }
  join_now();
}

void region_record_backward(const Graph& root, bool retain) {
  auto p = std::make_shared<Pending>();
  p->op = retain ? RO_BWD_RETAIN : RO_BWD;
  p->a = root;
  count_use(root, +1);
  t_queue.push_back(std::move(p));
}

namespace {
// somebody looks at a result before the join: run it (and what it needs) now.  The caller's own queue goes
// along -- program order for anything it recorded earlier, and the batch stays a batch when the whole queue is
// one thread's.
void force(const std::shared_ptr<Pending>& p) {
  if (p->state.load(std::memory_order_acquire) == 0) {
    std::vector<std::shared_ptr<Pending>> q;
    std::vector<std::shared_ptr<Weights>> st;
    q.swap(t_queue);
    st.swap(t_stage);
    bool mine = false;
    for (auto& c : q) mine = mine || c == p;
    if (!mine) {
      take_shared(q, st);  // handed in by its thread already?
      bool there = false;
      for (auto& c : q) there = there || c == p;
      if (!there) q.push_back(p);
    }
    std::exception_ptr err = execute(q, st);
    if (err && p->state.load(std::memory_order_acquire) != 2) {
      Shared& sh = shared();
      std::lock_guard<std::mutex> lk(sh.mu);
      if (!sh.first_error) sh.first_error = err;  // reported by the region's join
    }
  }
  if (p->state.load(std::memory_order_acquire) == 2) std::rethrow_exception(p->err);
}
}  // namespace

Graph& region_value(Graph& ph) {
  std::shared_ptr<Pending> p(ph.s, ph.s->pending);
  force(p);
  return result_graph(*p);
}

bool region_item(Graph& ph, float* out) {
  std::shared_ptr<Pending> p(ph.s, ph.s->pending);
  force(p);
  if (p->has_res.load(std::memory_order_acquire)) return false;
  BatchP x = p->batch;
  if (!x || x->kind != Batch::SCALAR || x->materialised) return false;
  std::lock_guard<std::recursive_mutex> lk(shared().exec);
  *out = batch_item_host(x, p->idx);
  return true;
}

bool region_items_device(Graph* const* hs, int n, void* dev_out) {
  if (n <= 0) return true;
  bool any = false;
  for (int i = 0; i < n && !any; ++i) any = is_placeholder(*hs[i]);
  if (!any) return false;
  std::vector<const float*> ptrs(static_cast<size_t>(n));
  Batch* first = nullptr;
  bool dense = true;  // element i of one record at position i: a plain copy
  for (int i = 0; i < n; ++i) {
    Graph& h = *hs[i];
    if (!is_placeholder(h)) return false;
    std::shared_ptr<Pending> p(h.s, h.s->pending);
    force(p);
    if (p->has_res.load(std::memory_order_acquire)) return false;
    Batch* x = p->batch.get();
    if (!x || x->kind != Batch::SCALAR || x->materialised) return false;
    ptrs[size_t(i)] = x->v_dev + p->idx;
    if (i == 0) first = x;
    dense = dense && x == first && p->idx == i;
  }
  Runtime& rt = Runtime::get();
  if (dense) {
    rt.d2d(dev_out, first->v_dev, sizeof(float) * size_t(n));
    return true;
  }
  DevMemP dp = upload_vec(ptrs);
  launch_gather_scalars(dp->as<const float*>(), static_cast<float*>(dev_out), n, rt.stream());
  return true;
}

void region_sync_thread() {
  if (t_queue.empty() && t_stage.empty()) return;
  std::vector<std::shared_ptr<Pending>> q;
  std::vector<std::shared_ptr<Weights>> st;
  q.swap(t_queue);
  st.swap(t_stage);
  std::exception_ptr err = execute(q, st);
  if (err) {
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    if (!sh.first_error) sh.first_error = err;
  }
}

void region_before_mutation(Graph& g) {
  if (!g.s || is_placeholder(g)) return;
  if (g.s->pending_uses.load() == 0 && (!g.w || g.w->pending_uses.load() == 0)) return;
  region_sync_thread();
  if (g.s->pending_uses.load() == 0 && (!g.w || g.w->pending_uses.load() == 0)) return;
  // queued by other threads (a graph shared across the region's tasks)
  std::vector<std::shared_ptr<Pending>> q;
  std::vector<std::shared_ptr<Weights>> st;
  take_shared(q, st);
  std::exception_ptr err = execute(q, st);
  if (err) {
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    if (!sh.first_error) sh.first_error = err;
  }
}

bool region_stage_weights(Graph& g, const float* p, bool device) {
  if (!region_active() || is_placeholder(g)) return false;
  Weights& w = *g.w;
  const int64_t n = g.s->A;
  if (n < 256 || w.host_escaped || g.w.use_count() > 2) return false;  // small, or aliased: the ordinary way
  auto st = std::make_shared<StagedWeights>();
  st->on_device = device;
  if (device) {
    st->src = p;
  } else {
    const size_t bytes = align_up(sizeof(float) * size_t(n), 16);
    float* dst = nullptr;
    for (;;) {
      if (!t_arena) {
        Shared& sh = shared();
        std::lock_guard<std::mutex> lk(sh.mu);
        if (!sh.arena) {
          auto a = std::make_shared<StageArena>();
          a->cap = std::max(sh.arena_hint, bytes);
          a->mem = Runtime::get().alloc_pinned(a->cap);
          a->cap = a->mem->bytes;
          sh.arena = std::move(a);
        }
        t_arena = sh.arena;
      }
      const size_t off = t_arena->used.fetch_add(bytes);
      if (off + bytes <= t_arena->cap) {
        dst = t_arena->mem->as<float>(off);
        break;
      }
      // full: the next block is twice as large (each thread that finds it full asks once)
      Shared& sh = shared();
      std::lock_guard<std::mutex> lk(sh.mu);
      if (sh.arena == t_arena) {
        sh.arena_hint = std::max(sh.arena_hint, 2 * t_arena->cap);
        sh.arena.reset();
      }
      t_arena.reset();
    }
    std::memcpy(dst, p, sizeof(float) * size_t(n));
    st->chunk = t_arena->mem;
    st->src = dst;
  }
  w.n = n;
  w.host.clear();
  w.host_valid = false;
  w.dev_valid = false;
  w.zero = false;
  w.staged = std::move(st);
  w.version++;
  t_stage.push_back(g.w);
  return true;
}

void region_trash(Graph* handle) { t_trash.push_back(handle); }

}  // namespace gtnx
