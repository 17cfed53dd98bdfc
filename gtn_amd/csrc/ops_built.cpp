// ops_built.cpp -- forwardScore / viterbiScore / viterbiPath of BUILT graphs (host-built graphs and materialised products):
// the level-scheduled kernels of shortest.hip and the linear-chain kernels; see ops.h
#include "ops_internal.h"

namespace gtnx {

// ======================================================================
// shortest distance: forwardScore / viterbiScore (functions.cpp:320-326)
// ======================================================================

// effective schedule view for this call: in_w only while it still matches the weights
DSched sched_view(Graph& g, bool need_full = true) {
  if (need_full) g.s->ensure_full();  // in_arc (= in_list) may not have been written yet
  Schedule& sc = *g.s->sched;
  DSched v = sc.view;
  v.in_w = (sc.in_w && sc.in_w_of == g.w.get() && sc.in_w_version == g.w->version) ? sc.in_w : nullptr;
  return v;
}

thread_local ChainGradPlan* t_chain_plan = nullptr;

// the LDS-ring kernels give a node to ONE lane (rows of composed lattices are short: mean in-degree 2.5); a graph
// whose nodes carry tens of arcs each (benchmarks/functions.cpp: makeLinear(1000, 1000)) belongs on the generic
// kernels, which spread a node's row over 8 or 64 lanes.  Sizes still on the device (a FAST chain product: at most
// four arcs per node by construction) count as narrow.
static bool narrow_degree_ok(const Structure& s) { return s.A < 0 || s.N <= 0 || s.A <= 16 * s.N; }
// deep, thin DAGs (a node or two per level, thousands of levels): one wave per graph with the score vector in
// LDS (shortest.hip: sd_*_deep_kernel) instead of a workgroup barrier and four global round trips per level
static bool deep_thin_ok(const Schedule& sc, const Structure& s) {
  return !getenv("GTNX_NO_DEEP") && sc.view.P >= 64 && sc.view.P <= sd_deep_node_cap() && int64_t(sc.view.L) * 4 >= sc.view.P &&
         s.A >= 0 && s.A <= 64 * int64_t(sc.view.P);
}

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
void linear_sd_backward_now(std::vector<Member>& ms, std::vector<Graph>& ins) {
  LinearSdOp lin;
  lin.tropical = false;
  lin.run_now(ms, ins);
}

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
                 sc.max_level_width <= sd_narrow_node_cap() && sc.max_reach <= sd_narrow_ring_backward() &&
                 narrow_degree_ok(*ms[i].out.g->inputs[0].s);
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
    std::vector<uint8_t> in_place_ok(size_t(n), 0);
    std::unordered_set<GradState*> place_seen;
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
        // (a chain that is accumulated into IN PLACE below needs no block of its own -- nor its share of the zero fill:
        //  T C floats per utterance)
        if (chain.calc_grad() && chain.is_grad_available() && place_seen.insert(chain.g.get()).second) {
          Weights& gw = *chain.grad().w;
          in_place_ok[i] = gw.dev_valid && !(gw.host_escaped && gw.host_valid) && gw.n == chain.num_arcs();
        }
        if (chain.calc_grad() && !in_place_ok[i]) fb = align_up(fb + 4 * size_t(chain.num_arcs()), 256);
      }
      fg = rt.alloc_zero(fb ? fb : 1);
    }
    std::vector<SdArgs> args(n);
    GradSink sink;
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
        a.grad_chain = chain.calc_grad() ? (in_place_ok[i] ? chain.grad().w->dev : fg->as<float>(off_c[i])) : nullptr;
        a.chunk_levels = std::max(1, std::min(a.chunk_levels, cap_c / std::max(sc.chain_C, 1)));
        if (a.grad_fixed) sink.add(fixed, fg, a.grad_fixed);
        // a chain that already holds a device gradient (e.g. from forwardScore(emissions),
        // run earlier in the sweep) is accumulated into in place: one pass, no axpy
        bool in_place = false;
        if (a.grad_chain && in_place_ok[i]) {  // (decided with the block sizes above)
          Weights& gw = *chain.grad().w;
          a.grad_chain = gw.dev;
          a.chain_accumulate = 1;
          gw.host_valid = false;
          gw.version++;
          in_place = true;
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
      bool deep = !narrow && mode == SD_LOG;
      int deep_p = 0;
      for (int i = 0; i < n && deep; ++i) {
        const Schedule& sc = *saved[ms[i].idx].sched;
        deep = deep_thin_ok(sc, *ms[i].out.g->inputs[0].s);
        deep_p = std::max(deep_p, sc.view.P);
      }
      if (deep)
        launch_sd_backward_deep(d->as<SdArgs>(), n, deep_p, rt.stream());
      else
        launch_sd_backward(d->as<SdArgs>(), n, mode, narrow ? (fuse ? 2 : 1) : 0,
                           int(tot_p ? (tot_out * 16) / tot_p : 0), rt.stream(), fuse_lds);
    }
    sink.flush();
  }
};


std::vector<Graph> op_shortest_distance(std::vector<Graph>& gs, bool tropical) {
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
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
               sc.max_reach <= sd_narrow_ring() && narrow_degree_ok(*gs[exp[k]].s);
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
    bool deep = !narrow && !tropical;
    int deep_p = 0;
    for (int k = 0; k < m && deep; ++k) {
      const Schedule& sc = *gs[exp[k]].s->sched;
      deep = deep_thin_ok(sc, *gs[exp[k]].s);
      deep_p = std::max(deep_p, sc.view.P);
    }
    if (deep)
      launch_sd_forward_deep(d->as<SdArgs>(), m, deep_p, rt.stream());
    else
      launch_sd_forward(d->as<SdArgs>(), m, op->mode, narrow ? (all_inw ? 2 : 1) : 0,
                        int(tot_p ? (tot_in * 16) / tot_p : 0), rt.stream());
  }
  return outs;
}

// ======================================================================
// viterbiPath (functions.cpp:328-330, shortest.cpp:190-272)
// ======================================================================
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

std::vector<Graph> op_viterbi_path(std::vector<Graph>& gs) {
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
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
  // the arena: what comes back to the host first and together (per graph: 4 ints + five arrays of `cap`), then the
  // per-position scratch (scores, back-pointers, predecessor positions: 12 bytes per node, never downloaded)
  size_t bytes = 0;
  std::vector<size_t> off_s(m), off_a(m), off_r(m), off_p(m), off_q(m), off_t(m);
  std::vector<int> cap(m);
  for (int k = 0; k < m; ++k) {
    const DSched& v = gs[k].s->sched->view;
    cap[k] = std::max(v.L, 1);
    off_p[k] = bytes;
    bytes = align_up(bytes + 20 * size_t(cap[k]) + 16, 256);
  }
  const size_t path_bytes = bytes;
  for (int k = 0; k < m; ++k) {
    const DSched& v = gs[k].s->sched->view;
    off_s[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(v.P), 256);
    off_a[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(v.P), 256);
    off_q[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(v.P), 256);
    off_t[k] = bytes;
    bytes = align_up(bytes + 4 * size_t(cap[k]), 256);
    off_r[k] = bytes;
    bytes += 256;
  }
  DevMemP arena = rt.alloc(bytes);
  std::vector<SdArgs> args(m);
  std::vector<PathArgs> pargs(m);
  int max_cap = 1, max_P = 1;
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
    p.pred = arena->as<int>(off_q[k]);
    p.tmp = arena->as<int>(off_t[k]);
    max_cap = std::max(max_cap, cap[k]);
    max_P = std::max(max_P, a.s.P);
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
               sc.max_level_width <= sd_narrow_node_cap() && sc.max_reach <= sd_narrow_ring() && narrow_degree_ok(*gs[k].s);
      tot_levels += sc.view.L;
    }
    narrow = narrow && tot_levels >= 32 * int64_t(m);
    launch_sd_forward(d->as<SdArgs>(), m, SD_PATH, narrow ? 2 : 0, int(tot_p ? (tot_in * 16) / tot_p : 0), rt.stream());
    launch_path_chase(dp->as<PathArgs>(), m, max_cap, max_P, rt.stream());
  }
  // the path is at most L arcs: bring it to the host (the path region of the arena only, into pinned memory: the
  // scores and back-pointers stay where they are -- 140 MB of them per 64 CTC lattices) and build the chain graph there
  PinnedMemP host_mem = rt.alloc_pinned(path_bytes ? path_bytes : 1);
  rt.d2h_sync(host_mem->ptr, arena->ptr, path_bytes);
  struct HostView {
    const char* p;
    const char* data() const { return p; }
  } host{host_mem->as<char>()};
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


} // namespace gtnx
