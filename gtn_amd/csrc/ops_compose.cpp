// ops_compose.cpp -- compose / intersect: capacity bounds, the choice between the FAST / general / wide-node kernels,
// symbolic products; see ops.h
#include "ops_internal.h"

namespace gtnx {

// ======================================================================
// compose / intersect (functions.cpp:225-251, compose.cpp:377-522)
// ======================================================================

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

std::vector<Graph> op_compose(std::vector<Graph>& av, std::vector<Graph>& bv, bool intersect) {
  GraphSlabScope slab_scope(std::max(av.size(), bv.size()));  // the results' pieces out of one allocation (graph.h)
  return op_compose_impl(av, bv, intersect, true);
}

// -1: the caller never said (gtnx_compose_mode).  The engine's own policy then: a chain product whose partner is
// a small graph resident on the HOST (a target the caller has just built: <= 512 nodes, <= 4 arcs per node) stays
// symbolic -- forwardScore / viterbi of it are one sweep kernel instead of a built lattice, the same choice a
// parallelMap region makes for its calls (region.h) -- and everything else is built.
thread_local int t_compose_mode = -1;
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
    const int mode_raw = env && env[0] >= '0' && env[0] <= '2' ? env[0] - '0' : t_compose_mode;
    const int mode = mode_raw < 0 ? 2 : mode_raw;
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
        if (!t_reclaim_at_wait) rt.drain_deferred();  // the step's reclamation point (see below)
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
        int64_t h = 0, ep = 0;
        if (!e.host_valid && e.dev_valid && e.A >= 0) {
          // a device-built partner (the product of an earlier compose): no download for a count -- every arc
          // may match, epsilons only if the structure has any (an upper bound is all the capacities need)
          h = e.A;
          ep = (e.dview.flags & GF_EPS_FREE) ? 0 : e.A;
        } else {
          e.ensure_host();
          const std::vector<int>& lab = l1 ? e.il : e.ol;
          for (int l : lab) {
            h += (l >= 0 && l < ch.C);
            ep += (l == GTNX_EPSILON);
          }
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
    const int mode_raw = env && env[0] >= '0' && env[0] <= '2' ? env[0] - '0' : t_compose_mode;
    const bool auto_mode = mode_raw < 0;  // nobody asked: symbolic for host-resident small partners only
    const int mode = auto_mode ? 2 : mode_raw;
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
      pairs = pairs && eligible && (!auto_mode || (l1 ? b : a).s->host_valid) &&
              lazy_pair_shape_ok(*(l1 ? a : b).s, *(l1 ? b : a).s);
      est += 44.0 * double(caps[i].Acap) + 30.0 * double(caps[i].Ncap) + 8.0 * double(caps[i].pairs);
    }
    if (eligible && (force || pairs || est > budget)) {
      // nothing downstream of a symbolic product waits for the GPU, so this is the step's
      // reclamation point (objects the caller let go of since the last one; cheap while
      // their memory is still warm for the allocator -- see Runtime::defer_delete).  A region that goes on to WAIT for
      // the GPU reclaims there instead (t_reclaim_at_wait).
      if (!t_reclaim_at_wait) rt.drain_deferred();
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
    x.trim_fwd_first = 0;
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
    if (!ex.host_valid) {  // a device-built partner: its degrees are not worth a download
      defer = false;
      break;
    }
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
        Structure& s2 = *bcast(bv, n, order[i]).s;
        x.s1_out = x.s1_in = x.s2_out = x.s2_in = nullptr;
        {
          // a narrow graph against a (nearly) complete one: trim from the start pairs first
          Structure& s1 = *bcast(av, n, order[i]).s;
          auto narrow_g = [](const Structure& s) { return s.A <= 4 * s.N; };
          auto complete_g = [](const Structure& s) { return s.N >= 8 && 2 * s.A >= s.N * s.N; };
          const char* env = getenv("GTNX_TRIM_FWD_FIRST");
          x.trim_fwd_first = env ? (env[0] != '0') : ((narrow_g(s1) && complete_g(s2)) || (narrow_g(s2) && complete_g(s1)));
        }
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
      rt.d2h_pinned_async(deferred->host->ptr, res->ptr, hdr.size());
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



} // namespace gtnx
