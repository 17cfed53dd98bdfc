// ops_band.cpp -- symbolic chain products with a BANDED partner (CTC targets, force-alignment acceptors): the band
// sweeps of band.hip, forward / backward / Viterbi; see ops.h
#include "ops_internal.h"

#include <memory>

namespace gtnx {

int band_vec_ok(const BandPair& p) { return p.C % 4 == 0 && (reinterpret_cast<uintptr_t>(p.em) & 15) == 0; }
// launches `tab` grouped by (C, nodes per lane, unit, G wants a gradient, 16-byte staging)
void band_launch_patched(const DevMemP& table, int n, const BandLaunchKey& k, int max_ns, const BandPatch& patch,
                         const char* prof_name, double prof_bytes) {
  Runtime& rt = Runtime::get();
  std::unique_ptr<Runtime::Scope> prof(prof_name ? new Runtime::Scope(&rt, prof_name, prof_bytes) : nullptr);
  launch_band_backward(table->as<BandPair>(), n, k.npl, k.C, max_ns, k.unit != 0, k.gradg != 0, k.vec != 0, rt.stream(), nullptr,
                       &patch);
}
void band_launch(std::vector<std::pair<BandLaunchKey, BandPair>>& tab, bool backward, const char* prof_name, double prof_bytes,
                 DevMemP* table_out) {
  Runtime& rt = Runtime::get();
  if (table_out) table_out->reset();
  if (tab.empty()) return;
  bool one_key = true;  // (a criterion step: every pair has the same shape -- nothing to group)
  for (size_t i = 1; i < tab.size() && one_key; ++i) one_key = tab[i].first == tab[0].first;
  if (!one_key) std::stable_sort(tab.begin(), tab.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  if (tab.size() == 1) {  // one utterance through the per-graph functions: the record travels with the launch
    const BandLaunchKey& k = tab[0].first;
    const BandPair& p = tab[0].second;
    if (band_one_ok(k.npl, k.C, p.NS, k.vec != 0, backward)) {
      std::unique_ptr<Runtime::Scope> prof(prof_name ? new Runtime::Scope(&rt, prof_name, prof_bytes) : nullptr);
      if (backward) launch_band_backward(nullptr, 1, k.npl, k.C, p.NS, k.unit != 0, k.gradg != 0, true, rt.stream(), &p);
      else launch_band_forward(nullptr, 1, k.npl, k.C, p.NS, k.unit != 0, true, rt.stream(), &p);
      return;
    }
  }
  std::vector<BandPair> flat;
  flat.reserve(tab.size());
  for (auto& e : tab) flat.push_back(e.second);
  DevMemP d = upload_vec(flat);
  const BandPair* dp = d->as<BandPair>();
  if (table_out && one_key) *table_out = d;
  // (the profiled span is the KERNEL launches: the table's upload -- a 5 us copy kernel -- is in front of it.  bench.py's
  //  roofline figure is algorithmic bytes over this span; until round 6 the span held the upload too and read 1.5 % low
  //  against rocprofv3's kernel time.)
  std::unique_ptr<Runtime::Scope> prof(prof_name ? new Runtime::Scope(&rt, prof_name, prof_bytes) : nullptr);
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
// ---- one workgroup per (chain, BANDED G) pair: band.hip.  CTC targets and force-alignment
// acceptors: a single wave carries the whole recursion, the other waves stage.
bool band_shape_ok(const Structure& cs, Structure& fs, bool chain_first) {
  if (cs.kind != KIND_LINEAR || cs.C < band_min_labels() || cs.C > band_max_labels() || cs.M < 0 || cs.M > (1 << 20)) return false;
  std::shared_ptr<BandInfo> b = band_info(fs, chain_first);
  return b->ok && b->max_label < cs.C;
}
bool band_ok(const LazyProduct& lp, std::shared_ptr<BandInfo>* out) {
  if (!band_shape_ok(*lp.chain.s, *lp.fixed.s, lp.chain_side == 1)) return false;
  if (out) *out = band_info(*lp.fixed.s, lp.chain_side == 1);
  return true;
}
// band records and the all-zero test of a batch of partners, on the worker pool (a training step
// brings one fresh target graph per utterance); both are cached on the graph afterwards
static bool tie_ranks(Structure& fs, BandInfo& bi, bool use_ilabel);
void band_prepare(const std::vector<Graph*>& fixed, const std::vector<uint8_t>& chain_first, bool want_ranks) {
  std::vector<size_t> todo;
  std::unordered_set<Structure*> seen;
  for (size_t i = 0; i < fixed.size(); ++i) {
    Structure* st = fixed[i]->s.get();
    const std::shared_ptr<BandInfo>& b = st->band[chain_first[i] ? 0 : 1];
    if ((!b || (want_ranks && b->ok && b->rank_state == 0)) && seen.insert(st).second) todo.push_back(i);
  }
  auto body = [&](size_t q) {
    Graph& g = *fixed[todo[q]];
    std::shared_ptr<BandInfo> b = band_info(*g.s, chain_first[todo[q]] != 0);
    (void)g.w->is_all_zero();
    // (a decode: the orders that decide exact ties, while a pool thread has the target in its cache)
    if (want_ranks && b && b->ok) (void)tie_ranks(*g.s, *b, chain_first[todo[q]] != 0);
  };
  if (todo.size() >= 64) gtn::detail::runIndexed(todo.size(), body, 32, false);
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
  static void launch(std::vector<std::pair<Key, BandPair>>& tab, bool backward, const char* prof_name = nullptr, double prof_bytes = 0.0) {
    band_launch(tab, backward, prof_name, prof_bytes);
  }

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
      p.delta = through_delta(ms[k].out);
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
    BandSdOp::launch(tab, true, "band_forward_score_grad", plan->bytes);
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
    linear_sd_backward_now(rest, rest_in);
  }
  plan->band.clear();
  plan->lin.clear();
  plan->keep.clear();
  plan->bytes = 0;
}

std::vector<Graph> band_forward_score(std::vector<Graph>& gs) {
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
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
    BandSdOp::launch(tab, false, "band_forward_score", abytes);
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
      a.delta = through_delta(m.out);
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

// ---- exact ties without the lattice (CTC-shaped targets) --------------------------------------------------------
// When two candidates into a node of the product are EQUAL, the reference's answer depends on orders its own data
// structures define: viterbiPath keeps the arc relaxed first, i.e. whose source node left its queue first
// (shortest.cpp:208-224: strictly greater replaces); viterbiScore's gradient keeps the first maximum in the node's
// in-list (shortest.cpp:118-127), i.e. the arc compose created first; both take the first best node of accept()
// (:233-244, :148-160), i.e. the accept node compose created first.  For the product of an emission chain with a
// CTC target acceptor both orders are, in EVERY layer of the lattice, the restriction of one fixed order of the
// target's nodes to the nodes alive in that layer (pruning by co-reachability does not disturb it), and that order
// is the fixed point of a recursion over the target graph alone:
//   queue order     a node enters the next layer when the LAST of its sources (in this layer's order, then in the
//                   source's out-list order) is processed            (in-degree reaches zero: shortest.cpp:221-223)
//   creation order  ... when the FIRST of them is                    (first discovery: compose.cpp's queue)
// -- checked against the unmodified reference's lattices (queue replayed on them, node ids read off them) for 1 680
// random targets with repeated labels, any blank, T down to the shortest feasible, both argument orders
// (tests/test_band_tie_ranks.py runs a committed sample of that on the CPU).  The out-list order of a product node
// is the target node's, provided its out-lists are strictly increasing in the matched label (arcSort'ed target:
// every matcher of compose.cpp then emits in label order); otherwise -- and for any target that is not exactly
// ctcGraph(labels) -- the tied utterance takes the built lattice as before.
static bool layer_order(const Structure& s, int start, bool last_touch, std::vector<int>& rank) {
  const int N = int(s.N);
  std::vector<int> layer{start}, next;
  std::vector<long long> key(size_t(N), -1);
  bool fixed = false;
  for (int it = 0; it < 4 * N + 8 && !fixed; ++it) {
    std::fill(key.begin(), key.end(), -1);
    for (size_t i = 0; i < layer.size(); ++i) {
      const int l = layer[i];
      for (int k = s.out_off[size_t(l)]; k < s.out_off[size_t(l) + 1]; ++k) {
        const int d = s.dst[size_t(s.out_list[size_t(k)])];
        const long long kk = (long long)(i) * 8 + (k - s.out_off[size_t(l)]);
        long long& slot = key[size_t(d)];
        slot = slot < 0 ? kk : (last_touch ? std::max(slot, kk) : std::min(slot, kk));
      }
    }
    next.clear();
    for (int d = 0; d < N; ++d)
      if (key[size_t(d)] >= 0) next.push_back(d);
    std::sort(next.begin(), next.end(), [&](int a, int b) { return key[size_t(a)] < key[size_t(b)]; });
    fixed = next == layer;
    layer.swap(next);
  }
  if (!fixed || int(layer.size()) != N) return false;
  rank.assign(size_t(N), 0);
  for (int i = 0; i < N; ++i) rank[size_t(layer[size_t(i)])] = i;
  return true;
}

static bool tie_ranks(Structure& fs, BandInfo& bi, bool use_ilabel) {
  if (bi.rank_state) return bi.rank_state > 0;
  bi.rank_state = -1;
  fs.ensure_host();
  detect_ctc_shape(fs);
  if (!fs.ctc_labels || fs.N > 512) return false;
  fs.ensure_csr();
  const std::vector<int>& lab = use_ilabel ? fs.il : fs.ol;
  for (int64_t n = 0; n < fs.N; ++n) {
    if (fs.out_off[size_t(n) + 1] - fs.out_off[size_t(n)] > 8) return false;
    for (int k = fs.out_off[size_t(n)] + 1; k < fs.out_off[size_t(n) + 1]; ++k)
      if (lab[size_t(fs.out_list[size_t(k)])] <= lab[size_t(fs.out_list[size_t(k) - 1])]) return false;
  }
  // ---- closed form (round 6) when the blank is below every label of the target (blank 0, labels >= 1: every CTC
  // criterion of the reference's tests, examples and benchmarks).  The out-list of a blank node is then (self, next),
  // of label node 2i+1 (next blank first, then self and the skip to 2i+3 by label value), and the fixed point of the
  // recursion above is, with p = the number of leading strict ascents t[0] < t[1] < ... < t[p]:
  //   queue order     0 | (2i, 2i-1) for i = 1..p | N-1 | for i = U-1 down to p+1: (2i, 2i+1) if t[i] != t[i-1] (a skip
  //                   arc enters 2i+1) else (2i+1, 2i) | 2p+1
  //   creation order  the identity
  // -- found by running layer_order() on samples and checked against it on 30 000 random targets (U = 0..40, alphabets
  // of 2..30 labels, up to 70 % repeats: no difference), in the tree as tests/test_band_tie_ranks.py under
  // GTNX_CHECK_CLOSED_RANKS=1, where BOTH are computed here and a difference throws.  The ranks of a whole batch then
  // cost microseconds, and a decode runs its FIRST launch with them (band_viterbi below): no second launch, no second
  // copy, for the utterance in five hundred that has an exact tie on its best path.
  static const bool no_closed = std::getenv("GTNX_NO_CLOSED_RANKS") != nullptr;
  static const bool check_closed = std::getenv("GTNX_CHECK_CLOSED_RANKS") != nullptr;
  const std::vector<int>& t = *fs.ctc_labels;
  bool below = !no_closed;
  for (size_t i = 0; i < t.size() && below; ++i) below = t[i] > fs.ctc_blank;
  if (below) {
    const int U = int(t.size()), N = 2 * U + 1;
    std::vector<int>& rk = bi.rank_kahn;
    rk.assign(size_t(N), 0);
    if (U > 0) {
      int p = 0;
      while (p + 1 < U && t[size_t(p)] < t[size_t(p) + 1]) ++p;
      int pos = 1;
      for (int i = 1; i <= p; ++i) {
        rk[size_t(2 * i)] = pos++;
        rk[size_t(2 * i - 1)] = pos++;
      }
      rk[size_t(N - 1)] = pos++;
      for (int i = U - 1; i > p; --i) {
        const bool skip = t[size_t(i)] != t[size_t(i) - 1];
        rk[size_t(skip ? 2 * i : 2 * i + 1)] = pos++;
        rk[size_t(skip ? 2 * i + 1 : 2 * i)] = pos++;
      }
      rk[size_t(2 * p + 1)] = pos++;
    }
    bi.rank_create.resize(size_t(N));
    for (int n = 0; n < N; ++n) bi.rank_create[size_t(n)] = n;
    if (check_closed) {
      std::vector<int> k2, c2;
      if (!layer_order(fs, 0, true, k2) || !layer_order(fs, 0, false, c2) || k2 != bi.rank_kahn || c2 != bi.rank_create)
        throw_runtime("[gtnx] tie ranks: the closed form differs from the fixed point of the recursion");
    }
    bi.rank_state = 1;
    return true;
  }
  if (!layer_order(fs, 0, /*last_touch=*/true, bi.rank_kahn) || !layer_order(fs, 0, /*last_touch=*/false, bi.rank_create))
    return false;
  bi.rank_state = 1;
  return true;
}

bool ctc_tie_ranks(Structure& s, bool use_ilabel, const std::vector<int>** kahn, const std::vector<int>** create) {
  std::shared_ptr<BandInfo> bi = band_info(s, use_ilabel);
  if (!bi || !bi->ok || !tie_ranks(s, *bi, use_ilabel)) return false;
  if (kahn) *kahn = &bi->rank_kahn;
  if (create) *create = &bi->rank_create;
  return true;
}

std::vector<Graph> band_viterbi(std::vector<Graph>& gs, bool want_path) {
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
  Runtime& rt = Runtime::get();
  const size_t n = gs.size();
  GTNX_HOST_T("band_viterbi.total");
  double t_mark = HostTimer::enabled() ? host_now_ms() : 0.0;
  auto lap = [&](const char* name) {
    if (!HostTimer::enabled()) return;
    const double t = host_now_ms();
    host_timer_add(name, t - t_mark);
    t_mark = t;
  };
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
    band_prepare(fx, cf, /*want_ranks=*/true);
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
  // per pair, device only: back-pointers [T][NS] bytes | pnode [T+1]; and what the host reads (an arena of its own:
  // a decode copies 12 T + 16 bytes per utterance back, not the back-pointer planes): path arc, label, weight [T]
  // each | len, score, tie
  size_t bytes = 0, wbytes = 0;
  std::vector<size_t> o_bp(n), o_pn(n), o_pa(n), o_hd(n);
  int max_c = 1;
  for (size_t i = 0; i < n; ++i) {
    const LazyProduct& lp = *gs[i].s->lazy;
    const size_t T = size_t(lp.chain.s->M), N = size_t(lp.fixed.s->N);
    const size_t ns = size_t(band_row_stride(int(N), band_npl(int(N))));
    o_bp[i] = wbytes;
    wbytes = align_up(wbytes + T * ns + 512, 256);
    o_pn[i] = wbytes;
    wbytes = align_up(wbytes + 4 * (T + 1), 256);
    o_pa[i] = bytes;
    bytes = align_up(bytes + 12 * (T ? T : 1), 256);
    o_hd[i] = bytes;
    bytes = align_up(bytes + 16, 256);
    max_c = std::max(max_c, lp.chain.s->C);
  }
  lap("band_viterbi.1_prepare");
  DevMemP arena = rt.alloc(bytes);
  DevMemP work = rt.alloc(wbytes);
  const int stage_floats = std::max(4096, max_c);
  int max_n = 1, vec = 1;
  double abytes = 0;  // 4TC in, T N / 2 of back-pointers out and in, 20 T of path out
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
    p.bp = work->as<uint8_t>(o_bp[i]);
    p.pnode = work->as<int>(o_pn[i]);
    p.path_arc = arena->as<int>(o_pa[i]);
    p.path_lab = p.path_arc + (p.T ? p.T : 1);
    p.path_w = reinterpret_cast<float*>(p.path_lab + (p.T ? p.T : 1));
    p.path_len = arena->as<int>(o_hd[i]);
    p.score = reinterpret_cast<float*>(p.path_len + 1);
    p.tie = p.path_len + 2;
    p.stage_floats = stage_floats;
    max_n = std::max(max_n, p.N);
    if (p.C % 4 != 0 || (reinterpret_cast<uintptr_t>(p.em) & 15) != 0 || int64_t(p.T) * p.C < 4) vec = 0;
    abytes += 4.0 * p.T * p.C + 0.5 * double(p.T) * p.N + 20.0 * p.T;
  }
  // Targets whose tie orders are known up front (tie_ranks: CTC-shaped, label-sorted -- a closed form, computed with
  // the band records) are decoded by the RANKED kernel at once: equal candidates are decided as the reference's queue
  // / in-lists decide them, no tie is ever reported, and the second launch (+ its copy and host round trip: 1.2 ms per
  // batch that has ONE tied utterance) never happens.  (The ranked kernel is 6 % slower: 0.41 against 0.39 ms.)
  static const bool no_ranked_first = std::getenv("GTNX_NO_RANKED_TIES") != nullptr || std::getenv("GTNX_NO_RANKED_FIRST") != nullptr;
  bool ranked_first = !no_ranked_first && n > 0 && band_viterbi_wave_ok(max_n, max_c, vec);
  for (size_t i = 0; i < n && ranked_first; ++i) ranked_first = infos[i]->rank_state == 1;
  {
    DevMemP drk0;
    if (ranked_first) {
      std::vector<int> rk;
      std::vector<size_t> off(n);
      size_t total = 0;
      for (size_t i = 0; i < n; ++i) total += 2 * size_t(tab[i].N);
      rk.reserve(total);
      for (size_t i = 0; i < n; ++i) {
        const BandInfo& b = *infos[i];
        off[i] = rk.size();
        const std::vector<int>& rin = want_path ? b.rank_kahn : b.rank_create;
        rk.insert(rk.end(), rin.begin(), rin.end());
        rk.insert(rk.end(), b.rank_create.begin(), b.rank_create.end());
      }
      drk0 = upload_vec(rk);
      for (size_t i = 0; i < n; ++i) {
        tab[i].rank_in = drk0->as<int>() + off[i];
        tab[i].rank_acc = tab[i].rank_in + tab[i].N;
      }
    }
    DevMemP d = upload_vec(tab);
    GTNX_PROF(want_path ? "band_viterbi_path" : "band_viterbi_score", abytes);
    launch_band_viterbi(d->as<BandDecode>(), int(n), stage_floats, max_n, max_c, vec, rt.stream(), ranked_first ? 1 : 0);
  }
  // heads (length, score, tie) of every pair; the paths themselves only when they become graphs
  // (a pinned block: the copy into pageable memory ran at a fifth of the link's rate)
  PinnedMemP host_pin = rt.alloc_pinned(bytes ? bytes : 1);
  struct HostView {
    char* p;
    char* data() const { return p; }
  } host{host_pin->as<char>()};
  auto fetch_results = [&]() {
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
  };
  fetch_results();
  lap("band_viterbi.2_launch_and_copy");
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
  auto head_of = [&](size_t i) { return reinterpret_cast<const int*>(host.data() + o_hd[i]); };
  // ---- exact ties on a best path.  CTC-shaped targets: a second launch over those utterances with the reference's
  // orders as node ranks (tie_ranks above); anything else: the built lattice decides (below)
  static const bool dbg_ties = std::getenv("GTNX_DEBUG_TIES") != nullptr;
  static const bool no_ranked = std::getenv("GTNX_NO_RANKED_TIES") != nullptr;
  std::vector<size_t> rerun;  // utterances whose ties a second launch (node ranks) is deciding
  {
    std::vector<size_t> ranked;
    size_t n_tied = 0;
    for (size_t i = 0; i < n; ++i) {
      const int* hd = head_of(i);
      if (!(hd[2] && hd[0] >= 0)) continue;
      ++n_tied;
      LazyProduct& lp = *gs[i].s->lazy;
      if (!no_ranked && band_viterbi_wave_ok(max_n, max_c, vec) && tie_ranks(*lp.fixed.s, *infos[i], lp.chain_side == 1))
        ranked.push_back(i);
    }
    if (dbg_ties && n_tied)
      std::fprintf(stderr, "[gtnx] band_viterbi: %zu of %zu utterances report an exact tie on their best path (%zu decided by node ranks)\n",
                   n_tied, n, ranked.size());
    if (!ranked.empty()) {
      std::vector<int> rk;
      std::vector<size_t> off(ranked.size());
      for (size_t k = 0; k < ranked.size(); ++k) {
        const BandInfo& b = *infos[ranked[k]];
        off[k] = rk.size();
        // viterbiPath: in-arc ties by the queue's order; viterbiScore: by the in-lists' (= creation) order; accept
        // ties by the accept list's (= creation) order either way
        const std::vector<int>& rin = want_path ? b.rank_kahn : b.rank_create;
        rk.insert(rk.end(), rin.begin(), rin.end());
        rk.insert(rk.end(), b.rank_create.begin(), b.rank_create.end());
      }
      DevMemP drk = upload_vec(rk);
      std::vector<BandDecode> tab2(ranked.size());
      for (size_t k = 0; k < ranked.size(); ++k) {
        tab2[k] = tab[ranked[k]];
        tab2[k].rank_in = drk->as<int>() + off[k];
        tab2[k].rank_acc = tab2[k].rank_in + tab2[k].N;
      }
      {
        DevMemP d2 = upload_vec(tab2);
        GTNX_PROF(want_path ? "band_viterbi_path_ranked" : "band_viterbi_score_ranked", 0.0);
        launch_band_viterbi(d2->as<BandDecode>(), int(tab2.size()), stage_floats, max_n, max_c, vec, rt.stream(), /*ranked=*/1);
      }
      // (the second launch runs while the host builds the path graphs of the utterances it does not concern:
      //  round 5 -- the two were in series, 0.65 ms per batch that has a tie)
      rerun = std::move(ranked);
      lap("band_viterbi.2b_ranked_rerun_enqueue");
    }
  }
  // The path graphs (a dozen host arrays of T entries per utterance): every element touches only its own objects, so
  // a large batch is built by a few threads of the caller's pool (idle at a join; graph.cpp's ensure_host_batch and
  // band_prepare above fan out the same way).  Those workers belong to the pool of device 0 and never take the
  // caller's device: what make_result() stamps on a result there (device, home list) is THEIR thread's, so the
  // caller's are captured here and written over it -- a result belongs to the thread that asked for it.  (Leaving
  // the HOME list the builder's, so that the pool threads take the path graphs apart, was measured in round 6:
  // 5.5-5.9 ms per decode batch against 3.9 -- the workers drain their lists inside the next batch's pool phase.)
  const int caller_device = Runtime::current_device();
  const Runtime::InboxP caller_home = Runtime::home();
  auto stamp = [&](Graph& out) {
    out.s->device = caller_device;
    out.s->home = caller_home;
  };
  std::mutex tied_mu;
  if (want_path)
    for (size_t i = 0; i < n; ++i) gs[i].s->lazy->fixed.s->ensure_host();
  auto build_output = [&](size_t i) {
    const int* hd = head_of(i);
    const int len = hd[0];
    if (hd[2] && len >= 0) {  // an exact tie the ranks do not cover: the built lattice decides (its node numbering breaks it)
      std::lock_guard<std::mutex> lk(tied_mu);
      tied.push_back(i);
      return;
    }
    LazyProduct& lp = *gs[i].s->lazy;
    const int chain_first = lp.chain_side == 1;
    if (want_path) {
      Graph out = make_output(pop, int(i), {gs[i]});
      stamp(out);
      if (len >= 0) {
        const int* harc = reinterpret_cast<const int*>(host.data() + o_pa[i]);
        const int* hlab = harc + (tab[i].T ? tab[i].T : 1);
        const float* hw = reinterpret_cast<const float*>(hlab + (tab[i].T ? tab[i].T : 1));
        // labels of the product's arcs: the chain's on its side, G's arc label on the other
        // (the partner's host arrays were made sure of before the threads started)
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
      stamp(out);
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
  };
  {
    std::vector<uint8_t> later(n, 0);
    for (size_t i : rerun) later[i] = 1;
    std::vector<size_t> now;
    now.reserve(n);
    for (size_t i = 0; i < n; ++i)
      if (!later[i]) now.push_back(i);
    auto build_many = [&](const std::vector<size_t>& idx) {
      if (want_path && idx.size() >= 64) gtn::detail::runIndexed(idx.size(), [&](size_t q) { build_output(idx[q]); }, 8, false);
      else for (size_t i : idx) build_output(i);
    };
    build_many(now);
    if (!rerun.empty()) {
      // only what the second launch rewrote comes back: the reruns' own path blocks and heads (a handful of 12 KB
      // blocks, not the whole 6 MB arena a second time)
      if (want_path && rerun.size() * 8 < n) {
        for (size_t i : rerun) {
          const size_t T = size_t(tab[i].T ? tab[i].T : 1);
          rt.d2h_pinned_async(host.data() + o_pa[i], arena->as<char>(o_pa[i]), 12 * T);
          rt.d2h_pinned_async(host.data() + o_hd[i], arena->as<char>(o_hd[i]), 16);
        }
        rt.sync_while_draining();
      } else {
        fetch_results();
      }
      lap("band_viterbi.2c_ranked_rerun_wait");
      build_many(rerun);
    }
    std::sort(tied.begin(), tied.end());
  }
  lap("band_viterbi.3_outputs");
  if (!tied.empty()) {
    if (dbg_ties) std::fprintf(stderr, "[gtnx] band_viterbi: %zu utterance(s) rerun on the built lattice\n", tied.size());
    std::vector<Graph> tg;
    for (size_t i : tied) {
      // The lattice is built and its level schedule taken by replaying the reference's queue on it
      // (graph.cpp: build_host_schedule, as for any host-built graph) instead of the id-order schedule a
      // layered product normally gets for free: under exact ties the winner is the arc whose source left
      // the queue first (shortest.cpp:212-227), and that order is not the node-id order.
      realize(gs[i]);
      lap("band_viterbi.4a_realize");
      gs[i].s->resolve_sizes();
      lap("band_viterbi.4b_sizes");
      gs[i].s->ensure_full();
      gs[i].s->ensure_host();
      lap("band_viterbi.4c_host_copy");
      gs[i].s->sched.reset();
      tg.push_back(gs[i]);
    }
    std::vector<Graph> to = want_path ? op_viterbi_path(tg) : op_shortest_distance(tg, true);
    for (size_t k = 0; k < tied.size(); ++k) outs[tied[k]] = std::move(to[k]);
    lap("band_viterbi.4d_rerun_op");
  }
  return outs;
}


} // namespace gtnx
