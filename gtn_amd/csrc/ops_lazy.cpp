// ops_lazy.cpp -- symbolic chain products scored by the record-walking, dense (MFMA), max-plus and per-pair kernels
// (lazy.hip, maxplus.hip, lazy_pair.hip), and viterbiPath of the record-walking route; see ops.h
#include "ops_internal.h"

namespace gtnx {

// ======================================================================
// Lazy chain products (kernels: lazy.hip).  compose(chain, G) / compose(G, chain)
// with an implicit linear chain and an epsilon-free G is kept SYMBOLIC when building
// it is infeasible (or GTNX_LAZY_COMPOSE=1): forwardScore / viterbiScore /
// viterbiPath and their gradients then run as time-synchronous dynamic programs
// over (t, node of G), batched over the utterances that share G.  Any other use of
// the result (inspection, another op) realises it through the ordinary compose.
// ======================================================================

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
  // The beta sweep of a dense product whose inputs want gradients, started WITH the alpha sweep on the runtime's
  // side stream (runtime.h side_launch): beta[t] depends on beta[t+1] and the emissions only, not on alpha, and a
  // sweep is T dependent launches of ~8 us whatever the batch (profiles/r04_c4_batch_scaling.json: 8.2 ms per pass
  // with 64 workgroups per step, 8.8 with 256) -- the chip holds both chains' workgroups side by side (120 VGPRs,
  // 39 KB of LDS: two per CU).  backward() then finds beta done and runs the gradient contractions only.  The
  // price: 4 (T+1) nb N bytes held from forward to backward like alpha is, and a wasted sweep when the caller never
  // calls backward on a graph it built with calcGrad = true (the default of Graph) -- a scoring loop.  So the early
  // sweep is only started while the process has been SEEN to differentiate such products: the first backward of a
  // dense product turns it on (that step runs the sweep in backward as before), an early sweep nobody used turns
  // it off again (eager_beta_follows below).  GTNX_NO_EAGER_BETA=1: never; GTNX_EAGER_BETA=1: always.
  struct EagerBeta {
    DevMemP beta, planes;
    Runtime* rt = nullptr;
    Runtime::SideJobP job;
    bool used = false;
    ~EagerBeta();
  };
  std::shared_ptr<EagerBeta> eager;
};

namespace {
std::atomic<int> eager_beta_follows{0};  // backward has followed the forward of a dense product (LazyGroupState::EagerBeta)
}
LazyGroupState::EagerBeta::~EagerBeta() {
  // the buffers go back to the pool in engine-stream order, so that stream has to be behind the side stream's
  // launches first (a sweep nobody asked the result of)
  if (!used) eager_beta_follows.store(0, std::memory_order_relaxed);
  try {
    if (rt && job) rt->side_join(job);
  } catch (...) {
  }
}

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
std::shared_ptr<Structure::DenseInfo> dense_info(Structure& fs, bool chain_first, int C);
// THE predicate of the dense regime (one label per node's in-arcs, G at least half complete, 8 .. 1024 nodes):
// the log semiring then runs in the probability domain (matrix cores unless GTNX_DENSE_VALU), the tropical one on
// the max-plus sweeps.  `di` comes back whenever the facts were taken (the walkers use its degree statistics).
bool lazy_dense_ok(Structure& fs, bool chain_first, int C, std::shared_ptr<Structure::DenseInfo>* di_out) {
  std::shared_ptr<Structure::DenseInfo> di;
  if (!getenv("GTNX_NO_DENSE") && fs.N <= 1024 && fs.N >= 8) di = dense_info(fs, chain_first, C);
  if (di_out) *di_out = di;
  return di && !di->lab.empty() && 2 * di->valid >= int64_t(fs.N) * fs.N;
}
SymbolicRoute lazy_group_route(const LazyProduct& lp, bool tropical) {
  Structure& fs = *lp.fixed.s;
  const Structure& cs = *lp.chain.s;
  std::shared_ptr<Structure::DenseInfo> di;
  if (!lazy_dense_ok(fs, lp.chain_side == 1, cs.C, &di)) return ROUTE_WALK;
  if (tropical) return (cs.M >= 1 && di->ncol > 0) ? ROUTE_MAXPLUS : ROUTE_WALK;
  return getenv("GTNX_DENSE_VALU") ? ROUTE_DENSE : ROUTE_DENSE_MFMA;
}
std::vector<int> lazy_node_labels(Structure& fs, bool chain_first, int C, int* max_in_deg);
// host facts about a fixed partner G for the dense regime, taken once per structure (a trainer keeps its
// transitions graph; only the weights move)
bool labels_unique(const std::vector<int>& lab) {
  std::unordered_set<int> seen;
  for (int l : lab)
    if (l >= 0 && !seen.insert(l).second) return false;
  return true;
}
// Exact ties in viterbiPath of a product that is never built: the reference keeps the arc whose SOURCE left its queue
// first (shortest.cpp:208-224).  When every node of G has exactly one arc to every node that has in-arcs ("core"
// nodes), listed in increasing node order -- an ASG transitions graph, arcSort'ed or as built -- that order is the
// node order in EVERY layer of the product: a core node enters the next layer when its last source is processed,
// the last source is the same node for all of them (the last one of the current layer: everybody reaches everybody),
// and they enter in that node's out-list order; layer 0 is the start nodes, whose out-lists have the same order.
// compose discovers the nodes of a layer in the out-list order of the FIRST node of the previous one, so the accept
// list of the product is in node order as well.  The max-plus walk breaks ties that way (smallest source node, first
// accept node), so for such graphs its answer is the reference's -- checked against the unmodified reference under
// integer weights at alphabets whose in-lists std::sort scrambles (tests/test_lazy_gpu.py).
static bool dense_ties_by_node_order(const Structure& fs, const std::vector<int>& matched, int C) {
  const int N = int(fs.N);
  std::vector<char> core(size_t(N), 0);
  int ncore = 0;
  for (int n = 0; n < N; ++n)
    if (fs.in_off[size_t(n) + 1] > fs.in_off[size_t(n)]) core[size_t(n)] = 1, ++ncore;
  if (ncore == 0) return false;
  for (int l : matched)
    if (l < 0 || l >= C) return false;  // (an arc the chain cannot match is missing from the product)
  for (int n = 0; n < N; ++n) {
    const int k0 = fs.out_off[size_t(n)], k1 = fs.out_off[size_t(n) + 1];
    if (k1 - k0 == 0 && !core[size_t(n)]) continue;  // (an isolated node takes no part)
    if (k1 - k0 != ncore) return false;
    int want = 0;
    for (int k = k0; k < k1; ++k) {
      while (want < N && !core[size_t(want)]) ++want;
      if (want >= N || fs.dst[size_t(fs.out_list[size_t(k)])] != want) return false;
      ++want;
    }
  }
  for (size_t i = 1; i < fs.accept.size(); ++i)
    if (fs.accept[i] <= fs.accept[i - 1]) return false;
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
    di->ties_by_node_order = dense_ties_by_node_order(fs, ml, C);
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
    const bool dense_ok = lazy_dense_ok(fs, st.view.chain_first != 0, C, &di);
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
    rt.h2d_pinned(st.arena->as<char>(o_em), pin->ptr, 8 * size_t(nb));
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
    if (mode != SD_LOG) continue;
    std::shared_ptr<Structure::DenseInfo> di;
    const bool dense_ok = lazy_dense_ok(fs, v.chain_first != 0, v.C, &di);
    if (di) {
      st.max_in_deg = di->max_in_deg;
      st.lab_unique = di->uniq;
    }
    if (!dense_ok) continue;
    const std::vector<int>& lab = di->lab;
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
    v.rel = st.mfma ? 1 : 0;  // the matrix-core sweeps keep their planes relative to per-row references (lazy.hip)
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
        {
          static const bool eager_off = getenv("GTNX_NO_EAGER_BETA") != nullptr || getenv("GTNX_CHAIN") != nullptr;
          static const bool eager_always = getenv("GTNX_EAGER_BETA") != nullptr;
          bool wants = st.fixed.calc_grad();
          for (size_t b = 0; b < st.chains.size() && !wants; ++b) wants = st.chains[b].calc_grad();
          if (wants && !eager_off && st.view.T >= 1 && (eager_always || eager_beta_follows.load(std::memory_order_relaxed))) {
            auto eb = std::make_shared<LazyGroupState::EagerBeta>();
            const LazyGroup& sv = st.view;
            const size_t xf = align_up(size_t(sv.Kpad) * size_t(sv.nbpad), 64);
            eb->beta = rt.alloc(4 * size_t(sv.nb) * size_t(sv.N) * size_t(sv.T + 1));
            eb->planes = rt.alloc(8 * xf);
            eb->rt = &rt;
            LazyGroup v = sv;  // the sweep's own argument: beta planes and operand planes of its own
            v.beta = eb->beta->as<float>();
            v.xt[0] = eb->planes->as<float>();
            v.xt[1] = v.xt[0] + xf;
            eb->job = rt.side_launch([v](hipStream_t side) {
              launch_lazy_init(v, 1, side);
              launch_lazy_mfma_init(v, 1, side);
              for (int t = v.T - 1; t >= 0; --t) launch_lazy_mfma_step(v, t, 1, side);
              launch_lazy_mfma_rowmax(v, 1, side);
            });
            st.eager = eb;
          }
        }
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
    if (st.dense && st.mfma) launch_lazy_mfma_score(st.view, rt.stream());  // + the sum of the row references, in float64
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
        delta[b] = through_delta(of_slot[b]->out);
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
      rt.h2d_pinned(tabs->ptr, pin->ptr, 16 * size_t(nb));
      v.delta = reinterpret_cast<const float* const*>(tabs->as<char>());
      v.grad_em = reinterpret_cast<float* const*>(tabs->as<char>(8 * size_t(nb)));
      if (mode == SD_LOG) {
        DevMemP beta = st.eager ? st.eager->beta : rt.alloc(4 * plane * size_t(T + 1));
        v.beta = beta->as<float>();
        GTNX_PROF("lazy_forward_score_grad", 0.0);
        if (st.dense && st.mfma) eager_beta_follows.store(1, std::memory_order_relaxed);
        if (st.eager) {
          st.eager->used = true;
          rt.side_join(st.eager->job);  // the sweep ran next to the alpha sweep (LazyGroupState::EagerBeta)
        } else {
          launch_lazy_init(v, 1, rt.stream());
        }
        if (st.eager) {
        } else if (st.dense && st.mfma) {
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
            const size_t pb = lazy_mfma_fixed_grad_scratch_bytes(v);
            DevMemP parts = pb ? rt.alloc(pb) : nullptr;
            launch_lazy_mfma_fixed_grad(v, pcm->ptr, rt.stream(), parts ? parts->ptr : nullptr, pb);
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
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
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
      p.delta = through_delta(ms[k].out);
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
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
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

std::atomic<int64_t> g_viterbi_ties_seen{0}, g_viterbi_ties_unresolved{0};

void LazyPathOp::backward(std::vector<Member>& ms) {
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
      a.delta = through_delta(m.out);
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

std::vector<Graph> lazy_viterbi_path(std::vector<Graph>& gs) {
  GraphSlabScope slab_scope(gs.size());  // the results' pieces out of one allocation (graph.h)
  {
    std::vector<Graph> br, gr;
    std::vector<size_t> bi, gi;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (symbolic_route(*gs[i].s->lazy, true) == ROUTE_BAND) { br.push_back(gs[i]); bi.push_back(i); }
      else { gr.push_back(gs[i]); gi.push_back(i); }
    }
    if (!br.empty()) {
      std::vector<Graph> outs(gs.size(), Graph(Graph::Empty{}));
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
    const size_t pbytes = nT * 16 + 8 * size_t(v.nb);  // ... | path length [nb] | exact tie on the best path [nb]
    DevMemP pm = rt.alloc(pbytes ? pbytes : 1);
    int* parc = pm->as<int>();
    int* pil = parc + nT;
    int* pol = pil + nT;
    float* pw = reinterpret_cast<float*>(pol + nT);
    int* plen = reinterpret_cast<int*>(pw + nT);
    HIP_CHECK(hipMemsetAsync(plen + v.nb, 0, 4 * size_t(v.nb), rt.stream()));  // (only the max-plus walk reports ties)
    // viterbiPath's ties go by the order the reference's queue visits the sources -- node order, where that is
    // provable for G (dense_ties_by_node_order); viterbiScore's gradient (LazySdOp::backward) keeps in-row order,
    // the reference's in-list order
    std::shared_ptr<Structure::DenseInfo> tdi = st.fixed.s->dense[v.chain_first ? 0 : 1];
    const bool by_node = st.maxplus && tdi && tdi->ties_by_node_order && !std::getenv("GTNX_NO_NODE_ORDER_TIES");
    {
      LazyGroup pv = v;
      pv.tie_by_node = by_node ? 1 : 0;
      launch_lazy_path(pv, parc, pil, pol, pw, plen, rt.stream());
    }
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
    // Exact ties between finite candidates ON the best path (reported by the max-plus walk): the reference breaks
    // them by its queue's order over the BUILT product (shortest.cpp:215-218).  Products small enough to build
    // (<= 2^22 arcs) are re-run on the built lattice with the queue-replaying schedule, as band_viterbi does;
    // larger ones (C4: 262 M arcs per utterance) keep the first maximum in in-row order -- counted
    // (gtnx_debug_viterbi_ties), INTEGRATION.md deviation 3.
    std::vector<uint8_t> rerun(gs.size(), 0);
    for (size_t i = 0; i < gs.size(); ++i) {
      if (slot[i].first != int(gi)) continue;
      const int b = slot[i].second;
      if (hlen[b] < 0 || !hlen[size_t(v.nb) + size_t(b)]) continue;
      g_viterbi_ties_seen.fetch_add(1);
      const LazyProduct& lp = *gs[i].s->lazy;
      // a transitions graph whose layers the reference's queue visits in node order: the walk broke the tie that
      // way already (dense_ties_by_node_order above)
      if (by_node) continue;
      const double arcs = double(lp.chain.s->M) * double(lp.fixed.s->A);
      if (arcs <= double(1 << 22) && !std::getenv("GTNX_NO_TIE_RERUN")) rerun[i] = 1;
      else g_viterbi_ties_unresolved.fetch_add(1);
    }
    // the path graphs (8 host arrays of T entries each per utterance): every element touches only its own
    // objects, so a large batch is built by a few threads (3.5 -> <1 ms of a 17 ms decode at C4)
    auto build = [&](size_t i) {
      if (slot[i].first != int(gi)) return;
      if (rerun[i]) return;
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
    std::vector<Graph> tg;
    std::vector<size_t> ti;
    for (size_t i = 0; i < gs.size(); ++i)
      if (rerun[i]) {
        realize(gs[i]);
        gs[i].s->resolve_sizes();
        gs[i].s->ensure_full();
        gs[i].s->ensure_host();
        gs[i].s->sched.reset();  // (the queue-replaying schedule: graph.cpp build_host_schedule)
        tg.push_back(gs[i]);
        ti.push_back(i);
      }
    if (!tg.empty()) {
      std::vector<Graph> to = op_viterbi_path(tg);
      for (size_t k = 0; k < ti.size(); ++k) outs[ti[k]] = std::move(to[k]);
    }
  }
  return outs;
}


} // namespace gtnx
