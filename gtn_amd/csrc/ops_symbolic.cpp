// ops_symbolic.cpp -- which kernels score a symbolic chain product: THE ROUTE TABLE.
//
// compose(chain, G) / compose(G, chain) of an implicit linear chain and an epsilon-free G may stay symbolic
// (ops_compose.cpp); forwardScore / viterbiScore / viterbiPath of such a product then run one of six kernel
// families.  Which one is decided here and nowhere else, per product, in this order:
//
//   route        kernels          eligibility (first match wins)                                   predicate
//   BAND         band.hip         G is banded: every arc n -> n + d with 0 <= d <= 2, one label    band_ok()           (ops_band.cpp)
//                                 per node's in-arcs, <= band_max_nodes() nodes, band_min_labels()
//                                 <= C <= band_max_labels() (CTC targets, force-alignment graphs)
//   PAIR         lazy_pair.hip    log semiring only; G small (<= lazy_pair_max_nodes() nodes,      lazy_pair_ok()      (ops_lazy.cpp)
//                                 degree <= lazy_pair_max_degree()), C within the block's LDS
//   DENSE_MFMA   lazy.hip         log semiring; 8 <= N <= 1024, one label per node's in-arcs,       lazy_dense_ok()     (ops_lazy.cpp)
//                                 G at least half complete: probability domain on the matrix cores
//   DENSE        lazy.hip         the same with GTNX_DENSE_VALU set (VALU form, diagnostics)
//   MAXPLUS      maxplus.hip      tropical semiring; the dense predicate, T >= 1, column labels     lazy_dense_ok()
//   WALK         lazy.hip         everything else: record-walking time-step kernels
//
// Environment switches that remove a route (tests flip them at run time, so they are read per call):
// GTNX_NO_BAND, GTNX_NO_LAZY_PAIRS (log semiring: also skips BAND, as it always did), GTNX_NO_DENSE.
// gtnx_debug_symbolic_route() (include/gtn_amd.h) exposes the decision; tests/test_lazy_gpu.py enumerates it.
#include "ops_internal.h"

namespace gtnx {

const char* symbolic_route_name(int r) {
  static const char* names[ROUTE_COUNT] = {"band", "pair", "dense_mfma", "dense", "maxplus", "walk"};
  return r >= 0 && r < ROUTE_COUNT ? names[r] : "?";
}

SymbolicRoute symbolic_route(const LazyProduct& lp, bool tropical) {
  const bool band = getenv("GTNX_NO_BAND") == nullptr;
  if (!tropical) {
    if (!getenv("GTNX_NO_LAZY_PAIRS")) {
      if (band && band_ok(lp)) return ROUTE_BAND;
      if (lazy_pair_ok(lp)) return ROUTE_PAIR;
    }
    return lazy_group_route(lp, false);
  }
  if (band && band_ok(lp)) return ROUTE_BAND;  // one launch per batch, back-pointers and all
  return lazy_group_route(lp, true);
}

std::vector<Graph> lazy_shortest_distance(std::vector<Graph>& gs, bool tropical) {
  // members by route family: banded / per-pair / one-G-per-group kernels
  std::vector<Graph> part[3];
  std::vector<size_t> idx[3];
  for (size_t i = 0; i < gs.size(); ++i) {
    const SymbolicRoute r = symbolic_route(*gs[i].s->lazy, tropical);
    const int k = r == ROUTE_BAND ? 0 : (r == ROUTE_PAIR ? 1 : 2);
    part[k].push_back(gs[i]);
    idx[k].push_back(i);
  }
  auto run = [&](int k) {
    if (k == 0) return tropical ? band_viterbi(part[0], false) : band_forward_score(part[0]);
    if (k == 1) return lazy_pair_forward_score(part[1]);
    return lazy_group_shortest_distance(part[2], tropical);
  };
  for (int k = 0; k < 3; ++k)
    if (part[k].size() == gs.size()) return run(k);
  std::vector<Graph> outs(gs.size(), Graph(Graph::Empty{}));
  for (int k = 0; k < 3; ++k) {
    if (part[k].empty()) continue;
    std::vector<Graph> r = run(k);
    for (size_t j = 0; j < idx[k].size(); ++j) outs[idx[k][j]] = std::move(r[j]);
  }
  return outs;
}

} // namespace gtnx
