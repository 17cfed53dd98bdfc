// gtn/creations.h -- reference gtn/creations.h:25,32
#pragma once
#include "gtn/functions.h"

namespace gtn {
inline Graph scalarGraph(float weight, bool calcGrad = true) {
  gtnx_graph_t h;
  detail::check(gtnx_scalar_graph(weight, calcGrad, &h));
  return Graph::fromHandle(h);
}
inline Graph linearGraph(int M, int N, bool calcGrad = true) {
  gtnx_graph_t h;
  detail::check(gtnx_linear_graph(M, N, calcGrad, &h));
  return Graph::fromHandle(h);
}
/** extension: B linear graphs over one device tensor [B][M][N] (weights copied) */
inline std::vector<Graph> linearGraphs(int B, int M, int N, const void* deviceWeights, bool calcGrad = true) {
  std::vector<gtnx_graph_t> h(B);
  if (B) detail::check(gtnx_linear_graph_n(B, M, N, calcGrad, deviceWeights, h.data()));
  return detail::adopt(h);
}
} // namespace gtn
