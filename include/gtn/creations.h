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
/** extension: B linear graphs over one device tensor [B][M][N]; weights copied (setWeights semantics,
 *  graph.cpp:179-181) unless `borrow`: then the graphs read the caller's tensor in place and the
 *  caller keeps it alive and unchanged while they are in use */
inline std::vector<Graph> linearGraphs(int B, int M, int N, const void* deviceWeights, bool calcGrad = true,
                                       bool borrow = false) {
  std::vector<gtnx_graph_t> h(B);
  if (B)
    detail::check(borrow ? gtnx_linear_graph_borrow_n(B, M, N, calcGrad, deviceWeights, h.data())
                         : gtnx_linear_graph_n(B, M, N, calcGrad, deviceWeights, h.data()));
  return detail::adopt(h);
}
} // namespace gtn
