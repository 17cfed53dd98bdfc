/*
 * gtn_oracle.h -- CPU restatement of the gtn hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X engine.  It restates, in plain C,
 * the algorithms of the reference (facebookresearch/gtn):
 *
 *   graph storage + arcSort     gtn/graph.cpp:33-67, 162-177
 *   linearGraph                 gtn/creations.cpp:20-33
 *   shortestDistance (fwd)      gtn/functions/shortest.cpp:86-188
 *   shortestDistanceGrad (bwd)  gtn/functions/shortest.cpp:33-82
 *   shortestPath                gtn/functions/shortest.cpp:190-272
 *   compose / intersect         gtn/functions/compose.cpp:64-104, 108-208, 377-522
 *     with the three matchers   gtn/functions/compose.cpp:211-374
 *     and matcher dispatch      gtn/functions.cpp:225-251
 *   compose gradient            gtn/functions/compose.cpp:496-518
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (gtn_amd) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 * the known-answer values of the reference's own tests (test/functions_test.cpp,
 * test/autograd_test.cpp, test/criterion_test.cpp) and against fixtures in
 * tests/golden/ produced by running the real reference (oracle/_ref) in the
 * build container (tests/golden/make_golden.py).
 */
#ifndef GTN_ORACLE_H
#define GTN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_EPSILON (-1)

typedef struct og_graph og_graph;

/* ---- graph storage (gtn/graph.h:56-465) ---- */
og_graph* og_new(void);
void og_free(og_graph* g);
int og_add_node(og_graph* g, int start, int accept);
int og_add_arc(og_graph* g, int src, int dst, int ilabel, int olabel, float w);
int og_num_nodes(const og_graph* g);
int og_num_arcs(const og_graph* g);
int og_num_start(const og_graph* g);
int og_num_accept(const og_graph* g);
/* copy-out getters; any pointer may be NULL */
void og_get_nodes(const og_graph* g, uint8_t* start, uint8_t* accept);
void og_get_arcs(const og_graph* g, int* src, int* dst, int* il, int* ol, float* w);
void og_set_weights(og_graph* g, const float* w);
/* arcSort: per-node in/out lists sorted by ilabel (olabel != 0: by olabel).
 * The reference uses std::sort (unstable); this oracle uses a stable sort, so
 * the order among equal labels may differ -- equal() ignores arc order. */
void og_arc_sort(og_graph* g, int olabel);
void og_mark_sorted(og_graph* g, int olabel);
int og_is_sorted(const og_graph* g, int olabel);
og_graph* og_linear_graph(int M, int N);

/* ---- shortest distance (log / tropical) ---- */
/* returns 0, or 1 for "Graph has a cycle, self-loop or is disconnected!".
 * node_scores: N floats; max_cache: N+1 floats; argmax_cache: N+1 int64
 * (-1 = none; last slot holds a NODE id) -- any may be NULL. */
int og_shortest_distance(og_graph* g, int tropical, float* out_score,
                         float* node_scores, float* max_cache,
                         int64_t* argmax_cache);
/* gradient wrt arc weights (arc_grads: A floats) for an upstream delta */
int og_shortest_distance_grad(og_graph* g, int tropical, float delta,
                              float* arc_grads);
/* best path; returns 0/1 like above. out_arcs (capacity N) receives the path's
 * arc ids first-arc-first; *n_arcs its length; *has_node = 1 if the output
 * graph has at least one node. */
int og_shortest_path(og_graph* g, int* out_arcs, int* n_arcs, int* has_node);

/* ---- composition ---- */
/* mode 0 = compose() dispatch (g1.olabelSorted / g2.ilabelSorted),
 * mode 1 = intersect() dispatch (either flag on each side).
 * grad_info (optional, caller-sized via og_num_arcs(result)*2) gets (i,j) pairs */
og_graph* og_compose(og_graph* g1, og_graph* g2, int mode);
const int* og_grad_info(const og_graph* composed); /* 2*A ints or NULL */
/* compose gradFunc: deltas over the composed graph's arcs -> grad1, grad2 */
void og_compose_grad(const og_graph* composed, const float* deltas, int A1,
                     int A2, float* grad1, float* grad2);

/* ---- CTC loss (benchmarks/ctc.cpp:40-58,150-160) for the cpu_baseline leg ---- */
og_graph* og_ctc_graph(const int* target, int U, int blank, int arc_sort);
/* loss = forwardScore(emissions) - forwardScore(intersect(ctc, emissions));
 * grad (T*C floats) = d loss / d emissions.  returns 0 on success. */
int og_ctc_loss(const float* emissions, int T, int C, const int* target, int U,
                float* loss, float* grad);

#ifdef __cplusplus
}
#endif
#endif
