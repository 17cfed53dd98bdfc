/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Exposes the *unmodified reference implementation* (compiled from the sources
 * where they lie under /root/reference by oracle/Makefile, output only into
 * oracle/_ref/) through the same C ABI as the product (include/gtn_amd.h), so
 * the parity tests and tests/golden/make_golden.py can drive both with the same
 * Python code.  Nothing here re-implements an algorithm: every entry point
 * forwards to the reference's public C++ API (gtn/gtn.h).
 *
 * Also exports ref_ctc_batch(): the reference's batched CTC benchmark pattern
 * (benchmarks/ctc.cpp:136-168, parallelMap(fwd) then parallelMap(bwd)) on
 * caller-supplied inputs -- the "reference" cpu_baseline of bench.py.
 */
#include <chrono>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gtn/gtn.h"
#include "gtn_amd.h"

using gtn::Graph;

namespace {
thread_local std::string g_err;

gtnx_status_t fail(gtnx_status_t s, const std::string& m) {
  g_err = m;
  return s;
}

template <class F>
gtnx_status_t guard(F&& f) {
  try {
    f();
    return GTNX_OK;
  } catch (const std::invalid_argument& e) {
    return fail(GTNX_INVALID_ARGUMENT, e.what());
  } catch (const std::out_of_range& e) {
    return fail(GTNX_OUT_OF_RANGE, e.what());
  } catch (const std::logic_error& e) {
    return fail(GTNX_LOGIC_ERROR, e.what());
  } catch (const std::exception& e) {
    return fail(GTNX_RUNTIME_ERROR, e.what());
  }
}

inline Graph& G(gtnx_graph_t h) { return *reinterpret_cast<Graph*>(h); }
inline gtnx_graph_t H(Graph&& g) {
  return reinterpret_cast<gtnx_graph_t>(new Graph(std::move(g)));
}
inline gtnx_graph_t H(const Graph& g) {
  return reinterpret_cast<gtnx_graph_t>(new Graph(g));
}

std::vector<Graph> vec(const gtnx_graph_t* a, int n) {
  std::vector<Graph> v;
  v.reserve(n);
  for (int i = 0; i < n; ++i) v.push_back(G(a[i]));
  return v;
}

void check_node(Graph& g, int n) {
  if (n < 0 || (size_t)n >= g.numNodes()) throw std::out_of_range("node index");
}
void check_arc(Graph& g, int a) {
  if (a < 0 || (size_t)a >= g.numArcs()) throw std::out_of_range("arc index");
}
} // namespace

extern "C" {

const char* gtnx_last_error(void) { return g_err.c_str(); }
const char* gtnx_version(void) { return "reference"; }
const char* gtnx_backend(void) { return "reference-cpu"; }
int gtnx_device_count(void) { return 0; }
gtnx_status_t gtnx_set_device(int) { return GTNX_OK; }
gtnx_status_t gtnx_get_device(int* d) {
  *d = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_set_stream(void*) { return GTNX_OK; }
gtnx_status_t gtnx_compose_mode(int, int* previous) {
  if (previous) *previous = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_synchronize(void) { return GTNX_OK; }
gtnx_status_t gtnx_memory_stats(uint64_t* r, uint64_t* u) {
  if (r) *r = 0;
  if (u) *u = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_empty_cache(void) { return GTNX_OK; }
gtnx_status_t gtnx_reclaim(void) { return GTNX_OK; }
gtnx_status_t gtnx_clone(gtnx_graph_t g, int projection, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::clone(G(g), static_cast<gtn::Projection>(projection))); });
}
gtnx_status_t gtnx_concat(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::concat(vec(g, n))); });
}
gtnx_status_t gtnx_closure(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::closure(G(g))); });
}
gtnx_status_t gtnx_union(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::union_(vec(g, n))); });
}
gtnx_status_t gtnx_parallel_enter(void) { return GTNX_OK; }
gtnx_status_t gtnx_parallel_leave(void) { return GTNX_OK; }
gtnx_status_t gtnx_parallel_flush(void) { return GTNX_OK; }

gtnx_status_t gtnx_graph_create(int calc_grad, gtnx_graph_t* out) {
  return guard([&] { *out = H(Graph(calc_grad != 0)); });
}
gtnx_status_t gtnx_graph_copy(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(G(g)); });
}
gtnx_status_t gtnx_graph_deep_copy(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(Graph::deepCopy(G(g))); });
}
gtnx_status_t gtnx_graph_destroy(gtnx_graph_t g) {
  delete reinterpret_cast<Graph*>(g);
  return GTNX_OK;
}
gtnx_status_t gtnx_graph_add_node(gtnx_graph_t g, int s, int a, int* id) {
  return guard([&] {
    int i = G(g).addNode(s != 0, a != 0);
    if (id) *id = i;
  });
}
gtnx_status_t gtnx_graph_add_arc(gtnx_graph_t g, int src, int dst, int il,
                                 int ol, float w, int* id) {
  return guard([&] {
    check_node(G(g), src);
    check_node(G(g), dst);
    int i = (int)G(g).addArc(src, dst, il, ol, w);
    if (id) *id = i;
  });
}
gtnx_status_t gtnx_graph_add_nodes(gtnx_graph_t g, int n, const uint8_t* s,
                                   const uint8_t* a) {
  return guard([&] {
    for (int i = 0; i < n; ++i) G(g).addNode(s && s[i], a && a[i]);
  });
}
gtnx_status_t gtnx_graph_add_arcs(gtnx_graph_t g, int n, const int* src,
                                  const int* dst, const int* il, const int* ol,
                                  const float* w) {
  return guard([&] {
    for (int i = 0; i < n; ++i) {
      check_node(G(g), src[i]);
      check_node(G(g), dst[i]);
      G(g).addArc(src[i], dst[i], il[i], ol[i], w ? w[i] : 0.0f);
    }
  });
}
#define COUNT_FN(name, expr)                                   \
  gtnx_status_t name(gtnx_graph_t g, int64_t* out) {           \
    return guard([&] { *out = (int64_t)(expr); });             \
  }
COUNT_FN(gtnx_graph_num_nodes, G(g).numNodes())
COUNT_FN(gtnx_graph_num_arcs, G(g).numArcs())
COUNT_FN(gtnx_graph_num_start, G(g).numStart())
COUNT_FN(gtnx_graph_num_accept, G(g).numAccept())
COUNT_FN(gtnx_graph_num_inputs, G(g).inputs().size())

gtnx_status_t gtnx_graph_item(gtnx_graph_t g, float* out) {
  return guard([&] { *out = G(g).item(); });
}
gtnx_status_t gtnx_graph_arc_sort(gtnx_graph_t g, int ol) {
  return guard([&] { G(g).arcSort(ol != 0); });
}
gtnx_status_t gtnx_graph_mark_arc_sorted(gtnx_graph_t g, int ol) {
  return guard([&] { G(g).markArcSorted(ol != 0); });
}
gtnx_status_t gtnx_graph_ilabel_sorted(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).ilabelSorted(); });
}
gtnx_status_t gtnx_graph_olabel_sorted(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).olabelSorted(); });
}
gtnx_status_t gtnx_graph_weights(gtnx_graph_t g, int, float** out) {
  return guard([&] { *out = G(g).weights(); });
}
gtnx_status_t gtnx_graph_get_weights(gtnx_graph_t g, float* out) {
  return guard([&] {
    std::memcpy(out, G(g).weights(), sizeof(float) * G(g).numArcs());
  });
}
gtnx_status_t gtnx_graph_set_weights(gtnx_graph_t g, const float* w) {
  return guard([&] { G(g).setWeights(w); });
}
gtnx_status_t gtnx_graph_set_weights_device(gtnx_graph_t, const void*) {
  return fail(GTNX_DEVICE_ERROR, "reference-cpu backend has no device");
}
gtnx_status_t gtnx_graph_weights_device(gtnx_graph_t, void**) {
  return fail(GTNX_DEVICE_ERROR, "reference-cpu backend has no device");
}
gtnx_status_t gtnx_graph_labels_to_array(gtnx_graph_t g, int* out, int il) {
  return guard([&] { G(g).labelsToArray(out, il != 0); });
}
gtnx_status_t gtnx_graph_get_start(gtnx_graph_t g, int* out) {
  return guard([&] {
    auto& v = G(g).start();
    std::memcpy(out, v.data(), sizeof(int) * v.size());
  });
}
gtnx_status_t gtnx_graph_get_accept(gtnx_graph_t g, int* out) {
  return guard([&] {
    auto& v = G(g).accept();
    std::memcpy(out, v.data(), sizeof(int) * v.size());
  });
}
gtnx_status_t gtnx_graph_is_start(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = G(g).isStart(n);
  });
}
gtnx_status_t gtnx_graph_is_accept(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = G(g).isAccept(n);
  });
}
gtnx_status_t gtnx_graph_make_accept(gtnx_graph_t g, int n) {
  return guard([&] {
    check_node(G(g), n);
    G(g).makeAccept(n);
  });
}
gtnx_status_t gtnx_graph_num_out(gtnx_graph_t g, int n, int64_t* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = (int64_t)G(g).numOut(n);
  });
}
gtnx_status_t gtnx_graph_num_in(gtnx_graph_t g, int n, int64_t* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = (int64_t)G(g).numIn(n);
  });
}
gtnx_status_t gtnx_graph_get_out(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    auto& v = G(g).out(n);
    std::memcpy(out, v.data(), sizeof(int) * v.size());
  });
}
gtnx_status_t gtnx_graph_get_in(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    auto& v = G(g).in(n);
    std::memcpy(out, v.data(), sizeof(int) * v.size());
  });
}
gtnx_status_t gtnx_graph_get_arcs(gtnx_graph_t g, int* src, int* dst, int* il,
                                  int* ol) {
  return guard([&] {
    Graph& gr = G(g);
    for (size_t a = 0; a < gr.numArcs(); ++a) {
      if (src) src[a] = gr.srcNode(a);
      if (dst) dst[a] = gr.dstNode(a);
      if (il) il[a] = gr.ilabel(a);
      if (ol) ol[a] = gr.olabel(a);
    }
  });
}
gtnx_status_t gtnx_graph_get_arc(gtnx_graph_t g, int a, int* src, int* dst,
                                 int* il, int* ol, float* w) {
  return guard([&] {
    Graph& gr = G(g);
    check_arc(gr, a);
    if (src) *src = gr.srcNode(a);
    if (dst) *dst = gr.dstNode(a);
    if (il) *il = gr.ilabel(a);
    if (ol) *ol = gr.olabel(a);
    if (w) *w = gr.weight(a);
  });
}
gtnx_status_t gtnx_graph_set_weight(gtnx_graph_t g, int a, float w) {
  return guard([&] {
    check_arc(G(g), a);
    G(g).setWeight(a, w);
  });
}
gtnx_status_t gtnx_graph_calc_grad(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).calcGrad(); });
}
gtnx_status_t gtnx_graph_set_calc_grad(gtnx_graph_t g, int c) {
  return guard([&] { G(g).setCalcGrad(c != 0); });
}
gtnx_status_t gtnx_graph_is_grad_available(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).isGradAvailable(); });
}
gtnx_status_t gtnx_graph_grad(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(G(g).grad()); });
}
gtnx_status_t gtnx_graph_zero_grad(gtnx_graph_t g) {
  return guard([&] { G(g).zeroGrad(); });
}
gtnx_status_t gtnx_graph_add_grad(gtnx_graph_t g, const float* v, int64_t n) {
  return guard([&] { G(g).addGrad(std::vector<float>(v, v + n)); });
}
gtnx_status_t gtnx_graph_add_grad_graph(gtnx_graph_t g, gtnx_graph_t o) {
  return guard([&] { G(g).addGrad(G(o)); });
}
gtnx_status_t gtnx_graph_id(gtnx_graph_t g, uintptr_t* out) {
  return guard([&] { *out = G(g).id(); });
}
gtnx_status_t gtnx_graph_create_op(gtnx_graph_t* inputs, int n,
                                   gtnx_grad_fn fn, void* ctx,
                                   void (*ctx_free)(void*), gtnx_graph_t* out) {
  return guard([&] {
    std::shared_ptr<void> holder(ctx, [ctx_free](void* p) {
      if (ctx_free) ctx_free(p);
    });
    Graph::GradFunc gf = nullptr;
    if (fn) {
      gf = [fn, holder](std::vector<Graph>& ins, Graph& deltas) {
        std::vector<gtnx_graph_t> hs;
        for (auto& i : ins) hs.push_back(reinterpret_cast<gtnx_graph_t>(&i));
        gtnx_status_t s = fn(holder.get(), hs.data(), (int)hs.size(),
                             reinterpret_cast<gtnx_graph_t>(&deltas));
        if (s != GTNX_OK) throw std::runtime_error("grad_fn failed");
      };
    }
    *out = H(Graph(gf, vec(inputs, n)));
  });
}

gtnx_status_t gtnx_graph_get_input(gtnx_graph_t g, int i, gtnx_graph_t* out) {
  return guard([&] { *out = H(G(g).inputs().at(i)); });
}
gtnx_status_t gtnx_graph_set_inputs(gtnx_graph_t g, const gtnx_graph_t* inputs, int n) {
  return guard([&] { G(g).setInputs(vec(inputs, n)); });
}
gtnx_status_t gtnx_graph_set_grad_fn(gtnx_graph_t g, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*)) {
  return guard([&] {
    std::shared_ptr<void> holder(ctx, [ctx_free](void* p) { if (ctx_free) ctx_free(p); });
    Graph::GradFunc gf = nullptr;
    if (fn) gf = [fn, holder](std::vector<Graph>& ins, Graph& deltas) {
      std::vector<gtnx_graph_t> hs;
      for (auto& i : ins) hs.push_back(reinterpret_cast<gtnx_graph_t>(&i));
      if (fn(holder.get(), hs.data(), (int)hs.size(), reinterpret_cast<gtnx_graph_t>(&deltas)) != GTNX_OK)
        throw std::runtime_error("grad_fn failed");
    };
    G(g).setGradFunc(gf);
  });
}
gtnx_status_t gtnx_graph_has_grad_fn(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).gradFunc() != nullptr; });
}

gtnx_status_t gtnx_scalar_graph(float v, int cg, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::scalarGraph(v, cg != 0)); });
}
gtnx_status_t gtnx_linear_graph(int M, int N, int cg, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::linearGraph(M, N, cg != 0)); });
}
gtnx_status_t gtnx_linear_graph_n(int, int, int, int, const void*,
                                  gtnx_graph_t*) {
  return fail(GTNX_DEVICE_ERROR, "reference-cpu backend has no device");
}

#define UNARY(name, fn)                                         \
  gtnx_status_t name(gtnx_graph_t g, gtnx_graph_t* out) {       \
    return guard([&] { *out = H(fn(G(g))); });                  \
  }
#define BINARY(name, fn)                                                  \
  gtnx_status_t name(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out) { \
    return guard([&] { *out = H(fn(G(a), G(b))); });                      \
  }
UNARY(gtnx_negate, gtn::negate)
BINARY(gtnx_add, gtn::add)
BINARY(gtnx_subtract, gtn::subtract)
BINARY(gtnx_compose, gtn::compose)
BINARY(gtnx_intersect, gtn::intersect)
UNARY(gtnx_forward_score, gtn::forwardScore)
UNARY(gtnx_viterbi_score, gtn::viterbiScore)
UNARY(gtnx_viterbi_path, gtn::viterbiPath)

/* batched forms == the binding's vector overloads (parallelMap) */
#define UNARY_N(name, fn)                                                  \
  gtnx_status_t name(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {    \
    return guard([&] {                                                     \
      auto v = vec(g, n);                                                  \
      auto f = [](const Graph& x) { return fn(x); };                       \
      auto r = gtn::parallelMap(f, v);                                     \
      for (int i = 0; i < n; ++i) out[i] = H(r[i]);                        \
    });                                                                    \
  }
#define BINARY_N(name, fn)                                                 \
  gtnx_status_t name(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, \
                     int nb, gtnx_graph_t* out) {                          \
    return guard([&] {                                                     \
      auto va = vec(a, na);                                                \
      auto vb = vec(b, nb);                                                \
      auto f = [](const Graph& x, const Graph& y) { return fn(x, y); };    \
      auto r = gtn::parallelMap(f, va, vb);                                \
      for (size_t i = 0; i < r.size(); ++i) out[i] = H(r[i]);              \
    });                                                                    \
  }
UNARY_N(gtnx_negate_n, gtn::negate)
BINARY_N(gtnx_add_n, gtn::add)
BINARY_N(gtnx_subtract_n, gtn::subtract)
BINARY_N(gtnx_compose_n, gtn::compose)
BINARY_N(gtnx_intersect_n, gtn::intersect)
UNARY_N(gtnx_forward_score_n, gtn::forwardScore)
UNARY_N(gtnx_viterbi_score_n, gtn::viterbiScore)
UNARY_N(gtnx_viterbi_path_n, gtn::viterbiPath)

gtnx_status_t gtnx_items_n(const gtnx_graph_t* g, int n, float* out) {
  return guard([&] {
    for (int i = 0; i < n; ++i) out[i] = G(g[i]).item();
  });
}
gtnx_status_t gtnx_items_device_n(const gtnx_graph_t*, int, void*) {
  return fail(GTNX_DEVICE_ERROR, "reference-cpu backend has no device");
}
gtnx_status_t gtnx_grads_device_n(const gtnx_graph_t*, int, void*,
                                  const int64_t*) {
  return fail(GTNX_DEVICE_ERROR, "reference-cpu backend has no device");
}

gtnx_status_t gtnx_backward(gtnx_graph_t g, int retain) {
  return guard([&] { gtn::backward(G(g), retain != 0); });
}
gtnx_status_t gtnx_backward_with_grad(gtnx_graph_t g, gtnx_graph_t grad,
                                      int retain) {
  return guard([&] { gtn::backward(G(g), G(grad), retain != 0); });
}
gtnx_status_t gtnx_backward_n(const gtnx_graph_t* g, int n, int retain) {
  return guard([&] {
    bool r = retain != 0;
    auto bwd = [r](const Graph& x) { gtn::backward(x, r); };
    auto v = vec(g, n);
    gtn::parallelMap(bwd, v);
  });
}

gtnx_status_t gtnx_equal(gtnx_graph_t a, gtnx_graph_t b, int* out) {
  return guard([&] { *out = gtn::equal(G(a), G(b)); });
}
gtnx_status_t gtnx_isomorphic(gtnx_graph_t a, gtnx_graph_t b, int* out) {
  return guard([&] { *out = gtn::isomorphic(G(a), G(b)); });
}

gtnx_status_t gtnx_remove(gtnx_graph_t g, int ilabel, int olabel, gtnx_graph_t* out) {
  return guard([&] { *out = H(gtn::remove(G(g), ilabel, olabel)); });
}
/* the reference's own load() (gtn/utils.cpp:185-225) over the file image */
gtnx_status_t gtnx_graph_load_buffer(const void* data, size_t bytes, gtnx_graph_t* out) {
  return guard([&] {
    std::istringstream in(std::string(static_cast<const char*>(data), bytes), std::ios::binary);
    *out = H(gtn::load(in));
  });
}

gtnx_status_t gtnx_prof_enable(int) { return GTNX_OK; }
gtnx_status_t gtnx_prof_reset(void) { return GTNX_OK; }
gtnx_status_t gtnx_prof_get(const char*, double* ms, int64_t* n, double* b) {
  if (ms) *ms = 0;
  if (n) *n = 0;
  if (b) *b = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_debug_symbolic_route(gtnx_graph_t, int, int* route) {
  *route = -1;  /* the reference builds every product */
  return GTNX_OK;
}
gtnx_status_t gtnx_debug_tie_ranks(gtnx_graph_t, int*, int*, int* applies) {
  if (applies) *applies = 0;  /* the reference decides ties on its lattices */
  return GTNX_OK;
}
gtnx_status_t gtnx_debug_viterbi_ties(int64_t* seen, int64_t* unresolved) {
  if (seen) *seen = 0;
  if (unresolved) *unresolved = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_debug_route_name(int, char* buf, size_t cap) {
  if (cap) buf[0] = 0;
  return GTNX_OK;
}
gtnx_status_t gtnx_prof_names(char* buf, size_t cap) {
  if (cap) buf[0] = 0;
  return GTNX_OK;
}

/* ---- the reference CPU baseline: benchmarks/ctc.cpp:136-168 on given inputs.
 * emissions: B*T*C floats, targets: B*U ints in [1, C-1], blank = 0.
 * Runs `iters` timed iterations of parallelMap(fwd) + parallelMap(bwd) after
 * `warmup` untimed ones; returns seconds per iteration, fills losses (B floats)
 * and, if non-NULL, grads (B*T*C) from the LAST iteration, and *threads with
 * the pool size parallelMap used (parallel_map.cpp:18-26). */
double ref_ctc_batch(const float* emissions, const int* targets, int B, int T,
                     int C, int U, int warmup, int iters, float* losses,
                     float* grads, int* threads) {
  std::vector<std::vector<int>> tg(B);
  std::vector<const float*> em(B);
  for (int b = 0; b < B; ++b) {
    tg[b].assign(targets + (size_t)b * U, targets + (size_t)(b + 1) * U);
    em[b] = emissions + (size_t)b * T * C;
  }
  auto ctcGraph = [](const std::vector<int>& target) {
    int blank = 0;
    size_t L = 2 * target.size() + 1;
    Graph ctc;
    for (size_t l = 0; l < L; l++) {
      size_t idx = (l - 1) / 2;
      ctc.addNode(l == 0, l == L - 1 || l == L - 2);
      int label = l % 2 ? target[idx] : blank;
      ctc.addArc(l, l, label);
      if (l > 0) ctc.addArc(l - 1, l, label);
      if (l % 2 && l > 1 && label != target[idx - 1]) ctc.addArc(l - 2, l, label);
    }
    ctc.arcSort();
    return ctc;
  };
  std::vector<Graph> ems(B);
  /* parallelMap copies every input vector into every task (thread_pool.h
   * enqueue -> std::bind), so hand it only an index vector and capture the
   * batch by reference: the cheapest legal use of the reference's API. */
  std::vector<int> idx(B);
  for (int b = 0; b < B; ++b) idx[b] = b;
  auto fwd = [&](int b) {
    auto ctc = ctcGraph(tg[b]);
    auto emissions = gtn::linearGraph(T, C);
    emissions.setWeights(em[b]);
    ems[b] = emissions;
    return gtn::subtract(gtn::forwardScore(emissions),
                         gtn::forwardScore(gtn::intersect(ctc, emissions)));
  };
  auto bwd = [](const Graph& g) { gtn::backward(g); };
  if (threads)
    *threads = (int)std::min<size_t>((size_t)B, std::thread::hardware_concurrency());
  double secs = 0;
  for (int it = 0; it < warmup + iters; ++it) {
    auto t0 = std::chrono::steady_clock::now();
    auto lossGraphs = gtn::parallelMap(fwd, idx);
    gtn::parallelMap(bwd, lossGraphs);
    auto t1 = std::chrono::steady_clock::now();
    if (it >= warmup) secs += std::chrono::duration<double>(t1 - t0).count();
    if (it == warmup + iters - 1) {
      for (int b = 0; b < B; ++b) {
        if (losses) losses[b] = lossGraphs[b].item();
        if (grads)
          std::memcpy(grads + (size_t)b * T * C, ems[b].grad().weights(),
                      sizeof(float) * (size_t)T * C);
      }
    }
  }
  return secs / (iters > 0 ? iters : 1);
}

/* ---- the same per-utterance computation (benchmarks/ctc.cpp:40-58,150-160) on RAGGED targets, with the gradient
 * of the target graph as well: labels back to back, lengths[b] labels each (0 allowed); losses [B]; grads [B*T*C] or
 * NULL; tgrads or NULL: the target graph's arc gradients (arc ids in benchmarks/ctc.cpp:40-58's addArc order) of
 * utterance b at tgrads + toff[b]; arcs[b] (or NULL) receives the graph's arc count.  Returns the pool size used. */
int ref_ctc_ragged(const float* emissions, const int* labels, const int* lengths, int B, int T, int C, int blank,
                   float* losses, float* grads, float* tgrads, const int64_t* toff, int* arcs) {
  std::vector<std::vector<int>> tg(B);
  size_t at = 0;
  for (int b = 0; b < B; ++b) {
    tg[b].assign(labels + at, labels + at + lengths[b]);
    at += (size_t)lengths[b];
  }
  std::vector<int> idx(B);
  for (int b = 0; b < B; ++b) idx[b] = b;
  std::vector<Graph> ems(B), ctcs(B);
  auto fwd = [&](int b) {
    const std::vector<int>& target = tg[b];
    size_t L = 2 * target.size() + 1;
    Graph ctc;
    for (size_t l = 0; l < L; l++) {
      size_t i = (l - 1) / 2;
      ctc.addNode(l == 0, l == L - 1 || l == L - 2);
      int label = l % 2 ? target[i] : blank;
      ctc.addArc(l, l, label);
      if (l > 0) ctc.addArc(l - 1, l, label);
      if (l % 2 && l > 1 && label != target[i - 1]) ctc.addArc(l - 2, l, label);
    }
    ctc.arcSort();
    auto e = gtn::linearGraph(T, C);
    e.setWeights(emissions + (size_t)b * T * C);
    ems[b] = e;
    ctcs[b] = ctc;
    return gtn::subtract(gtn::forwardScore(e), gtn::forwardScore(gtn::intersect(ctc, e)));
  };
  auto bwd = [](const Graph& g) { gtn::backward(g); };
  auto lossGraphs = gtn::parallelMap(fwd, idx);
  gtn::parallelMap(bwd, lossGraphs);
  for (int b = 0; b < B; ++b) {
    if (losses) losses[b] = lossGraphs[b].item();
    if (grads) std::memcpy(grads + (size_t)b * T * C, ems[b].grad().weights(), sizeof(float) * (size_t)T * C);
    if (arcs) arcs[b] = (int)ctcs[b].numArcs();
    if (tgrads) std::memcpy(tgrads + toff[b], ctcs[b].grad().weights(), sizeof(float) * ctcs[b].numArcs());
  }
  return (int)std::min<size_t>((size_t)B, std::thread::hardware_concurrency());
}

} // extern "C"
