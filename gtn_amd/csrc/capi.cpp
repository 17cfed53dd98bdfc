// capi.cpp -- the extern "C" boundary declared in include/gtn_amd.h.
// Thin: argument checks, handle <-> Graph, exception -> status mapping.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "batch.h"
#include "ops.h"
#include "ops_internal.h"
#include "region.h"

#define GTNX_API extern "C" __attribute__((visibility("default")))

using namespace gtnx;

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
  } catch (const Error& e) {
    return fail(e.status, e.what());
  } catch (const std::bad_alloc&) {
    return fail(GTNX_RUNTIME_ERROR, "out of host memory");
  } catch (const std::exception& e) {
    return fail(GTNX_RUNTIME_ERROR, e.what());
  }
}

// GL: the handle as it is (a composition may still be symbolic, ops.cpp "lazy chain
// products"); G: for everything that looks inside the graph -- builds it first
// RAW: the handle's own pieces, whatever they are (a placeholder of a parallelMap region included: region.h);
// GL: the graph behind the handle -- a placeholder's call runs now if it has not yet
inline Graph& RAW(gtnx_graph_t h) {
  if (!h) throw_invalid("null graph handle");
  return *reinterpret_cast<Graph*>(h);
}
inline Graph& GL(gtnx_graph_t h) {
  Graph& g = RAW(h);
  if (g.s->pending) return region_value(g);
  // a graph lives where it was made: its buffers are in that device's memory and its kernels run on that device's
  // stream (runtime.h) -- using it from a thread that is on another device is the caller's mistake, said so
  if (g.s->device >= 0 && g.s->device != Runtime::current_device()) {
    // (a graph that has nothing on its device yet -- built on the host, not uploaded -- simply moves)
    const bool host_only = !g.s->dev_valid && !g.s->lazy && !g.s->deferred && (!g.w || (!g.w->dev_valid && !g.w->staged)) &&
                           (!g.g || !g.is_grad_available());
    if (host_only) {
      g.s->device = Runtime::current_device();
      return g;
    }
    throw_invalid("[gtn_amd] this graph lives on device " + std::to_string(g.s->device) + ", the calling thread is on device " +
                  std::to_string(Runtime::current_device()) + " (gtnx_set_device)");
  }
  return g;
}
inline Graph& G(gtnx_graph_t h) {
  Graph& g = GL(h);
  if (g.s->lazy) realize(g);
  if (g.s->deferred) g.s->resolve_sizes();  // sizes a compose left on the device
  return g;
}
// Handles.  A gtnx_graph_t points at a Graph inside a SLAB of handles: the n results of a vector form (gtnx_*_n) are one
// allocation, not n -- a step of 512 utterances through the vector forms made and freed 2500 48-byte blocks on the
// thread everything waits for -- and a single result is a slab of one.  A handle's Graph is destroyed when the handle is
// (gtnx_graph_destroy); the slab's memory goes when its last handle has.
struct HandleSlab {
  std::atomic<int> live;
};
struct HandleEntry {
  Graph g;  // (first: the handle IS a Graph*)
  HandleSlab* slab;
};
HandleEntry* new_handles(size_t n) {
  const size_t head = (sizeof(HandleSlab) + alignof(HandleEntry) - 1) / alignof(HandleEntry) * alignof(HandleEntry);
  char* raw = static_cast<char*>(::operator new(head + sizeof(HandleEntry) * (n ? n : 1)));
  HandleSlab* s = new (raw) HandleSlab();
  s->live.store(int(n), std::memory_order_relaxed);
  HandleEntry* e = reinterpret_cast<HandleEntry*>(raw + head);
  for (size_t i = 0; i < n; ++i) {
    new (&e[i].g) Graph(Graph::Empty{});
    e[i].slab = s;
  }
  return e;
}
inline gtnx_graph_t H(Graph g) {
  HandleEntry* e = new_handles(1);
  e->g = std::move(g);
  return reinterpret_cast<gtnx_graph_t>(&e->g);
}

std::vector<Graph> vec(const gtnx_graph_t* a, int n, bool lazy_ok = false) {
  std::vector<Graph> v;
  v.reserve(n > 0 ? n : 0);
  for (int i = 0; i < n; ++i) v.push_back(lazy_ok ? GL(a[i]) : G(a[i]));
  return v;
}
void put(std::vector<Graph>& r, gtnx_graph_t* out) {
  if (r.empty()) return;
  HandleEntry* e = new_handles(r.size());
  for (size_t i = 0; i < r.size(); ++i) {
    e[i].g = std::move(r[i]);
    out[i] = reinterpret_cast<gtnx_graph_t>(&e[i].g);
  }
}
// Is p a HIP device address?  (hipPointerGetAttributes fails for ordinary host memory.)
bool is_device_pointer(const void* p) {
  static const bool have_gpu = Runtime::device_count() > 0;
  if (!have_gpu || !p) return false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}
void check_node(Graph& g, int n) {
  if (n < 0 || n >= g.num_nodes()) throw_range("node index out of range");
}
void check_arc(Graph& g, int a) {
  if (a < 0 || a >= g.num_arcs()) throw_range("arc index out of range");
}
} // namespace

namespace gtnx {
void destroy_handle(Graph* g) {  // (region.h: handles dropped inside a region are let go of by region_leave)
  HandleEntry* e = reinterpret_cast<HandleEntry*>(g);
  HandleSlab* s = e->slab;
  e->g.~Graph();
  if (s->live.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    s->~HandleSlab();
    ::operator delete(static_cast<void*>(s));
  }
}
}  // namespace gtnx

// ------------------------------------------------------------------ parallelMap regions
// A caller that maps the per-graph functions over a batch on host threads (gtn::parallelMap:
// parallel/parallel_map.h:153-188; benchmarks/ctc.cpp:150-165) issues B calls of batch size one.  Threads that
// announced themselves (gtnx_parallel_enter, made by include/gtn/parallel.h) have their calls DEFERRED: the
// functions return placeholder handles at once and the region's join (gtnx_parallel_flush) runs all of them as
// one batched launch per function -- region.h.  No thread waits for another one inside the region.
GTNX_API gtnx_status_t gtnx_parallel_enter(void) {
  return guard([&] { region_enter(); });
}
GTNX_API gtnx_status_t gtnx_parallel_leave(void) {
  return guard([&] { region_leave(); });
}
GTNX_API gtnx_status_t gtnx_parallel_flush(void) {
  return guard([&] { region_flush(); });
}

// ------------------------------------------------------------------ runtime
GTNX_API const char* gtnx_last_error(void) { return g_err.c_str(); }
namespace gtnx {
void set_last_error(const std::string& m) { g_err = m; }  // (comm.cpp reports through the same thread-local)
}
GTNX_API const char* gtnx_version(void) { return "0.1.0"; }
GTNX_API const char* gtnx_backend(void) { return "hip:gfx950"; }
GTNX_API int gtnx_device_count(void) { return Runtime::device_count(); }
GTNX_API gtnx_status_t gtnx_set_device(int d) {
  return guard([&] { Runtime::set_current_device(d); });
}
GTNX_API gtnx_status_t gtnx_get_device(int* d) {
  return guard([&] { *d = Runtime::current_device(); });
}
GTNX_API gtnx_status_t gtnx_compose_mode(int mode, int* previous) {
  return guard([&] {
    if (mode < -1 || mode > 2) throw_invalid("[gtnx_compose_mode] mode must be -1, 0, 1 or 2");
    const int old = compose_mode_hint(mode);
    if (previous) *previous = old;
  });
}
GTNX_API gtnx_status_t gtnx_set_stream(void* s) {
  return guard([&] { Runtime::get().set_stream(static_cast<hipStream_t>(s)); });
}
GTNX_API gtnx_status_t gtnx_synchronize(void) {
  return guard([&] {
    if (Runtime::initialized()) {
      Runtime::get().sync();
      deferred_resolve_all();
    }
  });
}
GTNX_API gtnx_status_t gtnx_memory_stats(uint64_t* r, uint64_t* u) {
  return guard([&] {
    if (r) *r = 0;
    if (u) *u = 0;
    if (Runtime::initialized()) Runtime::get().stats(r, u);
  });
}
GTNX_API gtnx_status_t gtnx_reclaim(void) {
  return guard([&] {
    // what THIS thread built, or let go of, and nobody refers to any more (runtime.h: every thread takes apart
    // what it allocated; the pool threads of a parallelMap call this when their share of a region is done)
    region_reclaim_thread();
  });
}
GTNX_API gtnx_status_t gtnx_debug_viterbi_ties(int64_t* seen, int64_t* unresolved) {
  return guard([&] {
    if (seen) *seen = g_viterbi_ties_seen.load();
    if (unresolved) *unresolved = g_viterbi_ties_unresolved.load();
  });
}
GTNX_API gtnx_status_t gtnx_debug_tie_ranks(gtnx_graph_t g, int* queue_rank, int* creation_rank, int* applies) {
  return guard([&] {
    Graph& x = GL(g);
    const std::vector<int>*k = nullptr, *c = nullptr;
    const bool ok = x.s->kind == KIND_EXPLICIT && !x.s->lazy && ctc_tie_ranks(*x.s, true, &k, &c);
    if (applies) *applies = ok ? 1 : 0;
    if (!ok) return;
    if (queue_rank) std::copy(k->begin(), k->end(), queue_rank);
    if (creation_rank) std::copy(c->begin(), c->end(), creation_rank);
  });
}
GTNX_API gtnx_status_t gtnx_empty_cache(void) {
  return guard([&] {
    if (Runtime::initialized()) Runtime::get().empty_cache();
  });
}

// ------------------------------------------------------------------ Graph
GTNX_API gtnx_status_t gtnx_graph_create(int calc_grad, gtnx_graph_t* out) {
  return guard([&] { *out = H(Graph(calc_grad != 0)); });
}
GTNX_API gtnx_status_t gtnx_graph_copy(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(RAW(g)); });  // (a placeholder's copies share its call)
}
GTNX_API gtnx_status_t gtnx_graph_deep_copy(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = H(Graph::deep_copy(G(g))); });
}
GTNX_API gtnx_status_t gtnx_graph_destroy(gtnx_graph_t g) {
  return guard([&] {
    Graph* p = reinterpret_cast<Graph*>(g);
    if (!p) return;
    if (region_active())
      region_trash(p);  // let go of when the thread leaves the region
    else if (p->s && p->s->pending)
      destroy_handle(p);  // a placeholder's handle is a reference to its slice: nothing to take apart here (region.cpp)
    else if (Runtime::initialized())  // taken apart later, by the thread that made the graph (runtime.h)
      Runtime::send(p->s && p->s->home ? p->s->home : Runtime::home(), p, [](void* q) { destroy_handle(static_cast<Graph*>(q)); });
    else
      destroy_handle(p);
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_node(gtnx_graph_t g, int s, int a, int* id) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    int i = gr.add_node(s != 0, a != 0);
    if (id) *id = i;
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_arc(gtnx_graph_t g, int src, int dst, int il, int ol, float w, int* id) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    int i = gr.add_arc(src, dst, il, ol, w);
    if (id) *id = i;
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_nodes(gtnx_graph_t g, int n, const uint8_t* s, const uint8_t* a) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    gr.add_nodes(n, s, a);
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_arcs(gtnx_graph_t g, int n, const int* src, const int* dst, const int* il,
                                           const int* ol, const float* w) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    gr.add_arcs(n, src, dst, il, ol, w);
  });
}
GTNX_API gtnx_status_t gtnx_graph_num_nodes(gtnx_graph_t g, int64_t* out) {
  return guard([&] { *out = G(g).num_nodes(); });
}
GTNX_API gtnx_status_t gtnx_graph_num_arcs(gtnx_graph_t g, int64_t* out) {
  return guard([&] { *out = G(g).num_arcs(); });
}
GTNX_API gtnx_status_t gtnx_graph_num_start(gtnx_graph_t g, int64_t* out) {
  return guard([&] { *out = G(g).num_start(); });
}
GTNX_API gtnx_status_t gtnx_graph_num_accept(gtnx_graph_t g, int64_t* out) {
  return guard([&] { *out = G(g).num_accept(); });
}
GTNX_API gtnx_status_t gtnx_graph_num_inputs(gtnx_graph_t g, int64_t* out) {
  return guard([&] { *out = int64_t(GL(g).g->inputs.size()); });
}
GTNX_API gtnx_status_t gtnx_graph_item(gtnx_graph_t g, float* out) {
  return guard([&] {
    Graph& raw = RAW(g);
    if (raw.s->pending && region_item(raw, out)) return;  // a batch record's scalar: no graph is built for it
    *out = G(g).item();
  });
}
GTNX_API gtnx_status_t gtnx_graph_arc_sort(gtnx_graph_t g, int ol) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    gr.arc_sort(ol != 0);
  });
}
GTNX_API gtnx_status_t gtnx_graph_mark_arc_sorted(gtnx_graph_t g, int ol) {
  return guard([&] {
    region_before_mutation(G(g));
    Structure& s = *G(g).s;
    bool& flag = ol ? s.olabel_sorted : s.ilabel_sorted;
    if (!flag) {
      flag = true;
      s.dev_valid = s.kind == KIND_LINEAR ? s.dev_valid : false;  // flags live in the device view
    }
  });
}
GTNX_API gtnx_status_t gtnx_graph_ilabel_sorted(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).s->ilabel_sorted; });
}
GTNX_API gtnx_status_t gtnx_graph_olabel_sorted(gtnx_graph_t g, int* out) {
  return guard([&] { *out = G(g).s->olabel_sorted; });
}
GTNX_API gtnx_status_t gtnx_graph_weights(gtnx_graph_t g, int mut, float** out) {
  return guard([&] {
    Graph& gr = G(g);
    if (mut) region_before_mutation(gr);
    *out = const_cast<float*>(gr.weights_host(mut != 0));
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_weights(gtnx_graph_t g, float* out) {
  return guard([&] {
    Graph& gr = G(g);
    const float* p = gr.weights_host(false);
    std::memcpy(out, p, sizeof(float) * size_t(gr.num_arcs()));
  });
}
GTNX_API gtnx_status_t gtnx_graph_set_weights(gtnx_graph_t g, const float* w) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    // a DEVICE address (e.g. a row of a torch tensor) is taken as one: the reference's one setWeights serves
    // both (pytorch_loss.py:53-61 has to go through .cpu() there)
    const bool dev = is_device_pointer(w);
    if (region_stage_weights(gr, w, dev)) return;
    if (dev)
      gr.set_weights_device(w);
    else
      gr.set_weights_host(w);
  });
}
GTNX_API gtnx_status_t gtnx_graph_set_weights_device(gtnx_graph_t g, const void* w) {
  return guard([&] {
    Graph& gr = G(g);
    region_before_mutation(gr);
    if (region_stage_weights(gr, static_cast<const float*>(w), true)) return;
    gr.set_weights_device(w);
  });
}
GTNX_API gtnx_status_t gtnx_graph_weights_device(gtnx_graph_t g, void** out) {
  return guard([&] {
    Graph& gr = G(g);
    std::vector<Weights*> v{gr.w.get()};
    ensure_weights_device_batch(v);
    *out = gr.w->dev;
  });
}
GTNX_API gtnx_status_t gtnx_graph_labels_to_array(gtnx_graph_t g, int* out, int il) {
  return guard([&] {
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      for (int64_t a = 0; a < s.A; ++a) out[a] = int(a % s.C);
      return;
    }
    s.ensure_host();
    std::memcpy(out, (il ? s.il : s.ol).data(), sizeof(int) * size_t(s.A));
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_start(gtnx_graph_t g, int* out) {
  return guard([&] {
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      out[0] = 0;
      return;
    }
    s.ensure_host();
    std::memcpy(out, s.start.data(), sizeof(int) * s.start.size());
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_accept(gtnx_graph_t g, int* out) {
  return guard([&] {
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      if (s.M > 0) out[0] = s.M;
      return;
    }
    s.ensure_host();
    std::memcpy(out, s.accept.data(), sizeof(int) * s.accept.size());
  });
}
GTNX_API gtnx_status_t gtnx_graph_is_start(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    Graph& gr = G(g);
    check_node(gr, n);
    Structure& s = *gr.s;
    if (s.kind == KIND_LINEAR) {
      *out = n == 0;
      return;
    }
    s.ensure_host();
    *out = (s.nflags[n] & NF_START) != 0;
  });
}
GTNX_API gtnx_status_t gtnx_graph_is_accept(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    Graph& gr = G(g);
    check_node(gr, n);
    Structure& s = *gr.s;
    if (s.kind == KIND_LINEAR) {
      *out = s.M > 0 && n == s.M;
      return;
    }
    s.ensure_host();
    *out = (s.nflags[n] & NF_ACCEPT) != 0;
  });
}
GTNX_API gtnx_status_t gtnx_graph_make_accept(gtnx_graph_t g, int n) {
  return guard([&] {
    Graph& gr = G(g);
    check_node(gr, n);
    region_before_mutation(gr);
    Structure& s = *gr.s;
    s.materialize();
    s.ensure_host();
    if (!(s.nflags[n] & NF_ACCEPT)) {  // graph.h:346-352 (sort flags are not touched)
      s.accept.push_back(n);
      s.nflags[n] |= NF_ACCEPT;
      s.dev_valid = false;
      s.dev_mem.reset();
      s.rec_mem.reset();
      s.sched.reset();
      // everything derived from the accept flags: the CTC-shape cache (graph.cpp: detect_ctc_shape compares them --
      // a ctcGraph with one more accept node is no longer the acceptor the device-built target records describe),
      // the leaf record made from it, the band and dense records (node flags are part of both)
      s.ctc_labels.reset();
      s.ctc_checked = false;
      s.leaf_batch.reset();
      s.band[0].reset();
      s.band[1].reset();
      s.dense[0].reset();
      s.dense[1].reset();
    }
  });
}
GTNX_API gtnx_status_t gtnx_graph_num_out(gtnx_graph_t g, int n, int64_t* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = G(g).s->num_out(n);
  });
}
GTNX_API gtnx_status_t gtnx_graph_num_in(gtnx_graph_t g, int n, int64_t* out) {
  return guard([&] {
    check_node(G(g), n);
    *out = G(g).s->num_in(n);
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_out(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      if (n < s.M)
        for (int c = 0; c < s.C; ++c) out[c] = n * s.C + c;
      return;
    }
    s.ensure_csr();
    std::memcpy(out, s.out_list.data() + s.out_off[n], sizeof(int) * size_t(s.out_off[n + 1] - s.out_off[n]));
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_in(gtnx_graph_t g, int n, int* out) {
  return guard([&] {
    check_node(G(g), n);
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      if (n > 0)
        for (int c = 0; c < s.C; ++c) out[c] = (n - 1) * s.C + c;
      return;
    }
    s.ensure_csr();
    std::memcpy(out, s.in_list.data() + s.in_off[n], sizeof(int) * size_t(s.in_off[n + 1] - s.in_off[n]));
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_arcs(gtnx_graph_t g, int* src, int* dst, int* il, int* ol) {
  return guard([&] {
    Structure& s = *G(g).s;
    if (s.kind == KIND_LINEAR) {
      for (int64_t a = 0; a < s.A; ++a) {
        if (src) src[a] = int(a / s.C);
        if (dst) dst[a] = int(a / s.C) + 1;
        if (il) il[a] = int(a % s.C);
        if (ol) ol[a] = int(a % s.C);
      }
      return;
    }
    s.ensure_host();
    size_t b = sizeof(int) * size_t(s.A);
    if (src) std::memcpy(src, s.src.data(), b);
    if (dst) std::memcpy(dst, s.dst.data(), b);
    if (il) std::memcpy(il, s.il.data(), b);
    if (ol) std::memcpy(ol, s.ol.data(), b);
  });
}
GTNX_API gtnx_status_t gtnx_graph_get_arc(gtnx_graph_t g, int a, int* src, int* dst, int* il, int* ol, float* w) {
  return guard([&] {
    Graph& gr = G(g);
    check_arc(gr, a);
    Structure& s = *gr.s;
    if (s.kind == KIND_LINEAR) {
      if (src) *src = a / s.C;
      if (dst) *dst = a / s.C + 1;
      if (il) *il = a % s.C;
      if (ol) *ol = a % s.C;
    } else {
      s.ensure_host();
      if (src) *src = s.src[a];
      if (dst) *dst = s.dst[a];
      if (il) *il = s.il[a];
      if (ol) *ol = s.ol[a];
    }
    if (w) *w = gr.weights_host(false)[a];
  });
}
GTNX_API gtnx_status_t gtnx_graph_set_weight(gtnx_graph_t g, int a, float w) {
  return guard([&] {
    Graph& gr = G(g);
    check_arc(gr, a);
    region_before_mutation(gr);
    gr.w->ensure_host();
    gr.w->host[a] = w;
    gr.w->dev_valid = false;
    gr.w->version++;
  });
}
GTNX_API gtnx_status_t gtnx_graph_calc_grad(gtnx_graph_t g, int* out) {
  return guard([&] { *out = GL(g).calc_grad(); });
}
GTNX_API gtnx_status_t gtnx_graph_set_calc_grad(gtnx_graph_t g, int c) {
  return guard([&] {
    region_sync_thread();
    GL(g).set_calc_grad(c != 0);
  });
}
GTNX_API gtnx_status_t gtnx_graph_is_grad_available(gtnx_graph_t g, int* out) {
  return guard([&] {
    region_sync_thread();  // (a backward queued by this thread inside a parallelMap region runs first)
    *out = GL(g).is_grad_available();
  });
}
GTNX_API gtnx_status_t gtnx_graph_grad(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] {
    region_sync_thread();
    *out = H(G(g).grad());
  });
}
GTNX_API gtnx_status_t gtnx_graph_zero_grad(gtnx_graph_t g) {
  return guard([&] {
    region_sync_thread();
    GL(g).zero_grad();
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_grad(gtnx_graph_t g, const float* v, int64_t n) {
  return guard([&] {
    region_sync_thread();
    G(g).add_grad_host(v, n);
  });
}
GTNX_API gtnx_status_t gtnx_graph_add_grad_graph(gtnx_graph_t g, gtnx_graph_t o) {
  return guard([&] {
    region_sync_thread();
    Graph& other = G(o);
    Graph& gr = G(g);
    if (!gr.calc_grad()) return;
    if (other.num_arcs() != gr.num_arcs()) throw_logic("[Graph::addGrad] Invalid grad size.");
    if (other.w->dev_valid && !other.w->host_escaped)
      gr.add_grad_device(other.w->dev_mem, other.w->dev, false);
    else
      gr.add_grad_host(other.weights_host(false), other.num_arcs());
  });
}
GTNX_API gtnx_status_t gtnx_graph_id(gtnx_graph_t g, uintptr_t* out) {
  return guard([&] { *out = GL(g).id(); });
}
GTNX_API gtnx_status_t gtnx_graph_create_op(gtnx_graph_t* inputs, int n, gtnx_grad_fn fn, void* ctx,
                                            void (*ctx_free)(void*), gtnx_graph_t* out) {
  return guard([&] {
    auto v = vec(inputs, n);
    *out = H(make_user_op(v, fn, ctx, ctx_free));
  });
}

GTNX_API gtnx_status_t gtnx_graph_get_input(gtnx_graph_t g, int i, gtnx_graph_t* out) {
  return guard([&] {
    auto& ins = GL(g).g->inputs;
    if (i < 0 || size_t(i) >= ins.size()) throw_range("input index out of range");
    *out = H(ins[size_t(i)]);
  });
}
GTNX_API gtnx_status_t gtnx_graph_set_inputs(gtnx_graph_t g, const gtnx_graph_t* inputs, int n) {
  return guard([&] {
    G(g).g->inputs = vec(inputs, n);
    for (auto& i : G(g).g->inputs) i.g->n_consumers++;
  });
}
GTNX_API gtnx_status_t gtnx_graph_set_grad_fn(gtnx_graph_t g, gtnx_grad_fn fn, void* ctx, void (*ctx_free)(void*)) {
  return guard([&] { set_user_grad_fn(G(g), fn, ctx, ctx_free); });
}
GTNX_API gtnx_status_t gtnx_graph_has_grad_fn(gtnx_graph_t g, int* out) {
  return guard([&] { *out = GL(g).g->has_grad_fn ? 1 : 0; });
}

// ------------------------------------------------------------------ creations
GTNX_API gtnx_status_t gtnx_scalar_graph(float v, int cg, gtnx_graph_t* out) {
  return guard([&] { *out = H(make_scalar_graph(v, cg != 0)); });
}
GTNX_API gtnx_status_t gtnx_linear_graph(int M, int N, int cg, gtnx_graph_t* out) {
  return guard([&] { *out = H(make_linear_graph(M, N, cg != 0)); });
}
GTNX_API gtnx_status_t gtnx_linear_graph_n(int B, int M, int N, int cg, const void* dev, gtnx_graph_t* out) {
  return guard([&] {
    auto r = make_linear_graphs_device(B, M, N, cg != 0, dev);
    put(r, out);
  });
}
GTNX_API gtnx_status_t gtnx_linear_graph_borrow_n(int B, int M, int N, int cg, const void* dev, gtnx_graph_t* out) {
  return guard([&] {
    if (!dev && B > 0 && M > 0 && N > 0) throw_invalid("[gtnx_linear_graph_borrow_n] null device tensor");
    auto r = make_linear_graphs_device(B, M, N, cg != 0, dev, /*borrow=*/true);
    put(r, out);
  });
}

// ------------------------------------------------------------------ functions
#define UNARY_FN(name, expr, LAZY_OK, ROP)                                  \
  GTNX_API gtnx_status_t name(gtnx_graph_t g, gtnx_graph_t* out) {          \
    return guard([&] {                                                      \
      if (region_active()) {                                                \
        *out = H(region_record(ROP, RAW(g), nullptr));                      \
        return;                                                             \
      }                                                                     \
      std::vector<Graph> v{LAZY_OK ? GL(g) : G(g)};                         \
      auto r = expr;                                                        \
      *out = H(std::move(r[0]));                                            \
    });                                                                     \
  }                                                                         \
  GTNX_API gtnx_status_t name##_n(const gtnx_graph_t* g, int n, gtnx_graph_t* out) { \
    return guard([&] {                                                      \
      if (n >= 2) {                                                         \
        run_vector(ROP, g, n, nullptr, 0, out);                             \
        return;                                                             \
      }                                                                     \
      auto v = vec(g, n, LAZY_OK);                                          \
      auto r = expr;                                                        \
      put(r, out);                                                          \
    });                                                                     \
  }
#define BINARY_FN(name, expr, ROP)                                                      \
  GTNX_API gtnx_status_t name(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out) {      \
    return guard([&] {                                                                  \
      if (region_active()) {                                                            \
        *out = H(region_record(ROP, RAW(a), &RAW(b)));                                  \
        return;                                                                         \
      }                                                                                 \
      std::vector<Graph> va{G(a)}, vb{G(b)};                                            \
      auto r = expr;                                                                    \
      *out = H(std::move(r[0]));                                                        \
    });                                                                                 \
  }                                                                                     \
  GTNX_API gtnx_status_t name##_n(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, \
                                  int nb, gtnx_graph_t* out) {                          \
    return guard([&] {                                                                  \
      if (na >= 2 || nb >= 2) {                                                         \
        run_vector(ROP, a, na, b, nb, out);                                             \
        return;                                                                         \
      }                                                                                 \
      auto va = vec(a, na);                                                             \
      auto vb = vec(b, nb);                                                             \
      auto r = expr;                                                                    \
      put(r, out);                                                                      \
    });                                                                                 \
  }

namespace {
std::vector<Graph> g_empty;
// a vector form with more than one element: through the machinery of the parallelMap regions, joined at once
// (region.h: region_run_vector) -- results over one batch record where the elements allow it
void run_vector(RegionOp op, const gtnx_graph_t* a, int na, const gtnx_graph_t* b, int nb, gtnx_graph_t* out) {
  std::vector<Graph*> pa(static_cast<size_t>(na > 0 ? na : 0)), pb(static_cast<size_t>(b && nb > 0 ? nb : 0));
  for (int i = 0; i < na; ++i) pa[size_t(i)] = &RAW(a[i]);
  for (size_t i = 0; i < pb.size(); ++i) pb[i] = &RAW(b[i]);
  const int n = b ? std::max(na, nb) : na;
  std::vector<Graph> res(static_cast<size_t>(n), Graph(Graph::Empty{}));
  region_run_vector(op, pa.data(), na, b ? pb.data() : nullptr, nb, res.data());
  put(res, out);
}
}
UNARY_FN(gtnx_negate, op_scalar(SK_NEGATE, v, g_empty), false, RO_NEG)
BINARY_FN(gtnx_add, op_scalar(SK_ADD, va, vb), RO_ADD)
BINARY_FN(gtnx_subtract, op_scalar(SK_SUBTRACT, va, vb), RO_SUB)
BINARY_FN(gtnx_compose, op_compose(va, vb, false), RO_COMPOSE)
BINARY_FN(gtnx_intersect, op_compose(va, vb, true), RO_INTERSECT)
UNARY_FN(gtnx_forward_score, op_shortest_distance(v, false), true, RO_FS)
UNARY_FN(gtnx_viterbi_score, op_shortest_distance(v, true), true, RO_VS)
UNARY_FN(gtnx_viterbi_path, op_viterbi_path(v), true, RO_VP)

GTNX_API gtnx_status_t gtnx_items_n(const gtnx_graph_t* g, int n, float* out) {
  return guard([&] {
    // results of a vector form / a parallelMap region held as batch-record scalars: read from the record (one
    // device->host copy per record, cached), no graph is built for them
    bool all = n > 0;
    for (int i = 0; i < n && all; ++i) {
      Graph& raw = RAW(g[i]);
      all = raw.s->pending && region_item(raw, out + i);
    }
    if (all) return;
    auto v = vec(g, n);
    items_host(v, out);
  });
}
GTNX_API gtnx_status_t gtnx_items_device_n(const gtnx_graph_t* g, int n, void* out) {
  return guard([&] {
    std::vector<Graph*> hs(static_cast<size_t>(n > 0 ? n : 0));
    for (int i = 0; i < n; ++i) hs[size_t(i)] = &RAW(g[i]);
    if (region_items_device(hs.data(), n, out)) return;
    auto v = vec(g, n);
    items_device(v, out);
  });
}
GTNX_API gtnx_status_t gtnx_grads_device_n(const gtnx_graph_t* g, int n, void* out, const int64_t* offsets) {
  return guard([&] {
    auto v = vec(g, n);
    grads_device(v, out, offsets);
  });
}
GTNX_API gtnx_status_t gtnx_grads_bind_device_n(const gtnx_graph_t* g, int n, void* out, const int64_t* offsets) {
  return guard([&] {
    auto v = vec(g, n);
    grads_bind_device(v, out, offsets);
  });
}

// ------------------------------------------------------------------ rational operations (device-built)
GTNX_API gtnx_status_t gtnx_clone(gtnx_graph_t g, int projection, gtnx_graph_t* out) {
  return guard([&] {
    if (projection < 0 || projection > 2) throw_invalid("[gtnx_clone] projection must be 0, 1 or 2");
    std::vector<Graph> v{G(g)};
    *out = H(op_rational(RAT_CLONE, v, projection));
  });
}
GTNX_API gtnx_status_t gtnx_concat(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] {
    auto v = vec(g, n);
    *out = H(op_rational(RAT_CONCAT, v, 0));
  });
}
GTNX_API gtnx_status_t gtnx_closure(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] {
    std::vector<Graph> v{G(g)};
    *out = H(op_rational(RAT_CLOSURE, v, 0));
  });
}
GTNX_API gtnx_status_t gtnx_union(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] {
    auto v = vec(g, n);
    *out = H(op_rational(RAT_UNION, v, 0));
  });
}

GTNX_API gtnx_status_t gtnx_remove(gtnx_graph_t g, int ilabel, int olabel, gtnx_graph_t* out) {
  return guard([&] { *out = H(op_remove(G(g), ilabel, olabel)); });
}

// ------------------------------------------------------------------ batch records
namespace {
inline BatchP& BH(gtnx_batch_t h) {
  if (!h) throw_invalid("null batch handle");
  return *reinterpret_cast<BatchP*>(h);
}
inline gtnx_batch_t HB(BatchP b) { return reinterpret_cast<gtnx_batch_t>(new BatchP(std::move(b))); }
} // namespace
GTNX_API gtnx_status_t gtnx_batch_from_graphs(const gtnx_graph_t* g, int n, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_from_graphs(vec(g, n, true))); });
}
GTNX_API gtnx_status_t gtnx_batch_ctc_targets(const int* labels, const int* lengths, int n, int blank, int cg,
                                              gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_ctc_targets(labels, lengths, n, blank, cg != 0)); });
}
GTNX_API gtnx_status_t gtnx_batch_asg_force_align(const int* labels, const int* lengths, int n, gtnx_graph_t transitions,
                                                  int n_labels, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_asg_force_align(labels, lengths, n, G(transitions), n_labels)); });
}
GTNX_API gtnx_status_t gtnx_batch_linear(int n, int M, int N, int cg, const void* dev, int borrow, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_linear(n, M, N, cg != 0, dev, borrow != 0)); });
}
GTNX_API gtnx_status_t gtnx_batch_destroy(gtnx_batch_t b) {
  return guard([&] {
    auto* p = reinterpret_cast<BatchP*>(b);
    if (!p) return;
    if (Runtime::initialized())  // like gtnx_graph_destroy: taken apart off the caller's critical path
      // (weight: the graphs a materialised record stands for -- a scoring loop that never blocks must not pile up
      //  records of hundreds of graphs behind one-pointer entries, runtime.cpp: kDeferFull)
      Runtime::get().defer_delete(p, [](void* q) { delete static_cast<BatchP*>(q); },
                                  (*p && (*p)->materialised) ? size_t((*p)->n > 0 ? (*p)->n : 1) : 1);
    else
      delete p;
  });
}
GTNX_API gtnx_status_t gtnx_batch_size(gtnx_batch_t b, int* out) {
  return guard([&] { *out = BH(b)->n; });
}
GTNX_API gtnx_status_t gtnx_batch_get(gtnx_batch_t b, int i, gtnx_graph_t* out) {
  return guard([&] { *out = H(batch_get(BH(b), i)); });
}
GTNX_API gtnx_status_t gtnx_batch_negate(gtnx_batch_t a, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_scalar(SK_NEGATE, BH(a), nullptr)); });
}
GTNX_API gtnx_status_t gtnx_batch_add(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_scalar(SK_ADD, BH(a), BH(b))); });
}
GTNX_API gtnx_status_t gtnx_batch_subtract(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_scalar(SK_SUBTRACT, BH(a), BH(b))); });
}
GTNX_API gtnx_status_t gtnx_batch_subtract_into(gtnx_batch_t a, gtnx_batch_t b, void* items_device, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_scalar(SK_SUBTRACT, BH(a), BH(b), items_device)); });
}
GTNX_API gtnx_status_t gtnx_batch_compose(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_compose(BH(a), BH(b), false)); });
}
GTNX_API gtnx_status_t gtnx_batch_intersect(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_compose(BH(a), BH(b), true)); });
}
GTNX_API gtnx_status_t gtnx_batch_forward_score(gtnx_batch_t a, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_shortest_distance(BH(a), false)); });
}
GTNX_API gtnx_status_t gtnx_batch_viterbi_score(gtnx_batch_t a, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_shortest_distance(BH(a), true)); });
}
GTNX_API gtnx_status_t gtnx_batch_viterbi_path(gtnx_batch_t a, gtnx_batch_t* out) {
  return guard([&] { *out = HB(batch_viterbi_path(BH(a))); });
}
GTNX_API gtnx_status_t gtnx_batch_backward(gtnx_batch_t a, int retain) {
  return guard([&] { batch_backward(BH(a), retain != 0); });
}
GTNX_API gtnx_status_t gtnx_batch_items(gtnx_batch_t a, float* out) {
  return guard([&] { batch_items_host(BH(a), out); });
}
GTNX_API gtnx_status_t gtnx_batch_items_device(gtnx_batch_t a, void* out) {
  return guard([&] { batch_items_device(BH(a), out); });
}
GTNX_API gtnx_status_t gtnx_batch_grads_bind_device(gtnx_batch_t a, void* out, const int64_t* offsets) {
  return guard([&] { batch_grads_bind(BH(a), out, offsets); });
}
GTNX_API gtnx_status_t gtnx_batch_grads_device(gtnx_batch_t a, void* out, const int64_t* offsets) {
  return guard([&] { batch_grads_device(BH(a), out, offsets); });
}

// ------------------------------------------------------------------ autograd
GTNX_API gtnx_status_t gtnx_backward(gtnx_graph_t g, int retain) {
  return guard([&] {
    if (region_active()) {
      region_record_backward(RAW(g), retain != 0);
      return;
    }
    std::vector<Graph> v{G(g)};
    op_backward(v, nullptr, retain != 0);
  });
}
GTNX_API gtnx_status_t gtnx_backward_with_grad(gtnx_graph_t g, gtnx_graph_t grad, int retain) {
  return guard([&] {
    std::vector<Graph> v{G(g)};
    op_backward(v, &G(grad), retain != 0);
  });
}
GTNX_API gtnx_status_t gtnx_backward_n(const gtnx_graph_t* g, int n, int retain) {
  return guard([&] {
    if (n >= 2) {
      std::vector<Graph*> roots(static_cast<size_t>(n));
      for (int i = 0; i < n; ++i) roots[size_t(i)] = &RAW(g[i]);
      region_run_backward_vector(roots.data(), n, retain != 0);
      return;
    }
    auto v = vec(g, n);
    op_backward(v, nullptr, retain != 0);
  });
}

// ------------------------------------------------------------------ utils
GTNX_API gtnx_status_t gtnx_equal(gtnx_graph_t a, gtnx_graph_t b, int* out) {
  return guard([&] { *out = graphs_equal(G(a), G(b)); });
}
GTNX_API gtnx_status_t gtnx_isomorphic(gtnx_graph_t a, gtnx_graph_t b, int* out) {
  return guard([&] { *out = graphs_isomorphic(G(a), G(b)); });
}

// ------------------------------------------------------------------ formats
GTNX_API gtnx_status_t gtnx_graph_load_buffer(const void* data, size_t bytes, gtnx_graph_t* out) {
  return guard([&] {
    if (!data) throw_invalid("[gtnx_graph_load_buffer] null buffer");
    *out = H(op_load_buffer(data, bytes));
  });
}

// ------------------------------------------------------------------ profiling
GTNX_API gtnx_status_t gtnx_prof_enable(int on) {
  return guard([&] {
    if (!on) deferred_resolve_all();  // algorithmic bytes owed by unresolved compositions
    Runtime::get().prof_enable(on != 0);
  });
}
GTNX_API gtnx_status_t gtnx_prof_reset(void) {
  return guard([&] { Runtime::get().prof_reset(); });
}
GTNX_API gtnx_status_t gtnx_prof_get(const char* name, double* ms, int64_t* n, double* bytes) {
  return guard([&] {
    deferred_resolve_all();
    ProfEntry e = Runtime::get().prof_get(name);
    if (ms) *ms = e.total_ms;
    if (n) *n = e.launches;
    if (bytes) *bytes = e.bytes;
  });
}
GTNX_API gtnx_status_t gtnx_debug_symbolic_route(gtnx_graph_t g, int tropical, int* route) {
  return guard([&] {
    Graph& raw = GL(g);
    *route = (raw.s && raw.s->lazy) ? int(symbolic_route(*raw.s->lazy, tropical != 0)) : -1;
  });
}
GTNX_API gtnx_status_t gtnx_debug_route_name(int route, char* buf, size_t cap) {
  return guard([&] {
    if (cap) {
      std::strncpy(buf, symbolic_route_name(route), cap - 1);
      buf[cap - 1] = 0;
    }
  });
}
GTNX_API gtnx_status_t gtnx_prof_names(char* buf, size_t cap) {
  return guard([&] {
    std::string s = Runtime::get().prof_names();
    if (cap) {
      std::strncpy(buf, s.c_str(), cap - 1);
      buf[cap - 1] = 0;
    }
  });
}
