// gtn/graph.h -- drop-in replacement for the reference's gtn/graph.h:75-415.
// Header-only shim over the C ABI of libgtn_amd.so (include/gtn_amd.h): same class,
// same members, same exception types and messages; the storage and every graph
// function live on the MI355X behind the ABI.
#pragma once

// (the standard headers the reference's graph.h:10-16 pulls in: callers rely on them transitively)
#include <atomic>
#include <cassert>
#include <climits>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "gtn_amd.h"

namespace gtn {

/** The index of the epsilon label (reference gtn/graph.h:21). */
constexpr int epsilon{-1};

namespace detail {
inline void check(gtnx_status_t st) {
  if (st == GTNX_OK) return;
  const std::string msg = gtnx_last_error();
  switch (st) {
    case GTNX_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case GTNX_LOGIC_ERROR: throw std::logic_error(msg);
    case GTNX_OUT_OF_RANGE: throw std::out_of_range(msg);
    default: throw std::runtime_error(msg);
  }
}
} // namespace detail

class Graph {
 public:
  using GradFunc = std::function<void(std::vector<Graph>& inputs, Graph& deltas)>;

  /** Graph(GradFunc, inputs), reference graph.h:78 */
  Graph(GradFunc gradFunc, std::vector<Graph> inputs) : fresh_(true) {
    std::vector<gtnx_graph_t> hs;
    for (auto& g : inputs) hs.push_back(g.handle());
    auto* ctx = gradFunc ? new GradFunc(std::move(gradFunc)) : nullptr;
    detail::check(gtnx_graph_create_op(hs.data(), static_cast<int>(hs.size()), ctx ? &Graph::trampoline : nullptr,
                                       ctx, ctx ? &Graph::freeCtx : nullptr, &h_));
  }
  /** Graph(bool calcGrad = true), reference graph.h:89 */
  Graph(bool calcGrad = true) : fresh_(true) { detail::check(gtnx_graph_create(calcGrad ? 1 : 0, &h_)); }
  Graph(const Graph& o) {  // aliases, like the reference
    detail::check(gtnx_graph_copy(o.h(), &h_));
    o.aliased_ = aliased_ = true;  // two objects over one graph: neither may keep arcs to itself any more
  }
  Graph(Graph&& o) noexcept
      : h_(o.h_), fresh_(o.fresh_), aliased_(o.aliased_), baseN_(o.baseN_), baseA_(o.baseA_), build_(std::move(o.build_)) {
    dirty_.store(o.dirty_.load(std::memory_order_relaxed), std::memory_order_relaxed);
    o.h_ = nullptr;
    o.dirty_.store(false, std::memory_order_relaxed);
  }
  Graph& operator=(const Graph& o) {
    if (this != &o) {
      gtnx_graph_t n;
      detail::check(gtnx_graph_copy(o.h(), &n));
      reset();
      h_ = n;
      o.aliased_ = aliased_ = true;
      fresh_ = false;
      std::lock_guard<std::mutex> lk(listMutex_);
      lists_.clear();  // (of the graph this object referred to before)
    }
    return *this;
  }
  Graph& operator=(Graph&& o) noexcept {
    if (this != &o) {
      reset();
      {
        std::lock_guard<std::mutex> lk(listMutex_);
        lists_.clear();
      }
      h_ = o.h_;
      fresh_ = o.fresh_;
      aliased_ = o.aliased_;
      baseN_ = o.baseN_;
      baseA_ = o.baseA_;
      build_ = std::move(o.build_);
      dirty_.store(o.dirty_.load(std::memory_order_relaxed), std::memory_order_relaxed);
      o.h_ = nullptr;
      o.dirty_.store(false, std::memory_order_relaxed);
    }
    return *this;
  }
  ~Graph() { reset(); }

  // addNode / addArc on a graph that only this object refers to are COLLECTED here and handed to the engine in
  // two bulk calls (gtnx_graph_add_nodes / gtnx_graph_add_arcs: same ids, same order) the first time anything
  // else is asked of the graph -- a target graph is several hundred of these calls, and a call across the ABI
  // costs ten times what the append does.  Ids are known without asking (a fresh graph starts at 0 and nobody
  // else adds to it).  What moves: an invalid arc (node out of range, label < epsilon -- the reference only
  // asserts both, graph.h:423-433) is reported by the call that hands the arcs over, not by its own addArc.
  int addNode(bool start = false, bool accept = false) {
    if (collecting()) {
      build_->start.push_back(start);
      build_->accept.push_back(accept);
      dirty_.store(true, std::memory_order_release);
      return baseN_ + static_cast<int>(build_->start.size()) - 1;
    }
    int id;
    detail::check(gtnx_graph_add_node(h_, start, accept, &id));
    return id;
  }
  size_t addArc(size_t srcNode, size_t dstNode, int label) { return addArc(srcNode, dstNode, label, label); }
  size_t addArc(size_t srcNode, size_t dstNode, int ilabel, int olabel, float weight = 0.0) {
    if (collecting()) {
      Build& b = *build_;
      b.src.push_back(static_cast<int>(srcNode));
      b.dst.push_back(static_cast<int>(dstNode));
      b.il.push_back(ilabel);
      b.ol.push_back(olabel);
      if (weight != 0.0f && b.w.empty()) b.w.assign(b.src.size() - 1, 0.0f);
      if (weight != 0.0f || !b.w.empty()) b.w.push_back(weight);
      dirty_.store(true, std::memory_order_release);
      return static_cast<size_t>(baseA_) + b.src.size() - 1;
    }
    int id;
    detail::check(gtnx_graph_add_arc(h_, static_cast<int>(srcNode), static_cast<int>(dstNode), ilabel, olabel,
                                     weight, &id));
    return static_cast<size_t>(id);
  }
  size_t numArcs() const { return count(&gtnx_graph_num_arcs); }
  size_t numNodes() const { return count(&gtnx_graph_num_nodes); }
  size_t numStart() const { return count(&gtnx_graph_num_start); }
  size_t numAccept() const { return count(&gtnx_graph_num_accept); }
  float item() const {
    float v;
    detail::check(gtnx_graph_item(h(), &v));
    return v;
  }
  static Graph deepCopy(const Graph& src) {
    gtnx_graph_t n;
    detail::check(gtnx_graph_deep_copy(src.h(), &n));
    return Graph(n);
  }
  void arcSort(bool olabel = false) { detail::check(gtnx_graph_arc_sort(h(), olabel)); }
  void markArcSorted(bool olabel = false) { detail::check(gtnx_graph_mark_arc_sorted(h(), olabel)); }
  bool ilabelSorted() const {
    int v;
    detail::check(gtnx_graph_ilabel_sorted(h(), &v));
    return v != 0;
  }
  bool olabelSorted() const {
    int v;
    detail::check(gtnx_graph_olabel_sorted(h(), &v));
    return v != 0;
  }
  float* weights() {
    float* p;
    detail::check(gtnx_graph_weights(h(), 1, &p));
    return p;
  }
  const float* weights() const {
    float* p;
    detail::check(gtnx_graph_weights(h(), 0, &p));
    return p;
  }
  void setWeights(const float* weights) { detail::check(gtnx_graph_set_weights(h(), weights)); }
  /** extension: copy numArcs floats from a DEVICE buffer (no host round trip) */
  void setWeightsDevice(const void* deviceWeights) { detail::check(gtnx_graph_set_weights_device(h(), deviceWeights)); }
  void labelsToArray(int* out, bool ilabel = true) { detail::check(gtnx_graph_labels_to_array(h(), out, ilabel)); }
  std::vector<int> labelsToVector(bool ilabel = true) {
    std::vector<int> out(numArcs());
    labelsToArray(out.data(), ilabel);
    return out;
  }

  void addGrad(std::vector<float>&& other) { detail::check(gtnx_graph_add_grad(h(), other.data(), (int64_t)other.size())); }
  void addGrad(const std::vector<float>& other) {
    detail::check(gtnx_graph_add_grad(h(), other.data(), (int64_t)other.size()));
  }
  void addGrad(const Graph& other) { detail::check(gtnx_graph_add_grad_graph(h(), other.h())); }
  bool calcGrad() const {
    int v;
    detail::check(gtnx_graph_calc_grad(h(), &v));
    return v != 0;
  }
  bool isGradAvailable() const {
    int v;
    detail::check(gtnx_graph_is_grad_available(h(), &v));
    return v != 0;
  }
  Graph& grad() { return const_cast<Graph&>(static_cast<const Graph&>(*this).grad()); }
  const Graph& grad() const {
    gtnx_graph_t n;
    detail::check(gtnx_graph_grad(h(), &n));
    grad_.reset(new Graph(n));
    return *grad_;
  }
  void setCalcGrad(bool calcGrad) { detail::check(gtnx_graph_set_calc_grad(h(), calcGrad)); }
  void zeroGrad() { detail::check(gtnx_graph_zero_grad(h())); }
  std::uintptr_t id() {
    std::uintptr_t v;
    detail::check(gtnx_graph_id(h(), &v));
    return v;
  }
  /** non-null iff a gradient function is attached (reference graph.h:286); the
   *  engine owns the function, so the returned callable only reports presence */
  GradFunc gradFunc() {
    int v;
    detail::check(gtnx_graph_has_grad_fn(h(), &v));
    if (!v) return nullptr;
    return [](std::vector<Graph>&, Graph&) {
      throw std::logic_error("[Graph::gradFunc] engine-owned gradient functions run through gtn::backward");
    };
  }
  void setGradFunc(GradFunc gradFunc) {
    auto* ctx = gradFunc ? new GradFunc(std::move(gradFunc)) : nullptr;
    detail::check(gtnx_graph_set_grad_fn(h(), ctx ? &Graph::trampoline : nullptr, ctx, ctx ? &Graph::freeCtx : nullptr));
  }
  std::vector<Graph>& inputs() const {
    int64_t n;
    detail::check(gtnx_graph_num_inputs(h(), &n));
    inputs_.clear();
    for (int64_t i = 0; i < n; ++i) {
      gtnx_graph_t in;
      detail::check(gtnx_graph_get_input(h(), (int)i, &in));
      inputs_.push_back(Graph(in));
    }
    return inputs_;
  }
  void setInputs(std::vector<Graph> inputs) {
    std::vector<gtnx_graph_t> hs;
    for (auto& g : inputs) hs.push_back(g.handle());
    detail::check(gtnx_graph_set_inputs(h(), hs.data(), (int)hs.size()));
  }
  /** reference graph.h:317-321 drops the weights to save memory on the tape; the
   *  engine keeps device buffers alive through the tape itself, so this aliases */
  Graph withoutWeights() const { return *this; }

  const std::vector<int>& start() const {
    return fillSlot(0, 0, numStart(), [this](int* p) { return gtnx_graph_get_start(h_, p); });
  }
  const std::vector<int>& accept() const {
    return fillSlot(1, 0, numAccept(), [this](int* p) { return gtnx_graph_get_accept(h_, p); });
  }
  bool isStart(size_t i) const {
    int v;
    detail::check(gtnx_graph_is_start(h(), (int)i, &v));
    return v != 0;
  }
  bool isAccept(size_t i) const {
    int v;
    detail::check(gtnx_graph_is_accept(h(), (int)i, &v));
    return v != 0;
  }
  void makeAccept(size_t i) { detail::check(gtnx_graph_make_accept(h(), (int)i)); }
  size_t numOut(size_t i) const {
    int64_t v;
    detail::check(gtnx_graph_num_out(h(), (int)i, &v));
    return (size_t)v;
  }
  const std::vector<int>& out(size_t i) const {
    return fillSlot(2, i, numOut(i), [this, i](int* p) { return gtnx_graph_get_out(h_, (int)i, p); });
  }
  int out(size_t i, size_t j) const { return out(i)[j]; }
  size_t numIn(size_t i) const {
    int64_t v;
    detail::check(gtnx_graph_num_in(h(), (int)i, &v));
    return (size_t)v;
  }
  const std::vector<int>& in(size_t i) const {
    return fillSlot(3, i, numIn(i), [this, i](int* p) { return gtnx_graph_get_in(h_, (int)i, p); });
  }
  size_t in(size_t i, size_t j) const { return (size_t)in(i)[j]; }

  int srcNode(size_t i) const { return arcField(i, 0); }
  int dstNode(size_t i) const { return arcField(i, 1); }
  int label(size_t i) const { return arcField(i, 2); }
  int ilabel(size_t i) const { return arcField(i, 2); }
  int olabel(size_t i) const { return arcField(i, 3); }
  float weight(size_t i) const {
    float w;
    detail::check(gtnx_graph_get_arc(h(), (int)i, nullptr, nullptr, nullptr, nullptr, &w));
    return w;
  }
  void setWeight(size_t i, float weight) { detail::check(gtnx_graph_set_weight(h(), (int)i, weight)); }

  /** the C-ABI handle (for the batched free functions) */
  gtnx_graph_t handle() const {
    aliased_ = true;  // the engine may keep a reference of its own from here on (an op's inputs)
    return h();
  }
  /** adopt a handle returned by the C ABI */
  static Graph fromHandle(gtnx_graph_t h) { return Graph(h); }

 private:
  size_t addArc(size_t srcNode, size_t dstNode, int label, float) = delete;   // reference graph.h:419-420
  size_t addArc(size_t srcNode, size_t dstNode, int label, double) = delete;
  explicit Graph(gtnx_graph_t h) : h_(h) {}
  void reset() {
    if (h_) gtnx_graph_destroy(h_);  // (collected nodes / arcs of a graph nobody ever looked at go with it)
    h_ = nullptr;
    if (build_) {  // (back to the thread's spare slot, capacity kept: see collecting())
      std::unique_ptr<Build>* spare = spareBuild();
      if (spare && !*spare && build_->src.capacity() <= (1u << 16)) {
        Build& b = *build_;
        b.start.clear(), b.accept.clear(), b.src.clear(), b.dst.clear(), b.il.clear(), b.ol.clear(), b.w.clear();
        *spare = std::move(build_);
      }
      build_.reset();
    }
    dirty_.store(false, std::memory_order_relaxed);
  }
  // ---- collected addNode / addArc calls (see addNode)
  struct Build {
    std::vector<uint8_t> start, accept;
    std::vector<int> src, dst, il, ol;
    std::vector<float> w;  // empty while every weight so far is 0
  };
  bool collecting() {
    if (aliased_ || !fresh_) return false;
    if (!build_) {
      // (a thread that builds one target graph per task builds them all alike: the vectors of the last graph
      //  handed over on this thread, capacity included, serve the next one -- no allocation per graph)
      std::unique_ptr<Build>* spare = spareBuild();
      if (spare && *spare) {
        build_ = std::move(*spare);
      } else {
        build_.reset(new Build());
        build_->src.reserve(64), build_->dst.reserve(64), build_->il.reserve(64), build_->ol.reserve(64);
        build_->start.reserve(32), build_->accept.reserve(32);
      }
    }
    return true;
  }
  // the thread's spare slot; null once the thread's thread_local objects are gone (graphs may still be destroyed
  // on it after that)
  static std::unique_ptr<Build>* spareBuild() {
    struct Holder {
      std::unique_ptr<Build> spare;
      bool* gone;
      explicit Holder(bool* g) : gone(g) {}
      ~Holder() { *gone = true; }
    };
    static thread_local bool gone = false;  // (trivially destructible: stays readable)
    if (gone) return nullptr;
    static thread_local Holder h(&gone);
    return &h.spare;
  }
  /** the handle, with everything collected so far handed over */
  gtnx_graph_t h() const {
    if (dirty_.load(std::memory_order_acquire)) handOver();
    return h_;
  }
  void handOver() const {
    std::lock_guard<std::mutex> lk(listMutex_);  // (const readers on several threads may arrive together)
    if (!dirty_.load(std::memory_order_acquire)) return;
    Build& b = *build_;
    const int nn = static_cast<int>(b.start.size()), na = static_cast<int>(b.src.size());
    // (counted as handed over even if the engine refuses them: the error is reported once)
    baseN_ += nn;
    baseA_ += na;
    std::vector<uint8_t> st, ac;
    std::vector<int> src, dst, il, ol;
    std::vector<float> w;
    st.swap(b.start), ac.swap(b.accept), src.swap(b.src), dst.swap(b.dst), il.swap(b.il), ol.swap(b.ol), w.swap(b.w);
    struct Done {  // readers on other threads wait (listMutex_) until the engine HAS the arcs
      std::atomic<bool>& d;
      ~Done() { d.store(false, std::memory_order_release); }
    } done{dirty_};
    if (nn) detail::check(gtnx_graph_add_nodes(h_, nn, st.data(), ac.data()));
    if (na) detail::check(gtnx_graph_add_arcs(h_, na, src.data(), dst.data(), il.data(), ol.data(), w.empty() ? nullptr : w.data()));
    // keep the capacity for the next round: of this graph or, more likely, of the next graph this thread builds
    st.clear(), ac.clear(), src.clear(), dst.clear(), il.clear(), ol.clear(), w.clear();
    b.start.swap(st), b.accept.swap(ac), b.src.swap(src), b.dst.swap(dst), b.il.swap(il), b.ol.swap(ol);
    std::unique_ptr<Build>* spare = spareBuild();
    if (spare && !*spare && b.src.capacity() <= (1u << 16)) *spare = std::move(build_);
  }
  size_t count(gtnx_status_t (*fn)(gtnx_graph_t, int64_t*)) const {
    int64_t v;
    detail::check(fn(h(), &v));
    return (size_t)v;
  }
  int arcField(size_t i, int which) const {
    int v[4];
    detail::check(gtnx_graph_get_arc(h(), (int)i, &v[0], &v[1], &v[2], &v[3], nullptr));
    return v[which];
  }
  static gtnx_status_t trampoline(void* ctx, gtnx_graph_t* inputs, int n, gtnx_graph_t deltas) {
    try {
      std::vector<Graph> ins;
      for (int i = 0; i < n; ++i) {
        gtnx_graph_t c;
        detail::check(gtnx_graph_copy(inputs[i], &c));
        ins.push_back(Graph(c));
      }
      gtnx_graph_t dc;
      detail::check(gtnx_graph_copy(deltas, &dc));
      Graph d(dc);
      (*static_cast<GradFunc*>(ctx))(ins, d);
      return GTNX_OK;
    } catch (const std::invalid_argument&) {
      return GTNX_INVALID_ARGUMENT;
    } catch (const std::logic_error&) {
      return GTNX_LOGIC_ERROR;
    } catch (...) {
      return GTNX_RUNTIME_ERROR;
    }
  }
  static void freeCtx(void* ctx) { delete static_cast<GradFunc*>(ctx); }

  gtnx_graph_t h_{nullptr};
  bool fresh_{false};             // created empty by this object: node / arc ids are known without asking
  mutable bool aliased_{false};   // another Graph object refers to the same graph
  mutable int baseN_{0}, baseA_{0};
  mutable std::unique_ptr<Build> build_;
  mutable std::atomic<bool> dirty_{false};
  mutable std::unique_ptr<Graph> grad_;
  mutable std::vector<Graph> inputs_;
  // start() / accept() / out(i) / in(i) hand out references like the reference does (graph.h:293-325 there):
  // one host copy per (list, node), refreshed on every call, so `g.out(n).begin(), g.out(n).end()` and nested
  // loops over different nodes see stable storage.  Slots of one Graph object are not shared with its copies.
  // Looked up, sized AND filled under the lock: two threads reading the same node of one Graph object (a graph
  // captured by reference in a parallelMap lambda) then write the same values into storage of the same size, and
  // nobody reads a vector while it is being resized.  (`n` was asked for before the lock: the count functions hand
  // collected arcs over first, under the same mutex.)
  template <class Get>
  const std::vector<int>& fillSlot(int which, size_t node, size_t n, Get&& get) const {
    std::lock_guard<std::mutex> lk(listMutex_);
    std::vector<int>& v = lists_[(uint64_t(node) << 2) | uint64_t(which)];
    if (v.size() != n) v.resize(n);
    detail::check(get(v.data()));
    return v;
  }
  mutable std::mutex listMutex_;
  mutable std::unordered_map<uint64_t, std::vector<int>> lists_;
};

} // namespace gtn
