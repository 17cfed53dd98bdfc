// gtn/batch.h -- extension: B graphs held as one record (gtnx_batch_* in gtn_amd.h).
//
// What the reference writes as parallelMap over per-utterance graphs
// (benchmarks/ctc.cpp:150-165) reads the same here with one Batch per call:
//   using namespace gtn::batched;
//   auto loss = subtract(forwardScore(ems), forwardScore(intersect(targets, ems)));
//   backward(loss);
// Elements are ordinary graphs whenever somebody asks for one (operator[]).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "gtn/functions.h"

namespace gtn {

class Batch {
 public:
  Batch() = default;
  /** the graphs of a vector as one record (they stay what they are) */
  explicit Batch(const std::vector<Graph>& graphs) {
    auto h = detail::handles(graphs);
    detail::check(gtnx_batch_from_graphs(h.data(), static_cast<int>(h.size()), &h_));
  }
  Batch(const Batch&) = delete;
  Batch& operator=(const Batch&) = delete;
  Batch(Batch&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  Batch& operator=(Batch&& o) noexcept {
    if (this != &o) {
      reset();
      h_ = o.h_;
      o.h_ = nullptr;
    }
    return *this;
  }
  ~Batch() { reset(); }

  /** CTC target acceptors (benchmarks/ctc.cpp:40-58) of label sequences, built on the device */
  static Batch ctcTargets(const std::vector<std::vector<int>>& targets, int blank = 0, bool calcGrad = true) {
    std::vector<int> flat, len;
    len.reserve(targets.size());
    size_t total = 0;
    for (auto& t : targets) total += t.size();
    flat.reserve(total);
    for (auto& t : targets) {
      flat.insert(flat.end(), t.begin(), t.end());
      len.push_back(static_cast<int>(t.size()));
    }
    return ctcTargets(flat.data(), len.data(), static_cast<int>(targets.size()), blank, calcGrad);
  }
  static Batch ctcTargets(const int* labels, const int* lengths, int n, int blank = 0, bool calcGrad = true) {
    Batch b;
    detail::check(gtnx_batch_ctc_targets(labels, lengths, n, blank, calcGrad, &b.h_));
    return b;
  }
  /** compose(forceAlign(target), transitions) of examples/asg.cpp:50-68 for every label sequence, built on the
   *  device; `transitions` in the arc layout of examples/asg.cpp:36-47 over `numLabels` labels */
  static Batch asgForceAlign(const int* labels, const int* lengths, int n, const Graph& transitions, int numLabels) {
    Batch b;
    detail::check(gtnx_batch_asg_force_align(labels, lengths, n, transitions.handle(), numLabels, &b.h_));
    return b;
  }
  /** n linear graphs over one device tensor [n][M][N] (see linearGraphs) */
  static Batch linear(int n, int M, int N, const void* deviceWeights, bool calcGrad = true, bool borrow = false) {
    Batch b;
    detail::check(gtnx_batch_linear(n, M, N, calcGrad, deviceWeights, borrow, &b.h_));
    return b;
  }

  int size() const {
    int n = 0;
    if (h_) detail::check(gtnx_batch_size(h_, &n));
    return n;
  }
  /** element i as an ordinary graph */
  Graph operator[](int i) const {
    gtnx_graph_t g;
    detail::check(gtnx_batch_get(h_, i, &g));
    return Graph::fromHandle(g);
  }
  /** item() of every element */
  std::vector<float> items() const {
    std::vector<float> v(static_cast<size_t>(size()));
    if (!v.empty()) detail::check(gtnx_batch_items(h_, v.data()));
    return v;
  }
  void itemsToDevice(void* deviceOut) const { detail::check(gtnx_batch_items_device(h_, deviceOut)); }
  /** where the elements' gradients should end up (element i at deviceOut + offsets[i] floats): before backward a
   *  hint that lets kernels store there directly, afterwards the gather */
  void bindGrads(void* deviceOut, const int64_t* offsets) { detail::check(gtnx_batch_grads_bind_device(h_, deviceOut, offsets)); }
  void gradsToDevice(void* deviceOut, const int64_t* offsets) const {
    detail::check(gtnx_batch_grads_device(h_, deviceOut, offsets));
  }
  gtnx_batch_t handle() const { return h_; }
  static Batch fromHandle(gtnx_batch_t h) {
    Batch b;
    b.h_ = h;
    return b;
  }

 private:
  void reset() {
    if (h_) gtnx_batch_destroy(h_);
    h_ = nullptr;
  }
  gtnx_batch_t h_ = nullptr;
};

namespace detail {
template <class F>
Batch batchUnary(F f, const Batch& a) {
  gtnx_batch_t out;
  check(f(a.handle(), &out));
  return Batch::fromHandle(out);
}
template <class F>
Batch batchBinary(F f, const Batch& a, const Batch& b) {
  gtnx_batch_t out;
  check(f(a.handle(), b.handle(), &out));
  return Batch::fromHandle(out);
}
} // namespace detail

// (in gtn::batched like the vector forms: the plain names stay un-overloaded, reference callers
//  pass them as function pointers -- parallelMap(compose, a, b))
namespace batched {
inline Batch negate(const Batch& a) { return detail::batchUnary(&gtnx_batch_negate, a); }
inline Batch add(const Batch& a, const Batch& b) { return detail::batchBinary(&gtnx_batch_add, a, b); }
inline Batch subtract(const Batch& a, const Batch& b) { return detail::batchBinary(&gtnx_batch_subtract, a, b); }
// ... with the values written straight into device memory of the caller's (which must outlive the result)
inline Batch subtract(const Batch& a, const Batch& b, void* itemsDevice) {
  gtnx_batch_t out;
  detail::check(gtnx_batch_subtract_into(a.handle(), b.handle(), itemsDevice, &out));
  return Batch::fromHandle(out);
}
inline Batch compose(const Batch& a, const Batch& b) { return detail::batchBinary(&gtnx_batch_compose, a, b); }
inline Batch intersect(const Batch& a, const Batch& b) { return detail::batchBinary(&gtnx_batch_intersect, a, b); }
inline Batch forwardScore(const Batch& a) { return detail::batchUnary(&gtnx_batch_forward_score, a); }
inline Batch viterbiScore(const Batch& a) { return detail::batchUnary(&gtnx_batch_viterbi_score, a); }
inline Batch viterbiPath(const Batch& a) { return detail::batchUnary(&gtnx_batch_viterbi_path, a); }
inline void backward(const Batch& a, bool retainGraph = false) { detail::check(gtnx_batch_backward(a.handle(), retainGraph)); }
} // namespace batched

} // namespace gtn
