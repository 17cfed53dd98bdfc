// batch.h -- B graphs held as ONE record.
//
// The reference runs a criterion as parallelMap over per-utterance graphs
// (benchmarks/ctc.cpp:136-168, bindings/python/examples/pytorch_loss.py:46-102); the vector
// overloads of ops.h already turn that into one launch per function, but they still hand back B
// graph objects per call (structure + weights + autograd node each), and at C3 the host spent
// four times the GPU's time creating and destroying them.  A Batch is what such a call returns
// when nobody looks at the elements: one object, one tape node, dense device arrays.  Taking an
// element out (batch_get) builds the ordinary per-graph expression once, tape included, so
// anything the batch functions do not cover natively still works -- through the vector ops.
//
// Native (no per-element objects):   CTC target acceptors built on the device from label
// sequences, linear chains over one [B][M][C] tensor, their composition (kept symbolic),
// forwardScore of both, negate / add / subtract, backward, item / gradient gathers.
#pragma once

#include <memory>
#include <vector>

#include "ops.h"

namespace gtnx {

struct Batch;
using BatchP = std::shared_ptr<Batch>;

struct BatchOp {
  uint64_t seq = 0;
  std::vector<BatchP> inputs;
  virtual void backward(Batch& out) = 0;
  virtual ~BatchOp() {}
};

struct Batch {
  enum Kind { GRAPHS, CTC_TARGETS, LINEAR, PRODUCT, SCALAR };
  Kind kind = GRAPHS;
  int n = 0;
  bool calc_grad = false;

  // the elements as graphs: always there for GRAPHS, built on demand otherwise (then THEY carry the
  // gradients: batch-level results are pushed into them)
  bool materialised = false;
  std::vector<Graph> graphs;
  // CTC_TARGETS / LINEAR made FROM the caller's graphs (region.cpp: the leaves of a parallelMap region): `graphs`
  // are those graphs from the start, the native arrays describe the same values, and every gradient the batch
  // functions produce is pushed into the graphs (that is where the caller looks)
  bool leaf = false;
  // ... taken from the digests of a region's slices (region.cpp): when the record dies, the graphs of each range
  // go back to the thread that built them (give_back(home, graphs)) instead of being taken apart here
  struct Origin {
    std::shared_ptr<void> home;
    size_t begin, end;
  };
  std::vector<Origin> origins;
  void (*give_back)(const std::shared_ptr<void>& home, std::vector<Graph>* part) = nullptr;
  // SCALAR: the values on the host, fetched once when an element's item() is asked for (batch_item_host)
  std::vector<float> host_vals;
  bool host_vals_valid = false;
  // ... or on their way: copied to pinned memory BEHIND the launch that produced them (batch_prefetch_items), so
  // that reading a loss waits for the forward sweep only, not for whatever was queued after it (the backward
  // sweep of the same step: the host prepares the next step meanwhile)
  PinnedMemP host_pin;
  void* host_ev = nullptr;  // hipEvent_t

  // ---- CTC_TARGETS: label sequences back to back; records (BandNode, flags, sorted lists) on the device
  std::vector<int> labels, lab_off;  // lab_off[n + 1]
  int blank = 0;
  int max_label = -1, max_nodes = 0;
  DevMemP rec_mem;
  std::vector<size_t> rec_off;       // byte offset of element b's records
  // ... or force-alignment acceptors composed with an ASG transitions graph (examples/asg.cpp:50-68):
  // U + 1 nodes and 2U weighted arcs per sequence, weights gathered from the transitions on the device
  bool fal = false;
  Graph trans;                       // the transitions graph (its gradient is scattered back into it)
  int trans_labels = 0;
  std::vector<size_t> w_off, map_off;  // byte offsets of element b's arc weights / arc -> transitions arc map
  // ---- LINEAR: [n][M][C] device tensor (owned copy or the caller's)
  int M = 0, C = 0;
  DevMemP w_mem;
  float* w_dev = nullptr;
  std::shared_ptr<PendingCopy> w_pend;  // LINEAR over a region's staged weights: the values may not be at w_dev yet
                                        // (graph.h PendingCopy; batch.cpp linear_values settles, the band forward fuses)
  DevMemP nc_mem;                    // forwardScore of every chain + per-row log-sum-exps, left behind by a sweep
  float* nc_norm = nullptr;
  float* nc_rowlse = nullptr;
  // ---- PRODUCT: compose(fixed, chain) / compose(chain, fixed), never built
  BatchP fixed, chain;
  bool chain_first = false, intersect = false;
  // ---- SCALAR: one float per element
  DevMemP v_mem;
  float* v_dev = nullptr;

  // ---- autograd
  std::shared_ptr<BatchOp> op;
  bool tape_cleared = false;         // backward without retain went through here (autograd.cpp:48-51)
  DevMemP g_mem;                     // gradient, elements back to back at g_off[b] (floats)
  float* g_dev = nullptr;
  std::vector<int64_t> g_off;        // n + 1
  DevMemP dest_mem;                  // batch_grads_bind: where the first gradient should be written
  float* dest = nullptr;

  int64_t elem_size(int b) const;    // gradient floats of element b (arcs; an upper bound for CTC_TARGETS)
  ~Batch();                          // the element graphs go to the runtime's deferred list in chunks
};

BatchP batch_from_graphs(std::vector<Graph> gs);
BatchP batch_ctc_targets(const int* labels, const int* lengths, int n, int blank, bool calc_grad);
BatchP batch_asg_force_align(const int* labels, const int* lengths, int n, Graph& transitions, int n_labels);
BatchP batch_linear(int n, int M, int C, bool calc_grad, const void* dev, bool borrow);
BatchP batch_compose(const BatchP& a, const BatchP& b, bool intersect);
BatchP batch_shortest_distance(const BatchP& x, bool tropical);
BatchP batch_viterbi_path(const BatchP& x);
// items_dev (optional): device memory of the CALLER's that the n result values are written into directly (borrowed: it
// must outlive the result); a later batch_items_device to the same address copies nothing
BatchP batch_scalar(ScalarKind k, const BatchP& a, const BatchP& b, void* items_dev = nullptr);
void batch_backward(const BatchP& root, bool retain);
void batch_items_host(const BatchP& x, float* out);
void batch_items_device(const BatchP& x, void* dev_out);
void batch_grads_device(const BatchP& x, void* dev_out, const int64_t* offsets);
void batch_grads_bind(const BatchP& x, void* dev_out, const int64_t* offsets);
Graph batch_get(const BatchP& x, int i);
float batch_item_host(const BatchP& x, int i);  // x SCALAR and not materialised
void batch_prefetch_items(const BatchP& x);      // x SCALAR: start the device->host copy of its values now
// the caller's graphs as native leaves (nullptr when they do not qualify): acceptors that are each exactly
// ctcGraph(labels) of benchmarks/ctc.cpp:40-58 (Structure::ctc_labels) / linear chains of one shape whose
// weights are (made) one [n][M][C] device tensor
BatchP batch_ctc_targets_from_graphs(const std::vector<Graph>& gs);
BatchP batch_linear_from_graphs(const std::vector<Graph>& gs);
void batch_materialise(Batch& x);

} // namespace gtnx
