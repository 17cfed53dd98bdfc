// region.h -- deferred execution of the per-graph functions inside a parallelMap region.
//
// The reference runs a criterion as parallelMap over per-utterance lambdas, each calling the per-graph
// functions (benchmarks/ctc.cpp:136-168; gtn/parallel/parallel_map.h:153-188).  On this engine a host thread
// that announced such a region (gtnx_parallel_enter, made by include/gtn/parallel.h) does not run these calls:
// negate .. viterbiPath return a PLACEHOLDER handle and backward returns at once; the calls are queued per
// thread (no lock, no rendezvous) and the region's join (gtnx_parallel_flush) runs them grouped by function
// and dependency depth -- one batched launch per group, and where the group is a whole criterion step over
// CTC-shaped targets and linear emission graphs, as the batch records of batch.h (no per-utterance objects).
// A placeholder that is looked at earlier (sizes, arcs, item(), another eager function) runs what it depends
// on right then; a graph that is mutated while a queued call still reads it has those calls run first, so the
// results per graph are those of immediate execution.  What moves is WHEN an error surfaces: a call that the
// reference would have thrown from throws when its result is first looked at, or from the region's join
// (parallelMap rethrows its first exception after the join either way, parallel_map.h:182-186).
#pragma once

#include <atomic>
#include <exception>
#include <memory>

#include "batch.h"

namespace gtnx {

enum RegionOp : uint8_t {
  RO_NEG = 0,
  RO_ADD,
  RO_SUB,
  RO_COMPOSE,
  RO_INTERSECT,
  RO_FS,
  RO_VS,
  RO_VP,
  RO_BWD,
  RO_BWD_RETAIN,
  RO_COUNT
};

// one queued call
struct Pending {
  RegionOp op = RO_NEG;
  int depth = 1;              // 1 + the depth of the deepest input that was still queued at the call
  // the inputs as handed in (either may be a placeholder itself); released after the run.  (Empty: a default-
  // constructed Graph allocates its structure / weights / gradient state -- six allocations per queued call)
  Graph a{Graph::Empty{}}, b{Graph::Empty{}};
  std::atomic<int> state{0};  // 0 queued, 1 done, 2 failed
  std::exception_ptr err;
  // the result: a graph of its own (vector path), or element `idx` of a batch record; `res` is then made on
  // first demand (batch_get: the per-graph expression is built once for the whole record)
  Graph res{Graph::Empty{}};
  std::atomic<bool> has_res{false};
  BatchP batch;
  int idx = -1;
  int group = -1;             // scratch of one run
  int8_t mode = -1;           // compose / intersect: the compose mode to run under (-1: the engine's own policy,
                              // gtn_amd.h gtnx_compose_mode -- symbolic for small partners built on the host)
};

bool region_active();  // the calling thread is inside a region (and not running queued calls itself)
void region_enter();
void region_leave();   // hands the thread's queue to the region
void region_flush();   // the join: runs everything handed in (and the caller's own queue); throws the first error

Graph region_record(RegionOp op, const Graph& a, const Graph* b);  // -> placeholder
// The vector forms of the C ABI (gtnx_*_n, n >= 2) run through the same machinery AT ONCE: the n calls are
// recorded and joined on the spot, so a vector of CTC-shaped targets and linear emission graphs takes the batch
// records too and the n results are placeholders over one record instead of n graphs.  `na` / `nb` of 1 broadcast
// (parallel_map.h:77-89).  compose / intersect keep the calling thread's compose mode (gtnx_compose_mode).
void region_run_vector(RegionOp op, Graph* const* a, int na, Graph* const* b, int nb, Graph* out /* max(na, nb) */);
void region_run_backward_vector(Graph* const* roots, int n, bool retain);
void region_record_backward(const Graph& root, bool retain);
Graph& region_value(Graph& placeholder);        // the graph behind a placeholder handle (runs what it needs)
bool region_item(Graph& placeholder, float* out);  // item() of a batch-record scalar without building its graph
// item() of n graphs, some of which may be placeholders over batch-record scalars (no graph is built for those):
// gathered to `dev_out` (n floats, device) with one launch.  false: not applicable (an input is not a one-arc graph
// resident as a scalar) -- the caller takes the ordinary path.
bool region_items_device(Graph* const* hs, int n, void* dev_out);
void region_before_mutation(Graph& g);          // runs queued calls that still read g
void region_sync_thread();                      // runs the calling thread's queue (gradient accessors)
// setWeights inside a region: host source copied to pinned staging now, device source read at the join
bool region_stage_weights(Graph& g, const float* p, bool device);
void region_trash(Graph* handle);               // gtnx_graph_destroy inside a region: handed over at leave

} // namespace gtnx
