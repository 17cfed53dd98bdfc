// region.h -- deferred execution of the per-graph functions inside a parallelMap region.
//
// The reference runs a criterion as parallelMap over per-utterance lambdas, each calling the per-graph
// functions (benchmarks/ctc.cpp:136-168; gtn/parallel/parallel_map.h:153-188).  On this engine a host thread
// that announced such a region (gtnx_parallel_enter, made by include/gtn/parallel.h) does not run these calls:
// negate .. viterbiPath return a PLACEHOLDER handle and backward returns at once.  The calls are recorded in
// the thread's own SLICE (no lock, no rendezvous) -- grouped by (dependency depth, function, compose mode) as
// they arrive, with everything the join would otherwise have to find out per call noted by the recording
// thread: which earlier call each input is the result of, the label sequence of a CTC-shaped target, where an
// emission graph's weights were staged.  The region's join (gtnx_parallel_flush) then runs one batched launch
// per group of the slices handed in; where the groups line up (the same lambda in every task) it touches the
// slices, not the calls: the inputs of a group are the RECORD an earlier group produced, leaves become the
// batch records of batch.h straight from the slices' digests, and a placeholder finds its result through its
// slice group (record + offset) when somebody asks.  Groups that do not line up are flattened and take the
// call-by-call path (value_of / as_batch below), with the same results.
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

struct SliceGroup;

// one queued call
struct Pending {
  RegionOp op = RO_NEG;
  int depth = 1;              // 1 + the depth of the deepest input that was still queued at the call
  // the inputs as handed in (either may be a placeholder itself); released after the run.  (Empty: a default-
  // constructed Graph allocates its structure / weights / gradient state -- six allocations per queued call)
  Graph a{Graph::Empty{}}, b{Graph::Empty{}};
  bool uses_counted = false;  // the inputs' pending_uses were raised for this call and not given back yet
  // 0 queued, 1 done, 2 failed -- of THIS call when it ran on its own (or failed); a call that ran with its
  // whole slice group reads the group's (st())
  std::atomic<int> state{0};
  std::exception_ptr err;
  // the result: a graph of its own, or element `idx` of a batch record; `res` is then made on first demand
  // (batch_get: the per-graph expression is built once for the whole record).  A call that ran with its slice
  // group has neither: record() finds the group's record and the call's element
  Graph res{Graph::Empty{}};
  std::atomic<bool> has_res{false};
  BatchP batch;
  int idx = -1;
  SliceGroup* sg = nullptr;   // the calls of one (depth, function, mode) recorded by one thread (region.cpp)
  int local = -1;             // position in it
  int8_t mode = -1;           // compose / intersect: the compose mode to run under (-1: the engine's own policy,
                              // gtn_amd.h gtnx_compose_mode -- symbolic for small partners built on the host)
  int st() const;                    // 0 queued, 1 done, 2 failed
  BatchP record(int* element) const; // the record this call's result is an element of (null: `res` / not run)
  void release_inputs();             // gives the inputs' pending_uses back and lets go of them
  void recycle();                    // back to the state of a fresh call (its slot is reused: region.cpp Slice::Chunk)
  ~Pending() { release_inputs(); }
};

bool region_active();  // the calling thread is inside a region (and not running queued calls itself)
void region_enter();
void region_leave();   // hands the thread's slice to the region
void region_flush();   // the join: runs everything handed in (and the caller's own slice); throws the first error

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
// setWeights inside a region: host source copied to pinned staging now; device source read at the join (the
// buffer must stay valid and unchanged until the parallelMap call returns: gtn_amd.h gtnx_graph_set_weights)
bool region_stage_weights(Graph& g, const float* p, bool device);
// gtnx_reclaim on a pool thread: what this thread built in earlier regions and nobody refers to any more comes
// home to be taken apart here (region.cpp: return to sender)
void region_reclaim_thread();
void destroy_handle(Graph* handle);             // capi.cpp: a handle is an entry of a slab of handles
void region_trash(Graph* handle);               // gtnx_graph_destroy inside a region: handed over at leave

} // namespace gtnx
