/*
 * gtn_amd.h -- C ABI of the MI355X-native WFST engine (libgtn_amd.so).
 *
 * The reference (facebookresearch/gtn) has no FFI layer of its own: its
 * language binding (bindings/python/gtn/_*.cpp, pybind11) binds the public C++
 * API of libgtn directly.  This header is therefore the C restatement of that
 * public API for the hot path, one entry point per reference member/function,
 * each citing what it replaces.  Everything crossing the boundary is a plain
 * pointer, an integer or an opaque handle; no C++/HIP/torch types.
 *
 * Conventions
 *  - every call returns gtnx_status_t; GTNX_OK == 0.  On failure the message is
 *    available (thread-local) from gtnx_last_error().  Status codes map 1:1 to
 *    the exception types the reference throws (see below); the header-only C++
 *    shim in include/gtn/ rethrows them.
 *  - gtnx_graph_t is an owning reference to a graph *value handle* with the
 *    reference's aliasing semantics (gtn/graph.h:461-464): gtnx_graph_copy()
 *    aliases structure, weights and grad; gtnx_graph_deep_copy() detaches.
 *  - the "_n" forms take arrays of n handles and run ONE batched device launch
 *    per kernel for the whole array.  They are the device analogue of the
 *    reference's batch entry points: the std::vector<Graph> overloads of the
 *    Python binding (bindings/python/gtn/_functions.cpp:84-135) which call
 *    parallelMap (gtn/parallel/parallel_map.h:153-188).  An input array of
 *    length 1 is broadcast, as parallelMap does (parallel_map.h:77-89).
 *  - device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()
 *    on a ROCm tensor); a stream is a hipStream_t passed as void*.
 */
#ifndef GTN_AMD_H
#define GTN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int gtnx_status_t;
#define GTNX_OK 0
#define GTNX_INVALID_ARGUMENT 1 /* std::invalid_argument  (shortest.cpp:149-152, autograd.cpp:43-45, graph.cpp:70-73) */
#define GTNX_LOGIC_ERROR 2      /* std::logic_error       (graph.cpp:82-95, functions.cpp:19-21,33-35,49-51) */
#define GTNX_RUNTIME_ERROR 3    /* std::runtime_error     (parallel_map.h:85-88) */
#define GTNX_OUT_OF_RANGE 4     /* std::out_of_range      (bad node / arc index; the reference only asserts) */
#define GTNX_DEVICE_ERROR 5     /* HIP failure or no usable gfx950 device; std::runtime_error in the C++ shim */

typedef struct gtnx_graph_s* gtnx_graph_t;

#define GTNX_EPSILON (-1) /* gtn/graph.h:21 */

/* ------------------------------------------------------------------ runtime */
const char* gtnx_last_error(void);
const char* gtnx_version(void);
/* "hip:gfx950" for libgtn_amd.so; the reference-backed test shim
 * (oracle/_ref/libgtn_ref.so) answers "reference-cpu". */
const char* gtnx_backend(void);
int gtnx_device_count(void);                /* 0 when no GPU is visible */
/* The device of the CALLING THREAD (hipSetDevice + the engine's per-device context: stream, memory pools).  A thread
 * that never chose is on the process default: the device of the first gtnx_set_device call, else the device the
 * first calling thread ALREADY has with HIP (a rank that only did torch.cuda.set_device(k) works on k and stays on k),
 * never a silent 0.  Every entry point tells HIP the engine's device again (a hipSetDevice by the host framework on the
 * same thread is not trusted to have been undone).  One rank per GPU (torch.distributed) sets it once; a host that
 * drives several GPUs gives each its own thread (gtn::parallelMapSharded,
 * include/gtn/parallel.h; the pool threads of a parallelMap are put on the device of the thread that called it).
 * Graphs live on the device they were made on: using one from a thread on another device is GTNX_INVALID_ARGUMENT. */
gtnx_status_t gtnx_set_device(int device);
gtnx_status_t gtnx_get_device(int* device);
gtnx_status_t gtnx_set_stream(void* hip_stream); /* NULL = the engine's own stream */
gtnx_status_t gtnx_synchronize(void);
/* How compose / intersect of an implicit emissions chain with an epsilon-free graph treat their
 * result, for the calling thread.  0: build it, unless the batch would not fit in memory.
 * 1: keep it symbolic whenever eligible.  2: keep it symbolic when the per-utterance sweep kernels
 * apply (partner of <= 512 nodes and <= 4 arcs per node: CTC targets).  -1 (default, "nobody asked"): as 2
 * when the partner is a graph the caller built on the host, else as 0.  The mode of the thread that CALLS
 * gtn::parallelMap goes with the tasks (include/gtn/parallel.h sets it on the pool's threads; a call made inside a
 * region runs under the mode it was recorded with -- 0 builds every composition there too).  A symbolic result supports
 * forwardScore / viterbiScore / viterbiPath and backward through them; any other use builds it.  What
 * a symbolic result does not have is its OWN gradient graph (Graph::grad() of the composition), so
 * this is a hint for criteria that never read it.  `previous` (may be NULL) receives the old mode.
 * The environment variable GTNX_LAZY_COMPOSE (0 / 1 / 2) overrides the hint. */
gtnx_status_t gtnx_compose_mode(int mode, int* previous);
/* bytes currently held by the engine's device arena pool / bytes in use */
gtnx_status_t gtnx_memory_stats(uint64_t* reserved, uint64_t* in_use);
/* Takes apart EVERY thread's deferred garbage (also that of threads which never come to a reclamation point), waits
 * for the stream and gives the pooled device and pinned blocks back to the HIP runtime; also what an allocation does
 * by itself before it reports out-of-memory. */
gtnx_status_t gtnx_empty_cache(void);
/* Destroys what the CALLING THREAD has let go of, or built for others and nobody refers to any more, since its last
 * reclamation point (released handles, finished tapes, the recorded calls of a parallelMap region); never waits for
 * the GPU, cheap when nothing is pending.  Every thread takes apart what it allocated: the engine reclaims by itself
 * wherever a thread would wait for the device and whenever a thread's own list stands for more than 8 192 graphs
 * (GTNX_DEFER_FULL); the pool threads of include/gtn/parallel.h call this when their share of a region is done.  A
 * thread's list takes at most 16 384 objects from OTHER threads: beyond that the thread that lets go destroys. */
gtnx_status_t gtnx_reclaim(void);
/* The calling thread is one of several host threads mapping per-graph functions over a batch
 * (gtn::parallelMap, parallel/parallel_map.h:153-188) from here until gtnx_parallel_leave: its
 * function calls (negate .. viterbiPath, backward) are DEFERRED -- they return placeholder handles at
 * once, no thread waits for another -- and gtnx_parallel_flush, called by the thread that joins the
 * region (parallel_map.h:182-186), runs the calls of all the region's threads as ONE batched launch per
 * function; a criterion step over CTC-shaped targets and linear emission graphs runs as the batch records
 * below.  Results per graph are unchanged: a placeholder that is looked at before the join (sizes, arcs,
 * item(), a gradient) runs what it depends on right then, and a graph that is changed while a deferred call
 * still reads it has those calls run first.  What moves is when an error surfaces: at the first look at the
 * result, or from gtnx_parallel_flush (parallelMap rethrows after its join either way).  setWeights inside
 * a region copies a host source at the call (graph.cpp:179-181) and reads a DEVICE source at the join, with one
 * launch for the whole region: LIFETIME REQUIREMENT -- a device buffer handed to setWeights inside a parallelMap
 * task must stay allocated and unchanged until that parallelMap call returns (rows of a tensor that outlives the
 * call, as in pytorch_loss.py:46-71, are; a buffer the task itself frees or overwrites is not).
 * GTNX_REGION_EAGER_WEIGHTS=1 copies at the call instead, like the reference, one launch per call.  Inputs of the
 * recorded calls stay referenced until the results of the thread that recorded them are released.
 * Made by include/gtn/parallel.h; a hint -- without it every call is a batch of one. */
gtnx_status_t gtnx_parallel_enter(void);
gtnx_status_t gtnx_parallel_leave(void);
gtnx_status_t gtnx_parallel_flush(void);

/* ------------------------------------------------------------------ Graph
 * class Graph, gtn/graph.h:75-415 */
gtnx_status_t gtnx_graph_create(int calc_grad, gtnx_graph_t* out);          /* graph.h:89  */
gtnx_status_t gtnx_graph_copy(gtnx_graph_t g, gtnx_graph_t* out);           /* copy-ctor: alias */
gtnx_status_t gtnx_graph_deep_copy(gtnx_graph_t g, gtnx_graph_t* out);      /* graph.h:150 */
gtnx_status_t gtnx_graph_destroy(gtnx_graph_t g);
gtnx_status_t gtnx_graph_add_node(gtnx_graph_t g, int start, int accept, int* id);      /* graph.h:97 */
gtnx_status_t gtnx_graph_add_arc(gtnx_graph_t g, int src, int dst, int ilabel,
                                 int olabel, float weight, int* id);                    /* graph.h:118-123 */
/* bulk forms of the two above (same order semantics as n successive calls) */
gtnx_status_t gtnx_graph_add_nodes(gtnx_graph_t g, int n, const uint8_t* start,
                                   const uint8_t* accept);
gtnx_status_t gtnx_graph_add_arcs(gtnx_graph_t g, int n, const int* src,
                                  const int* dst, const int* ilabel,
                                  const int* olabel, const float* weight /* NULL = 0 */);
gtnx_status_t gtnx_graph_num_nodes(gtnx_graph_t g, int64_t* out);   /* graph.h:130 */
gtnx_status_t gtnx_graph_num_arcs(gtnx_graph_t g, int64_t* out);    /* graph.h:126 */
gtnx_status_t gtnx_graph_num_start(gtnx_graph_t g, int64_t* out);   /* graph.h:134 */
gtnx_status_t gtnx_graph_num_accept(gtnx_graph_t g, int64_t* out);  /* graph.h:138 */
gtnx_status_t gtnx_graph_item(gtnx_graph_t g, float* out);          /* graph.h:143; sync point */
gtnx_status_t gtnx_graph_arc_sort(gtnx_graph_t g, int olabel);      /* graph.h:158 */
gtnx_status_t gtnx_graph_mark_arc_sorted(gtnx_graph_t g, int olabel); /* graph.h:166 */
gtnx_status_t gtnx_graph_ilabel_sorted(gtnx_graph_t g, int* out);   /* graph.h:178 */
gtnx_status_t gtnx_graph_olabel_sorted(gtnx_graph_t g, int* out);   /* graph.h:186 */
/* weights(): a pointer into the live host buffer of numArcs floats (graph.h:194-204).
 * Forces a device->host sync when the weights were produced on the GPU; the
 * non-const form marks the host copy authoritative. */
gtnx_status_t gtnx_graph_weights(gtnx_graph_t g, int mutable_, float** out);
gtnx_status_t gtnx_graph_get_weights(gtnx_graph_t g, float* out);   /* copy-out of the above */
/* graph.h:210; `weights` may also be a DEVICE address (detected: hipPointerGetAttributes) -- then as below.  Copies at
 * the call, except inside a parallelMap region with a device source: see gtnx_parallel_enter for the lifetime rule. */
gtnx_status_t gtnx_graph_set_weights(gtnx_graph_t g, const float* weights);
/* the same copy from a DEVICE buffer of numArcs floats (no host round-trip;
 * replaces pytorch_loss.py:53-61's inputs.cpu() + set_weights(data_ptr)) */
gtnx_status_t gtnx_graph_set_weights_device(gtnx_graph_t g, const void* device_weights);
/* device address of the weight buffer (numArcs floats), uploading if needed */
gtnx_status_t gtnx_graph_weights_device(gtnx_graph_t g, void** out);
gtnx_status_t gtnx_graph_labels_to_array(gtnx_graph_t g, int* out, int ilabel); /* graph.h:219 */
/* node accessors, graph.h:330-376 */
gtnx_status_t gtnx_graph_get_start(gtnx_graph_t g, int* out);   /* numStart ints  */
gtnx_status_t gtnx_graph_get_accept(gtnx_graph_t g, int* out);  /* numAccept ints */
gtnx_status_t gtnx_graph_is_start(gtnx_graph_t g, int node, int* out);
gtnx_status_t gtnx_graph_is_accept(gtnx_graph_t g, int node, int* out);
gtnx_status_t gtnx_graph_make_accept(gtnx_graph_t g, int node);
gtnx_status_t gtnx_graph_num_out(gtnx_graph_t g, int node, int64_t* out);
gtnx_status_t gtnx_graph_num_in(gtnx_graph_t g, int node, int64_t* out);
gtnx_status_t gtnx_graph_get_out(gtnx_graph_t g, int node, int* out); /* numOut arc ids, list order */
gtnx_status_t gtnx_graph_get_in(gtnx_graph_t g, int node, int* out);  /* numIn arc ids, list order  */
/* arc accessors, graph.h:384-414 (bulk; any pointer may be NULL) */
gtnx_status_t gtnx_graph_get_arcs(gtnx_graph_t g, int* src, int* dst, int* ilabel, int* olabel);
gtnx_status_t gtnx_graph_get_arc(gtnx_graph_t g, int arc, int* src, int* dst,
                                 int* ilabel, int* olabel, float* weight);
gtnx_status_t gtnx_graph_set_weight(gtnx_graph_t g, int arc, float weight);
/* autograd members, graph.h:244-321 */
gtnx_status_t gtnx_graph_calc_grad(gtnx_graph_t g, int* out);
gtnx_status_t gtnx_graph_set_calc_grad(gtnx_graph_t g, int calc_grad);
gtnx_status_t gtnx_graph_is_grad_available(gtnx_graph_t g, int* out);
gtnx_status_t gtnx_graph_grad(gtnx_graph_t g, gtnx_graph_t* out);  /* new handle aliasing the grad graph */
gtnx_status_t gtnx_graph_zero_grad(gtnx_graph_t g);
gtnx_status_t gtnx_graph_add_grad(gtnx_graph_t g, const float* host_grad, int64_t n); /* graph.h:244-250 */
gtnx_status_t gtnx_graph_add_grad_graph(gtnx_graph_t g, gtnx_graph_t other);           /* graph.h:256 */
gtnx_status_t gtnx_graph_id(gtnx_graph_t g, uintptr_t* out);                            /* graph.h:281 */
/* Graph(GradFunc, inputs), graph.h:76-78: a user-defined differentiable op.
 * grad_fn(ctx, inputs, n_inputs, deltas) is called by gtnx_backward on the host;
 * ctx_free(ctx) when the op is released (either may be NULL). */
typedef gtnx_status_t (*gtnx_grad_fn)(void* ctx, gtnx_graph_t* inputs, int n_inputs, gtnx_graph_t deltas);
gtnx_status_t gtnx_graph_create_op(gtnx_graph_t* inputs, int n_inputs, gtnx_grad_fn grad_fn,
                                   void* ctx, void (*ctx_free)(void*), gtnx_graph_t* out);
gtnx_status_t gtnx_graph_num_inputs(gtnx_graph_t g, int64_t* out);                      /* graph.h:303 */
gtnx_status_t gtnx_graph_get_input(gtnx_graph_t g, int i, gtnx_graph_t* out);           /* graph.h:303 (new handle) */
gtnx_status_t gtnx_graph_set_inputs(gtnx_graph_t g, const gtnx_graph_t* inputs, int n); /* graph.h:311 */
gtnx_status_t gtnx_graph_set_grad_fn(gtnx_graph_t g, gtnx_grad_fn grad_fn, void* ctx,
                                     void (*ctx_free)(void*));                          /* graph.h:293 */
gtnx_status_t gtnx_graph_has_grad_fn(gtnx_graph_t g, int* out);                         /* graph.h:286 */

/* ------------------------------------------------------------------ creations
 * gtn/creations.h:25,32 */
gtnx_status_t gtnx_scalar_graph(float value, int calc_grad, gtnx_graph_t* out);
gtnx_status_t gtnx_linear_graph(int M, int N, int calc_grad, gtnx_graph_t* out);
/* B linear graphs whose weights are COPIED from one contiguous device tensor
 * [B][M][N] (the (B,T,C) emissions of pytorch_loss.py:46-71), one launch. */
gtnx_status_t gtnx_linear_graph_n(int B, int M, int N, int calc_grad,
                                  const void* device_weights, gtnx_graph_t* out /* B handles */);
/* The same without the copy: the graphs' weights ARE the caller's tensor (what a
 * torch.autograd.Function holds on to between forward and backward anyway,
 * pytorch_loss.py:46-71).  The caller keeps it alive and unchanged while any of the
 * handles -- or a result computed from them -- is in use. */
gtnx_status_t gtnx_linear_graph_borrow_n(int B, int M, int N, int calc_grad,
                                         const void* device_weights, gtnx_graph_t* out /* B handles */);

/* ------------------------------------------------------------------ functions
 * gtn/functions.h:19-152.  Single-graph forms, then batched forms. */
gtnx_status_t gtnx_negate(gtnx_graph_t g, gtnx_graph_t* out);                   /* functions.cpp:18-30 */
gtnx_status_t gtnx_add(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out);      /* functions.cpp:32-46 */
gtnx_status_t gtnx_subtract(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out); /* functions.cpp:48-64 */
gtnx_status_t gtnx_compose(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out);  /* functions.cpp:225-237 */
gtnx_status_t gtnx_intersect(gtnx_graph_t a, gtnx_graph_t b, gtnx_graph_t* out);/* functions.cpp:239-251 */
gtnx_status_t gtnx_forward_score(gtnx_graph_t g, gtnx_graph_t* out);            /* functions.cpp:320-322 */
gtnx_status_t gtnx_viterbi_score(gtnx_graph_t g, gtnx_graph_t* out);            /* functions.cpp:324-326 */
gtnx_status_t gtnx_viterbi_path(gtnx_graph_t g, gtnx_graph_t* out);             /* functions.cpp:328-330 */

gtnx_status_t gtnx_negate_n(const gtnx_graph_t* g, int n, gtnx_graph_t* out);
gtnx_status_t gtnx_add_n(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, int nb, gtnx_graph_t* out);
gtnx_status_t gtnx_subtract_n(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, int nb, gtnx_graph_t* out);
gtnx_status_t gtnx_compose_n(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, int nb, gtnx_graph_t* out);
gtnx_status_t gtnx_intersect_n(const gtnx_graph_t* a, int na, const gtnx_graph_t* b, int nb, gtnx_graph_t* out);
gtnx_status_t gtnx_forward_score_n(const gtnx_graph_t* g, int n, gtnx_graph_t* out);
gtnx_status_t gtnx_viterbi_score_n(const gtnx_graph_t* g, int n, gtnx_graph_t* out);
gtnx_status_t gtnx_viterbi_path_n(const gtnx_graph_t* g, int n, gtnx_graph_t* out);
/* item() of n single-arc graphs with one device->host copy (batched graph.h:143) */
gtnx_status_t gtnx_items_n(const gtnx_graph_t* g, int n, float* out);
/* the same, written to a DEVICE buffer of n floats without a host sync */
gtnx_status_t gtnx_items_device_n(const gtnx_graph_t* g, int n, void* device_out);
/* grad().weights() of n graphs gathered into ONE device buffer, graph i at
 * byte offset offsets[i]*4 (replaces pytorch_loss.py:94-102's per-sample
 * weights_to_numpy + torch.from_numpy + .to(device)) */
gtnx_status_t gtnx_grads_device_n(const gtnx_graph_t* g, int n, void* device_out, const int64_t* offsets);
/* Called BEFORE backward: names the device buffer gtnx_grads_device_n will be asked to
 * fill, so that kernels which produce graph i's first gradient may store it at
 * device_out + offsets[i] directly (the later gtnx_grads_device_n then finds it in
 * place and copies nothing).  A hint: results are the same without it. */
gtnx_status_t gtnx_grads_bind_device_n(const gtnx_graph_t* g, int n, void* device_out, const int64_t* offsets);

/* ------------------------------------------------------------------ rational operations
 * gtn/functions.h:45-123, functions.cpp:66-223 -- built on the device: the output's arrays are the inputs'
 * arrays copied with node offsets, epsilon connectors written from the inputs' start / accept lists,
 * adjacency lists rebuilt by a stable sort of the arc ids.  Node and arc ids as the reference numbers them;
 * gradients of the inputs are slices of the output's. */
gtnx_status_t gtnx_clone(gtnx_graph_t g, int projection /* 0 none, 1 input, 2 output */, gtnx_graph_t* out); /* functions.cpp:66-92 */
gtnx_status_t gtnx_concat(const gtnx_graph_t* g, int n, gtnx_graph_t* out);                                  /* functions.cpp:97-153 */
gtnx_status_t gtnx_closure(gtnx_graph_t g, gtnx_graph_t* out);                                               /* functions.cpp:155-189 */
gtnx_status_t gtnx_union(const gtnx_graph_t* g, int n, gtnx_graph_t* out);                                   /* functions.cpp:191-223 */
/* remove(g, ilabel, olabel): arcs carrying the label pair are contracted (breadth-first over them from every kept
 * node, the reference's node / arc order); weights are dropped and backward through the result throws, as there */
gtnx_status_t gtnx_remove(gtnx_graph_t g, int ilabel, int olabel, gtnx_graph_t* out);                        /* functions.cpp:253-318 */

/* ------------------------------------------------------------------ batch records
 * B graphs held as ONE object: what gtn::parallelMap over the per-graph functions
 * (parallel/parallel_map.h:153-188; benchmarks/ctc.cpp:150-165) returns when nobody looks at
 * the elements -- one record and one tape node per call instead of B graphs.  Elements can be
 * taken out as ordinary graphs at any time (gtnx_batch_get): the per-graph expression is then
 * built once, tape included, and everything the batch functions do not cover natively runs
 * through the vector functions above.  Results per element equal the per-graph functions'. */
typedef struct gtnx_batch_s* gtnx_batch_t;
gtnx_status_t gtnx_batch_from_graphs(const gtnx_graph_t* g, int n, gtnx_batch_t* out);
/* the CTC target acceptor of benchmarks/ctc.cpp:40-58 / examples/ctc.cpp:21-41 for n label
 * sequences (labels back to back, lengths[i] each), built on the device */
gtnx_status_t gtnx_batch_ctc_targets(const int* labels, const int* lengths, int n, int blank, int calc_grad,
                                     gtnx_batch_t* out);
/* compose(forceAlign(target), transitions) of examples/asg.cpp:50-68 for n label sequences, built
 * on the device: `transitions` must have the arc layout of examples/asg.cpp:36-47 over n_labels labels
 * (arc i: start -> label i; arc n_labels + i * n_labels + j: label j -> label i); its weights are
 * gathered into the acceptors' arcs and their gradients are added back into its gradient */
gtnx_status_t gtnx_batch_asg_force_align(const int* labels, const int* lengths, int n, gtnx_graph_t transitions,
                                         int n_labels, gtnx_batch_t* out);
/* n linear graphs (creations.cpp:20-33) over one device tensor [n][M][N]; borrow != 0: read in
 * place (see gtnx_linear_graph_borrow_n) */
gtnx_status_t gtnx_batch_linear(int n, int M, int N, int calc_grad, const void* device_weights, int borrow,
                                gtnx_batch_t* out);
gtnx_status_t gtnx_batch_destroy(gtnx_batch_t b);
gtnx_status_t gtnx_batch_size(gtnx_batch_t b, int* out);
gtnx_status_t gtnx_batch_get(gtnx_batch_t b, int i, gtnx_graph_t* out);               /* new handle */
gtnx_status_t gtnx_batch_negate(gtnx_batch_t a, gtnx_batch_t* out);                   /* functions.cpp:18-30 */
gtnx_status_t gtnx_batch_add(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out);      /* functions.cpp:32-46 */
gtnx_status_t gtnx_batch_subtract(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out); /* functions.cpp:48-64 */
/* subtract with the n result values written straight into the caller's device memory (borrowed: it must outlive the
 * result batch) -- a criterion's losses land where the caller wants them without a copy; gtnx_batch_items_device to the
 * same address is then a no-op */
gtnx_status_t gtnx_batch_subtract_into(gtnx_batch_t a, gtnx_batch_t b, void* items_device, gtnx_batch_t* out);
gtnx_status_t gtnx_batch_compose(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out);  /* functions.cpp:225-237 */
gtnx_status_t gtnx_batch_intersect(gtnx_batch_t a, gtnx_batch_t b, gtnx_batch_t* out);/* functions.cpp:239-251 */
gtnx_status_t gtnx_batch_forward_score(gtnx_batch_t a, gtnx_batch_t* out);            /* functions.cpp:320-322 */
gtnx_status_t gtnx_batch_viterbi_score(gtnx_batch_t a, gtnx_batch_t* out);            /* functions.cpp:324-326 */
gtnx_status_t gtnx_batch_viterbi_path(gtnx_batch_t a, gtnx_batch_t* out);             /* functions.cpp:328-330 */
gtnx_status_t gtnx_batch_backward(gtnx_batch_t a, int retain_graph);                  /* autograd.cpp:17-67 */
gtnx_status_t gtnx_batch_items(gtnx_batch_t a, float* out);                           /* graph.h:143, n floats */
gtnx_status_t gtnx_batch_items_device(gtnx_batch_t a, void* device_out);
/* as gtnx_grads_bind_device_n / gtnx_grads_device_n, for the elements of a batch */
gtnx_status_t gtnx_batch_grads_bind_device(gtnx_batch_t a, void* device_out, const int64_t* offsets);
gtnx_status_t gtnx_batch_grads_device(gtnx_batch_t a, void* device_out, const int64_t* offsets);

/* ------------------------------------------------------------------ autograd
 * gtn/autograd.h:27,37 */
gtnx_status_t gtnx_backward(gtnx_graph_t g, int retain_graph);
gtnx_status_t gtnx_backward_with_grad(gtnx_graph_t g, gtnx_graph_t grad, int retain_graph);
gtnx_status_t gtnx_backward_n(const gtnx_graph_t* g, int n, int retain_graph);

/* ------------------------------------------------------------------ utils
 * gtn/utils.h:23-60 (test fixtures of the parity suite; host-side) */
gtnx_status_t gtnx_equal(gtnx_graph_t a, gtnx_graph_t b, int* out);
gtnx_status_t gtnx_isomorphic(gtnx_graph_t a, gtnx_graph_t b, int* out);

/* ------------------------------------------------------------------ formats
 * gtn/utils.h:115-150, utils.cpp:152-225: the binary graph format -- int32 {numNodes, numArcs, numStart, numAccept},
 * the start nodes, the accept nodes, {src, dst, ilabel, olabel} per arc, float32 weights.  `data` is the whole
 * file image: the arc table and the weights go to the device in ONE copy and are split into the graph's SoA
 * arrays there (adjacency lists built on the device as well); node and arc ids as load() numbers them. */
gtnx_status_t gtnx_graph_load_buffer(const void* data, size_t bytes, gtnx_graph_t* out);

/* ------------------------------------------------------------------ several GPUs, one process
 * The path shards by utterance (parallelMap has no cross-task communication, parallel/parallel_map.h:167-179): a host
 * that drives the GPUs of a node gives each device its own thread (gtnx_set_device; gtn::parallelMapSharded) and no
 * data-path collective is needed.  What is gathered afterwards goes over RCCL (xGMI), one communicator per device,
 * enqueued on each device's engine stream so that it orders with the kernels producing its inputs:
 *   all_gather   the per-utterance losses: device k contributes `count` floats at send[k] (ITS memory) and receives
 *                all n * count at recv[k], blocks in the order of `devices`
 *   all_reduce   the gradient of a graph every utterance shares (ASG transitions, criterion_test.cpp:289-305: the sum
 *                over utterances): bufs[k] (device k's partial sum, `count` floats) becomes the total, in place
 * librccl.so is looked up at first use; a communicator over ONE device needs none (gather = copy, reduce = nothing). */
typedef struct gtnx_comm_s* gtnx_comm_t;
gtnx_status_t gtnx_comm_create(const int* devices, int n, gtnx_comm_t* out);
gtnx_status_t gtnx_comm_destroy(gtnx_comm_t c);
gtnx_status_t gtnx_comm_size(gtnx_comm_t c, int* n);
gtnx_status_t gtnx_comm_all_gather_f32(gtnx_comm_t c, const void* const* send, void* const* recv, int64_t count);
gtnx_status_t gtnx_comm_all_reduce_sum_f32(gtnx_comm_t c, void* const* bufs, int64_t count);

/* ------------------------------------------------------------------ profiling
 * hipEvent timing of the engine's own kernel launches on the launch stream
 * (what bench.py's roofline leg reads).  name is a kernel family, e.g.
 * "forward_score", "compose", "forward_score_grad", "linear_forward". */
gtnx_status_t gtnx_prof_enable(int on);
gtnx_status_t gtnx_prof_reset(void);
gtnx_status_t gtnx_prof_get(const char* name, double* total_ms, int64_t* launches,
                            double* algorithmic_bytes);
/* Diagnostics: viterbiPath of SYMBOLIC products with a dense partner (max-plus walk) -- how many best paths ran through
 * a state with two exactly equal finite candidates since the process started (`seen`), and how many of those kept the
 * first maximum in in-row order because the product was too large to build and replay the reference's queue on
 * (`unresolved`; shortest.cpp:215-218, INTEGRATION.md "Where results can differ" 3). */
gtnx_status_t gtnx_debug_viterbi_ties(int64_t* seen, int64_t* unresolved);
/* Diagnostics (host only, no GPU needed): how exact ties in products of an emission chain with `g` are decided without
 * building the lattice when g is exactly a CTC target acceptor (benchmarks/ctc.cpp:40-58) with label-sorted out-lists:
 * queue_rank[n] = position of node n in the reference's queue within every layer of the lattice (viterbiPath keeps the
 * arc relaxed first, shortest.cpp:208-224), creation_rank[n] = its position in compose's creation order (in-list and
 * accept-list order: viterbiScore's gradient, shortest.cpp:118-127, :148-160); numNodes() ints each.  *applies = 0:
 * g is not such a graph (nothing written; tied utterances then take the built lattice). */
gtnx_status_t gtnx_debug_tie_ranks(gtnx_graph_t g, int* queue_rank, int* creation_rank, int* applies);
/* names, '\n'-separated, of the families seen since the last reset */
gtnx_status_t gtnx_prof_names(char* buf, size_t cap);
/* Diagnostics: which kernel family would score the SYMBOLIC chain product `g` (a compose / intersect result kept
 * symbolic, gtnx_compose_mode) -- forwardScore (tropical = 0) or viterbiScore / viterbiPath (tropical = 1).  The
 * decision table is gtn_amd/csrc/ops_symbolic.cpp; *route is an index into "band", "pair", "dense_mfma", "dense",
 * "maxplus", "walk" (gtnx_debug_route_name), or -1 when `g` is not a symbolic product.  No reference analogue. */
gtnx_status_t gtnx_debug_symbolic_route(gtnx_graph_t g, int tropical, int* route);
gtnx_status_t gtnx_debug_route_name(int route, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* GTN_AMD_H */
