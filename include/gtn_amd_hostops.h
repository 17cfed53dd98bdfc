/* gtn_amd_hostops.h -- C entry points of libgtn_hostops.so (gtn_amd/hostops/): the HOST-side graph
 * builders, samplers and file formats of the reference's Python binding, for language bindings that sit
 * on the C ABI of gtn_amd.h.  Nothing here is on the hot path and nothing runs on the device: each call
 * is the header-only implementation of include/gtn (functions.h / rand.h / utils.h -- structure builders
 * with the reference's gradient slices) applied to gtnx_graph_t handles.  Replaces, for a binding:
 *   bindings/python/gtn/_functions.cpp:36-83,109-121,151-203,221-233  (concat, clone, closure, project_*,
 *                                                                      remove, union)
 *   bindings/python/gtn/_rand.cpp:19-41   (sample, rand_equivalent)
 *   bindings/python/gtn/_utils.cpp:22-41  (write_dot, load, save, savetxt, loadtxt)
 *   bindings/python/gtn/_graph.cpp:104-108 (__repr__)
 * Status codes and gtnx_graph_t are those of gtn_amd.h; gtnh_last_error() is thread-local.  Outputs are new
 * owning handles (gtnx_graph_destroy). */
#ifndef GTN_AMD_HOSTOPS_H
#define GTN_AMD_HOSTOPS_H

#include <stddef.h>

#include "gtn_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* gtnh_last_error(void);
int gtnh_clone(gtnx_graph_t g, int projection /* 0 none, 1 input, 2 output */, gtnx_graph_t* out); /* functions.cpp:66-96 */
int gtnh_concat(const gtnx_graph_t* graphs, int n, gtnx_graph_t* out);                          /* functions.cpp:112-163 */
int gtnh_closure(gtnx_graph_t g, gtnx_graph_t* out);                                            /* functions.cpp:165-199 */
int gtnh_union(const gtnx_graph_t* graphs, int n, gtnx_graph_t* out);                           /* functions.cpp:201-223 */
int gtnh_remove(gtnx_graph_t g, int ilabel, int olabel, gtnx_graph_t* out);                     /* functions.cpp:253-318 */
int gtnh_sample(gtnx_graph_t g, size_t max_length, gtnx_graph_t* out);                          /* rand.cpp:14-75 */
int gtnh_rand_equivalent(gtnx_graph_t a, gtnx_graph_t b, size_t num_samples, double tol, size_t max_length,
                         int* out);                                                            /* rand.cpp:77-126 */
int gtnh_load(const char* file, gtnx_graph_t* out);                                             /* utils.cpp:310-345 */
int gtnh_save(const char* file, gtnx_graph_t g);                                                /* utils.cpp:281-308 */
int gtnh_loadtxt(const char* file, gtnx_graph_t* out);                                          /* utils.cpp:152-240 */
int gtnh_savetxt(const char* file, gtnx_graph_t g);                                             /* utils.cpp:242-279 */
int gtnh_write_dot(gtnx_graph_t g, const char* file, const int* ikeys, const char* const* inames, int n_isymbols,
                   const int* okeys, const char* const* onames, int n_osymbols);               /* utils.cpp:347-420 */
/* operator<< (utils.cpp:263-270); call with out = NULL to learn the size in *need (incl. the terminator) */
int gtnh_repr(gtnx_graph_t g, char* out, size_t cap, size_t* need);

#ifdef __cplusplus
}
#endif
#endif
