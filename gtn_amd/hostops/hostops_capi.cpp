// hostops_capi.cpp -- the host-side graph builders and formats of the reference's Python binding
// (bindings/python/gtn/_functions.cpp: clone / project_* / concat / closure / union / remove;
// _rand.cpp: sample / rand_equivalent; _utils.cpp: load / save / loadtxt / savetxt / write_dot;
// _graph.cpp: __repr__) for the Python mirror gtn_amd/.
//
// They already exist on this engine as header-only C++ over the C ABI (include/gtn/functions.h,
// rand.h, utils.h -- structure builders with the reference's gradient slices, functions.cpp:66-318).
// This file only gives them C entry points on gtnx_graph_t handles, so Python shares ONE
// implementation with C++ callers.  The gtnx_* symbols are left undefined on purpose: the Python
// package loads libgtn_amd.so (or, in the CPU tests, the reference shim) globally first and this
// library binds to whichever is there.
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "gtn/gtn.h"
#include "gtn_amd_hostops.h"

using namespace gtn;

namespace {
thread_local std::string g_err;

Graph borrow(gtnx_graph_t h) {  // an aliasing copy, as the binding passes `const Graph&`
  gtnx_graph_t c = nullptr;
  detail::check(gtnx_graph_copy(h, &c));
  return Graph::fromHandle(c);
}
gtnx_graph_t give(const Graph& g) {  // a new owning handle for the caller
  gtnx_graph_t c = nullptr;
  detail::check(gtnx_graph_copy(g.handle(), &c));
  return c;
}
std::vector<Graph> borrowAll(const gtnx_graph_t* h, int n) {
  std::vector<Graph> v;
  v.reserve(size_t(n));
  for (int i = 0; i < n; ++i) v.push_back(borrow(h[i]));
  return v;
}
int put(std::string& dst, char* out, size_t cap, size_t* need) {
  if (need) *need = dst.size() + 1;
  if (out && cap) {
    const size_t k = dst.size() < cap - 1 ? dst.size() : cap - 1;
    std::memcpy(out, dst.data(), k);
    out[k] = 0;
  }
  return 0;
}
SymbolMap symbols(const int* keys, const char* const* names, int n) {
  SymbolMap m;
  for (int i = 0; i < n; ++i) m[keys[i]] = names[i];
  return m;
}
// status codes of gtn_amd.h for the exception types the header mirror throws
template <class F>
int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument& e) {
    g_err = e.what();
    return GTNX_INVALID_ARGUMENT;
  } catch (const std::out_of_range& e) {
    g_err = e.what();
    return GTNX_OUT_OF_RANGE;
  } catch (const std::logic_error& e) {
    g_err = e.what();
    return GTNX_LOGIC_ERROR;
  } catch (const std::exception& e) {
    g_err = e.what();
    return GTNX_RUNTIME_ERROR;
  }
}
} // namespace

#define API extern "C" __attribute__((visibility("default")))  // signatures: include/gtn_amd_hostops.h

API const char* gtnh_last_error() { return g_err.c_str(); }

API int gtnh_clone(gtnx_graph_t g, int projection, gtnx_graph_t* out) {  // 0 none, 1 input, 2 output
  return guard([&] { *out = give(clone(borrow(g), static_cast<Projection>(projection))); });
}
API int gtnh_concat(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] { *out = give(concat(borrowAll(g, n))); });
}
API int gtnh_closure(gtnx_graph_t g, gtnx_graph_t* out) {
  return guard([&] { *out = give(closure(borrow(g))); });
}
API int gtnh_union(const gtnx_graph_t* g, int n, gtnx_graph_t* out) {
  return guard([&] { *out = give(union_(borrowAll(g, n))); });
}
API int gtnh_remove(gtnx_graph_t g, int ilabel, int olabel, gtnx_graph_t* out) {
  return guard([&] { *out = give(remove(borrow(g), ilabel, olabel)); });
}
API int gtnh_sample(gtnx_graph_t g, size_t max_length, gtnx_graph_t* out) {
  return guard([&] { *out = give(sample(borrow(g), max_length)); });
}
API int gtnh_rand_equivalent(gtnx_graph_t a, gtnx_graph_t b, size_t num_samples, double tol, size_t max_length,
                             int* out) {
  return guard([&] { *out = randEquivalent(borrow(a), borrow(b), num_samples, tol, max_length) ? 1 : 0; });
}
API int gtnh_load(const char* file, gtnx_graph_t* out) {
  return guard([&] { *out = give(load(std::string(file))); });
}
API int gtnh_save(const char* file, gtnx_graph_t g) {
  return guard([&] { save(std::string(file), borrow(g)); });
}
API int gtnh_loadtxt(const char* file, gtnx_graph_t* out) {
  return guard([&] { *out = give(loadTxt(std::string(file))); });
}
API int gtnh_savetxt(const char* file, gtnx_graph_t g) {
  return guard([&] { saveTxt(std::string(file), borrow(g)); });
}
API int gtnh_write_dot(gtnx_graph_t g, const char* file, const int* ikeys, const char* const* inames, int ni,
                       const int* okeys, const char* const* onames, int no) {
  return guard([&] { draw(borrow(g), std::string(file), symbols(ikeys, inames, ni), symbols(okeys, onames, no)); });
}
// the summarising operator<< of utils.h:106 (what the binding's __repr__ prints); call with out = NULL for the size
API int gtnh_repr(gtnx_graph_t g, char* out, size_t cap, size_t* need) {
  return guard([&] {
    std::ostringstream ss;
    ss << borrow(g);
    std::string s = ss.str();
    put(s, out, cap, need);
  });
}
