// runtime.h -- device, stream, memory pools and kernel-family profiler.
//
// MI355X-first choices:
//  * one process drives one GPU (one rank per GPU under torch.distributed); the
//    engine owns ONE in-order HIP stream (or borrows the caller's, e.g. torch's
//    current stream) so buffer reuse needs no cross-stream events;
//  * device memory comes from a caching pool sized for 288 GB of HBM: batched
//    ops carve ONE arena per launch family instead of per-graph hipMallocs;
//  * host<->device staging goes through pinned blocks recycled by event.
#pragma once

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace gtnx {

struct DevMem {
  void* ptr = nullptr;
  size_t bytes = 0;
  bool borrowed = false;  // the caller's memory (a tensor that outlives the graphs over it): never pooled
  ~DevMem();
  template <class T>
  T* as(size_t byte_off = 0) const { return reinterpret_cast<T*>(static_cast<char*>(ptr) + byte_off); }
};
using DevMemP = std::shared_ptr<DevMem>;

struct PinnedMem {
  void* ptr = nullptr;
  size_t bytes = 0;
  ~PinnedMem();
  template <class T>
  T* as(size_t byte_off = 0) const { return reinterpret_cast<T*>(static_cast<char*>(ptr) + byte_off); }
};
using PinnedMemP = std::shared_ptr<PinnedMem>;

struct ProfEntry {
  double total_ms = 0;
  int64_t launches = 0;
  double bytes = 0;
};

class Runtime {
 public:
  static Runtime& get();          // throws GTNX_DEVICE_ERROR when no GPU is usable
  static int device_count();      // never throws
  static bool initialized();

  hipStream_t stream() const { return stream_; }
  void set_stream(hipStream_t s);
  void set_device(int d);
  void sync();

  DevMemP alloc(size_t bytes);            // uninitialised
  DevMemP alloc_zero(size_t bytes);       // + hipMemsetAsync(0)
  PinnedMemP alloc_pinned(size_t bytes);  // recycled once the stream passed its release point
  void release_dev(void* p, size_t bytes);
  void release_pinned(void* p, size_t bytes);
  void empty_cache();
  void stats(uint64_t* reserved, uint64_t* in_use);

  // Deferred reclamation: objects the caller let go of (graph handles, the tape of
  // a finished backward) are destroyed at the next point where the host would wait
  // for the GPU anyway (sync / blocking copy), off the caller's critical path.
  void defer_delete(void* p, void (*del)(void*));
  void drain_deferred();
  bool drain_some(size_t max_items);
  void drain_while_busy();

  // copies (async on the engine stream; h2d source must be pinned or outlive sync())
  void h2d(void* dst, const void* src, size_t bytes);
  void d2h_sync(void* dst, const void* src, size_t bytes);  // returns after the data landed
  void d2d(void* dst, const void* src, size_t bytes);

  // ---- profiler: hipEvent pairs around kernel families on the launch stream
  void prof_enable(bool on);
  void prof_add_bytes(const std::string& name, double bytes);  // late algorithmic-byte credit
  void prof_reset();
  bool prof_on() const { return prof_on_; }
  struct Scope {
    Runtime* rt;
    int idx;
    Scope(Runtime* r, const char* name, double bytes);
    ~Scope();
  };
  ProfEntry prof_get(const std::string& name);
  std::string prof_names();
  int cu_count() const { return cu_count_; }

 private:
  Runtime();
  void collect_prof();
  hipStream_t own_stream_ = nullptr;
  hipStream_t stream_ = nullptr;
  int device_ = 0;
  int cu_count_ = 256;
  std::mutex mu_;
  std::multimap<size_t, void*> free_dev_;
  std::multimap<size_t, void*> free_pinned_;
  struct PendingPinned {
    void* ptr;
    size_t bytes;
    hipEvent_t ev;
  };
  std::vector<PendingPinned> pending_pinned_;
  uint64_t reserved_ = 0, in_use_ = 0;
  size_t pool_budget_ = size_t(64) << 30;  // reclaim before growing the pool past this (alloc)
  bool prof_on_ = false;
  struct ProfRec {
    std::string name;
    double bytes;
    hipEvent_t a, b;
  };
  std::vector<ProfRec> prof_recs_;
  std::vector<hipEvent_t> ev_pool_;
  std::map<std::string, ProfEntry> prof_;
  std::mutex defer_mu_;
  std::vector<std::pair<void*, void (*)(void*)>> deferred_;
  friend struct Scope;
};

#define GTNX_PROF(name, bytes) ::gtnx::Runtime::Scope _prof_scope(&::gtnx::Runtime::get(), name, bytes)

// host wall-clock phases (GTNX_HOST_TIMING=1 prints the table at exit); diagnostics only
struct HostTimer {
  const char* name;
  double t0;
  explicit HostTimer(const char* n);
  ~HostTimer();
  static bool enabled();
};
void host_timer_add(const char* name, double ms);  // one sample for the GTNX_HOST_TIMING table
#define GTNX_HT_CAT2(a, b) a##b
#define GTNX_HT_CAT(a, b) GTNX_HT_CAT2(a, b)
#define GTNX_HOST_T(name) ::gtnx::HostTimer GTNX_HT_CAT(_host_t_, __LINE__)(name)

} // namespace gtnx
