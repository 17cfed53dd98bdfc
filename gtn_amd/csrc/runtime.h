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

#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace gtnx {

class Runtime;

struct DevMem {
  void* ptr = nullptr;
  size_t bytes = 0;
  bool borrowed = false;  // the caller's memory (a tensor that outlives the graphs over it): never pooled
  Runtime* owner = nullptr;  // the device context whose pool the block goes back to (null: borrowed)
  std::shared_ptr<void> keep;  // borrowed from this (a pinned host block a kernel reads in place: ops.h upload_vec)
  ~DevMem();
  template <class T>
  T* as(size_t byte_off = 0) const { return reinterpret_cast<T*>(static_cast<char*>(ptr) + byte_off); }
};
using DevMemP = std::shared_ptr<DevMem>;

struct PinnedMem {
  void* ptr = nullptr;
  size_t bytes = 0;
  Runtime* owner = nullptr;
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

// One Runtime per DEVICE (stream, memory pools, profiler), made on first use.  Which one a call gets is a property
// of the calling THREAD: gtnx_set_device(d) (Runtime::set_current_device) selects device d for the calling thread
// only, and a thread that never chose inherits the process default -- the device of the first set_device call, else
// 0 -- so a one-GPU process (one rank per GPU under torch.distributed) behaves as before, and a C++ host that
// drives the 8 GPUs of a node gives each device its own thread (include/gtn/parallel.h: parallelMapSharded; the
// pool threads of a parallelMap inherit the device of the thread that called it).  Graphs remember the device they
// were made on; blocks go back to the pool of the device they came from whichever thread lets go of them.
class Runtime {
 public:
  static Runtime& get();          // the calling thread's device context; throws GTNX_DEVICE_ERROR when no GPU is usable
  static Runtime& of(int device); // a given device's context (made on first use)
  static int device_count();      // never throws
  static bool initialized();      // some device context exists
  static void set_current_device(int d);  // for the calling thread (and, the first time, the process default)
  static int current_device();            // of the calling thread
  int device() const { return device_; }
  void activate();                // hipSetDevice(device()) for the calling thread if it is on another one

  hipStream_t stream() const { return stream_; }
  void set_stream(hipStream_t s);
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
  // for the GPU anyway (sync / blocking copy), off the caller's critical path -- and ON THE THREAD THAT LET GO
  // OF THEM (usually the one that built them): every thread has its own list (an Inbox).  A block freed by
  // another thread goes back to the allocating thread's malloc arena under that arena's lock, and that thread is
  // by then allocating the next step's objects out of it: measured on the 256-thread host of an MI355X, a
  // step's host work ran two to five times slower while other threads took the previous step apart.
  // What a thread builds FOR others (the slices of a parallelMap region, region.cpp) is sent home the same way
  // when its last reference dies elsewhere: send(home_of_builder, ...).
  struct Inbox {
    std::mutex mu;
    std::vector<std::pair<void*, void (*)(void*)>> items;
    size_t load = 0;    // what the waiting items stand for, in graphs (an item may be a vector of them): reset when the list runs empty
    bool dead = false;  // the thread is gone: whoever has garbage for it takes it apart on the spot
  };
  using InboxP = std::shared_ptr<Inbox>;
  static InboxP home();                                             // the calling thread's list
  static void send(const InboxP& to, void* p, void (*del)(void*));  // (null / dead / over-full list: destroyed here and now)
  static void drain_all_inboxes();                                  // every live thread's list, by the caller (empty_cache, OOM retry)
  static void defer_delete(void* p, void (*del)(void*), size_t weight = 1);  // weight: graphs behind the pointer            // = send(home(), ...)
  static void drain_deferred();                                     // the calling thread's list, all of it
  static bool drain_some(size_t max_items);
  static size_t deferred_count();                                   // entries waiting on the calling thread's list
  void drain_until(void* hip_event);  // reclaim until the event has happened, then wait for it
  void drain_while_busy();
  void sync_while_draining();  // drain_while_busy(), then wait for the stream (what d2h_sync ends with)

  // ---- a SIDE stream with a host thread of its own to feed it.  For a launch chain that does not depend on what
  // the engine stream runs meanwhile (the beta sweep of a dense product next to its alpha sweep, ops_lazy.cpp:
  // a chain of T dependent launches is bound by its latencies -- two of them share the chip -- and the host
  // enqueues a launch of that argument size in ~7 us, so the second chain needs a second enqueuing thread).
  // side_launch: `fn(side)` is called on the side thread; its launches are ordered AFTER everything queued on the
  // engine stream up to this call.  side_join: everything queued on the engine stream from now on is ordered after
  // the job's launches (the host only waits until they have been enqueued); rethrows what fn threw.  Buffers the
  // job touches must stay alive until it was joined.
  struct SideJob;
  using SideJobP = std::shared_ptr<SideJob>;
  SideJobP side_launch(std::function<void(hipStream_t)> fn);
  void side_join(const SideJobP& job);

  // copies (async on the engine stream; h2d source must be pinned or outlive sync())
  void h2d(void* dst, const void* src, size_t bytes);
  void d2h_sync(void* dst, const void* src, size_t bytes);  // returns after the data landed
  void d2d(void* dst, const void* src, size_t bytes);
  // pinned host memory (alloc_pinned) to / from the device: up to GTNX_H2D_KERNEL_BYTES (default 1 MB) by a kernel of
  // ours, which is the next dispatch of the compute queue where a copy of the runtime's is ordered against it from outside
  void h2d_pinned(void* dst, const void* pinned_src, size_t bytes);
  void d2h_pinned_async(void* pinned_dst, const void* src, size_t bytes);  // ordered on the engine stream

  // ---- a ring of pinned floats for kernels that produce ONE scalar (the loss of a criterion written with the
  // per-graph functions): the kernel writes its value there as well, and item() of that result waits for the stream
  // instead of putting a copy on it.  A slot is handed out again after kMirrorSlots others: mirror_read says so.
  struct MirrorSlot {
    float* ptr = nullptr;
    uint64_t gen = 0;
  };
  static constexpr size_t kMirrorSlots = 1024;
  MirrorSlot mirror_slot();
  bool mirror_read(const MirrorSlot& m, float* out);  // waits for the stream; false: the slot has a new owner

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
  explicit Runtime(int device);
  void collect_prof();
  float* mirror_ring_ = nullptr;
  std::atomic<uint64_t>* mirror_gen_ = nullptr;
  std::atomic<uint64_t> mirror_next_{0};
  hipStream_t own_stream_ = nullptr;
  hipStream_t stream_ = nullptr;
  int device_ = 0;
  int cu_count_ = 256;
  std::mutex mu_;
  std::multimap<size_t, void*> free_dev_;
  std::multimap<size_t, void*> free_pinned_;
  // Released pinned blocks wait for the stream in GROUPS: a block let go of joins `unstamped_pinned_` without a HIP
  // call; one event, recorded after kPinnedStampEvery releases (or when the pool has nothing to offer), stands for
  // all of them -- the stream is in order, so an event recorded after a block's release covers every operation that
  // used it.  (One hipEventRecord per block was a marker in the stream and 2-4 us of the host per staged table.)
  struct PinnedGroup {
    hipEvent_t ev;
    std::vector<std::pair<void*, size_t>> blocks;
  };
  static constexpr size_t kPinnedStampEvery = 8;
  std::vector<PinnedGroup> pending_pinned_;
  std::vector<std::pair<void*, size_t>> unstamped_pinned_;
  void stamp_pinned_locked();  // mu_ held
  uint64_t reserved_ = 0, in_use_ = 0;
  bool prof_on_ = false;
  struct ProfRec {
    std::string name;
    double bytes;
    hipEvent_t a, b;
  };
  std::vector<ProfRec> prof_recs_;
  std::vector<hipEvent_t> ev_pool_;
  struct Side;
  std::atomic<Side*> side_{nullptr};  // made on first use, never destroyed (its thread outlives static destruction)
  std::map<std::string, ProfEntry> prof_;
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
double host_now_ms();
void host_timer_add(const char* name, double ms);  // one sample for the GTNX_HOST_TIMING table
#define GTNX_HT_CAT2(a, b) a##b
#define GTNX_HT_CAT(a, b) GTNX_HT_CAT2(a, b)
#define GTNX_HOST_T(name) ::gtnx::HostTimer GTNX_HT_CAT(_host_t_, __LINE__)(name)

} // namespace gtnx
