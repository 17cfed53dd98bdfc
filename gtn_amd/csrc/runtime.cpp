// runtime.cpp -- see runtime.h
#include "runtime.h"

#include "kernels.h"

#include <malloc.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

namespace gtnx {

namespace {
constexpr int kMaxDevices = 64;
std::atomic<Runtime*> g_rts[kMaxDevices];
std::atomic<bool> g_any_rt{false};
std::mutex g_rt_mu;
std::atomic<int> g_default_device{-1};   // the first gtnx_set_device of the process (else: what HIP says, below)
thread_local int t_device = -1;          // the calling thread's choice (-1: the process default)

// What HIP has as the calling thread's device.  Asked, never remembered: the host framework shares the thread
// (torch.cuda.device(k) is a hipSetDevice behind the engine's back), so a cached answer goes stale.
// hipGetDevice / hipSetDevice read / write one thread-local of the HIP runtime (tens of nanoseconds).
int hip_device_now() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return d;
}

// a HIP call that must be made with a given device current, from a thread that may be on another one (a block
// that goes home from wherever its last reference died)
struct OnDevice {
  int prev, mine;
  explicit OnDevice(int d) : prev(hip_device_now()), mine(d) {
    if (prev != d) (void)hipSetDevice(d);
  }
  ~OnDevice() {
    if (prev >= 0 && prev != mine) (void)hipSetDevice(prev);
  }
};

// every thread's list of deferred garbage (Runtime::Inbox), so that empty_cache() and the out-of-memory retry
// of alloc() can reach device blocks pinned by handles waiting on a thread that never comes to a reclamation
// point
std::mutex g_inbox_mu;
std::vector<std::weak_ptr<Runtime::Inbox>> g_inboxes;

// a foreign list no longer takes more than this: beyond it the sender destroys the object itself (a long-lived
// thread that builds graphs and never synchronises would otherwise pin everything destroyed elsewhere)
constexpr size_t kInboxBound = size_t(1) << 14;

size_t round_size(size_t b) {
  if (b < 512) return 512;
  if (b < (1u << 20)) return align_up(b, 512);
  return align_up(b, size_t(2) << 20);  // 2 MiB granules for arenas
}
} // namespace

DevMem::~DevMem() {
  if (ptr && owner && !borrowed) owner->release_dev(ptr, bytes);
}
PinnedMem::~PinnedMem() {
  if (ptr && owner) owner->release_pinned(ptr, bytes);
}

int Runtime::device_count() {
  // asked once (300 us per call on the GPU box; every pool thread asks when it is put on its device)
  static const int count = [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    return n;
  }();
  return count;
}

bool Runtime::initialized() { return g_any_rt.load(std::memory_order_acquire); }

int Runtime::current_device() {
  if (t_device >= 0) return t_device;
  const int d = g_default_device.load(std::memory_order_acquire);
  if (d >= 0) return d;
  // no gtnx_set_device anywhere yet: the device the calling thread already has with HIP (a rank that only did
  // torch.cuda.set_device(k) gets its graphs on k, and its thread is left on k), as the process default from here on
  int now = hip_device_now();
  if (now < 0 || now >= kMaxDevices) now = 0;
  int none = -1;
  g_default_device.compare_exchange_strong(none, now);
  return g_default_device.load(std::memory_order_acquire);
}

void Runtime::set_current_device(int d) {
  if (d < 0 || d >= kMaxDevices || d >= device_count()) throw_invalid("[gtnx_set_device] no such device");
  t_device = d;
  int none = -1;
  g_default_device.compare_exchange_strong(none, d);
  Runtime::of(d).activate();
}

void Runtime::activate() { HIP_CHECK(hipSetDevice(device_)); }

Runtime& Runtime::of(int d) {
  if (d < 0 || d >= kMaxDevices) throw_invalid("[gtn_amd] device index out of range");
  Runtime* r = g_rts[d].load(std::memory_order_acquire);
  if (!r) {
    std::lock_guard<std::mutex> lk(g_rt_mu);
    r = g_rts[d].load(std::memory_order_acquire);
    if (!r) {
      if (device_count() <= d)
        throw_device(
            "gtn_amd: no HIP device visible -- this engine runs its graph functions on an "
            "MI355X (gfx950) only and has no CPU fallback");
      r = new Runtime(d);
      g_rts[d].store(r, std::memory_order_release);
      g_any_rt.store(true, std::memory_order_release);
    }
  }
  return *r;
}

Runtime& Runtime::get() {
  Runtime& r = of(current_device());
  r.activate();
  return r;
}

Runtime::Runtime(int device) : device_(device) {
  // The host side of a step allocates and frees a few hundred KB of scratch (launch tables, per-utterance
  // records); with glibc's defaults the heap top is trimmed after every step and grown again in the next
  // (brk + page faults on the thread that joins the region: 17 % of its time in the stack samples of
  // tools/nullhip/region_step).  GTNX_MALLOPT=1 keeps freed heap in the process instead: opt-in, because it changes
  // the allocator of the WHOLE host process (a library should not) and because it stopped mattering once
  // every thread frees what it allocated and placeholder chunks are recycled (measured: 1.253 vs 1.265 ms per C3 batch).
  if (std::getenv("GTNX_MALLOPT")) {
    mallopt(M_TRIM_THRESHOLD, 256 << 20);
    mallopt(M_TOP_PAD, 16 << 20);
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
  }
  HIP_CHECK(hipSetDevice(device_));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device_));
  cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_CHECK(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  stream_ = own_stream_;
}

void Runtime::set_stream(hipStream_t s) {
  hipStream_t next = s ? s : own_stream_;
  if (next == stream_) return;  // (a loss called every step names the same stream every time)
  // what was queued on the old stream is ordered before anything the new one gets: an event, not a
  // host wait -- pooled buffers are handed out in enqueue order, which now spans both streams
  drain_deferred();
  hipEvent_t ev;
  HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(ev, stream_));
  HIP_CHECK(hipStreamWaitEvent(next, ev, 0));
  HIP_CHECK(hipEventDestroy(ev));
  stream_ = next;
}

// ---- side stream (runtime.h)
struct Runtime::SideJob {
  std::function<void(hipStream_t)> fn;
  hipEvent_t fork = nullptr, done_ev = nullptr;
  std::exception_ptr err;
  bool enqueued = false, joined = false;  // under Side::mu
  ~SideJob() {
    if (fork) (void)hipEventDestroy(fork);
    if (done_ev) (void)hipEventDestroy(done_ev);
  }
};
struct Runtime::Side {
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<SideJobP> q;
  std::thread th;
};

Runtime::SideJobP Runtime::side_launch(std::function<void(hipStream_t)> fn) {
  if (!side_.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!side_.load(std::memory_order_relaxed)) {
      Side* sd = new Side();
      HIP_CHECK(hipStreamCreateWithFlags(&sd->stream, hipStreamNonBlocking));
      const int dev = device_;
      sd->th = std::thread([sd, dev] {
        (void)hipSetDevice(dev);
        t_device = dev;
        for (;;) {
          SideJobP job;
          {
            std::unique_lock<std::mutex> lk(sd->mu);
            sd->cv_work.wait(lk, [&] { return !sd->q.empty(); });
            job = std::move(sd->q.front());
            sd->q.pop_front();
          }
          try {
            if (hipStreamWaitEvent(sd->stream, job->fork, 0) != hipSuccess) throw_runtime("gtn_amd: side stream: wait failed");
            job->fn(sd->stream);
          } catch (...) {
            job->err = std::current_exception();
          }
          (void)hipEventRecord(job->done_ev, sd->stream);  // also after a failure: whatever was queued is waited for
          job->fn = nullptr;
          {
            std::lock_guard<std::mutex> lk(sd->mu);
            job->enqueued = true;
          }
          sd->cv_done.notify_all();
        }
      });
      sd->th.detach();
      side_.store(sd, std::memory_order_release);
    }
  }
  Side* const side = side_.load(std::memory_order_acquire);
  auto job = std::make_shared<SideJob>();
  job->fn = std::move(fn);
  HIP_CHECK(hipEventCreateWithFlags(&job->fork, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&job->done_ev, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(job->fork, stream_));
  {
    std::lock_guard<std::mutex> lk(side->mu);
    side->q.push_back(job);
  }
  side->cv_work.notify_one();
  return job;
}

void Runtime::side_join(const SideJobP& job) {
  Side* const side = side_.load(std::memory_order_acquire);
  if (!job || !side) return;
  {
    std::unique_lock<std::mutex> lk(side->mu);
    side->cv_done.wait(lk, [&] { return job->enqueued; });
    if (job->joined) return;
    job->joined = true;
  }
  OnDevice on(device_);
  HIP_CHECK(hipStreamWaitEvent(stream_, job->done_ev, 0));
  if (job->err) std::rethrow_exception(job->err);
}

void Runtime::sync() {
  drain_while_busy();  // the GPU is (usually) still busy: reclaim while we would wait
  HIP_CHECK(hipStreamSynchronize(stream_));
}

namespace {
// The calling thread's list, reachable through TRIVIALLY DESTRUCTIBLE thread-locals: send() / defer_delete() also run
// from other thread_local destructors at thread exit (region.cpp's slice deleter, a slab cache), possibly after the
// holder below is gone -- its members must not be read then.  t_box: the live list; t_box_gone: the holder has been
// destroyed (whatever arrives now is destroyed by the sender); t_draining: drain_deferred() is on this thread's stack.
thread_local Runtime::Inbox* t_box = nullptr;
thread_local bool t_box_gone = false;
thread_local bool t_draining = false;
struct InboxHolder {
  Runtime::InboxP box = std::make_shared<Runtime::Inbox>();
  InboxHolder() {
    t_box = box.get();
    std::lock_guard<std::mutex> lk(g_inbox_mu);
    size_t live = 0;  // (drop the entries of threads that are gone while we are here)
    for (auto& w : g_inboxes)
      if (!w.expired()) g_inboxes[live++] = std::move(w);
    g_inboxes.resize(live);
    g_inboxes.push_back(box);
  }
  ~InboxHolder() {
    {
      std::lock_guard<std::mutex> lk(box->mu);
      box->dead = true;
    }
    for (;;) {  // (destructors may send more: to a dead list, i.e. destroyed by the sender)
      std::vector<std::pair<void*, void (*)(void*)>> batch;
      {
        std::lock_guard<std::mutex> lk(box->mu);
        if (box->items.empty()) break;
        batch.swap(box->items);
      }
      for (auto& e : batch) e.second(e.first);
    }
    t_box = nullptr;
    t_box_gone = true;
  }
};
thread_local InboxHolder t_home;
// the calling thread's list (made on first use), or null once the thread's holder has been destroyed
Runtime::Inbox* my_box() {
  if (t_box) return t_box;
  if (t_box_gone) return nullptr;
  return t_home.box.get();
}
}  // namespace

Runtime::InboxP Runtime::home() { return my_box() ? t_home.box : InboxP(); }

void Runtime::send(const InboxP& to, void* p, void (*del)(void*)) {
  if (to && to.get() == my_box()) {  // the caller's own list: counted, and taken apart when it is full
    defer_delete(p, del, 1);
    return;
  }
  if (to) {
    std::lock_guard<std::mutex> lk(to->mu);
    if (!to->dead && to->items.size() < kInboxBound) {
      to->items.push_back({p, del});
      to->load += 1;
      return;
    }
  }
  del(p);  // dead, or its owner is not keeping up: destroyed by the sender
}

// every live thread's list, taken apart by the caller (memory pressure: where an object dies matters less than
// that the device blocks it holds come back)
void Runtime::drain_all_inboxes() {
  std::vector<InboxP> boxes;
  {
    std::lock_guard<std::mutex> lk(g_inbox_mu);
    for (auto& w : g_inboxes)
      if (InboxP b = w.lock()) boxes.push_back(std::move(b));
  }
  for (int round = 0; round < 64; ++round) {  // (destructors may send more)
    bool any = false;
    for (auto& b : boxes) {
      std::vector<std::pair<void*, void (*)(void*)>> batch;
      {
        std::lock_guard<std::mutex> lk(b->mu);
        batch.swap(b->items);
        b->load = 0;
      }
      any |= !batch.empty();
      for (auto& e : batch) e.second(e.first);
    }
    if (!any) break;
  }
}

// a thread's own list is taken apart once it holds this many objects even if the thread never comes to a blocking
// point (a scoring loop that leaves its results on the device): beyond a few thousand the garbage of SEVERAL
// iterations is waiting, every new iteration allocates cold memory instead of what the last one let go of, and the
// host pays in page faults and cache misses (tools/nullhip/small_step c2: 155 page faults = 0.3 ms per batch of 256)
static const size_t kDeferFull = [] {
  const char* e = std::getenv("GTNX_DEFER_FULL");
  const long v = e ? std::atol(e) : 0;
  return size_t(v > 0 ? v : 8192);
}();

void Runtime::defer_delete(void* p, void (*del)(void*), size_t weight) {
  Inbox* own = my_box();
  if (!own) {  // (thread teardown, after the holder: destroyed now)
    del(p);
    return;
  }
  Inbox& b = *own;
  bool full = false, dead = false;
  {
    std::lock_guard<std::mutex> lk(b.mu);
    if (b.dead) {
      dead = true;  // (the thread is on its way out: destroyed now)
    } else {
      b.items.push_back({p, del});
      b.load += weight ? weight : 1;
      full = b.load >= kDeferFull;
    }
  }
  if (dead) del(p);
  // (a destructor run BY a drain may defer more: that lands on the list the drain is emptying -- no nested drain)
  if (full && !t_draining) drain_deferred();
}

void Runtime::drain_deferred() {
  GTNX_HOST_T("runtime.drain_deferred");
  if (t_draining) return;
  t_draining = true;
  try {
    while (drain_some(256)) {
    }
  } catch (...) {
    t_draining = false;
    throw;
  }
  t_draining = false;
}

// destroys up to `max_items` of what is waiting on the calling thread's list; false when nothing was
bool Runtime::drain_some(size_t max_items) {
  Inbox* own = my_box();
  if (!own) return false;
  Inbox& b = *own;
  std::vector<std::pair<void*, void (*)(void*)>> batch;
  {
    std::lock_guard<std::mutex> lk(b.mu);
    if (b.items.empty()) return false;
    const size_t n = std::min(max_items, b.items.size());
    batch.assign(b.items.end() - long(n), b.items.end());
    // (weights are not kept per item: the load shrinks in proportion, and is exact again when the list is empty)
    b.load = b.items.size() == n ? 0 : b.load - std::min(b.load, b.load * n / b.items.size());
    b.items.resize(b.items.size() - n);
  }
  for (auto& e : batch) e.second(e.first);  // (destructors may defer more)
  return true;
}

size_t Runtime::deferred_count() {
  Inbox* own = my_box();
  if (!own) return 0;
  Inbox& b = *own;
  std::lock_guard<std::mutex> lk(b.mu);
  return b.items.size();
}

void Runtime::drain_until(void* hip_event) {
  hipEvent_t ev = static_cast<hipEvent_t>(hip_event);
  while (hipEventQuery(ev) == hipErrorNotReady) {
    if (!drain_some(32)) break;
  }
  (void)hipGetLastError();
  HIP_CHECK(hipEventSynchronize(ev));
}

// reclaim while the GPU is busy and the host would only wait for it; what is left waits for the next such
// moment
void Runtime::drain_while_busy() {
  GTNX_HOST_T("runtime.drain_while_busy");
  Inbox* own = my_box();
  if (!own) return;
  Inbox& b = *own;
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(b.mu);
      if (b.items.empty()) return;
      if (b.items.size() >= (1u << 14)) break;  // too much waiting: all of it, now
    }
    if (hipStreamQuery(stream_) != hipErrorNotReady) {
      (void)hipGetLastError();
      return;
    }
    drain_some(32);
  }
  drain_deferred();
}

DevMemP Runtime::alloc(size_t bytes) {
  size_t sz = round_size(bytes ? bytes : 1);
  void* p = nullptr;
  auto from_pool = [&] {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = free_dev_.lower_bound(sz);
    // accept a cached block up to 1.5x the request (big arenas) / exact class (small)
    if (it != free_dev_.end() && it->first <= sz + sz / 2) {
      p = it->second;
      sz = it->first;
      free_dev_.erase(it);
    }
  };
  from_pool();
  if (!p && sz >= (size_t(32) << 20)) {
    // A big block that the pool does not have.  What this thread released since its last reclamation point may
    // hold it (a step's alpha planes are 0.4 GB at C3, the built lattices of a step 13 GB; a loop that never waits
    // for the device released a new set every step and the pool grew by that much per step -- 44 GB after 100
    // steps -- with a hipMalloc each): take the calling thread's own list apart and look again.  (Its OWN list:
    // what other threads built goes home to them, runtime.h -- this used to be every thread's garbage, 1.5 ms at
    // C3 on the one thread a step waits for.)
    drain_deferred();
    from_pool();
  }
  if (!p) {
    GTNX_HOST_T("runtime.alloc.hipMalloc (pool miss)");
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      empty_cache();  // (every thread's deferred garbage included)
      HIP_CHECK(hipMalloc(&p, sz));
    }
    std::lock_guard<std::mutex> lk(mu_);
    reserved_ += sz;
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    in_use_ += sz;
  }
  auto m = std::make_shared<DevMem>();
  m->ptr = p;
  m->bytes = sz;
  m->owner = this;
  return m;
}

DevMemP Runtime::alloc_zero(size_t bytes) {
  DevMemP m = alloc(bytes);
  // (blocks are multiples of 256 bytes: whole words.  Small ones by a kernel of ours, like d2d)
  if (bytes <= (size_t(16) << 20) && m->bytes % 4 == 0) launch_fill_i32(m->as<int>(), 0, (bytes + 3) / 4, stream_);
  else HIP_CHECK(hipMemsetAsync(m->ptr, 0, bytes ? bytes : 1, stream_));
  return m;
}

void Runtime::release_dev(void* p, size_t bytes) {
  // single in-order stream: a block freed on the host after its last kernel was
  // ENQUEUED can be handed to a later launch on the same stream safely.
  std::lock_guard<std::mutex> lk(mu_);
  free_dev_.emplace(bytes, p);
  in_use_ -= bytes;
}

PinnedMemP Runtime::alloc_pinned(size_t bytes) {
  size_t sz = round_size(bytes ? bytes : 1);
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu_);
    // recycle the groups whose event has completed
    auto recycle = [&] {
      for (size_t i = 0; i < pending_pinned_.size();) {
        if (hipEventQuery(pending_pinned_[i].ev) == hipSuccess) {
          for (auto& b : pending_pinned_[i].blocks) free_pinned_.emplace(b.second, b.first);
          ev_pool_.push_back(pending_pinned_[i].ev);
          if (i + 1 != pending_pinned_.size()) pending_pinned_[i] = std::move(pending_pinned_.back());
          pending_pinned_.pop_back();
        } else {
          (void)hipGetLastError();
          ++i;
        }
      }
    };
    recycle();
    {
      auto hit = free_pinned_.lower_bound(sz);
      // nothing to offer: whatever waits without an event gets one now, so that it comes back soon
      if ((hit == free_pinned_.end() || hit->first > sz * 2) && !unstamped_pinned_.empty()) stamp_pinned_locked();
    }
    auto it = free_pinned_.lower_bound(sz);
    if (it != free_pinned_.end() && it->first <= sz * 2) {
      p = it->second;
      sz = it->first;
      free_pinned_.erase(it);
    }
  }
  if (!p) {
    GTNX_HOST_T("runtime.alloc_pinned.hipHostMalloc (pool miss)");
    HIP_CHECK(hipHostMalloc(&p, sz, hipHostMallocDefault));
  }
  auto m = std::make_shared<PinnedMem>();
  m->ptr = p;
  m->bytes = sz;
  m->owner = this;
  return m;
}

void Runtime::release_pinned(void* p, size_t bytes) {
  OnDevice here(device_);
  std::lock_guard<std::mutex> lk(mu_);
  static const size_t every = [] {  // GTNX_PINNED_STAMP_EVERY: releases per event (1: an event per block, as before)
    const char* e = std::getenv("GTNX_PINNED_STAMP_EVERY");
    return e && std::atol(e) > 0 ? size_t(std::atol(e)) : kPinnedStampEvery;
  }();
  unstamped_pinned_.emplace_back(p, bytes);
  if (unstamped_pinned_.size() >= every) stamp_pinned_locked();
}
void Runtime::stamp_pinned_locked() {
  if (unstamped_pinned_.empty()) return;
  hipEvent_t ev;
  if (!ev_pool_.empty()) {
    ev = ev_pool_.back();
    ev_pool_.pop_back();
  } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream_);
    for (auto& b : unstamped_pinned_) free_pinned_.emplace(b.second, b.first);
    unstamped_pinned_.clear();
    return;
  }
  (void)hipEventRecord(ev, stream_);
  pending_pinned_.push_back({ev, std::move(unstamped_pinned_)});
  unstamped_pinned_.clear();
}

void Runtime::empty_cache() {
  OnDevice here(device_);
  drain_all_inboxes();
  (void)hipStreamSynchronize(stream_);
  std::lock_guard<std::mutex> lk(mu_);
  for (auto& kv : free_dev_) {
    (void)hipFree(kv.second);
    reserved_ -= kv.first;
  }
  free_dev_.clear();
  // the stream is idle: every released pinned block is free whether its group has an event yet or not
  for (auto& b : unstamped_pinned_) free_pinned_.emplace(b.second, b.first);
  unstamped_pinned_.clear();
  for (auto& g : pending_pinned_) {
    for (auto& b : g.blocks) free_pinned_.emplace(b.second, b.first);
    ev_pool_.push_back(g.ev);
  }
  pending_pinned_.clear();
  for (auto& kv : free_pinned_) (void)hipHostFree(kv.second);
  free_pinned_.clear();
}

void Runtime::stats(uint64_t* reserved, uint64_t* in_use) {
  drain_deferred();
  std::lock_guard<std::mutex> lk(mu_);
  if (reserved) *reserved = reserved_;
  if (in_use) *in_use = in_use_;
}

void Runtime::h2d(void* dst, const void* src, size_t bytes) {
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_));
}
void Runtime::d2h_sync(void* dst, const void* src, size_t bytes) {
  // a few bytes (item() of a loss): through a pinned block -- the copy into pageable memory is staged by the HIP
  // runtime and blocks inside the call
  if (bytes && bytes <= 4096) {
    PinnedMemP p = alloc_pinned(bytes);
    d2h_pinned_async(p->ptr, src, bytes);
    drain_while_busy();
    HIP_CHECK(hipStreamSynchronize(stream_));
    std::memcpy(dst, p->ptr, bytes);
    return;
  }
  // reclaim first: a device->host copy into pageable memory blocks inside the copy
  // call until the stream gets there, so this is the last moment the GPU is busy
  if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream_));
  drain_while_busy();
  HIP_CHECK(hipStreamSynchronize(stream_));
}
void Runtime::sync_while_draining() {
  drain_while_busy();
  HIP_CHECK(hipStreamSynchronize(stream_));
}
namespace {
// may a kernel on `dev` touch p?  (A caller's pointer may live on ANOTHER GPU of the process: the runtime's copy
// handles that with or without peer access, a kernel of ours does not.  One GPU in the process: nothing to ask.)
bool local_to(const void* p, int dev) {
  static const int ndev = Runtime::device_count();
  if (ndev <= 1) return true;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost || (a.type == hipMemoryTypeDevice && a.device == dev);
}
}  // namespace
// (up to a megabyte by a kernel of ours -- kernels.h: launch_copy_small; the runtime's copy costs the host about twice
// a launch, and the copies of this size are the ones at the head of a latency chain: setWeights of one utterance)
void Runtime::d2d(void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  if (bytes <= (size_t(1) << 20) && local_to(src, device_) && local_to(dst, device_)) {
    launch_copy_small(dst, src, bytes, stream_);
    return;
  }
  HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream_));
}
// device -> pinned host memory, asynchronous: the same way round (the kernel's stores travel over the host link)
void Runtime::d2h_pinned_async(void* pinned_dst, const void* src, size_t bytes) {
  if (!bytes) return;
  // (a source on ANOTHER GPU of the process -- Weights::ensure_host runs on the caller's device, not the owner's -- is
  //  the runtime's to copy, as in d2d)
  if (bytes <= (size_t(1) << 20) && local_to(src, device_)) {
    launch_copy_small(pinned_dst, src, bytes, stream_);
    return;
  }
  HIP_CHECK(hipMemcpyAsync(pinned_dst, src, bytes, hipMemcpyDeviceToHost, stream_));
}
void Runtime::h2d_pinned(void* dst, const void* pinned_src, size_t bytes) {
  if (!bytes) return;
  static const size_t lim = [] {  // GTNX_H2D_KERNEL_BYTES: largest pinned block a kernel of ours fetches
    const char* e = std::getenv("GTNX_H2D_KERNEL_BYTES");
    return e ? size_t(std::atol(e)) : size_t(1) << 20;
  }();
  // (pinned blocks are mapped into the device's address space: the kernel reads the host's copy.  Measured on the
  //  headline step, whose four table uploads of 28-200 KB were runtime copies: 0.688 -> 0.630 ms per step -- a copy of
  //  the runtime's is ordered against the compute queue from outside it, a kernel is just the next dispatch)
  if (bytes <= lim && local_to(dst, device_)) {
    launch_copy_small(dst, pinned_src, bytes, stream_);
    return;
  }
  HIP_CHECK(hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, stream_));
}

Runtime::MirrorSlot Runtime::mirror_slot() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (!mirror_ring_) {
      OnDevice here(device_);
      void* p = nullptr;
      HIP_CHECK(hipHostMalloc(&p, sizeof(float) * kMirrorSlots, hipHostMallocDefault));
      mirror_gen_ = new std::atomic<uint64_t>[kMirrorSlots];
      for (size_t i = 0; i < kMirrorSlots; ++i) mirror_gen_[i].store(0);
      mirror_ring_ = static_cast<float*>(p);
    }
  }
  const uint64_t g = mirror_next_.fetch_add(1) + 1;  // (never 0)
  const size_t i = size_t(g % kMirrorSlots);
  mirror_gen_[i].store(g);
  return {mirror_ring_ + i, g};
}
bool Runtime::mirror_read(const MirrorSlot& m, float* out) {
  if (!m.ptr || !mirror_ring_) return false;
  const size_t i = size_t(m.ptr - mirror_ring_);
  if (i >= kMirrorSlots || mirror_gen_[i].load() != m.gen) return false;
  sync();
  const float v = *static_cast<volatile float*>(m.ptr);
  if (mirror_gen_[i].load() != m.gen) return false;  // handed out again meanwhile: the value may be the next owner's
  *out = v;
  return true;
}

// ---------------------------------------------------------------- profiler
void Runtime::prof_enable(bool on) {
  if (!on) collect_prof();
  prof_on_ = on;
}
void Runtime::prof_reset() {
  collect_prof();
  prof_.clear();
}

Runtime::Scope::Scope(Runtime* r, const char* name, double bytes) : rt(r), idx(-1) {
  if (!rt->prof_on_) return;
  ProfRec rec;
  rec.name = name;
  rec.bytes = bytes;
  auto get_ev = [&]() {
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
  };
  rec.a = get_ev();
  rec.b = get_ev();
  HIP_CHECK(hipEventRecord(rec.a, rt->stream_));
  idx = (int)rt->prof_recs_.size();
  rt->prof_recs_.push_back(rec);
}
Runtime::Scope::~Scope() {
  if (idx >= 0) (void)hipEventRecord(rt->prof_recs_[idx].b, rt->stream_);
}

void Runtime::prof_add_bytes(const std::string& name, double bytes) { prof_[name].bytes += bytes; }

void Runtime::collect_prof() {
  if (prof_recs_.empty()) return;
  (void)hipStreamSynchronize(stream_);
  for (auto& r : prof_recs_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = prof_[r.name];
      e.total_ms += ms;
      e.launches += 1;
      e.bytes += r.bytes;
    } else {
      (void)hipGetLastError();
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  prof_recs_.clear();
}

ProfEntry Runtime::prof_get(const std::string& name) {
  collect_prof();
  auto it = prof_.find(name);
  return it == prof_.end() ? ProfEntry{} : it->second;
}

std::string Runtime::prof_names() {
  collect_prof();
  std::string s;
  for (auto& kv : prof_) {
    if (!s.empty()) s += "\n";
    s += kv.first;
  }
  return s;
}

} // namespace gtnx

// ---------------------------------------------------------------- host phase timer
#include <chrono>
namespace gtnx {
namespace {
struct HostTable {
  std::mutex mu;
  std::map<std::string, std::pair<double, long>> t;
  ~HostTable() {
    if (!HostTimer::enabled()) return;
    for (auto& kv : t)
      std::fprintf(stderr, "[gtnx host] %-28s calls %6ld  total %9.2f ms  avg %8.3f ms\n", kv.first.c_str(),
                   kv.second.second, kv.second.first, kv.second.first / double(kv.second.second));
  }
};
HostTable& host_table() {
  static HostTable h;
  return h;
}
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
} // namespace
double host_now_ms() { return now_ms(); }
bool HostTimer::enabled() {
  static const bool e = std::getenv("GTNX_HOST_TIMING") != nullptr;
  return e;
}
void host_timer_add(const char* name, double ms) {
  if (!HostTimer::enabled()) return;
  HostTable& h = host_table();
  std::lock_guard<std::mutex> lk(h.mu);
  auto& e = h.t[name];
  e.first += ms;
  e.second += 1;
}
HostTimer::HostTimer(const char* n) : name(n), t0(enabled() ? now_ms() : 0.0) {}
HostTimer::~HostTimer() {
  if (!enabled()) return;
  const double dt = now_ms() - t0;
  HostTable& h = host_table();
  std::lock_guard<std::mutex> lk(h.mu);
  auto& e = h.t[name];
  e.first += dt;
  e.second += 1;
}
} // namespace gtnx
