// gtn/parallel.h -- parallelMap (reference gtn/parallel/parallel_map.h:153-188).
// Same contract: maps `function` over the inputs element-wise on host threads,
// size-1 inputs broadcast, results in input order, the first exception is
// rethrown after every task finished.  The host threads do the HOST-side work of the
// tasks (building target graphs, handing weights over); the graph functions the tasks
// call are deferred by the engine and run at the join as one batched launch each
// (gtnx_parallel_enter / gtnx_parallel_flush, include/gtn_amd.h).
#pragma once

#include "gtn_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <semaphore.h>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <functional>
#include <exception>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace gtn {
namespace detail {
inline void throwStatus(gtnx_status_t st) {  // (as gtn/graph.h: detail::check -- this header stands on its own)
  if (st == GTNX_OK) return;
  const std::string msg = gtnx_last_error();
  switch (st) {
    case GTNX_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case GTNX_LOGIC_ERROR: throw std::logic_error(msg);
    case GTNX_OUT_OF_RANGE: throw std::out_of_range(msg);
    default: throw std::runtime_error(msg);
  }
}
template <class V>
auto pickElem(size_t size, size_t i, const V& v) -> decltype(v[0]) {
  if (v.size() == size) return v[i];
  if (v.size() == 1) return v[0];
  throw std::runtime_error("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
}
// Persistent worker pool (the reference keeps one too: thread_pool.h:27-91, parallel_map.cpp:18-46, grown on
// demand and never shrunk).  Differences that matter on a host that only PREPARES work for the GPU:
//  * the calling thread takes tasks as well, from the first instant -- a map of cheap tasks (512 backward()
//    calls are 512 records of a few dozen nanoseconds) is over before a sleeping thread could have been woken,
//    and nobody waits for threads that never took a task;
//  * a worker that ran out of tasks can keep looking for the next job for a while before it goes to sleep
//    (GTN_AMD_SPIN_US, default 0: measured on the 256-thread host of an MI355X, sleeping workers take their
//    first task 40 us after the call and spinning ones 35 us -- while thirty spinning threads made the tasks
//    themselves two to four times slower).
class Pool {
 public:
  struct Job {
    std::atomic<size_t> next{0};   // next task index to hand out
    std::atomic<size_t> done{0};   // tasks finished
    std::atomic<int> active{0};    // threads inside work() that may still hand in results
    size_t n = 0, grain = 1;
    size_t max_workers = 0;        // pool threads with an index below this take part
    int compose_mode = -1;         // the caller's gtnx_compose_mode: goes with the tasks
    int device = 0;                // ... and its device (gtnx_set_device): the pool's threads work there
    bool region = true;            // announce the threads to the engine (gtnx_parallel_enter / leave)
    void (*run)(void* ctx, size_t i) = nullptr;
    void* ctx = nullptr;
    std::exception_ptr first;
    std::mutex mu;
    std::chrono::steady_clock::time_point t0;  // GTN_AMD_POOL_TRACE
    int trace_slot = 0;                        // (even / odd jobs apart: a step is parallelMap(fwd), parallelMap(bwd))
  };
  // GTN_AMD_POOL_TRACE=1: where the time of a map goes, printed at exit (diagnostic)
  struct Trace {
    std::mutex mu;
    int slot = 0;
    double jobs = 0, threads = 0, wake_us = 0, wake_max_us = 0, busy_us = 0, busy_max_us = 0, caller_us = 0, total_us = 0,
           enter_us = 0, leave_us = 0;
    ~Trace() {
      if (jobs > 0)
        std::fprintf(stderr,
                     "[gtn pool] slot %d: jobs %.0f  threads/job %.1f  first-grab latency avg %.1f us (max %.1f)  busy/thread avg %.1f us (max %.1f)  "
                     "enter %.1f leave %.1f us/thread  caller's share %.1f us  job total %.1f us\n",
                     slot, jobs, threads / jobs, wake_us / std::max(1.0, threads), wake_max_us, busy_us / std::max(1.0, threads),
                     busy_max_us, enter_us / std::max(1.0, threads), leave_us / std::max(1.0, threads), caller_us / jobs,
                     total_us / jobs);
    }
  };
  static Trace& trace(int slot) {
    static Trace t[2];
    return t[slot & 1];
  }
  static bool tracing() {
    static const bool on = std::getenv("GTN_AMD_POOL_TRACE") != nullptr;
    return on;
  }
  /** the pool of a device: every GPU a host drives has its own workers (they are put on that device,
   *  gtnx_set_device, and stay there), so the maps of several device threads run side by side */
  static Pool& get(int device = 0) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<Pool>> pools;
    std::lock_guard<std::mutex> lk(mu);
    const size_t d = device < 0 ? 0 : size_t(device);
    if (pools.size() <= d) pools.resize(d + 1);
    if (!pools[d]) pools[d].reset(new Pool());
    return *pools[d];
  }
  ~Pool() {
    if (tracing())
      std::fprintf(stderr, "[gtn pool] workers at job start: idle %ld  still in tasks %ld  reclaiming %ld (sums over jobs)\n", phaseSum_[0],
                   phaseSum_[1], phaseSum_[2]);
    stop_.store(true, std::memory_order_seq_cst);
    wakeParked(slots_.size());
    for (auto& t : threads_) t.join();
  }
  /** run tasks 0 .. n-1 on up to `nthreads` pool threads and the calling thread.  Nested or concurrent calls run
   *  on the calling thread only. */
  void run(const std::shared_ptr<Job>& job, size_t nthreads) {
    std::unique_lock<std::mutex> call(callMutex_, std::try_to_lock);
    if (!call.owns_lock() || nthreads <= 1 || job->n <= 1) {
      work(*job, true);
      return;
    }
    grow(nthreads);
    job->max_workers = nthreads;
    if (tracing()) {
      int c[3] = {0, 0, 0};
      for (size_t i = 0; i < threads_.size() && i < 64; ++i) c[phase_[i].load(std::memory_order_relaxed)]++;
      phaseSum_[0] += c[0], phaseSum_[1] += c[1], phaseSum_[2] += c[2];
      job->trace_slot = int(traceSeq_++ & 1);
      job->t0 = std::chrono::steady_clock::now();
    }
    {
      lockJob();
      job_ = job;
      unlockJob();
    }
    // (sequentially consistent on both sides: a worker either sees the new epoch or is seen as parked)
    epoch_.fetch_add(1, std::memory_order_seq_cst);
    // a job narrower than the last one also wakes the workers that one used and this one does not: they take no
    // task, but they reclaim what the wider step sent home to them (slices that keep a batch's device memory alive)
    // instead of sleeping on it for as long as the jobs stay narrow
    wakeParked(std::max(nthreads, lastWorkers_));
    lastWorkers_ = nthreads;
    work(*job, true);
    // every task ran and every thread that took one has handed in what it recorded
    size_t spins = 0;
    while (job->done.load(std::memory_order_acquire) < job->n || job->active.load(std::memory_order_acquire) > 0) {
      if (++spins < 4096)
        cpuRelax();
      else
        std::this_thread::yield();
    }
    lockJob();
    job_.reset();
    unlockJob();
    if (tracing()) {
      Trace& t = trace(job->trace_slot);
      std::lock_guard<std::mutex> lk(t.mu);
      t.slot = job->trace_slot;
      t.jobs += 1;
      t.total_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - job->t0).count();
    }
  }

 private:
  Pool() = default;
  // Sleeping workers are woken ONE BY ONE through a semaphore of their own -- not by a condition variable: its
  // waiters all come back through one mutex, and on some kernels every hand-over of that mutex puts the woken
  // thread on the waker's core and the waker to sleep (measured in a microVM: a map over 512 tasks of 10 us ran
  // on one core at a time, 5.3 ms, with eight idle cores next to it).
  struct Slot {
    sem_t sem;
    std::atomic<int> parked{0};
    Slot() { sem_init(&sem, 0, 0); }
    ~Slot() { sem_destroy(&sem); }
  };
  void wakeParked(size_t n) {
    const size_t m = std::min(n, slots_.size());
    for (size_t i = 0; i < m; ++i)
      if (slots_[i]->parked.load(std::memory_order_seq_cst) == 1 && slots_[i]->parked.exchange(0, std::memory_order_seq_cst) == 1)
        sem_post(&slots_[i]->sem);
  }
  static void cpuRelax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void lockJob() {
    while (jobLock_.test_and_set(std::memory_order_acquire)) cpuRelax();
  }
  void unlockJob() { jobLock_.clear(std::memory_order_release); }
  static int64_t spinMicros() {
    static const int64_t us = [] {
      if (const char* e = std::getenv("GTN_AMD_SPIN_US")) return int64_t(std::atol(e));
      return int64_t(0);
    }();
    return us;
  }
  // the share of one thread: tasks in grains until none is left
  static void work(Job& j, bool caller) {
    j.active.fetch_add(1, std::memory_order_acq_rel);
    bool entered = false;
    int oldMode = -1;
    const bool tr = tracing();
    std::chrono::steady_clock::time_point a0, a1, a2, a3;
    for (size_t i0 = j.next.fetch_add(j.grain, std::memory_order_acq_rel); i0 < j.n;
         i0 = j.next.fetch_add(j.grain, std::memory_order_acq_rel)) {
      if (!entered) {
        entered = true;
        if (tr) a0 = std::chrono::steady_clock::now();
        if (j.region) {
          // graph functions called from here are deferred to the join (gtn_amd.h); they run under the compose
          // mode of the thread that called parallelMap
          if (!caller) gtnx_set_device(j.device);
          gtnx_parallel_enter();
          if (!caller) gtnx_compose_mode(j.compose_mode, &oldMode);
        }
        if (tr) a1 = std::chrono::steady_clock::now();
      }
      const size_t i1 = std::min(j.n, i0 + j.grain);
      for (size_t i = i0; i < i1; ++i) {
        try {
          j.run(j.ctx, i);
        } catch (...) {
          std::lock_guard<std::mutex> lk(j.mu);
          if (!j.first) j.first = std::current_exception();
        }
      }
      j.done.fetch_add(i1 - i0, std::memory_order_acq_rel);
    }
    if (tr && entered) a2 = std::chrono::steady_clock::now();
    if (entered && j.region) {
      if (!caller) gtnx_compose_mode(oldMode, nullptr);
      gtnx_parallel_leave();
    }
    if (tr && entered) {
      a3 = std::chrono::steady_clock::now();
      auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
      Trace& t = trace(j.trace_slot);
      std::lock_guard<std::mutex> lk(t.mu);
      t.threads += 1;
      const double w = us(j.t0, a0), b = us(a1, a2);
      t.wake_us += w;
      t.wake_max_us = std::max(t.wake_max_us, w);
      t.busy_us += b;
      t.busy_max_us = std::max(t.busy_max_us, b);
      t.enter_us += us(a0, a1);
      t.leave_us += us(a2, a3);
      if (caller) t.caller_us += us(a0, a3);
    }
    j.active.fetch_sub(1, std::memory_order_acq_rel);
  }
  void grow(size_t n) {
    while (threads_.size() < n) {
      const size_t idx = threads_.size();
      slots_.emplace_back(new Slot());
      Slot* slot = slots_.back().get();
      const uint64_t seen0 = epoch_.load(std::memory_order_acquire);
      threads_.emplace_back([this, idx, seen0, slot]() {
        uint64_t seen = seen0;
        for (;;) {
          // look for the next job: spinning for a while, then asleep
          const int64_t spinUs = spinMicros();
          bool have = false;
          if (spinUs > 0) {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spinUs);
            for (unsigned k = 0;; ++k) {
              if (epoch_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire)) {
                have = true;
                break;
              }
              cpuRelax();
              if ((k & 63) == 63 && std::chrono::steady_clock::now() >= until) break;
            }
          }
          if (!have) {
            // announce, look once more, sleep: whoever publishes a job after the announcement posts the semaphore
            slot->parked.store(1, std::memory_order_seq_cst);
            if (stop_.load(std::memory_order_seq_cst) || epoch_.load(std::memory_order_seq_cst) != seen) {
              if (slot->parked.exchange(0, std::memory_order_seq_cst) == 0) {  // (a post is on its way: take it)
                while (sem_wait(&slot->sem) != 0) {
                }
              }
            } else {
              while (sem_wait(&slot->sem) != 0) {
              }
            }
          }
          if (stop_.load(std::memory_order_acquire)) return;
          seen = epoch_.load(std::memory_order_acquire);
          std::shared_ptr<Job> job;
          lockJob();
          job = job_;
          unlockJob();
          static const bool noReclaim = std::getenv("GTN_AMD_NO_RECLAIM") != nullptr;  // (experiments)
          if (!job || idx >= job->max_workers) {
            job.reset();
            if (!noReclaim) gtnx_reclaim();  // (woken only to empty its list: see run())
            continue;
          }
          phase_[idx & 63].store(1, std::memory_order_relaxed);
          work(*job, false);
          job.reset();
          phase_[idx & 63].store(2, std::memory_order_relaxed);
          // the caller goes on (the engine runs the region's deferred calls now); this thread takes apart
          // what earlier steps let go of meanwhile -- off the caller's critical path, shared with the
          // pool's other threads
          if (!noReclaim) gtnx_reclaim();
          phase_[idx & 63].store(0, std::memory_order_relaxed);
        }
      });
    }
  }
  std::mutex callMutex_;
  std::vector<std::unique_ptr<Slot>> slots_;
  std::vector<std::thread> threads_;
  std::shared_ptr<Job> job_;
  std::atomic_flag jobLock_ = ATOMIC_FLAG_INIT;
  std::atomic<uint64_t> epoch_{0};
  std::atomic<bool> stop_{false};
  uint64_t traceSeq_ = 0;
  size_t lastWorkers_ = 0;  // (under callMutex_)
  std::atomic<int> phase_[64] = {};  // GTN_AMD_POOL_TRACE: 0 idle, 1 in tasks, 2 reclaiming
 public:
  long phaseSum_[3] = {0, 0, 0};
};

template <class Body>
void runIndexed(size_t n, Body&& body, size_t maxThreads = 64, bool region = true) {
  // min(size, hardware_concurrency) threads like parallel_map.cpp:18-26, capped.  The engine defers
  // the graph-function calls of the region's threads to the join (gtnx_parallel_enter), so the threads
  // only build graphs: a few dozen of them finish a batch of targets in well under a millisecond.
  // One process per GPU: the ranks of a node share its cores (torchrun exports LOCAL_WORLD_SIZE),
  // and a step is host-bound, so each rank takes its share instead of oversubscribing.
  static const size_t hw = [] {
    size_t h = std::max<size_t>(1, std::thread::hardware_concurrency());
    if (const char* lws = std::getenv("LOCAL_WORLD_SIZE")) {
      const long r = std::atol(lws);
      if (r > 1) h = std::max<size_t>(1, h / size_t(r));
    }
    return h;
  }();
  // The tasks of a region only do host-side work here (a target graph is a few tens of microseconds), so a
  // thread per 16 tasks is plenty and every further thread is only another wake-up; GTN_AMD_THREADS sets
  // the pool size for callers whose tasks do heavy work of their own.
  static const size_t fixed = [] {
    const char* e = std::getenv("GTN_AMD_THREADS");
    const long v = e ? std::atol(e) : 0;
    return v > 0 ? size_t(v) : size_t(0);
  }();
  if (n == 0) return;
  const size_t light = std::max<size_t>(n >= 2 ? 2 : 1, n / 16);
  const size_t nt = std::min<size_t>(std::min(n, hw), fixed ? fixed : std::min(light, maxThreads));
  auto job = std::make_shared<Pool::Job>();
  job->n = n;
  // a few indices per grab: tasks are small (one target graph each)
  job->grain = std::max<size_t>(1, n / ((nt + 1) * 4));
  job->region = region && nt > 1;
  using BodyT = typename std::remove_reference<Body>::type;
  job->ctx = const_cast<void*>(static_cast<const void*>(&body));
  job->run = [](void* ctx, size_t i) { (*static_cast<BodyT*>(ctx))(i); };
  int device = 0;
  if (job->region) {  // the mode the tasks' compose / intersect calls run under, and the device they run on: the caller's
    int cur = -1;
    if (gtnx_compose_mode(0, &cur) == GTNX_OK) gtnx_compose_mode(cur, nullptr);
    job->compose_mode = cur;
    if (gtnx_get_device(&device) != GTNX_OK) device = 0;
    job->device = device;
  }
  Pool::get(device).run(job, nt);
  // the join: everything the tasks asked the engine for runs now, batched (no-op when nothing was deferred)
  std::exception_ptr first = job->first;
  if (job->region) {
    const gtnx_status_t st = gtnx_parallel_flush();
    if (st != GTNX_OK && !first) {
      // the first failed call's error, as the exception its own function throws (gtn/graph.h: detail::check)
      const std::string msg = gtnx_last_error();
      switch (st) {
        case GTNX_INVALID_ARGUMENT: throw std::invalid_argument(msg);
        case GTNX_LOGIC_ERROR: throw std::logic_error(msg);
        case GTNX_OUT_OF_RANGE: throw std::out_of_range(msg);
        default: throw std::runtime_error(msg);
      }
    }
  }
  if (first) std::rethrow_exception(first);
}
} // namespace detail

template <typename FuncType, typename... Args>
auto parallelMap(FuncType&& function, Args&&... inputs) {
  size_t size = 0;
  (void)std::initializer_list<int>{(size = std::max(size, inputs.size()), 0)...};
  using OutType = decltype(function(detail::pickElem(1, 0, inputs)...));
  if constexpr (std::is_void<OutType>::value) {
    detail::runIndexed(size, [&](size_t i) { function(detail::pickElem(size, i, inputs)...); }, 64);
  } else {
    // results are constructed in place by the tasks (no default-constructed OutType per element first: for a
    // Graph that would be a graph created and thrown away per task), then moved into the vector in order
    struct Slots {
      size_t n;
      typename std::aligned_storage<sizeof(OutType), alignof(OutType)>::type* raw;
      std::vector<unsigned char> made;
      explicit Slots(size_t k)
          : n(k), raw(new typename std::aligned_storage<sizeof(OutType), alignof(OutType)>::type[k ? k : 1]), made(k, 0) {}
      OutType& at(size_t i) { return *reinterpret_cast<OutType*>(&raw[i]); }
      ~Slots() {
        for (size_t i = 0; i < n; ++i)
          if (made[i]) at(i).~OutType();
        delete[] raw;
      }
    } slots(size);
    detail::runIndexed(
        size,
        [&](size_t i) {
          new (&slots.raw[i]) OutType(function(detail::pickElem(size, i, inputs)...));
          slots.made[i] = 1;
        },
        64);
    std::vector<OutType> out;
    out.reserve(size);
    for (size_t i = 0; i < size; ++i) out.emplace_back(std::move(slots.at(i)));
    return out;
  }
}

/** parallelMap over SEVERAL GPUs of one node (every utterance's compose -> forwardScore -> backward is
 *  independent, parallel_map.h:167-179): the inputs are cut into contiguous blocks, one per entry of `devices`, and
 *  each block is mapped by its own host thread on its own device (gtnx_set_device: that device's stream, memory pools
 *  and worker pool) -- exactly as parallelMap would, deferred calls and all.  Results come back in input order; every
 *  result lives on the device of its block (resultDevice(i, n, devices.size()) says which), and so must the device
 *  buffers the tasks hand to setWeights.  No data moves between devices here: gather scalars with
 *  gtnx_comm_all_gather_f32 / sum a shared gradient with gtnx_comm_all_reduce_sum_f32 (include/gtn_amd.h, RCCL). */
inline size_t shardBegin(size_t n, size_t shards, size_t k) { return n * k / shards; }
inline size_t resultDevice(size_t i, size_t n, size_t shards) {
  size_t k = shards ? (i * shards) / (n ? n : 1) : 0;
  while (k + 1 < shards && shardBegin(n, shards, k + 1) <= i) ++k;
  while (k > 0 && shardBegin(n, shards, k) > i) --k;
  return k;
}
template <typename FuncType, typename... Args>
auto parallelMapSharded(const std::vector<int>& devices, FuncType&& function, Args&&... inputs) {
  size_t size = 0;
  (void)std::initializer_list<int>{(size = std::max(size, inputs.size()), 0)...};
  using OutType = decltype(function(detail::pickElem(1, 0, inputs)...));
  static_assert(!std::is_void<OutType>::value, "parallelMapSharded: the mapped function returns a value per input");
  const size_t S = devices.empty() ? 1 : devices.size();
  std::vector<std::vector<OutType>> parts(S);
  std::vector<std::exception_ptr> errs(S);
  auto shard = [&](size_t k) {
    try {
      if (!devices.empty()) detail::throwStatus(gtnx_set_device(devices[k]));
      const size_t b = shardBegin(size, S, k), e = shardBegin(size, S, k + 1);
      std::vector<size_t> idx(e - b);
      for (size_t i = b; i < e; ++i) idx[i - b] = i;
      parts[k] = parallelMap([&](size_t i) { return function(detail::pickElem(size, i, inputs)...); }, idx);
    } catch (...) {
      errs[k] = std::current_exception();
    }
  };
  std::vector<std::thread> threads;
  for (size_t k = 1; k < S; ++k) threads.emplace_back(shard, k);
  int mine = 0;
  const bool had = gtnx_get_device(&mine) == GTNX_OK;
  shard(0);
  if (had && !devices.empty()) gtnx_set_device(mine);  // (the calling thread goes back to where it was)
  for (auto& t : threads) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
  std::vector<OutType> out;
  out.reserve(size);
  for (auto& p : parts)
    for (auto& r : p) out.emplace_back(std::move(r));
  return out;
}
} // namespace gtn