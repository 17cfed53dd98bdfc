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
#include <condition_variable>
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
template <class V>
auto pickElem(size_t size, size_t i, const V& v) -> decltype(v[0]) {
  if (v.size() == size) return v[i];
  if (v.size() == 1) return v[0];
  throw std::runtime_error("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
}
// Persistent worker pool (the reference keeps one too: thread_pool.h:27-91,
// parallel_map.cpp:18-46, grown on demand and never shrunk).  A job is a callable
// every participating thread runs once; the caller takes part as well.
class Pool {
 public:
  static Pool& get() {
    static Pool p;
    return p;
  }
  /** run `job` on `nthreads` pool threads (the caller runs `callerFirst` and waits: what it allocates -- it
   *  runs the region's deferred calls afterwards -- stays apart from what the pool's threads build and take
   *  down).  Nested or concurrent calls run on the calling thread only. */
  template <class Job, class Pre>
  void run(size_t nthreads, Job&& job, Pre&& callerFirst) {
    std::unique_lock<std::mutex> call(callMutex_, std::try_to_lock);
    if (!call.owns_lock() || nthreads <= 1) {
      callerFirst();
      job();
      return;
    }
    grow(nthreads);
    {
      std::lock_guard<std::mutex> lk(mutex_);
      job_ = [&job] { job(); };
      want_ = nthreads;
      pending_ = nthreads;
      ++epoch_;
    }
    wake_.notify_all();
    callerFirst();  // the workers are already running
    std::unique_lock<std::mutex> lk(mutex_);
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  Pool() = default;
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mutex_);
      stop_ = true;
    }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
  }
  void grow(size_t n) {
    while (threads_.size() < n) {
      const size_t idx = threads_.size();
      uint64_t seen;
      {
        std::lock_guard<std::mutex> lk(mutex_);
        seen = epoch_;
      }
      threads_.emplace_back([this, idx, seen]() mutable {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mutex_);
            wake_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            if (idx >= want_) continue;
            job = job_;
          }
          job();
          {
            std::lock_guard<std::mutex> lk(mutex_);
            --pending_;
          }
          done_.notify_one();
          // the caller goes on (the engine runs the region's deferred calls now); this thread takes apart
          // what earlier steps let go of meanwhile -- off the caller's critical path, shared with the
          // pool's other threads
          gtnx_reclaim();
        }
      });
    }
  }
  std::mutex callMutex_, mutex_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> threads_;
  std::function<void()> job_;
  uint64_t epoch_ = 0;
  size_t want_ = 0, pending_ = 0;
  bool stop_ = false;
};

inline void noPrelude() {}

template <class Body, class Pre = void (*)()>
void runIndexed(size_t n, Body&& body, size_t maxThreads = 64, Pre callerFirst = &noPrelude) {
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
  const size_t light = std::max<size_t>(n >= 2 ? 2 : 1, n / 16);
  const size_t nt = std::min<size_t>(std::min(n, hw), fixed ? fixed : std::min(light, maxThreads));
  std::atomic<size_t> next{0};
  std::exception_ptr first;
  std::mutex mu;
  auto worker = [&]() {
    // graph functions called from here are deferred to the join below (gtn_amd.h)
    struct Region {
      bool on;
      explicit Region(bool o) : on(o) {
        if (on) gtnx_parallel_enter();
      }
      ~Region() {
        if (on) gtnx_parallel_leave();
      }
    } region(nt > 1);
    // a few indices per grab: tasks are small (one target graph each)
    const size_t grain = std::max<size_t>(1, n / (nt * 4));
    for (size_t i0 = next.fetch_add(grain); i0 < n; i0 = next.fetch_add(grain)) {
      for (size_t i = i0; i < std::min(n, i0 + grain); ++i) {
        try {
          body(i);
        } catch (...) {
          std::lock_guard<std::mutex> lk(mu);
          if (!first) first = std::current_exception();
        }
      }
    }
  };
  Pool::get().run(nt, worker, callerFirst);
  // the join: everything the tasks asked the engine for runs now, batched (no-op when nothing was deferred)
  if (nt > 1) {
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
  auto prelude = [] {};
  if constexpr (std::is_void<OutType>::value) {
    detail::runIndexed(size, [&](size_t i) { function(detail::pickElem(size, i, inputs)...); }, 64, prelude);
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
        64, prelude);
    std::vector<OutType> out;
    out.reserve(size);
    for (size_t i = 0; i < size; ++i) out.emplace_back(std::move(slots.at(i)));
    return out;
  }
}
} // namespace gtn
