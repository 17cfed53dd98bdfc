// gtn/parallel.h -- parallelMap (reference gtn/parallel/parallel_map.h:153-188).
// Same contract: maps `function` over the inputs element-wise on host threads,
// size-1 inputs broadcast, results in input order, the first exception is
// rethrown after every task finished.  On this engine it is for HOST-side work
// (building target graphs); graph functions batch through their vector overloads.
#pragma once

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <stdexcept>
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
template <class Body>
void runIndexed(size_t n, Body&& body) {
  // min(size, hardware_concurrency) threads like parallel_map.cpp:18-26, capped:
  // threads are spawned per call here, and host-side graph building saturates
  // long before 64 of them
  const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min(n, hw), 64);
  std::atomic<size_t> next{0};
  std::exception_ptr first;
  std::mutex mu;
  auto worker = [&]() {
    for (size_t i = next++; i < n; i = next++) {
      try {
        body(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(mu);
        if (!first) first = std::current_exception();
      }
    }
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (first) std::rethrow_exception(first);
}
} // namespace detail

template <typename FuncType, typename... Args>
auto parallelMap(FuncType&& function, Args&&... inputs) {
  size_t size = 0;
  (void)std::initializer_list<int>{(size = std::max(size, inputs.size()), 0)...};
  using OutType = decltype(function(detail::pickElem(1, 0, inputs)...));
  if constexpr (std::is_void<OutType>::value) {
    detail::runIndexed(size, [&](size_t i) { function(detail::pickElem(size, i, inputs)...); });
  } else {
    std::vector<OutType> out(size);
    detail::runIndexed(size, [&](size_t i) { out[i] = function(detail::pickElem(size, i, inputs)...); });
    return out;
  }
}
} // namespace gtn
