// inbox_step.cpp -- deferred garbage that belongs to a thread which never comes to a reclamation point:
// a worker thread makes N emission graphs with device-resident weights and then sleeps; the main thread destroys the
// handles (they go home to the worker's list, bounded) and asks for the memory back.  gtnx_empty_cache() has to reach
// every thread's list: the engine's in-use device bytes must drop to zero although the worker never synchronised.
// Diagnostic / CPU test only (tests/test_hostpath_cpu.py), against the null HIP device.
//   LD_PRELOAD=tools/nullhip/_bin/libnullhip.so tools/nullhip/_bin/inbox_step [N]
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gtn/gtn.h"

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20000, T = 16, C = 8;
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * T * C) != hipSuccess) return 2;
  std::vector<gtnx_graph_t> hs(static_cast<size_t>(N));
  std::atomic<int> made{0}, quit{0};
  std::thread worker([&] {
    for (int i = 0; i < N; ++i) {
      gtn::detail::check(gtnx_linear_graph(T, C, 1, &hs[size_t(i)]));
      gtn::detail::check(gtnx_graph_set_weights_device(hs[size_t(i)], dev));  // a device block of its own per graph
    }
    made.store(1);
    while (!quit.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // never syncs, never enters a region
  });
  while (!made.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  uint64_t res0 = 0, use0 = 0, res1 = 0, use1 = 0, res2 = 0, use2 = 0;
  gtn::detail::check(gtnx_memory_stats(&res0, &use0));
  for (int i = 0; i < N; ++i) gtn::detail::check(gtnx_graph_destroy(hs[size_t(i)]));  // home: the sleeping worker's list
  gtn::detail::check(gtnx_memory_stats(&res1, &use1));
  gtn::detail::check(gtnx_empty_cache());
  gtn::detail::check(gtnx_memory_stats(&res2, &use2));
  std::printf("in use: %llu bytes with the graphs alive, %llu after destroying the handles on another thread, %llu after empty_cache; "
              "reserved %llu -> %llu\n",
              (unsigned long long)use0, (unsigned long long)use1, (unsigned long long)use2, (unsigned long long)res0,
              (unsigned long long)res2);
  quit.store(1);
  worker.join();
  const bool ok = use0 >= uint64_t(N) * 512 && use2 == 0 && res2 == 0;
  std::printf("%s\n", ok ? "INBOX_OK" : "INBOX_FAIL");
  return ok ? 0 : 1;
}
