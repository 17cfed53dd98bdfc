// multidev_test.cpp -- ONE process, EIGHT devices: the sharding of BASELINE config C5's batch (4096 utterances, 512
// per GPU) through the C++ API, checked WITHOUT hardware against tools/nullhip's eight fake devices
// (NULLHIP_DEVICES=8: per-device streams, allocations and launch counters; kernels do not execute, NULLHIP_ZERO=1
// makes every "device" value 0).  What is checked is the host logic the engine needs for it (runtime.h: one context
// per device, the calling thread's device; include/gtn/parallel.h: parallelMapSharded; gtn_amd.h: gtnx_comm_*):
//   1. every device runs exactly the launches a single device runs for its 512 utterances, on its own stream, and no
//      launch is made on a stream while another device is current (what the real runtime rejects);
//   2. every device's memory pool is its own;
//   3. results come back in input order and live on their block's device (using one from a thread that is on another
//      device is std::invalid_argument);
//   4. the losses of all devices gathered (all_gather over the RCCL entry points) are in utterance order, and a shared
//      gradient summed over the devices (all_reduce) is the sum.
// Own test program; built by tests/dropin/Makefile, run by tests/test_multidevice_cpu.py under LD_PRELOAD=libnullhip.so.
// (T and C are reduced -- T = 40, C = 16, U = 6 -- the batch and its split are C5's: host logic does not depend on them,
// and eight fake devices' "HBM" is this container's RAM.)
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ++failures;                                                        \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
    }                                                                    \
  } while (0)

static Graph ctcGraph(const std::vector<int>& target) {
  Graph ctc;
  const int L = 2 * (int)target.size() + 1;
  for (int l = 0; l < L; l++) {
    const int idx = (l - 1) / 2;
    ctc.addNode(l == 0, l == L - 1 || l == L - 2);
    const int label = l % 2 ? target[idx] : 0;
    ctc.addArc(l, l, label);
    if (l > 0) ctc.addArc(l - 1, l, label);
    if (l % 2 && l > 1 && label != target[idx - 1]) ctc.addArc(l - 2, l, label);
  }
  ctc.arcSort();
  return ctc;
}

int main() {
  auto counter = [](const char* name, int d) -> long {
    auto fn = reinterpret_cast<long (*)(int)>(dlsym(RTLD_DEFAULT, name));
    return fn ? fn(d) : -1;
  };
  if (counter("nullhip_launches", 0) < 0) {
    std::printf("multidev_test needs LD_PRELOAD=tools/nullhip/_bin/libnullhip.so (NULLHIP_DEVICES=8)\n");
    return 2;
  }
  const int G = gtnx_device_count();
  EXPECT(G == 8);
  if (G < 2) return 1;
  const int B = 4096, per = B / G, T = 40, M = 16, U = 6;
  std::vector<int> devices(G);
  for (int d = 0; d < G; ++d) devices[d] = d;
  std::vector<std::vector<int>> targets(B);
  for (int b = 0; b < B; ++b)
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + (b * 7 + u * 3) % (M - 1));
  // every shard's emissions in ITS device's memory
  std::vector<float*> em(G);
  std::vector<const float*> scores(B);
  for (int d = 0; d < G; ++d) {
    detail::check(gtnx_set_device(d));
    if (hipMalloc(reinterpret_cast<void**>(&em[d]), sizeof(float) * size_t(per) * T * M) != hipSuccess) return 2;
    for (int i = 0; i < per; ++i) scores[size_t(d) * per + i] = em[d] + size_t(i) * T * M;
  }
  detail::check(gtnx_set_device(0));
  auto fwd = [T, M](const std::vector<int>& target, const float* e) {
    auto ctc = ctcGraph(target);
    auto emissions = linearGraph(T, M);
    emissions.setWeights(e);
    return subtract(forwardScore(emissions), forwardScore(intersect(ctc, emissions)));
  };
  auto bwd = [](const Graph& g) {
    backward(g);
    return 0;
  };

  // ---- the single-device run of one shard: what 512 utterances launch on one device
  long single = 0;
  {
    std::vector<std::vector<int>> t0(targets.begin(), targets.begin() + per);
    std::vector<const float*> s0(scores.begin(), scores.begin() + per);
    const long before = counter("nullhip_launches", 0);
    auto losses = parallelMap(fwd, t0, s0);
    parallelMap(bwd, losses);
    EXPECT(losses.size() == size_t(per));
    EXPECT(losses.back().item() == 0.0f);  // (NULLHIP_ZERO)
    single = counter("nullhip_launches", 0) - before;
    EXPECT(single > 0);
  }
  detail::check(gtnx_synchronize());

  // ---- the same over eight devices
  std::vector<long> before(G), alloc_before(G);
  for (int d = 0; d < G; ++d) {
    before[d] = counter("nullhip_launches", d);
    alloc_before[d] = counter("nullhip_alloc_bytes", d);
  }
  auto losses = parallelMapSharded(devices, fwd, targets, scores);
  auto done = parallelMapSharded(devices, bwd, losses);
  EXPECT(losses.size() == size_t(B) && done.size() == size_t(B));
  for (int d = 0; d < G; ++d) {
    const long got = counter("nullhip_launches", d) - before[d];
    if (got != single) std::printf("device %d: %ld launches, a single device runs %ld for its shard\n", d, got, single);
    EXPECT(got == single);
    EXPECT(counter("nullhip_wrong_device_launches", d) == 0);
    if (d > 0) EXPECT(counter("nullhip_alloc_bytes", d) > alloc_before[d]);  // its own pool grew (device 0's was warm)
  }
  int back = -1;
  detail::check(gtnx_get_device(&back));
  EXPECT(back == 0);  // the calling thread is where it was

  // ---- results live on their block's device
  for (int d = 0; d < G; ++d) EXPECT(resultDevice(size_t(d) * per, B, G) == size_t(d) && resultDevice(size_t(d + 1) * per - 1, B, G) == size_t(d));
  {
    bool threw = false;
    try {
      (void)losses[size_t(3) * per + 5].item();  // device 3's, asked for from device 0's thread
    } catch (const std::invalid_argument& e) {
      threw = std::string(e.what()).find("lives on device 3") != std::string::npos;
    }
    EXPECT(threw);
    detail::check(gtnx_set_device(3));
    EXPECT(losses[size_t(3) * per + 5].item() == 0.0f);
    detail::check(gtnx_set_device(0));
  }

  // ---- gather the losses, sum a shared gradient (RCCL entry points; here tools/nullhip's, on host memory)
  {
    gtnx_comm_t comm = nullptr;
    detail::check(gtnx_comm_create(devices.data(), G, &comm));
    std::vector<float*> send(G), recv(G), shared(G);
    for (int d = 0; d < G; ++d) {
      detail::check(gtnx_set_device(d));
      (void)hipMalloc(reinterpret_cast<void**>(&send[d]), 4 * size_t(per));
      (void)hipMalloc(reinterpret_cast<void**>(&recv[d]), 4 * size_t(B));
      (void)hipMalloc(reinterpret_cast<void**>(&shared[d]), 4 * size_t(M * M + M));
      std::vector<float> h(per), g(M * M + M, float(d + 1));
      for (int i = 0; i < per; ++i) h[i] = float(d * per + i);  // "the loss of utterance d * per + i"
      (void)hipMemcpy(send[d], h.data(), 4 * size_t(per), hipMemcpyHostToDevice);
      (void)hipMemcpy(shared[d], g.data(), 4 * g.size(), hipMemcpyHostToDevice);
    }
    detail::check(gtnx_set_device(0));
    std::vector<const void*> sp(send.begin(), send.end());
    std::vector<void*> rp(recv.begin(), recv.end()), gp(shared.begin(), shared.end());
    detail::check(gtnx_comm_all_gather_f32(comm, sp.data(), rp.data(), per));
    detail::check(gtnx_comm_all_reduce_sum_f32(comm, gp.data(), M * M + M));
    for (int d = 0; d < G; ++d) {
      std::vector<float> h(B), g(M * M + M);
      (void)hipMemcpy(h.data(), recv[d], 4 * size_t(B), hipMemcpyDeviceToHost);
      (void)hipMemcpy(g.data(), shared[d], 4 * g.size(), hipMemcpyDeviceToHost);
      bool ordered = true, summed = true;
      for (int i = 0; i < B; ++i) ordered = ordered && h[i] == float(i);
      for (float v : g) summed = summed && v == float(G * (G + 1) / 2);
      EXPECT(ordered);
      EXPECT(summed);
    }
    int n = 0;
    detail::check(gtnx_comm_size(comm, &n));
    EXPECT(n == G);
    detail::check(gtnx_comm_destroy(comm));
  }

  if (failures) {
    std::printf("%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("All tests passed\n");
  return 0;
}
