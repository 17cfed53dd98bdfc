// host_step.cpp -- runs the native CTC benchmark step (bench_native/libgtn_bench.so) against the
// null HIP device of nullhip.cpp and prints host milliseconds per step.  Diagnostic only.
//   make -C tools/nullhip && tools/nullhip/run.sh [steps] [B] [T] [C] [U]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <algorithm>
#include <vector>

#include "sampler.h"

using step_fn = int (*)(const void*, const int*, int, int, int, int, void*, void*);

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 20;
  const int B = argc > 2 ? atoi(argv[2]) : 512, T = argc > 3 ? atoi(argv[3]) : 1000;
  const int C = argc > 4 ? atoi(argv[4]) : 256, U = argc > 5 ? atoi(argv[5]) : 100;
  void* h = dlopen(argc > 6 ? argv[6] : "bench_native/libgtn_bench.so", RTLD_NOW);
  if (!h) { std::fprintf(stderr, "%s\n", dlerror()); return 1; }
  const char* fn = std::getenv("HOST_STEP_FN");  // e.g. gtn_bench_ctc_step_vector
  auto step = reinterpret_cast<step_fn>(dlsym(h, fn ? fn : "gtn_bench_ctc_step"));
  if (!step) { std::fprintf(stderr, "no such step function\n"); return 1; }
  void* eng = dlopen("libgtn_amd.so", RTLD_NOW | RTLD_NOLOAD);  // the copy the bench library pulled in
  if (!eng) eng = dlopen("gtn_amd/lib/libgtn_amd.so", RTLD_NOW);
  auto sync = reinterpret_cast<int (*)()>(dlsym(eng, "gtnx_synchronize"));  // reclaims what the step let go of
  std::vector<float> em(size_t(B) * T * C, 0.0f), grad(size_t(B) * T * C), loss(B);
  // HOST_STEP_DEVICE=1 (on a GPU box, WITHOUT the nullhip preload): the three buffers in device memory, so the
  // real step runs and the sampler sees the host side as it is next to a working GPU
  float *d_em = em.data(), *d_grad = grad.data(), *d_loss = loss.data();
  if (std::getenv("HOST_STEP_DEVICE")) {
    void* hip = dlopen("libamdhip64.so", RTLD_NOW);
    auto hmalloc = reinterpret_cast<int (*)(void**, size_t)>(dlsym(hip, "hipMalloc"));
    auto hmemset = reinterpret_cast<int (*)(void*, int, size_t)>(dlsym(hip, "hipMemset"));
    hmalloc(reinterpret_cast<void**>(&d_em), em.size() * 4);
    hmalloc(reinterpret_cast<void**>(&d_grad), grad.size() * 4);
    hmalloc(reinterpret_cast<void**>(&d_loss), loss.size() * 4);
    hmemset(d_em, 0, em.size() * 4);
  }
  std::vector<int> tg(size_t(B) * U);
  for (size_t i = 0; i < tg.size(); ++i) tg[i] = 1 + int((i * 2654435761u >> 7) % unsigned(C - 1));
  double best = 1e30, sum = 0, rsum = 0;
  long flt0 = 0;
  std::vector<double> all;
  const char* sample = std::getenv("GTN_HOST_SAMPLE");
  const bool no_sync = std::getenv("HOST_STEP_NO_SYNC") != nullptr;
  for (int s = 0; s < steps; ++s) {
    if (s == 3 && sample) start_sampler();
    if (s == 3) {
      struct rusage ru;
      getrusage(RUSAGE_SELF, &ru);
      flt0 = ru.ru_minflt;
    }
    auto t0 = std::chrono::steady_clock::now();
    if (step(d_em, tg.data(), B, T, C, U, d_loss, d_grad) != 0) { std::fprintf(stderr, "step failed\n"); return 2; }
    auto t1 = std::chrono::steady_clock::now();
    if (!no_sync) sync();  // HOST_STEP_NO_SYNC=1: like bench.py, nothing between the steps
    auto t2 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (s >= 3) { all.push_back(ms); best = ms < best ? ms : best; sum += ms; rsum += std::chrono::duration<double, std::milli>(t2 - t1).count(); }
  }
  if (sample) dump_samples(sample);
  {
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    std::printf("minor page faults per step: %.0f\n", double(ru.ru_minflt - flt0) / (steps - 3));
  }
  if (all.empty()) {  // (the first three steps are warm-up)
    std::printf("host ms/step: not measured (steps <= 3)\n");
    return 0;
  }
  std::sort(all.begin(), all.end());
  std::printf("host ms/step: best %.2f median %.2f mean %.2f + reclaim %.2f (B=%d T=%d C=%d U=%d)\n", best,
              all[all.size() / 2], sum / (steps - 3), rsum / (steps - 3), B, T, C, U);
  return 0;
}
