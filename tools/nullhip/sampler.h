// sampler.h -- main-thread stack sampler shared by the nullhip diagnostics (host_step.cpp, region_step.cpp)
#pragma once
#include <execinfo.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <cstdio>

// GTN_HOST_SAMPLE=<file>: sample the MAIN thread's call stack every 100 us of its CPU time
// and dump raw return addresses (+ /proc/self/maps) for tools/nullhip/report.py
namespace {
constexpr int kDepth = 24;
constexpr int kMaxSamples = 400000;
void* g_samples[kMaxSamples][kDepth];
int g_depth[kMaxSamples];
volatile int g_ns = 0;
void on_prof(int, siginfo_t*, void*) {
  const int i = g_ns;
  if (i >= kMaxSamples) return;
  g_depth[i] = backtrace(g_samples[i], kDepth);
  g_ns = i + 1;
}
void start_sampler() {
  void* warm[4];
  backtrace(warm, 4);  // loads libgcc outside the handler
  struct sigaction sa {};
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
  struct sigevent sev {};
  sev.sigev_notify = SIGEV_THREAD_ID;
  sev.sigev_signo = SIGPROF;
  sev._sigev_un._tid = int(syscall(SYS_gettid));
  timer_t tm;
  timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &tm);
  struct itimerspec its {};
  its.it_interval.tv_nsec = its.it_value.tv_nsec = 100000;
  timer_settime(tm, 0, &its, nullptr);
}
void dump_samples(const char* path) {
  FILE* f = std::fopen(path, "w");
  FILE* m = std::fopen("/proc/self/maps", "r");
  char line[512];
  while (std::fgets(line, sizeof line, m)) std::fprintf(f, "M %s", line);
  std::fclose(m);
  for (int i = 0; i < g_ns; ++i) {
    std::fprintf(f, "S");
    for (int d = 0; d < g_depth[i]; ++d) std::fprintf(f, " %p", g_samples[i][d]);
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}
} // namespace

