// asg_step.cpp -- the batched ASG criterion (gtn_amd/lib/libgtn_criteria.so: gtn_asg_loss_n) against the
// null HIP device: host-side smoke / sanitizer target.  With NULLHIP_ZERO=1 every device-built graph reads
// as EMPTY (sizes 0), which drives the empty-graph corners of compose and the symbolic products.
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <vector>

using asg_fn = int (*)(const void*, const int*, const int*, int, int, int, const void*, void*, void*, void*);

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 3, B = argc > 2 ? atoi(argv[2]) : 8;
  const int T = argc > 3 ? atoi(argv[3]) : 30, N = argc > 4 ? atoi(argv[4]) : 12, U = argc > 5 ? atoi(argv[5]) : 5;
  void* h = dlopen(argc > 6 ? argv[6] : "gtn_amd/lib/libgtn_criteria.so", RTLD_NOW);
  if (!h) { std::fprintf(stderr, "%s\n", dlerror()); return 1; }
  auto asg = reinterpret_cast<asg_fn>(dlsym(h, "gtn_asg_loss_n"));
  auto err = reinterpret_cast<const char* (*)()>(dlsym(h, "gtn_criteria_last_error"));
  std::vector<float> em(size_t(B) * T * N, 0.5f), gem(em.size()), tw(N + N * N, 0.1f), gtw(tw.size()), loss(B);
  std::vector<int> tg(size_t(B) * U), len(B, U);
  for (size_t i = 0; i < tg.size(); ++i) tg[i] = int(i * 7 % N);
  for (int s = 0; s < steps; ++s) {
    const int rc = asg(em.data(), tg.data(), len.data(), B, T, N, tw.data(), loss.data(), gem.data(), gtw.data());
    std::printf("step %d rc=%d %s\n", s, rc, rc ? err() : "");
  }
  return 0;
}
