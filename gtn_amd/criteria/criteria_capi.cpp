// criteria_capi.cpp -- C entry points of the batched criteria (libgtn_criteria.so), for
// callers that are not C++: gtn_amd/torch_loss.py binds gtn_ctc_loss_n with ctypes.
// Built with plain g++ against include/gtn and libgtn_amd.so.
#include <cstring>
#include <string>

#include "ctc_criterion.h"

namespace {
thread_local std::string g_err;
}

extern "C" __attribute__((visibility("default"))) const char* gtn_criteria_last_error(void) { return g_err.c_str(); }

// emissions: DEVICE float [B][T][C]; targets: host int32, concatenated; lengths: host int32 [B];
// loss: DEVICE float [B]; grad: DEVICE float [B][T][C] or null (then no backward pass).
// Returns 0, or -1 with the message in gtn_criteria_last_error().
extern "C" __attribute__((visibility("default"))) int gtn_ctc_loss_n(const void* emissions, const int* targets,
                                                                     const int* lengths, int B, int T, int C,
                                                                     int blank, void* loss, void* grad) {
  try {
    std::vector<std::vector<int>> tg(B);
    size_t o = 0;
    for (int b = 0; b < B; ++b) {
      tg[b].assign(targets + o, targets + o + lengths[b]);
      o += size_t(lengths[b]);
    }
    gtn::criteria::ctcLossBatch(emissions, tg, T, C, blank, loss, grad, /*targetGrad=*/false);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
