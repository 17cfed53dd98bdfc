// criteria_capi.cpp -- C entry points of the batched criteria (libgtn_criteria.so), for
// callers that are not C++: gtn_amd/torch_loss.py binds gtn_ctc_loss_n with ctypes.
// Built with plain g++ against include/gtn and libgtn_amd.so.
#include <cstring>
#include <string>

#include <map>
#include <mutex>

#include "asg_criterion.h"
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
    gtn::criteria::ctcLossBatch(emissions, targets, lengths, B, T, C, blank, loss, grad, /*targetGrad=*/false);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// The same, as benchmarks/ctc.cpp:150-165 runs it: target graphs with calcGrad = true, and THEIR gradients too.
// target_grad: DEVICE float, utterance b's arc gradients (arc ids of benchmarks/ctc.cpp:40-58's addArc order) at
// target_grad + target_grad_offsets[b]; grad must be non-null.
extern "C" __attribute__((visibility("default"))) int gtn_ctc_loss_target_grads_n(
    const void* emissions, const int* targets, const int* lengths, int B, int T, int C, int blank, void* loss,
    void* grad, void* target_grad, const int64_t* target_grad_offsets) {
  try {
    gtn::Batch ctcs;
    gtn::criteria::ctcLossBatch(emissions, targets, lengths, B, T, C, blank, loss, grad, /*targetGrad=*/true, nullptr, &ctcs);
    if (target_grad) ctcs.gradsToDevice(target_grad, target_grad_offsets);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// ASG over a shared transitions graph.  emissions: DEVICE float [B][T][N]; targets / lengths:
// host int32 (concatenated / [B]); trans_w: DEVICE float [N + N*N] in the arc order of
// gtn::criteria::asgTransitions BEFORE its arcSort (arc i: <s> -> i; arc N + i*N + j: j -> i);
// loss: DEVICE float [B]; grad_em: DEVICE [B][T][N] or null; grad_trans: DEVICE [N + N*N] or null.
extern "C" __attribute__((visibility("default"))) int gtn_asg_loss_n(const void* emissions, const int* targets,
                                                                     const int* lengths, int B, int T, int N,
                                                                     const void* trans_w, void* loss, void* grad_em,
                                                                     void* grad_trans) {
  try {
    // the transitions STRUCTURE is kept across calls (a trainer only changes the weights)
    static std::mutex mu;
    static auto* cache = new std::map<int, gtn::Graph>();  // never destroyed: outlives the engine's teardown
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache->find(N);
    if (it == cache->end()) it = cache->emplace(N, gtn::criteria::asgTransitions(N)).first;
    gtn::Graph& trans = it->second;
    trans.setCalcGrad(grad_trans != nullptr);
    trans.zeroGrad();
    trans.setWeightsDevice(trans_w);  // arc ids are creation order: arcSort permutes lists, not ids
    gtn::criteria::asgLossBatch(emissions, targets, lengths, B, T, N, trans, loss, grad_em);
    if (grad_trans) {
      gtnx_graph_t h = trans.handle();
      int64_t off = 0;
      gtn::detail::check(gtnx_grads_device_n(&h, 1, grad_trans, &off));
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
