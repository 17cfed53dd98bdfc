// levelize.hip -- the level schedule of shortest distance (kernels.h: DSched) built ON THE DEVICE from a
// graph's CSR arrays, for structures that only exist there (composition results that are not layered:
// level-skipping arcs).  The host pre-pass (graph.cpp: build_host_schedule) replays the reference's Kahn
// FIFO (shortest.cpp:96-145) and needs the arrays downloaded first; this one keeps them where they are.
//
// What a schedule has to get right for forwardScore / viterbiScore and their gradients is the SET of nodes the
// reference's queue ever reaches, their dependency depth, and -- for the gradient -- the set its reverse queue
// reaches (shortest.cpp:45-53); the order of the nodes INSIDE a level is free (rows keep in-list order, which is
// what ties are broken by).  Both sets are fixpoints of a pull rule and are found by Jacobi sweeps, one launch
// per level, both directions in the same launch:
//   forward : level[n] = it   once every in-arc's source has a level < it   (seeds: start nodes without in-arcs)
//   reverse : done[n]  = it   once every out-arc's destination is done < it (seeds: accept nodes without out-arcs)
// A launch that finds the previous one made no progress returns at once, and the host looks at the progress
// counters every 128 launches (one 8-byte read), so a 1000-level lattice costs ~1000 dependent launches of a few
// microseconds and 8 small reads -- no transfer of the graph in either direction.  viterbiPath's tie-break needs
// the queue ORDER itself: that stays with the host pre-pass (ensure_schedule_batch(need_rank)).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

namespace gtnx {
namespace {

constexpr int kLB = 256;
constexpr int kUnset = -1;

__device__ __forceinline__ int arc_of(const GTNX_G int* list, int k) { return list ? list[k] : k; }

__global__ void lv_init_kernel(DGraph g, int* level, int* done, int* progress, int n_progress) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < n_progress) progress[n] = 0;
  if (n >= g.N) return;
  const uint8_t f = g.nflags[n];
  const int indeg = g.in_off[n + 1] - g.in_off[n], outdeg = g.out_off[n + 1] - g.out_off[n];
  level[n] = (indeg == 0 && (f & NF_START)) ? 0 : kUnset;
  done[n] = (outdeg == 0 && (f & NF_ACCEPT)) ? 0 : kUnset;
}

// iteration `it` (>= 1).  progress[2 it], progress[2 it + 1]: nodes levelled / done in this iteration
__global__ void lv_sweep_kernel(DGraph g, int* level, int* done, int* progress, int it) {
  if (it >= 2 && progress[2 * (it - 1)] == 0 && progress[2 * (it - 1) + 1] == 0) return;  // fixpoint reached
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.N) return;
  if (level[n] == kUnset) {
    const int k0 = g.in_off[n], k1 = g.in_off[n + 1];
    bool ready = k1 > k0;  // (a node without in-arcs is a seed or never queued)
    for (int k = k0; k < k1 && ready; ++k) {
      const int lv = level[g.src[arc_of(g.in_list, k)]];
      ready = lv >= 0 && lv < it;  // a level written by THIS launch does not count yet
    }
    if (ready) {
      level[n] = it;
      atomicAdd(&progress[2 * it], 1);
    }
  }
  if (done[n] == kUnset) {
    const int k0 = g.out_off[n], k1 = g.out_off[n + 1];
    bool ready = k1 > k0;
    for (int k = k0; k < k1 && ready; ++k) {
      const int dv = done[g.dst[arc_of(g.out_list, k)]];
      ready = dv >= 0 && dv < it;
    }
    if (ready) {
      done[n] = it;
      atomicAdd(&progress[2 * it + 1], 1);
    }
  }
}

// shortest.cpp:148-152: an accept node the queue never reached is an error if it still waits on a
// predecessor; a non-start accept node WITHOUT in-arcs is never queued but takes part in the final
// reduction with its zero-initialised score: scheduled at level 0, flagged
__global__ void lv_accept_kernel(DGraph g, int* level, uint8_t* orphan, int* info) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.n_accept) return;
  const int n = g.accept_list[k];
  if (level[n] != kUnset) return;
  if (g.in_off[n + 1] - g.in_off[n] > 0) {
    info[LV_ERROR] = 1;
  } else {
    level[n] = 0;
    orphan[n] = 1;
  }
}

__global__ void lv_keys_kernel(int N, const int* level, unsigned* keys, int* ids, int* level_cnt, int* info) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int lv = level[n];
  keys[n] = lv >= 0 ? unsigned(lv) : 0x7fffffffu;
  ids[n] = n;
  if (lv >= 0) {
    atomicAdd(&level_cnt[lv], 1);
    atomicAdd(&info[LV_P], 1);
  }
}

// positions, flags and degrees in position order
__global__ void lv_pos_kernel(DGraph g, int P, const int* order, const uint8_t* orphan, int* pos, uint8_t* pflags, int* indeg_p) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.N) return;
  const int n = order[p];
  if (p < P) {
    pos[n] = p;
    pflags[p] = uint8_t(g.nflags[n] | (orphan[n] ? NF_ORPHAN : 0));
    indeg_p[p] = g.in_off[n + 1] - g.in_off[n];
  } else {
    pos[n] = -1;
  }
  if (p == 0) indeg_p[P] = 0;
}

__global__ void lv_rows_kernel(DGraph g, int P, const int* order, const int* pos, const int* done, const int* row_off,
                               int* in_srcpos, int* in_arc, int* outcnt_p) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > P) return;
  if (p == P) {
    outcnt_p[P] = 0;
    return;
  }
  const int n = order[p];
  int o = row_off[p];
  for (int k = g.in_off[n]; k < g.in_off[n + 1]; ++k, ++o) {
    const int a = arc_of(g.in_list, k);
    in_arc[o] = a;
    in_srcpos[o] = pos[g.src[a]];  // >= 0: every predecessor of a scheduled node is scheduled
  }
  // out rows keep the arcs whose destination is scheduled AND reached by the reverse queue (shortest.cpp:76-78)
  int cnt = 0;
  for (int k = g.out_off[n]; k < g.out_off[n + 1]; ++k) {
    const int d = g.dst[arc_of(g.out_list, k)];
    cnt += (pos[d] >= 0 && done[d] >= 0) ? 1 : 0;
  }
  outcnt_p[p] = cnt;
}

__global__ void lv_out_kernel(DGraph g, int P, const int* order, const int* pos, const int* done, const int* out_off,
                              int* out_dstpos, int* out_arc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int n = order[p];
  int o = out_off[p];
  for (int k = g.out_off[n]; k < g.out_off[n + 1]; ++k) {
    const int a = arc_of(g.out_list, k);
    const int d = g.dst[a];
    if (pos[d] >= 0 && done[d] >= 0) {
      out_dstpos[o] = pos[d];
      out_arc[o] = a;
      ++o;
    }
  }
}

// per level: width, in-arcs entering it, reach (last position + 1 - smallest source position); maxima into info
__global__ void lv_stats_kernel(int L, const int* level_off, const int* row_off, const int* in_srcpos, int* info) {
  const int l = blockIdx.x;
  if (l >= L) return;
  const int lo = level_off[l], hi = level_off[l + 1];
  int mn = lo;
  for (int k = row_off[lo] + threadIdx.x; k < row_off[hi]; k += blockDim.x) mn = min(mn, in_srcpos[k]);
  __shared__ int red[kLB];
  red[threadIdx.x] = mn;
  __syncthreads();
  for (int o = kLB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMax(&info[LV_MAX_WIDTH], hi - lo);
    atomicMax(&info[LV_MAX_LEVEL_ARCS], row_off[hi] - row_off[lo]);
    atomicMax(&info[LV_MAX_REACH], hi - red[0]);
  }
}

__global__ void lv_accpos_kernel(DGraph g, const int* pos, int* acc_pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < g.n_accept) acc_pos[k] = pos[g.accept_list[k]];
}

unsigned blocks(int64_t n) { return unsigned(std::max<int64_t>(1, (n + kLB - 1) / kLB)); }

size_t al(size_t b) { return (b + 255) & ~size_t(255); }

int bits_for(int n) {
  int b = 1;
  while ((1ll << b) < n) ++b;
  return b;
}

}  // namespace

// scratch layout (ints unless noted): level[N] done[N] order[N] ids[N] keys[N] keys_out[N] pos[N] indeg_p[N+1]
// outcnt_p[N+1] level_cnt[N+2] progress[2(N+3)] info[LV_INFO_INTS] orphan[N bytes] + the primitives' own scratch
size_t levelize_scratch_bytes(int N, int A) {
  (void)A;
  size_t sort_b = 0, scan_b = 0;
  const size_t n1 = size_t(N > 0 ? N : 1);
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, n1,
                                  0, 32);
  (void)rocprim::exclusive_scan(nullptr, scan_b, (const int*)nullptr, (int*)nullptr, 0, n1 + 2, rocprim::plus<int>());
  return 7 * al(4 * n1) + 2 * al(4 * (n1 + 1)) + al(4 * (n1 + 2)) + al(8 * (n1 + 3)) + al(4 * LV_INFO_INTS) + al(n1) +
         al(std::max(sort_b, scan_b)) + 256;
}

// Everything is queued on `st`; the call returns after its last (small) read-back.  `info_host`: LV_INFO_INTS ints.
void device_levelize(const DGraph& g, const LevelizeOut& out, void* scratch, int* info_host, hipStream_t st) {
  const int N = g.N;
  const size_t n1 = size_t(N > 0 ? N : 1);
  char* c = static_cast<char*>(scratch);
  auto take = [&](size_t bytes) {
    char* p = c;
    c += al(bytes);
    return p;
  };
  int* level = reinterpret_cast<int*>(take(4 * n1));
  int* done = reinterpret_cast<int*>(take(4 * n1));
  int* order = reinterpret_cast<int*>(take(4 * n1));
  int* ids = reinterpret_cast<int*>(take(4 * n1));
  unsigned* keys = reinterpret_cast<unsigned*>(take(4 * n1));
  unsigned* keys_out = reinterpret_cast<unsigned*>(take(4 * n1));
  int* pos = reinterpret_cast<int*>(take(4 * n1));
  int* indeg_p = reinterpret_cast<int*>(take(4 * (n1 + 1)));
  int* outcnt_p = reinterpret_cast<int*>(take(4 * (n1 + 1)));
  int* level_cnt = reinterpret_cast<int*>(take(4 * (n1 + 2)));
  int* progress = reinterpret_cast<int*>(take(8 * (n1 + 3)));
  int* info = reinterpret_cast<int*>(take(4 * LV_INFO_INTS));
  uint8_t* orphan = reinterpret_cast<uint8_t*>(take(n1));
  void* prim = c;
  size_t sort_b = 0, scan_b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, n1,
                                  0, 32);
  (void)rocprim::exclusive_scan(nullptr, scan_b, (const int*)nullptr, (int*)nullptr, 0, n1 + 2, rocprim::plus<int>());

  std::memset(info_host, 0, sizeof(int) * LV_INFO_INTS);
  (void)hipMemsetAsync(info, 0, 4 * LV_INFO_INTS, st);
  (void)hipMemsetAsync(orphan, 0, n1, st);
  (void)hipMemsetAsync(level_cnt, 0, 4 * (n1 + 2), st);
  if (N <= 0) return;
  const int n_progress = 2 * (N + 3);
  hipLaunchKernelGGL(lv_init_kernel, dim3(blocks(std::max(N, n_progress))), dim3(kLB), 0, st, g, level, done, progress, n_progress);
  // ---- the two fixpoints, one launch per level; a look at the counters every 128 launches
  int it = 1, last_fwd = 0;
  const int it_max = N + 1;  // a DAG of N nodes is at most N levels deep; beyond that nothing can change
  bool fix = false;
  while (!fix && it <= it_max) {
    const int batch_end = std::min(it + 127, it_max);
    for (; it <= batch_end; ++it) hipLaunchKernelGGL(lv_sweep_kernel, dim3(blocks(N)), dim3(kLB), 0, st, g, level, done, progress, it);
    int pr[2];
    (void)hipMemcpyAsync(pr, progress + 2 * batch_end, 8, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    fix = pr[0] == 0 && pr[1] == 0;
  }
  {  // the deepest level: the last iteration that levelled anything
    std::vector<int> pr(size_t(2) * size_t(it));
    (void)hipMemcpyAsync(pr.data(), progress, 8 * size_t(it), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    for (int i = 1; i < it; ++i)
      if (pr[size_t(2) * i] > 0) last_fwd = i;
  }
  const int L = last_fwd + 1;  // (orphans sit at level 0)
  if (g.n_accept > 0) hipLaunchKernelGGL(lv_accept_kernel, dim3(blocks(g.n_accept)), dim3(kLB), 0, st, g, level, orphan, info);
  hipLaunchKernelGGL(lv_keys_kernel, dim3(blocks(N)), dim3(kLB), 0, st, N, (const int*)level, keys, ids, level_cnt, info);
  size_t b = sort_b;
  (void)rocprim::radix_sort_pairs(prim, b, (const unsigned*)keys, keys_out, (const int*)ids, order, size_t(N), 0, 32, st);
  b = scan_b;
  (void)rocprim::exclusive_scan(prim, b, (const int*)level_cnt, out.level_off, 0, size_t(L) + 1, rocprim::plus<int>(), st);
  (void)hipMemcpyAsync(info_host, info, 4 * LV_INFO_INTS, hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
  const int P = info_host[LV_P];
  hipLaunchKernelGGL(lv_pos_kernel, dim3(blocks(N)), dim3(kLB), 0, st, g, P, (const int*)order, (const uint8_t*)orphan, pos, out.pflags,
                     indeg_p);
  b = scan_b;
  (void)rocprim::exclusive_scan(prim, b, (const int*)indeg_p, out.row_off, 0, size_t(P) + 1, rocprim::plus<int>(), st);
  hipLaunchKernelGGL(lv_rows_kernel, dim3(blocks(P + 1)), dim3(kLB), 0, st, g, P, (const int*)order, (const int*)pos, (const int*)done,
                     (const int*)out.row_off, out.in_srcpos, out.in_arc, outcnt_p);
  b = scan_b;
  (void)rocprim::exclusive_scan(prim, b, (const int*)outcnt_p, out.out_off, 0, size_t(P) + 1, rocprim::plus<int>(), st);
  if (P > 0)
    hipLaunchKernelGGL(lv_out_kernel, dim3(blocks(P)), dim3(kLB), 0, st, g, P, (const int*)order, (const int*)pos, (const int*)done,
                       (const int*)out.out_off, out.out_dstpos, out.out_arc);
  if (g.n_accept > 0) hipLaunchKernelGGL(lv_accpos_kernel, dim3(blocks(g.n_accept)), dim3(kLB), 0, st, g, (const int*)pos, out.acc_pos);
  if (P > 0)
    hipLaunchKernelGGL(lv_stats_kernel, dim3(unsigned(L)), dim3(kLB), 0, st, L, (const int*)out.level_off, (const int*)out.row_off,
                       (const int*)out.in_srcpos, info);
  int tails[2] = {0, 0};
  (void)hipMemcpyAsync(info_host, info, 4 * LV_INFO_INTS, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&tails[0], out.row_off + P, 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&tails[1], out.out_off + P, 4, hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
  info_host[LV_L] = P > 0 ? L : 0;
  info_host[LV_N_IN] = tails[0];
  info_host[LV_N_OUT] = tails[1];
}

}  // namespace gtnx
