// rational.hip -- clone / projectInput / projectOutput, concat, closure and union_
// (functions.cpp:66-223) as DEVICE-SIDE structure builders: the output's arc arrays are the
// inputs' arrays copied with node offsets (one launch over all arcs), the epsilon connectors of
// concat / closure are written from the inputs' start / accept lists, and the adjacency lists are
// rebuilt on the device by a stable sort of the arc ids (list order = arc-id order, graph.cpp:62-63).
// Node and arc ids are the reference's: inputs in order, each graph's arcs followed by the
// connectors into it (concat), new start node first (closure).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

namespace gtnx {
namespace {

__global__ void rat_arcs_kernel(const RationalSeg* __restrict__ segs, RationalOut out, int projection) {
  const RationalSeg s = segs[blockIdx.y];
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < s.g.A; a += gridDim.x * blockDim.x) {
    const int o = s.arc_off + a;
    int sn, dn, il, ol;
    if (s.g.kind == KIND_LINEAR) {  // implicit chain: arc a is (a / C) -> (a / C + 1), label a % C
      sn = a / s.g.C;
      dn = sn + 1;
      il = ol = a % s.g.C;
    } else {
      sn = s.g.src[a];
      dn = s.g.dst[a];
      il = s.g.il[a];
      ol = s.g.ol[a];
    }
    out.src[o] = sn + s.node_off;
    out.dst[o] = dn + s.node_off;
    out.il[o] = projection == 2 ? ol : il;  // Projection::OUTPUT: both labels are the output label
    out.ol[o] = projection == 1 ? il : ol;  // Projection::INPUT
    out.w[o] = s.g.w[a];
  }
}

// node flags: keep_start / keep_accept say whether the segment's own flags survive (concat keeps the
// first graph's starts and the last graph's accepts only; closure none)
__global__ void rat_nodes_kernel(const RationalSeg* __restrict__ segs, RationalOut out, int closure) {
  const RationalSeg s = segs[blockIdx.y];
  if (closure && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out.nflags[0] = NF_START | NF_ACCEPT;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < s.g.N; n += gridDim.x * blockDim.x) {
    uint8_t f;
    if (s.g.kind == KIND_LINEAR) f = uint8_t((n == 0 ? NF_START : 0) | (n == s.g.M ? NF_ACCEPT : 0));
    else f = s.g.nflags[n];
    uint8_t o = 0;
    if (s.keep_start && (f & NF_START)) o |= NF_START;
    if (s.keep_accept && (f & NF_ACCEPT)) o |= NF_ACCEPT;
    out.nflags[s.node_off + n] = o;
  }
}

// start / accept node i of a segment (an implicit chain has node 0 / node M)
__device__ __forceinline__ int seg_start(const RationalSeg& s, int i) { return s.g.kind == KIND_LINEAR ? 0 : s.g.start_list[i]; }
__device__ __forceinline__ int seg_accept(const RationalSeg& s, int i) { return s.g.kind == KIND_LINEAR ? s.g.M : s.g.accept_list[i]; }

// epsilon connectors.  concat (functions.cpp:139-149): for every accept p of the previous graph and every start
// q of this one, arc conn_off + p * n_start + q.  closure (functions.cpp:179-186): new start -> every old start,
// then every old accept -> new start.
__global__ void rat_conn_kernel(const RationalSeg* __restrict__ segs, RationalOut out, int closure) {
  const RationalSeg s = segs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (closure) {
    const int ns = s.g.n_start, na = s.g.n_accept;
    if (i >= ns + na) return;
    const int o = s.conn_off + i;
    if (i < ns) {
      out.src[o] = 0;
      out.dst[o] = seg_start(s, i) + s.node_off;
    } else {
      out.src[o] = seg_accept(s, i - ns) + s.node_off;
      out.dst[o] = 0;
    }
    out.il[o] = out.ol[o] = -1;
    out.w[o] = 0.0f;
    return;
  }
  if (blockIdx.y == 0) return;
  const RationalSeg p = segs[blockIdx.y - 1];
  const int ns = s.g.n_start, na = p.g.n_accept;
  if (i >= ns * na) return;
  const int o = s.conn_off + i;
  out.src[o] = seg_accept(p, i / ns) + p.node_off;
  out.dst[o] = seg_start(s, i % ns) + s.node_off;
  out.il[o] = out.ol[o] = -1;
  out.w[o] = 0.0f;
}

// the arc table of the binary graph format (utils.cpp:152-225: {src, dst, ilabel, olabel} per arc, then the weights)
// split into the structure's arrays; node flags copied
__global__ void rat_load_kernel(const gtnx_i4* __restrict__ rows, const float* __restrict__ w, const uint8_t* __restrict__ flags,
                                RationalOut out) {
  const int stride = gridDim.x * blockDim.x;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < out.A; a += stride) {
    const gtnx_i4 r = rows[a];
    out.src[a] = r.x;
    out.dst[a] = r.y;
    out.il[a] = r.z;
    out.ol[a] = r.w;
    out.w[a] = w[a];
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < out.N; n += stride) out.nflags[n] = flags[n];
}

// ---- remove (functions.cpp:253-318)
__global__ void remove_keep_kernel(DGraph g, int ilabel, int olabel, int* keep) {
  const int stride = gridDim.x * blockDim.x;
  // (keep[] starts as the start flags: remove_keep_init_kernel)
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < g.A; a += stride)
    if (!(g.il[a] == ilabel && g.ol[a] == olabel)) keep[g.dst[a]] = 1;
}
__global__ void remove_keep_init_kernel(DGraph g, int* keep) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < g.N) keep[n] = (g.nflags[n] & NF_START) ? 1 : 0;
}
__global__ void remove_roots_kernel(const int* __restrict__ keep, const int* __restrict__ new_id, int N, int* roots) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && keep[n]) roots[new_id[n]] = n;
}
template <bool EMIT>
__global__ void remove_walk_kernel(RemoveArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = a.root0 + r;
  if (r >= a.rows || k >= a.K) return;
  const DGraph& g = a.g;
  GTNX_G int* stamp = a.stamp + size_t(r) * g.N;
  GTNX_G int* queue = a.queue + size_t(r) * g.N;
  const int tag = k + 1;  // unique per kept node: the rows are reused by later batches without clearing
  const int n0 = a.roots[k];
  int head = 0, tail = 0, cnt = 0;
  bool acc = false;
  queue[tail++] = n0;
  stamp[n0] = tag;
  const int base = EMIT ? a.arc_off[k] : 0;
  while (head < tail) {
    const int next = queue[head++];
    acc = acc || (g.nflags[next] & NF_ACCEPT);
    for (int j = g.out_off[next]; j < g.out_off[next + 1]; ++j) {
      const int arc = g.out_list[j];
      const int dn = g.dst[arc], il = g.il[arc], ol = g.ol[arc];
      if (il == a.ilabel && ol == a.olabel) {
        if (stamp[dn] != tag) {
          stamp[dn] = tag;
          queue[tail++] = dn;
        }
      } else {
        if (EMIT) {
          const int o = base + cnt;
          a.out.src[o] = k;
          a.out.dst[o] = a.new_id[dn];
          a.out.il[o] = il;
          a.out.ol[o] = ol;
          a.out.w[o] = 0.0f;  // (remove drops the weights: functions.cpp:308)
        }
        ++cnt;
      }
    }
  }
  if (EMIT) a.out.nflags[k] = uint8_t(((g.nflags[n0] & NF_START) ? NF_START : 0) | (acc ? NF_ACCEPT : 0));
  else a.arc_cnt[k] = cnt;
}

__global__ void rat_iota_kernel(int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
__global__ void rat_count_kernel(const int* __restrict__ key, int n, int* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1);
}
// ordered start / accept lists (node-id order, graph.cpp:33-45): one workgroup, chunked block scan
__global__ __launch_bounds__(1024) void rat_lists_kernel(const uint8_t* __restrict__ nflags, int N, int* start_list,
                                                        int* accept_list) {
  __shared__ int sc[2][1024];
  __shared__ int base[2];
  if (threadIdx.x == 0) base[0] = base[1] = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int n = n0 + threadIdx.x;
    const uint8_t f = n < N ? nflags[n] : 0;
    const int fs = (f & NF_START) ? 1 : 0, fa = (f & NF_ACCEPT) ? 1 : 0;
    sc[0][threadIdx.x] = fs;
    sc[1][threadIdx.x] = fa;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int a = threadIdx.x >= o ? sc[0][threadIdx.x - o] : 0, b = threadIdx.x >= o ? sc[1][threadIdx.x - o] : 0;
      __syncthreads();
      sc[0][threadIdx.x] += a;
      sc[1][threadIdx.x] += b;
      __syncthreads();
    }
    if (fs) start_list[base[0] + sc[0][threadIdx.x] - 1] = n;
    if (fa) accept_list[base[1] + sc[1][threadIdx.x] - 1] = n;
    __syncthreads();
    if (threadIdx.x == 1023) {
      base[0] += sc[0][1023];
      base[1] += sc[1][1023];
    }
    __syncthreads();
  }
}

int bits_for(int n) {
  int b = 1;
  while ((1ll << b) < n) ++b;
  return b;
}

}  // namespace

void launch_remove_keep(const DGraph& g, int ilabel, int olabel, int* keep, hipStream_t st) {
  if (g.N > 0) hipLaunchKernelGGL(remove_keep_init_kernel, dim3((g.N + 255) / 256), dim3(256), 0, st, g, keep);
  if (g.A > 0)
    hipLaunchKernelGGL(remove_keep_kernel, dim3(std::min((g.A + 255) / 256, 4096)), dim3(256), 0, st, g, ilabel, olabel, keep);
}
void launch_remove_roots(const int* keep, const int* new_id, int N, int* roots, hipStream_t st) {
  if (N > 0) hipLaunchKernelGGL(remove_roots_kernel, dim3((N + 255) / 256), dim3(256), 0, st, keep, new_id, N, roots);
}
void launch_remove_walk(const RemoveArgs& a, bool emit, hipStream_t st) {
  const int n = std::min(a.rows, a.K - a.root0);
  if (n <= 0) return;
  if (emit) hipLaunchKernelGGL(remove_walk_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, st, a);
  else hipLaunchKernelGGL(remove_walk_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, st, a);
}
size_t scan_temp_bytes(int n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const int*)nullptr, (int*)nullptr, 0, size_t(n > 0 ? n : 1), rocprim::plus<int>());
  return b + 256;
}
void launch_exclusive_scan(const int* in, int* out, int n, void* temp, size_t temp_bytes, hipStream_t st) {
  if (n > 0) (void)rocprim::exclusive_scan(temp, temp_bytes, in, out, 0, size_t(n), rocprim::plus<int>(), st);
}

void launch_rational_load(const void* rows, const float* w, const uint8_t* flags, const RationalOut& out, void* temp,
                          hipStream_t st) {
  const int work = std::max(out.A, out.N);
  if (work > 0)
    hipLaunchKernelGGL(rat_load_kernel, dim3(std::min((work + 255) / 256, 4096)), dim3(256), 0, st,
                       static_cast<const gtnx_i4*>(rows), w, flags, out);
  launch_rational_adjacency(out, temp, st);
}

size_t rational_csr_temp_bytes(int N, int A) {
  size_t sort_b = 0, scan_b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, size_t(A > 0 ? A : 1), 0, 32);
  (void)rocprim::exclusive_scan(nullptr, scan_b, (const int*)nullptr, (int*)nullptr, 0, size_t(N + 1), rocprim::plus<int>());
  // keys out + iota + counts, then the primitives' own scratch
  return 256 + 4 * size_t(A > 0 ? A : 1) * 2 + 4 * size_t(N + 2) + std::max(sort_b, scan_b) + 256;
}

void launch_rational_build(const RationalSeg* d_segs, int nseg, int max_A, int max_N, int max_conn, const RationalOut& out,
                           int projection, int closure, void* temp, hipStream_t st) {
  const int N = out.N, A = out.A;
  if (max_A > 0)
    hipLaunchKernelGGL(rat_arcs_kernel, dim3(std::min((max_A + 255) / 256, 4096), nseg), dim3(256), 0, st, d_segs, out, projection);
  hipLaunchKernelGGL(rat_nodes_kernel, dim3(std::max(1, std::min((max_N + 255) / 256, 4096)), nseg), dim3(256), 0, st, d_segs,
                     out, closure);
  if (max_conn > 0)
    hipLaunchKernelGGL(rat_conn_kernel, dim3((max_conn + 255) / 256, nseg), dim3(256), 0, st, d_segs, out, closure);
  launch_rational_adjacency(out, temp, st);
}

// adjacency lists + ordered start / accept lists of a structure whose arc arrays and node flags are in place
void launch_rational_adjacency(const RationalOut& out, void* temp, hipStream_t st) {
  const int N = out.N, A = out.A;
  // ---- adjacency: lists in arc-id order = stable sort of the arc ids by source / destination
  char* t = static_cast<char*>(temp);
  unsigned* keys_out = reinterpret_cast<unsigned*>(t);
  int* iota = reinterpret_cast<int*>(t + 4 * size_t(A > 0 ? A : 1));
  int* cnt = iota + (A > 0 ? A : 1);
  void* scratch = reinterpret_cast<char*>(cnt + N + 2);
  size_t sort_b = 0, scan_b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                  (int*)nullptr, size_t(A > 0 ? A : 1), 0, 32);
  (void)rocprim::exclusive_scan(nullptr, scan_b, (const int*)nullptr, (int*)nullptr, 0, size_t(N + 1), rocprim::plus<int>());
  const int nb = bits_for(N > 1 ? N : 2);
  for (int side = 0; side < 2; ++side) {
    const int* key = side ? out.dst : out.src;
    int* off = side ? out.in_off : out.out_off;
    int* list = side ? out.in_list : out.out_list;
    (void)hipMemsetAsync(cnt, 0, 4 * size_t(N + 1), st);
    if (A > 0) {
      hipLaunchKernelGGL(rat_count_kernel, dim3((A + 255) / 256), dim3(256), 0, st, key, A, cnt);
      hipLaunchKernelGGL(rat_iota_kernel, dim3((A + 255) / 256), dim3(256), 0, st, iota, A);
      size_t b = sort_b;
      (void)rocprim::radix_sort_pairs(scratch, b, reinterpret_cast<const unsigned*>(key), keys_out, (const int*)iota, list,
                                      size_t(A), 0, nb, st);
    }
    size_t b2 = scan_b;
    (void)rocprim::exclusive_scan(scratch, b2, (const int*)cnt, off, 0, size_t(N + 1), rocprim::plus<int>(), st);
  }
  if (N > 0) hipLaunchKernelGGL(rat_lists_kernel, dim3(1), dim3(1024), 0, st, out.nflags, N, out.start_list, out.accept_list);
}

} // namespace gtnx
