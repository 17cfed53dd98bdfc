// graph.cpp -- host graph model, HBM residency and level scheduling.
#include "graph.h"

#include <new>

#include <unordered_set>

#include "gtn/parallel.h"  // header-only worker pool (no engine dependency)

#include <algorithm>
#include <cstring>
#include <list>
#include <map>
#include <queue>

namespace gtnx {

GradState::~GradState() {
  for (auto& in : inputs)
    if (in.g) in.g->n_consumers--;
}

// ======================================================================
// Structure
// ======================================================================
void Structure::touch() {
  dev_valid = false;
  dev_mem.reset();
  rec_mem.reset();
  sched.reset();
  csr_valid = false;
  sort_pending = 0;
  max_deg = -1;
  band[0].reset();
  band[1].reset();
  dense[0].reset();
  dense[1].reset();
  ctc_labels.reset();
  leaf_batch.reset();
  ctc_checked = false;
  ilabel_sorted = olabel_sorted = false;  // graph.cpp:42-43, 64-65
}

// Is this exactly ctcGraph(labels) of benchmarks/ctc.cpp:40-58 (same nodes, same arcs in the same order,
// blank = the label of node 0's self-loop)?  O(A) integer compares on the host arrays.
void detect_ctc_shape(Structure& s) {
  if (s.ctc_checked) return;
  s.ctc_checked = true;
  s.ctc_labels.reset();
  if (s.kind != KIND_EXPLICIT || !s.host_valid || s.N < 1 || (s.N & 1) == 0 || s.A < s.N) return;
  const int L = int(s.N), U = (L - 1) / 2;
  if (s.il[0] < 0 || s.src[0] != 0 || s.dst[0] != 0) return;
  const int blank = s.il[0];
  auto lab = std::make_shared<std::vector<int>>(size_t(U));
  size_t a = 0;
  const size_t A = size_t(s.A);
  for (int l = 0; l < L; ++l) {
    if (a >= A || s.src[a] != l || s.dst[a] != l) return;
    const int label = s.il[a];
    if (label < 0 || s.ol[a] != label) return;
    if (l % 2) (*lab)[size_t(l / 2)] = label;
    else if (label != blank) return;
    const uint8_t want = uint8_t((l == 0 ? NF_START : 0) | ((l == L - 1 || l == L - 2) ? NF_ACCEPT : 0));
    if (s.nflags[size_t(l)] != want) return;
    ++a;
    if (l > 0) {
      if (a >= A || s.src[a] != l - 1 || s.dst[a] != l || s.il[a] != label || s.ol[a] != label) return;
      ++a;
    }
    if (l % 2 && l > 1 && label != (*lab)[size_t(l / 2) - 1]) {
      if (a >= A || s.src[a] != l - 2 || s.dst[a] != l || s.il[a] != label || s.ol[a] != label) return;
      ++a;
    }
  }
  if (a != A) return;
  s.ctc_labels = std::move(lab);
  s.ctc_blank = blank;
}

bool Weights::is_all_zero() {
  if (zero) return true;
  if (!host_valid || host_escaped) return false;
  if (zero_version != version) {
    all_zero = true;
    for (int64_t i = 0; i < n && all_zero; ++i) all_zero = host[size_t(i)] == 0.0f;
    zero_version = version;
  }
  return all_zero;
}

// Band records of a host-built structure (kernels.h: BandNode).  O(A); cached until touch().
std::shared_ptr<BandInfo> band_info(Structure& s, bool use_ilabel) {
  std::shared_ptr<BandInfo>& slot = s.band[use_ilabel ? 0 : 1];
  if (slot) return slot;
  auto b = std::make_shared<BandInfo>();
  slot = b;
  if (s.kind != KIND_EXPLICIT || !s.host_valid || s.lazy || s.N < 1 || s.N > band_max_nodes()) return b;
  const size_t N = size_t(s.N), A = size_t(s.A);
  b->nodes.assign(N, BandNode{-1, {-1, -1, -1}});
  const std::vector<int>& lab = use_ilabel ? s.il : s.ol;
  for (size_t a = 0; a < A; ++a) {
    const int k = s.dst[a] - s.src[a], l = lab[a];
    if (k < 0 || k > 2 || l < 0) return b;
    BandNode& nd = b->nodes[size_t(s.dst[a])];
    if (nd.aid[k] >= 0 || (nd.lab >= 0 && nd.lab != l)) return b;
    nd.lab = l;
    nd.aid[k] = int(a);
    b->max_label = std::max(b->max_label, l);
  }
  b->ok = true;
  b->unit_shape = true;
  for (size_t m = 0; m < N && b->unit_shape; ++m)
    b->unit_shape = b->nodes[m].aid[0] >= 0 && (m == 0 || b->nodes[m].aid[1] >= 0);
  // nodes by label (the backward sweep gathers posteriors by label); the label most nodes
  // share gets a wave of its own
  std::vector<uint64_t> ln;  // (label, node) keys
  ln.reserve(N);
  for (size_t m = 0; m < N; ++m)
    if (b->nodes[m].lab >= 0) ln.push_back((uint64_t(uint32_t(b->nodes[m].lab)) << 32) | uint64_t(m));
  std::sort(ln.begin(), ln.end());
  b->snode.resize(ln.size());
  b->slab.resize(ln.size());
  size_t best = 0;
  for (size_t i = 0; i < ln.size();) {
    size_t j = i;
    while (j < ln.size() && (ln[j] >> 32) == (ln[i] >> 32)) ++j;
    if (j - i > best) {
      best = j - i;
      b->hot = int(ln[i] >> 32);
    }
    i = j;
  }
  for (size_t i = 0; i < ln.size(); ++i) {
    b->slab[i] = int(ln[i] >> 32);
    b->snode[i] = int(ln[i] & 0xffffffffu);
  }
  if (best < 8) b->hot = -1;
  return b;
}

// one staging copy + one H2D for the band records (and node flags) of a batch
void ensure_band_device_batch(const std::vector<BandInfo*>& bs, const std::vector<Structure*>& ss) {
  Runtime& rt = Runtime::get();
  std::vector<size_t> todo;
  std::unordered_set<BandInfo*> seen;
  for (size_t i = 0; i < bs.size(); ++i)
    if (!bs[i]->dev && seen.insert(bs[i]).second) todo.push_back(i);
  if (todo.empty()) return;
  size_t total = 0;
  std::vector<size_t> on(todo.size()), of(todo.size());
  for (size_t q = 0; q < todo.size(); ++q) {
    const size_t N = bs[todo[q]]->nodes.size();
    on[q] = total;
    of[q] = total + sizeof(BandNode) * N;
    total = align_up(of[q] + N, 64) + 8 * bs[todo[q]]->snode.size();
    total = align_up(total, 64);
  }
  PinnedMemP pin = rt.alloc_pinned(total);
  DevMemP dev = rt.alloc(total);
  auto body = [&](size_t q) {
    BandInfo* b = bs[todo[q]];
    const size_t N = b->nodes.size();
    std::memcpy(pin->as<char>(on[q]), b->nodes.data(), sizeof(BandNode) * N);
    std::memcpy(pin->as<char>(of[q]), ss[todo[q]]->nflags.data(), N);
    b->dev_mem = dev;
    b->dev = dev->as<BandNode>(on[q]);
    b->dev_flags = dev->as<uint8_t>(of[q]);
    const size_t os = align_up(of[q] + N, 64), nl = b->snode.size();
    std::memcpy(pin->as<char>(os), b->snode.data(), 4 * nl);
    std::memcpy(pin->as<char>(os + 4 * nl), b->slab.data(), 4 * nl);
    b->dev_snode = dev->as<int>(os);
    b->dev_slab = dev->as<int>(os + 4 * nl);
  };
  for (size_t q = 0; q < todo.size(); ++q) body(q);
  rt.h2d_pinned(dev->ptr, pin->ptr, total);
}

int Structure::max_degree() {
  if (max_deg >= 0) return max_deg;
  if (kind == KIND_LINEAR) return max_deg = (M > 0 ? C : 0);
  ensure_host();
  ensure_csr();
  int d = 0;
  for (int64_t n = 0; n < N; ++n)
    d = std::max(d, std::max(out_off[n + 1] - out_off[n], in_off[n + 1] - in_off[n]));
  return max_deg = d;
}

void Structure::materialize() {
  if (kind != KIND_LINEAR) return;
  // creations.cpp:20-33: node m -> m+1, arc id m*N+n, label n
  src.resize(A);
  dst.resize(A);
  il.resize(A);
  ol.resize(A);
  for (int64_t a = 0; a < A; ++a) {
    src[a] = int(a / C);
    dst[a] = int(a / C) + 1;
    il[a] = ol[a] = int(a % C);
  }
  nflags.assign(N, 0);
  nflags[0] |= NF_START;
  start.assign(1, 0);
  accept.clear();
  if (M > 0) {
    nflags[M] |= NF_ACCEPT;
    accept.assign(1, M);
  }
  kind = KIND_EXPLICIT;
  host_valid = true;
  csr_valid = false;
  dev_valid = false;
  dev_mem.reset();
  sched.reset();
}

// ---------------------------------------------------------------- deferred sizes
void apply_compose_sizes(Structure& s, Weights* w, const ComposeOut& co, int n_start, int n_accept) {
  s.N = co.N;
  s.A = co.A;
  DGraph& v = s.dview;
  v.N = co.N;
  v.A = co.A;
  v.n_start = n_start;
  v.n_accept = n_accept;
  if (w) w->n = co.A;
  if (s.partial) {
    s.partial->args.N = co.N;
    s.partial->args.A = co.A;
  }
  if (s.sched) {
    Schedule& sc = *s.sched;
    sc.n_in = co.A;
    sc.n_out = co.A;
    sc.max_level_width = co.max_width;
    sc.max_level_arcs = co.max_level_arcs;
    sc.max_reach = 2 * co.max_width;  // in-arcs come from the previous level only
    sc.view.P = co.N;
    sc.view.L = co.L;
    sc.view.n_accept = n_accept;
  }
}

namespace {
std::mutex g_def_mu;
std::vector<std::shared_ptr<DeferredSizes>> g_deferred;
}
void deferred_register(const std::shared_ptr<DeferredSizes>& d) {
  std::lock_guard<std::mutex> lk(g_def_mu);
  g_deferred.push_back(d);
}
void deferred_limit(size_t keep) {
  for (;;) {
    std::shared_ptr<DeferredSizes> d;
    {
      std::lock_guard<std::mutex> lk(g_def_mu);
      // drop what is resolved already
      while (!g_deferred.empty() && g_deferred.front()->done) g_deferred.erase(g_deferred.begin());
      if (g_deferred.size() <= keep) return;
      d = g_deferred.front();
      g_deferred.erase(g_deferred.begin());
    }
    d->resolve();
  }
}
void deferred_resolve_all() { deferred_limit(0); }

DeferredSizes::~DeferredSizes() {
  if (ev) (void)hipEventDestroy(ev);
}

void DeferredSizes::resolve() {
  // (two threads may get here for the same record -- one through deferred_limit, one through a member's
  //  resolve_sizes: the second has to wait until the first has applied the sizes, not just see the flag)
  std::lock_guard<std::mutex> lk(mu);
  if (done.load(std::memory_order_relaxed)) return;
  done.store(true, std::memory_order_relaxed);
  HIP_CHECK(hipEventSynchronize(ev));
  const ComposeOut* outs = reinterpret_cast<const ComposeOut*>(host->as<char>(hdr_out));
  const int* cnts = reinterpret_cast<const int*>(host->as<char>(hdr_cnt));
  for (size_t i = 0; i < members.size(); ++i) {
    const ComposeOut& co = outs[i];
    if (co.overflow || !co.layered || !co.csr_built)
      throw_runtime("[gtn::compose] internal: a deferred-size composition left the proven fast path (overflow " +
                    std::to_string(co.overflow) + ", layered " + std::to_string(co.layered) + ", csr " +
                    std::to_string(co.csr_built) + ", N " + std::to_string(co.N) + ", A " + std::to_string(co.A) + ")");
    std::shared_ptr<Structure> s = members[i].s.lock();
    if (!s) continue;
    std::shared_ptr<Weights> w = members[i].w.lock();
    apply_compose_sizes(*s, w.get(), co, cnts[2 * i], cnts[2 * i + 1]);
    s->deferred.reset();
  }
  for (auto& gw : grads)
    if (auto w = gw.w.lock()) w->n = outs[gw.member].A;
  if (!prof.empty() && Runtime::initialized()) {
    Runtime& rt = Runtime::get();
    for (auto& p : prof)
      rt.prof_add_bytes(p.name, p.per_arc * double(outs[p.member].A) + p.per_node * double(outs[p.member].N));
  }
  prof.clear();
}

void Structure::resolve_sizes() {
  if (!deferred) return;
  std::shared_ptr<DeferredSizes> d = deferred;  // resolve() clears the member
  d->resolve();
}

void Structure::ensure_full() {
  resolve_sizes();
  if (!partial) return;
  Runtime& rt = Runtime::get();
  std::shared_ptr<PartialInfo> p = partial;
  partial.reset();
  ComposeFillArgs a = p->args;
  DevMemP cursor = rt.alloc(sizeof(int) * size_t(a.N > 0 ? a.N : 1));
  if (a.N > 0) rt.d2d(cursor->ptr, dview.in_off, sizeof(int) * size_t(a.N));
  a.in_cursor = cursor->as<int>();
  launch_compose_fill(a, rt.stream());
  // `cursor` goes back to the pool here; the pool is stream-ordered, so a later
  // allocation cannot touch it before the kernels above have run
}

void Structure::ensure_host() {
  if (kind == KIND_LINEAR || host_valid) return;
  ensure_full();
  // device-built (composition result): pull the SoA arrays once
  Runtime& rt = Runtime::get();
  src.resize(A);
  dst.resize(A);
  il.resize(A);
  ol.resize(A);
  nflags.resize(N);
  size_t ab = sizeof(int) * size_t(A);
  if (A) {
    HIP_CHECK(hipMemcpyAsync(src.data(), dview.src, ab, hipMemcpyDeviceToHost, rt.stream()));
    HIP_CHECK(hipMemcpyAsync(dst.data(), dview.dst, ab, hipMemcpyDeviceToHost, rt.stream()));
    HIP_CHECK(hipMemcpyAsync(il.data(), dview.il, ab, hipMemcpyDeviceToHost, rt.stream()));
    HIP_CHECK(hipMemcpyAsync(ol.data(), dview.ol, ab, hipMemcpyDeviceToHost, rt.stream()));
  }
  if (N) HIP_CHECK(hipMemcpyAsync(nflags.data(), dview.nflags, size_t(N), hipMemcpyDeviceToHost, rt.stream()));
  rt.sync();
  start.clear();
  accept.clear();
  for (int n = 0; n < N; ++n) {
    if (nflags[n] & NF_START) start.push_back(n);
    if (nflags[n] & NF_ACCEPT) accept.push_back(n);
  }
  host_valid = true;
  csr_valid = false;
}

void Structure::ensure_csr() {
  if (kind == KIND_LINEAR) materialize();
  ensure_host();
  if (csr_valid) return;
  // lists in arc-id order == the push_back order of graph.cpp:62-63
  in_off.assign(N + 1, 0);
  out_off.assign(N + 1, 0);
  for (int64_t a = 0; a < A; ++a) {
    out_off[src[a] + 1]++;
    in_off[dst[a] + 1]++;
  }
  for (int64_t n = 0; n < N; ++n) {
    out_off[n + 1] += out_off[n];
    in_off[n + 1] += in_off[n];
  }
  in_list.resize(A);
  out_list.resize(A);
  std::vector<int> ci(in_off.begin(), in_off.end() - 1), co(out_off.begin(), out_off.end() - 1);
  for (int64_t a = 0; a < A; ++a) {
    out_list[co[src[a]]++] = int(a);
    in_list[ci[dst[a]]++] = int(a);
  }
  csr_valid = true;
  if (sort_pending) {  // graph.cpp:162-177, asked for before anything needed the lists
    const std::vector<int>& key = sort_pending == 2 ? ol : il;
    auto cmp = [&key](int a, int b) { return key[a] < key[b]; };
    for (int64_t n = 0; n < N; ++n) {
      std::sort(in_list.begin() + in_off[n], in_list.begin() + in_off[n + 1], cmp);
      std::sort(out_list.begin() + out_off[n], out_list.begin() + out_off[n + 1], cmp);
    }
    sort_pending = 0;
  }
}

int Structure::num_in(int n) {
  if (kind == KIND_LINEAR) return n == 0 ? 0 : C;
  ensure_csr();
  return in_off[n + 1] - in_off[n];
}
int Structure::num_out(int n) {
  if (kind == KIND_LINEAR) return n == M ? 0 : C;
  ensure_csr();
  return out_off[n + 1] - out_off[n];
}

// ======================================================================
// Weights
// ======================================================================
void PendingCopy::settle() {
  if (done.load(std::memory_order_acquire)) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.load(std::memory_order_relaxed)) return;
  if (!segs.empty()) {
    Runtime& rt = Runtime::of(device);  // (whoever asks, from whichever thread: the copy runs on the owner's stream)
    rt.activate();
    const size_t bytes = sizeof(CopySeg) * segs.size();
    DevMemP d = rt.alloc(bytes);
    PinnedMemP p = rt.alloc_pinned(bytes);
    std::memcpy(p->ptr, segs.data(), bytes);
    rt.h2d_pinned(d->ptr, p->ptr, bytes);
    launch_copy_segments(d->as<CopySeg>(), int(segs.size()), max_bytes, rt.stream());
  }
  done.store(true, std::memory_order_release);
}
const void* PendingCopy::src_of(const void* dst) const {
  auto it = std::lower_bound(segs.begin(), segs.end(), dst, [](const CopySeg& a, const void* d) { return a.dst < d; });
  return it != segs.end() && it->dst == dst ? it->src : nullptr;
}

bool Weights::settle_staged() {
  if (!staged || !staged->on_device || !staged->blk) return false;
  float* base = staged->blk->base.load(std::memory_order_acquire);
  if (!base) return false;
  if (staged->blk->pend) staged->blk->pend->settle();  // whoever looks at the values finds them there
  dev_mem = staged->blk->mem;
  dev = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + staged->off);
  dev_valid = true;
  host_valid = false;
  staged.reset();
  return true;
}

void Weights::ensure_host() {
  if (host_valid) return;
  settle_staged();
  if (zero) {
    host.assign(size_t(n), 0.0f);
    zero = false;
    host_valid = true;
    return;
  }
  if (staged && !staged->on_device) {  // handed over inside a parallelMap region, not uploaded yet
    host.assign(staged->src, staged->src + n);
    staged.reset();
    host_valid = true;
    return;
  }
  if (staged) {  // the caller's device buffer: take the copy now
    Runtime& rt = Runtime::get();
    dev_mem = rt.alloc(sizeof(float) * size_t(n ? n : 1));
    dev = dev_mem->as<float>();
    rt.d2d(dev, staged->src, sizeof(float) * size_t(n));
    staged.reset();
    dev_valid = true;
  }
  host.resize(n);
  float mv;
  if (n == 1 && mirror.ptr && mirror_version == version && dev_valid && Runtime::get().mirror_read(mirror, &mv)) {
    host[0] = mv;
    mirror = {};
    host_valid = true;
    return;
  }
  if (n) Runtime::get().d2h_sync(host.data(), dev, sizeof(float) * size_t(n));
  host_valid = true;
}

// ======================================================================
// Graph
// ======================================================================
namespace {
// (graph.h: GraphSlabScope)  One raw buffer, handed out front to back by the one thread whose scope it is --
// as the ALLOCATOR of the results' shared pieces (std::allocate_shared: object and reference count side by side in
// the buffer), so that every piece still dies on its own when its last reference goes, and only the MEMORY is
// pooled: the buffer is given back when the last piece carved out of it has been destroyed.  (A slab whose pieces
// all lived as long as the slab did -- aliases of one owner -- leaked: a result's gradient state holds its op
// record, the op record may hold another result's structure, and nothing could break that ring.)
struct SlabBuf {
  std::atomic<long> live{1};  // the scope's own reference + one per allocation
  char* base;
  size_t cap, used = 0;
};
constexpr size_t kSlabPerGraph =
    ((sizeof(Structure) + 63) & ~size_t(63)) + ((sizeof(Weights) + 63) & ~size_t(63)) + ((sizeof(GradState) + 63) & ~size_t(63)) + 192;
// The raw buffers go round a small per-thread cache: a slab of 256 graphs is 330 KB, which the allocator maps and
// unmaps on every use (above its mmap threshold) -- 80 fresh pages to fault in per slab
// (tools/nullhip/small_step c2b).  A buffer retires to the cache of whichever thread destroys the last piece.
struct SlabCache {
  struct Buf {
    char* p;
    size_t cap;
  };
  std::vector<Buf> bufs;
  size_t bytes = 0;  // held: at most 16 MB per thread (the garbage of ~10 C2 batches comes back at once, runtime.cpp kDeferFull)
  ~SlabCache() {
    for (auto& b : bufs) ::operator delete(b.p, std::align_val_t(64));
  }
};
// (null once the thread's destructors have run: buffers that retire later -- the thread's own deferred garbage is
//  taken apart by a thread-exit destructor too, in unspecified order -- are freed directly)
SlabCache* slab_cache() {
  // (the pointer and the flag are trivially destructible thread-locals: they may be read after `h` is gone, which a
  //  member of `h` may not)
  static thread_local SlabCache* cache = nullptr;
  static thread_local bool gone = false;
  struct Holder {
    Holder() { cache = new SlabCache(); }
    ~Holder() {
      SlabCache* dead = cache;
      cache = nullptr;
      gone = true;
      delete dead;
    }
  };
  if (!cache && !gone) {
    thread_local Holder h;
    (void)h;
  }
  return cache;
}
std::atomic<long> g_slab_hit{0}, g_slab_miss{0}, g_slab_drop{0};
struct SlabStats {
  ~SlabStats() {
    if (std::getenv("GTNX_SLAB_STATS"))
      std::fprintf(stderr, "[gtnx] graph slabs: %ld from the cache, %ld fresh, %ld freed past a full cache\n", g_slab_hit.load(),
                   g_slab_miss.load(), g_slab_drop.load());
  }
} g_slab_stats;
SlabBuf* slab_open(size_t bytes) {
  SlabBuf* b = new SlabBuf();
  if (SlabCache* c = slab_cache()) {
    for (size_t i = 0; i < c->bufs.size(); ++i)
      if (c->bufs[i].cap >= bytes && c->bufs[i].cap <= 2 * bytes) {
        b->base = c->bufs[i].p;
        b->cap = c->bufs[i].cap;
        c->bytes -= b->cap;
        c->bufs[i] = c->bufs.back();
        c->bufs.pop_back();
        g_slab_hit.fetch_add(1, std::memory_order_relaxed);
        return b;
      }
  }
  g_slab_miss.fetch_add(1, std::memory_order_relaxed);
  if (std::getenv("GTNX_SLAB_DEBUG")) {
    SlabCache* c = slab_cache();
    std::fprintf(stderr, "slab miss: want %zu; cache %p holds", bytes, (void*)c);
    if (c) for (auto& x : c->bufs) std::fprintf(stderr, " %zu", x.cap);
    std::fprintf(stderr, "\n");
  }
  b->base = static_cast<char*>(::operator new(bytes, std::align_val_t(64)));
  b->cap = bytes;
  return b;
}
void slab_release(SlabBuf* b) {
  if (b->live.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  SlabCache* c = slab_cache();
  if (c && c->bufs.size() < 64 && b->cap <= (size_t(4) << 20) && c->bytes + b->cap <= (size_t(16) << 20)) {
    c->bufs.push_back({b->base, b->cap});
    c->bytes += b->cap;
  } else {
    if (std::getenv("GTNX_SLAB_DEBUG")) std::fprintf(stderr, "slab drop: cache %p size %zu cap %zu\n", (void*)c, c ? c->bufs.size() : 0, b->cap);
    g_slab_drop.fetch_add(1, std::memory_order_relaxed);
    ::operator delete(b->base, std::align_val_t(64));
  }
  delete b;
}
template <class T>
struct SlabAlloc {
  using value_type = T;
  SlabBuf* buf;
  explicit SlabAlloc(SlabBuf* b) : buf(b) {}
  template <class U>
  SlabAlloc(const SlabAlloc<U>& o) : buf(o.buf) {}
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T), off = (buf->used + 63) & ~size_t(63);
    buf->live.fetch_add(1, std::memory_order_relaxed);  // (also for what does not fit: deallocate looks at the buffer's range)
    if (off + bytes <= buf->cap) {
      buf->used = off + bytes;
      return reinterpret_cast<T*>(buf->base + off);
    }
    return static_cast<T*>(::operator new(bytes));
  }
  void deallocate(T* p, size_t) {
    const char* c = reinterpret_cast<const char*>(p);
    if (!(c >= buf->base && c < buf->base + buf->cap)) ::operator delete(p);
    slab_release(buf);
  }
  template <class U>
  bool operator==(const SlabAlloc<U>& o) const { return buf == o.buf; }
  template <class U>
  bool operator!=(const SlabAlloc<U>& o) const { return buf != o.buf; }
};
thread_local SlabBuf* t_slab = nullptr;
}  // namespace

GraphSlabScope::GraphSlabScope(size_t n) {
  static const bool off = std::getenv("GTNX_NO_GRAPH_SLAB") != nullptr;
  if (off || n < 8) return;
  prev_ = t_slab;  // (scopes nest: every scope has a buffer of its own)
  t_slab = slab_open(n * kSlabPerGraph);
  active_ = true;
}
GraphSlabScope::~GraphSlabScope() {
  if (!active_) return;
  slab_release(t_slab);
  t_slab = static_cast<SlabBuf*>(prev_);
}

Graph::Graph(bool calc_grad)
    : s(std::make_shared<Structure>()), w(std::make_shared<Weights>()), g(std::make_shared<GradState>()) {
  s->home = Runtime::home();
  s->device = Runtime::current_device();
  g->calc_grad = calc_grad;
}

// fresh pieces for an op RESULT: out of the calling thread's slab scope when there is one (graph.h)
Graph Graph::make_result(bool calc_grad) {
  SlabBuf* sl = t_slab;
  if (!sl || sl->used + kSlabPerGraph > sl->cap) return Graph(calc_grad);
  Graph out{Graph::Empty{}};
  out.s = std::allocate_shared<Structure>(SlabAlloc<Structure>(sl));
  out.w = std::allocate_shared<Weights>(SlabAlloc<Weights>(sl));
  out.g = std::allocate_shared<GradState>(SlabAlloc<GradState>(sl));
  out.s->home = Runtime::home();
  out.s->device = Runtime::current_device();
  out.g->calc_grad = calc_grad;
  return out;
}

Graph::Graph(bool calc_grad, std::shared_ptr<Structure> shared)
    : s(std::move(shared)), w(std::make_shared<Weights>()), g(std::make_shared<GradState>()) {
  g->calc_grad = calc_grad;
}


int Graph::add_node(bool start, bool accept) {
  s->materialize();
  s->ensure_host();
  int idx = int(s->N);
  s->nflags.push_back(uint8_t((start ? NF_START : 0) | (accept ? NF_ACCEPT : 0)));
  if (start) s->start.push_back(idx);
  if (accept) s->accept.push_back(idx);
  s->N++;
  s->touch();
  return idx;
}

int Graph::add_arc(int src, int dst, int il, int ol, float wt) {
  s->materialize();
  s->ensure_host();
  if (src < 0 || src >= s->N || dst < 0 || dst >= s->N) throw_range("[Graph::addArc] node index out of range");
  if (il < GTNX_EPSILON || ol < GTNX_EPSILON) throw_invalid("[Graph::addArc] labels must be >= epsilon");  // graph.cpp:57
  w->ensure_host();
  int idx = int(s->A);
  s->src.push_back(src);
  s->dst.push_back(dst);
  s->il.push_back(il);
  s->ol.push_back(ol);
  w->host.push_back(wt);
  w->n = int64_t(w->host.size());
  w->dev_valid = false;
  w->version++;
  s->A++;
  s->touch();
  return idx;
}

void Graph::add_nodes(int n, const uint8_t* start, const uint8_t* accept) {
  if (n <= 0) return;
  s->materialize();
  s->ensure_host();
  s->nflags.reserve(s->nflags.size() + size_t(n));
  for (int i = 0; i < n; ++i) {
    const bool st = start && start[i], ac = accept && accept[i];
    const int idx = int(s->N) + i;
    s->nflags.push_back(uint8_t((st ? NF_START : 0) | (ac ? NF_ACCEPT : 0)));
    if (st) s->start.push_back(idx);
    if (ac) s->accept.push_back(idx);
  }
  s->N += n;
  s->touch();
}

void Graph::add_arcs(int n, const int* src, const int* dst, const int* il, const int* ol, const float* wt) {
  if (n <= 0) return;
  s->materialize();
  s->ensure_host();
  for (int i = 0; i < n; ++i) {
    if (src[i] < 0 || src[i] >= s->N || dst[i] < 0 || dst[i] >= s->N) throw_range("[Graph::addArc] node index out of range");
    if (il[i] < GTNX_EPSILON || ol[i] < GTNX_EPSILON) throw_invalid("[Graph::addArc] labels must be >= epsilon");  // graph.cpp:57
  }
  w->ensure_host();
  s->src.insert(s->src.end(), src, src + n);
  s->dst.insert(s->dst.end(), dst, dst + n);
  s->il.insert(s->il.end(), il, il + n);
  s->ol.insert(s->ol.end(), ol, ol + n);
  if (wt) w->host.insert(w->host.end(), wt, wt + n);
  else w->host.resize(w->host.size() + size_t(n), 0.0f);
  w->n = int64_t(w->host.size());
  w->dev_valid = false;
  w->version++;
  s->A += n;
  s->touch();
}

int64_t Graph::num_start() {
  if (s->kind == KIND_LINEAR) return 1;
  s->ensure_host();
  return int64_t(s->start.size());
}
int64_t Graph::num_accept() {
  if (s->kind == KIND_LINEAR) return s->M > 0 ? 1 : 0;
  s->ensure_host();
  return int64_t(s->accept.size());
}

float Graph::item() {
  if (s->A != 1)  // graph.cpp:70-73
    throw_invalid("[Graph::item] Cannot convert Graph with more than 1 arc to a scalar.");
  w->ensure_host();
  return w->host[0];
}

void Graph::arc_sort(bool olabel) {
  // graph.cpp:162-177
  if ((olabel && s->olabel_sorted) || (!olabel && s->ilabel_sorted)) return;
  if (s->kind == KIND_EXPLICIT && s->host_valid && !s->csr_valid) {
    // nobody has looked at the per-node lists yet: sort them when they are first built.  A
    // criterion that only takes forwardScore of a symbolic composition never needs them.
    s->sort_pending = olabel ? 2 : 1;
    s->dev_valid = false;
    s->dev_mem.reset();
    s->sched.reset();
    s->olabel_sorted = olabel;
    s->ilabel_sorted = !olabel;
    // A small graph that has just been sorted is about to be composed.  Host code builds such
    // graphs on many threads (parallelMap): take the band records now, on this thread, instead of
    // on the one thread that later calls the batched compose.
    if (s->N <= band_max_nodes()) {
      detect_ctc_shape(*s);
      if (!s->ctc_labels) {  // (a CTC target acceptor's records are built on the device from its labels: batch.cpp)
        band_info(*s, true);
        if (s->il == s->ol) s->band[1] = s->band[0];  // an acceptor matches the same labels either way round
      }
      (void)w->is_all_zero();
    }
    return;
  }
  s->ensure_csr();
  const std::vector<int>& key = olabel ? s->ol : s->il;
  auto cmp = [&key](int a, int b) { return key[a] < key[b]; };
  for (int64_t n = 0; n < s->N; ++n) {
    std::sort(s->in_list.begin() + s->in_off[n], s->in_list.begin() + s->in_off[n + 1], cmp);
    std::sort(s->out_list.begin() + s->out_off[n], s->out_list.begin() + s->out_off[n + 1], cmp);
  }
  s->dev_valid = false;
  s->dev_mem.reset();
  s->sched.reset();
  s->olabel_sorted = olabel;
  s->ilabel_sorted = !olabel;
}

Graph Graph::deep_copy(const Graph& srcg) {
  // graph.cpp:152-160: structure + weights copied, not on the tape, sort flags dropped
  Graph out(srcg.g->calc_grad);
  Structure& a = *srcg.s;
  if (a.kind == KIND_LINEAR) {
    out.s->kind = KIND_LINEAR;
    out.s->N = a.N;
    out.s->A = a.A;
    out.s->M = a.M;
    out.s->C = a.C;
    // the reference's deepCopy does not carry the sort flags; a linear graph
    // stays implicitly sorted only while flagged, so materialise the copy
    out.s->materialize();
  } else {
    a.ensure_host();
    out.s->N = a.N;
    out.s->A = a.A;
    out.s->src = a.src;
    out.s->dst = a.dst;
    out.s->il = a.il;
    out.s->ol = a.ol;
    out.s->nflags = a.nflags;
    out.s->start = a.start;
    out.s->accept = a.accept;
    // adjacency lists are copied too (deepCopy copies the Node vectors)
    if (a.csr_valid) {
      out.s->in_off = a.in_off;
      out.s->in_list = a.in_list;
      out.s->out_off = a.out_off;
      out.s->out_list = a.out_list;
      out.s->csr_valid = true;
    }
  }
  srcg.w->ensure_host();
  out.w->host = srcg.w->host;
  out.w->n = srcg.w->n;
  return out;
}

const float* Graph::weights_host(bool mut) {
  if (!w) throw_logic("[Graph::weights] graph has no weights");
  w->ensure_host();
  if (mut) {
    w->host_escaped = true;
    w->dev_valid = false;
    w->version++;
  }
  return w->host.data();
}

void Graph::set_weights_host(const float* p) {
  // graph.cpp:179-181
  w->host.assign(p, p + s->A);
  w->n = s->A;
  w->zero = false;
  w->staged.reset();
  w->host_valid = true;
  w->dev_valid = false;
  w->version++;
}

void Graph::set_weights_device(const void* p) {
  Runtime& rt = Runtime::get();
  int64_t n = s->A;
  if (!w->dev || w->n != n || !w->dev_mem || w->dev_mem.use_count() > 1) {
    w->dev_mem = rt.alloc(sizeof(float) * size_t(n));
    w->dev = w->dev_mem->as<float>();
  }
  w->n = n;
  rt.d2d(w->dev, p, sizeof(float) * size_t(n));
  w->zero = false;
  w->staged.reset();
  w->dev_valid = true;
  w->host_valid = false;
  w->host_escaped = false;
  w->version++;
}

Graph& Graph::grad() {
  // graph.cpp:81-89
  if (!g->calc_grad) throw_logic("[Graph::grad] Gradient calculation disabled.");
  if (g->lazy_ptr) {
    std::lock_guard<std::mutex> lk(s->grad_lock);
    materialize_grad();
  }
  if (!g->grad) throw_logic("[Graph::grad] Gradient not calculated yet.");
  return *g->grad;
}

void Graph::set_calc_grad(bool c) {
  // graph.cpp:131-138
  g->calc_grad = c;
  if (!c) {
    g->op.reset();
    g->has_grad_fn = false;
    g->inputs.clear();
    zero_grad();
  }
}

static void make_grad_graph(Graph& self) {
  self.g->grad = std::unique_ptr<Graph>(new Graph(false, self.s));  // shares the structure (graph.cpp:102-103)
}

// (caller holds grad_lock)
void Graph::materialize_grad() {
  if (!g->lazy_ptr) return;
  make_grad_graph(*this);
  Weights& gw = *g->grad->w;
  gw.n = s->A;
  gw.host_valid = false;
  gw.dev_mem = std::move(g->lazy_owner);
  gw.dev = g->lazy_ptr;
  gw.dev_valid = true;
  g->lazy_ptr = nullptr;
}

void Graph::add_grad_host(const float* v, int64_t n) {
  if (!calc_grad()) return;
  if (n != s->A) throw_logic("[Graph::addGrad] Invalid grad size.");  // graph.cpp:93-95
  std::lock_guard<std::mutex> lk(s->grad_lock);
  materialize_grad();
  if (is_grad_available()) {
    Weights& gw = *g->grad->w;
    gw.ensure_host();
    for (int64_t i = 0; i < n; ++i) gw.host[i] += v[i];
    gw.dev_valid = false;
  } else {
    make_grad_graph(*this);
    Weights& gw = *g->grad->w;
    gw.host.assign(v, v + n);
    gw.n = n;
  }
}

void Graph::add_grad_device(const DevMemP& owner, float* dev, bool adopt) {
  if (!calc_grad()) return;
  Runtime& rt = Runtime::get();
  // sizes still on the device: only the "first gradient, adopt the buffer" case can go
  // on without them (the gradient graph's arc count is filled in when they arrive)
  if (s->deferred && !(adopt && !is_grad_available())) s->resolve_sizes();
  int64_t n = s->A;
  std::lock_guard<std::mutex> lk(s->grad_lock);
  if (!is_grad_available() && adopt && !s->deferred) {  // noted only: materialize_grad() builds the graph on demand
    g->lazy_owner = owner;
    g->lazy_ptr = dev;
    return;
  }
  materialize_grad();
  if (!is_grad_available()) {
    make_grad_graph(*this);
    Weights& gw = *g->grad->w;
    gw.n = n;
    gw.host_valid = false;
    if (s->deferred) s->deferred->grads.push_back({g->grad->w, s->deferred_idx});
    if (adopt) {
      gw.dev_mem = owner;
      gw.dev = dev;
    } else {
      gw.dev_mem = rt.alloc(sizeof(float) * size_t(n));
      gw.dev = gw.dev_mem->as<float>();
      rt.d2d(gw.dev, dev, sizeof(float) * size_t(n));
    }
    gw.dev_valid = true;
    return;
  }
  Weights& gw = *g->grad->w;
  if (!gw.dev_valid || (gw.host_escaped && gw.host_valid)) {
    std::vector<Weights*> v{&gw};
    ensure_weights_device_batch(v);
  }
  AxpyArgs a{gw.dev, dev, n, 1.0f};
  PinnedMemP pin = rt.alloc_pinned(sizeof(AxpyArgs));
  *pin->as<AxpyArgs>() = a;
  DevMemP d = rt.alloc(sizeof(AxpyArgs));
  rt.h2d_pinned(d->ptr, pin->ptr, sizeof(AxpyArgs));
  launch_axpy_batch(d->as<AxpyArgs>(), 1, n, 0, rt.stream());
  gw.host_valid = false;
  gw.host_escaped = false;  // the device copy is the live one now
  gw.version++;
}

// ======================================================================
// residency: one staging buffer + one copy per batch
// ======================================================================
namespace {
struct Packer {
  // lays out blobs at 256-byte aligned offsets inside one allocation
  size_t total = 0;
  size_t add(size_t bytes) {
    size_t off = total;
    total = align_up(total + bytes, 256);
    return off;
  }
};
} // namespace

void ensure_device_batch(const std::vector<Structure*>& ss) {
  Runtime& rt = Runtime::get();
  std::vector<Structure*> todo;
  std::unordered_set<Structure*> todo_set;
  for (Structure* s : ss) {
    if (s->kind == KIND_LINEAR) {
      DGraph& v = s->dview;
      std::memset(&v, 0, sizeof(v));
      v.kind = KIND_LINEAR;
      v.N = int(s->N);
      v.A = int(s->A);
      v.M = s->M;
      v.C = s->C;
      v.n_start = 1;
      v.n_accept = s->M > 0 ? 1 : 0;
      v.flags = 3 | GF_EPS_FREE | GF_ACCEPTOR;
      s->dev_valid = true;
      continue;
    }
    if (!s->dev_valid && todo_set.insert(s).second) todo.push_back(s);
  }
  if (todo.empty()) return;
  struct Off {
    size_t src, dst, il, ol, nf, st, ac, oo, ol_, io, il_, orec, irec;
  };
  Packer pk;
  std::vector<Off> offs(todo.size());
  // per-graph host work (CSR build, packing) fans out over a few host threads when
  // the batch is large -- a training step uploads one small target graph per utterance
  auto for_each_todo = [&](auto&& body) {
    if (todo.size() >= 64) gtn::detail::runIndexed(todo.size(), body, 16, false);
    else for (size_t i = 0; i < todo.size(); ++i) body(i);
  };
  for_each_todo([&](size_t i) { todo[i]->ensure_csr(); });
  for (size_t i = 0; i < todo.size(); ++i) {
    Structure* s = todo[i];
    size_t A = size_t(s->A), N = size_t(s->N);
    Off& o = offs[i];
    o.src = pk.add(4 * A);
    o.dst = pk.add(4 * A);
    o.il = pk.add(4 * A);
    o.ol = pk.add(4 * A);
    o.nf = pk.add(N);
    o.st = pk.add(4 * s->start.size());
    o.ac = pk.add(4 * s->accept.size());
    o.oo = pk.add(4 * (N + 1));
    o.ol_ = pk.add(4 * A);
    o.io = pk.add(4 * (N + 1));
    o.il_ = pk.add(4 * A);
    o.orec = pk.add(16 * A);
    o.irec = pk.add(16 * A);
  }
  PinnedMemP pin = rt.alloc_pinned(pk.total);
  DevMemP dev = rt.alloc(pk.total);
  char* hb = pin->as<char>();
  char* db = dev->as<char>();
  for_each_todo([&](size_t i) {
    Structure* s = todo[i];
    size_t A = size_t(s->A), N = size_t(s->N);
    const Off& o = offs[i];
    std::memcpy(hb + o.src, s->src.data(), 4 * A);
    std::memcpy(hb + o.dst, s->dst.data(), 4 * A);
    std::memcpy(hb + o.il, s->il.data(), 4 * A);
    std::memcpy(hb + o.ol, s->ol.data(), 4 * A);
    std::memcpy(hb + o.nf, s->nflags.data(), N);
    std::memcpy(hb + o.st, s->start.data(), 4 * s->start.size());
    std::memcpy(hb + o.ac, s->accept.data(), 4 * s->accept.size());
    std::memcpy(hb + o.oo, s->out_off.data(), 4 * (N + 1));
    std::memcpy(hb + o.ol_, s->out_list.data(), 4 * A);
    std::memcpy(hb + o.io, s->in_off.data(), 4 * (N + 1));
    std::memcpy(hb + o.il_, s->in_list.data(), 4 * A);
    bool eps_free = true, acceptor = true;
    {
      int* orec = reinterpret_cast<int*>(hb + o.orec);
      int* irec = reinterpret_cast<int*>(hb + o.irec);
      for (size_t k = 0; k < A; ++k) eps_free = eps_free && s->il[k] >= 0 && s->ol[k] >= 0;
      for (size_t k = 0; k < A && acceptor; ++k) acceptor = s->il[k] == s->ol[k];
      for (size_t k = 0; k < A; ++k) {
        const int ao = s->out_list[k], ai = s->in_list[k];
        orec[4 * k + 0] = s->il[ao]; orec[4 * k + 1] = s->ol[ao]; orec[4 * k + 2] = s->dst[ao]; orec[4 * k + 3] = ao;
        irec[4 * k + 0] = s->il[ai]; irec[4 * k + 1] = s->ol[ai]; irec[4 * k + 2] = s->src[ai]; irec[4 * k + 3] = ai;
      }
    }
    DGraph& v = s->dview;
    std::memset(&v, 0, sizeof(v));
    v.kind = KIND_EXPLICIT;
    v.N = int(N);
    v.A = int(A);
    v.n_start = int(s->start.size());
    v.n_accept = int(s->accept.size());
    v.flags = (s->ilabel_sorted ? 1 : 0) | (s->olabel_sorted ? 2 : 0) | (eps_free ? GF_EPS_FREE : 0) | (acceptor ? GF_ACCEPTOR : 0);
    v.src = reinterpret_cast<const int*>(db + o.src);
    v.dst = reinterpret_cast<const int*>(db + o.dst);
    v.il = reinterpret_cast<const int*>(db + o.il);
    v.ol = reinterpret_cast<const int*>(db + o.ol);
    v.nflags = reinterpret_cast<const uint8_t*>(db + o.nf);
    v.start_list = reinterpret_cast<const int*>(db + o.st);
    v.accept_list = reinterpret_cast<const int*>(db + o.ac);
    v.out_off = reinterpret_cast<const int*>(db + o.oo);
    v.out_list = reinterpret_cast<const int*>(db + o.ol_);
    v.in_off = reinterpret_cast<const int*>(db + o.io);
    v.in_list = reinterpret_cast<const int*>(db + o.il_);
    v.out_rec = reinterpret_cast<const gtnx_i4*>(db + o.orec);
    v.in_rec = reinterpret_cast<const gtnx_i4*>(db + o.irec);
    s->dev_mem = dev;
    s->dev_valid = true;
  });
  rt.h2d_pinned(dev->ptr, pin->ptr, pk.total);
}

void ensure_weights_device_batch(const std::vector<Weights*>& ws) {
  Runtime& rt = Runtime::get();
  std::vector<Weights*> todo;
  std::unordered_set<Weights*> todo_set;
  for (Weights* w : ws) {
    w->settle_staged();  // (copied by a region's join already: region.cpp)
    // (a mutable host pointer that escaped matters only while the host copy is the live one: after a
    //  device-side write the device copy is authoritative and host_valid is false)
    bool stale = !w->dev_valid || (w->host_escaped && w->host_valid);
    if (stale && todo_set.insert(w).second) todo.push_back(w);
  }
  if (todo.empty()) return;
  Packer pk;
  std::vector<size_t> offs(todo.size());
  for (size_t i = 0; i < todo.size(); ++i) {
    Weights* w = todo[i];
    if (w->staged && w->staged->on_device) {  // region.cpp: the caller's device buffer, not copied yet
      w->dev_mem = rt.alloc(sizeof(float) * size_t(w->n ? w->n : 1));
      w->dev = w->dev_mem->as<float>();
      rt.d2d(w->dev, w->staged->src, sizeof(float) * size_t(w->n));
      w->staged.reset();
      w->dev_valid = true;
      todo[i] = nullptr;
      continue;
    }
    if (!w->host_valid && !w->zero && !w->staged) throw_logic("weights valid on neither host nor device");
    offs[i] = pk.add(4 * size_t(w->n));
  }
  if (pk.total == 0) return;
  PinnedMemP pin = rt.alloc_pinned(pk.total);
  DevMemP dev = rt.alloc(pk.total);
  for (size_t i = 0; i < todo.size(); ++i) {
    Weights* w = todo[i];
    if (!w) continue;
    if (w->zero) {
      std::memset(pin->as<char>(offs[i]), 0, 4 * size_t(w->n));
      w->zero = false;  // (the device copy is the live one now; the host copy is made on demand)
    } else if (w->staged) {
      std::memcpy(pin->as<char>(offs[i]), w->staged->src, 4 * size_t(w->n));
      w->staged.reset();
    } else {
      std::memcpy(pin->as<char>(offs[i]), w->host.data(), 4 * size_t(w->n));
    }
    w->dev_mem = dev;
    w->dev = dev->as<float>(offs[i]);
    w->dev_valid = true;
  }
  rt.h2d_pinned(dev->ptr, pin->ptr, pk.total);
}

DGraph device_view(Graph& gr) {
  gr.s->ensure_full();
  DGraph v = gr.s->dview;
  v.w = gr.w ? gr.w->dev : nullptr;
  return v;
}

// ======================================================================
// level schedule (host pre-pass for host-built structures)
// Replays the reference's Kahn FIFO (shortest.cpp:96-145) on the structure only,
// so positions == the reference's processing order; levels are the dependency
// depth, which the FIFO order is sorted by.
// ======================================================================
namespace {
struct HostSched {
  bool error = false;
  int P = 0, L = 0;
  std::vector<int> level_off, row_off, in_srcpos, in_arc, in_rank, acc_pos, out_off, out_dstpos, out_arc;
  std::vector<uint8_t> pflags;
  int max_width = 0;
  int max_level_arcs = 0, max_reach = 0;
  bool all_written = false;
};

void build_host_schedule(Structure& s, HostSched& h, bool need_rank) {
  s.ensure_csr();
  const int N = int(s.N);
  std::vector<int> deg(N), order, pos(N, -1), level(N, 0);
  order.reserve(N);
  for (int n = 0; n < N; ++n) deg[n] = s.in_off[n + 1] - s.in_off[n];
  size_t head = 0;
  for (int n : s.start)
    if (deg[n] == 0) order.push_back(n);
  const size_t n_seed = order.size();
  while (head < order.size()) {
    int n = order[head];
    ++head;
    for (int k = s.out_off[n]; k < s.out_off[n + 1]; ++k) {
      int d = s.dst[s.out_list[k]];
      level[d] = std::max(level[d], level[n] + 1);
      if (--deg[d] == 0) order.push_back(d);
    }
  }
  // shortest.cpp:148-152: an accept node the queue never reached is an error only
  // if it still waits on a predecessor; a non-start accept node WITHOUT in-arcs is
  // never queued but keeps its zero-initialised score (shortest.cpp:89) and takes
  // part in the final reduction with 0.0.  Schedule those at level 0, flagged.
  std::vector<uint8_t> scheduled(N, 0);
  for (int n : order) scheduled[n] = 1;
  std::vector<int> orphans;
  for (int n : s.accept) {
    if (scheduled[n]) continue;
    if (deg[n] > 0)
      h.error = true;
    else
      orphans.push_back(n);
  }
  std::vector<uint8_t> orphan_flag(N, 0);
  if (!orphans.empty()) {
    for (int n : orphans) {
      orphan_flag[n] = 1;
      level[n] = 0;
    }
    order.insert(order.begin() + n_seed, orphans.begin(), orphans.end());
  }
  for (size_t p = 0; p < order.size(); ++p) pos[order[p]] = int(p);
  h.P = int(order.size());
  // levels (FIFO order is level-sorted)
  h.level_off.clear();
  int cur = -1;
  for (int p = 0; p < h.P; ++p) {
    int lv = level[order[p]];
    while (cur < lv) {
      h.level_off.push_back(p);
      ++cur;
    }
  }
  h.L = int(h.level_off.size());
  h.level_off.push_back(h.P);
  for (int l = 0; l < h.L; ++l) h.max_width = std::max(h.max_width, h.level_off[l + 1] - h.level_off[l]);
  // reverse-Kahn "processed" flags (shortest.cpp:45-53, 76-78): a node's arcs get
  // gradients only if it is popped from the reverse queue
  std::vector<int> odeg(N);
  std::vector<uint8_t> proc(N, 0);
  std::vector<int> rq;
  for (int n = 0; n < N; ++n) odeg[n] = s.out_off[n + 1] - s.out_off[n];
  for (int n : s.accept)
    if (odeg[n] == 0) rq.push_back(n);
  for (size_t i = 0; i < rq.size(); ++i) {
    int n = rq[i];
    proc[n] = 1;
    for (int k = s.in_off[n]; k < s.in_off[n + 1]; ++k) {
      int u = s.src[s.in_list[k]];
      if (--odeg[u] == 0) rq.push_back(u);
    }
  }
  // rows
  h.row_off.assign(h.P + 1, 0);
  h.out_off.assign(h.P + 1, 0);
  h.pflags.assign(h.P, 0);
  std::vector<int> out_index_of_arc;  // push rank: index in the position-ordered out CSR
  if (need_rank) out_index_of_arc.assign(size_t(s.A), 0);
  for (int p = 0; p < h.P; ++p) {
    int n = order[p];
    h.pflags[p] = uint8_t(s.nflags[n] | (orphan_flag[n] ? NF_ORPHAN : 0));
    h.row_off[p + 1] = h.row_off[p] + (s.in_off[n + 1] - s.in_off[n]);
  }
  h.in_srcpos.resize(h.row_off[h.P]);
  h.in_arc.resize(h.row_off[h.P]);
  for (int p = 0; p < h.P; ++p) {
    int n = order[p], o = h.row_off[p];
    for (int k = s.in_off[n]; k < s.in_off[n + 1]; ++k, ++o) {
      int a = s.in_list[k];
      h.in_arc[o] = a;
      h.in_srcpos[o] = pos[s.src[a]];  // >= 0: all predecessors of a scheduled node are scheduled
    }
  }
  for (int l = 0; l < h.L; ++l) {
    const int lo = h.level_off[l], hi = h.level_off[l + 1];
    h.max_level_arcs = std::max(h.max_level_arcs, h.row_off[hi] - h.row_off[lo]);
    int mn = lo;
    for (int k = h.row_off[lo]; k < h.row_off[hi]; ++k) mn = std::min(mn, h.in_srcpos[k]);
    h.max_reach = std::max(h.max_reach, hi - mn);
  }
  bool all = (h.P == N);
  int64_t written = 0;
  for (int p = 0; p < h.P; ++p) {
    int n = order[p];
    int cnt = 0;
    for (int k = s.out_off[n]; k < s.out_off[n + 1]; ++k) {
      int d = s.dst[s.out_list[k]];
      if (pos[d] >= 0 && proc[d]) ++cnt;
    }
    h.out_off[p + 1] = h.out_off[p] + cnt;
  }
  h.out_dstpos.resize(h.out_off[h.P]);
  h.out_arc.resize(h.out_off[h.P]);
  {
    int rank = 0;
    for (int p = 0; p < h.P; ++p) {
      int n = order[p], o = h.out_off[p];
      for (int k = s.out_off[n]; k < s.out_off[n + 1]; ++k, ++rank) {
        int a = s.out_list[k];
        if (need_rank) out_index_of_arc[a] = rank;
        int d = s.dst[a];
        if (pos[d] >= 0 && proc[d]) {
          h.out_dstpos[o] = pos[d];
          h.out_arc[o] = a;
          ++o;
          ++written;
        }
      }
    }
  }
  h.all_written = all && written == s.A;
  if (need_rank) {
    h.in_rank.resize(h.in_arc.size());
    for (size_t i = 0; i < h.in_arc.size(); ++i) h.in_rank[i] = out_index_of_arc[h.in_arc[i]];
  }
  h.acc_pos.resize(s.accept.size());
  for (size_t k = 0; k < s.accept.size(); ++k) h.acc_pos[k] = pos[s.accept[k]];
}
} // namespace

// The schedule of a structure that only exists on the device (a composition result that is not layered),
// built there (levelize.hip).  Without the queue's order inside a level, so not for viterbiPath (need_rank).
// GTNX_DEVICE_LEVELIZE=1 takes this way for every explicit structure (parity suite), =0 never.
namespace {
bool device_schedule_wanted(const Structure& s, bool need_rank) {
  if (need_rank || s.kind == KIND_LINEAR) return false;
  static const char* env = getenv("GTNX_DEVICE_LEVELIZE");
  if (env) return env[0] == '1';
  return !s.host_valid && s.dev_valid;
}
void build_device_schedule(Structure& s) {
  Runtime& rt = Runtime::get();
  GTNX_PROF("device_levelize", 0.0);
  if (!s.dev_valid) {
    std::vector<Structure*> one{&s};
    ensure_device_batch(one);
  }
  s.ensure_full();
  const DGraph g = s.dview;
  const size_t N = size_t(g.N), A = size_t(g.A), na = size_t(g.n_accept);
  Packer pk;
  const size_t o_lv = pk.add(4 * (N + 2)), o_ro = pk.add(4 * (N + 1)), o_sp = pk.add(4 * (A ? A : 1)), o_ia = pk.add(4 * (A ? A : 1)),
               o_pf = pk.add(N ? N : 1), o_ap = pk.add(4 * (na ? na : 1)), o_oo = pk.add(4 * (N + 1)), o_od = pk.add(4 * (A ? A : 1)),
               o_oa = pk.add(4 * (A ? A : 1));
  DevMemP dev = rt.alloc(pk.total);
  char* db = dev->as<char>();
  LevelizeOut out{reinterpret_cast<int*>(db + o_lv), reinterpret_cast<int*>(db + o_ro), reinterpret_cast<int*>(db + o_sp),
                  reinterpret_cast<int*>(db + o_ia), reinterpret_cast<uint8_t*>(db + o_pf), reinterpret_cast<int*>(db + o_ap),
                  reinterpret_cast<int*>(db + o_oo), reinterpret_cast<int*>(db + o_od), reinterpret_cast<int*>(db + o_oa)};
  DevMemP scratch = rt.alloc(levelize_scratch_bytes(int(N), int(A)));
  int info[LV_INFO_INTS];
  device_levelize(g, out, scratch->ptr, info, rt.stream());
  auto sc = std::make_shared<Schedule>();
  sc->error = info[LV_ERROR] != 0;
  sc->mem = dev;
  sc->max_level_width = info[LV_MAX_WIDTH];
  sc->max_level_arcs = info[LV_MAX_LEVEL_ARCS];
  sc->max_reach = info[LV_MAX_REACH];
  sc->n_in = info[LV_N_IN];
  sc->n_out = info[LV_N_OUT];
  sc->all_written = info[LV_P] == int(N) && info[LV_N_OUT] == int(A);
  sc->has_rank = false;
  DSched& v = sc->view;
  v.P = info[LV_P];
  v.L = info[LV_L];
  v.n_accept = int(na);
  // rows are copies of the device in-lists: the reference's list order for an uploaded host graph (ties by row
  // slot), but UNORDERED for a composition result (compose fills them through atomic cursors): ties by arc id,
  // which is the order the reference's lists would have
  v.flags = g.in_rec ? 0 : SCHED_TIE_BY_ARC;
  v.level_off = out.level_off;
  v.row_off = out.row_off;
  v.in_srcpos = out.in_srcpos;
  v.in_arc = out.in_arc;
  v.in_rank = nullptr;
  v.in_w = nullptr;
  v.pflags = out.pflags;
  v.acc_pos = out.acc_pos;
  v.out_off = out.out_off;
  v.out_dstpos = out.out_dstpos;
  v.out_arc = out.out_arc;
  s.sched = sc;
}
}  // namespace

void ensure_schedule_batch(const std::vector<Structure*>& ss, bool need_rank) {
  Runtime& rt = Runtime::get();
  std::vector<Structure*> todo;
  for (Structure* s : ss) {
    if (s->kind == KIND_LINEAR) continue;
    if (s->sched && (!need_rank || s->sched->has_rank || (s->sched->view.flags & SCHED_TIE_BY_ARC))) continue;
    if (std::find(todo.begin(), todo.end(), s) != todo.end()) continue;
    if (device_schedule_wanted(*s, need_rank)) build_device_schedule(*s);
    else todo.push_back(s);
  }
  if (todo.empty()) return;
  std::vector<HostSched> hs(todo.size());
  struct Off {
    size_t lv, ro, sp, ia, ir, pf, ap, oo, od, oa;
  };
  std::vector<Off> offs(todo.size());
  Packer pk;
  for (size_t i = 0; i < todo.size(); ++i) {
    {
      GTNX_HOST_T("schedule.host_queue_replay");
      build_host_schedule(*todo[i], hs[i], need_rank);
    }
    HostSched& h = hs[i];
    Off& o = offs[i];
    o.lv = pk.add(4 * h.level_off.size());
    o.ro = pk.add(4 * h.row_off.size());
    o.sp = pk.add(4 * h.in_srcpos.size());
    o.ia = pk.add(4 * h.in_arc.size());
    o.ir = pk.add(4 * h.in_rank.size());
    o.pf = pk.add(h.pflags.size());
    o.ap = pk.add(4 * h.acc_pos.size());
    o.oo = pk.add(4 * h.out_off.size());
    o.od = pk.add(4 * h.out_dstpos.size());
    o.oa = pk.add(4 * h.out_arc.size());
  }
  PinnedMemP pin = rt.alloc_pinned(pk.total);
  DevMemP dev = rt.alloc(pk.total);
  char* hb = pin->as<char>();
  char* db = dev->as<char>();
  auto put = [&](size_t off, const void* p, size_t bytes) {
    if (bytes) std::memcpy(hb + off, p, bytes);
  };
  for (size_t i = 0; i < todo.size(); ++i) {
    HostSched& h = hs[i];
    const Off& o = offs[i];
    put(o.lv, h.level_off.data(), 4 * h.level_off.size());
    put(o.ro, h.row_off.data(), 4 * h.row_off.size());
    put(o.sp, h.in_srcpos.data(), 4 * h.in_srcpos.size());
    put(o.ia, h.in_arc.data(), 4 * h.in_arc.size());
    put(o.ir, h.in_rank.data(), 4 * h.in_rank.size());
    put(o.pf, h.pflags.data(), h.pflags.size());
    put(o.ap, h.acc_pos.data(), 4 * h.acc_pos.size());
    put(o.oo, h.out_off.data(), 4 * h.out_off.size());
    put(o.od, h.out_dstpos.data(), 4 * h.out_dstpos.size());
    put(o.oa, h.out_arc.data(), 4 * h.out_arc.size());
    auto sc = std::make_shared<Schedule>();
    sc->error = h.error;
    sc->mem = dev;
    sc->max_level_width = h.max_width;
    sc->max_level_arcs = h.max_level_arcs;
    sc->max_reach = h.max_reach;
    sc->n_in = int64_t(h.in_arc.size());
    sc->n_out = int64_t(h.out_arc.size());
    sc->all_written = h.all_written;
    sc->has_rank = need_rank;
    DSched& v = sc->view;
    v.P = h.P;
    v.L = h.L;
    v.n_accept = int(h.acc_pos.size());
    v.flags = 0;
    v.level_off = reinterpret_cast<const int*>(db + o.lv);
    v.row_off = reinterpret_cast<const int*>(db + o.ro);
    v.in_srcpos = reinterpret_cast<const int*>(db + o.sp);
    v.in_arc = reinterpret_cast<const int*>(db + o.ia);
    v.in_rank = need_rank ? reinterpret_cast<const int*>(db + o.ir) : nullptr;
    v.in_w = nullptr;
    v.pflags = reinterpret_cast<const uint8_t*>(db + o.pf);
    v.acc_pos = reinterpret_cast<const int*>(db + o.ap);
    v.out_off = reinterpret_cast<const int*>(db + o.oo);
    v.out_dstpos = reinterpret_cast<const int*>(db + o.od);
    v.out_arc = reinterpret_cast<const int*>(db + o.oa);
    todo[i]->sched = sc;
  }
  rt.h2d_pinned(dev->ptr, pin->ptr, pk.total);
}

// ======================================================================
// equal / isomorphic (host fixtures of the parity suite)
// ======================================================================
namespace {
struct HostView {
  Structure* s;
  const float* w;
};
HostView host_view(Graph& g) {
  g.s->ensure_csr();
  g.w->ensure_host();
  return {g.s.get(), g.w->host.data()};
}
} // namespace

bool graphs_equal(Graph& a, Graph& b) {
  // utils.cpp:45-77: same node ids, per-node multiset of out arcs, arc order ignored
  if (a.num_nodes() != b.num_nodes() || a.num_start() != b.num_start() ||
      a.num_accept() != b.num_accept() || a.num_arcs() != b.num_arcs())
    return false;
  HostView x = host_view(a), y = host_view(b);
  for (int n = 0; n < x.s->N; ++n) {
    if (x.s->num_in(n) != y.s->num_in(n) || x.s->num_out(n) != y.s->num_out(n) ||
        x.s->nflags[n] != y.s->nflags[n])
      return false;
    std::list<int> bout(y.s->out_list.begin() + y.s->out_off[n], y.s->out_list.begin() + y.s->out_off[n + 1]);
    for (int k = x.s->out_off[n]; k < x.s->out_off[n + 1]; ++k) {
      int a1 = x.s->out_list[k];
      auto it = bout.begin();
      for (; it != bout.end(); ++it) {
        int a2 = *it;
        if (x.s->dst[a1] == y.s->dst[a2] && x.s->src[a1] == y.s->src[a2] && x.s->il[a1] == y.s->il[a2] &&
            x.s->ol[a1] == y.s->ol[a2] && x.w[a1] == y.w[a2])
          break;
      }
      if (it == bout.end()) return false;
      bout.erase(it);
    }
  }
  return true;
}

namespace {
bool iso_rec(HostView& x, HostView& y, int n1, int n2, std::map<std::pair<int, int>, int>& visited) {
  // utils.cpp:79-123
  auto ins = visited.insert({{n1, n2}, n1});
  if (!ins.second) return ins.first->second >= 0;
  if (x.s->num_in(n1) != y.s->num_in(n2) || x.s->num_out(n1) != y.s->num_out(n2) ||
      x.s->nflags[n1] != y.s->nflags[n2]) {
    ins.first->second = -1;
    return false;
  }
  std::list<int> bout(y.s->out_list.begin() + y.s->out_off[n2], y.s->out_list.begin() + y.s->out_off[n2 + 1]);
  for (int k = x.s->out_off[n1]; k < x.s->out_off[n1 + 1]; ++k) {
    int a1 = x.s->out_list[k];
    auto it = bout.begin();
    for (; it != bout.end(); ++it) {
      int a2 = *it;
      if (x.s->il[a1] != y.s->il[a2] || x.s->ol[a1] != y.s->ol[a2] || x.w[a1] != y.w[a2]) continue;
      if (iso_rec(x, y, x.s->dst[a1], y.s->dst[a2], visited)) break;
    }
    if (it == bout.end()) {
      visited[{n1, n2}] = -1;
      return false;
    }
    bout.erase(it);
  }
  return true;
}
} // namespace

bool graphs_isomorphic(Graph& a, Graph& b) {
  // utils.cpp:125-150
  if (a.num_nodes() != b.num_nodes() || a.num_start() != b.num_start() ||
      a.num_accept() != b.num_accept() || a.num_arcs() != b.num_arcs())
    return false;
  HostView x = host_view(a), y = host_view(b);
  std::map<std::pair<int, int>, int> visited;
  std::list<int> s2(y.s->start.begin(), y.s->start.end());
  for (int s1 : x.s->start) {
    auto it = s2.begin();
    for (; it != s2.end(); ++it)
      if (iso_rec(x, y, s1, *it, visited)) break;
    if (it == s2.end()) return false;
    s2.erase(it);
  }
  return true;
}

} // namespace gtnx
