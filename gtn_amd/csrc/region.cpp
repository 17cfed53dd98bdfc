// region.cpp -- see region.h.  Replaces the reference's per-utterance execution inside
// gtn::parallelMap (gtn/parallel/parallel_map.h:153-188, benchmarks/ctc.cpp:150-165) with deferred calls
// that the region's join runs as batch records (batch.h).
//
// Layout of the file: slices (what a thread records) -> staging of weights -> the join (Run: merged groups,
// the slice-level path, the call-by-call path) -> the entry points of region.h.
#include "region.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>

namespace gtnx {
extern thread_local bool t_reclaim_at_wait;  // ops_internal.h: a run that waits for the GPU reclaims at that wait

namespace {

// ---- slices ---------------------------------------------------------------------------------------------
// how the calls of a slice group read input side k
enum InKind : uint8_t {
  IN_NONE = 0,  // (no call yet)
  IN_LEAF,      // every call reads a plain graph
  IN_ALIGNED,   // call i reads the result of call i of ONE earlier group of the same slice (still queued)
  IN_DONE,      // every call reads an element of ONE record that exists already (`done_idx`)
  IN_MIXED      // anything else: the group takes the call-by-call path
};
enum LeafKind : uint8_t {
  LK_NONE = 0,
  LK_CTC,     // acceptors that are each exactly ctcGraph(labels) with all-zero weights (batch.cpp: CTC_TARGETS)
  LK_LINEAR,  // linear chains of one shape whose weights are one block in call order (batch.cpp: LINEAR)
  LK_OTHER
};
enum LinSrc : uint8_t { LS_NONE = 0, LS_STAGED, LS_RESIDENT };

struct Slice;

// what the recording thread found out about the plain graphs on one input side of a group
struct LeafDigest {
  uint8_t kind = LK_NONE;
  std::vector<Graph> graphs;        // the inputs themselves, in call order (they become Batch::graphs)
  // LK_CTC
  std::vector<int> labels, len;
  int blank = 0;
  // LK_LINEAR
  uint8_t src = LS_NONE;
  int M = 0, C = 0;
  size_t first_off = 0;             // LS_STAGED: byte offset of call 0's weights in the slice's stage block
  const float* first_dev = nullptr; // LS_RESIDENT: call 0's device address
  DevMem* mem = nullptr;            //              and its owner
  // the same weights were the leaf of call i of another group of this slice on `alias_side`: one record serves both
  SliceGroup* alias = nullptr;
  int alias_side = 0;
  // LS_RESIDENT: the record an earlier call over the same graphs made (Weights::leaf_batch), if every call agrees
  BatchP reuse;
  // gtnx_grads_bind_device_n: call i's first gradient goes to dest0 + i * M * C
  float* dest0 = nullptr;
  DevMemP dest_mem;
  bool dest_ok = false;
  bool cg = false;
};

}  // namespace

// the calls of one (depth, function, mode) recorded by ONE thread, in the order they were made
struct SliceGroup {
  Slice* slice = nullptr;
  RegionOp op = RO_NEG;
  int depth = 0;
  int8_t mode = -1;
  std::vector<Pending*> calls;
  uint8_t in_kind[2] = {IN_NONE, IN_NONE};
  SliceGroup* prod[2] = {nullptr, nullptr};  // IN_ALIGNED
  Batch* done_rec[2] = {nullptr, nullptr};   // IN_DONE: the record ...
  BatchP done_hold[2];
  std::vector<int> done_idx[2];              // ... and the element each call reads
  bool done_identity[2] = {true, true};      // done_idx[k][i] == i
  LeafDigest leaf[2];
  std::atomic<int> individual{0};  // members that ran (or failed) on their own: the group is no longer one unit
  // the result of the group when it ran as one: elements base .. base + calls.size() of `result` (or, through
  // `perm`, element perm[i] for call i)
  BatchP result;
  int base = 0;
  std::vector<int> perm;      // (empty: call i is element base + i)
  std::atomic<int> state{0};  // 0 queued, 1 done (failures are per call)
  // scratch of the run that executes the slice
  void* run = nullptr;
  int mgroup = -1;
};

int Pending::st() const {
  const int s = state.load(std::memory_order_acquire);
  if (s != 0 || !sg) return s;
  return sg->state.load(std::memory_order_acquire);
}

BatchP Pending::record(int* element) const {
  if (state.load(std::memory_order_acquire) == 1) {
    *element = idx;
    return batch;
  }
  if (sg && sg->state.load(std::memory_order_acquire) == 1) {
    *element = sg->perm.empty() ? sg->base + local : sg->perm[size_t(local)];
    return sg->result;
  }
  *element = -1;
  return nullptr;
}

namespace {

inline bool is_placeholder(const Graph& g) { return g.s && g.s->pending; }

void count_use(const Graph& g, int d) {
  if (!g.s || is_placeholder(g)) return;
  g.s->pending_uses += d;
  if (g.w) g.w->pending_uses += d;
}

}  // namespace

void Pending::release_inputs() {
  if (uses_counted) {
    count_use(a, -1);
    count_use(b, -1);
    uses_counted = false;
  }
  a = Graph(Graph::Empty{});
  b = Graph(Graph::Empty{});
}

void Pending::recycle() {
  release_inputs();
  op = RO_NEG;
  depth = 1;
  state.store(0, std::memory_order_relaxed);
  err = nullptr;
  res = Graph(Graph::Empty{});
  has_res.store(false, std::memory_order_relaxed);
  batch.reset();
  idx = -1;
  sg = nullptr;
  local = -1;
  mode = -1;
}

namespace {

// a placeholder's structure and its call in one piece (a slice hands them out of chunks)
struct PlaceholderStructure : Structure {
  Pending call;
};

constexpr size_t kChunk = 64;

struct Slice {
  // placeholders and backward calls live in chunks owned by the slice; a placeholder handle is an aliasing
  // shared_ptr of the slice, so the slice (and every call in it) lives as long as any of its results
  // (A chunk outlives its slice: when the slice dies -- on the thread that recorded it, runtime.h -- its chunks go
  //  to that thread's spare list with their placeholders still constructed, only the calls reset.  A placeholder's
  //  Structure is never written to apart from `pending`, and constructing and destroying its forty members was
  //  a fifth of a microsecond per recorded call -- half a millisecond per step of the vector forms at B = 512.)
  struct Chunk {
    typename std::aligned_storage<sizeof(PlaceholderStructure), alignof(PlaceholderStructure)>::type raw[kChunk];
    size_t used = 0;   // handed out to the slice that holds the chunk
    size_t built = 0;  // constructed (>= used)
    PlaceholderStructure* at(size_t i) { return reinterpret_cast<PlaceholderStructure*>(&raw[i]); }
    ~Chunk() {
      for (size_t i = 0; i < built; ++i) at(i)->~PlaceholderStructure();
    }
  };
  // the thread's spare chunks; null once the thread is shutting down (slices may still die on it after its
  // thread_local objects are gone: the runtime's list is emptied last)
  static std::vector<std::unique_ptr<Chunk>>* spare_chunks() {
    struct Holder {
      std::vector<std::unique_ptr<Chunk>> v;
      bool* gone;
      explicit Holder(bool* g) : gone(g) {}
      ~Holder() { *gone = true; }
    };
    static thread_local bool gone = false;  // (trivially destructible: stays readable)
    if (gone) return nullptr;
    static thread_local Holder h(&gone);
    return &h.v;
  }
  std::vector<std::unique_ptr<Chunk>> chunks;
  ~Slice() {
    std::vector<std::unique_ptr<Chunk>>* spare = spare_chunks();
    if (!spare) return;  // (the chunks die with the slice)
    for (auto& c : chunks) {
      for (size_t i = 0; i < c->used; ++i) c->at(i)->call.recycle();
      c->used = 0;
      if (spare->size() < 128) spare->push_back(std::move(c));
    }
  }
  std::deque<Pending> plain;  // backward calls (no placeholder)
  std::deque<SliceGroup> groups;
  size_t n_calls = 0;
  // weights handed over inside the region
  std::vector<std::shared_ptr<Weights>> stage_host;  // host sources (pinned staging, region_stage_weights)
  struct DevSeg {
    const float* src;
    size_t off, bytes;
  };
  std::vector<DevSeg> stage_dev;                     // device sources, in call order
  std::shared_ptr<StageBlock> blk;
  size_t dev_bytes = 0;
  // the last linear leaf this thread recorded (the same emissions are usually read by two calls in a row)
  struct {
    Weights* w = nullptr;
    SliceGroup* sg = nullptr;
    int side = 0, local = -1;
  } last_lin;
  bool executed = false;
  Runtime::InboxP home;  // of the thread that recorded the slice: where it is taken apart
  int device = 0;        // the recording thread's device (runtime.h): where its calls run and its results live

  PlaceholderStructure* new_placeholder() {
    if (chunks.empty() || chunks.back()->used == kChunk) {
      std::vector<std::unique_ptr<Chunk>>* spare = spare_chunks();
      if (spare && !spare->empty()) {
        chunks.push_back(std::move(spare->back()));
        spare->pop_back();
      } else {
        chunks.emplace_back(new Chunk());
      }
    }
    Chunk& c = *chunks.back();
    PlaceholderStructure* ps;
    if (c.used < c.built) {
      ps = c.at(c.used);  // (constructed by an earlier slice, its call reset)
    } else {
      ps = new (&c.raw[c.used]) PlaceholderStructure();
      ++c.built;
    }
    ++c.used;
    return ps;
  }
  SliceGroup& group(RegionOp op, int depth, int8_t mode) {
    for (auto it = groups.rbegin(); it != groups.rend(); ++it)
      if (it->op == op && it->depth == depth && it->mode == mode) return *it;
    groups.emplace_back();
    SliceGroup& g = groups.back();
    g.slice = this;
    g.op = op;
    g.depth = depth;
    g.mode = mode;
    return g;
  }
  bool empty() const { return n_calls == 0 && stage_host.empty() && stage_dev.empty(); }
};
using SliceP = std::shared_ptr<Slice>;

// pinned staging of the region's host-source setWeights calls: ONE block for all threads, bump-allocated, so
// that the join moves the region's weights to the device with one copy (the device arena is its image)
struct StageArena {
  PinnedMemP mem;
  std::atomic<size_t> used{0};
  size_t cap = 0;
};

thread_local int t_depth = 0;  // nesting of gtnx_parallel_enter on this thread
thread_local int t_exec = 0;   // > 0: this thread is running queued calls (its own graph functions run eagerly)
thread_local SliceP t_slice;   // what this thread has recorded and not handed in yet
thread_local std::vector<Graph*> t_trash;
thread_local bool t_vector_call = false;  // recording the calls of a gtnx_*_n vector form (region_run_vector)
thread_local std::shared_ptr<StageArena> t_arena;
thread_local bool t_reclaims = false;  // this thread calls gtnx_reclaim (a pool thread): its list is emptied there

// ---- return to sender ------------------------------------------------------------------------------------
// What a region's thread builds in its tasks (target graphs, emission graphs, placeholders: a few dozen heap
// blocks per task) dies long after the task, when the caller drops the step's results -- on the caller's thread,
// or on whichever thread reclaims the runtime's garbage.  A block freed by another thread goes back to the
// allocating thread's arena under that arena's lock, and the allocating thread is by then building the NEXT
// step's graphs out of the same arena: measured on the 256-thread host of an MI355X, the tasks of a region ran two
// to five times slower while other threads freed the previous step's objects (16 -> 40-80 us per task).  So
// garbage goes home (runtime.h: Runtime::Inbox): a slice (and the leaf graphs its tasks built) is sent to the list
// of the thread that recorded it when its last reference dies, and a thread empties its own list when it has
// finished its share of a region (gtnx_reclaim), when it waits for the device, or when it next enters a region.
// a slice dies with the last of its placeholders -- usually on the thread that drops a step's results, the one
// everything waits for: it goes home
void retire_slice(Slice* s) {
  Runtime::InboxP home = std::move(s->home);
  Runtime::send(home, s, [](void* q) { delete static_cast<Slice*>(q); });
}
// the leaf graphs a batch record took from the slices' digests go home the same way (batch.cpp: Batch::~Batch)
void give_back_graphs(const std::shared_ptr<void>& home, std::vector<Graph>* part) {
  Runtime::send(std::static_pointer_cast<Runtime::Inbox>(home), part,
                [](void* q) { delete static_cast<std::vector<Graph>*>(q); });
}
Slice& my_slice() {
  if (!t_slice) {
    t_slice = SliceP(new Slice(), &retire_slice);
    t_slice->home = Runtime::home();
    t_slice->device = Runtime::current_device();
  }
  return *t_slice;
}

struct Shared {
  std::mutex mu;
  // what the region's threads handed in: a lock-free stack (thirty threads leave a region within microseconds
  // of each other; a mutex here was a convoy of 20 us per thread)
  struct Node {
    SliceP slice;
    Node* next;
  };
  std::atomic<Node*> handed{nullptr};
  void hand_in(SliceP s) {
    Node* n = new Node{std::move(s), handed.load(std::memory_order_relaxed)};
    while (!handed.compare_exchange_weak(n->next, n, std::memory_order_release, std::memory_order_relaxed)) {
    }
  }
  void take_all(std::vector<SliceP>& out) {  // oldest first
    Node* n = handed.exchange(nullptr, std::memory_order_acquire);
    const size_t at = out.size();
    while (n) {
      out.push_back(std::move(n->slice));
      Node* d = n;
      n = n->next;
      delete d;
    }
    std::reverse(out.begin() + long(at), out.end());
  }
  std::exception_ptr first_error;  // of a run nobody was there to catch (a forced closure's other members)
  std::shared_ptr<StageArena> arena;  // the current staging block (replaced when full and at every join)
  size_t arena_hint = size_t(64) << 20;  // bytes the next block starts with (what the last region needed)
  std::atomic<int> draining{0};       // threads taking the runtime's deferred list apart right now
  std::recursive_mutex exec;  // one runner at a time (recursive: a user gradFunc may call back into the engine)
};
// one per device: a region belongs to the device of the thread that joins it (runtime.h), and regions of different
// devices run side by side (gtn::parallelMapSharded)
Shared& shared() {
  static std::atomic<Shared*> per_device[64];  // never destroyed: worker threads may outlive static destruction
  const int d = Runtime::current_device() & 63;
  Shared* s = per_device[d].load(std::memory_order_acquire);
  if (!s) {
    Shared* fresh = new Shared();
    if (per_device[d].compare_exchange_strong(s, fresh, std::memory_order_acq_rel))
      s = fresh;
    else
      delete fresh;
  }
  return *s;
}

struct ExecScope {
  ExecScope() { ++t_exec; }
  ~ExecScope() { --t_exec; }
};

// ---- recording --------------------------------------------------------------------------------------------
// input side k of call `local` of group sg is the plain graph x
void note_leaf(SliceGroup& sg, int k, const Graph& x, int local) {
  Slice& sl = *sg.slice;
  LeafDigest& d = sg.leaf[k];
  uint8_t kind = LK_OTHER;
  const Structure& s = *x.s;
  const bool comp = sg.op == RO_COMPOSE || sg.op == RO_INTERSECT;
  if (x.w && x.g && s.kind == KIND_LINEAR && !s.lazy && s.M >= 1 && s.C >= 1 && !x.w->host_escaped && (comp || sg.op == RO_FS)) {
    Weights& w = *x.w;
    const size_t bytes = sizeof(float) * size_t(s.M) * size_t(s.C);
    bool ok = bytes % 16 == 0;
    uint8_t src = LS_NONE;
    size_t off = 0;
    if (w.staged && w.staged->on_device && w.staged->blk && w.staged->blk == sl.blk &&
        !w.staged->blk->base.load(std::memory_order_acquire)) {
      src = LS_STAGED;
      off = w.staged->off;
    } else {
      w.settle_staged();  // (staged by an earlier region whose join has made the copy)
      if (!w.staged && w.dev_valid && w.dev)
        src = LS_RESIDENT;
      else
        ok = false;  // host values (uploaded by the call-by-call path) / staged elsewhere
    }
    if (ok && local == 0) {
      d.src = src;
      d.M = s.M;
      d.C = s.C;
      d.cg = x.calc_grad();
      d.first_off = off;
      d.first_dev = w.dev;
      d.mem = w.dev_mem.get();
      d.dest_ok = d.cg && x.g->grad_dest && !x.is_grad_available();
      d.dest0 = x.g->grad_dest;
      d.dest_mem = x.g->grad_dest_mem;
      if (src == LS_RESIDENT) d.reuse = w.leaf_batch.lock();
    } else if (ok) {
      ok = d.kind == LK_LINEAR && d.src == src && d.M == s.M && d.C == s.C && d.cg == x.calc_grad();
      if (ok && src == LS_STAGED) ok = off == d.first_off + size_t(local) * bytes;
      if (ok && src == LS_RESIDENT)
        ok = w.dev == d.first_dev + size_t(local) * size_t(s.M) * size_t(s.C) && w.dev_mem.get() == d.mem;
      if (ok && d.dest_ok)
        d.dest_ok = x.g->grad_dest == d.dest0 + size_t(local) * size_t(s.M) * size_t(s.C) && !x.is_grad_available();
    }
    if (ok && src == LS_RESIDENT && d.reuse) {
      Batch& r = *d.reuse;
      const bool same = r.kind == Batch::LINEAR && r.leaf && size_t(local) < r.graphs.size() &&
                        r.graphs[size_t(local)].w == x.w && r.graphs[size_t(local)].s == x.s &&
                        r.graphs[size_t(local)].g == x.g && r.calc_grad == d.cg && w.leaf_version == w.version &&
                        w.leaf_batch.lock() == d.reuse;
      if (!same) d.reuse.reset();
    }
    if (ok) {
      kind = LK_LINEAR;
      // the same emissions as the previous linear leaf of this thread, at the same position?
      const bool al = sl.last_lin.w == &w && sl.last_lin.local == local && sl.last_lin.sg != &sg;
      if (local == 0) {
        d.alias = al ? sl.last_lin.sg : nullptr;
        d.alias_side = sl.last_lin.side;
      } else if (d.alias && !(al && sl.last_lin.sg == d.alias && sl.last_lin.side == d.alias_side)) {
        d.alias = nullptr;
      }
      sl.last_lin.w = &w;
      sl.last_lin.sg = &sg;
      sl.last_lin.side = k;
      sl.last_lin.local = local;
    }
  } else if (comp && x.w && x.g && s.kind == KIND_EXPLICIT && s.host_valid && !s.lazy && s.ctc_checked && s.ctc_labels &&
             s.N <= band_max_nodes() && x.w->known_all_zero()) {
    // (only what is cached already -- arcSort() checks the shape on the building thread, graph.cpp -- so that a
    //  target shared by several tasks is never written to from here)
    const bool cg = x.calc_grad();
    if (local == 0) {
      d.blank = s.ctc_blank;
      d.cg = cg;
    }
    if (d.blank == s.ctc_blank && d.cg == cg && (local == 0 || d.kind == LK_CTC)) {
      kind = LK_CTC;
      d.labels.insert(d.labels.end(), s.ctc_labels->begin(), s.ctc_labels->end());
      d.len.push_back(int(s.ctc_labels->size()));
    }
  }
  if (local == 0)
    d.kind = kind;
  else if (d.kind != kind)
    d.kind = LK_OTHER;
  d.graphs.push_back(x);
}

// input side k of call `local` of group sg is x: returns the depth x contributes
int note_input(SliceGroup& sg, int k, const Graph& x, int local, Pending& call) {
  uint8_t kind;
  int depth = 0;
  if (!is_placeholder(x)) {
    kind = IN_LEAF;
    if (sg.in_kind[k] == IN_NONE || sg.in_kind[k] == IN_LEAF) note_leaf(sg, k, x, local);
  } else {
    Pending& q = *x.s->pending;
    const int qs = q.st();
    if (qs == 0) {
      depth = q.depth;
      kind = IN_MIXED;
      if (q.sg && q.sg->slice == sg.slice && q.local == local && q.state.load(std::memory_order_acquire) == 0 &&
          (local == 0 || sg.prod[k] == q.sg)) {
        kind = IN_ALIGNED;
        sg.prod[k] = q.sg;
      }
    } else {
      kind = IN_MIXED;
      int e = -1;
      BatchP r = qs == 1 && !q.has_res.load(std::memory_order_acquire) ? q.record(&e) : nullptr;
      if (r && e >= 0 && (local == 0 || sg.done_rec[k] == r.get())) {
        kind = IN_DONE;
        if (local == 0) {
          sg.done_rec[k] = r.get();
          sg.done_hold[k] = r;
        }
        sg.done_idx[k].push_back(e);
        sg.done_identity[k] = sg.done_identity[k] && e == local;
      }
    }
  }
  (void)call;
  if (local == 0)
    sg.in_kind[k] = kind;
  else if (sg.in_kind[k] != kind)
    sg.in_kind[k] = IN_MIXED;
  return depth;
}

inline bool binary(RegionOp op) { return op == RO_ADD || op == RO_SUB || op == RO_COMPOSE || op == RO_INTERSECT; }

Graph record_call(RegionOp op, const Graph& a, const Graph* b) {
  Slice& sl = my_slice();
  PlaceholderStructure* ps = sl.new_placeholder();
  Pending& p = ps->call;
  ps->pending = &p;
  p.op = op;
  // (an input that is a placeholder of THIS slice is referred to without owning it: the slice owns both calls,
  //  and a counted reference to itself would keep it alive for ever)
  auto hold = [&sl](const Graph& x) {
    if (is_placeholder(x) && x.s->pending->sg && x.s->pending->sg->slice == &sl) {
      Graph r{Graph::Empty{}};
      r.s = std::shared_ptr<Structure>(std::shared_ptr<Structure>(), x.s.get());
      return r;
    }
    return x;
  };
  p.a = hold(a);
  if (b) p.b = hold(*b);
  count_use(a, +1);
  if (b) count_use(*b, +1);
  p.uses_counted = true;
  int d = 0;
  auto depth_of = [](const Graph& x) {
    if (!is_placeholder(x)) return 0;
    const Pending& q = *x.s->pending;
    return q.st() == 0 ? q.depth : 0;
  };
  d = std::max(depth_of(a), b ? depth_of(*b) : 0);
  p.depth = d + 1;
  if (op == RO_COMPOSE || op == RO_INTERSECT) {
    // the calling thread's compose mode goes with the call (gtn_amd.h gtnx_compose_mode: -1 the engine's own
    // policy, 0 built, 1 / 2 symbolic where possible); include/gtn/parallel.h hands the mode of the thread that
    // called parallelMap to the pool's threads
    const int hint = compose_mode_hint(0);
    compose_mode_hint(hint);
    p.mode = int8_t(hint);
  }
  SliceGroup& sg = sl.group(op, p.depth, p.mode);
  p.sg = &sg;
  p.local = int(sg.calls.size());
  note_input(sg, 0, a, p.local, p);
  if (b) note_input(sg, 1, *b, p.local, p);
  sg.calls.push_back(&p);
  ++sl.n_calls;
  Graph ph{Graph::Empty{}};
  ph.s = std::shared_ptr<Structure>(t_slice, ps);
  return ph;
}

// ---- weights handed over inside the region: one copy per staging block / one launch for device sources
thread_local std::vector<std::shared_ptr<PendingCopy>> t_pending_copies;  // of the join this thread is running
void apply_stage(std::vector<SliceP>& slices) {
  Runtime* rtp = nullptr;
  // device sources: one arena, slice after slice; the weights find their copy through their slice's block
  size_t dtotal = 0, nseg = 0;
  for (auto& sl : slices)
    if (!sl->executed && sl->blk && !sl->blk->base.load(std::memory_order_acquire)) {
      dtotal += sl->dev_bytes;
      nseg += sl->stage_dev.size();
    }
  if (nseg) {
    GTNX_HOST_T("region.apply_stage.device");
    rtp = &Runtime::get();
    DevMemP darena = rtp->alloc(dtotal ? dtotal : 16);
    auto pend = std::make_shared<PendingCopy>();
    pend->device = rtp->device();
    std::vector<CopySeg>& segs = pend->segs;
    segs.reserve(nseg);
    size_t off = 0;
    for (auto& sl : slices) {
      if (sl->executed || !sl->blk || sl->blk->base.load(std::memory_order_acquire)) continue;
      char* base = darena->as<char>(off);
      for (const Slice::DevSeg& sg : sl->stage_dev) {
        segs.push_back({base + sg.off, sg.src, int64_t(sg.bytes)});
        pend->max_bytes = std::max<int64_t>(pend->max_bytes, int64_t(sg.bytes));
      }
      sl->blk->mem = darena;
      sl->blk->pend = pend;
      sl->blk->base.store(reinterpret_cast<float*>(base), std::memory_order_release);
      off += sl->dev_bytes;  // (multiples of 16: the slices' blocks are back to back)
      sl->stage_dev.clear();
    }
    std::sort(segs.begin(), segs.end(), [](const CopySeg& a, const CopySeg& b) { return a.dst < b.dst; });
    // (not launched here: graph.h PendingCopy -- the band forward sweep of this join makes the copy on its way, or
    //  the join's end does)
    static const bool eager_copy = std::getenv("GTNX_NO_FUSED_COPY") != nullptr;
    if (eager_copy) pend->settle();
    t_pending_copies.push_back(pend);
  }
  // host sources: the device arena is the image of the used span of the pinned block
  bool any_host = false;
  for (auto& sl : slices) any_host = any_host || !sl->stage_host.empty();
  if (!any_host) return;
  GTNX_HOST_T("region.apply_stage.host");
  Runtime& rt = Runtime::get();
  struct Span {
    const char* lo;
    const char* hi;
    DevMemP dev;
  };
  std::unordered_map<PinnedMem*, Span> spans;
  for (auto& sl : slices)
    for (auto& w : sl->stage_host) {
      if (!w->staged || w->staged->on_device) continue;  // read (and settled) in the meantime, or overwritten
      const char* src = reinterpret_cast<const char*>(w->staged->src);
      const char* end = src + align_up(sizeof(float) * size_t(w->n), 16);
      auto it = spans.find(w->staged->chunk.get());
      if (it == spans.end())
        spans.emplace(w->staged->chunk.get(), Span{src, end, nullptr});
      else {
        it->second.lo = std::min(it->second.lo, src);
        it->second.hi = std::max(it->second.hi, end);
      }
    }
  for (auto& kv : spans) {
    Span& sp = kv.second;
    sp.dev = rt.alloc(size_t(sp.hi - sp.lo));
    rt.h2d(sp.dev->ptr, sp.lo, size_t(sp.hi - sp.lo));
  }
  for (auto& sl : slices) {
    for (auto& w : sl->stage_host) {
      if (!w->staged || w->staged->on_device) continue;
      const Span& sp = spans[w->staged->chunk.get()];
      w->dev_mem = sp.dev;
      w->dev = sp.dev->as<float>(size_t(reinterpret_cast<const char*>(w->staged->src) - sp.lo));
      w->dev_valid = true;
      w->host_valid = false;
      w->staged.reset();  // (the pinned block goes back to the pool behind the copy: stream order)
    }
    sl->stage_host.clear();
  }
}

// ---- one run --------------------------------------------------------------------------------------------
// a call's value as the batch functions see it
struct Val {
  BatchP batch;               // element `idx` of this record, or
  int idx = -1;
  const Graph* g = nullptr;   // an ordinary graph (the call's own input, or an earlier call's result)
  std::exception_ptr err;
};

void finish(Pending& p, int state) {
  p.release_inputs();
  if (p.sg) p.sg->individual.fetch_add(1, std::memory_order_acq_rel);
  p.state.store(state, std::memory_order_release);
}

void fail(Pending& p, std::exception_ptr e) {
  p.err = e;
  finish(p, 2);
}

void set_result(Pending& p, const BatchP& r, int idx) {
  p.batch = r;
  p.idx = idx;
  finish(p, 1);
}

Graph& result_graph(Pending& p) {
  if (!p.has_res.load(std::memory_order_acquire)) {
    std::lock_guard<std::recursive_mutex> lk(shared().exec);
    if (!p.has_res.load(std::memory_order_acquire)) {
      ExecScope es;
      int e = -1;
      BatchP r = p.record(&e);
      p.res = batch_get(r, e);
      p.has_res.store(true, std::memory_order_release);
    }
  }
  return p.res;
}

// the calls of one (depth, function, mode) of all slices of a run
struct MGroup {
  RegionOp op = RO_NEG;
  int depth = 0;
  int8_t mode = -1;
  bool postponed = false;  // forwardScore of plain linear chains: after the sweeps over the same chains (they
                           // leave it behind, batch.cpp: nc_norm), unless somebody needs it earlier
  bool ran = false;
  bool as_one = false;     // ran on the slice-level path: `result` holds part i at offset off[i]
  bool consumed = false;   // another group of the run read `result`
  std::vector<SliceGroup*> parts;
  int n = 0;
  BatchP result;
  BatchP leaf_rec[2];      // the leaf records made for its plain inputs (groups over the same leaves reuse them)
  std::vector<int> perm;   // element of `result` per call, when that is not the call's position
};

struct Run {
  std::vector<SliceP>& slices;
  std::deque<MGroup> groups;
  std::unordered_map<Weights*, BatchP> linear_of;      // first element's weights -> leaf LINEAR record
  std::unordered_map<Structure*, BatchP> targets_of;   // first element's structure -> leaf CTC_TARGETS record
  std::exception_ptr first_error;

  explicit Run(std::vector<SliceP>& s) : slices(s) {}

  void note_error(std::exception_ptr e) {
    if (!first_error) first_error = e;
  }

  MGroup* group_of(const SliceGroup* sg) {
    if (!sg || sg->run != this || sg->mgroup < 0) return nullptr;
    return &groups[size_t(sg->mgroup)];
  }

  // ---- the call-by-call path: values, records from values, one group's calls
  Val value_of(Graph& x) {
    Val v;
    if (!is_placeholder(x)) {
      v.g = &x;
      return v;
    }
    Pending* q = x.s->pending;
    if (q->st() == 0) {
      if (MGroup* g = group_of(q->sg)) {
        run_group(*g);
      }
      if (q->st() == 0) {  // queued by a thread that has not handed its calls in (or in an outer run): that one now
        std::vector<Pending*> one{q};
        run_calls(q->op, one, false);
      }
    }
    if (q->st() == 2) {
      v.err = q->err;
      return v;
    }
    if (q->has_res.load(std::memory_order_acquire)) {
      v.g = &q->res;
    } else {
      v.batch = q->record(&v.idx);
      if (!v.batch) {  // (cannot happen: done without a result)
        try {
          throw_logic("[region] a queued call finished without a result");
        } catch (...) {
          v.err = std::current_exception();
        }
      }
    }
    return v;
  }

  static Graph graph_of(Val& v) {
    if (v.g) return *v.g;
    return batch_get(v.batch, v.idx);
  }

  // all values are the elements of ONE record, each exactly once: that record and the element of each call
  static BatchP aligned(std::vector<Val>& vs, std::vector<int>& perm) {
    if (vs.empty() || !vs[0].batch) return nullptr;
    Batch* x = vs[0].batch.get();
    if (size_t(x->n) != vs.size()) return nullptr;
    std::vector<uint8_t> seen(vs.size(), 0);
    perm.resize(vs.size());
    for (size_t k = 0; k < vs.size(); ++k) {
      if (vs[k].batch.get() != x || vs[k].idx < 0 || vs[k].idx >= x->n || seen[size_t(vs[k].idx)]) return nullptr;
      seen[size_t(vs[k].idx)] = 1;
      perm[k] = vs[k].idx;
    }
    return vs[0].batch;
  }

  // the inputs of a group as ONE batch record + the element each call reads
  BatchP as_batch(std::vector<Val>& vs, std::vector<int>& perm, bool want_targets) {
    if (BatchP x = aligned(vs, perm)) return x;
    perm.resize(vs.size());
    for (size_t k = 0; k < vs.size(); ++k) perm[k] = int(k);
    bool all_graphs = true;
    for (auto& v : vs) all_graphs = all_graphs && v.g;
    std::vector<Graph> gs;
    gs.reserve(vs.size());
    for (auto& v : vs) gs.push_back(graph_of(v));
    if (all_graphs && vs.size() >= 2) {
      const Structure& s0 = *gs[0].s;
      if (s0.kind == KIND_LINEAR && !s0.lazy) {
        auto it = linear_of.find(gs[0].w.get());
        if (it != linear_of.end() && same_leaves(*it->second, gs)) return it->second;
        if (BatchP old = gs[0].w->leaf_batch.lock())  // made by an earlier call over the same graphs, unchanged since
          if (old->kind == Batch::LINEAR && same_leaves(*old, gs)) {
            linear_of[gs[0].w.get()] = old;
            return old;
          }
        if (BatchP b = batch_linear_from_graphs(gs)) {
          linear_of[gs[0].w.get()] = b;
          return b;
        }
      } else if (want_targets && s0.kind == KIND_EXPLICIT && s0.host_valid && !s0.lazy && s0.N <= band_max_nodes()) {
        auto it = targets_of.find(gs[0].s.get());
        if (it != targets_of.end() && same_leaves(*it->second, gs)) return it->second;
        if (BatchP old = gs[0].s->leaf_batch.lock())
          if (old->kind == Batch::CTC_TARGETS && same_leaves(*old, gs)) {
            targets_of[gs[0].s.get()] = old;
            return old;
          }
        if (BatchP b = batch_ctc_targets_from_graphs(gs)) {
          targets_of[gs[0].s.get()] = b;
          return b;
        }
      }
    }
    return batch_from_graphs(std::move(gs));
  }

  static bool same_leaves(const Batch& b, const std::vector<Graph>& gs) {
    if (size_t(b.n) != gs.size() || b.graphs.size() != gs.size()) return false;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (b.graphs[i].s != gs[i].s || b.graphs[i].w != gs[i].w || b.graphs[i].g != gs[i].g) return false;
      if (gs[i].calc_grad() != b.calc_grad) return false;
      // (a LINEAR record holds the weights' values: they must not have changed since; a target record holds
      //  labels only -- touch() forgets it when the structure changes -- and needs all-zero weights)
      if (b.kind == Batch::LINEAR && gs[i].w->leaf_version != gs[i].w->version) return false;
      if (b.kind == Batch::CTC_TARGETS && !gs[i].w->is_all_zero()) return false;
    }
    return true;
  }

  // the batch function of a group over whole records
  BatchP apply(RegionOp op, const BatchP& a, const BatchP& b, int mode) {
    switch (op) {
      case RO_NEG: return batch_scalar(SK_NEGATE, a, nullptr);
      case RO_ADD: return batch_scalar(SK_ADD, a, b);
      case RO_SUB: return batch_scalar(SK_SUBTRACT, a, b);
      case RO_COMPOSE:
      case RO_INTERSECT: {
        // run under the mode the calls were made with (gtnx_compose_mode; looking inside a symbolic product
        // still builds it)
        struct Mode {
          int old;
          explicit Mode(int m) : old(compose_mode_hint(m)) {}
          ~Mode() { compose_mode_hint(old); }
        } scope(mode);
        return batch_compose(a, b, op == RO_INTERSECT);
      }
      case RO_FS: return batch_shortest_distance(a, false);
      case RO_VS: return batch_shortest_distance(a, true);
      case RO_VP: return batch_viterbi_path(a);
      default: return nullptr;
    }
  }

  // compose / intersect: the mode the calls were made under (-1, the engine's own policy -- gtn_amd.h: symbolic
  // for small partners built on the host -- unless the caller set one); the process-wide override on top
  static int effective_mode(RegionOp op, int mode) {
    if (op != RO_COMPOSE && op != RO_INTERSECT) return 0;
    const char* env = std::getenv("GTNX_LAZY_COMPOSE");  // (read per call: ops.cpp)
    if (env && env[0] >= '0' && env[0] <= '2') mode = env[0] - '0';
    return mode;
  }
  // target records (always symbolic) only when the mode allows symbolic results
  static bool records_allowed(RegionOp op, int mode) {
    return (op == RO_COMPOSE || op == RO_INTERSECT) && mode != 0 && !std::getenv("GTNX_NO_BAND");
  }

  void run_calls(RegionOp op, std::vector<Pending*>& cs, bool retry_singly) {
    const size_t n = cs.size();
    std::vector<Pending*> live;
    std::vector<Val> la, lb;
    live.reserve(n);
    la.reserve(n);
    if (binary(op)) lb.reserve(n);
    {
      GTNX_HOST_T("region.run_calls.resolve");
      for (size_t k = 0; k < n; ++k) {
        // (the records were written by the pool's threads: every one of them is a miss in this core's caches --
        //  fetch the record eight calls ahead and what its inputs point to four calls ahead)
        if (k + 8 < n) __builtin_prefetch(cs[k + 8], 1);
        if (k + 4 < n) {
          __builtin_prefetch(cs[k + 4]->a.s.get());
          if (binary(op)) __builtin_prefetch(cs[k + 4]->b.s.get());
        }
        Pending& p = *cs[k];
        if (p.st() != 0) continue;  // ran already (somebody looked at it, or a callback forced it mid-run)
        Val a = value_of(p.a), b;
        if (p.st() != 0) continue;  // (resolving the input ran the call itself: a forced closure)
        if (binary(op)) b = value_of(p.b);
        if (p.st() != 0) continue;
        if (a.err || b.err) {  // an input failed: so does this call, with the same error
          fail(p, a.err ? a.err : b.err);
          continue;
        }
        live.push_back(&p);
        la.push_back(std::move(a));
        if (binary(op)) lb.push_back(std::move(b));
      }
    }
    if (live.empty()) return;
    try {
      if (op == RO_BWD || op == RO_BWD_RETAIN) {
        run_backward(live, la, op == RO_BWD_RETAIN);
        return;
      }
      std::vector<int> pa, pb;
      const int mode = effective_mode(op, live[0]->mode);
      const bool records = records_allowed(op, mode);
      BatchP A = as_batch(la, pa, records), B;
      if (binary(op)) {
        B = as_batch(lb, pb, records);
        if (pa != pb) {  // the two sides are elements of records in different orders: line them up as graphs
          std::vector<Graph> ga, gb;
          for (auto& v : la) ga.push_back(graph_of(v));
          for (auto& v : lb) gb.push_back(graph_of(v));
          A = batch_from_graphs(std::move(ga));
          B = batch_from_graphs(std::move(gb));
          for (size_t k = 0; k < pa.size(); ++k) pa[k] = int(k);
        }
      }
      BatchP R = apply(op, A, B, mode);
      GTNX_HOST_T("region.run_calls.results");
      for (size_t k = 0; k < live.size(); ++k) {
        if (k + 8 < live.size()) __builtin_prefetch(live[k + 8], 1);
        set_result(*live[k], R, pa[k]);
      }
    } catch (...) {
      if (!retry_singly || live.size() == 1) {
        const std::exception_ptr e = std::current_exception();
        note_error(e);
        for (Pending* p : live)
          if (p->st() == 0) fail(*p, e);
        return;
      }
      // one by one (these functions do not change their inputs): every call gets its own result or error
      for (size_t k = 0; k < live.size(); ++k) {
        if (live[k]->st() != 0) continue;
        std::vector<Pending*> one{live[k]};
        run_calls(op, one, false);
      }
    }
  }

  void run_backward(std::vector<Pending*>& live, std::vector<Val>& roots, bool retain) {
    std::vector<int> perm;
    if (BatchP x = aligned(roots, perm)) {
      batch_backward(x, retain);  // (throws before it changes anything: batch.cpp)
      for (Pending* p : live) finish(*p, 1);
      return;
    }
    // per-graph tapes: a root that cannot run (backward twice without retain) fails alone, the others go
    // through ONE sweep -- which either happens for all of them or, if it throws midway, is reported to all
    std::vector<Graph> gs;
    std::vector<Pending*> ok;
    for (size_t k = 0; k < live.size(); ++k) {
      try {
        Graph g = graph_of(roots[k]);
        realize(g);
        backward_validate(g);
        gs.push_back(std::move(g));
        ok.push_back(live[k]);
      } catch (...) {
        note_error(std::current_exception());
        fail(*live[k], std::current_exception());
      }
    }
    if (gs.empty()) return;
    try {
      op_backward(gs, nullptr, retain);
      for (Pending* p : ok) finish(*p, 1);
    } catch (...) {
      note_error(std::current_exception());
      for (Pending* p : ok) fail(*p, std::current_exception());
    }
  }

  // ---- the slice-level path ------------------------------------------------------------------------------
  // the leaf record of input side k of group g, from the slices' digests (null: not that simple)
  BatchP leaf_record(MGroup& g, int k, bool records) {
    if (g.leaf_rec[k]) return g.leaf_rec[k];
    const LeafDigest& d0 = g.parts[0]->leaf[k];
    if (d0.kind == LK_CTC) {
      if (!records) return nullptr;
      GTNX_HOST_T("region.leaf.ctc_targets");
      size_t total = 0;
      for (SliceGroup* sg : g.parts) {
        const LeafDigest& d = sg->leaf[k];
        if (d.kind != LK_CTC || d.blank != d0.blank || d.cg != d0.cg) return nullptr;
        total += d.labels.size();
      }
      std::vector<int> flat, len;
      flat.reserve(total);
      len.reserve(size_t(g.n));
      for (SliceGroup* sg : g.parts) {
        const LeafDigest& d = sg->leaf[k];
        flat.insert(flat.end(), d.labels.begin(), d.labels.end());
        len.insert(len.end(), d.len.begin(), d.len.end());
      }
      BatchP b = batch_ctc_targets(flat.data(), len.data(), g.n, d0.blank, d0.cg);
      if (b->kind != Batch::CTC_TARGETS) return nullptr;  // (labels the records cannot hold: the per-graph way)
      take_graphs(*b, g, k);
      g.leaf_rec[k] = b;
      return b;
    }
    if (d0.kind != LK_LINEAR) return nullptr;
    // one record for two groups over the same emissions
    for (SliceGroup* sg : g.parts) {
      const LeafDigest& d = sg->leaf[k];
      if (d.kind != LK_LINEAR) return nullptr;
    }
    if (d0.alias) {
      MGroup* other = group_of(d0.alias);
      bool same = other && other != &g && other->parts.size() == g.parts.size();
      for (size_t i = 0; same && i < g.parts.size(); ++i) {
        const LeafDigest& d = g.parts[i]->leaf[k];
        same = d.alias == other->parts[i] && d.alias_side == d0.alias_side &&
               other->parts[i]->calls.size() == g.parts[i]->calls.size();
      }
      if (same && other->leaf_rec[d0.alias_side]) {
        g.leaf_rec[k] = other->leaf_rec[d0.alias_side];
        return g.leaf_rec[k];
      }
      if (same) {  // we are first: the other group finds the record here
        if (BatchP b = linear_record(g, k)) {
          other->leaf_rec[d0.alias_side] = b;
          return b;
        }
        return nullptr;
      }
    }
    return linear_record(g, k);
  }

  void take_graphs(Batch& b, MGroup& g, int k) {
    b.graphs.reserve(size_t(g.n));
    b.give_back = &give_back_graphs;
    for (SliceGroup* sg : g.parts) {
      LeafDigest& d = sg->leaf[k];
      const size_t at = b.graphs.size();
      for (Graph& x : d.graphs) b.graphs.push_back(std::move(x));
      d.graphs.clear();
      if (sg->slice->home) b.origins.push_back({sg->slice->home, at, b.graphs.size()});  // (they go home with the record)
    }
    b.leaf = true;
  }

  BatchP linear_record(MGroup& g, int k) {
    GTNX_HOST_T("region.leaf.linear");
    const LeafDigest& d0 = g.parts[0]->leaf[k];
    if (g.parts.size() == 1 && d0.reuse && d0.reuse->n == g.n) {  // made by an earlier call over the same graphs
      g.leaf_rec[k] = d0.reuse;
      return d0.reuse;
    }
    const size_t A = size_t(d0.M) * size_t(d0.C);
    const float* expect = nullptr;
    DevMemP mem;
    std::shared_ptr<PendingCopy> pend;  // staged weights: the copy into `mem` may still be owed (graph.h)
    for (SliceGroup* sg : g.parts) {
      const LeafDigest& d = sg->leaf[k];
      if (d.src != d0.src || d.M != d0.M || d.C != d0.C || d.cg != d0.cg) return nullptr;
      const float* first;
      if (d.src == LS_STAGED) {
        StageBlock& blk = *sg->slice->blk;
        float* base = blk.base.load(std::memory_order_acquire);
        if (!base) return nullptr;  // (not copied: cannot happen after apply_stage)
        first = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + d.first_off);
        if (!mem) {
          mem = blk.mem;
          pend = blk.pend;
        }
        if (blk.mem != mem) return nullptr;
      } else {
        first = d.first_dev;
        if (!mem && !d.graphs.empty()) mem = d.graphs[0].w->dev_mem;
        if (d.mem != mem.get()) return nullptr;
      }
      if (expect && first != expect) return nullptr;  // the slices' blocks are not back to back in this order
      expect = first + sg->calls.size() * A;
    }
    // gtnx_grads_bind_device_n on the graphs: their first gradients go straight to the caller's tensor when
    // that is one block in element order
    float* dest = d0.dest_ok ? d0.dest0 : nullptr;
    size_t done = 0;
    for (SliceGroup* sg : g.parts) {
      const LeafDigest& d = sg->leaf[k];
      if (!dest) break;
      if (!d.dest_ok || d.dest0 != d0.dest0 + done * A) dest = nullptr;
      done += sg->calls.size();
    }
    const LeafDigest& f = g.parts[0]->leaf[k];
    const float* w0 = f.src == LS_STAGED
                          ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.parts[0]->slice->blk->base.load()) + f.first_off)
                          : f.first_dev;
    BatchP b = std::make_shared<Batch>();
    b->kind = Batch::LINEAR;
    b->n = g.n;
    b->calc_grad = d0.cg;
    b->M = d0.M;
    b->C = d0.C;
    b->w_mem = mem;
    b->w_dev = const_cast<float*>(w0);
    b->w_pend = pend;
    if (dest) {
      b->dest_mem = d0.dest_mem;
      b->dest = dest;
    }
    const bool resident = d0.src == LS_RESIDENT;
    take_graphs(*b, g, k);
    if (resident)  // a later call over the same graphs finds the record (and what a sweep left behind in it) again
      for (Graph& x : b->graphs) {
        x.w->leaf_batch = b;
        x.w->leaf_version = x.w->version;
      }
    g.leaf_rec[k] = b;
    return b;
  }

  // input side k of group g as one record, every call reading the element at its own position (perm: or not)
  BatchP side_record(MGroup& g, int k, bool records, const std::vector<int>** perm) {
    *perm = nullptr;
    const uint8_t kind = g.parts[0]->in_kind[k];
    for (SliceGroup* sg : g.parts)
      if (sg->in_kind[k] != kind) return nullptr;
    switch (kind) {
      case IN_LEAF: return leaf_record(g, k, records);
      case IN_ALIGNED: {
        MGroup* p = group_of(g.parts[0]->prod[k]);
        if (!p || p->parts.size() != g.parts.size()) return nullptr;
        for (size_t i = 0; i < g.parts.size(); ++i)
          if (g.parts[i]->prod[k] != p->parts[i] || p->parts[i]->calls.size() != g.parts[i]->calls.size()) return nullptr;
        run_group(*p);
        if (!p->as_one || !p->result) return nullptr;
        for (SliceGroup* sg : p->parts)
          if (sg->individual.load(std::memory_order_acquire) != 0) return nullptr;
        if (!p->perm.empty()) *perm = &p->perm;
        p->consumed = true;
        return p->result;
      }
      case IN_DONE: {
        Batch* r = g.parts[0]->done_rec[k];
        if (!r || r->n != g.n) return nullptr;
        bool identity = g.parts.size() == 1 && g.parts[0]->done_identity[k];
        for (SliceGroup* sg : g.parts)
          if (sg->done_rec[k] != r) return nullptr;
        if (!identity) {  // every element exactly once?
          std::vector<int>& pm = g.perm;
          std::vector<int> tmp;
          tmp.reserve(size_t(g.n));
          for (SliceGroup* sg : g.parts) tmp.insert(tmp.end(), sg->done_idx[k].begin(), sg->done_idx[k].end());
          std::vector<uint8_t> seen(size_t(g.n), 0);
          for (int e : tmp) {
            if (e < 0 || e >= g.n || seen[size_t(e)]) return nullptr;
            seen[size_t(e)] = 1;
          }
          if (!pm.empty() && pm != tmp) return nullptr;  // (the two sides of a binary call disagree)
          pm = std::move(tmp);
          *perm = &pm;
        }
        return g.parts[0]->done_hold[k];
      }
      default: return nullptr;
    }
  }

  // the whole group as ONE call over records; false: flatten it
  bool run_as_one(MGroup& g) {
    for (SliceGroup* sg : g.parts)
      if (sg->individual.load(std::memory_order_acquire) != 0) return false;
    const bool bwd = g.op == RO_BWD || g.op == RO_BWD_RETAIN;
    const int mode = effective_mode(g.op, g.mode);
    const bool records = records_allowed(g.op, mode);
    const std::vector<int>*pa = nullptr, *pb = nullptr;
    BatchP A = side_record(g, 0, records, &pa), B;
    if (!A) return false;
    if (bwd) {
      // every element of ONE record exactly once (in any order): its backward
      if (A->n != g.n) return false;
      batch_backward(A, g.op == RO_BWD_RETAIN);  // (throws before it changes anything: batch.cpp)
      publish(g, nullptr);
      return true;
    }
    if (binary(g.op)) {
      B = side_record(g, 1, records, &pb);
      if (!B) return false;
      const bool ia = !pa, ib = !pb;
      if (ia != ib || (pa && pb && *pa != *pb)) return false;  // elements in different orders: line them up as graphs
    }
    if (A->n != g.n || (B && B->n != g.n)) return false;
    BatchP R = apply(g.op, A, B, mode);
    if (!R || R->n != g.n) return false;
    if (pa && g.perm.empty()) g.perm = *pa;
    publish(g, R);
    return true;
  }

  void publish(MGroup& g, const BatchP& R) {
    g.result = R;
    g.as_one = true;
    int off = 0;
    for (SliceGroup* sg : g.parts) {
      sg->result = R;
      sg->base = off;
      if (!g.perm.empty())  // (the group's own copy: the slice outlives the run)
        sg->perm.assign(g.perm.begin() + off, g.perm.begin() + off + int(sg->calls.size()));
      off += int(sg->calls.size());
      sg->state.store(1, std::memory_order_release);
    }
  }

  void run_group(MGroup& g) {
    if (g.ran) return;
    g.ran = true;
    static const char* names[RO_COUNT] = {"region.negate", "region.add", "region.subtract", "region.compose", "region.intersect",
                                          "region.forward_score", "region.viterbi_score", "region.viterbi_path",
                                          "region.backward", "region.backward_retain"};
    GTNX_HOST_T(names[g.op]);
    static const bool no_fast = std::getenv("GTNX_REGION_NO_SLICE_PATH") != nullptr;
    if (!no_fast) {
      try {
        if (run_as_one(g)) return;
      } catch (...) {
        // one member's failure must stay that member's: call by call below (these functions do not change
        // their inputs, and a batch backward throws before it changes anything)
      }
    }
    GTNX_HOST_T("region.flattened");
    std::vector<Pending*> cs;
    cs.reserve(size_t(g.n));
    for (SliceGroup* sg : g.parts)
      for (Pending* p : sg->calls)
        if (p->st() == 0) cs.push_back(p);
    run_calls(g.op, cs, true);
  }

  void run_all() {
    // merge the slices' groups by (depth, function, mode); backward calls last, in one group per retain flag
    for (auto& sl : slices) {
      if (sl->executed) continue;
      for (SliceGroup& sg : sl->groups) {
        if (sg.state.load(std::memory_order_acquire) != 0 || sg.run) continue;
        const bool bwd = sg.op == RO_BWD || sg.op == RO_BWD_RETAIN;
        int idx = -1;
        for (size_t i = 0; i < groups.size(); ++i)
          if (groups[i].op == sg.op && groups[i].mode == sg.mode && (bwd || groups[i].depth == sg.depth)) {
            idx = int(i);
            break;
          }
        if (idx < 0) {
          idx = int(groups.size());
          groups.emplace_back();
          groups.back().op = sg.op;
          groups.back().depth = bwd ? 0x7fffffff : sg.depth;
          groups.back().mode = sg.mode;
        }
        sg.run = this;
        sg.mgroup = idx;
        groups[size_t(idx)].parts.push_back(&sg);
        groups[size_t(idx)].n += int(sg.calls.size());
      }
    }
    for (auto& g : groups) {
      if (g.op != RO_FS) continue;
      g.postponed = true;
      for (SliceGroup* sg : g.parts)
        if (sg->in_kind[0] != IN_LEAF || sg->leaf[0].kind != LK_LINEAR) {
          bool linear = sg->in_kind[0] == IN_LEAF;
          for (const Graph& x : sg->leaf[0].graphs) linear = linear && x.s->kind == KIND_LINEAR;
          if (!linear) {
            g.postponed = false;
            break;
          }
        }
    }
    std::vector<int> order(groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = int(i);
    auto prio = [](RegionOp op) { return (op == RO_COMPOSE || op == RO_INTERSECT) ? 0 : 1; };
    std::sort(order.begin(), order.end(), [&](int x, int y) {
      const MGroup &a = groups[size_t(x)], &b = groups[size_t(y)];
      const bool ba = a.depth == 0x7fffffff, bb = b.depth == 0x7fffffff;
      if (ba != bb) return bb;
      if (a.postponed != b.postponed) return b.postponed;
      if (a.depth != b.depth) return a.depth < b.depth;
      if (prio(a.op) != prio(b.op)) return prio(a.op) < prio(b.op);
      return int(a.op) < int(b.op);
    });
    // (does this run end in a function that waits for the GPU?  Then the step's reclamation moves to that wait.)
    bool waits = false;
    for (auto& g : groups) waits = waits || g.op == RO_VS || g.op == RO_VP;
    struct ReclaimScope {
      bool prev;
      explicit ReclaimScope(bool w) : prev(t_reclaim_at_wait) { t_reclaim_at_wait = w; }
      ~ReclaimScope() { t_reclaim_at_wait = prev; }
    } reclaim_scope(waits);
    for (int gi : order) run_group(groups[size_t(gi)]);
    // scalars nobody in this run reads are what the caller will ask for (the losses of a step): their values start
    // for the host now, behind the launches that compute them -- item() then waits for THEM, not for what the
    // caller queues next (the step's backward)
    static const bool no_prefetch = std::getenv("GTNX_NO_ITEM_PREFETCH") != nullptr;
    if (!no_prefetch)
      for (auto& g : groups)
        if (g.as_one && !g.consumed && g.result && g.result->kind == Batch::SCALAR) {
          try {
            batch_prefetch_items(g.result);
          } catch (...) {
          }
        }
    for (auto& sl : slices) {
      for (SliceGroup& sg : sl->groups) {
        if (sg.run == this) {
          sg.run = nullptr;
          sg.mgroup = -1;
        }
      }
      sl->executed = true;
    }
  }
};

// runs the slices (and what their calls depend on); returns the first error of the run
std::exception_ptr execute(std::vector<SliceP>& slices) {
  std::lock_guard<std::recursive_mutex> lk(shared().exec);
  GTNX_HOST_T("region.execute");
  ExecScope es;
  std::exception_ptr err;
  // (a nested join -- a placeholder looked at from inside a batched call -- has copies of its own)
  std::vector<std::shared_ptr<PendingCopy>> outer;
  outer.swap(t_pending_copies);
  try {
    apply_stage(slices);
    Run run(slices);
    run.run_all();
    err = run.first_error;
  } catch (...) {
    err = std::current_exception();
    for (auto& sl : slices)
      for (SliceGroup& sg : sl->groups)
        for (Pending* p : sg.calls)
          if (p->st() == 0) fail(*p, err);
  }
  // the caller's buffers are promised until parallelMap returns: a copy no sweep of this join made is made now
  try {
    for (auto& pc : t_pending_copies) pc->settle();
  } catch (...) {
    if (!err) err = std::current_exception();
  }
  t_pending_copies.swap(outer);
  // (the calls keep their inputs until their slice dies -- with the last result handle of the thread that
  //  recorded it -- and goes home: nothing is taken apart here)
  slices.clear();
  return err;
}

void take_shared(std::vector<SliceP>& q) {
  Shared& sh = shared();
  sh.take_all(q);
  std::lock_guard<std::mutex> lk(sh.mu);
  if (sh.arena) {  // the next region starts a fresh block, sized by what this one used
    sh.arena_hint = std::max<size_t>(size_t(16) << 20, std::min(sh.arena_hint, 2 * sh.arena->used.load()));
    sh.arena.reset();
  }
}

void take_mine(std::vector<SliceP>& q) {
  if (t_slice && !t_slice->empty()) q.push_back(std::move(t_slice));
  t_slice.reset();
}

}  // namespace

bool region_active() { return t_depth > 0 && t_exec == 0; }

namespace {
std::atomic<int64_t> g_first_enter_us{0};  // GTNX_HOST_TIMING: when the current region's first thread arrived
int64_t now_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

void region_enter() {
  if (HostTimer::enabled() && t_depth == 0) {
    int64_t zero = 0;
    g_first_enter_us.compare_exchange_strong(zero, now_us());
  }
  // what earlier steps let go of is taken apart by the region's threads together (a batch's graphs were built
  // by such threads too), not by the one thread that joins them
  // (a handful of them: the allocator's locks are what more threads would wait on)
  // (a thread that never reclaims empties its list while it waits for the device -- Runtime::drain_while_busy --
  //  and here at the latest, once a few steps' worth has piled up)
  if (t_depth == 0 && !t_reclaims && Runtime::deferred_count() >= 4096) Runtime::drain_deferred();
  if (t_depth++ == 0 && Runtime::initialized()) {
    Shared& sh = shared();
    static const int max_drainers = [] {
      const char* e = std::getenv("GTNX_DRAIN_THREADS");
      return e ? std::atoi(e) : 0;
    }();
    if (sh.draining.fetch_add(1) < max_drainers) Runtime::get().drain_deferred();
    sh.draining.fetch_sub(1);
  }
}

void region_leave() {
  if (t_depth <= 0 || --t_depth > 0) return;
  if (t_slice && !t_slice->empty()) shared().hand_in(std::move(t_slice));
  t_slice.reset();
  t_arena.reset();
  // handles the tasks dropped: allocated on this thread, and their graphs are held by the recorded calls anyway
  for (Graph* g : t_trash) destroy_handle(g);
  t_trash.clear();
}

void region_reclaim_thread() {
  t_reclaims = true;
  Runtime::drain_deferred();
}

void region_flush() {
  if (HostTimer::enabled() && t_depth == 0) {
    const int64_t t0 = g_first_enter_us.exchange(0);
    if (t0) host_timer_add("region.pool_phase(first enter -> join)", double(now_us() - t0) * 1e-3);
  }
  std::vector<SliceP> q;
  std::exception_ptr stored;
  if (t_depth == 0) {
    take_shared(q);
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    stored = sh.first_error;
    sh.first_error = nullptr;
  }
  // (inside an enclosing region -- a nested parallelMap -- only the caller's own calls: they are what it joins)
  take_mine(q);
  std::exception_ptr err;
  if (!q.empty()) err = execute(q);
  if (stored) std::rethrow_exception(stored);
  if (err) std::rethrow_exception(err);
}

Graph region_record(RegionOp op, const Graph& a, const Graph* b) { return record_call(op, a, b); }

namespace {
// the calls of one vector form: recorded in this thread's slice (whatever was there before stays in front of
// them, in order) and joined at once
struct VectorCall {
  int depth0;
  bool exec0;
  VectorCall() : depth0(t_depth), exec0(t_vector_call) {
    ++t_depth;  // (record even when the thread is not inside a parallelMap region)
    t_vector_call = true;
  }
  ~VectorCall() {
    t_vector_call = exec0;
    t_depth = depth0;
  }
};
void join_now() {
  std::vector<SliceP> q;
  take_mine(q);
  if (q.empty()) return;
  std::exception_ptr err = execute(q);
  if (err) std::rethrow_exception(err);
}
}  // namespace

void region_run_vector(RegionOp op, Graph* const* a, int na, Graph* const* b, int nb, Graph* out) {
  const int n = b ? std::max(na, nb) : na;
  if ((na != n && na != 1) || (b && nb != n && nb != 1))  // parallel_map.h:85-88
    throw_runtime("parallelMap getIdxOrBroadcast got invalid size or unbroadcastable vector");
  {
    VectorCall scope;
    for (int i = 0; i < n; ++i) out[i] = record_call(op, *a[na == 1 ? 0 : i], b ? b[nb == 1 ? 0 : i] : nullptr);
  }
  join_now();
}

void region_run_backward_vector(Graph* const* roots, int n, bool retain) {
  {
    VectorCall scope;
    for (int i = 0; i < n; ++i) region_record_backward(*roots[i], retain);
  }
  join_now();
}

void region_record_backward(const Graph& root, bool retain) {
  Slice& sl = my_slice();
  sl.plain.emplace_back();
  Pending& p = sl.plain.back();
  p.op = retain ? RO_BWD_RETAIN : RO_BWD;
  if (is_placeholder(root) && root.s->pending->sg && root.s->pending->sg->slice == &sl)  // (as in record_call)
    p.a.s = std::shared_ptr<Structure>(std::shared_ptr<Structure>(), root.s.get());
  else
    p.a = root;
  count_use(root, +1);
  p.uses_counted = true;
  p.depth = 0x7fffffff;
  SliceGroup& sg = sl.group(p.op, p.depth, -1);
  p.sg = &sg;
  p.local = int(sg.calls.size());
  note_input(sg, 0, root, p.local, p);
  sg.calls.push_back(&p);
  ++sl.n_calls;
}

namespace {
// somebody looks at a result before the join: run it (and what it needs) now.  The caller's own slice goes
// along -- program order for anything it recorded earlier, and the batch stays a batch when the whole slice is
// one thread's.
void force(Pending* p) {
  // (a result lives on the device of the thread that asked for it: capi.cpp GL says the same of ordinary graphs)
  if (p->sg && p->sg->slice->device != Runtime::current_device())
    throw_invalid("[gtn_amd] this graph lives on device " + std::to_string(p->sg->slice->device) +
                  ", the calling thread is on device " + std::to_string(Runtime::current_device()) + " (gtnx_set_device)");
  if (p->st() == 0) {
    std::vector<SliceP> q;
    const bool mine = t_slice && p->sg && p->sg->slice == t_slice.get();
    take_mine(q);
    if (!mine) take_shared(q);  // handed in by its thread already?
    std::exception_ptr err;
    if (!q.empty()) err = execute(q);
    if (p->st() == 0) {  // still with the thread that recorded it: that one call (and what it needs) alone
      std::lock_guard<std::recursive_mutex> lk(shared().exec);
      ExecScope es;
      std::vector<SliceP> none;
      Run run(none);
      std::vector<Pending*> one{p};
      run.run_calls(p->op, one, false);
      if (!err) err = run.first_error;
    }
    if (err && p->st() != 2) {
      Shared& sh = shared();
      std::lock_guard<std::mutex> lk(sh.mu);
      if (!sh.first_error) sh.first_error = err;  // reported by the region's join
    }
  }
  if (p->st() == 2) std::rethrow_exception(p->err);
}
}  // namespace

Graph& region_value(Graph& ph) {
  std::shared_ptr<Structure> keep = ph.s;  // (the slice, through the aliasing handle)
  Pending* p = ph.s->pending;
  force(p);
  return result_graph(*p);
}

bool region_item(Graph& ph, float* out) {
  std::shared_ptr<Structure> keep = ph.s;
  Pending* p = ph.s->pending;
  force(p);
  if (p->has_res.load(std::memory_order_acquire)) return false;
  int e = -1;
  BatchP x = p->record(&e);
  if (!x || x->kind != Batch::SCALAR || x->materialised) return false;
  std::lock_guard<std::recursive_mutex> lk(shared().exec);
  *out = batch_item_host(x, e);
  return true;
}

bool region_items_device(Graph* const* hs, int n, void* dev_out) {
  if (n <= 0) return true;
  bool any = false;
  for (int i = 0; i < n && !any; ++i) any = is_placeholder(*hs[i]);
  if (!any) return false;
  std::vector<const float*> ptrs(static_cast<size_t>(n));
  Batch* first = nullptr;
  bool dense = true;  // element i of one record at position i: a plain copy
  for (int i = 0; i < n; ++i) {
    Graph& h = *hs[i];
    if (!is_placeholder(h)) return false;
    std::shared_ptr<Structure> keep = h.s;
    Pending* p = h.s->pending;
    force(p);
    if (p->has_res.load(std::memory_order_acquire)) return false;
    int e = -1;
    BatchP xp = p->record(&e);
    Batch* x = xp.get();
    if (!x || x->kind != Batch::SCALAR || x->materialised) return false;
    ptrs[size_t(i)] = x->v_dev + e;
    if (i == 0) first = x;
    dense = dense && x == first && e == i;
  }
  Runtime& rt = Runtime::get();
  if (dense) {
    rt.d2d(dev_out, first->v_dev, sizeof(float) * size_t(n));
    return true;
  }
  DevMemP dp = upload_vec(ptrs);
  launch_gather_scalars(dp->as<const float*>(), static_cast<float*>(dev_out), n, rt.stream());
  return true;
}

void region_sync_thread() {
  if (!t_slice || t_slice->empty()) return;
  std::vector<SliceP> q;
  take_mine(q);
  std::exception_ptr err = execute(q);
  if (err) {
    Shared& sh = shared();
    std::lock_guard<std::mutex> lk(sh.mu);
    if (!sh.first_error) sh.first_error = err;
  }
}

void region_before_mutation(Graph& g) {
  if (!g.s || is_placeholder(g)) return;
  if (g.s->pending_uses.load() == 0 && (!g.w || g.w->pending_uses.load() == 0)) return;
  region_sync_thread();
  if (g.s->pending_uses.load() == 0 && (!g.w || g.w->pending_uses.load() == 0)) return;
  // queued by other threads (a graph shared across the region's tasks), or calls that ran already and have not
  // let go of their inputs yet: what was handed in runs
  std::vector<SliceP> q;
  take_shared(q);
  if (!q.empty()) {
    std::exception_ptr err = execute(q);
    if (err) {
      Shared& sh = shared();
      std::lock_guard<std::mutex> lk(sh.mu);
      if (!sh.first_error) sh.first_error = err;
    }
  }
  // (a count that is still raised belongs to calls that have RUN: they give it back when their slice dies)
}

bool region_stage_weights(Graph& g, const float* p, bool device) {
  if (!region_active() || is_placeholder(g)) return false;
  Weights& w = *g.w;
  const int64_t n = g.s->A;
  if (n < 256 || w.host_escaped || g.w.use_count() > 2) return false;  // small, or aliased: the ordinary way
  auto st = std::make_shared<StagedWeights>();
  st->on_device = device;
  Slice& sl = my_slice();
  if (device) {
    // read at the join (gtn_amd.h: the buffer stays valid and unchanged until the parallelMap call returns);
    // GTNX_REGION_EAGER_WEIGHTS=1: copied at the call like graph.cpp:179-181, one launch per call
    static const bool eager = std::getenv("GTNX_REGION_EAGER_WEIGHTS") != nullptr;
    if (eager) return false;
    if (!sl.blk || sl.blk->base.load(std::memory_order_acquire)) {
      sl.blk = std::make_shared<StageBlock>();
      sl.dev_bytes = 0;
    }
    const size_t bytes = sizeof(float) * size_t(n);
    st->src = p;
    st->blk = sl.blk;
    st->off = sl.dev_bytes;
    sl.stage_dev.push_back({p, sl.dev_bytes, bytes});
    sl.dev_bytes += align_up(bytes, 16);
  } else {
    const size_t bytes = align_up(sizeof(float) * size_t(n), 16);
    float* dst = nullptr;
    for (;;) {
      if (!t_arena) {
        Shared& sh = shared();
        std::lock_guard<std::mutex> lk(sh.mu);
        if (!sh.arena) {
          auto a = std::make_shared<StageArena>();
          a->cap = std::max(sh.arena_hint, bytes);
          a->mem = Runtime::get().alloc_pinned(a->cap);
          a->cap = a->mem->bytes;
          sh.arena = std::move(a);
        }
        t_arena = sh.arena;
      }
      const size_t off = t_arena->used.fetch_add(bytes);
      if (off + bytes <= t_arena->cap) {
        dst = t_arena->mem->as<float>(off);
        break;
      }
      // full: the next block is twice as large (each thread that finds it full asks once)
      Shared& sh = shared();
      std::lock_guard<std::mutex> lk(sh.mu);
      if (sh.arena == t_arena) {
        sh.arena_hint = std::max(sh.arena_hint, 2 * t_arena->cap);
        sh.arena.reset();
      }
      t_arena.reset();
    }
    std::memcpy(dst, p, sizeof(float) * size_t(n));
    st->chunk = t_arena->mem;
    st->src = dst;
    sl.stage_host.push_back(g.w);
  }
  w.n = n;
  w.host.clear();
  w.host_valid = false;
  w.dev_valid = false;
  w.zero = false;
  w.staged = std::move(st);
  w.version++;
  return true;
}

void region_trash(Graph* handle) { t_trash.push_back(handle); }

}  // namespace gtnx
