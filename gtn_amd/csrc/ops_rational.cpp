// ops_rational.cpp -- rational operations, remove and the binary format as device builders (rational.hip); see ops.h
#include "ops_internal.h"

namespace gtnx {

// ======================================================================
// rational operations (functions.cpp:66-223), built on the device: rational.hip
// ======================================================================
struct RationalOp : OpRecord {
  // per output: where each input's arcs start in the output's arc order (functions.cpp:98-110, 159-164, 191-200:
  // the gradient of an input is a slice of the deltas)
  std::vector<std::vector<int64_t>> arc_off;
  void backward(std::vector<Member>& ms) override {
    // addGrad COPIES the slice (graph.cpp:91-129): an input's gradient must not alias the output's buffer -- a
    // retained tape run twice, concat({g, g}) or a later accumulation into the input would otherwise write into
    // the output's gradient too.  One arena for the record's slices, one batched copy, then the sink adopts it.
    Runtime& rt = Runtime::get();
    GradSink sink;
    std::vector<CopySeg> segs;
    struct Item {
      Graph* in;
      size_t off;
    };
    std::vector<Item> items;
    size_t total = 0;
    int64_t longest = 0;
    for (auto& m : ms) {
      Graph& gr = m.out.grad();
      if (!gr.w->dev_valid || gr.w->host_escaped) {
        std::vector<Weights*> v{gr.w.get()};
        ensure_weights_device_batch(v);
      }
      auto& ins = m.out.g->inputs;
      for (size_t i = 0; i < ins.size(); ++i) {
        if (!ins[i].calc_grad()) continue;
        const int64_t bytes = int64_t(sizeof(float)) * ins[i].num_arcs();
        items.push_back({&ins[i], total});
        segs.push_back({nullptr, gr.w->dev + arc_off[m.idx][i], bytes});
        total += align_up(size_t(bytes), 16);
        longest = std::max(longest, bytes);
      }
    }
    if (items.empty()) return;
    DevMemP arena = rt.alloc(total ? total : 16);
    for (size_t k = 0; k < items.size(); ++k) segs[k].dst = arena->as<char>(items[k].off);
    DevMemP d = upload_vec(segs);
    launch_copy_segments(d->as<CopySeg>(), int(segs.size()), longest, rt.stream());
    for (auto& it : items) sink.add(*it.in, arena, arena->as<float>(it.off));
    sink.flush();
  }
};

Graph op_rational(int kind, std::vector<Graph>& ins, int projection) {
  Runtime& rt = Runtime::get();
  const bool closure = kind == RAT_CLOSURE, concat = kind == RAT_CONCAT;
  auto op = std::make_shared<RationalOp>();
  op->seq = next_seq();
  if (ins.empty()) {  // a^0 accepts the empty string (functions.cpp:117-121); the empty union is the empty graph
    Graph out = make_output(op, 0, {});
    if (concat) out.add_node(true, true);
    op->arc_off.push_back({});
    return out;
  }
  std::vector<Structure*> ss;
  std::vector<Weights*> ws;
  for (auto& g : ins) {
    g.s->resolve_sizes();
    if (g.s->kind != KIND_LINEAR) ss.push_back(g.s.get());
    ws.push_back(g.w.get());
  }
  ensure_device_batch(ss);
  ensure_weights_device_batch(ws);
  const int k = int(ins.size());
  std::vector<RationalSeg> segs;
  segs.resize(size_t(k));
  int64_t N = closure ? 1 : 0, A = 0;
  int max_A = 0, max_N = 0, max_conn = 0;
  bool eps_free = true;
  std::vector<int64_t> offs;
  for (int i = 0; i < k; ++i) {
    RationalSeg& s = segs[size_t(i)];
    s = RationalSeg{};
    s.g = device_view(ins[size_t(i)]);
    if (s.g.kind == KIND_LINEAR) {
      s.g.N = int(ins[size_t(i)].s->N);
      s.g.A = int(ins[size_t(i)].s->A);
      s.g.M = ins[size_t(i)].s->M;
      s.g.C = ins[size_t(i)].s->C;
      s.g.n_start = s.g.n_accept = 1;
    } else if (!(s.g.flags & 4)) {
      eps_free = false;
    }
    s.node_off = int(N);
    if (concat && i > 0) {  // the connectors into graph i come right after graph i's own arcs (functions.cpp:139-149)
      s.arc_off = int(A);
      s.conn_off = int(A) + s.g.A;
    } else {
      s.arc_off = int(A);
      s.conn_off = int(A) + s.g.A;
    }
    offs.push_back(A);
    int conn = 0;
    if (concat && i > 0) conn = segs[size_t(i) - 1].g.n_accept * s.g.n_start;
    if (closure) conn = s.g.n_start + s.g.n_accept;
    if (conn) eps_free = false;
    s.keep_start = closure ? 0 : (concat ? i == 0 : 1);
    s.keep_accept = closure ? 0 : (concat ? i == k - 1 : 1);
    N += s.g.N;
    A += int64_t(s.g.A) + conn;
    max_A = std::max(max_A, s.g.A);
    max_N = std::max(max_N, s.g.N);
    max_conn = std::max(max_conn, conn);
  }
  if (N > (int64_t(1) << 30) || A > (int64_t(1) << 30)) throw_runtime("[gtn] rational operation: result too large");
  // one arena: arc arrays, weights, flags, lists, adjacency
  size_t bytes = 0;
  auto add = [&](size_t b) {
    const size_t at = bytes;
    bytes = align_up(bytes + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(N);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_w = add(a4), o_fl = add(size_t(N)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(bytes);
  RationalOut ro{};
  ro.N = int(N);
  ro.A = int(A);
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_w);
  ro.nflags = arena->as<uint8_t>(o_fl);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  DevMemP dsegs = upload_vec(segs);
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(int(N), int(A)));
  launch_rational_build(dsegs->as<RationalSeg>(), k, max_A, max_N, max_conn, ro, projection, closure ? 1 : 0, temp->ptr, rt.stream());
  // counts of the output's start / accept nodes follow from the inputs'
  int n_start = 0, n_accept = 0;
  if (closure) n_start = n_accept = 1;
  else if (concat) n_start = segs.front().g.n_start, n_accept = segs.back().g.n_accept;
  else
    for (auto& s : segs) n_start += s.g.n_start, n_accept += s.g.n_accept;
  Graph out = make_output(op, 0, ins);
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = N;
  st.A = A;
  st.host_valid = false;
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = int(N);
  v.A = int(A);
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  op->arc_off.push_back(std::move(offs));
  return out;
}

// ======================================================================
// remove (functions.cpp:253-318), built on the device: rational.hip
// ======================================================================
struct RemoveOp : OpRecord {
  void backward(std::vector<Member>&) override {
    throw_logic("[gtn::remove] gradient compuation not implemented");  // functions.cpp:271-273
  }
};

Graph op_remove(Graph& gin, int ilabel, int olabel) {
  Runtime& rt = Runtime::get();
  auto op = std::make_shared<RemoveOp>();
  op->seq = next_seq();
  std::vector<Graph> ins{gin};
  Graph out = make_output(op, 0, ins);
  gin.s->resolve_sizes();
  // the structure the walk reads: explicit, with out-lists on the device (an implicit chain is written out into a
  // copy; a composition result that keeps its out-lists implicit goes through its host mirror once)
  Graph src = gin;
  if (src.s->kind == KIND_LINEAR) src = Graph::deep_copy(gin);
  std::vector<Structure*> ss{src.s.get()};
  ensure_device_batch(ss);
  DGraph g = device_view(src);
  if (g.A > 0 && (!g.out_off || !g.out_list)) {
    src = Graph::deep_copy(src);  // (host arrays, uploaded with their adjacency lists)
    ss[0] = src.s.get();
    ensure_device_batch(ss);
    g = device_view(src);
  }
  const int N = g.N;
  if (N == 0) return out;
  const size_t scan_b = scan_temp_bytes(N + 1);
  // keep flags -> ids of the kept nodes
  DevMemP ids = rt.alloc(4 * (3 * size_t(N) + 8) + scan_b);
  int* keep = ids->as<int>();
  int* new_id = keep + (N + 1);
  int* roots = new_id + (N + 1);
  void* scan_tmp = roots + N + 2;
  HIP_CHECK(hipMemsetAsync(keep, 0, 4 * size_t(N + 1), rt.stream()));
  launch_remove_keep(g, ilabel, olabel, keep, rt.stream());
  launch_exclusive_scan(keep, new_id, N + 1, scan_tmp, scan_b, rt.stream());
  int K = 0;
  rt.d2h_sync(&K, new_id + N, sizeof(int));
  if (K == 0) return out;
  launch_remove_roots(keep, new_id, N, roots, rt.stream());
  // walks: batches of `rows` kept nodes share 2 x rows x N ints of scratch (at most ~256 MB)
  const int rows = int(std::max<int64_t>(1, std::min<int64_t>(K, (int64_t(32) << 20) / std::max(N, 1))));
  DevMemP scratch = rt.alloc_zero(8 * size_t(rows) * size_t(N));
  DevMemP counts = rt.alloc(4 * (2 * size_t(K) + 4) + scan_temp_bytes(K + 1));
  RemoveArgs ra{};
  ra.g = g;
  ra.ilabel = ilabel;
  ra.olabel = olabel;
  ra.new_id = new_id;
  ra.roots = roots;
  ra.K = K;
  ra.rows = rows;
  ra.stamp = scratch->as<int>();
  ra.queue = ra.stamp + size_t(rows) * size_t(N);
  ra.arc_cnt = counts->as<int>();
  int* arc_off = ra.arc_cnt + (K + 1);
  HIP_CHECK(hipMemsetAsync(ra.arc_cnt, 0, 4 * size_t(K + 1), rt.stream()));
  for (int r0 = 0; r0 < K; r0 += rows) {
    ra.root0 = r0;
    launch_remove_walk(ra, false, rt.stream());
  }
  launch_exclusive_scan(ra.arc_cnt, arc_off, K + 1, arc_off + K + 2, scan_temp_bytes(K + 1), rt.stream());
  int A = 0;
  rt.d2h_sync(&A, arc_off + K, sizeof(int));
  // the result's arena (as op_rational lays it out)
  size_t total = 0;
  auto add = [&](size_t b) {
    const size_t at = total;
    total = align_up(total + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(K);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_w = add(a4), o_nf = add(size_t(K)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(total);
  RationalOut ro{};
  ro.N = K;
  ro.A = A;
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_w);
  ro.nflags = arena->as<uint8_t>(o_nf);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  ra.out = ro;
  ra.arc_off = arc_off;
  // (the stamps of the count pass are k + 1: the emit pass uses K + k + 1 through a shifted tag base)
  HIP_CHECK(hipMemsetAsync(ra.stamp, 0, 4 * size_t(rows) * size_t(N), rt.stream()));
  for (int r0 = 0; r0 < K; r0 += rows) {
    ra.root0 = r0;
    launch_remove_walk(ra, true, rt.stream());
  }
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(K, A));
  launch_rational_adjacency(ro, temp->ptr, rt.stream());
  // start / accept counts for the view (the ordered lists are built by the adjacency pass)
  std::vector<uint8_t> fl(static_cast<size_t>(K));
  rt.d2h_sync(fl.data(), ro.nflags, size_t(K));
  int n_start = 0, n_accept = 0;
  bool eps_free = false;  // (not known without a pass over the labels: claim nothing)
  for (uint8_t f : fl) n_start += (f & NF_START) ? 1 : 0, n_accept += (f & NF_ACCEPT) ? 1 : 0;
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = K;
  st.A = A;
  st.host_valid = false;
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = K;
  v.A = A;
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  return out;
}

// ======================================================================
// binary graph format (utils.cpp:152-225) straight into device buffers
// ======================================================================
Graph op_load_buffer(const void* data, size_t bytes) {
  // layout: int32 {N, A, n_start, n_accept} | start[n_start] | accept[n_accept] | {src, dst, ilabel, olabel} x A | float w[A]
  const char* p = static_cast<const char*>(data);
  if (bytes < 16) throw_invalid("[gtn::load] truncated graph file");
  int head[4];
  std::memcpy(head, p, 16);
  const int64_t N = head[0], A = head[1], ns = head[2], na = head[3];
  if (N < 0 || A < 0 || ns < 0 || na < 0 || ns > N || na > N) throw_invalid("[gtn::load] corrupt graph file header");
  const size_t need = 16 + 4 * size_t(ns + na) + 20 * size_t(A);
  if (bytes < need) throw_invalid("[gtn::load] truncated graph file");
  const int* start = reinterpret_cast<const int*>(p + 16);
  const int* accept = start + ns;
  const int* rows = accept + na;
  const float* w = reinterpret_cast<const float*>(rows + 4 * A);
  Graph out(true);
  if (N == 0) return out;
  // A small graph -- or a host without a GPU: building a graph is host work everywhere in this engine, only the
  // graph FUNCTIONS need the device -- is put together like addNode / addArc would, in two bulk appends; large
  // decoding graphs take the device route below.
  if (A < 4096 || Runtime::device_count() == 0) {
    std::vector<uint8_t> st(static_cast<size_t>(N), 0), ac(static_cast<size_t>(N), 0);
    for (int64_t i = 0; i < ns; ++i) {
      if (start[i] < 0 || start[i] >= N) throw_range("[gtn::load] start node out of range");
      st[size_t(start[i])] = 1;
    }
    for (int64_t i = 0; i < na; ++i) {
      if (accept[i] < 0 || accept[i] >= N) throw_range("[gtn::load] accept node out of range");
      ac[size_t(accept[i])] = 1;
    }
    out.add_nodes(int(N), st.data(), ac.data());
    std::vector<int> src(static_cast<size_t>(A)), dst(src.size()), il(src.size()), ol(src.size());
    for (int64_t a = 0; a < A; ++a) {
      src[size_t(a)] = rows[4 * a];
      dst[size_t(a)] = rows[4 * a + 1];
      il[size_t(a)] = rows[4 * a + 2];
      ol[size_t(a)] = rows[4 * a + 3];
    }
    std::vector<float> ww(w, w + A);  // (the file image need not be aligned for floats in place)
    out.add_arcs(int(A), src.data(), dst.data(), il.data(), ol.data(), ww.data());
    return out;
  }
  Runtime& rt = Runtime::get();
  // host: node flags from the two lists; the arc table is checked like addArc checks it (graph.cpp:47-66)
  std::vector<uint8_t> flags(static_cast<size_t>(N), 0);
  for (int64_t i = 0; i < ns; ++i) {
    if (start[i] < 0 || start[i] >= N) throw_range("[gtn::load] start node out of range");
    flags[size_t(start[i])] |= NF_START;
  }
  for (int64_t i = 0; i < na; ++i) {
    if (accept[i] < 0 || accept[i] >= N) throw_range("[gtn::load] accept node out of range");
    flags[size_t(accept[i])] |= NF_ACCEPT;
  }
  bool eps_free = true;
  for (int64_t a = 0; a < A; ++a) {
    const int* r = rows + 4 * a;
    if (r[0] < 0 || r[0] >= N || r[1] < 0 || r[1] >= N) throw_range("[Graph::addArc] node index out of range");
    if (r[2] < GTNX_EPSILON || r[3] < GTNX_EPSILON) throw_invalid("[Graph::addArc] labels must be >= epsilon");
    eps_free = eps_free && r[2] >= 0 && r[3] >= 0;
  }
  // one staging copy: rows (16-byte aligned) | weights | flags
  const size_t o_rows = 0, o_w = align_up(16 * size_t(A), 256), o_fl = align_up(o_w + 4 * size_t(A), 256);
  const size_t in_bytes = align_up(o_fl + size_t(N), 256);
  PinnedMemP pin = rt.alloc_pinned(in_bytes);
  std::memcpy(pin->as<char>(o_rows), rows, 16 * size_t(A));
  std::memcpy(pin->as<char>(o_w), w, 4 * size_t(A));
  std::memcpy(pin->as<char>(o_fl), flags.data(), size_t(N));
  DevMemP raw = rt.alloc(in_bytes);
  rt.h2d_pinned(raw->ptr, pin->ptr, in_bytes);
  // the structure's own arena: arc arrays, weights, flags, lists, adjacency (as op_rational lays it out)
  size_t total = 0;
  auto add = [&](size_t b) {
    const size_t at = total;
    total = align_up(total + (b ? b : 4), 256);
    return at;
  };
  const size_t a4 = 4 * size_t(A), n4 = 4 * size_t(N);
  const size_t o_src = add(a4), o_dst = add(a4), o_il = add(a4), o_ol = add(a4), o_ww = add(a4), o_nf = add(size_t(N)),
               o_st = add(n4), o_ac = add(n4), o_oo = add(n4 + 4), o_ol2 = add(a4), o_io = add(n4 + 4), o_il2 = add(a4);
  DevMemP arena = rt.alloc(total);
  RationalOut ro{};
  ro.N = int(N);
  ro.A = int(A);
  ro.src = arena->as<int>(o_src);
  ro.dst = arena->as<int>(o_dst);
  ro.il = arena->as<int>(o_il);
  ro.ol = arena->as<int>(o_ol);
  ro.w = arena->as<float>(o_ww);
  ro.nflags = arena->as<uint8_t>(o_nf);
  ro.start_list = arena->as<int>(o_st);
  ro.accept_list = arena->as<int>(o_ac);
  ro.out_off = arena->as<int>(o_oo);
  ro.out_list = arena->as<int>(o_ol2);
  ro.in_off = arena->as<int>(o_io);
  ro.in_list = arena->as<int>(o_il2);
  DevMemP temp = rt.alloc(rational_csr_temp_bytes(int(N), int(A)));
  launch_rational_load(raw->as<char>(o_rows), raw->as<float>(o_w), raw->as<uint8_t>(o_fl), ro, temp->ptr, rt.stream());
  int n_start = 0, n_accept = 0;  // (a node listed twice counts once: addNode's lists have one entry per node)
  for (uint8_t f : flags) n_start += (f & NF_START) ? 1 : 0, n_accept += (f & NF_ACCEPT) ? 1 : 0;
  Structure& st = *out.s;
  st.kind = KIND_EXPLICIT;
  st.N = N;
  st.A = A;
  st.host_valid = false;  // the host mirror is pulled when somebody inspects the graph
  st.csr_valid = false;
  st.dev_valid = true;
  st.dev_mem = arena;
  DGraph& v = st.dview;
  v = DGraph{};
  v.kind = KIND_EXPLICIT;
  v.N = int(N);
  v.A = int(A);
  v.n_start = n_start;
  v.n_accept = n_accept;
  v.flags = eps_free ? 4 : 0;
  v.src = ro.src;
  v.dst = ro.dst;
  v.il = ro.il;
  v.ol = ro.ol;
  v.nflags = ro.nflags;
  v.start_list = ro.start_list;
  v.accept_list = ro.accept_list;
  v.out_off = ro.out_off;
  v.out_list = ro.out_list;
  v.in_off = ro.in_off;
  v.in_list = ro.in_list;
  set_dev_weights(out, arena, ro.w, A);
  return out;
}


} // namespace gtnx
