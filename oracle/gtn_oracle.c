/*
 * gtn_oracle.c -- CPU restatement of the gtn hot path.  TEST INFRASTRUCTURE ONLY
 * (see gtn_oracle.h for the scope, the reference citations and the pinning
 * status).  Plain C99, scalar, single-threaded; written for clarity, not speed.
 */
#include "gtn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* tiny growable int vector                                            */
/* ------------------------------------------------------------------ */
typedef struct {
  int* d;
  int n, cap;
} ivec;

static void iv_push(ivec* v, int x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 4;
    v->d = (int*)realloc(v->d, sizeof(int) * (size_t)v->cap);
  }
  v->d[v->n++] = x;
}
static void iv_free(ivec* v) {
  free(v->d);
  v->d = NULL;
  v->n = v->cap = 0;
}

/* ------------------------------------------------------------------ */
/* graph storage  (gtn/graph.h:58-73, 439-451; gtn/graph.cpp:33-67)     */
/* ------------------------------------------------------------------ */
struct og_graph {
  int N, A, capN, capA;
  uint8_t *start, *accept; /* per-node flags */
  ivec* in;                /* per-node arc-id lists, reference order */
  ivec* out;
  ivec start_list, accept_list;
  int *src, *dst, *il, *ol;
  float* w;
  int ilabel_sorted, olabel_sorted;
  /* set on graphs made by og_compose */
  int* grad_info; /* 2*A */
  /* caches of the last og_shortest_distance call */
};

og_graph* og_new(void) { return (og_graph*)calloc(1, sizeof(og_graph)); }

void og_free(og_graph* g) {
  if (!g) return;
  for (int n = 0; n < g->N; ++n) {
    iv_free(&g->in[n]);
    iv_free(&g->out[n]);
  }
  free(g->in);
  free(g->out);
  free(g->start);
  free(g->accept);
  iv_free(&g->start_list);
  iv_free(&g->accept_list);
  free(g->src);
  free(g->dst);
  free(g->il);
  free(g->ol);
  free(g->w);
  free(g->grad_info);
  free(g);
}

/* gtn/graph.cpp:33-45 */
int og_add_node(og_graph* g, int start, int accept) {
  if (g->N == g->capN) {
    g->capN = g->capN ? g->capN * 2 : 16;
    g->start = (uint8_t*)realloc(g->start, (size_t)g->capN);
    g->accept = (uint8_t*)realloc(g->accept, (size_t)g->capN);
    g->in = (ivec*)realloc(g->in, sizeof(ivec) * (size_t)g->capN);
    g->out = (ivec*)realloc(g->out, sizeof(ivec) * (size_t)g->capN);
  }
  int idx = g->N++;
  g->start[idx] = (uint8_t)(start != 0);
  g->accept[idx] = (uint8_t)(accept != 0);
  memset(&g->in[idx], 0, sizeof(ivec));
  memset(&g->out[idx], 0, sizeof(ivec));
  if (start) iv_push(&g->start_list, idx);
  if (accept) iv_push(&g->accept_list, idx);
  g->ilabel_sorted = g->olabel_sorted = 0;
  return idx;
}

/* gtn/graph.cpp:51-67 */
int og_add_arc(og_graph* g, int src, int dst, int ilabel, int olabel, float w) {
  if (g->A == g->capA) {
    g->capA = g->capA ? g->capA * 2 : 32;
    size_t c = (size_t)g->capA;
    g->src = (int*)realloc(g->src, sizeof(int) * c);
    g->dst = (int*)realloc(g->dst, sizeof(int) * c);
    g->il = (int*)realloc(g->il, sizeof(int) * c);
    g->ol = (int*)realloc(g->ol, sizeof(int) * c);
    g->w = (float*)realloc(g->w, sizeof(float) * c);
  }
  int idx = g->A++;
  g->src[idx] = src;
  g->dst[idx] = dst;
  g->il[idx] = ilabel;
  g->ol[idx] = olabel;
  g->w[idx] = w;
  iv_push(&g->out[src], idx);
  iv_push(&g->in[dst], idx);
  g->ilabel_sorted = g->olabel_sorted = 0;
  return idx;
}

int og_num_nodes(const og_graph* g) { return g->N; }
int og_num_arcs(const og_graph* g) { return g->A; }
int og_num_start(const og_graph* g) { return g->start_list.n; }
int og_num_accept(const og_graph* g) { return g->accept_list.n; }

void og_get_nodes(const og_graph* g, uint8_t* start, uint8_t* accept) {
  if (start) memcpy(start, g->start, (size_t)g->N);
  if (accept) memcpy(accept, g->accept, (size_t)g->N);
}

void og_get_arcs(const og_graph* g, int* src, int* dst, int* il, int* ol,
                 float* w) {
  size_t b = sizeof(int) * (size_t)g->A;
  if (src) memcpy(src, g->src, b);
  if (dst) memcpy(dst, g->dst, b);
  if (il) memcpy(il, g->il, b);
  if (ol) memcpy(ol, g->ol, b);
  if (w) memcpy(w, g->w, sizeof(float) * (size_t)g->A);
}

/* gtn/graph.cpp:179-181 */
void og_set_weights(og_graph* g, const float* w) {
  memcpy(g->w, w, sizeof(float) * (size_t)g->A);
}

/* stable insertion/merge sort of an arc-id list by a label array */
static void sort_list(ivec* v, const int* key) {
  /* lists are short in every use of the oracle; binary-insertion keeps it
   * stable and allocation-free */
  for (int i = 1; i < v->n; ++i) {
    int a = v->d[i], k = key[a], j = i - 1;
    while (j >= 0 && key[v->d[j]] > k) {
      v->d[j + 1] = v->d[j];
      --j;
    }
    v->d[j + 1] = a;
  }
}

/* gtn/graph.cpp:162-177 */
void og_arc_sort(og_graph* g, int olabel) {
  if ((olabel && g->olabel_sorted) || (!olabel && g->ilabel_sorted)) return;
  g->olabel_sorted = olabel != 0;
  g->ilabel_sorted = olabel == 0;
  const int* key = olabel ? g->ol : g->il;
  for (int n = 0; n < g->N; ++n) {
    sort_list(&g->in[n], key);
    sort_list(&g->out[n], key);
  }
}

void og_mark_sorted(og_graph* g, int olabel) {
  if (olabel)
    g->olabel_sorted = 1;
  else
    g->ilabel_sorted = 1;
}

int og_is_sorted(const og_graph* g, int olabel) {
  return olabel ? g->olabel_sorted : g->ilabel_sorted;
}

/* gtn/creations.cpp:20-33 */
og_graph* og_linear_graph(int M, int N) {
  og_graph* g = og_new();
  og_add_node(g, 1, 0); /* creations.cpp:22: start, never accepting */
  for (int m = 1; m <= M; ++m) {
    og_add_node(g, 0, m == M);
    for (int n = 0; n < N; ++n) og_add_arc(g, m - 1, m, n, n, 0.0f);
  }
  g->ilabel_sorted = g->olabel_sorted = 1;
  return g;
}

/* ------------------------------------------------------------------ */
/* shortest distance  (gtn/functions/shortest.cpp:86-188)               */
/* ------------------------------------------------------------------ */

/* shortest.cpp:102-114 (the getScore lambda) */
static float reduce_scores(const float* in, int n, float max_score,
                           int tropical) {
  if (n == 0) return -INFINITY;
  if (tropical || max_score == INFINITY || max_score == -INFINITY)
    return max_score;
  float s = -1.0f;
  for (int i = 0; i < n; ++i) s += expf(in[i] - max_score);
  return max_score + log1pf(s);
}

static int sd_forward(og_graph* g, int tropical, float* out_score,
                      float* scores, float* maxc, int64_t* argc) {
  int N = g->N;
  int* queue = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  int qh = 0, qt = 0;
  int* deg = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  for (int n = 0; n < N; ++n) {
    scores[n] = 0.0f;
    maxc[n] = -INFINITY;
    argc[n] = -1;
    deg[n] = g->in[n].n;
  }
  maxc[N] = -INFINITY;
  argc[N] = -1;
  /* shortest.cpp:96-100: seed with start nodes of in-degree 0 */
  for (int k = 0; k < g->start_list.n; ++k) {
    int n = g->start_list.d[k];
    if (g->in[n].n == 0) queue[qt++] = n;
  }
  int cap = 16;
  float* ins = (float*)malloc(sizeof(float) * (size_t)cap);
  while (qh < qt) {
    int n = queue[qh++];
    int cnt = 0;
    if (g->in[n].n + 1 > cap) {
      cap = g->in[n].n + 1;
      ins = (float*)realloc(ins, sizeof(float) * (size_t)cap);
    }
    /* shortest.cpp:121-128 */
    for (int k = 0; k < g->in[n].n; ++k) {
      int a = g->in[n].d[k];
      float s = scores[g->src[a]] + g->w[a];
      ins[cnt++] = s;
      if (s > maxc[n]) {
        maxc[n] = s;
        argc[n] = a;
      }
    }
    /* shortest.cpp:129-135: start nodes see a virtual 0.0 in-score */
    if (g->start[n]) {
      ins[cnt++] = 0.0f;
      if (0.0f > maxc[n]) {
        maxc[n] = 0.0f;
        argc[n] = -1;
      }
    }
    scores[n] = reduce_scores(ins, cnt, maxc[n], tropical);
    /* shortest.cpp:139-144 */
    for (int k = 0; k < g->out[n].n; ++k) {
      int d = g->dst[g->out[n].d[k]];
      if (--deg[d] == 0) queue[qt++] = d;
    }
  }
  /* shortest.cpp:148-159 */
  int err = 0;
  int na = g->accept_list.n;
  if (na + 1 > cap) ins = (float*)realloc(ins, sizeof(float) * (size_t)(na + 1));
  int cnt = 0;
  for (int k = 0; k < na; ++k) {
    int n = g->accept_list.d[k];
    if (deg[n] > 0) {
      err = 1;
      break;
    }
    ins[cnt++] = scores[n];
    if (scores[n] > maxc[N]) {
      maxc[N] = scores[n];
      argc[N] = n; /* a NODE id, as in the reference */
    }
  }
  if (!err) *out_score = reduce_scores(ins, cnt, maxc[N], tropical);
  free(ins);
  free(deg);
  free(queue);
  return err;
}

int og_shortest_distance(og_graph* g, int tropical, float* out_score,
                         float* node_scores, float* max_cache,
                         int64_t* argmax_cache) {
  int N = g->N;
  float* sc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  float* mc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  int64_t* ac = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N + 1));
  float out = 0.0f;
  int err = sd_forward(g, tropical, &out, sc, mc, ac);
  if (!err) {
    if (out_score) *out_score = out;
    if (node_scores) memcpy(node_scores, sc, sizeof(float) * (size_t)N);
    if (max_cache) memcpy(max_cache, mc, sizeof(float) * (size_t)(N + 1));
    if (argmax_cache) memcpy(argmax_cache, ac, sizeof(int64_t) * (size_t)(N + 1));
  }
  free(sc);
  free(mc);
  free(ac);
  return err;
}

/* gtn/functions/shortest.cpp:33-82 */
int og_shortest_distance_grad(og_graph* g, int tropical, float delta,
                              float* arc_grads) {
  int N = g->N, A = g->A;
  float* sc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  float* mc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  int64_t* ac = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N + 1));
  float output = 0.0f;
  int err = sd_forward(g, tropical, &output, sc, mc, ac);
  if (err) {
    free(sc);
    free(mc);
    free(ac);
    return err;
  }
  int* queue = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  int qh = 0, qt = 0;
  int* deg = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  float* ng = (float*)calloc((size_t)(N + 1), sizeof(float));
  for (int a = 0; a < A; ++a) arc_grads[a] = 0.0f;
  for (int n = 0; n < N; ++n) deg[n] = g->out[n].n;
  /* shortest.cpp:48-60 */
  float denom = tropical ? 0.0f : expf(output - mc[N]);
  for (int k = 0; k < g->accept_list.n; ++k) {
    int n = g->accept_list.d[k];
    if (g->out[n].n == 0) queue[qt++] = n;
    float cur;
    if (tropical)
      cur = ((int64_t)n == ac[N]) ? 1.0f : 0.0f;
    else
      cur = expf(sc[n] - mc[N]) / denom;
    ng[n] += cur;
  }
  /* shortest.cpp:62-80 */
  while (qh < qt) {
    int n = queue[qh++];
    denom = tropical ? 0.0f : expf(sc[n] - mc[n]);
    for (int k = 0; k < g->in[n].n; ++k) {
      int a = g->in[n].d[k];
      int un = g->src[a];
      float cur;
      if (tropical)
        cur = ((int64_t)a == ac[n]) ? ng[n] : 0.0f;
      else
        cur = ng[n] * expf(sc[un] + g->w[a] - mc[n]) / denom;
      ng[un] += cur;
      arc_grads[a] = cur * delta;
      if (--deg[un] == 0) queue[qt++] = un;
    }
  }
  free(queue);
  free(deg);
  free(ng);
  free(sc);
  free(mc);
  free(ac);
  return 0;
}

/* gtn/functions/shortest.cpp:190-272 */
int og_shortest_path(og_graph* g, int* out_arcs, int* n_arcs, int* has_node) {
  int N = g->N;
  int* queue = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  int qh = 0, qt = 0;
  int* deg = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  int* bp = (int*)calloc((size_t)(N + 1), sizeof(int));
  float* sc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  for (int n = 0; n < N; ++n) {
    deg[n] = g->in[n].n;
    sc[n] = -INFINITY;
  }
  for (int k = 0; k < g->start_list.n; ++k) {
    int n = g->start_list.d[k];
    sc[n] = 0.0f;
    bp[n] = -1;
    if (g->in[n].n == 0) queue[qt++] = n;
  }
  /* shortest.cpp:208-223: push-style relaxation, strict '>' */
  while (qh < qt) {
    int n = queue[qh++];
    float s = sc[n];
    for (int k = 0; k < g->out[n].n; ++k) {
      int a = g->out[n].d[k];
      int d = g->dst[a];
      float ns = s + g->w[a];
      if (ns > sc[d]) {
        sc[d] = ns;
        bp[d] = a;
      }
      if (--deg[d] == 0) queue[qt++] = d;
    }
  }
  /* shortest.cpp:226-237 */
  int err = 0, best = -1;
  float score = -INFINITY;
  for (int k = 0; k < g->accept_list.n; ++k) {
    int a = g->accept_list.d[k];
    if (deg[a] > 0) {
      err = 1;
      break;
    }
    if (sc[a] > score) {
      score = sc[a];
      best = a;
    }
  }
  if (!err) {
    /* shortest.cpp:240-245: chase back-pointers (collected last-arc-first) */
    int cnt = 0;
    int* rev = (int*)malloc(sizeof(int) * (size_t)(N + 1));
    while (best != -1 && bp[best] != -1) {
      int a = bp[best];
      best = g->src[a];
      rev[cnt++] = a;
    }
    for (int i = 0; i < cnt; ++i) out_arcs[i] = rev[cnt - 1 - i];
    *n_arcs = cnt;
    *has_node = (best != -1);
    free(rev);
  }
  free(queue);
  free(deg);
  free(bp);
  free(sc);
  return err;
}

/* ------------------------------------------------------------------ */
/* matchers  (gtn/functions/compose.cpp:211-374)                        */
/* The reference drives a stateful hasNext()/next() iterator; here each   */
/* matcher is restated as "enumerate all (i, j) into a list", which gives */
/* the same pairs in the same order.                                      */
/* ------------------------------------------------------------------ */
enum { M_UNSORTED = 0, M_SINGLY_G1 = 1, M_SINGLY_G2 = 2, M_DOUBLY = 3 };

typedef struct {
  int *i, *j;
  int n, cap;
} pairlist;

static void pl_push(pairlist* p, int i, int j) {
  if (p->n == p->cap) {
    p->cap = p->cap ? p->cap * 2 : 16;
    p->i = (int*)realloc(p->i, sizeof(int) * (size_t)p->cap);
    p->j = (int*)realloc(p->j, sizeof(int) * (size_t)p->cap);
  }
  p->i[p->n] = i;
  p->j[p->n] = j;
  p->n++;
}

/* first position in lst[lo..n) whose key is >= val (std::lower_bound) */
static int lower_bound_from(const ivec* lst, int lo, const int* key, int val) {
  int hi = lst->n;
  while (lo < hi) {
    int mid = lo + (hi - lo) / 2;
    if (key[lst->d[mid]] < val)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

static void match_pairs(const og_graph* g1, const og_graph* g2, int kind,
                        int n1, int n2, int match_in, pairlist* out) {
  const ivec* lv = match_in ? &g1->in[n1] : &g1->out[n1];
  const ivec* rv = match_in ? &g2->in[n2] : &g2->out[n2];
  out->n = 0;
  if (kind == M_UNSORTED) {
    /* compose.cpp:220-234 */
    for (int a = 0; a < lv->n; ++a)
      for (int b = 0; b < rv->n; ++b)
        if (g1->ol[lv->d[a]] == g2->il[rv->d[b]])
          pl_push(out, lv->d[a], rv->d[b]);
    return;
  }
  int search_g1;
  if (kind == M_DOUBLY)
    search_g1 = lv->n > rv->n; /* compose.cpp:319 */
  else
    search_g1 = (kind == M_SINGLY_G1); /* compose.cpp:236-259 */
  const ivec* search = search_g1 ? lv : rv;
  const ivec* query = search_g1 ? rv : lv;
  const int* skey = search_g1 ? g1->ol : g2->il;
  const int* qkey = search_g1 ? g2->il : g1->ol;
  int sbegin = 0;
  for (int q = 0; q < query->n; ++q) {
    int ql = qkey[query->d[q]];
    /* singly: restart the search at the list head for every query
     * (compose.cpp:284-285); doubly: the search start only moves forward and
     * an exhausted search list ends the enumeration (compose.cpp:352-356) */
    int pos = lower_bound_from(search, kind == M_DOUBLY ? sbegin : 0, skey, ql);
    if (kind == M_DOUBLY) {
      sbegin = pos;
      if (pos == search->n) break;
    }
    for (int s = pos; s < search->n && skey[search->d[s]] == ql; ++s) {
      if (search_g1)
        pl_push(out, search->d[s], query->d[q]);
      else
        pl_push(out, query->d[q], search->d[s]);
    }
  }
}

/* ------------------------------------------------------------------ */
/* compose  (gtn/functions/compose.cpp:377-522)                         */
/* ------------------------------------------------------------------ */
typedef struct {
  int *a, *b;
  size_t h, t, cap;
} pairq;

static void pq_push(pairq* q, int a, int b) {
  if (q->t == q->cap) {
    q->cap = q->cap ? q->cap * 2 : 64;
    q->a = (int*)realloc(q->a, sizeof(int) * q->cap);
    q->b = (int*)realloc(q->b, sizeof(int) * q->cap);
  }
  q->a[q->t] = a;
  q->b[q->t] = b;
  q->t++;
}

#define IDX(n1, n2) ((size_t)(n1) + (size_t)N1 * (size_t)(n2)) /* compose.cpp:17-19 */

/* compose.cpp:22-53 */
static void eps_reach_back(int second, const og_graph* g1, const og_graph* g2,
                           int n1, int n2, uint8_t* reach, pairq* q) {
  size_t N1 = (size_t)g1->N;
  const ivec* edges = second ? &g2->in[n2] : &g1->in[n1];
  int sorted = second ? g2->ilabel_sorted : g1->olabel_sorted;
  for (int k = 0; k < edges->n; ++k) {
    int a = edges->d[k];
    int label = second ? g2->il[a] : g1->ol[a];
    if (label != OG_EPSILON) {
      if (sorted) break;
      continue;
    }
    int un = second ? g2->src[a] : g1->src[a];
    size_t idx = second ? IDX(n1, un) : IDX(un, n2);
    if (!reach[idx]) {
      if (second)
        pq_push(q, n1, un);
      else
        pq_push(q, un, n2);
    }
    reach[idx] = 1;
  }
}

/* compose.cpp:64-104 */
static uint8_t* find_reachable(const og_graph* g1, const og_graph* g2,
                               int kind) {
  size_t N1 = (size_t)g1->N;
  uint8_t* reach = (uint8_t*)calloc((size_t)g1->N * (size_t)g2->N + 1, 1);
  pairq q = {0};
  pairlist pl = {0};
  for (int a = 0; a < g1->accept_list.n; ++a)
    for (int b = 0; b < g2->accept_list.n; ++b) {
      int f = g1->accept_list.d[a], s = g2->accept_list.d[b];
      pq_push(&q, f, s);
      reach[IDX(f, s)] = 1;
    }
  while (q.h < q.t) {
    int c1 = q.a[q.h], c2 = q.b[q.h];
    q.h++;
    match_pairs(g1, g2, kind, c1, c2, 1, &pl);
    for (int k = 0; k < pl.n; ++k) {
      int u1 = g1->src[pl.i[k]], u2 = g2->src[pl.j[k]];
      size_t idx = IDX(u1, u2);
      if (!reach[idx]) pq_push(&q, u1, u2);
      reach[idx] = 1;
    }
    eps_reach_back(0, g1, g2, c1, c2, reach, &q);
    eps_reach_back(1, g1, g2, c1, c2, reach, &q);
  }
  free(q.a);
  free(q.b);
  free(pl.i);
  free(pl.j);
  return reach;
}

typedef struct {
  const og_graph *g1, *g2;
  const uint8_t* reach;
  int* new_nodes;
  pairq q;
  og_graph* out;
  ivec ginfo; /* flattened (i, j) */
} cstate;

/* compose.cpp:108-136 */
static int add_reachable(cstate* S, int cur, int d1, int d2, float w, int il,
                         int ol) {
  size_t N1 = (size_t)S->g1->N;
  size_t idx = IDX(d1, d2);
  if (S->reach[idx]) {
    if (S->new_nodes[idx] < 0) {
      S->new_nodes[idx] =
          og_add_node(S->out, S->g1->start[d1] && S->g2->start[d2],
                      S->g1->accept[d1] && S->g2->accept[d2]);
      pq_push(&S->q, d1, d2);
    }
    og_add_arc(S->out, cur, S->new_nodes[idx], il, ol, w);
  }
  return S->reach[idx];
}

/* compose.cpp:146-208 */
static void add_eps(cstate* S, int second, int cur, int n1, int n2) {
  const og_graph *g1 = S->g1, *g2 = S->g2;
  const ivec* edges = second ? &g2->out[n2] : &g1->out[n1];
  int sorted = second ? g2->ilabel_sorted : g1->olabel_sorted;
  for (int k = 0; k < edges->n; ++k) {
    int a = edges->d[k];
    int label = second ? g2->il[a] : g1->ol[a];
    if (label != OG_EPSILON) {
      if (sorted) break;
      continue;
    }
    int ok = add_reachable(S, cur, second ? n1 : g1->dst[a],
                           second ? g2->dst[a] : n2,
                           second ? g2->w[a] : g1->w[a],
                           second ? OG_EPSILON : g1->il[a],
                           second ? g2->ol[a] : OG_EPSILON);
    if (ok) {
      iv_push(&S->ginfo, second ? -1 : a);
      iv_push(&S->ginfo, second ? a : -1);
    }
  }
}

og_graph* og_compose(og_graph* g1, og_graph* g2, int mode) {
  /* gtn/functions.cpp:225-251 */
  int s1 = mode ? (g1->ilabel_sorted || g1->olabel_sorted) : g1->olabel_sorted;
  int s2 = mode ? (g2->ilabel_sorted || g2->olabel_sorted) : g2->ilabel_sorted;
  int kind = (s1 && s2) ? M_DOUBLY
                        : (s1 ? M_SINGLY_G1 : (s2 ? M_SINGLY_G2 : M_UNSORTED));
  size_t N1 = (size_t)g1->N;
  cstate S;
  memset(&S, 0, sizeof(S));
  S.g1 = g1;
  S.g2 = g2;
  uint8_t* reach = find_reachable(g1, g2, kind);
  S.reach = reach;
  size_t np = (size_t)g1->N * (size_t)g2->N;
  S.new_nodes = (int*)malloc(sizeof(int) * (np + 1));
  for (size_t i = 0; i < np; ++i) S.new_nodes[i] = -1;
  S.out = og_new();
  /* compose.cpp:392-401 */
  for (int a = 0; a < g1->start_list.n; ++a)
    for (int b = 0; b < g2->start_list.n; ++b) {
      int s1n = g1->start_list.d[a], s2n = g2->start_list.d[b];
      size_t idx = IDX(s1n, s2n);
      if (reach[idx]) {
        S.new_nodes[idx] =
            og_add_node(S.out, 1, g1->accept[s1n] && g2->accept[s2n]);
        pq_push(&S.q, s1n, s2n);
      }
    }
  pairlist pl = {0};
  /* compose.cpp:409-489 */
  while (S.q.h < S.q.t) {
    int c1 = S.q.a[S.q.h], c2 = S.q.b[S.q.h];
    S.q.h++;
    int cur = S.new_nodes[IDX(c1, c2)];
    int eps_matched = 0;
    match_pairs(g1, g2, kind, c1, c2, 0, &pl);
    for (int k = 0; k < pl.n; ++k) {
      int i = pl.i[k], j = pl.j[k];
      if (g1->ol[i] == OG_EPSILON) { /* compose.cpp:425-428 */
        eps_matched = 1;
        continue;
      }
      int ok = add_reachable(&S, cur, g1->dst[i], g2->dst[j],
                             g1->w[i] + g2->w[j], g1->il[i], g2->ol[j]);
      if (ok) {
        iv_push(&S.ginfo, i);
        iv_push(&S.ginfo, j);
      }
    }
    /* compose.cpp:461-488 */
    if (!eps_matched || g2->accept[c2] || !g1->accept[c1])
      add_eps(&S, 0, cur, c1, c2);
    if (!eps_matched || g1->accept[c1]) add_eps(&S, 1, cur, c1, c2);
  }
  og_graph* out = S.out;
  out->grad_info = (int*)malloc(sizeof(int) * (size_t)(S.ginfo.n + 1));
  memcpy(out->grad_info, S.ginfo.d, sizeof(int) * (size_t)S.ginfo.n);
  iv_free(&S.ginfo);
  free(pl.i);
  free(pl.j);
  free(S.q.a);
  free(S.q.b);
  free(S.new_nodes);
  free(reach);
  return out;
}

const int* og_grad_info(const og_graph* c) { return c->grad_info; }

/* gtn/functions/compose.cpp:496-518 */
void og_compose_grad(const og_graph* c, const float* deltas, int A1, int A2,
                     float* grad1, float* grad2) {
  if (grad1) memset(grad1, 0, sizeof(float) * (size_t)A1);
  if (grad2) memset(grad2, 0, sizeof(float) * (size_t)A2);
  for (int k = 0; k < c->A; ++k) {
    int i = c->grad_info[2 * k], j = c->grad_info[2 * k + 1];
    if (grad1 && i >= 0) grad1[i] += deltas[k];
    if (grad2 && j >= 0) grad2[j] += deltas[k];
  }
}

/* ------------------------------------------------------------------ */
/* CTC loss  (benchmarks/ctc.cpp:40-58, 150-160)                        */
/* ------------------------------------------------------------------ */
og_graph* og_ctc_graph(const int* target, int U, int blank, int arc_sort) {
  int L = 2 * U + 1;
  og_graph* ctc = og_new();
  for (int l = 0; l < L; ++l) {
    int idx = (l - 1) / 2;
    og_add_node(ctc, l == 0, l == L - 1 || l == L - 2);
    int label = (l % 2) ? target[idx] : blank;
    og_add_arc(ctc, l, l, label, label, 0.0f);
    if (l > 0) og_add_arc(ctc, l - 1, l, label, label, 0.0f);
    if ((l % 2) && l > 1 && label != target[idx - 1])
      og_add_arc(ctc, l - 2, l, label, label, 0.0f);
  }
  if (arc_sort) og_arc_sort(ctc, 0);
  return ctc;
}

int og_ctc_loss(const float* emissions, int T, int C, const int* target, int U,
                float* loss, float* grad) {
  og_graph* ctc = og_ctc_graph(target, U, 0, 1);
  og_graph* em = og_linear_graph(T, C);
  og_set_weights(em, emissions);
  og_graph* comp = og_compose(ctc, em, 1);
  float z = 0.0f, s = 0.0f;
  int err = og_shortest_distance(em, 0, &z, NULL, NULL, NULL);
  if (!err) err = og_shortest_distance(comp, 0, &s, NULL, NULL, NULL);
  if (!err) {
    *loss = z - s;
    if (grad) {
      /* backward of subtract(forwardScore(em), forwardScore(comp)):
       * +1 into forwardScore(em), -1 into forwardScore(comp), the latter
       * scattered to the emissions through the compose gradInfo. */
      float* gz = (float*)malloc(sizeof(float) * (size_t)em->A);
      float* gc = (float*)malloc(sizeof(float) * (size_t)(comp->A + 1));
      float* g2 = (float*)malloc(sizeof(float) * (size_t)em->A);
      og_shortest_distance_grad(em, 0, 1.0f, gz);
      og_shortest_distance_grad(comp, 0, -1.0f, gc);
      og_compose_grad(comp, gc, ctc->A, em->A, NULL, g2);
      for (int a = 0; a < em->A; ++a) grad[a] = gz[a] + g2[a];
      free(gz);
      free(gc);
      free(g2);
    }
  }
  og_free(comp);
  og_free(em);
  og_free(ctc);
  return err;
}
