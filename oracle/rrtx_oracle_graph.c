/*
 * rrtx_oracle_graph.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped or linked into the product)
 * of the cost-propagation half of RRT^X as the reference runs it:
 *   BinaryHeap            R/heap.jl:80-273 (bubbleUp, bubbleDown, addToHeap, popHeap, removeFromHeap, updateHeap)
 *   keyQ / lessQ / greaterQ R/DRRT_Q.jl:2052-2077
 *   verifyInQueue / verifyInOSQueue R/DRRT_Q.jl:2364-2385
 *   cullCurrentNeighbors  R/DRRT_Q.jl:2388-2403
 *   nextOutNeighbor / nextInNeighbor (initial list, then current list) R/DRRT_Q.jl:2408-2455
 *   makeParentOf          R/DRRT_Q.jl:2459-2486
 *   recalculateLMCMineVTwo R/DRRT_Q.jl:2490-2541
 *   rewire                R/DRRT_Q.jl:2647-2700
 *   reduceInconsistency   R/DRRT_Q.jl:2703-2717
 *   propogateDescendants  R/DRRT_Q.jl:2724-2817
 *   the edge / parent part of addNewObstacle R/DRRT_Q.jl:3244-3268
 * Nodes are indices, edges are ids; lists keep the reference's order (JlistPush inserts at the FRONT,
 * R/jlist.jl:79-97; iteration runs front to back).
 *
 * Parity status: unpinned like the rest of the oracle (the reference holds no golden vectors and cannot
 * run here); pinned by hand-derived cases and by an independent shortest-path computation in tests/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rrtx_oracle.h"

typedef struct { int64_t *v; int64_t n, cap; } ivec;
static void iv_push(ivec *a, int64_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? 2 * a->cap : 8;
    a->v = (int64_t *)realloc(a->v, sizeof(int64_t) * (size_t)a->cap);
  }
  a->v[a->n++] = x;
}

struct orc_graph {
  int64_t n;
  double *lmc, *g;                 /* rrtLMC, rrtTreeCost */
  int64_t *parent_edge;            /* rrtParentEdge (edge id), valid when parent_used */
  uint8_t *parent_used, *in_q, *in_os, *move_goal;
  int64_t *heap_index;
  int64_t *succ_slot;              /* position of this node in its parent's successor vector */
  ivec *out_init, *out_cur, *in_init, *in_cur;   /* edge ids, appended = pushed to the FRONT: iterate from the end */
  ivec *succ;                      /* SuccessorList: child node ids (or -1 removed), appended = pushed to the front */
  /* edges */
  int64_t m, mcap;
  int64_t *es, *ee;
  double *ed;
  uint8_t *e_valid, *e_alive;
  /* priority queue and orphan stack */
  int64_t *heap;                   /* 1-based */
  int64_t heap_last, heap_cap;
  ivec os;                         /* Q.OS: appended = pushed to the front; the BACK of the list is index 0 */
  int64_t os_len;                  /* live length (Q.OS.length) */
};

orc_graph *orc_graph_create(int64_t n) {
  orc_graph *G = (orc_graph *)calloc(1, sizeof(orc_graph));
  G->n = n;
  G->lmc = (double *)malloc(sizeof(double) * (size_t)n);
  G->g = (double *)malloc(sizeof(double) * (size_t)n);
  G->parent_edge = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  G->parent_used = (uint8_t *)calloc((size_t)n, 1);
  G->in_q = (uint8_t *)calloc((size_t)n, 1);
  G->in_os = (uint8_t *)calloc((size_t)n, 1);
  G->move_goal = (uint8_t *)calloc((size_t)n, 1);
  G->heap_index = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  G->succ_slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  G->out_init = (ivec *)calloc((size_t)n, sizeof(ivec));
  G->out_cur = (ivec *)calloc((size_t)n, sizeof(ivec));
  G->in_init = (ivec *)calloc((size_t)n, sizeof(ivec));
  G->in_cur = (ivec *)calloc((size_t)n, sizeof(ivec));
  G->succ = (ivec *)calloc((size_t)n, sizeof(ivec));
  for (int64_t i = 0; i < n; ++i) { G->lmc[i] = INFINITY; G->g[i] = INFINITY; G->parent_edge[i] = -1; G->heap_index[i] = -1; G->succ_slot[i] = -1; }
  G->heap_cap = 64;
  G->heap = (int64_t *)malloc(sizeof(int64_t) * (size_t)(G->heap_cap + 1));
  return G;
}

void orc_graph_destroy(orc_graph *G) {
  if (!G) return;
  for (int64_t i = 0; i < G->n; ++i) { free(G->out_init[i].v); free(G->out_cur[i].v); free(G->in_init[i].v); free(G->in_cur[i].v); free(G->succ[i].v); }
  free(G->out_init); free(G->out_cur); free(G->in_init); free(G->in_cur); free(G->succ);
  free(G->lmc); free(G->g); free(G->parent_edge); free(G->parent_used); free(G->in_q); free(G->in_os); free(G->move_goal);
  free(G->heap_index); free(G->succ_slot); free(G->es); free(G->ee); free(G->ed); free(G->e_valid); free(G->e_alive);
  free(G->heap); free(G->os.v); free(G);
}

/* an edge start -> end placed in start's out list and end's in list: `initial` = the lists extend() fills for
 * a new node's own first neighbours (makeInitialOutNeighborOf / makeInitialInNeighborOf, never culled), otherwise
 * the current lists (makeNeighborOf).  Returns the edge id. */
int64_t orc_graph_add_edge(orc_graph *G, int64_t start, int64_t end, double dist, int initial, int valid_move) {
  if (G->m == G->mcap) {
    G->mcap = G->mcap ? 2 * G->mcap : 64;
    G->es = (int64_t *)realloc(G->es, sizeof(int64_t) * (size_t)G->mcap);
    G->ee = (int64_t *)realloc(G->ee, sizeof(int64_t) * (size_t)G->mcap);
    G->ed = (double *)realloc(G->ed, sizeof(double) * (size_t)G->mcap);
    G->e_valid = (uint8_t *)realloc(G->e_valid, (size_t)G->mcap);
    G->e_alive = (uint8_t *)realloc(G->e_alive, (size_t)G->mcap);
  }
  const int64_t e = G->m++;
  G->es[e] = start; G->ee[e] = end; G->ed[e] = dist; G->e_valid[e] = valid_move ? 1 : 0; G->e_alive[e] = 1;
  iv_push(initial ? &G->out_init[start] : &G->out_cur[start], e);
  iv_push(initial ? &G->in_init[end] : &G->in_cur[end], e);
  return e;
}

void orc_graph_set_node(orc_graph *G, int64_t v, double lmc, double tree_cost) { G->lmc[v] = lmc; G->g[v] = tree_cost; }
void orc_graph_set_move_goal(orc_graph *G, int64_t v, int flag) { G->move_goal[v] = flag ? 1 : 0; }
void orc_graph_set_edge_dist(orc_graph *G, int64_t e, double dist) { G->ed[e] = dist; }
double orc_graph_lmc(const orc_graph *G, int64_t v) { return G->lmc[v]; }
double orc_graph_tree_cost(const orc_graph *G, int64_t v) { return G->g[v]; }
int64_t orc_graph_parent_edge(const orc_graph *G, int64_t v) { return G->parent_used[v] ? G->parent_edge[v] : -1; }
int64_t orc_graph_queue_length(const orc_graph *G) { return G->heap_last; }
int64_t orc_graph_n_edges(const orc_graph *G) { return G->m; }

/* ---- keys, R/DRRT_Q.jl:2052-2077 ---- */
static int less_q(const orc_graph *G, int64_t a, int64_t b) {
  const double ga = fmin(G->g[a], G->lmc[a]), gb = fmin(G->g[b], G->lmc[b]);     /* keyQ: (g_min + 0.0, g_min) */
  const double a1 = ga + 0.0, a2 = ga, b1 = gb + 0.0, b2 = gb;
  return (a1 < b1) || (a1 == b1 && a2 < b2) || (a1 == b1 && a2 == b2 && G->move_goal[a]);
}
static int greater_q(const orc_graph *G, int64_t a, int64_t b) {
  const double ga = fmin(G->g[a], G->lmc[a]), gb = fmin(G->g[b], G->lmc[b]);
  const double a1 = ga + 0.0, a2 = ga, b1 = gb + 0.0, b2 = gb;
  return (a1 > b1) || (a1 == b1 && a2 > b2) || (a1 == b1 && a2 == b2 && G->move_goal[b]);
}

/* ---- BinaryHeap, R/heap.jl:138-273 ---- */
static void bubble_up(orc_graph *G, int64_t n) {
  if (n == 1) return;
  int64_t parent = n / 2;
  while (n != 1 && greater_q(G, G->heap[parent], G->heap[n])) {
    const int64_t t = G->heap[parent]; G->heap[parent] = G->heap[n]; G->heap[n] = t;
    G->heap_index[G->heap[parent]] = parent;
    G->heap_index[G->heap[n]] = n;
    n = parent; parent = n / 2;
  }
}
static void bubble_down(orc_graph *G, int64_t n) {
  int64_t child;
  const int64_t last = G->heap_last, parent_of_last = last / 2;
  if (2 * n == last) child = 2 * n;
  else if (2 * n + 1 > last) return;
  else if (less_q(G, G->heap[2 * n], G->heap[2 * n + 1])) child = 2 * n;
  else child = 2 * n + 1;
  while (n <= parent_of_last && less_q(G, G->heap[child], G->heap[n])) {
    const int64_t t = G->heap[child]; G->heap[child] = G->heap[n]; G->heap[n] = t;
    G->heap_index[G->heap[child]] = child;
    G->heap_index[G->heap[n]] = n;
    n = child;
    if (2 * n == last) child = 2 * n;
    else if (2 * n + 1 > last) return;
    else if (less_q(G, G->heap[2 * n], G->heap[2 * n + 1])) child = 2 * n;
    else child = 2 * n + 1;
  }
}
static void add_to_heap(orc_graph *G, int64_t v) {
  if (G->heap_last == G->heap_cap) {
    G->heap_cap *= 2;
    G->heap = (int64_t *)realloc(G->heap, sizeof(int64_t) * (size_t)(G->heap_cap + 1));
  }
  if (G->in_q[v]) return;                       /* the reference crashes here on purpose (heap.jl:221-222) */
  G->heap_last += 1;
  G->heap[G->heap_last] = v;
  G->heap_index[v] = G->heap_last;
  bubble_up(G, G->heap_last);
  G->in_q[v] = 1;
}
static int64_t pop_heap(orc_graph *G) {
  const int64_t top = G->heap[1];
  G->heap[1] = G->heap[G->heap_last];
  G->heap_index[G->heap[1]] = 1;
  G->heap_last -= 1;
  bubble_down(G, 1);
  G->in_q[top] = 0;
  G->heap_index[top] = -1;
  return top;
}
static void remove_from_heap(orc_graph *G, int64_t v) {
  const int64_t n = G->heap_index[v];
  const int64_t moved = G->heap[G->heap_last];
  G->heap[n] = moved;
  G->heap_index[moved] = n;
  G->heap_last -= 1;
  bubble_up(G, n);
  bubble_down(G, G->heap_index[moved]);
  G->in_q[v] = 0;
  G->heap_index[v] = -1;
}
static void update_heap(orc_graph *G, int64_t v) {
  bubble_up(G, G->heap_index[v]);
  bubble_down(G, G->heap_index[v]);
}

/* verifyInQueue / verifyInOSQueue, R/DRRT_Q.jl:2364-2385 */
void orc_graph_verify_in_queue(orc_graph *G, int64_t v) {
  if (G->in_q[v]) update_heap(G, v); else add_to_heap(G, v);
}
void orc_graph_verify_in_os(orc_graph *G, int64_t v) {
  if (G->in_q[v]) { update_heap(G, v); remove_from_heap(G, v); }
  if (!G->in_os[v]) { G->in_os[v] = 1; iv_push(&G->os, v); G->os_len += 1; }
}

/* makeParentOf, R/DRRT_Q.jl:2459-2486 */
static void succ_remove(orc_graph *G, int64_t node) {
  const int64_t par = G->ee[G->parent_edge[node]];
  G->succ[par].v[G->succ_slot[node]] = -1;
}
void orc_graph_make_parent_of(orc_graph *G, int64_t new_parent, int64_t node, int64_t edge) {
  if (G->parent_used[node]) succ_remove(G, node);
  G->parent_edge[node] = edge;
  G->parent_used[node] = 1;
  iv_push(&G->succ[new_parent], node);
  G->succ_slot[node] = G->succ[new_parent].n - 1;
}

/* cullCurrentNeighbors, R/DRRT_Q.jl:2388-2403: current out-edges longer than the ball are dropped from both
 * ends' current lists (initial neighbours are kept) */
static void cull_current_neighbors(orc_graph *G, int64_t v, double ball) {
  ivec *L = &G->out_cur[v];
  for (int64_t k = L->n - 1; k >= 0; --k) {
    const int64_t e = L->v[k];
    if (e >= 0 && G->e_alive[e] && G->ed[e] > ball) G->e_alive[e] = 0;
  }
}

/* iteration of RRTNodeNeighborIterator: initial list front to back, then current list (R/DRRT_Q.jl:2408-2455) */
#define FOR_NEIGHBOR_EDGES(L_INIT, L_CUR, e_var, body)                          \
  for (int _pass = 0; _pass < 2; ++_pass) {                                      \
    const ivec *_L = _pass == 0 ? (L_INIT) : (L_CUR);                           \
    for (int64_t _k = _L->n - 1; _k >= 0; --_k) {                                \
      const int64_t e_var = _L->v[_k];                                          \
      if (e_var < 0 || !G->e_alive[e_var]) continue;                            \
      body                                                                      \
    }                                                                           \
  }

/* recalculateLMCMineVTwo, R/DRRT_Q.jl:2490-2541 */
static void recalculate_lmc(orc_graph *G, int64_t v, int64_t root, double ball) {
  if (v == root) return;
  int found = 0;
  int64_t best_parent = -1, best_edge = -1;
  cull_current_neighbors(G, v, ball);
  FOR_NEIGHBOR_EDGES(&G->out_init[v], &G->out_cur[v], e, {
    const int64_t u = G->ee[e];
    const double nd = G->ed[e];
    if (G->in_os[u]) continue;
    if (G->lmc[v] > G->lmc[u] + nd && (!G->parent_used[u] || G->ee[G->parent_edge[u]] != v) && G->e_valid[e]) {
      G->lmc[v] = G->lmc[u] + nd;
      best_parent = u; best_edge = e; found = 1;
    }
  })
  if (found) orc_graph_make_parent_of(G, best_parent, v, best_edge);
}

/* rewire, R/DRRT_Q.jl:2647-2700 */
static void rewire(orc_graph *G, int64_t v, int64_t root, double ball, double change_thresh) {
  (void)root;
  const double delta = G->g[v] - G->lmc[v];
  if (delta <= change_thresh) return;
  cull_current_neighbors(G, v, ball);
  FOR_NEIGHBOR_EDGES(&G->in_init[v], &G->in_cur[v], e, {
    const int64_t u = G->es[e];
    if ((G->parent_used[v] && G->ee[G->parent_edge[v]] == u) || !G->e_valid[e]) continue;
    const double dn = G->lmc[u] - (G->lmc[v] + G->ed[e]);
    if (dn > 0) {
      G->lmc[u] = G->lmc[v] + G->ed[e];
      if (!G->parent_used[u] || G->ee[G->parent_edge[u]] != v) orc_graph_make_parent_of(G, v, u, e);
      if (G->g[u] - G->lmc[u] > change_thresh) orc_graph_verify_in_queue(G, u);
    }
  })
}

/* reduceInconsistency, R/DRRT_Q.jl:2703-2717.  goal < 0: no goal node (the loop runs until the queue is empty,
 * as with a goal whose rrtLMC is Inf). */
void orc_graph_reduce_inconsistency(orc_graph *G, int64_t goal, int64_t root, double ball, double change_thresh) {
  while (G->heap_last > 0 &&
         (goal < 0 || less_q(G, G->heap[1], goal) || G->lmc[goal] == INFINITY || G->g[goal] == INFINITY || G->in_q[goal])) {
    const int64_t v = pop_heap(G);
    if (G->g[v] - G->lmc[v] > change_thresh) {
      recalculate_lmc(G, v, root, ball);
      rewire(G, v, root, ball, change_thresh);
    }
    G->g[v] = G->lmc[v];
  }
}

/* the edge loop of addNewObstacle for ONE blocked edge, R/DRRT_Q.jl:3248-3268: the edge costs Inf from now
 * on; if it was its start node's parent edge, that node loses its parent and goes to the orphan stack */
void orc_graph_block_edge(orc_graph *G, int64_t e) {
  G->ed[e] = INFINITY;
  const int64_t v = G->es[e];
  if (G->parent_used[v] && G->parent_edge[v] == e) {
    succ_remove(G, v);
    /* `thisNode.rrtParentEdge.endNode = thisNode` (:3258): the parent edge IS the neighbour edge object, so the
     * out-edge to the former parent becomes a self loop of cost Inf -- reproduced */
    G->ee[e] = v;
    G->parent_used[v] = 0;
    orc_graph_verify_in_os(G, v);
  }
}

/* propogateDescendants, R/DRRT_Q.jl:2724-2817 (the robot's move target is the caller's business) */
void orc_graph_propagate_descendants(orc_graph *G) {
  if (G->os_len <= 0) return;
  /* first pass: back to front, successors pushed to the front while walking */
  for (int64_t k = 0; k < G->os.n; ++k) {
    const int64_t v = G->os.v[k];
    const ivec *S = &G->succ[v];
    for (int64_t j = S->n - 1; j >= 0; --j)
      if (S->v[j] >= 0) orc_graph_verify_in_os(G, S->v[j]);
  }
  /* second pass: the out-neighbours (and the parent) of every orphan that are not orphans themselves */
  for (int64_t k = 0; k < G->os.n; ++k) {
    const int64_t v = G->os.v[k];
    FOR_NEIGHBOR_EDGES(&G->out_init[v], &G->out_cur[v], e, {
      const int64_t u = G->ee[e];
      if (G->in_os[u]) continue;
      G->g[u] = INFINITY;
      orc_graph_verify_in_queue(G, u);
    })
    if (G->parent_used[v] && !G->in_os[G->ee[G->parent_edge[v]]]) {
      const int64_t p = G->ee[G->parent_edge[v]];
      G->g[p] = INFINITY;
      orc_graph_verify_in_queue(G, p);
    }
  }
  /* third pass: pop from the front */
  while (G->os_len > 0) {
    const int64_t v = G->os.v[G->os.n - 1];
    G->os.n -= 1; G->os_len -= 1;
    G->in_os[v] = 0;
    if (G->parent_used[v]) {
      succ_remove(G, v);
      G->parent_used[v] = 0;
    }
    G->g[v] = INFINITY;
    G->lmc[v] = INFINITY;
  }
}
