/*
 * rrtx_oracle.c -- CPU restatement (plain C) of the RRT^X extend/rewire hot
 * path of jnetter6/RRTQX_3D.  TEST INFRASTRUCTURE ONLY -- see rrtx_oracle.h.
 *
 * Every function cites the reference lines it follows (R/ = code_RRTQx_3D/).
 * Julia semantics reproduced on purpose:
 *   - sum() of < 16 elements is a sequential left fold (Base reduce.jl);
 *   - x.^2 and x^2 are x*x; x^2.0 (one site, R/DRRT_DubinsEdge_functions.jl:367) is pow(x, 2.0), whose
 *     correctly rounded value is fl(x*x): written x*x here and on the device;
 *   - min/max propagate NaN and order signed zeros;
 *   - no implicit FMA (build with -ffp-contract=off);
 *   - a:s:b float ranges: literal-fallback length rule of Base (twiceprecision.jl).
 *
 * Transcendentals on the Dubins paths (sin, cos, atan, acos) are NOT libm calls: they come from
 * include/rrtx_detmath.h, one deterministic IEEE implementation compiled into this file and into
 * the HIP kernels alike, so that device and checker agree to the last bit where the reference's
 * branches (`theta < 0`, strict `bestDist > len`) decide on it.  What stays un-pinnable is the
 * distance of that implementation from Julia's libm (< 2 ulp against glibc, tests/test_detmath.py).
 * Build with -DORC_LIBM_TRIG (librrtx_oracle_libm.so) for the literal form -- glibc sin / cos / atan2 /
 * acos, one cos and one sin per arc row -- which the CPU tests hold against the default build within
 * rounding on poses in general position.
 */
#include "rrtx_oracle.h"

#include "../include/rrtx_detmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793 /* Float64(pi) */

#ifdef ORC_LIBM_TRIG
#define t_sin sin
#define t_cos cos
#define t_atan2 atan2
#define t_acos acos
#else
#define t_sin rrtx_dm_sin
#define t_cos rrtx_dm_cos
#define t_atan2 rrtx_dm_atan2
#define t_acos rrtx_dm_acos
#endif

/* Julia 1.0 Base.min/max for Float64 (base/math.jl) */
static double jl_min(double x, double y) {
  if ((y < x) || (signbit(y) > signbit(x))) return isnan(x) ? x : y;
  return isnan(y) ? y : x;
}
static double jl_max(double x, double y) {
  if ((y > x) || (signbit(y) < signbit(x))) return isnan(x) ? x : y;
  return isnan(y) ? y : x;
}

/* ------------------------------------------------------------------------ */
/* A1  euclidianDist(x,y) = sqrt(sum((x-y).^2))                              */
/*     R/DRRT_distance_functions.jl:37                                       */
/* ------------------------------------------------------------------------ */
double orc_euclid(const double *x, const double *y, int d) {
  double t = x[0] - y[0];
  double s = t * t;
  for (int i = 1; i < d; ++i) {
    t = x[i] - y[i];
    s = s + t * t;
  }
  return sqrt(s);
}

/* R/rrtqx.jl:382  min(delta, ballConstant*((log(1+n)/n)^(1/d))) */
double orc_ball_radius(double delta, double ball_constant, int64_t n, int d) {
  double nn = (double)n;
  double v = ball_constant * pow(log(1.0 + nn) / nn, 1.0 / (double)d);
  return jl_min(delta, v);
}

/* ------------------------------------------------------------------------ */
/* A2-A5  kd-tree                                                            */
/* ------------------------------------------------------------------------ */
#define ORC_MAX_WRAPS 8
#define ORC_MAX_DIM 8

struct orc_kd {
  int d;
  int64_t n, cap;
  double *pos;       /* n x d */
  int32_t *split;    /* 0-based split dimension */
  int64_t *parent, *cl, *cr; /* -1 = absent */
  uint8_t *in_heap;
  int nwraps;
  int wraps[ORC_MAX_WRAPS];
  double wrap_points[ORC_MAX_WRAPS];
};

struct orc_list {
  int64_t n, cap;
  int32_t *idx; /* stored in push order; the list front is the LAST pushed */
  double *key;
};

orc_kd *orc_kd_create(int d) {
  orc_kd *t = (orc_kd *)calloc(1, sizeof(orc_kd));
  t->d = d;
  return t;
}

void orc_kd_destroy(orc_kd *t) {
  if (!t) return;
  free(t->pos); free(t->split); free(t->parent); free(t->cl); free(t->cr); free(t->in_heap);
  free(t);
}

void orc_kd_set_wraps(orc_kd *t, int nwraps, const int *wraps, const double *wrap_points) {
  if (nwraps > ORC_MAX_WRAPS) nwraps = ORC_MAX_WRAPS;
  t->nwraps = nwraps;
  for (int i = 0; i < nwraps; ++i) { t->wraps[i] = wraps[i]; t->wrap_points[i] = wrap_points[i]; }
}

int64_t orc_kd_size(const orc_kd *t) { return t->n; }
const double *orc_kd_position(const orc_kd *t, int64_t idx) { return t->pos + idx * t->d; }

static void kd_grow(orc_kd *t) {
  int64_t nc = t->cap ? t->cap * 2 : 1024;
  t->pos = (double *)realloc(t->pos, sizeof(double) * nc * t->d);
  t->split = (int32_t *)realloc(t->split, sizeof(int32_t) * nc);
  t->parent = (int64_t *)realloc(t->parent, sizeof(int64_t) * nc);
  t->cl = (int64_t *)realloc(t->cl, sizeof(int64_t) * nc);
  t->cr = (int64_t *)realloc(t->cr, sizeof(int64_t) * nc);
  t->in_heap = (uint8_t *)realloc(t->in_heap, nc);
  t->cap = nc;
}

/* kdInsert, R/kdTree_general.jl:121-170 */
int64_t orc_kd_insert(orc_kd *t, const double *pos) {
  if (t->n == t->cap) kd_grow(t);
  int64_t me = t->n;
  memcpy(t->pos + me * t->d, pos, sizeof(double) * t->d);
  t->parent[me] = t->cl[me] = t->cr[me] = -1;
  t->in_heap[me] = 0;
  if (t->n == 0) {            /* :127-132 */
    t->split[me] = 0;
    t->n = 1;
    return me;
  }
  int64_t parent = 0;
  for (;;) {                  /* :136-160 */
    int s = t->split[parent];
    if (pos[s] < t->pos[parent * t->d + s]) {
      if (t->cl[parent] < 0) { t->cl[parent] = me; break; }
      parent = t->cl[parent];
    } else {
      if (t->cr[parent] < 0) { t->cr[parent] = me; break; }
      parent = t->cr[parent];
    }
  }
  t->parent[me] = parent;     /* :162-169 */
  t->split[me] = (t->split[parent] == t->d - 1) ? 0 : t->split[parent] + 1;
  t->n += 1;
  return me;
}

void orc_kd_insert_many(orc_kd *t, const double *pos, int64_t n) {
  for (int64_t i = 0; i < n; ++i) orc_kd_insert(t, pos + i * t->d);
}

static int64_t kd_depth_rec(const orc_kd *t, int64_t i) {
  /* iterative to survive degenerate trees */
  (void)i;
  int64_t best = 0;
  int64_t *depth = (int64_t *)malloc(sizeof(int64_t) * (t->n ? t->n : 1));
  for (int64_t k = 0; k < t->n; ++k) {
    depth[k] = (t->parent[k] < 0) ? 1 : depth[t->parent[k]] + 1; /* parents precede children */
    if (depth[k] > best) best = depth[k];
  }
  free(depth);
  return best;
}
int64_t orc_kd_depth(const orc_kd *t) { return kd_depth_rec(t, 0); }

#define KPOS(t, i) ((t)->pos + (i) * (t)->d)

/* kdFindNearestInSubtree, R/kdTree_general.jl:254-354 */
static void kd_nearest_in_subtree(orc_kd *t, int64_t root, const double *q,
                                  int64_t sug_node, double sug_dist,
                                  int64_t *out_node, double *out_dist) {
  int64_t parent = root;
  int64_t cur_node = sug_node;
  double cur_dist = sug_dist;
  for (;;) {                                   /* :262-280 descend */
    int s = t->split[parent];
    if (q[s] < KPOS(t, parent)[s]) {
      if (t->cl[parent] < 0) break;
      parent = t->cl[parent];
    } else {
      if (t->cr[parent] < 0) break;
      parent = t->cr[parent];
    }
  }
  double nd = orc_euclid(q, KPOS(t, parent), t->d); /* :282-286 */
  if (nd < cur_dist) { cur_node = parent; cur_dist = nd; }

  for (;;) {                                   /* :289-353 walk back up */
    int s = t->split[parent];
    double hyper = q[s] - KPOS(t, parent)[s];
    if (hyper > cur_dist) {                    /* :295-306 */
      if (parent == root) { *out_node = cur_node; *out_dist = cur_dist; return; }
      parent = t->parent[parent];
      continue;
    }
    if (cur_node != parent) {                  /* :312-318 */
      nd = orc_euclid(q, KPOS(t, parent), t->d);
      if (nd < cur_dist) { cur_node = parent; cur_dist = nd; }
    }
    if (q[s] < KPOS(t, parent)[s] && t->cr[parent] >= 0) {         /* :321-332 */
      int64_t rn; double rd;
      kd_nearest_in_subtree(t, t->cr[parent], q, cur_node, cur_dist, &rn, &rd);
      if (rd < cur_dist) { cur_dist = rd; cur_node = rn; }
    } else if (KPOS(t, parent)[s] <= q[s] && t->cl[parent] >= 0) { /* :334-345 */
      int64_t ln; double ld;
      kd_nearest_in_subtree(t, t->cl[parent], q, cur_node, cur_dist, &ln, &ld);
      if (ld < cur_dist) { cur_dist = ld; cur_node = ln; }
    }
    if (parent == root) { *out_node = cur_node; *out_dist = cur_dist; return; }
    parent = t->parent[parent];
  }
}

/* ghostPointIterator / getNextGhostPoint, R/ghostPoint.jl:32-111 */
typedef struct {
  const orc_kd *t;
  const double *q;
  int flags[ORC_MAX_WRAPS + 1]; /* 1-based like the reference */
  int depth;
  double ghost[16];
  double closest[16];
} ghost_iter;

static void ghost_init(ghost_iter *g, const orc_kd *t, const double *q) {
  g->t = t; g->q = q;
  for (int i = 0; i <= ORC_MAX_WRAPS; ++i) g->flags[i] = 0;
  g->depth = t->nwraps;
  for (int i = 0; i < t->d; ++i) { g->ghost[i] = q[i]; g->closest[i] = q[i]; }
}

/* returns 1 and fills g->ghost, or 0 when exhausted */
static int ghost_next(ghost_iter *g, double best_dist) {
  const orc_kd *t = g->t;
  for (;;) {
    while (g->depth > 0 && g->flags[g->depth] != 0) g->depth -= 1;   /* :67-69 */
    if (g->depth == 0) return 0;                                     /* :71-74 */
    g->flags[g->depth] = 1;                                          /* :77 */
    int wd = t->wraps[g->depth - 1];
    double wp = t->wrap_points[g->depth - 1];
    double dim_val = g->q[wd];                                       /* :80-89 */
    double dim_closest = 0.0;
    if (g->q[wd] < wp / 2.0) { dim_val += wp; dim_closest += wp; }
    else { dim_val -= wp; }
    g->ghost[wd] = dim_val;                                          /* :90-91 */
    g->closest[wd] = dim_closest;
    while (g->depth < t->nwraps) {                                   /* :96-101 */
      g->depth += 1;
      g->flags[g->depth] = 0;
      int wd2 = t->wraps[g->depth - 1];
      g->ghost[wd2] = g->q[wd2];
      g->closest[wd2] = g->ghost[wd2];
    }
    if (orc_euclid(g->closest, g->ghost, t->d) > best_dist) continue; /* :104-106 */
    return 1;
  }
}

int orc_ghost_points(const orc_kd *t, const double *q, double best_dist, int cap, double *out) {
  ghost_iter g;
  ghost_init(&g, t, q);
  int n = 0;
  while (ghost_next(&g, best_dist)) {
    if (n < cap) memcpy(out + (size_t)n * t->d, g.ghost, sizeof(double) * t->d);
    n++;
  }
  return n;
}

/* kdFindNearest, R/kdTree_general.jl:357-385 */
void orc_kd_nearest(orc_kd *t, const double *q, int64_t *idx, double *dist) {
  double d_root = orc_euclid(q, KPOS(t, 0), t->d);
  int64_t ln; double ld;
  kd_nearest_in_subtree(t, 0, q, 0, d_root, &ln, &ld);
  if (t->nwraps > 0) {
    ghost_iter g;
    ghost_init(&g, t, q);
    while (ghost_next(&g, ld)) {
      double dgr = orc_euclid(g.ghost, KPOS(t, 0), t->d);
      int64_t tn; double td;
      kd_nearest_in_subtree(t, 0, g.ghost, 0, dgr, &tn, &td);
      if (td < ld) { ld = td; ln = tn; }
    }
  }
  *idx = ln; *dist = ld;
}

/* kdFindNearestNaive, R/kdTree_general.jl:215-247: pre-order walk, strict <.
 * (ignores wraps, like the reference's naive version) */
void orc_kd_nearest_naive(orc_kd *t, const double *q, int64_t *idx, double *dist) {
  /* explicit stack pre-order: node, left subtree, right subtree */
  int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (t->n + 1));
  int64_t sp = 0;
  stack[sp++] = 0;
  double best = INFINITY; int64_t bn = 0; int first = 1;
  while (sp > 0) {
    int64_t i = stack[--sp];
    double dd = orc_euclid(q, KPOS(t, i), t->d);
    if (first || dd < best) { best = dd; bn = i; first = 0; }
    if (t->cr[i] >= 0) stack[sp++] = t->cr[i];
    if (t->cl[i] >= 0) stack[sp++] = t->cl[i];
  }
  free(stack);
  *idx = bn; *dist = best;
}

static orc_list *list_new(void) { return (orc_list *)calloc(1, sizeof(orc_list)); }

/* addToRangeList, R/kdTree_general.jl:765-771 (JlistPush = push to front) */
static void add_to_range_list(orc_kd *t, orc_list *l, int64_t node, double key) {
  if (t->in_heap[node]) return;
  t->in_heap[node] = 1;
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 64;
    l->idx = (int32_t *)realloc(l->idx, sizeof(int32_t) * l->cap);
    l->key = (double *)realloc(l->key, sizeof(double) * l->cap);
  }
  l->idx[l->n] = (int32_t)node;
  l->key[l->n] = key;
  l->n += 1;
}

/* kdFindWithinRangeInSubtree, R/kdTree_general.jl:800-884 */
static void kd_range_in_subtree(orc_kd *t, int64_t root, double range, const double *q, orc_list *l) {
  int64_t parent = root;
  for (;;) {                                   /* :805-823 */
    int s = t->split[parent];
    if (q[s] < KPOS(t, parent)[s]) {
      if (t->cl[parent] < 0) break;
      parent = t->cl[parent];
    } else {
      if (t->cr[parent] < 0) break;
      parent = t->cr[parent];
    }
  }
  double nd = orc_euclid(q, KPOS(t, parent), t->d); /* :829-832 */
  if (nd < range) add_to_range_list(t, l, parent, nd);

  for (;;) {                                   /* :835-883 */
    int s = t->split[parent];
    double hyper = q[s] - KPOS(t, parent)[s];
    if (hyper > range) {                       /* :842-853 */
      if (parent == root) return;
      parent = t->parent[parent];
      continue;
    }
    if (!t->in_heap[parent]) {                 /* :859-864 */
      nd = orc_euclid(q, KPOS(t, parent), t->d);
      if (nd < range) add_to_range_list(t, l, parent, nd);
    }
    if (q[s] < KPOS(t, parent)[s] && t->cr[parent] >= 0) {          /* :867-870 */
      kd_range_in_subtree(t, t->cr[parent], range, q, l);
    } else if (KPOS(t, parent)[s] <= q[s] && t->cl[parent] >= 0) {  /* :871-875 */
      kd_range_in_subtree(t, t->cl[parent], range, q, l);
    }
    if (parent == root) return;
    parent = t->parent[parent];
  }
}

/* kdFindMoreWithinRange, R/kdTree_general.jl:927-955 */
void orc_kd_find_more_within_range(orc_kd *t, double r, const double *q, orc_list *l) {
  double d_root = orc_euclid(q, KPOS(t, 0), t->d);
  if (d_root <= r) add_to_range_list(t, l, 0, d_root);   /* root uses <=  (:932) */
  kd_range_in_subtree(t, 0, r, q, l);
  if (t->nwraps > 0) {
    ghost_iter g;
    ghost_init(&g, t, q);
    while (ghost_next(&g, r)) kd_range_in_subtree(t, 0, r, g.ghost, l);
  }
}

/* kdFindWithinRange, R/kdTree_general.jl:889-919 */
orc_list *orc_kd_find_within_range(orc_kd *t, double r, const double *q) {
  orc_list *l = list_new();
  orc_kd_find_more_within_range(t, r, q, l);
  return l;
}

int64_t orc_list_length(const orc_list *l) { return l->n; }

int64_t orc_list_read(const orc_list *l, int64_t cap, int32_t *idx, double *key) {
  for (int64_t k = 0; k < l->n && k < cap; ++k) {
    int64_t src = l->n - 1 - k; /* front = last pushed */
    if (idx) idx[k] = l->idx[src];
    if (key) key[k] = l->key[src];
  }
  return l->n;
}

/* emptyRangeList, R/kdTree_general.jl:782-787 */
void orc_kd_empty_range_list(orc_kd *t, orc_list *l) {
  if (!l) return;
  for (int64_t k = 0; k < l->n; ++k) t->in_heap[l->idx[k]] = 0;
  free(l->idx); free(l->key); free(l);
}

/* naive scan: same inclusivity rules as kdFindWithinRange, ascending index.
 * Design follows the reference's commented differential test
 * (R/kdTree_general.jl:732-761, 1039-1148). */
int64_t orc_range_naive(orc_kd *t, double r, const double *q, int64_t cap, int32_t *idx, double *key) {
  int64_t n = t->n;
  double *k = (double *)malloc(sizeof(double) * (n ? n : 1));
  uint8_t *in = (uint8_t *)calloc(n ? n : 1, 1);
  for (int64_t i = 0; i < n; ++i) {
    double dd = orc_euclid(q, KPOS(t, i), t->d);
    if ((i == 0) ? (dd <= r) : (dd < r)) { in[i] = 1; k[i] = dd; }
  }
  if (t->nwraps > 0) {
    ghost_iter g;
    ghost_init(&g, t, q);
    while (ghost_next(&g, r)) {
      for (int64_t i = 0; i < n; ++i) {
        if (in[i]) continue;
        double dd = orc_euclid(g.ghost, KPOS(t, i), t->d);
        if (dd < r) { in[i] = 1; k[i] = dd; }
      }
    }
  }
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!in[i]) continue;
    if (cnt < cap) { if (idx) idx[cnt] = (int32_t)i; if (key) key[cnt] = k[i]; }
    cnt++;
  }
  free(k); free(in);
  return cnt;
}

/* ------------------------------------------------------------------------ */
/* A3b  k nearest (no caller in the reference; restated for completeness)     */
/* ------------------------------------------------------------------------ */

/* The "B" (max-on-top) binary heap of R/heap.jl:358-461 over (node, key) pairs;
 * node -1 is the dummy of R/kdTree_general.jl:704-706.  1-based like the source. */
typedef struct {
  int64_t *node; double *key;
  int64_t last, parent_of_last, cap;
  orc_kd *t; int dummy_marked;
} knn_heap;

static int kh_marked(const knn_heap *h, int64_t n) { return n < 0 ? h->dummy_marked : h->t->in_heap[n]; }
static void kh_mark(knn_heap *h, int64_t n, int v) { if (n < 0) h->dummy_marked = v; else h->t->in_heap[n] = (uint8_t)v; }
static void kh_swap(knn_heap *h, int64_t a, int64_t b) {
  int64_t tn = h->node[a]; h->node[a] = h->node[b]; h->node[b] = tn;
  double tk = h->key[a]; h->key[a] = h->key[b]; h->key[b] = tk;
}

static void kh_bubble_up(knn_heap *h, int64_t n) {          /* heap.jl:358-378 */
  if (n == 1) return;
  int64_t parent = n / 2;
  while (n != 1 && h->key[parent] < h->key[n]) {
    kh_swap(h, parent, n);
    n = parent; parent = n / 2;
  }
}

static void kh_bubble_down(knn_heap *h, int64_t n) {        /* heap.jl:383-419 */
  int64_t child;
  if (2 * n == h->last) child = 2 * n;
  else if (2 * n + 1 > h->last) return;
  else if (h->key[2 * n] > h->key[2 * n + 1]) child = 2 * n;
  else child = 2 * n + 1;
  while (n <= h->parent_of_last && h->key[child] > h->key[n]) {
    kh_swap(h, child, n);
    n = child;
    if (2 * n == h->last) child = 2 * n;
    else if (2 * n + 1 > h->last) return;
    else if (h->key[2 * n] > h->key[2 * n + 1]) child = 2 * n;
    else child = 2 * n + 1;
  }
}

static void kh_add(knn_heap *h, int64_t node, double key) { /* heap.jl:422-440 */
  if (h->last == h->cap) {
    h->cap *= 2;
    h->node = (int64_t *)realloc(h->node, sizeof(int64_t) * (h->cap + 1));
    h->key = (double *)realloc(h->key, sizeof(double) * (h->cap + 1));
  }
  if (!kh_marked(h, node)) {
    h->last += 1;
    h->parent_of_last = h->last / 2;
    h->node[h->last] = node; h->key[h->last] = key;
    kh_bubble_up(h, h->last);
    kh_mark(h, node, 1);
  }
}

static void kh_pop(knn_heap *h) {                           /* heap.jl:447-461 */
  if (h->last < 1) return;
  int64_t old = h->node[1];
  h->node[1] = h->node[h->last]; h->key[1] = h->key[h->last];
  h->last -= 1;
  h->parent_of_last = h->last / 2;
  kh_bubble_down(h, 1);
  kh_mark(h, old, 0);
}

/* addToKNNHeap, R/kdTree_general.jl:580-593 */
static void add_to_knn_heap(knn_heap *h, int64_t node, double key, int64_t k) {
  if (kh_marked(h, node)) return;
  if (h->last < k) kh_add(h, node, key);
  else if (h->key[1] > key) { kh_pop(h); kh_add(h, node, key); }
}

/* kdFindKNearestInSubtree, R/kdTree_general.jl:605-692 */
static void kd_knearest_in_subtree(orc_kd *t, int64_t root, int64_t k, const double *q, knn_heap *h) {
  int64_t parent = root;
  double worst = h->key[1];
  for (;;) {                                   /* :612-630 */
    int s = t->split[parent];
    if (q[s] < KPOS(t, parent)[s]) {
      if (t->cl[parent] < 0) break;
      parent = t->cl[parent];
    } else {
      if (t->cr[parent] < 0) break;
      parent = t->cr[parent];
    }
  }
  double nd = orc_euclid(q, KPOS(t, parent), t->d);   /* :632-636 */
  if (nd < worst) { add_to_knn_heap(h, parent, nd, k); worst = h->key[1]; }
  for (;;) {                                   /* :639-691 */
    int s = t->split[parent];
    double hyper = q[s] - KPOS(t, parent)[s];
    if (hyper > worst) {                       /* :646-657 */
      if (parent == root) return;
      parent = t->parent[parent];
      continue;
    }
    if (!t->in_heap[parent]) {                 /* :663-669 */
      nd = orc_euclid(q, KPOS(t, parent), t->d);
      if (nd < worst) { add_to_knn_heap(h, parent, nd, k); worst = h->key[1]; }
    }
    if (q[s] < KPOS(t, parent)[s] && t->cr[parent] >= 0) {          /* :672-676 */
      kd_knearest_in_subtree(t, t->cr[parent], k, q, h);
      worst = h->key[1];
    } else if (KPOS(t, parent)[s] <= q[s] && t->cl[parent] >= 0) {  /* :677-682 */
      kd_knearest_in_subtree(t, t->cl[parent], k, q, h);
      worst = h->key[1];
    }
    if (parent == root) return;
    parent = t->parent[parent];
  }
}

/* kdFindKNearest, R/kdTree_general.jl:696-723.  Writes the heap array front to
 * back (the order cleanHeapB hands out) and returns its length, or -1 where the
 * reference raises (wrapped spaces, :711-713).  The heap is seeded with the
 * root and an Inf-keyed dummy, so k = 1 ends with TWO nodes. */
int64_t orc_kd_knearest(orc_kd *t, int64_t k, const double *q, int64_t cap, int32_t *idx, double *key) {
  if (t->nwraps > 0) return -1;
  knn_heap h;
  h.cap = k > 0 ? k : 1;
  h.node = (int64_t *)malloc(sizeof(int64_t) * (h.cap + 1));
  h.key = (double *)malloc(sizeof(double) * (h.cap + 1));
  h.last = 0; h.parent_of_last = -1; h.t = t; h.dummy_marked = 0;
  kh_add(&h, 0, orc_euclid(q, KPOS(t, 0), t->d));   /* :699-701 */
  kh_add(&h, -1, INFINITY);                         /* :703-706 */
  kd_knearest_in_subtree(t, 0, k, q, &h);
  if (h.node[1] == -1) kh_pop(&h);                  /* :716-720 */
  int64_t n = h.last;
  for (int64_t i = 1; i <= n; ++i) {                /* cleanHeap, heap.jl:338-350 */
    if (i - 1 < cap) { if (idx) idx[i - 1] = (int32_t)h.node[i]; if (key) key[i - 1] = h.key[i]; }
    kh_mark(&h, h.node[i], 0);
  }
  free(h.node); free(h.key);
  return n;
}

/* kdFindKNearestNaive, R/kdTree_general.jl:563-574: every node into the heap, pop
 * down to k.  Returned in heap order like the above. */
int64_t orc_kd_knearest_naive(orc_kd *t, int64_t k, const double *q, int64_t cap, int32_t *idx, double *key) {
  knn_heap h;
  h.cap = 64;
  h.node = (int64_t *)malloc(sizeof(int64_t) * (h.cap + 1));
  h.key = (double *)malloc(sizeof(double) * (h.cap + 1));
  h.last = 0; h.parent_of_last = -1; h.t = t; h.dummy_marked = 0;
  for (int64_t i = 0; i < t->n; ++i) kh_add(&h, i, orc_euclid(q, KPOS(t, i), t->d));
  while (h.last > k) kh_pop(&h);
  int64_t n = h.last;
  for (int64_t i = 1; i <= n; ++i) {
    if (i - 1 < cap) { if (idx) idx[i - 1] = (int32_t)h.node[i]; if (key) key[i - 1] = h.key[i]; }
    kh_mark(&h, h.node[i], 0);
  }
  free(h.node); free(h.key);
  return n;
}

/* ------------------------------------------------------------------------ */
/* A9  sphere edge check                                                     */
/* ------------------------------------------------------------------------ */

/* distancePointToSegment, R/DRRT_Q.jl:1205-1210.  NOTE the reference divides
 * the dot product by edgeLen, not edgeLen^2 -- reproduced on purpose.
 * dot() is LinearAlgebra.dot (BLAS ddot, n=3): restated as the unfused left
 * fold (parity unpinned at the last bit, see header). */
double orc_distance_point_to_segment3(const double *c, const double *p0, const double *p1) {
  double edge_len = orc_euclid(p0, p1, 3);
  double a0 = c[0] - p0[0], a1 = c[1] - p0[1], a2 = c[2] - p0[2];
  double b0 = p1[0] - p0[0], b1 = p1[1] - p0[1], b2 = p1[2] - p0[2];
  double dot = (a0 * b0 + a1 * b1) + a2 * b2;
  double t = jl_max(0.0, jl_min(1.0, dot / edge_len));
  double q[3];
  q[0] = p0[0] + t * b0;
  q[1] = p0[1] + t * b1;
  q[2] = p0[2] + t * b2;
  return orc_euclid(c, q, 3);
}

/* explicitEdgeCheck3D, R/DRRT_Q.jl:1775-1795 */
int orc_edge_check_sphere(const orc_sphere *ob, const double *p0, const double *p1, double robot_radius) {
  if (ob->unused || ob->life_span <= 0) return 0;   /* (radius == NaN is always false) */
  double dist_s = orc_distance_point_to_segment3(ob->c, p0, p1);
  if (dist_s > (robot_radius + ob->radius)) return 0;
  return 1;
}

/* explicitEdgeCheck(C, edge), R/DRRT_Q.jl:1802-1826 (inWarmupTime=false) */
int orc_edge_check_spheres(const orc_sphere *obs, int m, const double *p0, const double *p1,
                           double robot_radius, int32_t *first_hit) {
  for (int i = 0; i < m; ++i) {
    if (orc_edge_check_sphere(&obs[i], p0, p1, robot_radius)) {
      if (first_hit) *first_hit = i;
      return 1;
    }
  }
  if (first_hit) *first_hit = -1;
  return 0;
}

/* A12  explicitPointCheck / explicitPointCheck3D over spheres,
 * R/DRRT_Q.jl:1402-1415 (quickCheck2D), :1463-1487, :1520-1590.
 * Wdist = euclidianDist(x[1:3], y[1:3]) (R/DRRT_SimpleEdge_functions.jl:61). */
int orc_point_check_spheres(const orc_sphere *obs, int m, const double *p, double robot_radius,
                            int quick, double *clearance) {
  if (quick) {
    for (int i = 0; i < m; ++i) {               /* quickCheck, :1434-1451 */
      const orc_sphere *ob = &obs[i];
      if (ob->unused || ob->life_span <= 0) continue;
      if (orc_euclid(ob->c, p, 3) > ob->radius) continue;
      if (clearance) *clearance = 0.0;
      return 1;
    }
  }
  double ret_cert = INFINITY;
  for (int i = 0; i < m; ++i) {
    const orc_sphere *ob = &obs[i];
    /* explicitPointCheck2D(ob, point, retCert, robotRadius), :1463-1487 */
    double this_cert = ret_cert;
    if (!(ob->unused || ob->life_span <= 0)) {
      double this_dist = orc_euclid(ob->c, p, 3) - robot_radius;
      if (!(this_dist - ob->radius > ret_cert)) {
        this_dist = this_dist - ob->radius;
        if (this_dist < 0.0) { if (clearance) *clearance = 0.0; return 1; }
        this_cert = jl_min(ret_cert, this_dist);
      }
    }
    if (this_cert < ret_cert) ret_cert = this_cert;
  }
  if (clearance) *clearance = ret_cert;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* A10  polygon obstacles (legacy 2-D path)                                  */
/* ------------------------------------------------------------------------ */

/* Obstacle(kind, polygon) ctor, R/DRRT_data_structures.jl:229-241 */
void orc_polygon_ctor(const double *v, int n, double *cx, double *cy, double *radius) {
  double maxx = v[0], minx = v[0], maxy = v[1], miny = v[1];
  for (int i = 1; i < n; ++i) {
    maxx = jl_max(maxx, v[2 * i]); minx = jl_min(minx, v[2 * i]);
    maxy = jl_max(maxy, v[2 * i + 1]); miny = jl_min(miny, v[2 * i + 1]);
  }
  double px = (maxx + minx) / 2.0, py = (maxy + miny) / 2.0;
  double best = -INFINITY;
  for (int i = 0; i < n; ++i) {
    double dx = v[2 * i] - px, dy = v[2 * i + 1] - py;
    double s = dx * dx + dy * dy;
    best = (i == 0) ? s : jl_max(best, s);
  }
  *cx = px; *cy = py; *radius = sqrt(best);
}

/* distanceSqrdPointToSegment, R/DRRT.jl:1060-1083 */
double orc_dist_sqrd_point_to_segment(const double *pt, const double *a, const double *b) {
  double vx = pt[0] - a[0];
  double vy = pt[1] - a[1];
  double ux = b[0] - a[0];
  double uy = b[1] - a[1];
  double det = vx * ux + vy * uy;
  if (det <= 0) {
    return vx * vx + vy * vy;
  } else {
    double len = ux * ux + uy * uy;
    if (det >= len) {
      double ex = b[0] - pt[0], ey = b[1] - pt[1];
      return ex * ex + ey * ey;
    } else {
      double cr = ux * vy - uy * vx;
      return (cr * cr) / len;
    }
  }
}

/* segmentDistSqrd, R/DRRT.jl:1144-1202 (copy at R/DRRT_Q.jl:1288-1346) */
double orc_segment_dist_sqrd(const double *PA, const double *PB, const double *QA, const double *QB) {
  int possible = 1;
  if (fabs(PB[0] - PA[0]) < .000001) {
    if ((QA[0] >= PA[0] && QB[0] >= PA[0]) || (QA[0] <= PA[0] && QB[0] <= PA[0])) possible = 0;
  } else {
    double m = (PB[1] - PA[1]) / (PB[0] - PA[0]);
    double diffA = (m * (QA[0] - PA[0]) + PA[1]) - QA[1];
    double diffB = (m * (QB[0] - PA[0]) + PA[1]) - QB[1];
    if ((diffA > 0.0 && diffB > 0.0) || (diffA < 0.0 && diffB < 0.0)) possible = 0;
  }
  if (possible) {
    if (fabs(QB[0] - QA[0]) < .000001) {
      if ((PA[0] >= QA[0] && PB[0] >= QA[0]) || (PA[0] <= QA[0] && PB[0] <= QA[0])) possible = 0;
    } else {
      double m = (QB[1] - QA[1]) / (QB[0] - QA[0]);
      double diffA = (m * (PA[0] - QA[0]) + QA[1]) - PA[1];
      double diffB = (m * (PB[0] - QA[0]) + QA[1]) - PB[1];
      if ((diffA > 0.0 && diffB > 0.0) || (diffA < 0.0 && diffB < 0.0)) possible = 0;
    }
  }
  if (possible) return 0.0;
  /* Julia min(a,b,c,d) = min(min(min(a,b),c),d) */
  double r = orc_dist_sqrd_point_to_segment(PA, QA, QB);
  r = jl_min(r, orc_dist_sqrd_point_to_segment(PB, QA, QB));
  r = jl_min(r, orc_dist_sqrd_point_to_segment(QA, PA, PB));
  r = jl_min(r, orc_dist_sqrd_point_to_segment(QB, PA, PB));
  return r;
}

/* pointInPolygon (MacMartin crossings), R/DRRT.jl:1009-1056 */
int orc_point_in_polygon(const double *pt, const double *v, int P) {
  if (P < 2) return 0;
  int crossings = 0;
  double sx = v[2 * (P - 1)], sy = v[2 * (P - 1) + 1];
  for (int i = 0; i < P; ++i) {
    double ex = v[2 * i], ey = v[2 * i + 1];
    if ((sy > pt[1] && ey < pt[1]) || (sy < pt[1] && ey > pt[1])) {
      if (sx > pt[0] && ex > pt[0]) {
        crossings += 1;
      } else if (sx < pt[0] && ex < pt[0]) {
        /* no crossing */
      } else {
        double T = 2 * jl_max(sx, ex);
        double x = (-((sx * ey - sy * ex) * (pt[0] - T)) + ((sx - ex) * (pt[0] * pt[1] - pt[1] * T))) /
                   ((sy - ey) * (pt[0] - T));
        if (x > pt[0]) crossings += 1;
      }
    }
    sx = ex; sy = ey;
  }
  return (crossings % 2) != 0;
}

/* distToPolygonSqrd, R/DRRT.jl:1087-1106 */
double orc_dist_to_polygon_sqrd(const double *pt, const double *v, int P) {
  double best = INFINITY;
  double s[2] = {v[2 * (P - 1)], v[2 * (P - 1) + 1]};
  for (int i = 0; i < P; ++i) {
    double e[2] = {v[2 * i], v[2 * i + 1]};
    double dd = orc_dist_sqrd_point_to_segment(pt, s, e);
    if (dd < best) best = dd;
    s[0] = e[0]; s[1] = e[1];
  }
  return best;
}

/* findIndexBeforeTime, R/DRRT_Q.jl:1351-1362 (= R/DRRT.jl:1207-1218); 1-based result, -1 for no path */
static int index_before_time(const double *path, int rows, double t) {
  if (rows < 1) return -1;
  int i = 0;
  while (i + 1 <= rows && path[3 * i + 2] < t) i += 1;
  return i;
}

/* findTransformObsToTimeOfPoint, R/DRRT_Q.jl:1367-1391: offset of a moving obstacle at time t */
static void transform_obs_to_time(const orc_polygon *ob, double t, double *dx, double *dy) {
  const double *path = ob->path;
  int before = index_before_time(path, ob->npath, t);
  if (before < 1) { *dx = path[0]; *dy = path[1]; return; }
  if (before == ob->npath) { *dx = path[3 * (before - 1)]; *dy = path[3 * (before - 1) + 1]; return; }
  const double *b = path + 3 * (before - 1), *a = path + 3 * before;
  double along = (t - b[2]) / (a[2] - b[2]);
  *dx = b[0] + along * (a[0] - b[0]);
  *dy = b[1] + along * (a[1] - b[1]);
}

/* explicitEdgeCheck2D, kinds 6 and 7 (R/DRRT_Q.jl:1699-1771 = R/DRRT.jl:1579-1651): the robot moves
 * start -> end in (x, y, time); every obstacle path segment that overlaps it in time is tested at the
 * time of closest approach of the two centres, against the bounding circle only. */
static int edge_check_moving(const orc_polygon *ob, const double *start, const double *end, double robot_radius) {
  const double *early, *late;
  if (start[2] < end[2]) { early = start; late = end; } else { late = start; early = end; }
  int first = index_before_time(ob->path, ob->npath, early[2]);
  if (first < 1) first = 1;
  int last = 1 + index_before_time(ob->path, ob->npath, late[2]);
  if (last > ob->npath) last = ob->npath;
  if (last <= first) return 0;
  for (int is = first; is <= last - 1; ++is) {
    const double *pa = ob->path + 3 * (is - 1), *pb = ob->path + 3 * is;
    double x_1 = early[0], y_1 = early[1], T_1 = early[2];
    double x_2 = pa[0] + ob->cx, y_2 = pa[1] + ob->cy, T_2 = pa[2];
    double m_x1 = (late[0] - x_1) / (late[2] - T_1);
    double m_y1 = (late[1] - y_1) / (late[2] - T_1);
    double m_x2 = ((pb[0] + ob->cx) - x_2) / (pb[2] - T_2);
    double m_y2 = ((pb[1] + ob->cy) - y_2) / (pb[2] - T_2);
    double num = (((m_x1 * m_x1) * T_1 + m_x2 * (((m_x2 * T_2) + x_1) - x_2)) -
                  m_x1 * (((m_x2 * (T_1 + T_2)) + x_1) - x_2)) +
                 (m_y1 - m_y2) * ((((m_y1 * T_1) - (m_y2 * T_2)) - y_1) + y_2);
    double den = (m_x1 - m_x2) * (m_x1 - m_x2) + (m_y1 - m_y2) * (m_y1 - m_y2);
    double T_c = num / den;
    if (T_c < jl_max(T_1, T_2)) T_c = jl_max(T_1, T_2);
    else if (T_c > jl_min(late[2], pb[2])) T_c = jl_min(late[2], pb[2]);
    double r_x = m_x1 * (T_c - T_1) + x_1, r_y = m_y1 * (T_c - T_1) + y_1;
    double o_x = m_x2 * (T_c - T_2) + x_2, o_y = m_y2 * (T_c - T_2) + y_2;
    double rr = ob->radius + robot_radius;
    if ((r_x - o_x) * (r_x - o_x) + (r_y - o_y) * (r_y - o_y) < rr * rr) return 1;
  }
  return 0;
}

/* explicitEdgeCheck2D, R/DRRT.jl:1523-1653 (kinds 1 and 3 read only coords [1:2]; 6/7 also [3] = time) */
int orc_edge_check_polygon(const orc_polygon *ob, const double *p0, const double *p1, double robot_radius) {
  if (ob->unused || ob->life_span <= 0) return 0;
  if (1 <= ob->kind && ob->kind <= 5) {
    double c[2] = {ob->cx, ob->cy};
    double dsq = orc_dist_sqrd_point_to_segment(c, p0, p1);
    double rr = robot_radius + ob->radius;
    if (dsq > rr * rr) return 0;
  }
  if (ob->kind == 1) return 1;
  if (ob->kind == 3) {
    int P = ob->nverts;
    if (P < 2) return 0;
    double A[2] = {ob->verts[2 * (P - 1)], ob->verts[2 * (P - 1) + 1]};
    for (int i = 0; i < P; ++i) {
      double B[2] = {ob->verts[2 * i], ob->verts[2 * i + 1]};
      if (orc_segment_dist_sqrd(p0, p1, A, B) < robot_radius * robot_radius) return 1;
      A[0] = B[0]; A[1] = B[1];
    }
  }
  if (ob->kind == 6 || ob->kind == 7) return edge_check_moving(ob, p0, p1, robot_radius);
  return 0;
}

/* explicitEdgeCheck(C, edge) over polygon list, R/DRRT.jl:1660-1678 */
int orc_edge_check_polygons(const orc_polygon *obs, int m, const double *p0, const double *p1,
                            double robot_radius, int32_t *first_hit) {
  for (int i = 0; i < m; ++i) {
    if (orc_edge_check_polygon(&obs[i], p0, p1, robot_radius)) {
      if (first_hit) *first_hit = i;
      return 1;
    }
  }
  if (first_hit) *first_hit = -1;
  return 0;
}

/* polygon[:, 1:2] = originalPolygon .+ (dx, dy)  (R/DRRT.jl:1301-1302, 1411-1412) */
static double *moved_polygon(const orc_polygon *ob, double dx, double dy) {
  double *v = (double *)malloc(sizeof(double) * 2 * (ob->nverts > 0 ? ob->nverts : 1));
  for (int i = 0; i < ob->nverts; ++i) { v[2 * i] = ob->verts[2 * i] + dx; v[2 * i + 1] = ob->verts[2 * i + 1] + dy; }
  return v;
}

/* explicitPointCheck over polygons: quickCheck pass (R/DRRT.jl:1258-1284) then
 * explicitPointCheck2D (:1340-1427), driver :1433-1465.  Wdist over [1:2]
 * (legacy 2-D / Dubins Wdist, R/DRRT_DubinsEdge_functions.jl:62). */
int orc_point_check_polygons(const orc_polygon *obs, int m, const double *p, double robot_radius,
                             double *clearance) {
  for (int i = 0; i < m; ++i) {
    const orc_polygon *ob = &obs[i];
    if (ob->unused || ob->life_span <= 0) continue;
    double c[2] = {ob->cx, ob->cy};
    if ((1 <= ob->kind && ob->kind <= 5) && orc_euclid(c, p, 2) > ob->radius) continue;
    if (ob->kind == 1) { if (clearance) *clearance = 0.0; return 1; }
    if (ob->kind == 3 && orc_point_in_polygon(p, ob->verts, ob->nverts)) {
      if (clearance) *clearance = 0.0;
      return 1;
    }
    if (ob->kind == 6 || ob->kind == 7) {           /* R/DRRT.jl:1289-1305 */
      double dx, dy;
      transform_obs_to_time(ob, p[2], &dx, &dy);
      double cm[2] = {ob->cx + dx, ob->cy + dy};
      if (orc_euclid(cm, p, 2) > ob->radius) continue;
      double *mv = moved_polygon(ob, dx, dy);
      int in = orc_point_in_polygon(p, mv, ob->nverts);
      free(mv);
      if (in) { if (clearance) *clearance = 0.0; return 1; }
    }
  }
  double ret_cert = INFINITY;
  for (int i = 0; i < m; ++i) {
    const orc_polygon *ob = &obs[i];
    double this_cert = ret_cert;
    if (!(ob->unused || ob->life_span <= 0)) {
      double c[2] = {ob->cx, ob->cy};
      double dx = 0.0, dy = 0.0;
      const int moving = (ob->kind == 6 || ob->kind == 7);
      if (moving) {                                  /* R/DRRT.jl:1395-1408 */
        transform_obs_to_time(ob, p[2], &dx, &dy);
        c[0] = ob->cx + dx; c[1] = ob->cy + dy;
      }
      double this_dist = orc_euclid(c, p, 2) - robot_radius;
      if (!(this_dist - ob->radius > ret_cert)) {
        if (moving) {                                /* :1410-1420 */
          double *mv = moved_polygon(ob, dx, dy);
          int in = orc_point_in_polygon(p, mv, ob->nverts);
          if (!in) this_dist = sqrt(orc_dist_to_polygon_sqrd(p, mv, ob->nverts)) - robot_radius;
          free(mv);
          if (in || this_dist < 0.0) { if (clearance) *clearance = 0.0; return 1; }
        } else if (ob->kind == 1) {
          this_dist = this_dist - ob->radius;
          if (this_dist < 0.0) { if (clearance) *clearance = 0.0; return 1; }
        } else if (ob->kind == 3) {
          if (orc_point_in_polygon(p, ob->verts, ob->nverts)) { if (clearance) *clearance = 0.0; return 1; }
          this_dist = sqrt(orc_dist_to_polygon_sqrd(p, ob->verts, ob->nverts)) - robot_radius;
          if (this_dist < 0.0) { if (clearance) *clearance = 0.0; return 1; }
        }
        this_cert = jl_min(ret_cert, this_dist);
      }
    }
    if (this_cert < ret_cert) ret_cert = this_cert;
  }
  if (clearance) *clearance = ret_cert;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* A14b  findPointsInConflictWithObstacle(S, KD, ob::Obstacle, root), R/DRRT.jl:3048-3125: the nodes whose   */
/* edges addNewObstacle / removeObstacle (:3127-3290) re-check against a polygon obstacle.                   */
/*   kinds 1-5, Euclidean space without time: range = robotRadius + delta + ob.radius around ob.position     */
/*     (the legacy planner is 2-D; a d = 3 tree is that space embedded at z = 0: the query gets a 0.0 third   */
/*     coordinate -- the reference's own 1x2 query would not broadcast against 1x3 node rows);                */
/*   kinds 1-5, Dubins space without time: query [x y 0.0 pi], range + pi (:3061-3065);                       */
/*   kinds 1-5 in a space with time: the reference raises (:3067) -> NULL;                                    */
/*   kinds 6-7: one query per path segment i -> j = i + 1 (a single row: i = j = 1) at                        */
/*     [ob.position 0.0] + (path[i, :] + path[j, :]) / 2.0 (x, y AND time), range = base +                    */
/*     euclidianDist(path[i, :], path[j, :]) / 2.0, both + pi / [.. pi] for the Dubins car; the first query   */
/*     makes the list, the others add to it (kdFindMoreWithinRange; inHeap keeps a node from entering twice).  */
/* Returns the range list (the caller empties it), NULL where the reference raises.                           */
/* ------------------------------------------------------------------------ */
orc_list *orc_find_points_in_conflict_polygon(orc_kd *t, const orc_polygon *ob, double robot_radius, double delta,
                                              int has_time, int has_theta) {
  const int d = t->d;
  double q[ORC_MAX_DIM];
  for (int k = 0; k < ORC_MAX_DIM; ++k) q[k] = 0.0;
  if (ob->kind >= 1 && ob->kind <= 5) {
    if (!has_time && !has_theta) {
      if (d != 2 && d != 3) return NULL;
      const double search_range = robot_radius + delta + ob->radius;
      q[0] = ob->cx; q[1] = ob->cy;
      return orc_kd_find_within_range(t, search_range, q);
    }
    if (!has_time && has_theta) {
      if (d != 4) return NULL;
      const double search_range = robot_radius + delta + ob->radius + ORC_PI;
      q[0] = ob->cx; q[1] = ob->cy; q[2] = 0.0; q[3] = ORC_PI;
      return orc_kd_find_within_range(t, search_range, q);
    }
    return NULL;                          /* error("this type of obstacle not coded for this type of space") */
  }
  if (ob->kind >= 6 && ob->kind <= 7) {
    if (ob->npath < 1 || d != (has_theta ? 4 : 3)) return NULL;
    const double base = robot_radius + delta + ob->radius;
    orc_list *L = NULL;
    for (int i = 0; i < ob->npath; ++i) {
      const int j = (ob->npath == 1) ? 0 : i + 1;
      const double *pi_ = ob->path + 3 * i, *pj = ob->path + 3 * j;
      /* queryPose = [ob.position 0.0] + val/2.0, val = path[i, :] + path[j, :] */
      q[0] = ob->cx + (pi_[0] + pj[0]) / 2.0;
      q[1] = ob->cy + (pi_[1] + pj[1]) / 2.0;
      q[2] = 0.0 + (pi_[2] + pj[2]) / 2.0;
      if (has_theta) q[3] = ORC_PI;
      double search_range = base + orc_euclid(pi_, pj, 3) / 2.0;
      if (has_theta) search_range += ORC_PI;
      if (i == 0) L = orc_kd_find_within_range(t, search_range, q);
      else orc_kd_find_more_within_range(t, search_range, q, L);
      if (j == ob->npath - 1) break;
    }
    return L;
  }
  return NULL;                            /* error("this case not coded yet") */
}

/* ------------------------------------------------------------------------ */
/* A7  Dubins steering                                                       */
/* ------------------------------------------------------------------------ */

/* rightTurnDist / leftTurnDist, R/DRRT_distance_functions.jl:62-80 */
static double right_turn_dist(const double *a, const double *b, const double *c, double r) {
  double theta = t_atan2(a[1] - c[1], a[0] - c[0]) - t_atan2(b[1] - c[1], b[0] - c[0]);
  if (theta < 0) theta = theta + 2 * ORC_PI;
  return theta * r;
}
static double left_turn_dist(const double *a, const double *b, const double *c, double r) {
  double theta = t_atan2(b[1] - c[1], b[0] - c[0]) - t_atan2(a[1] - c[1], a[0] - c[0]);
  if (theta < 0) theta = theta + 2 * ORC_PI;
  return theta * r;
}
static double seg_len2(const double *a, const double *b) {
  double dx = a[0] - b[0], dy = a[1] - b[1];
  return sqrt(dx * dx + dy * dy);
}

int orc_dm_eval(int op, const double *x, const double *y, int64_t n, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    switch (op) {
      case 0: out[i] = rrtx_dm_sin(x[i]); break;
      case 1: out[i] = rrtx_dm_cos(x[i]); break;
      case 2: out[i] = rrtx_dm_atan2(y[i], x[i]); break;
      default: out[i] = rrtx_dm_acos(x[i]); break;
    }
  }
#ifdef ORC_LIBM_TRIG
  return 1;
#else
  return 0;
#endif
}

/* Length of start:step:stop for Float64 in Julia's literal fallback branch
 * (base/twiceprecision.jl `(:)(start::T, step::T, stop::T)`), reached whenever
 * start/stop have no small exact rational form -- the case for atan outputs. */
int64_t orc_julia_range_len(double start, double step, double stop) {
  double lf = (stop - start) / step;
  int64_t len;
  if (lf < 0) {
    len = 0;
  } else if (lf == 0) {
    len = 1;
  } else {
    len = (int64_t)nearbyint(lf) + 1; /* round-half-even like Julia round(Int, x) */
    double stop2 = start + (double)(len - 1) * step;
    len -= ((start < stop && stop < stop2) ? 1 : 0) + ((start > stop && stop > stop2) ? 1 : 0);
  }
  return len;
}

typedef struct { double *xy; int cap; int n; } traj_sink;
static void sink_push(traj_sink *s, double x, double y) {
  if (s->xy && s->n < s->cap) { s->xy[2 * s->n] = x; s->xy[2 * s->n + 1] = y; }
  s->n++;
}
/* one arc: phis = (phi_end == phi_start) ? phi_start : collect(phi_start:step:phi_end);
 * x = cx .+ r*cos.(phis)  (R/DRRT_DubinsEdge_functions.jl:529-532 and siblings).  Row k is at
 * phi_start + k * step; its cos / sin are the shared definition of include/rrtx_detmath.h (one angle
 * addition on cos / sin of phi_start), the literal cos(phi), sin(phi) in the ORC_LIBM_TRIG build. */
static void sink_arc(traj_sink *s, const double *c, double r, double phi_start, double phi_end, double step) {
  const int64_t len = (phi_end == phi_start) ? 1 : orc_julia_range_len(phi_start, step, phi_end);
#ifdef ORC_LIBM_TRIG
  for (int64_t i = 0; i < len; ++i) {
    double phi = phi_start + (double)i * step;
    sink_push(s, c[0] + r * cos(phi), c[1] + r * sin(phi));
  }
#else
  static const double arc_cos[RRTX_DM_ARC_TAB] = RRTX_DM_ARC_COS_INIT;
  static const double arc_sin[RRTX_DM_ARC_TAB] = RRTX_DM_ARC_SIN_INIT;
  double a, b;
  rrtx_dm_sincos(phi_start, &b, &a);
  for (int64_t i = 0; i < len; ++i) {
    double ck, sk, x, y;
    if (i < RRTX_DM_ARC_TAB) { ck = arc_cos[i]; sk = arc_sin[i]; }
    else rrtx_dm_sincos((double)i * .1, &sk, &ck);
    rrtx_dm_arc_row(c[0], c[1], r, a, b, ck, sk, step < 0.0, &x, &y);
    sink_push(s, x, y);
  }
#endif
}

/* calculateTrajectory(S, edge::DubinsEdge), R/DRRT_DubinsEdge_functions.jl:329-709,
 * space without time (S.spaceHasTime == false). */
static void dubins_steer_pieces(const double *s, const double *g, double r_min, double *cost,
                                char *word, double *traj, int traj_cap, int *traj_len, int *piece_len) {
  if (piece_len) piece_len[0] = piece_len[1] = piece_len[2] = 0;
  const double il[2] = {s[0], s[1]};
  const double it = s[3];
  const double gl[2] = {g[0], g[1]};
  const double gt = g[3];

  /* circle centres, :348-357 */
  double irc[2] = {il[0] + r_min * t_cos(it - ORC_PI / 2.0), il[1] + r_min * t_sin(it - ORC_PI / 2.0)};
  double ilc[2] = {il[0] + r_min * t_cos(it + ORC_PI / 2.0), il[1] + r_min * t_sin(it + ORC_PI / 2.0)};
  double grc[2] = {gl[0] + r_min * t_cos(gt - ORC_PI / 2.0), gl[1] + r_min * t_sin(gt - ORC_PI / 2.0)};
  double glc[2] = {gl[0] + r_min * t_cos(gt + ORC_PI / 2.0), gl[1] + r_min * t_sin(gt + ORC_PI / 2.0)};

  double best = INFINITY;
  const char *best_type = "xxx";
  double D, v[2], R, sq, a, b, first, second, third;

  /* rsl, :367-388 */
  double rsl_t1[2] = {NAN, NAN}, rsl_t2[2] = {NAN, NAN};
  D = sqrt((glc[0] - irc[0]) * (glc[0] - irc[0]) + (glc[1] - irc[1]) * (glc[1] - irc[1])); /* ^2.0, see header */
  v[0] = (glc[0] - irc[0]) / D; v[1] = (glc[1] - irc[1]) / D;
  R = -2.0 * r_min / D;
  if (!(fabs(R) > 1.0)) {
    sq = sqrt(1.0 - R * R);
    a = r_min * (R * v[0] + v[1] * sq);
    b = r_min * (R * v[1] - v[0] * sq);
    rsl_t1[0] = irc[0] - a; rsl_t2[0] = glc[0] + a;
    rsl_t1[1] = irc[1] - b; rsl_t2[1] = glc[1] + b;
    first = right_turn_dist(il, rsl_t1, irc, r_min);
    second = seg_len2(rsl_t2, rsl_t1);
    third = left_turn_dist(rsl_t2, gl, glc, r_min);
    double len = first + second + third;
    if (best > len) { best = len; best_type = "rsl"; }
  }

  /* rsr, :394-407 */
  double rsr_t1[2], rsr_t2[2];
  D = sqrt((grc[0] - irc[0]) * (grc[0] - irc[0]) + (grc[1] - irc[1]) * (grc[1] - irc[1]));
  v[0] = (grc[0] - irc[0]) / D; v[1] = (grc[1] - irc[1]) / D;
  rsr_t1[0] = -r_min * v[1] + irc[0]; rsr_t2[0] = -r_min * v[1] + grc[0];
  rsr_t1[1] = r_min * v[0] + irc[1];  rsr_t2[1] = r_min * v[0] + grc[1];
  first = right_turn_dist(il, rsr_t1, irc, r_min);
  second = seg_len2(rsr_t2, rsr_t1);
  third = right_turn_dist(rsr_t2, gl, grc, r_min);
  {
    double len = first + second + third;
    if (best > len) { best = len; best_type = "rsr"; }
  }

  /* rlr, :411-431 (D, v from rsr) */
  double rlr_rl[2] = {NAN, NAN}, rlr_lr[2] = {NAN, NAN}, rlr_c[2] = {NAN, NAN};
  if (D < 4.0 * r_min) {
    double theta = -t_acos(D / (4 * r_min)) + t_atan2(v[1], v[0]);
    rlr_c[0] = irc[0] + 2 * r_min * t_cos(theta);
    rlr_c[1] = irc[1] + 2 * r_min * t_sin(theta);
    rlr_rl[0] = (rlr_c[0] + irc[0]) / 2.0; rlr_rl[1] = (rlr_c[1] + irc[1]) / 2.0;
    rlr_lr[0] = (rlr_c[0] + grc[0]) / 2.0; rlr_lr[1] = (rlr_c[1] + grc[1]) / 2.0;
    first = right_turn_dist(il, rlr_rl, irc, r_min);
    second = left_turn_dist(rlr_rl, rlr_lr, rlr_c, r_min);
    third = right_turn_dist(rlr_lr, gl, grc, r_min);
    double len = first + second + third;
    if (best > len) { best = len; best_type = "rlr"; }
  }

  /* lsr, :436-458 */
  double lsr_t1[2] = {NAN, NAN}, lsr_t2[2] = {NAN, NAN};
  D = sqrt((grc[0] - ilc[0]) * (grc[0] - ilc[0]) + (grc[1] - ilc[1]) * (grc[1] - ilc[1]));
  v[0] = (grc[0] - ilc[0]) / D; v[1] = (grc[1] - ilc[1]) / D;
  R = 2.0 * r_min / D;
  if (!(fabs(R) > 1)) {
    sq = sqrt(1 - R * R);
    a = R * v[0] + v[1] * sq;
    b = R * v[1] - v[0] * sq;
    lsr_t1[0] = ilc[0] + a * r_min; lsr_t2[0] = grc[0] - a * r_min;
    lsr_t1[1] = ilc[1] + b * r_min; lsr_t2[1] = grc[1] - b * r_min;
    first = left_turn_dist(il, lsr_t1, ilc, r_min);
    second = seg_len2(lsr_t2, lsr_t1);
    third = right_turn_dist(lsr_t2, gl, grc, r_min);
    double len = first + second + third;
    if (best > len) { best = len; best_type = "lsr"; }
  }

  /* lsl, :464-477 */
  double lsl_t1[2], lsl_t2[2];
  D = sqrt((glc[0] - ilc[0]) * (glc[0] - ilc[0]) + (glc[1] - ilc[1]) * (glc[1] - ilc[1]));
  v[0] = (glc[0] - ilc[0]) / D; v[1] = (glc[1] - ilc[1]) / D;
  lsl_t1[0] = r_min * v[1] + ilc[0];  lsl_t2[0] = r_min * v[1] + glc[0];
  lsl_t1[1] = -r_min * v[0] + ilc[1]; lsl_t2[1] = -r_min * v[0] + glc[1];
  first = left_turn_dist(il, lsl_t1, ilc, r_min);
  second = seg_len2(lsl_t2, lsl_t1);
  third = left_turn_dist(lsl_t2, gl, glc, r_min);
  {
    double len = first + second + third;
    if (best > len) { best = len; best_type = "lsl"; }
  }

  /* lrl, :481-501 (D, v from lsl) */
  double lrl_rl[2] = {NAN, NAN}, lrl_lr[2] = {NAN, NAN}, lrl_c[2] = {NAN, NAN};
  if (D < 4.0 * r_min) {
    double theta = t_acos(D / (4 * r_min)) + t_atan2(v[1], v[0]);
    lrl_c[0] = ilc[0] + 2.0 * r_min * t_cos(theta);
    lrl_c[1] = ilc[1] + 2.0 * r_min * t_sin(theta);
    lrl_lr[0] = (lrl_c[0] + ilc[0]) / 2.0; lrl_lr[1] = (lrl_c[1] + ilc[1]) / 2.0;
    lrl_rl[0] = (lrl_c[0] + glc[0]) / 2.0; lrl_rl[1] = (lrl_c[1] + glc[1]) / 2.0;
    first = left_turn_dist(il, lrl_lr, ilc, r_min);
    second = right_turn_dist(lrl_lr, lrl_rl, lrl_c, r_min);
    third = left_turn_dist(lrl_rl, gl, glc, r_min);
    double len = first + second + third;
    if (best > len) { best = len; best_type = "lrl"; }
  }

  *cost = best;                       /* edge.Wdist = edge.dist = bestDist, :659,:698 */
  if (word) { memcpy(word, best_type, 3); word[3] = 0; }

  traj_sink sink = {traj, traj_cap, 0};
  if (best == INFINITY || best_type[0] == 'x') { /* :661-662: no trajectory is built */
    if (traj_len) *traj_len = 0;
    return;
  }

  const double delta_phi = .1;        /* :506 */
  const int is_rsl = !strcmp(best_type, "rsl"), is_rsr = !strcmp(best_type, "rsr");
  const int is_lsl = !strcmp(best_type, "lsl"), is_lsr = !strcmp(best_type, "lsr");
  double phi_start, phi_end;
  const double *p;

  /* first piece, :511-555 */
  if (best_type[0] == 'r') {
    p = is_rsl ? rsl_t1 : (is_rsr ? rsr_t1 : rlr_rl);
    phi_start = t_atan2(il[1] - irc[1], il[0] - irc[0]);
    phi_end = t_atan2(p[1] - irc[1], p[0] - irc[0]);
    if (phi_end > phi_start) phi_end = phi_end - 2.0 * ORC_PI;
    sink_arc(&sink, irc, r_min, phi_start, phi_end, -delta_phi);
  } else {
    p = is_lsl ? lsl_t1 : (is_lsr ? lsr_t1 : lrl_lr);
    phi_start = t_atan2(il[1] - ilc[1], il[0] - ilc[0]);
    phi_end = t_atan2(p[1] - ilc[1], p[0] - ilc[0]);
    if (phi_end < phi_start) phi_end = phi_end + 2.0 * ORC_PI;
    sink_arc(&sink, ilc, r_min, phi_start, phi_end, delta_phi);
  }
  if (piece_len) piece_len[0] = sink.n;

  /* second piece, :559-608 */
  if (best_type[1] == 's') {
    const double *p1 = is_lsr ? lsr_t1 : (is_lsl ? lsl_t1 : (is_rsr ? rsr_t1 : rsl_t1));
    const double *p2 = is_lsr ? lsr_t2 : (is_lsl ? lsl_t2 : (is_rsr ? rsr_t2 : rsl_t2));
    sink_push(&sink, p1[0], p1[1]);
    sink_push(&sink, p2[0], p2[1]);
  } else if (best_type[1] == 'r') {   /* lrl */
    phi_start = t_atan2(lrl_lr[1] - lrl_c[1], lrl_lr[0] - lrl_c[0]);
    phi_end = t_atan2(lrl_rl[1] - lrl_c[1], lrl_rl[0] - lrl_c[0]);
    if (phi_end > phi_start) phi_end = phi_end - 2.0 * ORC_PI;
    sink_arc(&sink, lrl_c, r_min, phi_start, phi_end, -delta_phi);
  } else {                            /* rlr */
    phi_start = t_atan2(rlr_rl[1] - rlr_c[1], rlr_rl[0] - rlr_c[0]);
    phi_end = t_atan2(rlr_lr[1] - rlr_c[1], rlr_lr[0] - rlr_c[0]);
    if (phi_end < phi_start) phi_end = phi_end + 2.0 * ORC_PI;
    sink_arc(&sink, rlr_c, r_min, phi_start, phi_end, delta_phi);
  }
  if (piece_len) piece_len[1] = sink.n - piece_len[0];

  /* third piece, :611-655 */
  if (best_type[2] == 'r') {
    p = is_rsr ? rsr_t2 : (is_lsr ? lsr_t2 : rlr_lr);
    phi_start = t_atan2(p[1] - grc[1], p[0] - grc[0]);
    phi_end = t_atan2(gl[1] - grc[1], gl[0] - grc[0]);
    if (phi_end > phi_start) phi_end = phi_end - 2.0 * ORC_PI;
    sink_arc(&sink, grc, r_min, phi_start, phi_end, -delta_phi);
  } else {
    p = is_lsl ? lsl_t2 : (is_rsl ? rsl_t2 : lrl_rl);
    phi_start = t_atan2(p[1] - glc[1], p[0] - glc[0]);
    phi_end = t_atan2(gl[1] - glc[1], gl[0] - glc[0]);
    if (phi_end < phi_start) phi_end = phi_end + 2.0 * ORC_PI;
    sink_arc(&sink, glc, r_min, phi_start, phi_end, delta_phi);
  }
  if (piece_len) piece_len[2] = sink.n - piece_len[0] - piece_len[1];
  if (traj_len) *traj_len = sink.n;
}
void orc_dubins_steer(const double *s, const double *g, double r_min, double *cost,
                      char *word, double *traj, int traj_cap, int *traj_len) {
  dubins_steer_pieces(s, g, r_min, cost, word, traj, traj_cap, traj_len, NULL);
}

/* explicitEdgeCheck(S, edge::DubinsEdge, obstacle) over the obstacle list,
 * R/DRRT_DubinsEdge_functions.jl:750-774 + R/DRRT.jl:1660-1678 */
int orc_dubins_edge_check_polygons(const orc_polygon *obs, int m, const double *s, const double *g,
                                   const double *traj, int traj_len, double robot_radius,
                                   double r_min, int32_t *first_hit) {
  for (int i = 0; i < m; ++i)
    if ((obs[i].kind == 6 || obs[i].kind == 7) && !(obs[i].unused || obs[i].life_span <= 0)) return -1;
  for (int i = 0; i < m; ++i) {
    const orc_polygon *ob = &obs[i];
    if (!orc_edge_check_polygon(ob, s, g, robot_radius + 2 * r_min)) continue;
    for (int k = 1; k < traj_len; ++k) {
      if (orc_edge_check_polygon(ob, traj + 2 * (k - 1), traj + 2 * k, robot_radius)) {
        if (first_hit) *first_hit = i;
        return 1;
      }
    }
  }
  if (first_hit) *first_hit = -1;
  return 0;
}

/* calculateTrajectory(S, edge::DubinsEdge) with S.spaceHasTime (R/DRRT_DubinsEdge_functions.jl:660-697):
 * the same six-word steering in (x, y); edge.Wdist = bestDist, edge.dist = sqrt(bestDist^2 + dt^2)
 * (:667, `^2` on a Float64 is x*x), edge.velocity = Wdist / dt with dt = start time - end time (:682;
 * planning runs in reverse time, the start of an edge is LATER than its end), and the trajectory gets a
 * third column: row 1 carries the start time, rows 2 .. P-1 the start time minus the distance walked
 * so far (sum of the straight pieces between stored rows, accumulated left to right, :691-695) over the
 * velocity, and the last row is overwritten with the end node's (x, y, t) (:696).  traj3 (may be NULL)
 * receives up to traj_cap rows of (x, y, t). */
static void dubins_steer_time_impl(const double *s, const double *g, double r_min, double *dist, double *wdist,
                                   double *velocity, char *word, double *traj3, int traj_cap, int *traj_len,
                                   int piecewise) {
  double best;
  int n = 0, plen[3];
  double *xy = (double *)malloc(sizeof(double) * 2 * (size_t)(traj_cap > 0 ? traj_cap : 1));
  dubins_steer_pieces(s, g, r_min, &best, word, xy, traj_cap, &n, plen);
  *wdist = best;
  if (best == INFINITY) {                 /* :661-662: no trajectory, no velocity */
    *dist = INFINITY;
    if (velocity) *velocity = NAN;
    if (traj_len) *traj_len = 0;
    free(xy);
    return;
  }
  const double dt = s[2] - g[2];
  *dist = sqrt(best * best + dt * dt);
  const double vel = best / dt;
  if (velocity) *velocity = vel;
  if (traj_len) *traj_len = n;
  if (traj3 && n > 0) {
    const int rows = n < traj_cap ? n : traj_cap;
    for (int i = 0; i < rows; ++i) { traj3[3 * i] = xy[2 * i]; traj3[3 * i + 1] = xy[2 * i + 1]; traj3[3 * i + 2] = 0.0; }
    traj3[2] = s[2];
    if (!piecewise) {
      double cumulative = 0.0;
      for (int i = 1; i < rows - 1; ++i) {
        cumulative += orc_euclid(xy + 2 * (i - 1), xy + 2 * i, 2);
        traj3[3 * i + 2] = s[2] - cumulative / vel;
      }
    } else if (n <= traj_cap) {
      /* distance walked at row k of a piece = (distance at the piece's first row) + k x (the piece's first
       * chord); the junctions between pieces are measured once */
      double run = 0.0;
      int first = 0, have_last = 0, last = 0;
      for (int pi = 0; pi < 3; ++pi) {
        if (plen[pi] <= 0) continue;
        if (have_last) run = run + orc_euclid(xy + 2 * last, xy + 2 * first, 2);
        const double cum0 = run;
        double chord = 0.0;
        if (plen[pi] > 1) {
          chord = orc_euclid(xy + 2 * first, xy + 2 * (first + 1), 2);
          run = run + (double)(plen[pi] - 1) * chord;
        }
        for (int k = 0; k < plen[pi]; ++k) {
          const int row = first + k;
          if (row > 0 && row < n - 1) traj3[3 * row + 2] = s[2] - (cum0 + (double)k * chord) / vel;
        }
        last = first + plen[pi] - 1;
        first += plen[pi];
        have_last = 1;
      }
    }
    if (n <= traj_cap) { traj3[3 * (n - 1)] = g[0]; traj3[3 * (n - 1) + 1] = g[1]; traj3[3 * (n - 1) + 2] = g[2]; }
  }
  free(xy);
}
void orc_dubins_steer_time(const double *s, const double *g, double r_min, double *dist, double *wdist,
                           double *velocity, char *word, double *traj3, int traj_cap, int *traj_len) {
  dubins_steer_time_impl(s, g, r_min, dist, wdist, velocity, word, traj3, traj_cap, traj_len, 0);
}
/* The same edge with the time column as the HIP kernels form it (they cannot afford a running sum over up to
 * 190 rows per directed edge): inside a piece every step is the piece's first chord, so the distance walked at
 * row k of a piece is (distance at its first row) + k * chord.  Differs from the reference's left-to-right sum
 * (above) by rounding only -- tests/test_oracle_kat.py bounds the difference at 1e-12 relative and requires the
 * same collision booleans on the suite's scenes; the device is compared with THIS form bit for bit. */
void orc_dubins_steer_time_pw(const double *s, const double *g, double r_min, double *dist, double *wdist,
                              double *velocity, char *word, double *traj3, int traj_cap, int *traj_len) {
  dubins_steer_time_impl(s, g, r_min, dist, wdist, velocity, word, traj3, traj_cap, traj_len, 1);
}

/* validMove(S, edge::DubinsEdge) with S.spaceHasTime, R/DRRT_DubinsEdge_functions.jl:115-121 */
int orc_dubins_valid_move_time(const double *s, const double *g, double velocity, double v_min, double v_max) {
  return (s[2] > g[2]) && (v_min <= velocity && velocity <= v_max);
}

/* explicitEdgeCheck(S, edge::DubinsEdge, obstacle) over the list in a space with time (:750-774):
 * stage 1 the chord start -> end ([x y t theta]: the moving kinds read time from [3]) with
 * robotRadius + 2 minTurningRadius, stage 2 every stored piece trajectory[i-1, :] -> [i, :] (rows of
 * x, y, t) with robotRadius. */
int orc_dubins_edge_check_polygons_time(const orc_polygon *obs, int m, const double *s, const double *g,
                                        const double *traj3, int traj_len, double robot_radius, double r_min,
                                        int32_t *first_hit) {
  for (int i = 0; i < m; ++i) {
    const orc_polygon *ob = &obs[i];
    if (!orc_edge_check_polygon(ob, s, g, robot_radius + 2 * r_min)) continue;
    for (int k = 1; k < traj_len; ++k) {
      if (orc_edge_check_polygon(ob, traj3 + 3 * (k - 1), traj3 + 3 * k, robot_radius)) {
        if (first_hit) *first_hit = i;
        return 1;
      }
    }
  }
  if (first_hit) *first_hit = -1;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline: the per-sample inner loop of extend/findBestParent          */
/* R/rrtqx.jl:926 (kdFindNearest), R/DRRT_Q.jl:2551 (kdFindWithinRange),     */
/* :1951-1963 (newNode->near: calculateTrajectory + explicitEdgeCheck),      */
/* :2600-2602 (near->newNode).  SimpleEdge: cost = dist(start, end).         */
/* ------------------------------------------------------------------------ */
int64_t orc_extend_batch_spheres(orc_kd *t, const orc_sphere *obs, int m, const double *queries,
                                 int64_t nq, double r, double robot_radius,
                                 int64_t *nearest_idx, int64_t *n_neighbors_total,
                                 int64_t *n_hits_total) {
  int64_t edges = 0, neigh = 0, hits = 0;
  volatile double sink = 0.0;
  for (int64_t i = 0; i < nq; ++i) {
    const double *q = queries + i * t->d;
    int64_t ni; double nd;
    orc_kd_nearest(t, q, &ni, &nd);
    if (nearest_idx) nearest_idx[i] = ni;
    orc_list *l = orc_kd_find_within_range(t, r, q);
    for (int64_t k = l->n - 1; k >= 0; --k) {      /* front -> back */
      const double *pn = KPOS(t, l->idx[k]);
      double c_out = orc_euclid(q, pn, t->d);      /* calculateTrajectory newNode->near */
      hits += orc_edge_check_spheres(obs, m, q, pn, robot_radius, NULL);
      double c_in = orc_euclid(pn, q, t->d);       /* calculateTrajectory near->newNode */
      hits += orc_edge_check_spheres(obs, m, pn, q, robot_radius, NULL);
      sink += c_out + c_in;
      edges += 2;
    }
    neigh += l->n;
    orc_kd_empty_range_list(t, l);
  }
  (void)sink;
  if (n_neighbors_total) *n_neighbors_total = neigh;
  if (n_hits_total) *n_hits_total = hits;
  return edges;
}

/* the same loop with CSpace.obstacles a list of polygon Obstacles: explicitEdgeCheck over the list with the first
 * hit's early return (R/DRRT.jl:1660-1678), explicitPointCheck of the sample (R/DRRT.jl:1434-1470) */
int64_t orc_extend_batch_polygons(orc_kd *t, const orc_polygon *obs, int m, const double *queries, int64_t nq, double r,
                                  double robot_radius, int64_t *nearest_idx, int64_t *n_neighbors_total,
                                  int64_t *n_hits_total) {
  int64_t edges = 0, neigh = 0, hits = 0;
  volatile double sink = 0.0;
  for (int64_t i = 0; i < nq; ++i) {
    const double *q = queries + i * t->d;
    int64_t ni; double nd, cl;
    orc_kd_nearest(t, q, &ni, &nd);
    if (nearest_idx) nearest_idx[i] = ni;
    hits += orc_point_check_polygons(obs, m, q, robot_radius, &cl);
    orc_list *l = orc_kd_find_within_range(t, r, q);
    for (int64_t k = l->n - 1; k >= 0; --k) {
      const double *pn = KPOS(t, l->idx[k]);
      double c_out = orc_euclid(q, pn, t->d);
      hits += orc_edge_check_polygons(obs, m, q, pn, robot_radius, NULL);
      double c_in = orc_euclid(pn, q, t->d);
      hits += orc_edge_check_polygons(obs, m, pn, q, robot_radius, NULL);
      sink += c_out + c_in;
      edges += 2;
    }
    neigh += l->n;
    orc_kd_empty_range_list(t, l);
  }
  (void)sink;
  if (n_neighbors_total) *n_neighbors_total = neigh;
  if (n_hits_total) *n_hits_total = hits;
  return edges;
}
