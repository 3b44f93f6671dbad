/*
 * rrtx_oracle.h -- CPU restatement of the RRT^X extend/rewire hot path of
 * jnetter6/RRTQX_3D (R/ = code_RRTQx_3D/).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (rrtqx_3d_amd/, librrtx_hip.so) never links or calls it.
 *
 * Parity status: the reference is Julia and cannot be executed in the build
 * container (no julia binary, no network); the reference holds no golden
 * vectors or asserting tests for this path.  The oracle is therefore pinned by
 * (1) the hand-derived known-answer tests K1-K10 of SURVEY.md section 8(c) and
 * (2) the reference's own differential design (kd-tree vs naive scan,
 * R/kdTree_general.jl:1039-1148).  LinearAlgebra.dot (OpenBLAS ddot) and
 * Julia's libm are third-party code outside R/: results that depend on their
 * last-bit behaviour are "parity unpinned" (see DESIGN.md).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction: Julia
 * never contracts a*b+c on its own).
 */
#ifndef RRTX_ORACLE_H
#define RRTX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: metric (R/DRRT_distance_functions.jl:37) ---------------------- */
double orc_euclid(const double *x, const double *y, int d);
/* hyper-ball radius, R/rrtqx.jl:382 */
double orc_ball_radius(double delta, double ball_constant, int64_t n, int d);

/* ---- A2-A5: kd-tree (R/kdTree_general.jl, R/ghostPoint.jl) ------------- */
typedef struct orc_kd orc_kd;
typedef struct orc_list orc_list;

orc_kd *orc_kd_create(int d);
void orc_kd_destroy(orc_kd *t);
/* wraps are 0-based dimension indices; wrap_points[i] is the period */
void orc_kd_set_wraps(orc_kd *t, int nwraps, const int *wraps, const double *wrap_points);
int64_t orc_kd_insert(orc_kd *t, const double *pos);  /* returns node index (insertion order) */
void orc_kd_insert_many(orc_kd *t, const double *pos, int64_t n);  /* n rows of d doubles, in order */
int64_t orc_kd_size(const orc_kd *t);
int64_t orc_kd_depth(const orc_kd *t);
const double *orc_kd_position(const orc_kd *t, int64_t idx);
void orc_kd_nearest(orc_kd *t, const double *q, int64_t *idx, double *dist);
void orc_kd_nearest_naive(orc_kd *t, const double *q, int64_t *idx, double *dist);

/* kdFindWithinRange returns a JList; list order is front -> back, i.e. the
 * reverse of discovery order (JlistPush inserts at the front). */
orc_list *orc_kd_find_within_range(orc_kd *t, double r, const double *q);
void orc_kd_find_more_within_range(orc_kd *t, double r, const double *q, orc_list *l);
int64_t orc_list_length(const orc_list *l);
/* copy list front->back into idx/key (cap entries at most); returns length */
int64_t orc_list_read(const orc_list *l, int64_t cap, int32_t *idx, double *key);
/* emptyRangeList: clears the inHeap flags and frees the list */
void orc_kd_empty_range_list(orc_kd *t, orc_list *l);
/* naive scan with the same inclusivity rule (root <=, others <); ascending idx.
 * With wraps, ghosts are applied the same way as the tree search. */
int64_t orc_range_naive(orc_kd *t, double r, const double *q, int64_t cap,
                        int32_t *idx, double *key);

/* kdFindKNearest (R/kdTree_general.jl:696-723): heap order, returns the length
 * (k nodes, but TWO for k = 1: the heap starts with root + dummy), -1 where the
 * reference raises (wrapped space).  _naive = kdFindKNearestNaive (:563-574). */
int64_t orc_kd_knearest(orc_kd *t, int64_t k, const double *q, int64_t cap, int32_t *idx, double *key);
int64_t orc_kd_knearest_naive(orc_kd *t, int64_t k, const double *q, int64_t cap, int32_t *idx, double *key);

/* ghost iterator exposed for tests (R/ghostPoint.jl:60-111).  Writes up to cap
 * ghosts (each d doubles) and returns how many were produced for bestDist. */
int orc_ghost_points(const orc_kd *t, const double *q, double best_dist, int cap, double *out);

/* ---- A9/A12: sphere obstacles (R/DRRT_Q.jl:1205-1210,1402-1595,1775-1826) */
typedef struct {
  double c[3];
  double radius;
  double life_span;    /* lifeSpan <= 0 => inactive */
  int32_t unused;      /* obstacleUnused */
  int32_t pad;
} orc_sphere;

double orc_distance_point_to_segment3(const double *c, const double *p0, const double *p1);
int orc_edge_check_sphere(const orc_sphere *ob, const double *p0, const double *p1, double robot_radius);
/* explicitEdgeCheck(C, edge): list order, early-out; returns 1 on hit and the
 * list position of the first hit (or -1) */
int orc_edge_check_spheres(const orc_sphere *obs, int m, const double *p0, const double *p1,
                           double robot_radius, int32_t *first_hit);
/* explicitPointCheck (quick=1: with quickCheck first pass, R/DRRT_Q.jl:1520;
 * quick=0: explicitPointCheck3D, :1558). wdims = 3 (SimpleEdge Wdist). */
int orc_point_check_spheres(const orc_sphere *obs, int m, const double *p, double robot_radius,
                            int quick, double *clearance);

/* ---- A10: polygon obstacles (R/DRRT.jl:1009-1106,1144-1202,1258-1470,1523-1653).
 * For the moving kinds 6/7 the points handed to the checks carry time in their third coordinate. */
typedef struct {
  int32_t kind;        /* 1 ball, 3 polygon, 6 / 7 polygon moving along `path` (R/DRRT_data_structures.jl:136-143) */
  int32_t nverts;
  const double *verts; /* nverts x 2 row-major (kinds 6/7: originalPolygon) */
  double cx, cy;       /* Obstacle(kind, polygon) ctor centre */
  double radius;
  double life_span;
  int32_t unused;
  int32_t npath;       /* kinds 6/7: rows of path */
  const double *path;  /* npath x 3 row-major (dx, dy, t): offsets from the ctor position vs time */
} orc_polygon;

/* Obstacle(kind=3, polygon) ctor: bbox centre + max vertex distance
 * (R/DRRT_data_structures.jl:229-241) */
void orc_polygon_ctor(const double *verts, int nverts, double *cx, double *cy, double *radius);
double orc_dist_sqrd_point_to_segment(const double *pt, const double *a, const double *b);
double orc_segment_dist_sqrd(const double *pa, const double *pb, const double *qa, const double *qb);
int orc_point_in_polygon(const double *pt, const double *verts, int nverts);
double orc_dist_to_polygon_sqrd(const double *pt, const double *verts, int nverts);
int orc_edge_check_polygon(const orc_polygon *ob, const double *p0, const double *p1, double robot_radius);
int orc_edge_check_polygons(const orc_polygon *obs, int m, const double *p0, const double *p1,
                            double robot_radius, int32_t *first_hit);
int orc_point_check_polygons(const orc_polygon *obs, int m, const double *p, double robot_radius,
                             double *clearance);

/* findPointsInConflictWithObstacle(S, KD, ob::Obstacle, root) for the polygon list (R/DRRT.jl:3048-3125): the
 * range list of the nodes whose edges addNewObstacle / removeObstacle (:3127-3290) re-check; NULL where the
 * reference raises.  The caller empties the list (orc_kd_empty_range_list). */
orc_list *orc_find_points_in_conflict_polygon(orc_kd *t, const orc_polygon *ob, double robot_radius, double delta,
                                              int has_time, int has_theta);

/* ---- A6/A7/A11: steering ------------------------------------------------ */
/* Dubins calculateTrajectory (R/DRRT_DubinsEdge_functions.jl:329-709), space
 * without time.  s, g are [x y t theta].  traj (may be NULL) receives up to
 * traj_cap rows of (x, y); *traj_len receives the number of rows the reference
 * would produce.  word is 3 chars + NUL. */
void orc_dubins_steer(const double *s, const double *g, double r_min, double *cost,
                      char *word, double *traj, int traj_cap, int *traj_len);
/* explicitEdgeCheck(S, DubinsEdge, obstacle) (:750-774) over a list */
/* (returns -1 when the list holds a moving obstacle: those need the time-parameterised
 * trajectory of :660-697, which is not restated) */
int orc_dubins_edge_check_polygons(const orc_polygon *obs, int m, const double *s, const double *g,
                                   const double *traj, int traj_len, double robot_radius,
                                   double r_min, int32_t *first_hit);
/* The same with S.spaceHasTime (R/DRRT_DubinsEdge_functions.jl:660-697, 115-121, 750-774): edge.dist,
 * edge.Wdist, edge.velocity, the trajectory with its time column (rows of x, y, t), validMove, and the
 * two-stage edge check whose pieces carry time (kinds 6 / 7 are tested at their time stamps). */
void orc_dubins_steer_time(const double *s, const double *g, double r_min, double *dist, double *wdist,
                           double *velocity, char *word, double *traj3, int traj_cap, int *traj_len);
/* the same with the time column formed piece by piece as the HIP kernels do (see rrtx_oracle.c) */
void orc_dubins_steer_time_pw(const double *s, const double *g, double r_min, double *dist, double *wdist,
                              double *velocity, char *word, double *traj3, int traj_cap, int *traj_len);
int orc_dubins_valid_move_time(const double *s, const double *g, double velocity, double v_min, double v_max);
int orc_dubins_edge_check_polygons_time(const orc_polygon *obs, int m, const double *s, const double *g,
                                        const double *traj3, int traj_len, double robot_radius, double r_min,
                                        int32_t *first_hit);
/* the shared deterministic transcendentals of include/rrtx_detmath.h, element-wise (tests compare them with
 * libm here and with the device's build of the same header bit for bit).
 * op: 0 sin(x)  1 cos(x)  2 atan2(y, x)  3 acos(x);  returns 1 in the ORC_LIBM_TRIG build, else 0 */
int orc_dm_eval(int op, const double *x, const double *y, int64_t n, double *out);
/* Julia float range length for start:step:stop in the literal fallback branch */
int64_t orc_julia_range_len(double start, double step, double stop);

/* ---- A13/A8: the per-sample extend inner loop, used as the CPU baseline --
 * For every query: kdFindNearest, kdFindWithinRange, then for every neighbour
 * both directed edges: SimpleEdge cost + explicitEdgeCheck over the obstacle
 * list with first-hit early-out.  Outputs are optional (NULL to skip).
 * Returns the number of directed edges checked. */
int64_t orc_extend_batch_spheres(orc_kd *t, const orc_sphere *obs, int m, const double *queries,
                                 int64_t nq, double r, double robot_radius,
                                 int64_t *nearest_idx, int64_t *n_neighbors_total,
                                 int64_t *n_hits_total);

int64_t orc_extend_batch_polygons(orc_kd *t, const orc_polygon *obs, int m, const double *queries, int64_t nq, double r,
                                  double robot_radius, int64_t *nearest_idx, int64_t *n_neighbors_total,
                                  int64_t *n_hits_total);

/* ---- N4: cost propagation (rrtx_oracle_graph.c): rewire / reduceInconsistency / propogateDescendants with the
 * reference's BinaryHeap and list orders, R/DRRT_Q.jl:2052-2077, 2364-2541, 2647-2817, 3244-3268, R/heap.jl:138-273 */
typedef struct orc_graph orc_graph;
orc_graph *orc_graph_create(int64_t n_nodes);
void orc_graph_destroy(orc_graph *g);
int64_t orc_graph_add_edge(orc_graph *g, int64_t start, int64_t end, double dist, int initial, int valid_move);
void orc_graph_set_node(orc_graph *g, int64_t v, double lmc, double tree_cost);
void orc_graph_set_move_goal(orc_graph *g, int64_t v, int flag);
void orc_graph_set_edge_dist(orc_graph *g, int64_t e, double dist);
double orc_graph_lmc(const orc_graph *g, int64_t v);
double orc_graph_tree_cost(const orc_graph *g, int64_t v);
int64_t orc_graph_parent_edge(const orc_graph *g, int64_t v);
int64_t orc_graph_queue_length(const orc_graph *g);
int64_t orc_graph_n_edges(const orc_graph *g);
void orc_graph_verify_in_queue(orc_graph *g, int64_t v);
void orc_graph_verify_in_os(orc_graph *g, int64_t v);
void orc_graph_make_parent_of(orc_graph *g, int64_t new_parent, int64_t node, int64_t edge);
void orc_graph_reduce_inconsistency(orc_graph *g, int64_t goal, int64_t root, double ball, double change_thresh);
void orc_graph_block_edge(orc_graph *g, int64_t e);
void orc_graph_propagate_descendants(orc_graph *g);

#ifdef __cplusplus
}
#endif
#endif
