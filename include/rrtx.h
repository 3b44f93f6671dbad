/*
 * rrtx.h -- C ABI of librrtx_hip.so: the MI355X (gfx950) implementation of the
 * RRT^X extend/rewire hot path of jnetter6/RRTQX_3D.
 *
 * The reference (Julia, R/ = code_RRTQx_3D/) has no FFI; the seam is a set of
 * free functions selected by multiple dispatch (R/README.txt:85-99).  Each entry
 * point below names the reference function(s) it replaces.  The Julia binding a
 * maintainer would add is shown in INTEGRATION.md and julia/RRTXHip.jl.
 *
 * Conventions
 *   - plain pointers and sizes only; all host buffers are caller-owned and only
 *     read/written for the duration of the call (GC.@preserve on the Julia side);
 *   - every function returns RRTX_OK (0) or a negative RRTX_E_* code and never
 *     throws or aborts; rrtx_last_error(ctx) gives the message;
 *   - points are passed as n x dim row-major doubles (a Julia dim x n Array);
 *   - node indices are 0-based insertion order (index 0 is the kd-tree root);
 *   - a ctx is bound to ONE GPU and is not thread-safe; distinct ctxs are
 *     independent (one per planner/agent tree, R/rrtqx.jl:29-31);
 *   - all arithmetic is IEEE fp64 without FMA contraction, so neighbour sets and
 *     collision booleans are bit-identical to the reference's CPU path; NaN
 *     inputs are not errors (a zero-length edge is a "hit", R/DRRT_Q.jl:1208);
 *   - host-pointer calls are synchronous; the *_dev calls take DEVICE pointers,
 *     enqueue on the ctx stream and return immediately (rrtx_sync to wait).
 */
#ifndef RRTX_H
#define RRTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRTX_OK 0
#define RRTX_E_INVALID (-1)  /* bad argument */
#define RRTX_E_CAPACITY (-2) /* caller output buffer too small; see `needed` */
#define RRTX_E_DEVICE (-3)   /* HIP runtime error */
#define RRTX_E_NOMEM (-4)
#define RRTX_E_STATE (-5)    /* call not valid in this state (e.g. empty tree) */

typedef struct rrtx_ctx rrtx_ctx;

typedef struct {
  int64_t n_nodes;
  int32_t dim;
  int32_t n_spheres;         /* as set (active + inactive) */
  int32_t n_polygons;
  int32_t n_wraps;
  /* per-kernel-family device time, accumulated while profiling is enabled
   * (HIP events on the ctx stream around every launch of that family) */
  double ms_nn_scan;     int64_t launches_nn_scan;
  double ms_nn_finish;   int64_t launches_nn_finish;
  double ms_nn_nearest;  int64_t launches_nn_nearest;
  double ms_edges;       int64_t launches_edges;
  double ms_points;      int64_t launches_points;
  double ms_dubins;      int64_t launches_dubins;
  /* work counters of the last rrtx_nn_radius* / rrtx_extend_candidates* call */
  int64_t last_pairs;        /* (query copy, node) visits */
  int64_t last_neighbors;    /* sum of k */
  int32_t last_tile_q;       /* query copies that shared one streamed pass of the node arrays */
  int32_t last_scan_units;   /* slab-culled range scan: (tile of last_tile_q copies, 512-node chunk) pairs the
                              * last call streamed; 0 when it streamed every node for every tile */
  /* (round 3, appended) the steering launches of the Dubins edge paths on their own: ms_dubins above then counts
   * the check kernels (and the stand-alone steer / trajectory calls) */
  double ms_dubins_steer; int64_t launches_dubins_steer;
  int64_t last_sweep_candidates;  /* mirrored edges the last rrtx_obstacle_sweep_polygon put through explicitEdgeCheck */
} rrtx_stats_t;

/* ---- lifetime ------------------------------------------------------------ */
/* replaces KDTree{T}(d, KDdist) + CSpace obstacle list construction
 * (R/kdTree_general.jl:94-112, R/rrtqx.jl:29-31).  dim is 3 (SimpleEdge, 3-D)
 * or 4 ([x y t theta], Dubins).  device = HIP device ordinal. */
int rrtx_create(rrtx_ctx **out, int dim, int device, int64_t node_capacity);
int rrtx_destroy(rrtx_ctx *ctx);
const char *rrtx_last_error(rrtx_ctx *ctx);
/* message of the last failing rrtx_create (no ctx exists then) */
const char *rrtx_create_error(void);
/* run on a caller-provided hipStream_t (e.g. the stream the caller's framework is using); NULL
 * restores the ctx-owned stream */
int rrtx_set_stream(rrtx_ctx *ctx, void *hip_stream);
void *rrtx_get_stream(rrtx_ctx *ctx);
int rrtx_sync(rrtx_ctx *ctx);
/* per-kernel-family HIP-event timing (resets the sums): 0 = off, 1 = only the range-scan kernel
 * (two event records per call), 2 = every family (each record costs ~10 us between kernels) */
int rrtx_profile(rrtx_ctx *ctx, int enable);
int rrtx_stats(rrtx_ctx *ctx, rrtx_stats_t *out);
/* tuning switches; none of them changes a result.
 *   RRTX_OPT_NN_FILTER (default 1): range search screens (query, node) pairs with a
 *   conservative fp32 bound before the exact unfused fp64 test; 0 = exact test on
 *   every pair. */
#define RRTX_OPT_NN_FILTER 1
/*   RRTX_OPT_SCAN_BLOCKS: target workgroup count of the range scan (launch geometry);
 *   RRTX_OPT_SCAN_TILE_Q: query copies per workgroup tile, 0 = default. */
#define RRTX_OPT_SCAN_BLOCKS 2
#define RRTX_OPT_SCAN_TILE_Q 3
#define RRTX_OPT_SCAN_ITEMS 4  /* target number of (tile, node segment) work items */
/*   RRTX_OPT_NN_CULL (default 1): the range search keeps a second copy of the node shadow
 *   ordered by (x, y) grid cell and buckets each call's query copies by cell, so a tile of
 *   copies only streams the 512-node chunks whose x/y extent can reach it.  Purely a skip of
 *   pairs that provably fail the exact test; results are identical.  0 = off (every tile
 *   streams every node), 1 = on for trees of at least 8192 nodes, 2 = always on. */
#define RRTX_OPT_NN_CULL 5
/*   RRTX_OPT_PROFILE_EVERY (default 1): with rrtx_profile level 1, only every n-th launch of the
 *   range-search kernel is bracketed by HIP events (an event record between two kernels drains the
 *   pipeline for several microseconds); rrtx_stats then reports the timed launches only. */
#define RRTX_OPT_PROFILE_EVERY 6
/*   RRTX_OPT_KNN_LISTS (default 1): rrtx_nn_knearest takes the k nearest of a query from its
 *   range-search list (radius guessed from a sample of the batch) and runs the exhaustive
 *   selection kernel only for queries whose list holds fewer than k nodes; 0 = exhaustive for all. */
#define RRTX_OPT_KNN_LISTS 7
/*   RRTX_OPT_EXTEND_OBSTACLES (default 0): which obstacle list rrtx_extend_candidates* checks the
 *   candidate edges and the samples against -- 0 = the sphere list (explicitEdgeCheck3D, the 3-D
 *   planner of R/rrtqx.jl), 1 = the polygon list (explicitEdgeCheck2D on the (x, y) projection and the
 *   polygon explicitPointCheck, R/DRRT.jl:1434-1470, 1523-1678; time in the third coordinate for the
 *   moving kinds).  This one DOES select behaviour: it says which CSpace.obstacles the caller has. */
#define RRTX_OPT_EXTEND_OBSTACLES 8
/*   RRTX_OPT_NEAREST_REC_CAP (testing, default 0 = sized from the batch): candidate records the screened
 *   nearest scan may keep; queries that lose one are answered again exactly on the device. */
#define RRTX_OPT_NEAREST_REC_CAP 9
/*   RRTX_OPT_BUCKET_MULT (default 2, grows by itself after a call whose lists overflowed): capacity of the
 *   per-query hit buckets of the range search in units of cap / nq; 2, 4, 8 or 16. */
#define RRTX_OPT_BUCKET_MULT 10
/*   RRTX_OPT_TUNE (default 0): bit mask of kernel variants under measurement; results are identical. */
#define RRTX_OPT_TUNE 11
/*   RRTX_OPT_SPACE_HAS_TIME (default 0; dim = 4 only): CSpace.spaceHasTime (R/DRRT_data_structures.jl:330) for the
 *   Dubins entry points.  The third coordinate of [x y t theta] is then time (planning runs in reverse time:
 *   an edge's start node is LATER than its end node): edge.dist = sqrt(Wdist^2 + dt^2), edge.velocity =
 *   Wdist / dt and edge.trajectory carries a time column (R/DRRT_DubinsEdge_functions.jl:660-697), validMove
 *   wants start time > end time and a velocity within rrtx_set_dubins_velocity's bounds (:115-121), and the two-stage
 *   edge check hands its chord and its pieces to explicitEdgeCheck2D with their times, so polygons that move
 *   in time (kinds 6 / 7) are tested where they are when the robot passes (:750-774, R/DRRT.jl:1579-1651).
 *   This one DOES select behaviour: it says which space the caller plans in. */
#define RRTX_OPT_SPACE_HAS_TIME 12
/*   RRTX_OPT_ROOT_RULE (default 1): node 0 of this context is the kd-tree's root, which the range search takes
 *   with <= (R/kdTree_general.jl:896).  0 for a context that holds a LATER index range of a tree sharded over
 *   several GPUs (rrtqx_3d_amd/parallel.py): its node 0 is an ordinary node. */
#define RRTX_OPT_ROOT_RULE 13
int rrtx_set_option(rrtx_ctx *ctx, int option, int64_t value);
/* The value an option currently has (as rrtx_set_option normalised it): callers that size buffers by an
 * option -- the row width of rrtx_dubins_trajectory -- read it here instead of keeping a shadow copy. */
int rrtx_get_option(rrtx_ctx *ctx, int option, int64_t *value);
/* Host-pointer entry points move their results through a pinned staging arena owned by the context (DMA at PCIe rate,
 * then one memcpy per output array).  A caller that keeps its output arrays alive across calls -- the Julia host
 * preallocates them -- can have them page-locked instead: after rrtx_host_register(ptr, bytes) every output pointer
 * inside [ptr, ptr + bytes) receives its DMA directly (hipHostRegister; unregister before freeing the array).
 * Results are identical either way. */
int rrtx_host_register(rrtx_ctx *ctx, void *ptr, size_t bytes);
int rrtx_host_unregister(rrtx_ctx *ctx, void *ptr);

/* Host-only helper (no GPU needed): the exact thresholds on SQUARED distances the kernels
 * compare against, so that no device sqrt sits on a decision path:
 *   *first_ge = min{ s >= 0 : sqrt(s) >= r }   (sqrt(s) <  r  <=>  s < *first_ge)
 *   *first_gt = min{ s >= 0 : sqrt(s) >  r }   (sqrt(s) <= r  <=>  s < *first_gt)
 * (NaN when no such s exists).  Exposed for testing the host logic. */
int rrtx_sq_thresholds(double r, double *first_ge, double *first_gt);

/* ---- tree (A2, A5) --------------------------------------------------------- */
/* kdInsert (R/kdTree_general.jl:121-170): appends n nodes; *first_index receives
 * the index of the first one (== treeSize before the call). */
int rrtx_nodes_append(rrtx_ctx *ctx, const double *pos, int64_t n, int64_t *first_index);
int64_t rrtx_nodes_count(rrtx_ctx *ctx);
/* device-to-device variant: pos is a device pointer */
int rrtx_nodes_append_dev(rrtx_ctx *ctx, const double *pos_dev, int64_t n);
/* KDTree(d, f, wraps, wrapPoints) (R/kdTree_general.jl:108): dimension
 * dim_index (0-based) wraps with the given period (Dubins theta: 3, 2*pi;
 * R/DRRT.jl:3312).  At most 3 wrapped dimensions. */
int rrtx_set_wrap(rrtx_ctx *ctx, int dim_index, double period);

/* ---- obstacles (A15) ------------------------------------------------------- */
/* CSpace.obstacles as List{SphereObstacle} in LIST ORDER (front first,
 * R/list.jl:53-58).  cxyzr is m x 4; active[i] = !(obstacleUnused || lifeSpan<=0)
 * (R/DRRT_Q.jl:1777); NULL = all active. */
int rrtx_spheres_set(rrtx_ctx *ctx, const double *cxyzr, const uint8_t *active, int m);
/* polygon Obstacles (legacy 2-D path, R/DRRT_data_structures.jl:135-265), list
 * order.  vert_off is m+1 CSR offsets into vxy (2 doubles per vertex);
 * centre_radius is m x 3 (the Obstacle(3, polygon) ctor values, :229-241, or
 * NULL to have the library apply that ctor); kind[i] is 1 (ball), 3 (polygon), or 6 / 7 (polygon
 * moving in time along a path, :140-143; give the paths with rrtx_polygon_paths_set).  Kinds 2, 4
 * and 5 raise or cannot be constructed in the reference (R/DRRT.jl:1546, data_structures:256) and
 * are refused. */
int rrtx_polygons_set(rrtx_ctx *ctx, const int32_t *vert_off, const double *vxy,
                      const double *centre_radius, const uint8_t *kind, const uint8_t *active, int m);
/* Obstacle.path of the moving kinds 6 and 7 (R/DRRT_data_structures.jl:184-187; read by
 * readTimeObstaclesFromfile, R/DRRT_Q.jl:1022-1061): path_off is m+1 CSR row offsets into path_xyt,
 * 3 doubles per row (dx, dy, t) = offset of the obstacle from its ctor position at time t, t
 * ascending; m is the count last given to rrtx_polygons_set (which clears all paths); static kinds
 * have empty ranges.  For kind 7 this is the path the robot currently assumes -- call again after
 * the host recomputed it (changeObstacleDirection, R/DRRT.jl:370-443).  With such obstacles in the
 * list, rrtx_edges_check* / rrtx_points_check* read TIME from the third coordinate of their points
 * (startPoint[3], R/DRRT_Q.jl:1703; point[3], :1369): edges are tested at the closest approach of
 * the two centres against the bounding circle (:1699-1771), points against the polygon moved to its
 * place at that time (R/DRRT.jl:1289-1305, 1395-1420).  A moving obstacle without a path, and Dubins
 * edge checks against moving obstacles in a space WITHOUT time (their pieces carry no time stamp; see
 * RRTX_OPT_SPACE_HAS_TIME), fail with RRTX_E_STATE. */
int rrtx_polygon_paths_set(rrtx_ctx *ctx, const int32_t *path_off, const double *path_xyt, int m);
/* obstacleAugmentation / expiry (R/obstacleAugmentation.jl:106-114,
 * R/DRRT_Q.jl:3301): change radius and/or active flag of sphere `which`. */
int rrtx_obstacle_update(rrtx_ctx *ctx, int which, double radius, uint8_t active);

/* ---- nearest neighbours (A3, A4) ------------------------------------------ */
/* kdFindNearest (R/kdTree_general.jl:357-385), batched.  idx/dist: nq entries.
 * Ties on distance resolve to the lowest index (the reference's tie order is
 * its tree-visit order). */
int rrtx_nn_nearest(rrtx_ctx *ctx, const double *q, int nq, int32_t *idx, double *dist);
/* kdFindKNearest (R/kdTree_general.jl:696-723; helpers :580-593, :605-692), batched.  The
 * reference seeds its heap with the root and an Inf-keyed dummy, so it hands back max(k, 2)
 * nodes (fewer when the tree is smaller): rows of idx/dist are max(k, 2) wide, count[i] says
 * how many entries of row i are filled (unused slots: idx -1, dist +Inf).  Rows come sorted by
 * ascending (distance, index) -- the reference returns heap order -- and ties at the last place
 * go to the lowest indices.  Nodes at a non-finite distance are never returned.  1 <= k <= 2048.
 * Like the reference (:711-713) the call fails on a wrapped space (RRTX_E_STATE).  The reference
 * itself never calls this search (RRT^X uses the radius search).  See RRTX_OPT_KNN_LISTS. */
int rrtx_nn_knearest(rrtx_ctx *ctx, const double *q, int nq, int k, int32_t *idx, double *dist, int32_t *count);
/* kdFindWithinRange (R/kdTree_general.jl:889-919), batched: for query i the
 * nodes with KDdist < r[i] (the root, index 0, with <=), wrapped dimensions
 * handled with the reference's ghost rule; each list sorted by node index,
 * dist = the key the reference stores.  CSR output; if the total exceeds cap
 * the call returns RRTX_E_CAPACITY with *needed set (offsets are still valid).
 * r_stride: 0 = one radius r[0] for all queries, 1 = r[i] per query. */
int rrtx_nn_radius(rrtx_ctx *ctx, const double *q, const double *r, int r_stride, int nq,
                   int64_t *offsets /* nq+1 */, int32_t *idx, double *dist, int64_t cap,
                   int64_t *needed);

/* ---- collision (A8-A12) ----------------------------------------------------- */
/* explicitEdgeCheck(C, edge) (R/DRRT_Q.jl:1802-1826) for ne straight edges
 * p0[i] -> p1[i]; obstacle_or_minus1 >= 0 restricts the test to that one
 * obstacle (explicitEdgeCheck(S, edge, ob), R/DRRT_Q.jl:3248).  kind selects
 * the obstacle list: 0 = spheres (explicitEdgeCheck3D, :1775-1795, uses the
 * first 3 coordinates), 1 = polygons (explicitEdgeCheck2D, R/DRRT.jl:1523-1578,
 * uses the first 2).  hit[i] in {0,1}; first_hit[i] = list position of the
 * first colliding obstacle or -1 (may be NULL). */
int rrtx_edges_check(rrtx_ctx *ctx, int kind, const double *p0, const double *p1, int64_t ne,
                     double robot_radius, int obstacle_or_minus1, uint8_t *hit, int32_t *first_hit);
/* The same test for edges given as NODE INDEX pairs (the planner's graph edges), as the obstacle
 * sweeps need it: addNewObstacle tests every out-edge of the nodes near a new obstacle against
 * that one obstacle (R/DRRT_Q.jl:3220-3290: obstacle_or_minus1 = its list position); removeObstacle
 * re-tests freed edges against the OTHER obstacles that are active in their time window
 * (R/DRRT_Q.jl:3321-3337: obstacle_mask[i] != 0 selects them; NULL = all).  Sphere list only. */
int rrtx_edges_check_idx(rrtx_ctx *ctx, const int32_t *start_idx, const int32_t *end_idx, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, const uint8_t *obstacle_mask,
                         uint8_t *hit, int32_t *first_hit);
/* Obstacle sweep against a device mirror of the planner's directed edges.
 * rrtx_graph_edges_append registers edges start -> end (node indices; what RRTNodeNeighborIterator
 * walks: the out-neighbour edges and the parent edge of every node) and returns the id of the first
 * one; ids are consecutive.  The mirror may be a superset of the live graph (edges the planner has
 * dropped are simply ignored by the caller).  rrtx_obstacle_sweep is the edge loop of addNewObstacle
 * (R/DRRT_Q.jl:3195-3290): nodes within search_range (= robotRadius + delta + ob.radius) of sphere
 * `obstacle`'s centre -- kdFindWithinRange, root with <= -- and, among the registered edges that
 * START at such a node, those for which explicitEdgeCheck(S, edge, ob) is true.  edge_ids receives
 * their ids in ascending order (two-call capacity pattern).  An inactive obstacle collides with
 * nothing (R/DRRT_Q.jl:1777). */
int rrtx_graph_edges_append(rrtx_ctx *ctx, const int32_t *start_idx, const int32_t *end_idx, int64_t n,
                            int64_t *first_id);
int64_t rrtx_graph_edges_count(rrtx_ctx *ctx);
int rrtx_graph_edges_clear(rrtx_ctx *ctx);
int rrtx_obstacle_sweep(rrtx_ctx *ctx, int obstacle, double search_range, double robot_radius, int32_t *edge_ids,
                        int64_t cap, int64_t *needed);
/* The obstacle sweeps of the POLYGON list -- the 2-D Euclidean and the Dubins space, with or without time
 * (legacy planner, R/DRRT.jl:3048-3290; BASELINE config 5's "dynamic discoverable obstacles" run these).  The edge
 * type is the context's: dim = 3 SimpleEdge, dim = 4 DubinsEdge (r_min = S.minTurningRadius; with
 * RRTX_OPT_SPACE_HAS_TIME the pieces carry time).  `obstacle` is a list position of rrtx_polygons_set.
 *   nodes: findPointsInConflictWithObstacle (:3048-3125) -- static kinds: range robotRadius + delta + ob.radius
 *     around ob.position (Dubins space: around [x y 0.0 pi] with range + pi; a dim = 3 tree is the 2-D space at
 *     z = 0); kinds 6 / 7: one query per path segment at [ob.position 0.0] + (path[i] + path[i+1]) / 2 with range +
 *     half the segment's length, accumulated (kdFindMoreWithinRange); the root with <=, ghosts of wrapped dimensions
 *     as the range search takes them.  A static obstacle in a space with time is the reference's
 *     error("this type of obstacle not coded for this type of space") -> RRTX_E_STATE;
 *   mode 0, addNewObstacle's loop (:3127-3200): the mirrored edges that START at such a node (its out-neighbour
 *     edges and parent edge) for which explicitEdgeCheck(S, edge, ob) is true -- the caller sets their dist = Inf
 *     (rrtx_graph_edges_block);
 *   mode 1, removeObstacle's loop (:3202-3290): of those edges the ones that are blocked in the mirror
 *     (dist == Inf), collide with ob, and with no OTHER obstacle that is in use -- the caller resets them to
 *     distOriginal (the reference also asks startTime <= timeElapsed <= startTime + lifeSpan of the others: fold
 *     it into their `active` flags).  ob itself must still be in use, as it is in the reference until the loop
 *     is over (:3287); an obstacle not in use collides with nothing and the sweep returns no edge.
 * edge_ids: ascending, two-call capacity pattern (RRTX_E_CAPACITY with *needed set). */
int rrtx_obstacle_sweep_polygon(rrtx_ctx *ctx, int obstacle, double robot_radius, double delta, double r_min, int mode,
                                int32_t *edge_ids, int64_t cap, int64_t *needed);
/* Cost propagation over the edge mirror: the fixed point that rewire / reduceInconsistency /
 * propogateDescendants (R/DRRT_Q.jl:2490-2541, 2647-2817) drive rrtLMC to when changeThresh = 0 and the queue
 * runs dry -- lmc(root) = 0, lmc(v) = min over mirrored edges v -> u with finite dist of lmc(u) + dist (one
 * rounded addition per edge, the value the reference's heap order also ends at), Inf for a node that cannot
 * reach the root (an orphan).  Every mirrored edge carries edge.dist: the SimpleEdge cost of its two nodes when
 * appended, overwritten with rrtx_graph_edges_set_dist (Dubins costs, costs in a space with time), set to Inf
 * with rrtx_graph_edges_block for the ids rrtx_obstacle_sweep returned (addNewObstacle: `dist = Inf`,
 * R/DRRT_Q.jl:3249).  parent_edge[v] (may be NULL) = the id of the mirrored edge v -> rrtParent(v): the lowest id
 * among the edges that attain the minimum (-1: the root, or an orphan).  passes (may be NULL) = relaxation
 * passes run.  CONTRACT: changeThresh = 0 run to exhaustion only -- a positive changeThresh (the value the one
 * runnable script passes, R/experimentsForRRTQX.jl:30,132) and the goal-bounded loop of reduceInconsistency
 * (R/DRRT_Q.jl:2706) make the reference's result depend on its pop order; that epsilon-consistent variant is not
 * offered and stays on the host.  A solve that cannot reach a fixed point (a pass limit, a parent structure that is
 * no forest, a HIP error part-way) returns RRTX_E_STATE / RRTX_E_DEVICE and FORGETS the previous solve: the next
 * rrtx_graph_cost_update then solves in full.
 * rrtx_graph_cost_update gives the same answer starting from the state the previous call (either function, same
 * root) left on the device: nodes and edges appended since then, costs changed with set_dist / block.  This is
 * the replanning step: edges whose cost was touched and that were parent edges orphan their subtrees
 * (propogateDescendants, R/DRRT_Q.jl:2760-2817), orphans restart at Inf, and only the region that changes is
 * relaxed again.  Without a previous solve for this root it is rrtx_graph_cost_to_root. */
int rrtx_graph_edges_set_dist(rrtx_ctx *ctx, int64_t first_id, const double *dist, int64_t n);
int rrtx_graph_edges_block(rrtx_ctx *ctx, const int32_t *edge_ids, int64_t n);
int rrtx_graph_cost_to_root(rrtx_ctx *ctx, int root_idx, double *lmc /* n_nodes */, int32_t *parent_edge /* n_nodes */,
                            int32_t *passes);
int rrtx_graph_cost_update(rrtx_ctx *ctx, int root_idx, double *lmc /* n_nodes */, int32_t *parent_edge /* n_nodes */,
                           int32_t *passes);
/* explicitPointCheck (R/DRRT_Q.jl:1520-1556; quick=0: explicitPointCheck3D,
 * :1558-1590).  unsafe[i] in {0,1}; clearance[i] = the returned certificate
 * (0.0 when unsafe); clearance may be NULL when only the flag is wanted (the
 * same obstacles are looked at either way). kind as above. */
int rrtx_points_check(rrtx_ctx *ctx, int kind, const double *p, int64_t np, double robot_radius,
                      int quick, uint8_t *unsafe, double *clearance);

/* ---- steering (A6, A7, A11) -------------------------------------------------- */
/* calculateTrajectory(S, ::SimpleEdge) (R/DRRT_SimpleEdge_functions.jl:177-181):
 * dist over all dim coordinates, wdist over the first 3. */
int rrtx_simple_steer(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double *dist,
                      double *wdist);
/* calculateTrajectory(S, ::DubinsEdge) (R/DRRT_DubinsEdge_functions.jl:329-501),
 * space without time: cost = edge.dist = edge.Wdist, word = 3 chars per edge
 * ("rsl","rsr","rlr","lsr","lsl","lrl","xxx").  s, g are ne x 4 [x y t theta]. */
int rrtx_dubins_steer(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                      double *cost, uint8_t *word /* ne x 3 */);
/* S.dubinsMinVelocity / S.dubinsMaxVelocity (R/DRRT_data_structures.jl:354-355), read by validMove in a space
 * with time.  Default: no bounds. */
int rrtx_set_dubins_velocity(rrtx_ctx *ctx, double v_min, double v_max);
/* calculateTrajectory's scalar results in full: dist = edge.dist (== Wdist without time, sqrt(Wdist^2 + dt^2)
 * with), wdist = edge.Wdist, velocity = edge.velocity (0 without time), valid_move = validMove(S, edge)
 * (always 1 without time).  Any output may be NULL. */
int rrtx_dubins_steer_full(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min, double *dist,
                           double *wdist, double *velocity, uint8_t *word /* ne x 3 */, uint8_t *valid_move);
/* Same, plus the discretised trajectory (:506-701) and the two-stage Dubins
 * collision check against the polygon list (:750-774).  traj_len[i] = number of
 * polyline rows the reference builds (may be NULL). */
int rrtx_dubins_edges_check(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne,
                            double r_min, double robot_radius, double *cost, uint8_t *word,
                            uint8_t *hit, int32_t *traj_len);

/* explicitEdgeCheck(S, edge::DubinsEdge, ob) (:750-774) against ONE obstacle of the polygon list (list position;
 * an obstacle not in use collides with nothing, R/DRRT.jl:1525): what addNewObstacle asks per edge (R/DRRT.jl:3157). */
int rrtx_dubins_edges_check_obstacle(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                                     double robot_radius, int obstacle, uint8_t *hit);

/* edge.trajectory of calculateTrajectory(S, ::DubinsEdge) (:506-701): the discretised polyline
 * (0.1 rad arc steps, Julia float-range length rule), P_i rows of (x, y) per edge -- rows of (x, y, t) with
 * RRTX_OPT_SPACE_HAS_TIME, the last row being the end node's (x, y, t) (:684-696) -- CSR layout:
 * traj_off[ne+1] (rows), traj_xy[cols * total rows].  `cols` is the row width the caller's buffer was sized
 * for and must be the context's (2, or 3 with RRTX_OPT_SPACE_HAS_TIME; rrtx_get_option tells): anything else is
 * RRTX_E_INVALID and nothing is written -- a caller whose idea of the space drifted from the context's gets an
 * error, not a buffer overrun.  Two-call pattern: if the total exceeds cap_rows the call returns
 * RRTX_E_CAPACITY with *needed_rows set (traj_off is still valid). */
int rrtx_dubins_trajectory(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                           int64_t *traj_off, double *traj_xy, int cols, int64_t cap_rows, int64_t *needed_rows);

/* Diagnostics: the deterministic transcendentals of include/rrtx_detmath.h evaluated ON THE DEVICE, element-wise
 * over host arrays (op 0 sin(x), 1 cos(x), 2 atan2(y, x), 3 acos(x); y may be NULL for the one-argument ops).
 * The parity suite holds the device build of that header against the host build bit for bit with it. */
int rrtx_detmath_eval(rrtx_ctx *ctx, int op, const double *x, const double *y, int64_t n, double *out);

/* ---- fused per-sample preamble of extend() (A13) ----------------------------- */
/* For each of nq samples: kdFindWithinRange + for every neighbour both directed
 * SimpleEdges sample->near and near->sample: calculateTrajectory cost and
 * explicitEdgeCheck over the sphere list (R/DRRT_Q.jl:1927-1979, 2581-2637),
 * plus kdFindNearest (R/rrtqx.jl:926) and explicitPointCheck of the sample
 * (R/rrtqx.jl:940).  CSR layout as rrtx_nn_radius; per neighbour entry:
 * cost (same both ways for SimpleEdge), hit_out (sample->near), hit_in. */
int rrtx_extend_candidates(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                           int64_t *offsets, int32_t *idx, double *cost, uint8_t *hit_out,
                           uint8_t *hit_in, int64_t cap, int64_t *needed, int32_t *nearest_idx,
                           double *nearest_dist, uint8_t *sample_unsafe);

/* The same preamble for Edge = DubinsEdge (BASELINE config 3; R/dubinsExperimentsForPaper.jl): tree in
 * [x y t theta] with theta wrapped (rrtx_set_wrap), polygon obstacle list.  Per neighbour entry:
 * key = the KDdist the range search stores, Dubins cost and word for sample->near (out) and
 * near->sample (in), and the two-stage Dubins collision flags (R/DRRT_DubinsEdge_functions.jl:750-774).
 * With RRTX_OPT_SPACE_HAS_TIME the costs are edge.dist in [x y t theta] and a flag byte also carries
 * bit 1 (value 2) = !validMove(S, edge): findBestParent blocks an edge on either (R/DRRT_Q.jl:1960), so the
 * caller's test is simply hit != 0.
 * word_out / word_in (3 bytes per entry) and nearest_* / sample_unsafe may be NULL. */
int rrtx_extend_candidates_dubins(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                                  double r_min, int64_t *offsets, int32_t *idx, double *key, double *cost_out,
                                  double *cost_in, uint8_t *word_out, uint8_t *word_in, uint8_t *hit_out,
                                  uint8_t *hit_in, int64_t cap, int64_t *needed, int32_t *nearest_idx,
                                  double *nearest_dist, uint8_t *sample_unsafe);

/* ---- device-resident variants (inputs/outputs are DEVICE pointers) ------------ */
int rrtx_nn_nearest_dev(rrtx_ctx *ctx, const double *q, int nq, int32_t *idx, double *dist);
/* rows max(k, 2) wide as in rrtx_nn_knearest.  The list path (RRTX_OPT_KNN_LISTS) reads two small values
 * back from the device on the way (its radius guess and the size of the lists), so this call waits on the
 * stream internally; the result kernels themselves are only enqueued. */
int rrtx_nn_knearest_dev(rrtx_ctx *ctx, const double *q, int nq, int k, int32_t *idx, double *dist, int32_t *count);
int rrtx_nn_radius_dev(rrtx_ctx *ctx, const double *q, double r, int nq, int64_t *offsets,
                       int32_t *idx, double *dist, int64_t cap, int64_t *needed_dev);
int rrtx_edges_check_dev(rrtx_ctx *ctx, int kind, const double *p0, const double *p1, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end,
                         uint8_t *hit, int32_t *first_hit);
int rrtx_points_check_dev(rrtx_ctx *ctx, int kind, const double *p, int64_t np, double robot_radius,
                          int quick, uint8_t *unsafe, double *clearance);
int rrtx_extend_candidates_dev(rrtx_ctx *ctx, const double *q, int nq, double r,
                               double robot_radius, int64_t *offsets, int32_t *idx, double *cost,
                               uint8_t *hit_out, uint8_t *hit_in, int64_t cap, int64_t *needed_dev,
                               int32_t *nearest_idx, double *nearest_dist, uint8_t *sample_unsafe);

/* device-pointer form of rrtx_extend_candidates_dubins; *needed_dev receives the number of entries (entries
 * beyond cap are not written).  Only enqueues work, never waits on the stream: with wrapped dimensions nearest_*
 * come from the nearest scan (rrtx_nn_nearest_dev: records a full candidate buffer dropped are re-decided by its
 * fix-up pass on the device); without them they come off the lists, and a sample whose ball is empty gets
 * kdFindNearest's answer from the expanding search of the finish kernel -- no -1 leaves the device. */
int rrtx_extend_candidates_dubins_dev(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                                      double r_min, int64_t *offsets, int32_t *idx, double *key, double *cost_out,
                                      double *cost_in, uint8_t *word_out, uint8_t *word_in, uint8_t *hit_out,
                                      uint8_t *hit_in, int64_t cap, int64_t *needed_dev, int32_t *nearest_idx,
                                      double *nearest_dist, uint8_t *sample_unsafe);

/* Per-edge collision bitmask for the multi-GPU exchange (RCCL all-reduce over
 * xGMI): bit e (e < cap) = hit_out[e], bit cap+e = hit_in[e]; entries at or
 * beyond *n_valid_dev read as 0.  words has (2*cap+63)/64 uint64 entries. */
/* device-pointer forms of rrtx_graph_cost_to_root / _update (they wait on the stream between groups of passes to learn whether
 * the fixed point is reached) */
int rrtx_graph_cost_to_root_dev(rrtx_ctx *ctx, int root_idx, double *lmc_dev, int32_t *parent_edge_dev);
int rrtx_graph_cost_update_dev(rrtx_ctx *ctx, int root_idx, double *lmc_dev, int32_t *parent_edge_dev);

int rrtx_pack_hits_dev(rrtx_ctx *ctx, const uint8_t *hit_out, const uint8_t *hit_in,
                       const int64_t *n_valid_dev, int64_t cap, uint64_t *words);

#ifdef __cplusplus
}
#endif
#endif
