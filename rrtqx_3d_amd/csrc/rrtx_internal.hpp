// rrtx_internal.hpp -- context, device buffers and launch declarations shared by
// the HIP translation units of librrtx_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrtx.h"

namespace rrtx {

constexpr int kMaxWraps = 3;
constexpr int kMaxSlots = 1 << kMaxWraps;  // query copies per query (original + ghosts)

// ---- records shared between host and device --------------------------------
// One query copy as the scan kernel reads it with scalar loads.
struct alignas(32) QRec3 { double x, y, z, thr; };                 // thr: hit <=> d2 < thr
struct alignas(64) QRec4 { double x, y, z, w, thr, pad0, pad1, pad2; };
template <int D> struct QRecT;
template <> struct QRecT<3> { using type = QRec3; };
template <> struct QRecT<4> { using type = QRec4; };

// Fixed-slot table entry: slot j of query i lives at [i * n_slots + j].  Slot 0 is
// the query itself, slots 1.. are its ghosts in the reference's iterator order
// (R/ghostPoint.jl:60-111).  thr_lt < 0 marks a ghost the iterator skips.
struct alignas(64) SlotRec { double x, y, z, w, thr_lt, thr_gt, pad0, pad1; };

// One range-search hit: node `idx` is within range of query `owner`; d2 is the
// squared distance to the copy that discovered it first.
struct alignas(16) HitRec { int32_t owner; int32_t idx; double d2; };

// Active sphere as the edge kernel reads it: centre + threshold on the squared
// distance (hit <=> !(s >= thr), thr = first s with sqrt(s) > robotRadius+radius).
struct alignas(32) SphRec { double cx, cy, cz, thr; };

// host-side mirror of exact_math.hpp's ChunkExt (this header is also read by plain C++)
struct ChunkExtHost { unsigned long long xlo, xhi, ylo, yhi; };

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t need) {
    if (need <= bytes) return hipSuccess;
    size_t nb = bytes ? bytes : 4096;
    while (nb < need) nb *= 2;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, nb);
    if (e == hipSuccess) bytes = nb;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// cost propagation over the edge mirror (kernels_graph.hip): the state of the last solve stays on the device
struct GraphCost {
  DevBuf lmc, parent, stamp, flags, orph, anc, ids;
  DevBuf in_cnt, in_start, in_cursor, in_tiles;     // CSR of in-edges: per end node, ...
  DevBuf in_src, in_w, in_pos;                      // ... start node and cost of every in-edge; slot of every edge id
  int64_t in_ne = 0;                                // edges the CSR holds ([in_ne, ge_n) is the tail)
  int in_nn = 0;                                    // nodes the CSR holds
  int solved_root = -1;
  int64_t solved_nodes = 0, solved_edges = 0;
  bool touched_old = false;                         // set_dist / block since the last solve
};

enum KernelFamily { KF_NN_SCAN = 0, KF_NN_FINISH, KF_NN_NEAREST, KF_EDGES, KF_POINTS, KF_DUBINS, KF_DUBINS_STEER, KF_COUNT };

struct TimedSpan { hipEvent_t a, b; int family; };

}  // namespace rrtx

struct rrtx_ctx {
  int dim = 3;
  int device = 0;
  hipStream_t own_stream = nullptr;
  // host-pointer entry points: results leave the device through ONE pinned staging arena (hipHostMalloc; DMA at PCIe
  // rate, no driver-side staging of pageable memory) and are copied into the caller's arrays after the final sync --
  // unless the caller registered its arrays (rrtx_host_register), in which case the DMA goes straight to them
  char *h_arena = nullptr;
  size_t h_arena_bytes = 0, h_arena_used = 0;
  struct HostCopy { void *dst; const void *src; size_t bytes; };
  std::vector<HostCopy> h_pending;                       // arena -> caller copies to run after the next sync
  std::vector<std::pair<const char *, size_t>> h_registered;
  hipStream_t stream = nullptr;
  std::string err;

  // node SoA (fp64), one array per coordinate
  double *nodes[4] = {nullptr, nullptr, nullptr, nullptr};
  // the same coordinates, four doubles per node (x y z w): one 32-byte read per random node access
  double *nodes_aos = nullptr;
  // fp32 shadow of the node SoA for the conservative range prefilter
  // (kernels_nn.hip, nn_scan_f32_kernel); never used to decide a result
  float *nodes_f[4] = {nullptr, nullptr, nullptr, nullptr};
  float *nodes_pp = nullptr;        // |p~|^2 of the shifted fp32 coordinates
  double origin[4] = {0, 0, 0, 0};  // shift applied before the fp32 conversion (first node)
  bool origin_set = false;
  int64_t n_nodes = 0, cap_nodes = 0;
  rrtx::DevBuf d_absmax;            // uint64: bit pattern of max |coordinate| over all nodes

  // Slab-ordered copy of the fp32 shadow for the culled range scan (kernels_nn.hip, "slab
  // index").  Positions [0, sl_n_sorted) hold nodes 0..sl_n_sorted-1 ordered by equal-width
  // (x, y) grid cell and, inside a cell, by bin of the third coordinate (sl_kz bins); positions
  // >= sl_n_sorted hold node p at position p (appended since the last rebuild; a batch appended
  // as a sorted run is in cell order inside its own range of positions).  chunk_ext: exact fp64 x and y extent of every 512-position chunk (enc_ord).
  // Inside a chunk the positions are lane-major: scan lane L owns positions 8 L .. 8 L + 7, so
  // its eight nodes are two 16-byte loads per array and the exact re-test of a flagged lane
  // reads eight consecutive doubles (sl_d: the same order in fp64).
  float *sl_f[4] = {nullptr, nullptr, nullptr, nullptr};
  double *sl_d[4] = {nullptr, nullptr, nullptr, nullptr};
  float *sl_pp = nullptr;
  int32_t *sl_id = nullptr;
  rrtx::ChunkExtHost *chunk_ext = nullptr;   // x / y extent of every chunk (enc_ord), see exact_math.hpp ChunkExt
  int64_t cap_chunks = 0;
  int64_t sl_n_sorted = 0;
  double sl_run_chunks = 0.0;       // chunks of the tail that were appended as sorted runs
  int sl_cells = 0;                 // cells of the grid of the last rebuild
  int sl_kz = 0;                    // bins of the third coordinate inside a cell (0: no index yet)
  double sl_debt_us = 0.0;          // what the appended tail has cost the searches since the last rebuild (estimate)
  rrtx::DevBuf d_xrange;            // uint64[6]: enc_ord of min / max node x, min / max node y, min / max of the third coordinate
  rrtx::DevBuf ws_slab_hist, ws_slab_start, ws_slab_sr, ws_slab_params, ws_run_hist, ws_run_sr;
  int run_hist_flip = 0;            // the sorted runs alternate between two cell histograms (each run's place kernel clears the other)

  // options (rrtx_set_option)
  int opt_nn_filter = 1;            // 1: fp32 prefilter + exact fp64 confirm, 0: exact fp64 scan
  int opt_scan_blocks = 1280;       // persistent workgroups of the range scan (256 CUs x 5 resident)
  int opt_scan_items = 2048;        // target number of (tile, segment) work items
  int opt_tile_q = 0;               // query copies per workgroup tile (0 = kernel default)
  int opt_nn_cull = 1;              // 0 off, 1 auto (trees of >= 8192 nodes), 2 always
  long long opt_nearest_rec_cap = 0; // testing: candidate record capacity of the screened nearest scan (0 = default)
  int opt_space_has_time = 0;       // CSpace.spaceHasTime for the Dubins entry points ([x y t theta], R/DRRT_data_structures.jl:330)
  double dubins_vmin = 0.0, dubins_vmax = 1e300;   // S.dubinsMinVelocity / dubinsMaxVelocity (validMove)
  int opt_root_rule = 1;            // 0: node 0 is not the tree's root (this context holds a later node range)
  int opt_tune = 0;                 // experiment switches (RRTX_OPT_TUNE), never change a result
  int opt_profile_every = 1;        // profiling level 1 times every n-th launch of the search kernel
  long long span_tick = 0;

  // wrapped dimensions
  int n_wraps = 0;
  int wrap_dim[rrtx::kMaxWraps] = {0, 0, 0};
  double wrap_period[rrtx::kMaxWraps] = {0, 0, 0};

  // sphere obstacles: host truth (list order) + packed active device copy
  std::vector<double> sph;          // m x 4
  std::vector<uint8_t> sph_active;  // m
  bool sph_dirty = true;
  double sph_packed_rr = -1.0;      // robot radius the packed thresholds were built for
  int sph_n_active = 0;
  rrtx::DevBuf d_sph;               // SphRec[n_active]
  rrtx::DevBuf d_sph_reach;         // SphRec[n_active]: centre + inflated reach (conservative skip test)
  rrtx::DevBuf d_sph_reach_f;       // fp32, origin-relative, pair-interleaved copy for the packed screen
  double sph_packed_origin[3] = {0, 0, 0};
  rrtx::DevBuf d_sph_aux;           // double radius[n_active] then int32 orig[n_active]
  rrtx::DevBuf d_sph_sample;        // SampleSph[n_active]: one record per sphere for sample_spheres_kernel

  // polygon obstacles
  std::vector<int32_t> poly_off;    // m+1
  std::vector<double> poly_vxy;
  std::vector<double> poly_cr;      // m x 3
  std::vector<uint8_t> poly_kind, poly_active;
  bool poly_dirty = true;
  int poly_n_active = 0;
  rrtx::DevBuf d_poly_slope;   // per vertex v: slope of the side that ends at v, (y_v - y_prev) / (x_v - x_prev) (R/DRRT.jl:1178)
  rrtx::DevBuf d_poly_off, d_poly_vxy, d_poly_meta; // meta: per active obstacle {cx, cy, radius, kind} doubles
  rrtx::DevBuf d_poly_orig;
  // flag-only point checks (points_polygons_kernel's fast path): per packed obstacle the exact bounding box of its
  // vertices (xmin, xmax, ymin, ymax); the sorted y coordinates of every vertex of the packed kind-3 polygons
  rrtx::DevBuf d_poly_bbox, d_poly_ytab;
  rrtx::DevBuf d_poly_pbox;    // per packed obstacle: box of its centre over its whole path (kinds 6 / 7; a point otherwise)
  int poly_n_ytab = 0;
  // uniform grid over the packed obstacles for the flag-only point check (kernels_collide.hip, sync_polygon_grid): per cell
  // the obstacles whose bounding circle or box, padded by poly_grid_pad, reaches the cell
  rrtx::DevBuf d_poly_grid_start, d_poly_grid_items;
  std::vector<double> poly_h_meta, poly_h_bbox;      // host copies of the packed records the grid is built from
  double poly_grid_pad = -1.0;                        // < 0: no grid
  double poly_grid_x0 = 0.0, poly_grid_y0 = 0.0, poly_grid_inv_wx = 0.0, poly_grid_inv_wy = 0.0;
  int poly_grid_g = 0;
  rrtx::DevBuf ws_knn_off, ws_knn_idx, ws_knn_dist, ws_knn_misc;   // k-nearest via range-search lists
  int opt_knn_lists = 1;
  int opt_extend_polygons = 0;   // rrtx_extend_candidates checks against the polygon list instead of the spheres
  // kinds 6 / 7 (polygons moving in time): per obstacle rows of (dx, dy, t), CSR over all m obstacles
  std::vector<int32_t> poly_path_off;
  std::vector<double> poly_path;
  rrtx::DevBuf d_poly_path_off, d_poly_path;   // packed over the active obstacles like d_poly_off
  bool poly_has_moving = false;                // an active obstacle of kind 6 or 7 exists

  // workspaces
  rrtx::DevBuf ws_q;        // staged queries / points
  rrtx::DevBuf ws_q2;
  rrtx::DevBuf ws_slots;    // SlotRec table
  rrtx::DevBuf ws_copies;   // QRec copies
  rrtx::DevBuf ws_copies_f; // fp32 prefilter copies
  rrtx::DevBuf ws_copy_meta;// int32 owner, slot per copy
  rrtx::DevBuf ws_copies_s, ws_meta_s;  // copies / meta in bucket order (culled scan)
  rrtx::DevBuf ws_cb, ws_qhist;         // (bucket, rank) per copy; bucket histogram
  rrtx::DevBuf ws_bkt;      // per-query hit buckets (16-byte records)
  int bkt_mult = 2;         // bucket capacity in units of the average list length the caller made room for
  unsigned *mailbox = nullptr;   // host-mapped words the finish kernel reports to: [0] overflow records of a call
  rrtx::DevBuf ws_ev_a, ws_ev_cnt, ws_confirm_args;   // per-wave entry slices, their counts, confirm arguments
  rrtx::DevBuf ws_recs;     // HitRec
  rrtx::DevBuf ws_counts;   // int32 count[nq], cursor[nq]
  rrtx::DevBuf ws_bsum;     // int64 per-256-query sums of count (first level of the offsets scan)
  rrtx::DevBuf ws_scalars;  // device scalars of the range search: two records used alternately
  rrtx::DevBuf ws_scalars_nn;  // ... of the nearest search
  int scalars_flip = 0;
  rrtx::DevBuf ws_tmp;      // list entries that did not fit their bucket, in CSR position
  rrtx::DevBuf ws_owner;    // int32 owner query of every CSR entry (extend_candidates)
  rrtx::DevBuf ws_dub_rec;  // 128-byte records of the steered Dubins edges of one chunk (kernels_dubins.hip)
  rrtx::DevBuf ws_out_off, ws_out_idx, ws_out_dist, ws_out_u8a, ws_out_u8b, ws_out_i32, ws_out_f64;
  rrtx::DevBuf ws_partial;  // nearest partials
  rrtx::DevBuf ws_thr;      // per-query thresholds
  rrtx::DevBuf ws_mask;     // per-call obstacle mask (packed order)
  rrtx::DevBuf ws_i32a, ws_i32b;  // staged index arrays
  rrtx::DevBuf ws_sph_lists;      // per sample: spheres its candidate edges can touch (+ counts)
  rrtx::DevBuf ws_poly_lists;     // per sample: polygons its candidate edges can reach (+ counts), uint16

  // device mirror of the planner's directed edges (obstacle sweeps, kernels_sweep.hip)
  int32_t *ge_start = nullptr, *ge_end = nullptr;
  double *ge_dist = nullptr;        // edge.dist of every mirrored edge (Inf = blocked); SimpleEdge cost by default
  uint8_t *ge_dirty = nullptr;      // edge cost touched since the last cost solve (set_dist / block)
  rrtx::GraphCost gc;               // cost propagation state (kernels_graph.hip)
  int64_t ge_n = 0, ge_cap = 0;
  rrtx::DevBuf ws_sweep_mark, ws_sweep_flag, ws_sweep_cnt, ws_sweep_start;

  // radius -> threshold cache
  double thr_cache_r = -1.0, thr_cache_lt = 0.0, thr_cache_gt = 0.0;

  // profiling
  int profiling = 0;                // 0 off, 1 range-scan family only, 2 every family
  bool span_open = false;
  std::vector<rrtx::TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  double fam_ms[rrtx::KF_COUNT] = {0};
  int64_t fam_launches[rrtx::KF_COUNT] = {0};
  int64_t last_pairs = 0, last_neighbors = 0;
  int last_tile_q = 0;
  bool last_culled = false;         // the last range search used the slab-culled scan
  int64_t last_sweep_candidates = 0;
  int last_visit_slices = 0;        // entries of ws_ev_cnt holding its per-wave chunk counts
};

namespace rrtx {

int fail(rrtx_ctx *ctx, int code, const char *fmt, ...);

// roctx range for the duration of one C-ABI call (rrtx_capi.hip); a no-op without a marker library
struct ApiRange {
  explicit ApiRange(const char *name);
  ~ApiRange();
  bool active;
};

#define RRTX_HIP(ctx, expr)                                                               \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return ::rrtx::fail((ctx), RRTX_E_DEVICE, "%s failed: %s (%s:%d)", #expr,           \
                          hipGetErrorString(_e), __FILE__, __LINE__);                     \
  } while (0)

// profiling spans around a kernel family
void span_begin(rrtx_ctx *ctx, int family);
void span_end(rrtx_ctx *ctx);

// thresholds on squared distances (host, exact):
//   first_ge(r): smallest s >= 0 with sqrt(s) >= r   (sqrt(s) <  r  <=>  s <  first_ge(r))
//   first_gt(r): smallest s >= 0 with sqrt(s) >  r   (sqrt(s) <= r  <=>  s <  first_gt(r))
double thr_first_ge(double r);
double thr_first_gt(double r);
double thr_point_clear(double robot_radius, double radius);

// ---- launchers (device pointers, enqueue on ctx->stream) ----------------------
// Fused extend() work of the range search (sphere list, culled search only): both directed edges of
// every list entry and explicitPointCheck of the samples; r = radius the lists are built with.
// `fused` tells the caller whether the search did it (otherwise: launch_candidate_edges).
struct ExtendFuse { double r; uint8_t *hit_out, *hit_in, *sample_unsafe; bool fused; };
// the last launch of the range search (kernels_finish.hip); device types passed as void *
struct FinishLaunch {
  const int *count; int nq; int bcap;
  const void *bkt; void *tmp; const void *ovf; long long ovf_cap; const void *scalars;
  bool prescattered;
  int64_t *offsets; int64_t *needed; int32_t *idx; double *dist; long long out_cap;
  int32_t *owner; int32_t *nearest_idx; double *nearest_dist;
  int *qhist; int n_qhist; unsigned *mailbox;
  const double *q; double r_start;
  uint8_t *hit_out, *hit_in;   // fused extend(): the records' edge flags go here (null: not fused)
};
int launch_nn_finish(rrtx_ctx *ctx, const FinishLaunch &f);
int launch_nn_radius(rrtx_ctx *ctx, const double *q_dev, const double *r_dev_or_null, double r_scalar,
                     int nq, int64_t *offsets_dev, int32_t *idx_dev, double *dist_dev, int64_t cap,
                     int64_t *needed_dev, int32_t *owner_dev = nullptr, int32_t *nearest_idx_dev = nullptr,
                     double *nearest_dist_dev = nullptr, ExtendFuse *ext = nullptr);
int knearest_row(int k, int64_t n_nodes);
int launch_nn_knearest(rrtx_ctx *ctx, const double *q_dev, int nq, int k, int32_t *idx_dev, double *dist_dev,
                       int32_t *count_dev);
int launch_nn_nearest(rrtx_ctx *ctx, const double *q_dev, int nq, int32_t *idx_dev, double *dist_dev,
                      bool exact = false);
int scan_units(rrtx_ctx *ctx, int *units);   // (tile, chunk) units of the last culled range scan
int slab_refresh(rrtx_ctx *ctx, long long n_tiles);
bool slab_run_wanted(const rrtx_ctx *ctx, int64_t n);
// what the append kernel needs to draw the (cell, rank) of every node of a sorted run (kernels_slab.hip)
// the grid of the slab index, written on the device at every rebuild (slab_params_kernel)
struct SlabParams { double x0, inv_wx, y0, inv_wy, z0, inv_wz; int Kx, Ky, Kz, pad; };
struct RunRank { const SlabParams *sp; int *hist; int2 *sr; };
int slab_run_prepare(rrtx_ctx *ctx, int64_t n, RunRank *rr);
int slab_append_run(rrtx_ctx *ctx, int64_t base, int64_t n);   // bring the slab index up to date when it pays (kernels_slab.hip)
constexpr int kSlabChunk = 512;              // node positions per chunk of the slab index
int launch_edges_spheres(rrtx_ctx *ctx, const double *p0_dev, const double *p1_dev, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end,
                         uint8_t *hit_dev, int32_t *first_hit_dev, const int32_t *sidx_dev = nullptr,
                         const int32_t *eidx_dev = nullptr, const uint8_t *mask_host = nullptr);
int launch_edges_polygons(rrtx_ctx *ctx, const double *p0_dev, const double *p1_dev, int64_t ne,
                          double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end,
                          uint8_t *hit_dev, int32_t *first_hit_dev);
int launch_points_spheres(rrtx_ctx *ctx, const double *p_dev, int64_t np, double robot_radius, int quick,
                          uint8_t *unsafe_dev, double *clearance_dev);
int launch_points_polygons(rrtx_ctx *ctx, const double *p_dev, int64_t np, double robot_radius,
                           uint8_t *unsafe_dev, double *clearance_dev, double list_r = -1.0,
                           const unsigned short **lists_out = nullptr, const unsigned short **list_cnt_out = nullptr);
int launch_simple_steer(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne,
                        double *dist_dev, double *wdist_dev);
int launch_dubins_steer(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double r_min,
                        double *cost_dev, uint8_t *word_dev, double *wdist_dev = nullptr,
                        double *velocity_dev = nullptr, uint8_t *valid_dev = nullptr);
int launch_dubins_edges_check(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne,
                              double r_min, double robot_radius, double *cost_dev, uint8_t *word_dev,
                              uint8_t *hit_dev, int32_t *traj_len_dev, int pb = -1, int pe = -1);
int launch_dubins_edges_idx(rrtx_ctx *ctx, const int32_t *ids_dev, int64_t n, double r_min, double robot_radius, int pb,
                            int pe, uint8_t *hit_dev);
int launch_sweep_mark_multi(rrtx_ctx *ctx, const void *queries_host, int nqs);
int launch_sweep_select(rrtx_ctx *ctx, int blocked_only, int32_t *out_dev, long long cap, long long **total_dev);
int launch_sweep_finish(rrtx_ctx *ctx, const int32_t *ids_dev, long long n, const uint8_t *hit, const uint8_t *o1,
                        const uint8_t *o2, int32_t *out_dev, long long cap, long long **total_dev);
int launch_sweep_gather(rrtx_ctx *ctx, const int32_t *ids_dev, long long n, double *p0_dev, double *p1_dev);
void packed_range(const std::vector<int32_t> &orig, int begin, int end, int &pb, int &pe);
std::vector<int32_t> active_positions(const std::vector<uint8_t> &active);
int launch_candidate_dubins(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                            const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap, double r_min,
                            double robot_radius, double *cost_out, double *cost_in, uint8_t *word_out,
                            uint8_t *word_in, uint8_t *hit_out, uint8_t *hit_in);
int launch_detmath_eval(rrtx_ctx *ctx, int op, const double *x_dev, const double *y_dev, int64_t n, double *out_dev);
int launch_dubins_trajectory(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double r_min,
                             const int64_t *traj_off_dev, double *traj_xy_dev, int64_t cap_rows,
                             int32_t *traj_len_dev);
// candidate edges of extend(): for every CSR entry both directed edges vs the sphere list
int launch_candidate_edges(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                           const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap, double robot_radius,
                           uint8_t *hit_out_dev, uint8_t *hit_in_dev, double r = -1.0,
                           uint8_t *sample_unsafe_dev = nullptr);
int launch_candidate_edges_polygons(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                                    const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap,
                                    double robot_radius, uint8_t *hit_out_dev, uint8_t *hit_in_dev,
                                    uint8_t *sample_unsafe_dev, double r = -1.0);
int launch_nearest_from_lists(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                              const int32_t *idx_dev, const double *dist_dev, int32_t *nearest_idx_dev,
                              double *nearest_dist_dev);

int launch_obstacle_sweep(rrtx_ctx *ctx, const double centre[3], double thr_lt, double thr_gt, const SphRec &ob,
                          int active, int32_t *out_dev, int64_t cap, long long **total_dev);

int launch_graph_edge_dist(rrtx_ctx *ctx, long long first, long long n);
int launch_graph_touch(rrtx_ctx *ctx, long long first, long long n);
int launch_graph_block(rrtx_ctx *ctx, const int32_t *ids_host, long long n);
int launch_graph_cost(rrtx_ctx *ctx, int root, bool update, double *lmc_dev, int32_t *parent_dev, int *passes_out);
void graph_cost_forget(rrtx_ctx *ctx);

int launch_pack_hits(rrtx_ctx *ctx, const uint8_t *hit_out, const uint8_t *hit_in, const int64_t *n_valid_dev,
                     int64_t cap, uint64_t *words);

// make sure the packed obstacle tables on the device match the host truth
int sync_spheres(rrtx_ctx *ctx, double robot_radius);
int sync_polygons(rrtx_ctx *ctx);
int sync_polygon_grid(rrtx_ctx *ctx, double pad_needed);

}  // namespace rrtx
