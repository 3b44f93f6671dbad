// kernels_dubins.hip -- Dubins steering (six-word shortest path) and the
// two-stage Dubins edge check against polygon obstacles.  Replaces
// calculateTrajectory(S, ::DubinsEdge) and explicitEdgeCheck(S, ::DubinsEdge, ob)
// (R/DRRT_DubinsEdge_functions.jl:329-709, 750-774; helpers
// R/DRRT_distance_functions.jl:62-80), in a space without time and -- template TIME,
// CSpace.spaceHasTime -- in [x y t theta] with the time-stamped trajectory of :660-697, validMove of
// :115-121 and obstacles that move in time (kinds 6 / 7, R/DRRT.jl:1579-1651).  gfx950 only.
//
// Transcendentals (sin, cos, atan2, acos, the rows of a sampled arc) are NOT the ROCm device library's: they are
// include/rrtx_detmath.h, one deterministic sequence of IEEE operations compiled into this file and into the CPU
// checker alike, so the two agree to the last bit where the reference's branches decide on it (`theta < 0 ->
// + 2 pi`, strict `bestDist > len`, a piece grazing a polygon side).  Against Julia's own libm the costs keep the
// 1e-6 relative tolerance north_star allows; everything else keeps the reference's operation order (no FMA
// contraction).
#include "../../include/rrtx_detmath.h"
#include "collide_device.hpp"
#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {

namespace {

constexpr double kPi = 3.141592653589793;

__device__ __forceinline__ double right_turn_dist(double ax, double ay, double bx, double by, double cx,
                                                  double cy, double r) {
  double theta = rrtx_dm_atan2(ay - cy, ax - cx) - rrtx_dm_atan2(by - cy, bx - cx);
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
__device__ __forceinline__ double left_turn_dist(double ax, double ay, double bx, double by, double cx,
                                                 double cy, double r) {
  double theta = rrtx_dm_atan2(by - cy, bx - cx) - rrtx_dm_atan2(ay - cy, ax - cx);
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
// The same two functions with one of the two angles given: the angle of the start pose seen from its right / left
// circle and of the goal pose seen from its circles enter three words each (and the stored trajectory), always
// through the same expression rrtx_dm_atan2(y - cy, x - cx) of the same operands, so they are computed once per edge
// (16 atan2 per edge instead of 24 + 2 of the trajectory's 6; the values are the identical doubles).
__device__ __forceinline__ double right_turn_from(double angle_a, double bx, double by, double cx, double cy, double r) {
  double theta = angle_a - rrtx_dm_atan2(by - cy, bx - cx);
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
__device__ __forceinline__ double right_turn_to(double ax, double ay, double angle_b, double cx, double cy, double r) {
  double theta = rrtx_dm_atan2(ay - cy, ax - cx) - angle_b;
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
__device__ __forceinline__ double left_turn_from(double angle_a, double bx, double by, double cx, double cy, double r) {
  double theta = rrtx_dm_atan2(by - cy, bx - cx) - angle_a;
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
__device__ __forceinline__ double left_turn_to(double ax, double ay, double angle_b, double cx, double cy, double r) {
  double theta = angle_b - rrtx_dm_atan2(ay - cy, ax - cx);
  if (theta < 0) theta = theta + 2 * kPi;
  return theta * r;
}
__device__ __forceinline__ double seg_len2(double ax, double ay, double bx, double by) {
  return sqrt_rn(sq2(ax, ay, bx, by));
}

// one piece of the stored trajectory: an arc sampled every 0.1 rad or a straight
// segment given by its two end points
struct Piece {
  double cx, cy;      // arc centre | line p1
  double a, b;        // arc cos(phi_start), sin(phi_start) | line p2
  int len;            // number of polyline rows
  int kind;           // 0 arc with step +0.1, 2 arc with step -0.1, 1 line
};

// cos / sin of k * 0.1: row k of an arc is at phi_start -+ k * 0.1 (StepRangeLen element of
// collect(phi_start:-+0.1:phi_end)), so its point is one angle addition on the arc's own cos / sin of
// phi_start instead of a cos and a sin per row (the polyline walk is most of the Dubins edge check).
// The two forms differ by an ulp or two in the coordinates -- as device and Julia libm do anyway.
constexpr int kArcTab = RRTX_DM_ARC_TAB;           // an arc spans less than 2 pi: at most 63 rows
__device__ const double kArcCos[kArcTab] = RRTX_DM_ARC_COS_INIT;
__device__ const double kArcSin[kArcTab] = RRTX_DM_ARC_SIN_INIT;

struct Steer {
  double cost;
  int word;           // 0 rsl 1 rsr 2 rlr 3 lsr 4 lsl 5 lrl 6 xxx
  Piece pc[3];
};

// length of start:step:stop (Julia float range, literal fallback branch)
__device__ __forceinline__ int julia_range_len(double start, double step, double stop) {
  double lf = (stop - start) / step;
  if (lf < 0) return 0;
  if (lf == 0) return 1;
  long long len = (long long)rint(lf) + 1;
  double stop2 = start + (double)(len - 1) * step;
  len -= ((start < stop && stop < stop2) ? 1 : 0) + ((start > stop && stop > stop2) ? 1 : 0);
  return (int)len;
}

__device__ __forceinline__ Piece make_arc(double cx, double cy, double phi_start, double phi_end, double step) {
  Piece p;
  p.cx = cx; p.cy = cy; p.kind = step < 0.0 ? 2 : 0;
  rrtx_dm_sincos(phi_start, &p.b, &p.a);
  p.len = (phi_end == phi_start) ? 1 : julia_range_len(phi_start, step, phi_end);
  return p;
}
__device__ __forceinline__ Piece make_line(double x1, double y1, double x2, double y2) {
  Piece p;
  p.cx = x1; p.cy = y1; p.a = x2; p.b = y2; p.kind = 1; p.len = 2;
  return p;
}

// What calculateTrajectory derives from ONE pose alone: the centres of its right / left turning circles (:348-357) and
// the pose's angle on each of them (see right_turn_from).  An edge and its reverse use the same two poses with the
// roles of start and goal swapped -- the expressions are the same, so the values are shared.
struct PoseCircles { double rcx, rcy, lcx, lcy, a_r, a_l; };
__device__ __forceinline__ PoseCircles pose_circles(double x, double y, double th, double r_min) {
  PoseCircles p;
  double sn, cs;
  rrtx_dm_sincos(th - kPi / 2.0, &sn, &cs);
  p.rcx = x + r_min * cs; p.rcy = y + r_min * sn;
  rrtx_dm_sincos(th + kPi / 2.0, &sn, &cs);
  p.lcx = x + r_min * cs; p.lcy = y + r_min * sn;
  p.a_r = rrtx_dm_atan2(y - p.rcy, x - p.rcx);
  p.a_l = rrtx_dm_atan2(y - p.lcy, x - p.lcx);
  return p;
}

template <bool WANT_TRAJ>
__device__ __forceinline__ void dubins_steer_poses(const double *__restrict__ s, const double *__restrict__ g, const PoseCircles &ps,
                                   const PoseCircles &pg, double r_min, Steer &out) {
  const double ilx = s[0], ily = s[1];
  const double glx = g[0], gly = g[1];
  const double ircx = ps.rcx, ircy = ps.rcy, ilcx = ps.lcx, ilcy = ps.lcy;
  const double grcx = pg.rcx, grcy = pg.rcy, glcx = pg.lcx, glcy = pg.lcy;
  const double a_ir = ps.a_r, a_il = ps.a_l, a_gr = pg.a_r, a_gl = pg.a_l;
  double sn, cs;

  double best = __builtin_inf();
  int word = 6;
  double D, vx, vy, R, sq, a, b, first, second, third, len;

  // rsl (:367-388)
  double rsl1x = 0, rsl1y = 0, rsl2x = 0, rsl2y = 0;
  {
    double dx = glcx - ircx, dy = glcy - ircy;
    D = sqrt_rn(dx * dx + dy * dy);
    vx = dx / D; vy = dy / D;
    R = -2.0 * r_min / D;
    if (!(fabs(R) > 1.0)) {
      sq = sqrt_rn(1.0 - R * R);
      a = r_min * (R * vx + vy * sq);
      b = r_min * (R * vy - vx * sq);
      rsl1x = ircx - a; rsl2x = glcx + a;
      rsl1y = ircy - b; rsl2y = glcy + b;
      first = right_turn_from(a_ir, rsl1x, rsl1y, ircx, ircy, r_min);
      second = seg_len2(rsl2x, rsl2y, rsl1x, rsl1y);
      third = left_turn_to(rsl2x, rsl2y, a_gl, glcx, glcy, r_min);
      len = first + second + third;
      if (best > len) { best = len; word = 0; }
    }
  }
  // rsr (:394-407)
  double rsr1x, rsr1y, rsr2x, rsr2y;
  {
    double dx = grcx - ircx, dy = grcy - ircy;
    D = sqrt_rn(dx * dx + dy * dy);
    vx = dx / D; vy = dy / D;
    rsr1x = -r_min * vy + ircx; rsr2x = -r_min * vy + grcx;
    rsr1y = r_min * vx + ircy;  rsr2y = r_min * vx + grcy;
    first = right_turn_from(a_ir, rsr1x, rsr1y, ircx, ircy, r_min);
    second = seg_len2(rsr2x, rsr2y, rsr1x, rsr1y);
    third = right_turn_to(rsr2x, rsr2y, a_gr, grcx, grcy, r_min);
    len = first + second + third;
    if (best > len) { best = len; word = 1; }
  }
  // rlr (:411-431; D, v from rsr)
  double rlr_cx = 0, rlr_cy = 0, rlr_rlx = 0, rlr_rly = 0, rlr_lrx = 0, rlr_lry = 0;
  if (D < 4.0 * r_min) {
    double theta = -rrtx_dm_acos(D / (4 * r_min)) + rrtx_dm_atan2(vy, vx);
    rrtx_dm_sincos(theta, &sn, &cs);
    rlr_cx = ircx + 2 * r_min * cs;
    rlr_cy = ircy + 2 * r_min * sn;
    rlr_rlx = (rlr_cx + ircx) / 2.0; rlr_rly = (rlr_cy + ircy) / 2.0;
    rlr_lrx = (rlr_cx + grcx) / 2.0; rlr_lry = (rlr_cy + grcy) / 2.0;
    first = right_turn_from(a_ir, rlr_rlx, rlr_rly, ircx, ircy, r_min);
    second = left_turn_dist(rlr_rlx, rlr_rly, rlr_lrx, rlr_lry, rlr_cx, rlr_cy, r_min);
    third = right_turn_to(rlr_lrx, rlr_lry, a_gr, grcx, grcy, r_min);
    len = first + second + third;
    if (best > len) { best = len; word = 2; }
  }
  // lsr (:436-458)
  double lsr1x = 0, lsr1y = 0, lsr2x = 0, lsr2y = 0;
  {
    double dx = grcx - ilcx, dy = grcy - ilcy;
    D = sqrt_rn(dx * dx + dy * dy);
    vx = dx / D; vy = dy / D;
    R = 2.0 * r_min / D;
    if (!(fabs(R) > 1)) {
      sq = sqrt_rn(1 - R * R);
      a = R * vx + vy * sq;
      b = R * vy - vx * sq;
      lsr1x = ilcx + a * r_min; lsr2x = grcx - a * r_min;
      lsr1y = ilcy + b * r_min; lsr2y = grcy - b * r_min;
      first = left_turn_from(a_il, lsr1x, lsr1y, ilcx, ilcy, r_min);
      second = seg_len2(lsr2x, lsr2y, lsr1x, lsr1y);
      third = right_turn_to(lsr2x, lsr2y, a_gr, grcx, grcy, r_min);
      len = first + second + third;
      if (best > len) { best = len; word = 3; }
    }
  }
  // lsl (:464-477)
  double lsl1x, lsl1y, lsl2x, lsl2y;
  {
    double dx = glcx - ilcx, dy = glcy - ilcy;
    D = sqrt_rn(dx * dx + dy * dy);
    vx = dx / D; vy = dy / D;
    lsl1x = r_min * vy + ilcx;  lsl2x = r_min * vy + glcx;
    lsl1y = -r_min * vx + ilcy; lsl2y = -r_min * vx + glcy;
    first = left_turn_from(a_il, lsl1x, lsl1y, ilcx, ilcy, r_min);
    second = seg_len2(lsl2x, lsl2y, lsl1x, lsl1y);
    third = left_turn_to(lsl2x, lsl2y, a_gl, glcx, glcy, r_min);
    len = first + second + third;
    if (best > len) { best = len; word = 4; }
  }
  // lrl (:481-501; D, v from lsl)
  double lrl_cx = 0, lrl_cy = 0, lrl_lrx = 0, lrl_lry = 0, lrl_rlx = 0, lrl_rly = 0;
  if (D < 4.0 * r_min) {
    double theta = rrtx_dm_acos(D / (4 * r_min)) + rrtx_dm_atan2(vy, vx);
    rrtx_dm_sincos(theta, &sn, &cs);
    lrl_cx = ilcx + 2.0 * r_min * cs;
    lrl_cy = ilcy + 2.0 * r_min * sn;
    lrl_lrx = (lrl_cx + ilcx) / 2.0; lrl_lry = (lrl_cy + ilcy) / 2.0;
    lrl_rlx = (lrl_cx + glcx) / 2.0; lrl_rly = (lrl_cy + glcy) / 2.0;
    first = left_turn_from(a_il, lrl_lrx, lrl_lry, ilcx, ilcy, r_min);
    second = right_turn_dist(lrl_lrx, lrl_lry, lrl_rlx, lrl_rly, lrl_cx, lrl_cy, r_min);
    third = left_turn_to(lrl_rlx, lrl_rly, a_gl, glcx, glcy, r_min);
    len = first + second + third;
    if (best > len) { best = len; word = 5; }
  }

  out.cost = best;
  out.word = word;
  if (!WANT_TRAJ) return;
  out.pc[0].len = out.pc[1].len = out.pc[2].len = 0;
  if (word == 6 || best == __builtin_inf()) return;   // no trajectory is built (:661-662)

  const double dphi = .1;
  // The three pieces (:511-655).  Which circle, which tangent point and which sense of rotation a piece has depends on the
  // word; the operands are SELECTED first and every expression is then written once, so that a wave whose lanes hold
  // different words evaluates one atan2 / sincos per piece instead of one per branch (the values are the reference's:
  // the same expressions on the same operands).
  const bool fr = (word <= 2);                        // first letter 'r'
  double px, py, cx, cy, phi_start, phi_end;
  // first piece (:511-555)
  cx = fr ? ircx : ilcx; cy = fr ? ircy : ilcy;
  if (fr) { px = word == 0 ? rsl1x : (word == 1 ? rsr1x : rlr_rlx); py = word == 0 ? rsl1y : (word == 1 ? rsr1y : rlr_rly); }
  else { px = word == 4 ? lsl1x : (word == 3 ? lsr1x : lrl_lrx); py = word == 4 ? lsl1y : (word == 3 ? lsr1y : lrl_lry); }
  phi_start = fr ? a_ir : a_il;
  phi_end = rrtx_dm_atan2(py - cy, px - cx);
  if (fr) { if (phi_end > phi_start) phi_end = phi_end - 2.0 * kPi; }
  else { if (phi_end < phi_start) phi_end = phi_end + 2.0 * kPi; }
  out.pc[0] = make_arc(cx, cy, phi_start, phi_end, fr ? -dphi : dphi);
  // second piece (:559-608)
  if (word == 2 || word == 5) {
    const bool mr = word == 5;                        // lrl: the middle turn is a right turn; rlr: a left turn
    cx = mr ? lrl_cx : rlr_cx; cy = mr ? lrl_cy : rlr_cy;
    const double ax = mr ? lrl_lrx : rlr_rlx, ay = mr ? lrl_lry : rlr_rly;
    const double bx = mr ? lrl_rlx : rlr_lrx, by = mr ? lrl_rly : rlr_lry;
    phi_start = rrtx_dm_atan2(ay - cy, ax - cx);
    phi_end = rrtx_dm_atan2(by - cy, bx - cx);
    if (mr) { if (phi_end > phi_start) phi_end = phi_end - 2.0 * kPi; }
    else { if (phi_end < phi_start) phi_end = phi_end + 2.0 * kPi; }
    out.pc[1] = make_arc(cx, cy, phi_start, phi_end, mr ? -dphi : dphi);
  } else {
    const double x1 = word == 0 ? rsl1x : (word == 1 ? rsr1x : (word == 3 ? lsr1x : lsl1x));
    const double y1 = word == 0 ? rsl1y : (word == 1 ? rsr1y : (word == 3 ? lsr1y : lsl1y));
    const double x2 = word == 0 ? rsl2x : (word == 1 ? rsr2x : (word == 3 ? lsr2x : lsl2x));
    const double y2 = word == 0 ? rsl2y : (word == 1 ? rsr2y : (word == 3 ? lsr2y : lsl2y));
    out.pc[1] = make_line(x1, y1, x2, y2);
  }
  // third piece (:611-655)
  const bool tr = (word == 1 || word == 3 || word == 2);   // last letter 'r'
  cx = tr ? grcx : glcx; cy = tr ? grcy : glcy;
  if (tr) { px = word == 1 ? rsr2x : (word == 3 ? lsr2x : rlr_lrx); py = word == 1 ? rsr2y : (word == 3 ? lsr2y : rlr_lry); }
  else { px = word == 4 ? lsl2x : (word == 0 ? rsl2x : lrl_rlx); py = word == 4 ? lsl2y : (word == 0 ? rsl2y : lrl_rly); }
  phi_start = rrtx_dm_atan2(py - cy, px - cx);
  phi_end = tr ? a_gr : a_gl;
  if (tr) { if (phi_end > phi_start) phi_end = phi_end - 2.0 * kPi; }
  else { if (phi_end < phi_start) phi_end = phi_end + 2.0 * kPi; }
  out.pc[2] = make_arc(cx, cy, phi_start, phi_end, tr ? -dphi : dphi);
}

template <bool WANT_TRAJ>
__device__ void dubins_steer(const double *__restrict__ s, const double *__restrict__ g, double r_min, Steer &out) {
  const PoseCircles ps = pose_circles(s[0], s[1], s[3], r_min), pg = pose_circles(g[0], g[1], g[3], r_min);
  dubins_steer_poses<WANT_TRAJ>(s, g, ps, pg, r_min, out);
}

__device__ __forceinline__ void piece_point(const Piece &p, int k, double r_min, double &x, double &y) {
  if (p.kind == 1) {
    x = (k == 0) ? p.cx : p.a;
    y = (k == 0) ? p.cy : p.b;
  } else {
    // phi = phi_start -+ k * 0.1
    const int kc = k < kArcTab ? k : kArcTab - 1;
    double ck = kArcCos[kc], sk = kArcSin[kc];
    if (k >= kArcTab) rrtx_dm_sincos((double)k * .1, &sk, &ck);     // (cannot happen: an arc is < 2 pi)
    rrtx_dm_arc_row(p.cx, p.cy, r_min, p.a, p.b, ck, sk, p.kind == 2, &x, &y);
  }
}

// Time column of edge.trajectory (R/DRRT_DubinsEdge_functions.jl:682-696): row 1 = the start node's
// time, rows 2 .. P-1 = start time - (distance walked along the stored polyline) / velocity, last
// row = the end node's (x, y, t).  The reference adds the straight pieces up one by one; here the
// distance at a row is (distance at the first row of its piece) + (rows into the piece) x (the
// piece's step): inside an arc every step is the same chord, and the three junctions are measured
// once.  The two sums differ by rounding only (~1e-15 relative), which is inside the tolerance
// north_star gives Dubins edge costs; only a collision test on a knife edge can tell.
struct TimeInfo {
  double st, et, vel;      // start / end time, edge.velocity = Wdist / (st - et)
  double gx, gy;           // end node (the last row is made exact)
  double cum[3], chord[3];
  int P;
};

__device__ __forceinline__ TimeInfo time_info(const Steer &st, const double *__restrict__ s,
                                              const double *__restrict__ g, double r_min) {
  TimeInfo ti;
  ti.st = s[2]; ti.et = g[2];
  ti.vel = st.cost / (s[2] - g[2]);
  ti.gx = g[0]; ti.gy = g[1];
  ti.P = st.pc[0].len + st.pc[1].len + st.pc[2].len;
  double run = 0.0, lx = 0.0, ly = 0.0;
  bool have_last = false;
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    const Piece &p = st.pc[pi];
    ti.cum[pi] = run; ti.chord[pi] = 0.0;
    if (p.len <= 0) continue;
    double x0, y0;
    piece_point(p, 0, r_min, x0, y0);
    if (have_last) { run = run + seg_len2(lx, ly, x0, y0); ti.cum[pi] = run; }
    if (p.len > 1) {
      double x1, y1;
      piece_point(p, 1, r_min, x1, y1);
      ti.chord[pi] = seg_len2(x0, y0, x1, y1);
      run = run + (double)(p.len - 1) * ti.chord[pi];
      piece_point(p, p.len - 1, r_min, lx, ly);
    } else {
      lx = x0; ly = y0;
    }
    have_last = true;
  }
  return ti;
}

// row `row` of the stored polyline with its time stamp
__device__ __forceinline__ void polyline_point_t(const Piece *pc, const TimeInfo &ti, int row, double r_min, double &x,
                                                 double &y, double &t) {
  int pi = 0, k = row;
  if (k >= pc[0].len) { k -= pc[0].len; pi = 1; }
  if (pi == 1 && k >= pc[1].len) { k -= pc[1].len; pi = 2; }
  piece_point(pc[pi], k, r_min, x, y);
  if (row == 0) t = ti.st;
  else if (row == ti.P - 1) { x = ti.gx; y = ti.gy; t = ti.et; }
  else t = ti.st - (ti.cum[pi] + (double)k * ti.chord[pi]) / ti.vel;
}

// edge.dist with S.spaceHasTime: sqrt(bestDist^2 + (start time - end time)^2), Inf stays Inf (:661-667)
__device__ __forceinline__ double dist_with_time(double best, double st, double et) {
  if (best == __builtin_inf()) return best;
  const double dt = st - et;
  return sqrt_rn(best * best + dt * dt);
}
// validMove with S.spaceHasTime (:115-121)
__device__ __forceinline__ bool valid_move_time(double st, double et, double vel, double vmin, double vmax) {
  return (st > et) && (vmin <= vel) && (vel <= vmax);
}

// the polygon list as the Dubins kernels read it
struct PolyTab {
  const double *meta;        // per active obstacle {cx, cy, radius, kind}
  const int32_t *off;        // vertex CSR
  const double *vxy;
  const double *vslope;      // per vertex: slope of the side that ends there (sync_polygons)
  const int32_t *poff;       // path CSR (kinds 6 / 7)
  const double *path;        // rows (dx, dy, t)
  const double *pbox;        // per obstacle: box (xlo, xhi, ylo, yhi) of its centre over its whole path
  int m;
};

__device__ __forceinline__ void write_word(uint8_t *__restrict__ word, long long i, int w) {
  const char *tab = "rslrsrrlrlsrlsllrlxxx";
  word[3 * i + 0] = (uint8_t)tab[3 * w + 0];
  word[3 * i + 1] = (uint8_t)tab[3 * w + 1];
  word[3 * i + 2] = (uint8_t)tab[3 * w + 2];
}

// calculateTrajectory's scalar results.  has_time: edge.dist = sqrt(Wdist^2 + dt^2), edge.velocity and
// validMove (R/DRRT_DubinsEdge_functions.jl:661-682, 115-121); otherwise dist = Wdist and every move is valid.
__global__ __launch_bounds__(256) void dubins_steer_kernel(const double *__restrict__ s,
                                                           const double *__restrict__ g, long long ne,
                                                           double r_min, int has_time, double vmin, double vmax,
                                                           double *__restrict__ cost, double *__restrict__ wdist,
                                                           double *__restrict__ velocity,
                                                           uint8_t *__restrict__ word, uint8_t *__restrict__ valid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  Steer st;
  dubins_steer<false>(s + 4 * i, g + 4 * i, r_min, st);
  const double t0 = s[4 * i + 2], t1 = g[4 * i + 2];
  const double vel = st.cost / (t0 - t1);
  if (cost) cost[i] = has_time ? dist_with_time(st.cost, t0, t1) : st.cost;
  if (wdist) wdist[i] = st.cost;
  if (velocity) velocity[i] = has_time ? vel : 0.0;
  if (valid) valid[i] = (!has_time || valid_move_time(t0, t1, vel, vmin, vmax)) ? 1 : 0;
  if (word) write_word(word, i, st.word);
}

// include/rrtx_detmath.h element-wise (rrtx_detmath_eval: the suite compares this build of the header with the host's)
__global__ __launch_bounds__(256) void detmath_eval_kernel(int op, const double *__restrict__ x, const double *__restrict__ y,
                                                           long long n, double *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  if (op == 0) r = rrtx_dm_sin(x[i]);
  else if (op == 1) r = rrtx_dm_cos(x[i]);
  else if (op == 2) r = rrtx_dm_atan2(y[i], x[i]);
  else r = rrtx_dm_acos(x[i]);
  out[i] = r;
}

// polygon helpers live in kernels_collide.hip; the Dubins check needs the same
// arithmetic, restated here to keep the translation units independent.
__device__ __forceinline__ double dspts(double px, double py, double ax, double ay, double bx, double by) {
  double vx = px - ax, vy = py - ay;
  double ux = bx - ax, uy = by - ay;
  double det = vx * ux + vy * uy;
  if (det <= 0) return vx * vx + vy * vy;
  double len = ux * ux + uy * uy;
  if (det >= len) { double ex = bx - px, ey = by - py; return ex * ex + ey * ey; }
  double cr = ux * vy - uy * vx;
  return (cr * cr) / len;
}
__device__ double seg_dist_sqrd(double pax, double pay, double pbx, double pby, double qax, double qay,
                                double qbx, double qby) {
  bool possible = true;
  if (fabs(pbx - pax) < .000001) {
    if ((qax >= pax && qbx >= pax) || (qax <= pax && qbx <= pax)) possible = false;
  } else {
    double m = (pby - pay) / (pbx - pax);
    double diffA = (m * (qax - pax) + pay) - qay;
    double diffB = (m * (qbx - pax) + pay) - qby;
    if ((diffA > 0.0 && diffB > 0.0) || (diffA < 0.0 && diffB < 0.0)) possible = false;
  }
  if (possible) {
    if (fabs(qbx - qax) < .000001) {
      if ((pax >= qax && pbx >= qax) || (pax <= qax && pbx <= qax)) possible = false;
    } else {
      double m = (qby - qay) / (qbx - qax);
      double diffA = (m * (pax - qax) + qay) - pay;
      double diffB = (m * (pbx - qax) + qay) - pby;
      if ((diffA > 0.0 && diffB > 0.0) || (diffA < 0.0 && diffB < 0.0)) possible = false;
    }
  }
  if (possible) return 0.0;
  double r = dspts(pax, pay, qax, qay, qbx, qby);
  r = jl_min(r, dspts(pbx, pby, qax, qay, qbx, qby));
  r = jl_min(r, dspts(qax, qay, pax, pay, pbx, pby));
  r = jl_min(r, dspts(qbx, qby, pax, pay, pbx, pby));
  return r;
}
// explicitEdgeCheck2D (R/DRRT.jl:1523-1578) in two steps: the bounding-circle test (:1536-1539) ...
__device__ __forceinline__ bool seg_outside_circle(double ax, double ay, double bx, double by, double robot_radius,
                                                   const double *__restrict__ meta, int j) {
  const double dsq = dspts(meta[4 * j + 0], meta[4 * j + 1], ax, ay, bx, by);
  const double rr = robot_radius + meta[4 * j + 2];
  return dsq > rr * rr;
}
// ... and what follows it (:1542-1577)
__device__ bool seg_hits_polygon_past_circle(double ax, double ay, double bx, double by, double robot_radius,
                                             const double *__restrict__ meta, const int32_t *__restrict__ off,
                                             const double *__restrict__ vxy, int j);
__device__ bool seg_hits_polygon(double ax, double ay, double bx, double by, double robot_radius,
                                 const double *__restrict__ meta, const int32_t *__restrict__ off,
                                 const double *__restrict__ vxy, int j) {
  if (seg_outside_circle(ax, ay, bx, by, robot_radius, meta, j)) return false;
  return seg_hits_polygon_past_circle(ax, ay, bx, by, robot_radius, meta, off, vxy, j);
}
__device__ bool seg_hits_polygon_past_circle(double ax, double ay, double bx, double by, double robot_radius,
                                             const double *__restrict__ meta, const int32_t *__restrict__ off,
                                             const double *__restrict__ vxy, int j) {
  const int kind = (int)meta[4 * j + 3];
  if (kind == 1) return true;
  if (kind == 3) {
    const int b = off[j], e = off[j + 1];
    if (e - b < 2) return false;
    double Ax = vxy[2 * (e - 1)], Ay = vxy[2 * (e - 1) + 1];
    const double rr2 = robot_radius * robot_radius;
    for (int v = b; v < e; ++v) {
      double Bx = vxy[2 * v], By = vxy[2 * v + 1];
      if (seg_dist_sqrd(ax, ay, bx, by, Ax, Ay, Bx, By) < rr2) return true;
      Ax = Bx; Ay = By;
    }
  }
  return false;
}

// Two-stage test of the steered edges of one wave against the polygon list (:750-774).  The
// reference walks the obstacle list and, for an obstacle whose inflated chord test (stage 1) does
// not clear it, every polyline piece (stage 2); the answer is the OR over all (obstacle, piece)
// tests, so the work may be dealt differently:
//   stage 1, a: lane = edge drops the obstacles the chord cannot reach (box test); b: the surviving
//     (edge, obstacle) pairs of the wave are dealt one per lane for the inflated chord test and mark
//     up to 64 obstacles in the edge's bit mask;
//   stage 2, lane = (edge, polyline piece): the next eight pieces of every still-undecided edge of
//     the wave are numbered through (prefix sum) and handed out 64 at a time, so lanes stay busy
//     whatever the lengths of the individual polylines are, and an edge that has collided drops
//     out of the numbering.
// Same set of tests, same arithmetic in each.  Every lane of the wave calls this together.
// Measuring build (python -m rrtqx_3d_amd.build --clocks, tools/dubins_clocks.py): wall-clock ticks (100 MHz) every
// wave spends in the stages of the fused Dubins preamble, summed over the launch:
//   0 steering  1 stage 1a (boxes)  2 stage 1b (chord tests)  3 arc screen  4 stage 2a (pieces, bounding circles)
//   5 stage 2b (polygon tests of the queued pairs)  6 waves
#ifdef RRTX_TILE_CLOCKS
__device__ unsigned long long g_dub_clk[8];
#define RRTX_DUB_T(var) const unsigned long long var = wall_clock64()
#define RRTX_DUB_ADD(k, a, b) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dub_clk[k], (b) - (a)); } while (0)
#define RRTX_DUB_ACC(acc, a, b) acc += (b) - (a)
#else
#define RRTX_DUB_T(var) do { } while (0)
#define RRTX_DUB_ADD(k, a, b) do { } while (0)
#define RRTX_DUB_ACC(acc, a, b) do { } while (0)
#endif

template <bool TIME>
struct WaveDubinsT {
  Piece pc[64][3];
  unsigned long long mask[64];
  unsigned long long cand[64];    // stage 1: obstacles the chord's box reaches; stage 2: those a piece reaches
  union {
    double chord[TIME ? 6 : 4][64];              // stage 1: the edges' chords
    unsigned int pq[(TIME ? 6 : 4) * 128];       // stage 2: queue of (edge, row, obstacle) pairs awaiting the full test
  };
  unsigned long long pm[3][64];   // per piece of the edge: the marked obstacles that piece can touch at all (arc screen)
  unsigned char piece_edge[64];
  unsigned char lc[2][64];        // rows to walk in piece 0 / piece 1 of the edge (0: that piece touches nothing)
  unsigned short nseg[64], rot[64];   // pieces the edge walks in all; where its walk starts (the order is free, see stage 2)
  int pstart[65];
  int pqn;
  unsigned char done[64];
  TimeInfo ti[TIME ? 64 : 1];
};

__device__ __forceinline__ void polyline_point(const Piece *pc, int row, double r_min, double &x, double &y) {
  int pi = 0;
  if (row >= pc[0].len) { row -= pc[0].len; pi = 1; }
  if (pi == 1 && row >= pc[1].len) { row -= pc[1].len; pi = 2; }
  piece_point(pc[pi], row, r_min, x, y);
}

template <bool TIME>
__device__ bool wave_dubins_collides(WaveDubinsT<TIME> &w, bool valid, const Steer &st, const double *__restrict__ sp,
                                     const double *__restrict__ gp, double r_min, double robot_radius,
                                     const PolyTab &tab) {
  const double *__restrict__ meta = tab.meta;
  const int32_t *__restrict__ off = tab.off;
  const double *__restrict__ vxy = tab.vxy;
  const int m = tab.m;
  const double sx = sp[0], sy = sp[1], gx = gp[0], gy = gp[1];
  const int lane = threadIdx.x & 63;
  w.pc[lane][0] = st.pc[0]; w.pc[lane][1] = st.pc[1]; w.pc[lane][2] = st.pc[2];
  if constexpr (TIME) w.ti[lane] = time_info(st, sp, gp, r_min);
  w.done[lane] = 0;
  const int rows = st.pc[0].len + st.pc[1].len + st.pc[2].len;
  for (int j0 = 0; j0 < m; j0 += 64) {
    const int j1 = (j0 + 64 < m) ? j0 + 64 : m;
    RRTX_DUB_T(t_1a);
    // ---- stage 1a (lane = edge): obstacles whose bounding circle, inflated like the chord test, the
    // chord's box cannot reach fail that test for certain (box widened by 1e-9 against ~1e-15 of
    // rounding; NaN / overflow keep the obstacle)
    w.chord[0][lane] = sx; w.chord[1][lane] = sy; w.chord[2][lane] = gx; w.chord[3][lane] = gy;   // (stage 2 reuses it)
    if constexpr (TIME) { w.chord[4][lane] = sp[2]; w.chord[5][lane] = gp[2]; }
    // The obstacles' boxes (bounding circle inflated like the chord test, widened by 1e-9, rounded outward to fp32)
    // are worked out once per wave -- lane b = obstacle j0 + b, kept in w.pm, which is free until the arc screen --
    // and every edge then compares its chord's box (widened and rounded outward the same way) with them: four fp32
    // compares per (edge, obstacle) instead of a dozen fp64 operations.  Disjoint boxes fail the chord test for certain.
    float4 *obox = reinterpret_cast<float4 *>(&w.pm[0][0]);
    if (j0 + lane < j1) {
      const int j = j0 + lane;
      const float inf = __builtin_inff();
      float4 o = {-inf, inf, -inf, inf};
      if (!(TIME && meta[4 * j + 3] >= 6.0)) {
        const double cx = meta[4 * j + 0], cy = meta[4 * j + 1];
        const double R = fabs((robot_radius + 2 * r_min) + meta[4 * j + 2]) * (1.0 + 1e-9) + 1e-9 * (1.0 + fabs(cx) + fabs(cy));
        o.x = __double2float_rd(cx - R); o.y = __double2float_ru(cx + R);
        o.z = __double2float_rd(cy - R); o.w = __double2float_ru(cy + R);
      } else {
        // An obstacle that moves is not where its record says -- but wherever the closest-approach test places it (a point
        // of one of its path segments, R/DRRT.jl:1607-1640: the time of closest approach is clamped into the segment's
        // window) its centre lies in the box of position + path rows, and the robot's centre on the chord.  Boxes farther
        // apart than the test radius + the obstacle's radius: the squared distance the test compares is at least that
        // gap squared -- no hit for any segment.  (NaN sides, NaN comparisons: the obstacle is kept.)
        const double4 pb = reinterpret_cast<const double4 *>(tab.pbox)[j];
        const double R = fabs((robot_radius + 2 * r_min) + meta[4 * j + 2]) * (1.0 + 1e-9) +
                         1e-9 * (1.0 + fabs(pb.x) + fabs(pb.y) + fabs(pb.z) + fabs(pb.w));
        if (R == R && pb.x == pb.x && pb.y == pb.y && pb.z == pb.z && pb.w == pb.w) {
          o.x = __double2float_rd(pb.x - R); o.y = __double2float_ru(pb.y + R);
          o.z = __double2float_rd(pb.z - R); o.w = __double2float_ru(pb.w + R);
        }
      }
      obox[lane] = o;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long cand = 0ull;
    if (valid && !w.done[lane]) {
      // (NaN-propagating min / max: a chord with a NaN coordinate keeps every obstacle -- every compare below is false)
      const double cslack = 1e-9 * jl_max(jl_max(fabs(sx), fabs(gx)), jl_max(fabs(sy), fabs(gy)));
      const float exlo = __double2float_rd(jl_min(sx, gx) - cslack), exhi = __double2float_ru(jl_max(sx, gx) + cslack);
      const float eylo = __double2float_rd(jl_min(sy, gy) - cslack), eyhi = __double2float_ru(jl_max(sy, gy) + cslack);
      for (int b = 0; b < j1 - j0; ++b) {
        const float4 o = obox[b];
        const bool c = !(exhi < o.x || exlo > o.y || eyhi < o.z || eylo > o.w);
        cand |= (c ? 1ull : 0ull) << b;
      }
    }
    __builtin_amdgcn_wave_barrier();      // (the boxes are read before stage 1b's queue takes w.pm)
    w.cand[lane] = cand;
    w.mask[lane] = 0ull;
    int incl = __popcll(cand);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    w.pstart[lane + 1] = incl;
    if (lane == 0) w.pstart[0] = 0;
    __builtin_amdgcn_wave_barrier();
    const int pairs = __shfl(incl, 63);
    RRTX_DUB_T(t_1b);
    RRTX_DUB_ADD(1, t_1a, t_1b);
    // ---- stage 1b: the inflated chord test (:757-760) of the surviving (edge, obstacle) pairs, in two steps that
    // keep the lanes together (as edges_polygons_kernel does, kernels_collide.hip).  A, lane = pair: the polygon
    // test's bounding-circle step, which settles balls and moving obstacles; for a polygon that passes, every
    // segment whose box comes within the test radius of the chord's box is queued (a segment farther away than
    // that in x or in y cannot come out of the segment test below radius^2).  B, lane = queued (pair, segment): the
    // segment test; any segment within the radius marks the obstacle (an OR over the segments).  The queue
    // borrows w.pm, which the arc screen writes afterwards. ----
    {
      const double rad1 = robot_radius + 2 * r_min;
      const double rr2 = rad1 * rad1;
      const double gap_min = fabs(rad1) * (1.0 + 1e-9);
      unsigned int *sq = reinterpret_cast<unsigned int *>(&w.pm[0][0]);
      constexpr int kSqCap = 3 * 64 * 2;                         // 32-bit entries in w.pm
      int nsq = 0;                                               // wave-uniform
      auto seg_round = [&]() {                                   // B: up to 64 queued tests from the end of the queue
        __builtin_amdgcn_wave_barrier();
        const int take = nsq < 64 ? nsq : 64;
        if (lane < take) {
          const unsigned int ent = sq[nsq - take + lane];
          const int e = (int)(ent & 63u), b = (int)((ent >> 6) & 63u), sg = (int)(ent >> 12);
          const int j = j0 + b;
          const int vb0 = off[j], P = off[j + 1] - vb0;
          const int va = vb0 + (sg == 0 ? P - 1 : sg - 1), vb = vb0 + sg;
          if (seg_dist_sqrd(w.chord[0][e], w.chord[1][e], w.chord[2][e], w.chord[3][e], vxy[2 * va], vxy[2 * va + 1],
                            vxy[2 * vb], vxy[2 * vb + 1]) < rr2)
            atomicOr(&w.mask[e], 1ull << b);
        }
        __builtin_amdgcn_wave_barrier();
        nsq -= take;
      };
      for (int p0 = 0; p0 < pairs; p0 += 64) {
        const int p = p0 + lane;
        int e = 0, b = 0, vb0 = 0, P = 0;                        // P > 0: a polygon past the bounding circle
        double elx = 0.0, ehx = 0.0, ely = 0.0, ehy = 0.0, slack = 0.0;
        double pax = 0.0, pay = 0.0, pbx = 0.0, pby = 0.0, em = 0.0;   // the chord's ends and slope (R/DRRT.jl:1158)
        bool evert = false;                                      // the chord is "close to vertical" (:1151)
        if (p < pairs) {
          for (int step = 32; step > 0; step >>= 1)
            if (w.pstart[e + step] <= p) e += step;
          unsigned long long bits = w.cand[e];
          for (int r = p - w.pstart[e]; r > 0; --r) bits &= bits - 1ull;
          b = __ffsll((long long)bits) - 1;
          const int j = j0 + b;
          const double ax = w.chord[0][e], ay = w.chord[1][e], bx = w.chord[2][e], by = w.chord[3][e];
          if (TIME && meta[4 * j + 3] >= 6.0) {
            if (edge_hits_moving(ax, ay, w.chord[TIME ? 4 : 0][e], bx, by, w.chord[TIME ? 5 : 0][e], rad1, meta[4 * j + 0],
                                 meta[4 * j + 1], meta[4 * j + 2], tab.path + 3 * (size_t)tab.poff[j],
                                 tab.poff[j + 1] - tab.poff[j]))
              atomicOr(&w.mask[e], 1ull << b);
          } else if (!seg_outside_circle(ax, ay, bx, by, rad1, meta, j)) {
            const int kind = (int)meta[4 * j + 3];
            if (kind == 1) atomicOr(&w.mask[e], 1ull << b);
            else if (kind == 3) {
              vb0 = off[j];
              P = off[j + 1] - vb0;
              if (P < 2) P = 0;
              elx = fmin(ax, bx); ehx = fmax(ax, bx); ely = fmin(ay, by); ehy = fmax(ay, by);
              slack = gap_min + 1e-9 * (fabs(elx) + fabs(ehx) + fabs(ely) + fabs(ehy));
              // (fmin / fmax drop a NaN operand: a chord with a non-finite coordinate keeps every segment)
              if (!(ax - ax == 0.0 && ay - ay == 0.0 && bx - bx == 0.0 && by - by == 0.0)) slack = __builtin_inf();
              pax = ax; pay = ay; pbx = bx; pby = by;
              evert = fabs(bx - ax) < .000001;
              if (!evert) em = (by - ay) / (bx - ax);
            }
          }
        }
        int pmax = P;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pmax = max(pmax, __shfl_xor(pmax, o));
        // (at least one round per group of pairs, so that the last group always reaches the final drain below)
        const int n_sg = pmax > 0 ? pmax : 1;
        // (a side starts where the one before it ends: that vertex, whether it is finite and its difference in the
        // first side test are carried from round to round)
        double Ax = 0.0, Ay = 0.0, diff_a1 = 0.0;
        bool fin_a = false;
        if (P > 0) {
          Ax = vxy[2 * (vb0 + P - 1)]; Ay = vxy[2 * (vb0 + P - 1) + 1];
          fin_a = (Ax - Ax == 0.0) && (Ay - Ay == 0.0);
          diff_a1 = (em * (Ax - pax) + pay) - Ay;
        }
        for (int sg = 0; sg < n_sg; ++sg) {
          bool push = false;
          if (sg < P) {
            const int vb = vb0 + sg;
            const double Bx = vxy[2 * vb], By = vxy[2 * vb + 1];
            const bool apart = (fmin(Ax, Bx) - ehx > slack) || (elx - fmax(Ax, Bx) > slack) ||
                               (fmin(Ay, By) - ehy > slack) || (ely - fmax(Ay, By) > slack);
            const bool fin_b = (Bx - Bx == 0.0) && (By - By == 0.0);
            const bool finite = fin_a && fin_b;
            const double diff_b1 = (em * (Bx - pax) + pay) - By;
            // a segment is dropped only if one of segmentDistSqrd's side tests, computed as the reference computes them
            // (R/DRRT.jl:1150-1188), also separates it from the chord: without a separating side test the reference
            // answers 0.0 however far apart the two are (segments on one common line), see kernels_collide.hip
            bool one_side;
            if (evert) one_side = (Ax >= pax && Bx >= pax) || (Ax <= pax && Bx <= pax);
            else one_side = diff_a1 * diff_b1 > 0.0;           // (a product that rounds to zero: the side is kept)
            if (fabs(Bx - Ax) < .000001) one_side = one_side || (pax >= Ax && pbx >= Ax) || (pax <= Ax && pbx <= Ax);
            else {
              const double qm = tab.vslope[vb];
              const double diff_a = (qm * (pax - Ax) + Ay) - pay;
              const double diff_b = (qm * (pbx - Ax) + Ay) - pby;
              one_side = one_side || diff_a * diff_b > 0.0;
            }
            push = !(apart && one_side) || !finite;              // (slack = +inf or NaN: never apart)
            Ax = Bx; Ay = By; fin_a = fin_b; diff_a1 = diff_b1;
          }
          const unsigned long long sv = __ballot(push);
          if (push)
            sq[nsq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sv, 0u))] =
                (unsigned)e | ((unsigned)b << 6) | ((unsigned)sg << 12);
          nsq += __popcll(sv);
          // one site for the segment test: rounds of 64 while the queue could overflow, the rest after the last pair
          const bool last = (p0 + 64 >= pairs) && (sg + 1 == n_sg);
          while (nsq > kSqCap - 64 || (last && nsq > 0)) seg_round();
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    RRTX_DUB_T(t_arc);
    RRTX_DUB_ADD(2, t_1b, t_arc);
    const unsigned long long mask = w.mask[lane];
    // ---- arc screen (lane = edge): every stored row of an arc lies on its circle (centre c, radius r_min),
    // so a piece of the arc -- and the short piece that joins it to the next one, whose far end is the
    // tangent point on the same circle -- stays inside that disk.  A static polygon whose every edge is
    // farther than r_min + robotRadius (+ slack) from c, or a ball whose centre is farther than that plus
    // its radius, cannot be hit by any of them: most stage-1 survivors are near misses of the inflated
    // chord, and this spares their ~60 pieces per arc the full test.  Moving obstacles are kept.
    {
      unsigned long long pm[3] = {mask, mask, mask};
      if (mask != 0ull) {
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) {
          const Piece &pc = st.pc[pi];
          if (pc.kind == 1 || pc.len <= 0) continue;
          const double reach = (r_min + robot_radius) * (1.0 + 1e-9) + 1e-9 * (1.0 + fabs(pc.cx) + fabs(pc.cy));
          unsigned long long keep = 0ull, mm = mask;
          while (mm != 0ull) {
            const int b = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            const int j = j0 + b;
            const int kind = (int)meta[4 * j + 3];
            bool touch = true;
            if (kind == 3) {
              const int vb = off[j], ve = off[j + 1];
              double d2min = __builtin_inf();
              if (ve - vb >= 2) {
                double Ax = vxy[2 * (ve - 1)], Ay = vxy[2 * (ve - 1) + 1];
                for (int v = vb; v < ve; ++v) {
                  const double Bx = vxy[2 * v], By = vxy[2 * v + 1];
                  d2min = jl_min(d2min, dspts(pc.cx, pc.cy, Ax, Ay, Bx, By));
                  Ax = Bx; Ay = By;
                }
              }
              touch = !(d2min > reach * reach);
            } else if (kind == 1) {
              const double dx = pc.cx - meta[4 * j + 0], dy = pc.cy - meta[4 * j + 1];
              const double R = reach + fabs(meta[4 * j + 2]) * (1.0 + 1e-9);
              touch = !(dx * dx + dy * dy > R * R);
            }
            if (touch) keep |= 1ull << b;
          }
          pm[pi] = keep;
        }
      }
      w.pm[0][lane] = pm[0]; w.pm[1][lane] = pm[1]; w.pm[2][lane] = pm[2];
    }
    // Which stored pieces are walked at all: piece k of the polyline (row k-1 -> row k) belongs to the arc its
    // first row lies on, else to the arc its second row starts (the two short joins), else it is the straight
    // middle piece; the rows of one arc are contiguous, so an edge walks up to three row ranges, and only
    // those whose arc (or line) can touch one of the marked obstacles.
    int segs = 0;
    {
      const int la = st.pc[0].len, lb = st.pc[1].len;
      const bool line1 = st.pc[1].kind == 1;
      const int last = rows - 1;
      int c0 = min(la, last); c0 = c0 > 0 ? c0 : 0;                                   // rows 1 .. la
      int c1 = line1 ? ((last >= la + 1) ? 1 : 0) : (min(la + lb, last) - la);        // row la+1 | rows la+1 .. la+lb
      c1 = c1 > 0 ? c1 : 0;
      int c2 = last - (line1 ? la + lb : la + lb + 1) + 1;                            // the rest
      c2 = c2 > 0 ? c2 : 0;
      const bool l1 = line1 ? (mask != 0ull) : (w.pm[1][lane] != 0ull);
      if (w.pm[0][lane] == 0ull) c0 = 0;
      if (!l1) c1 = 0;
      if (w.pm[2][lane] == 0ull) c2 = 0;
      w.lc[0][lane] = (unsigned char)c0;
      w.lc[1][lane] = (unsigned char)c1;
      if (mask != 0ull && rows > 1) segs = c0 + c1 + c2;
    }
    // The answer is an OR over the pieces, so the ORDER in which an edge's pieces are walked is free -- and most candidate
    // edges near an obstacle do collide somewhere (60 % at C3, 88 % at C5), after which the edge leaves the walk.  The walk
    // therefore starts where a hit is most likely: at the piece whose share of the chord lies nearest the first marked
    // obstacle's centre (a guess: any value gives the same result), and wraps around.
    {
      int rot = 0;
      if (segs > 8 && segs < 65536) {
        const int j = j0 + (__ffsll((long long)mask) - 1);
        const double ux = gx - sx, uy = gy - sy;
        double t = ((meta[4 * j + 0] - sx) * ux + (meta[4 * j + 1] - sy) * uy) / (ux * ux + uy * uy);
        if (!(t > 0.0)) t = 0.0;               // (NaN too)
        if (t > 1.0) t = 1.0;
        rot = (int)(t * (double)segs) - 3;
        rot = rot < 0 ? 0 : (rot >= segs ? segs - 1 : rot);
      }
      w.rot[lane] = (unsigned short)rot;
      w.nseg[lane] = (unsigned short)(segs < 65536 ? segs : 0);
    }
    // ---- stage 2 (lane = one polyline piece of one edge), a window of rows of every live edge at a
    // time: an edge that has collided leaves the numbering at the next window, so its remaining pieces
    // stop taking up lanes (most candidate edges do collide somewhere along the polyline).  The window is
    // 8 rows, more when few edges are left so that a round still fills the wave.  Step a) gives every
    // piece the obstacles whose bounding circle it reaches; those (piece, obstacle) pairs are QUEUED and the
    // full polygon test -- by far the longest part -- runs on 64 queued pairs at a time, whatever round
    // they came from, so its lanes are full too. ----
    constexpr int kPqCap = (TIME ? 6 : 4) * 128;
    if (lane == 0) w.pqn = 0;
    __builtin_amdgcn_wave_barrier();
    RRTX_DUB_T(t_2);
    RRTX_DUB_ADD(3, t_arc, t_2);
#ifdef RRTX_TILE_CLOCKS
    unsigned long long acc_b = 0ull;
#endif
    // b) lane = one queued (piece, obstacle) pair that got past the bounding circle: the full test
    auto test_pair = [&](unsigned int ent) {
      const int ee = (int)(ent & 63u), row = (int)((ent >> 6) & 255u), j = j0 + (int)(ent >> 14);
      if (w.done[ee]) return;
      double px, py, x, y, pt = 0.0, t = 0.0;
      if constexpr (TIME) {
        polyline_point_t(w.pc[ee], w.ti[ee], row - 1, r_min, px, py, pt);
        polyline_point_t(w.pc[ee], w.ti[ee], row, r_min, x, y, t);
      } else {
        polyline_point(w.pc[ee], row - 1, r_min, px, py);
        polyline_point(w.pc[ee], row, r_min, x, y);
      }
      bool h2;
      if (TIME && meta[4 * j + 3] >= 6.0)
        h2 = edge_hits_moving(px, py, pt, x, y, t, robot_radius, meta[4 * j + 0], meta[4 * j + 1], meta[4 * j + 2],
                              tab.path + 3 * (size_t)tab.poff[j], tab.poff[j + 1] - tab.poff[j]);
      else
        h2 = seg_hits_polygon_past_circle(px, py, x, y, robot_radius, meta, off, vxy, j);
      if (h2) w.done[ee] = 1;
    };
    // (one copy of the round body and ONE site of the drain loop: the polygon test is long, and every
    //  inlined copy of it costs registers)
    bool finishing = false;
    int base = 0;
    for (;;) {
      int total = 0, win = 8;
      if (!finishing) {
        const bool live = segs > base && !w.done[lane];
        const unsigned long long lm = __ballot(live);
        if (lm == 0ull) {
          finishing = true;
        } else {
          const int n_live = __popcll(lm);
          if (n_live * win < 64) win = (64 + n_live - 1) / n_live;
          incl = live ? min(win, segs - base) : 0;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
          }
          w.pstart[lane + 1] = incl;
          if (lane == 0) w.pstart[0] = 0;
          __builtin_amdgcn_wave_barrier();
          total = __shfl(incl, 63);
        }
      }
      const int n_rounds = finishing ? 1 : (total + 63) / 64;
      for (int rd = 0; rd < n_rounds; ++rd) {
        // a) lane = piece: its end points, and which of the edge's marked obstacles it can touch at all
        //    (the polygon test's own first step, :1536-1539: outside the bounding circle => no hit)
        const int i = rd * 64 + lane;
        unsigned long long pending = 0ull;
        int e = 0, row = 0;
        if (!finishing && i < total) {
          int lo = 0, hi = 64;                     // edge e with pstart[e] <= i < pstart[e + 1]
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (w.pstart[mid] <= i) lo = mid; else hi = mid;
          }
          e = lo;
          if (!w.done[e]) {
            // live piece number -> row and the piece it belongs to
            int tt = base + (i - w.pstart[e]) + (int)w.rot[e];
            if (tt >= (int)w.nseg[e] && w.nseg[e] != 0) tt -= (int)w.nseg[e];
            const int la = w.pc[e][0].len, lb = w.pc[e][1].len;
            const bool line1 = w.pc[e][1].kind == 1;
            const int c0 = w.lc[0][e], c1 = w.lc[1][e];
            int own;
            if (tt < c0) { row = 1 + tt; own = 0; }
            else if (tt - c0 < c1) { row = la + 1 + (tt - c0); own = 1; }
            else { row = (line1 ? la + lb : la + lb + 1) + (tt - c0 - c1); own = 2; }
            double px, py, x, y;
            polyline_point(w.pc[e], row - 1, r_min, px, py);
            polyline_point(w.pc[e], row, r_min, x, y);
            if constexpr (TIME) {
              if (row == w.ti[e].P - 1) { x = w.ti[e].gx; y = w.ti[e].gy; }    // the last row is the end node itself
            }
            // the obstacles this piece can touch: those of the arc it belongs to (the straight piece: all marked)
            unsigned long long mm = (own == 1 && line1) ? w.mask[e] : w.pm[own][e];
            while (mm != 0ull) {
              const int b = __ffsll((long long)mm) - 1;
              mm &= mm - 1ull;
              const bool moving = TIME && meta[4 * (j0 + b) + 3] >= 6.0;   // no bounding-circle step for those (:1579)
              bool keep;
              if (moving) {
                // ... but the same box argument as in stage 1a: the obstacle's centre stays in the box of its path, the
                // robot's on this piece; farther apart than robotRadius + radius in x or in y: no hit at any instant
                const double4 pb = reinterpret_cast<const double4 *>(tab.pbox)[j0 + b];
                const double R = fabs(robot_radius + meta[4 * (j0 + b) + 2]) * (1.0 + 1e-9) +
                                 1e-9 * (1.0 + fabs(pb.x) + fabs(pb.y) + fabs(pb.z) + fabs(pb.w) + fabs(px) + fabs(py));
                keep = !((fmin(px, x) - pb.y > R) || (pb.x - fmax(px, x) > R) || (fmin(py, y) - pb.w > R) || (pb.z - fmax(py, y) > R));
                if (!(px - px == 0.0 && py - py == 0.0 && x - x == 0.0 && y - y == 0.0)) keep = true;   // (fmin / fmax drop a NaN)
              } else keep = !seg_outside_circle(px, py, x, y, robot_radius, meta, j0 + b);
              if (keep) pending |= 1ull << b;
            }
          }
        }
        // queue the pairs (whatever does not fit waits for the drain below), then drain
        for (;;) {
          const int cnt = __popcll(pending);
          int pin = cnt;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(pin, o);
            if (lane >= o) pin += v;
          }
          const int n_new = __shfl(pin, 63);
          const int qb = w.pqn;
          int at = qb + pin - cnt;
          unsigned long long nb = pending;
          while (nb != 0ull) {
            const int b = __ffsll((long long)nb) - 1;
            nb &= nb - 1ull;
            if (at < kPqCap) {
              w.pq[at] = (unsigned)e | ((unsigned)row << 6) | ((unsigned)b << 14);
              pending &= ~(1ull << b);
            }
            ++at;
          }
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) w.pqn = (qb + n_new < kPqCap) ? qb + n_new : kPqCap;
          __builtin_amdgcn_wave_barrier();
          const bool more = __ballot(pending != 0ull) != 0ull;
          // between windows what is known about collisions decides who walks on: drain down to 32 there
          const int at_least = (more || finishing) ? 1 : ((rd == n_rounds - 1) ? 32 : 64);
          for (;;) {
            const int n = w.pqn;                       // wave-uniform
            if (n < at_least || n == 0) break;
            const int take = n < 64 ? n : 64;
            RRTX_DUB_T(t_b0);
            if (lane < take) test_pair(w.pq[n - take + lane]);
            __builtin_amdgcn_wave_barrier();
            RRTX_DUB_T(t_b1);
            RRTX_DUB_ACC(acc_b, t_b0, t_b1);
            if (lane == 0) w.pqn = n - take;
            __builtin_amdgcn_wave_barrier();
          }
          if (!more) break;
        }
      }
      if (finishing) break;
      base += win;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef RRTX_TILE_CLOCKS
    {
      const unsigned long long t_end = wall_clock64();
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_dub_clk[5], acc_b);
        atomicAdd(&g_dub_clk[4], (t_end - t_2) - acc_b);
      }
    }
#endif
  }
  return valid && w.done[lane] != 0;
}

// ---- steering and checking are two launches -------------------------------------------------------------------
// Round 3: calculateTrajectory's six words (16 atan2, 6 sincos, 2 acos -- ~4 000 instructions once the shared
// deterministic transcendentals are inlined) and the two-stage check no longer share a kernel: the steering kernel
// leaves a 128-byte record per directed edge (the three pieces, the cost, lengths / kinds / word) and the check
// kernel starts at stage 1a with the record in hand.  The check kernel's registers are then the check's alone, and
// the steering kernel runs at full occupancy with every lane busy.  Edges go through in chunks of kDubChunk, so the
// records of a chunk (256 MB) are written and read back while they are still in the Infinity Cache.
constexpr long long kDubChunk = 2ll << 20;

// where the edges of a launch come from
struct EdgeSrc {
  int mode;                        // 0: rows s[i], g[i];  1: candidate edges of extend() (CSR entry e = (sample, node));
                                   // 2: mirrored edges (start node, end node) through a list of edge ids
  int dir;                         // mode 1: 0 = sample -> node, 1 = node -> sample
  const double *s, *g;             // mode 0 (rows of 4)
  const double *q;                 // mode 1: samples (rows of 4)
  const int64_t *offsets;          //         offsets[nq] = number of CSR entries
  const int32_t *idx, *owner;      //         node and sample of every entry
  int nq;
  const int32_t *ids;              // mode 2: edge ids (null: ids are first, first + 1, ...)
  const int32_t *es, *ee;          //         start / end node of every mirrored edge
  long long first;
  const double *nx, *ny, *nz, *nw; // node table (modes 1, 2)
  int n_nodes;
  long long cap;                   // mode 1: the caller's capacity (an overflowing CSR is left alone)
};

// start and goal pose of edge i; false: no such edge (past the end, or indices that point nowhere)
__device__ __forceinline__ bool load_edge(const EdgeSrc &src, long long i, long long n, double *s, double *g) {
  if (src.mode == 0) {
    if (i >= n) return false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = src.s[4 * i + k]; g[k] = src.g[4 * i + k]; }
    return true;
  }
  if (src.mode == 1) {
    const long long total = src.offsets[src.nq];
    if (total > src.cap || i >= total) return false;   // capacity overflow: the CSR arrays are only partly written
    const int qi = src.owner[i], nd = src.idx[i];
    if ((unsigned)qi >= (unsigned)src.nq || (unsigned)nd >= (unsigned)src.n_nodes) return false;   // defensive
    const double n0 = src.nx[nd], n1 = src.ny[nd], n2 = src.nz[nd], n3 = src.nw[nd];
    const double q0 = src.q[4 * (size_t)qi], q1 = src.q[4 * (size_t)qi + 1], q2 = src.q[4 * (size_t)qi + 2],
                 q3 = src.q[4 * (size_t)qi + 3];
    const bool rev = src.dir != 0;
    s[0] = rev ? n0 : q0; s[1] = rev ? n1 : q1; s[2] = rev ? n2 : q2; s[3] = rev ? n3 : q3;
    g[0] = rev ? q0 : n0; g[1] = rev ? q1 : n1; g[2] = rev ? q2 : n2; g[3] = rev ? q3 : n3;
    return true;
  }
  if (i >= n) return false;
  const long long id = src.ids ? (long long)src.ids[i] : src.first + i;
  const int a = src.es[id], b = src.ee[id];
  if ((unsigned)a >= (unsigned)src.n_nodes || (unsigned)b >= (unsigned)src.n_nodes) return false;
  s[0] = src.nx[a]; s[1] = src.ny[a]; s[2] = src.nz[a]; s[3] = src.nw[a];
  g[0] = src.nx[b]; g[1] = src.ny[b]; g[2] = src.nz[b]; g[3] = src.nw[b];
  return true;
}

// the record: doubles 0-11 the three pieces (cx, cy, a, b), 12 the cost (bestDist), 13 the packed integers
// (len0 | len1 << 16 | len2 << 32 | kinds << 48 (2 bits each) | word << 56); 14, 15 unused (128-byte records)
constexpr int kRecDoubles = 16;
__device__ __forceinline__ void store_rec(double *__restrict__ rec, const Steer &st) {
  double2 *r2 = reinterpret_cast<double2 *>(rec);
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    r2[2 * pi] = make_double2(st.pc[pi].cx, st.pc[pi].cy);
    r2[2 * pi + 1] = make_double2(st.pc[pi].a, st.pc[pi].b);
  }
  unsigned long long m = 0ull;
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    const int len = st.pc[pi].len < 0 ? 0 : (st.pc[pi].len > 65535 ? 65535 : st.pc[pi].len);
    m |= (unsigned long long)len << (16 * pi);
    m |= (unsigned long long)(st.pc[pi].kind & 3) << (48 + 2 * pi);
  }
  m |= (unsigned long long)(st.word & 7) << 56;
  r2[6] = make_double2(st.cost, __longlong_as_double((long long)m));
}
__device__ __forceinline__ void load_rec(const double *__restrict__ rec, Steer &st) {
  const double2 *r2 = reinterpret_cast<const double2 *>(rec);
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    const double2 c = r2[2 * pi], ab = r2[2 * pi + 1];
    st.pc[pi].cx = c.x; st.pc[pi].cy = c.y; st.pc[pi].a = ab.x; st.pc[pi].b = ab.y;
  }
  const double2 cm = r2[6];
  st.cost = cm.x;
  const unsigned long long m = (unsigned long long)__double_as_longlong(cm.y);
#pragma unroll
  for (int pi = 0; pi < 3; ++pi) {
    st.pc[pi].len = (int)((m >> (16 * pi)) & 0xffffull);
    st.pc[pi].kind = (int)((m >> (48 + 2 * pi)) & 3ull);
  }
  st.word = (int)((m >> 56) & 7ull);
}

// calculateTrajectory of the edges [base, base + count) of the source: the record of edge base + k at rec[k],
// cost = edge.dist (sqrt(Wdist^2 + dt^2) in a space with time), the word, the number of polyline rows.
// rec2 / cost2 / word2 (candidate edges of extend() only): the REVERSE edge g -> s in the same thread, its record at
// rec2[k] -- the two poses' circle centres and angles (4 sincos + 4 atan2 of the ~30 transcendentals of a steer) are
// computed once for both directions.
__global__ __launch_bounds__(256) void dubins_steer_rec_kernel(const EdgeSrc src, long long base, long long count, long long n,
                                                               double r_min, int has_time, double *__restrict__ rec,
                                                               double *__restrict__ cost, uint8_t *__restrict__ word,
                                                               int32_t *__restrict__ traj_len, double *__restrict__ rec2,
                                                               double *__restrict__ cost2, uint8_t *__restrict__ word2) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const long long i = base + k;
  double s[4] = {0, 0, 0, 0}, g[4] = {1, 0, 0, 0};
  const bool valid = load_edge(src, i, n, s, g);
  if (__ballot(valid) == 0ull) return;          // (mode 1: the grid covers the caller's capacity)
  PoseCircles pa = pose_circles(s[0], s[1], s[3], r_min), pb = pose_circles(g[0], g[1], g[3], r_min);
  double a0 = s[0], a1 = s[1], a2 = s[2], b0 = g[0], b1 = g[1], b2 = g[2];
  // ONE copy of the six-word evaluation in the code, run once or twice (the second time with the poses swapped)
#pragma unroll 1
  for (int d = 0; d < (rec2 ? 2 : 1); ++d) {
    const double sa[4] = {a0, a1, a2, 0.0}, ga[4] = {b0, b1, b2, 0.0};
    Steer st;
    dubins_steer_poses<true>(sa, ga, pa, pb, r_min, st);
    store_rec((d ? rec2 : rec) + kRecDoubles * k, st);
    if (valid) {
      double *co = d ? cost2 : cost;
      uint8_t *wo = d ? word2 : word;
      if (co) co[i] = has_time ? dist_with_time(st.cost, a2, b2) : st.cost;
      if (wo) write_word(wo, i, st.word);
      if (!d && traj_len) traj_len[i] = st.pc[0].len + st.pc[1].len + st.pc[2].len;
    }
    const PoseCircles t = pa; pa = pb; pb = t;
    double x;
    x = a0; a0 = b0; b0 = x;
    x = a1; a1 = b1; b1 = x;
    x = a2; a2 = b2; b2 = x;
  }
}

// explicitEdgeCheck(S, ::DubinsEdge, ob) over the polygon list (:750-774) for the steered edges of a chunk:
// stage 1 = straight chord with radius robotRadius + 2*minTurningRadius, stage 2 = every stored polyline piece
// with robotRadius.  TIME: the pieces carry time; for the candidate edges of extend() (mode 1) bit 1 of the hit
// byte says !validMove (findBestParent blocks an edge on explicitEdgeCheck || !validMove, R/DRRT_Q.jl:1960).
// spread: a wave takes every n_waves-th edge of the chunk instead of 64 neighbours -- edges that arrive grouped
// by their sample cost what the sample's surroundings make them cost, and all waves should get the same mix.
template <bool TIME>
__global__ __launch_bounds__(256, TIME ? 2 : 3) void dubins_check_rec_kernel(
    const EdgeSrc src, long long base, long long count, long long n, int spread, double r_min, double robot_radius,
    double vmin, double vmax, const PolyTab tab, const double *__restrict__ rec, uint8_t *__restrict__ hit) {
  __shared__ WaveDubinsT<TIME> wd[4];
  WaveDubinsT<TIME> &w = wd[threadIdx.x >> 6];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long k = t;
  if (spread) {
    const long long n_waves = (count + 63) / 64;
    const long long gw = t >> 6;
    k = (gw < n_waves) ? (long long)(threadIdx.x & 63) * n_waves + gw : count;
  }
  double s[4] = {0, 0, 0, 0}, g[4] = {1, 0, 0, 0};
  const bool valid = k < count && load_edge(src, base + k, n, s, g);
  if (__ballot(valid) == 0ull) return;
  Steer st;
  load_rec(rec + kRecDoubles * (valid ? k : 0), st);
  if (!valid) { st.cost = __builtin_inf(); st.word = 6; st.pc[0].len = st.pc[1].len = st.pc[2].len = 0; }
  RRTX_DUB_ADD(6, 0ull, 1ull);
  int bad_move = 0;
  if (TIME && src.mode == 1 && valid && !valid_move_time(s[2], g[2], st.cost / (s[2] - g[2]), vmin, vmax)) bad_move = 2;
  const bool h = wave_dubins_collides<TIME>(w, valid, st, s, g, r_min, robot_radius, tab);
  if (valid) hit[base + k] = (uint8_t)((h ? 1 : 0) | bad_move);
}

// edge.trajectory (R/DRRT_DubinsEdge_functions.jl:684-701): the polyline of every edge, written at
// traj_off[i] (row units), rows of (x, y) -- with has_time rows of (x, y, t); traj == null only
// reports P per edge.
__global__ __launch_bounds__(256) void dubins_trajectory_kernel(const double *__restrict__ s,
                                                                const double *__restrict__ g, long long ne,
                                                                double r_min, int has_time,
                                                                const int64_t *__restrict__ traj_off,
                                                                double *__restrict__ traj, long long cap_rows,
                                                                int32_t *__restrict__ traj_len) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  Steer st;
  dubins_steer<true>(s + 4 * i, g + 4 * i, r_min, st);
  const int P = st.pc[0].len + st.pc[1].len + st.pc[2].len;
  if (traj_len) traj_len[i] = P;
  if (!traj) return;
  long long row = traj_off[i];
  if (has_time) {
    const TimeInfo ti = time_info(st, s + 4 * i, g + 4 * i, r_min);
    for (int k = 0; k < P; ++k, ++row) {
      double x, y, t;
      polyline_point_t(st.pc, ti, k, r_min, x, y, t);
      if (row < cap_rows) { traj[3 * row] = x; traj[3 * row + 1] = y; traj[3 * row + 2] = t; }
    }
    return;
  }
  for (int pi = 0; pi < 3; ++pi)
    for (int k = 0; k < st.pc[pi].len; ++k, ++row) {
      double x, y;
      piece_point(st.pc[pi], k, r_min, x, y);
      if (row < cap_rows) { traj[2 * row] = x; traj[2 * row + 1] = y; }
    }
}

PolyTab poly_tab(rrtx_ctx *ctx) {
  PolyTab t;
  t.meta = ctx->d_poly_meta.as<double>();
  t.off = ctx->d_poly_off.as<int32_t>();
  t.vxy = ctx->d_poly_vxy.as<double>();
  t.vslope = ctx->d_poly_slope.as<double>();
  t.poff = ctx->d_poly_path_off.as<int32_t>();
  t.path = ctx->d_poly_path.as<double>();
  t.pbox = ctx->d_poly_pbox.as<double>();
  t.m = ctx->poly_n_active;
  return t;
}

// obstacles that move in time are tested at the pieces' time stamps: only a space with time has them
int check_space(rrtx_ctx *ctx) {
  if (ctx->poly_has_moving && !ctx->opt_space_has_time)
    return fail(ctx, RRTX_E_STATE, "Dubins edges against moving obstacles (kind 6/7) need the time-parameterised "
                                   "trajectory: set RRTX_OPT_SPACE_HAS_TIME (CSpace.spaceHasTime)");
  return RRTX_OK;
}

}  // namespace

#ifdef RRTX_TILE_CLOCKS
extern "C" int rrtx_debug_dubins_clocks(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dub_clk), sizeof(unsigned long long) * 8);
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dub_clk), z, sizeof(z));
  }
  return rc;
}
#endif

// steer + check of the edges [0, n) of a source, chunk by chunk
// cost2 / word2 / hit2 (candidate edges, src.mode == 1 with dir == 0): the reverse edges too, steered in the same launch
static int run_dubins_edges(rrtx_ctx *ctx, const EdgeSrc &src, long long n, int spread, double r_min, double robot_radius,
                            const PolyTab &tab, bool check, double *cost, uint8_t *word, uint8_t *hit, int32_t *traj_len,
                            double *cost2 = nullptr, uint8_t *word2 = nullptr, uint8_t *hit2 = nullptr) {
  const bool both = hit2 != nullptr;
  const long long chunk = n < kDubChunk ? n : kDubChunk;
  RRTX_HIP(ctx, ctx->ws_dub_rec.ensure(sizeof(double) * kRecDoubles * (size_t)chunk * (both ? 2 : 1)));
  double *rec = ctx->ws_dub_rec.as<double>();
  double *rec2 = both ? rec + kRecDoubles * (size_t)chunk : nullptr;
  const int has_time = ctx->opt_space_has_time ? 1 : 0;
  EdgeSrc rsrc = src;
  rsrc.dir = 1;
  for (long long base = 0; base < n; base += chunk) {
    const long long count = (n - base < chunk) ? n - base : chunk;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    span_begin(ctx, KF_DUBINS_STEER);
    hipLaunchKernelGGL(dubins_steer_rec_kernel, grid, block, 0, ctx->stream, src, base, count, n, r_min, has_time, rec, cost,
                       word, traj_len, rec2, cost2, word2);
    span_end(ctx);
    if (!check) continue;
    for (int d = 0; d < (both ? 2 : 1); ++d) {
      const EdgeSrc &cs = d ? rsrc : src;
      const double *rc_ = d ? rec2 : rec;
      uint8_t *h = d ? hit2 : hit;
      span_begin(ctx, KF_DUBINS);
      if (has_time)
        hipLaunchKernelGGL(dubins_check_rec_kernel<true>, grid, block, 0, ctx->stream, cs, base, count, n, spread, r_min,
                           robot_radius, ctx->dubins_vmin, ctx->dubins_vmax, tab, rc_, h);
      else
        hipLaunchKernelGGL(dubins_check_rec_kernel<false>, grid, block, 0, ctx->stream, cs, base, count, n, spread, r_min,
                           robot_radius, ctx->dubins_vmin, ctx->dubins_vmax, tab, rc_, h);
      span_end(ctx);
    }
  }
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// Candidate Dubins edges of extend(): CSR entry e = (sample qi, node idx[e]); both directed edges
// sample->near and near->sample are steered and checked (R/DRRT_Q.jl:1951-1963, 2600-2602 with
// Edge = DubinsEdge).
int launch_candidate_dubins(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                            const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap, double r_min,
                            double robot_radius, double *cost_out, double *cost_in, uint8_t *word_out,
                            uint8_t *word_in, uint8_t *hit_out, uint8_t *hit_in) {
  if (nq <= 0 || cap <= 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  if ((rc = check_space(ctx))) return rc;
  const PolyTab tab = poly_tab(ctx);
  EdgeSrc src = {};
  src.mode = 1;
  src.q = q_dev; src.offsets = offsets_dev; src.idx = idx_dev; src.owner = owner_dev; src.nq = nq;
  src.nx = ctx->nodes[0]; src.ny = ctx->nodes[1]; src.nz = ctx->nodes[2]; src.nw = ctx->nodes[3];
  src.n_nodes = (int)ctx->n_nodes;
  src.cap = (long long)cap;
  src.dir = 0;
  return run_dubins_edges(ctx, src, (long long)cap, 0, r_min, robot_radius, tab, true, cost_out, word_out, hit_out, nullptr,
                          cost_in, word_in, hit_in);
}

int launch_detmath_eval(rrtx_ctx *ctx, int op, const double *x_dev, const double *y_dev, int64_t n, double *out_dev) {
  if (n <= 0) return RRTX_OK;
  hipLaunchKernelGGL(detmath_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, x_dev, y_dev,
                     (long long)n, out_dev);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_dubins_trajectory(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double r_min,
                             const int64_t *traj_off_dev, double *traj_dev, int64_t cap_rows,
                             int32_t *traj_len_dev) {
  if (ne <= 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  span_begin(ctx, KF_DUBINS);
  hipLaunchKernelGGL(dubins_trajectory_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream,
                     s_dev, g_dev, (long long)ne, r_min, ctx->opt_space_has_time ? 1 : 0, traj_off_dev, traj_dev,
                     (long long)cap_rows, traj_len_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_dubins_steer(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double r_min,
                        double *cost_dev, uint8_t *word_dev, double *wdist_dev, double *velocity_dev,
                        uint8_t *valid_dev) {
  if (ne <= 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  span_begin(ctx, KF_DUBINS);
  hipLaunchKernelGGL(dubins_steer_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, s_dev,
                     g_dev, (long long)ne, r_min, ctx->opt_space_has_time ? 1 : 0, ctx->dubins_vmin, ctx->dubins_vmax,
                     cost_dev, wdist_dev, velocity_dev, word_dev, valid_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// the packed (active-only) obstacles [pb, pe) as a table of their own: obstacle-relative pointers shift, the vertex /
// slope / path arrays keep their global offsets
static PolyTab tab_range(const PolyTab &t, int pb, int pe) {
  PolyTab r = t;
  if (pb < 0) pb = 0;
  if (pe > t.m) pe = t.m;
  if (pe < pb) pe = pb;
  r.meta = t.meta + 4 * (size_t)pb;
  r.off = t.off + pb;
  r.poff = t.poff + pb;
  r.pbox = t.pbox + 4 * (size_t)pb;
  r.m = pe - pb;
  return r;
}

// explicitEdgeCheck(S, edge::DubinsEdge, ob) for MIRRORED edges ids_dev[k] (start node -> end node) against the packed
// obstacles [pb, pe): what the obstacle sweeps of the Dubins space run (R/DRRT.jl:3157, 3236-3249)
int launch_dubins_edges_idx(rrtx_ctx *ctx, const int32_t *ids_dev, int64_t n, double r_min, double robot_radius, int pb,
                            int pe, uint8_t *hit_dev) {
  if (n <= 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  if ((rc = check_space(ctx))) return rc;
  const PolyTab tab = tab_range(poly_tab(ctx), pb, pe);
  if (tab.m <= 0) { RRTX_HIP(ctx, hipMemsetAsync(hit_dev, 0, (size_t)n, ctx->stream)); return RRTX_OK; }
  EdgeSrc src = {};
  src.mode = 2;
  src.ids = ids_dev; src.es = ctx->ge_start; src.ee = ctx->ge_end;
  src.nx = ctx->nodes[0]; src.ny = ctx->nodes[1]; src.nz = ctx->nodes[2]; src.nw = ctx->nodes[3];
  src.n_nodes = (int)ctx->n_nodes;
  RRTX_HIP(ctx, hipMemsetAsync(hit_dev, 0, (size_t)n, ctx->stream));     // (an id that points nowhere is not written)
  return run_dubins_edges(ctx, src, (long long)n, 1, r_min, robot_radius, tab, true, nullptr, nullptr, hit_dev, nullptr);
}

int launch_dubins_edges_check(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double r_min,
                              double robot_radius, double *cost_dev, uint8_t *word_dev, uint8_t *hit_dev,
                              int32_t *traj_len_dev, int pb, int pe) {
  if (ne <= 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  if ((rc = check_space(ctx))) return rc;
  const PolyTab tab = (pb >= 0) ? tab_range(poly_tab(ctx), pb, pe) : poly_tab(ctx);
  EdgeSrc src = {};
  src.mode = 0;
  src.s = s_dev; src.g = g_dev;
  return run_dubins_edges(ctx, src, (long long)ne, 1, r_min, robot_radius, tab, true, cost_dev, word_dev, hit_dev, traj_len_dev);
}

}  // namespace rrtx
