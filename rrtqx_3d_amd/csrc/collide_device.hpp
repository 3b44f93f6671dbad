// collide_device.hpp -- the exact sphere edge test as a device function, shared by
// kernels_collide.hip (edge kernels) and kernels_sweep.hip (obstacle sweeps).
#pragma once
#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {
namespace {

// distancePointToSegment + explicitEdgeCheck3D for one (edge, sphere) pair.
// Returns true on collision.  t = dot/edgeLen (NOT edgeLen^2) is the
// reference's formula (R/DRRT_Q.jl:1208) and is reproduced on purpose.
__device__ __forceinline__ bool edge_hits_sphere(double p0x, double p0y, double p0z, double bx, double by,
                                                 double bz, double edge_len, const SphRec &ob) {
  double a0 = ob.cx - p0x, a1 = ob.cy - p0y, a2 = ob.cz - p0z;
  double dot = (a0 * bx + a1 * by) + a2 * bz;
  double t = jl_clamp01(dot / edge_len);
  double qx = p0x + t * bx, qy = p0y + t * by, qz = p0z + t * bz;
  double s = sq3(ob.cx, ob.cy, ob.cz, qx, qy, qz);
  // distS > robotRadius + radius  <=>  s >= thr ; collision unless that holds
  return !(s >= ob.thr);
}

// ---- polygons moving in time (kinds 6 and 7) ------------------------------------------------
// findIndexBeforeTime, R/DRRT_Q.jl:1351-1362: 1-based row count with path[i, 3] < t
__device__ __forceinline__ int index_before_time(const double *__restrict__ path, int rows, double t) {
  if (rows < 1) return -1;
  int i = 0;
  while (i + 1 <= rows && path[3 * i + 2] < t) i += 1;
  return i;
}

// findTransformObsToTimeOfPoint, R/DRRT_Q.jl:1367-1391
__device__ __forceinline__ void transform_obs_to_time(const double *__restrict__ path, int rows, double t,
                                                      double &dx, double &dy) {
  const int before = index_before_time(path, rows, t);
  if (before < 1) { dx = path[0]; dy = path[1]; return; }
  if (before == rows) { dx = path[3 * (before - 1)]; dy = path[3 * (before - 1) + 1]; return; }
  const double *b = path + 3 * (before - 1), *a = path + 3 * before;
  const double along = (t - b[2]) / (a[2] - b[2]);
  dx = b[0] + along * (a[0] - b[0]);
  dy = b[1] + along * (a[1] - b[1]);
}

// explicitEdgeCheck2D, kinds 6 and 7 (R/DRRT_Q.jl:1699-1771 = R/DRRT.jl:1579-1651): robot edge in
// (x, y, time) against the obstacle's bounding circle carried along its path; every path segment that
// overlaps the edge in time is tested at the time of closest approach of the two centres.
__device__ bool edge_hits_moving(double sx, double sy, double st, double ex, double ey, double et,
                                 double robot_radius, double cx, double cy, double rad,
                                 const double *__restrict__ path, int rows) {
  double x_1, y_1, T_1, lx, ly, lt;
  if (st < et) { x_1 = sx; y_1 = sy; T_1 = st; lx = ex; ly = ey; lt = et; }
  else { lx = sx; ly = sy; lt = st; x_1 = ex; y_1 = ey; T_1 = et; }
  int first = index_before_time(path, rows, T_1);
  if (first < 1) first = 1;
  int last = 1 + index_before_time(path, rows, lt);
  if (last > rows) last = rows;
  if (last <= first) return false;
  const double m_x1 = (lx - x_1) / (lt - T_1);
  const double m_y1 = (ly - y_1) / (lt - T_1);
  const double rr = rad + robot_radius;
  for (int is = first; is <= last - 1; ++is) {
    const double *pa = path + 3 * (is - 1), *pb = path + 3 * is;
    const double x_2 = pa[0] + cx, y_2 = pa[1] + cy, T_2 = pa[2];
    const double m_x2 = ((pb[0] + cx) - x_2) / (pb[2] - T_2);
    const double m_y2 = ((pb[1] + cy) - y_2) / (pb[2] - T_2);
    const double num = (((m_x1 * m_x1) * T_1 + m_x2 * (((m_x2 * T_2) + x_1) - x_2)) -
                        m_x1 * (((m_x2 * (T_1 + T_2)) + x_1) - x_2)) +
                       (m_y1 - m_y2) * ((((m_y1 * T_1) - (m_y2 * T_2)) - y_1) + y_2);
    const double den = (m_x1 - m_x2) * (m_x1 - m_x2) + (m_y1 - m_y2) * (m_y1 - m_y2);
    double T_c = num / den;
    if (T_c < jl_max(T_1, T_2)) T_c = jl_max(T_1, T_2);
    else if (T_c > jl_min(lt, pb[2])) T_c = jl_min(lt, pb[2]);
    const double r_x = m_x1 * (T_c - T_1) + x_1, r_y = m_y1 * (T_c - T_1) + y_1;
    const double o_x = m_x2 * (T_c - T_2) + x_2, o_y = m_y2 * (T_c - T_2) + y_2;
    if ((r_x - o_x) * (r_x - o_x) + (r_y - o_y) * (r_y - o_y) < rr * rr) return true;
  }
  return false;
}

// Per-sample sphere lists of the fused extend path (sample_spheres_kernel / nn_finish_kernel):
// everything the sample pass needs about one active sphere, one 64-byte record:
//   thr_in  : quickCheck, inside the sphere            <=> !(s >= thr_in)   (R/DRRT_Q.jl:1410)
//   thr_pt  : (sqrt(s) - robotRadius) - radius < 0     <=>  s < thr_pt      (thr_point_clear)
//   reach   : inflated robotRadius + radius (or +inf)
struct alignas(64) SampleSph { double cx, cy, cz, thr_in, thr_pt, reach, pad0, pad1; };
constexpr int kSphListCap = 8;   // spheres listed per sample; a longer list sends the sample's edges to the full loop

// what the fused extend() path reads about the sphere list (tables of sync_spheres)
struct ExtendDev {
  const SphRec *sph;           // exact records, packed (active only), list order
  const SampleSph *stab;
  const float *reach_f;        // fp32 reach table, pair-interleaved, origin-relative
  const double4 *naos;         // node coordinates by node index, one 32-byte record each
  double ox, oy, oz;
  int m;                       // active spheres
  int pad;
  double r_bound;              // radius of the ball the lists are built for; < 0: no lists (full loop)
  uint8_t *sample_unsafe;
};

// fp32 screen state of a point or segment midpoint for the reach table: centre relative to the
// origin and the inflated half length  h~ = RU[(half + 3 eps |m|_1)(1 + 8 eps)]  (+inf disables it)
struct ReachProbe { float mx, my, mz, h; };
__device__ __forceinline__ ReachProbe reach_probe(const ExtendDev &x, double cx, double cy, double cz, double half,
                                                  bool usable) {
  const double mx = cx - x.ox, my = cy - x.oy, mz = cz - x.oz;
  const double eps = 5.9604644775390625e-08;
  const double h = (half * (1.0 + 1e-12) + 3.0 * eps * (fabs(mx) + fabs(my) + fabs(mz)) + 1e-30) * (1.0 + 8.0 * eps);
  ReachProbe p;
  p.h = usable ? __double2float_ru(h) : __builtin_inff();
  p.mx = usable ? (float)mx : 0.f; p.my = usable ? (float)my : 0.f; p.mz = usable ? (float)mz : 0.f;
  return p;
}

// Both directed edges sample <-> node against the sphere list: explicitEdgeCheck(S, edge) for
// newNode -> near and near -> newNode (R/DRRT_Q.jl:1951-1963, 2600-2602, 1802-1826).  `list` holds the
// nl spheres within reach of any edge inside the sample's ball (nl > kSphListCap: not all fit); edges
// the list does not cover (zero length: t = 0/0 = NaN collides with every active sphere, longer than the
// ball, non-finite) walk the whole table behind the packed fp32 midpoint screen.  The screens only
// ever skip pairs that cannot collide; every decision is edge_hits_sphere.  All lanes of the wave call
// this together.  len = sqrt(sq3(sample, node)).
__device__ __forceinline__ void edge_flags(const ExtendDev &x, bool act, double sx, double sy, double sz, double tx,
                                           double ty, double tz, double len, int nl_in, const int *list, bool &out_hit,
                                           bool &in_hit, const SphRec *list_recs = nullptr) {
  // list_recs: the listed spheres' records where the caller keeps them at hand (LDS), same order as `list`
  // out: sample -> near ; in: near -> sample.  edgeLen is the same value either way
  const double bx = tx - sx, by = ty - sy, bz = tz - sz;
  const double cx = sx - tx, cy = sy - ty, cz = sz - tz;
  out_hit = false; in_hit = false;
  if (__ballot(act) == 0ull) return;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const double cmax = fmax(fmax(fmax(fabs(sx), fabs(sy)), fmax(fabs(sz), fabs(tx))), fmax(fabs(ty), fabs(tz)));
  const bool usable = (len > 0.0) && (len < 1e30) && (cmax < 1e30) && (sx == sx) && (sy == sy) && (sz == sz) &&
                      (tx == tx) && (ty == ty) && (tz == tz);
  bool need_full = act;
  if (x.r_bound >= 0.0) {
    int nl = 0;
    if (act) {
      nl = nl_in;
      need_full = !usable || nl > kSphListCap || !(len <= x.r_bound);
      if (need_full) nl = 0;
    }
    for (int c = 0; __ballot(c < nl) != 0ull; ++c) {
      if (c < nl && !(out_hit && in_hit)) {
        const SphRec ob = list_recs ? list_recs[c] : x.sph[list[c]];
        if (!out_hit) out_hit = edge_hits_sphere(sx, sy, sz, bx, by, bz, len, ob);
        if (!in_hit) in_hit = edge_hits_sphere(tx, ty, tz, cx, cy, cz, len, ob);
      }
    }
    if (__ballot(need_full) == 0ull) return;
  }
  const ReachProbe rp = reach_probe(x, 0.5 * (sx + tx), 0.5 * (sy + ty), 0.5 * (sz + tz), 0.5 * len, usable);
  const f32x2 m2x = {rp.mx, rp.mx}, m2y = {rp.my, rp.my}, m2z = {rp.mz, rp.mz}, h2 = {rp.h, rp.h};
  constexpr int G = 8;
  const int m = x.m;
  for (int j0 = 0; j0 < m; j0 += G) {
    const float *gp = x.reach_f + (size_t)(j0 / 8) * 32;      // wave-uniform
    unsigned touch = 0u;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const f32x2 ccx = {gp[8 * pr + 0], gp[8 * pr + 1]}, ccy = {gp[8 * pr + 2], gp[8 * pr + 3]};
      const f32x2 ccz = {gp[8 * pr + 4], gp[8 * pr + 5]}, rr = {gp[8 * pr + 6], gp[8 * pr + 7]};
      const f32x2 dx = ccx - m2x, dy = ccy - m2y, dz = ccz - m2z;
      f32x2 dm2 = dx * dx;
      dm2 = __builtin_elementwise_fma(dy, dy, dm2);
      dm2 = __builtin_elementwise_fma(dz, dz, dm2);
      const f32x2 bound = rr + h2;
      const f32x2 b2 = bound * bound;
      touch |= (!(dm2.x > b2.x) ? 1u : 0u) << (2 * pr);
      touch |= (!(dm2.y > b2.y) ? 1u : 0u) << (2 * pr + 1);
    }
    if (j0 + G > m) touch &= (1u << (m - j0)) - 1u;
    if (!need_full) touch = 0u;
    if (__ballot(touch != 0u) == 0ull) continue;
    for (int g = 0; g < G; ++g) {
      if (__ballot((touch >> g) & 1u) == 0ull) continue;
      if (((touch >> g) & 1u) && !(out_hit && in_hit)) {
        const SphRec ob = x.sph[j0 + g];
        if (!out_hit) out_hit = edge_hits_sphere(sx, sy, sz, bx, by, bz, len, ob);
        if (!in_hit) in_hit = edge_hits_sphere(tx, ty, tz, cx, cy, cz, len, ob);
      }
    }
    if (__ballot(need_full && !(out_hit && in_hit)) == 0ull) break;
  }
}

// Sample pass, exact part for one (sample, sphere) the fp32 screen could not rule out:
// explicitPointCheck's per-sphere terms (R/DRRT_Q.jl:1402-1415, 1463-1487 as thresholds on the squared
// distance) and membership in the sample's sphere list.  Spheres the screen rules out are farther
// than reach + ball radius from the sample: neither inside, nor closer than the robot radius, nor
// touchable by an edge in the ball.  Returns true if the sphere makes the sample unsafe; *listed says
// whether it belongs on the list.
__device__ __forceinline__ bool sample_exact(const ExtendDev &x, int j, double px, double py, double pz, double base_b,
                                             bool *listed) {
  const SampleSph sp = x.stab[j];
  const double s = sq3(sp.cx, sp.cy, sp.cz, px, py, pz);
  const double B = base_b + sp.reach;
  *listed = x.r_bound >= 0.0 && !(s > B * B * (1.0 + 1e-12));
  return !(s >= sp.thr_in) | (s < sp.thr_pt);
}

}  // namespace
}  // namespace rrtx
