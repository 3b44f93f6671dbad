// kernels_collide.hip -- per-edge and per-point collision checks against the
// obstacle lists.  Replaces explicitEdgeCheck / explicitPointCheck and their
// geometry helpers (sphere: R/DRRT_Q.jl:1205-1210, 1402-1595, 1775-1826;
// polygon: R/DRRT.jl:1009-1106, 1144-1202, 1258-1470, 1523-1578, 1660-1678).
// gfx950 only.  Lane = edge (or point); the obstacle loop is wave-uniform so
// obstacle records arrive through scalar loads; a wave leaves the loop as soon
// as every lane has its answer (ballot early-out).
#include "collide_device.hpp"

#include <limits>

namespace rrtx {

namespace {

// ------------------------------------------------------------ spheres -------
// (edge_hits_sphere: collide_device.hpp, shared with kernels_sweep.hip)

// ---- conservative reach test (never decides a result) ---------------------------
// A sphere can only collide with an edge if its centre is within
//   |c - mid| <= L/2 + (robotRadius + radius) (+ rounding slack)
// of the edge midpoint: the reference's foot point q = p0 + t*(p1-p0), t in [0,1],
// lies on the segment whatever t is (the dot/edgeLen quirk moves it along the
// segment, not off it).  Pairs failing this bound are skipped; every other pair
// gets the exact unfused evaluation above.  Degenerate edges (L == 0 -> t = NaN
// -> "hit", R/DRRT_Q.jl:1208) and non-finite/huge coordinates disable the test.
struct EdgeReach {
  double mx, my, mz;   // midpoint
  double hls;          // (L/2 + slack) inflated, or +inf when the test is disabled
};
__device__ __forceinline__ EdgeReach edge_reach(double ax, double ay, double az, double ex, double ey,
                                                double ez, double edge_len) {
  EdgeReach r;
  r.mx = 0.5 * (ax + ex); r.my = 0.5 * (ay + ey); r.mz = 0.5 * (az + ez);
  double cmax = fmax(fmax(fmax(fabs(ax), fabs(ay)), fmax(fabs(az), fabs(ex))), fmax(fabs(ey), fabs(ez)));
  bool usable = (edge_len > 0.0) && (edge_len < 1e100) && (cmax < 1e100) &&
                (ax == ax) && (ay == ay) && (az == az) && (ex == ex) && (ey == ey) && (ez == ez);
  r.hls = usable ? (0.5 * edge_len + 1e-12 * (cmax + 1.0)) * (1.0 + 1e-12) : __builtin_inf();
  return r;
}
// true = the pair must be evaluated exactly (NaN-safe: NaN compares pass)
__device__ __forceinline__ bool may_touch(const EdgeReach &r, const SphRec &b) {
  double dx = b.cx - r.mx, dy = b.cy - r.my, dz = b.cz - r.mz;
  double dm2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  double bound = r.hls + b.thr;   // b.thr holds the inflated reach of the sphere here
  return !(dm2 > bound * bound);
}

// Edges are given either as positions (p0/p1, `stride` doubles per point) or, when sidx != NULL,
// as node index pairs into the node SoA (the planner's graph edges in obstacle sweeps).
// maskp (optional, one byte per packed obstacle) restricts the test to obstacles with mask != 0.
__global__ __launch_bounds__(256) void edges_spheres_kernel(const double *__restrict__ p0,
                                                            const double *__restrict__ p1, int stride,
                                                            const int32_t *__restrict__ sidx,
                                                            const int32_t *__restrict__ eidx,
                                                            const double *__restrict__ nx,
                                                            const double *__restrict__ ny,
                                                            const double *__restrict__ nz,
                                                            long long ne, const SphRec *__restrict__ sph,
                                                            const SphRec *__restrict__ reach,
                                                            const uint8_t *__restrict__ maskp,
                                                            const int32_t *__restrict__ orig, int m_begin,
                                                            int m_end, uint8_t *__restrict__ hit,
                                                            int32_t *__restrict__ first_hit) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = i < ne;
  double ax = 0, ay = 0, az = 0, ex = 0, ey = 0, ez = 0;
  if (act) {
    if (sidx) {
      const int a = sidx[i], b = eidx[i];
      ax = nx[a]; ay = ny[a]; az = nz[a];
      ex = nx[b]; ey = ny[b]; ez = nz[b];
    } else {
      ax = p0[i * stride + 0]; ay = p0[i * stride + 1]; az = p0[i * stride + 2];
      ex = p1[i * stride + 0]; ey = p1[i * stride + 1]; ez = p1[i * stride + 2];
    }
  }
  const double bx = ex - ax, by = ey - ay, bz = ez - az;
  const double edge_len = sqrt_rn(sq3(ax, ay, az, ex, ey, ez));
  const EdgeReach er = edge_reach(ax, ay, az, ex, ey, ez, edge_len);
  bool done = !act;
  int first = -1;
  if (__ballot(act) == 0ull) return;
  // groups of 4 obstacles: the four wave-uniform reach records are requested
  // together (one SMEM round trip per group instead of per obstacle)
  for (int j0 = m_begin; j0 < m_end; j0 += 4) {
    SphRec rb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rb[g] = reach[min(j0 + g, m_end - 1)];
    bool cand[4];
    bool anyc = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jj = min(j0 + g, m_end - 1);
      const bool allowed = (j0 + g < m_end) && (maskp == nullptr || maskp[jj] != 0);
      cand[g] = !done && allowed && may_touch(er, rb[g]);
      anyc = anyc || cand[g];
    }
    if (__ballot(anyc) == 0ull) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (__ballot(cand[g]) == 0ull) continue;
      if (cand[g] && !done) {
        const SphRec ob = sph[j0 + g];
        if (edge_hits_sphere(ax, ay, az, bx, by, bz, edge_len, ob)) { done = true; first = orig[j0 + g]; }
      }
    }
    if (__ballot(!done) == 0ull) break;   // every edge of this wave already collides
  }
  if (act) {
    hit[i] = first >= 0 ? 1 : 0;
    if (first_hit) first_hit[i] = first;
  }
}

// Candidate edges of extend(): CSR entry e = (query qi, node idx[e]); checks
// sample->near and near->sample (R/DRRT_Q.jl:1951-1963, 2600-2602).  Both
// directions share the segment, so one reach test serves both.
__global__ __launch_bounds__(256) void candidate_edges_kernel(
    const double *__restrict__ q, int stride, const int64_t *__restrict__ offsets, int nq,
    const int32_t *__restrict__ idx, const int32_t *__restrict__ owner, const double4 *__restrict__ naos,
    int n_nodes, long long cap,
    const SphRec *__restrict__ sph, const float *__restrict__ reach_f, double ox, double oy, double oz, int m,
    const int32_t *__restrict__ lists, const int32_t *__restrict__ list_n, int list_cap, double r_bound,
    uint8_t *__restrict__ hit_out, uint8_t *__restrict__ hit_in) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = offsets[nq];
  // capacity overflow: the CSR arrays are only partly written (the caller gets RRTX_E_CAPACITY and
  // retries); nothing may be dereferenced through them
  if (total > cap) return;
  bool act = e < total;
  double sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0;
  int lo = 0;
  if (act) {
    lo = owner[e];             // owning query (written by nn_order_kernel)
    const int n = idx[e];
    act = (unsigned)lo < (unsigned)nq && (unsigned)n < (unsigned)n_nodes;   // defensive: never index out of range
    if (act) {
      sx = q[(size_t)lo * stride + 0]; sy = q[(size_t)lo * stride + 1]; sz = q[(size_t)lo * stride + 2];
      const double4 nd = naos[n];          // one 32-byte record instead of three scattered reads
      tx = nd.x; ty = nd.y; tz = nd.z;
    }
  }
  // out: sample -> near ; in: near -> sample.  edgeLen is the same value either
  // way (squares of negated differences); the direction vector flips.
  const double bx = tx - sx, by = ty - sy, bz = tz - sz;
  const double cx = sx - tx, cy = sy - ty, cz = sz - tz;
  const double len = sqrt_rn(sq3(sx, sy, sz, tx, ty, tz));
  bool out_hit = false, in_hit = false;
  if (__ballot(act) == 0ull) return;   // the grid covers the caller's capacity, most waves are past the end
  // a segment the screens below may reason about: positive finite length, finite coordinates.
  // (A zero-length edge collides with every active sphere: t = 0/0 = NaN, R/DRRT_Q.jl:1208.)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const double cmax = fmax(fmax(fmax(fabs(sx), fabs(sy)), fmax(fabs(sz), fabs(tx))), fmax(fabs(ty), fabs(tz)));
  const bool usable = (len > 0.0) && (len < 1e30) && (cmax < 1e30) && (sx == sx) && (sy == sy) && (sz == sz) &&
                      (tx == tx) && (ty == ty) && (tz == tz);
  // ---- fast path: only the spheres on the sample's list (sample_spheres_kernel) ----
  bool need_full = act;
  if (lists) {
    int nl = 0;
    if (act) {
      nl = list_n[lo];
      // the list only covers edges inside the ball it was built for
      need_full = !usable || nl > list_cap || !(len <= r_bound);
      if (need_full) nl = 0;
    }
    for (int c = 0; __ballot(c < nl) != 0ull; ++c) {
      if (c < nl && !(out_hit && in_hit)) {
        const SphRec ob = sph[lists[(size_t)lo * list_cap + c]];
        if (!out_hit) out_hit = edge_hits_sphere(sx, sy, sz, bx, by, bz, len, ob);
        if (!in_hit) in_hit = edge_hits_sphere(tx, ty, tz, cx, cy, cz, len, ob);
      }
    }
    if (__ballot(need_full) == 0ull) {
      if (act) { hit_out[e] = out_hit ? 1 : 0; hit_in[e] = in_hit ? 1 : 0; }
      return;
    }
  }
  // ---- full obstacle loop for the lanes that need it ----
  // fp32 screen state of this lane's segment: midpoint relative to the context origin and the
  // inflated half length  hls~ = RU[(L/2 + 3 eps |m|_1)(1 + 8 eps)]  (+inf disables the screen)
  float mxf, myf, mzf, hlsf;
  {
    const double mx = 0.5 * (sx + tx) - ox, my = 0.5 * (sy + ty) - oy, mz = 0.5 * (sz + tz) - oz;
    mxf = (float)mx; myf = (float)my; mzf = (float)mz;
    const double eps = 5.9604644775390625e-08;
    const double h = (0.5 * len * (1.0 + 1e-12) + 3.0 * eps * (fabs(mx) + fabs(my) + fabs(mz)) + 1e-30) * (1.0 + 8.0 * eps);
    hlsf = usable ? __double2float_ru(h) : __builtin_inff();
    if (!usable) { mxf = 0.f; myf = 0.f; mzf = 0.f; }
  }
  const f32x2 m2x = {mxf, mxf}, m2y = {myf, myf}, m2z = {mzf, mzf}, h2 = {hlsf, hlsf};
  // groups of 8 obstacles: two s_load_dwordx16 bring four pair-interleaved records; each pair costs
  // 8 packed fp32 instructions (3 sub, mul + 2 fma, add, mul) + 2 compares, ONE wave-level branch per group
  constexpr int G = 8;
  for (int j0 = 0; j0 < m; j0 += G) {
    const float *gp = reach_f + (size_t)(j0 / 8) * 32;
    unsigned touch = 0u;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const f32x2 cx = {gp[8 * pr + 0], gp[8 * pr + 1]}, cy = {gp[8 * pr + 2], gp[8 * pr + 3]};
      const f32x2 cz = {gp[8 * pr + 4], gp[8 * pr + 5]}, rr = {gp[8 * pr + 6], gp[8 * pr + 7]};
      const f32x2 dx = cx - m2x, dy = cy - m2y, dz = cz - m2z;
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
    // rare: some lane is within reach of one of these spheres -> exact evaluation for that lane
    for (int g = 0; g < G; ++g) {
      if (__ballot((touch >> g) & 1u) == 0ull) continue;
      if (((touch >> g) & 1u) && !(out_hit && in_hit)) {
        const SphRec ob = sph[j0 + g];
        if (!out_hit) out_hit = edge_hits_sphere(sx, sy, sz, bx, by, bz, len, ob);
        if (!in_hit) in_hit = edge_hits_sphere(tx, ty, tz, cx, cy, cz, len, ob);
      }
    }
    if (__ballot(need_full && !(out_hit && in_hit)) == 0ull) break;
  }
  if (act) {
    hit_out[e] = out_hit ? 1 : 0;
    hit_in[e] = in_hit ? 1 : 0;
  }
}

// explicitPointCheck over spheres (R/DRRT_Q.jl:1520-1590).  aux holds, per
// active sphere, radius and thr_in = first s with sqrt(s) > radius.
// 16 lanes share one point and stride over the obstacle list; the reference's
// sequential loop is order-independent except for which of two equal clearances
// (+-0.0) survives, so the lane combine prefers the lower list position.
constexpr int kPtLanes = 16;
__global__ __launch_bounds__(256) void points_spheres_kernel(const double *__restrict__ p, int stride,
                                                             long long np, const SphRec *__restrict__ sph,
                                                             const double *__restrict__ radius,
                                                             const double *__restrict__ thr_in, int m,
                                                             double robot_radius, int quick,
                                                             uint8_t *__restrict__ unsafe,
                                                             double *__restrict__ clearance) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = gid / kPtLanes;
  const int sub = (int)(gid % kPtLanes);
  const bool act = i < np;
  double px = 0, py = 0, pz = 0;
  if (act) { px = p[i * stride + 0]; py = p[i * stride + 1]; pz = p[i * stride + 2]; }
  bool bad = false;
  double best = __builtin_inf();
  int best_j = 0x7fffffff;
  for (int j = sub; j < m; j += kPtLanes) {
    const double s = sq3(sph[j].cx, sph[j].cy, sph[j].cz, px, py, pz);
    // quickCheck: inside the sphere  (Wdist > radius => outside, :1410)
    if (quick && !(s >= thr_in[j])) bad = true;
    // explicitPointCheck2D: thisDist = Wdist - robotRadius; thisDist - radius < 0 => collision
    const double td = (sqrt_rn(s) - robot_radius) - radius[j];
    if (td < 0.0) bad = true;
    if (td < best) { best = td; best_j = j; }
  }
#pragma unroll
  for (int off = kPtLanes / 2; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off);
    const int oj = __shfl_xor(best_j, off);
    const bool obad = __shfl_xor((int)bad, off) != 0;
    bad = bad || obad;
    if ((ob < best) || (ob == best && oj < best_j)) { best = ob; best_j = oj; }
  }
  if (act && sub == 0) {
    unsafe[i] = bad ? 1 : 0;
    if (clearance) clearance[i] = bad ? 0.0 : best;
  }
}

// Samples of extend(): explicitPointCheck (as points_spheres_kernel, quick = 1) plus, per
// sample, the short list of spheres that any candidate edge of that sample can possibly touch.
// Every point the edge test looks at lies on the segment between the sample and a neighbour
// within the search radius r (the reference's foot point p0 + t (p1 - p0), t in [0, 1],
// R/DRRT_Q.jl:1208), hence within r of the sample; a sphere whose centre is farther than
// r + robotRadius + radius (+ slack) from the sample cannot collide with any of those edges.
// A list longer than kSphListCap is marked as overflowed (count = cap + 1) and the edges of
// that sample take the full obstacle loop.  Never decides a result by itself.
// (SampleSph, kSphListCap: collide_device.hpp)

// lane = sample; the kSampleWaves waves of a workgroup share the same 64 samples and split the
// sphere table between them, so every sphere record is a wave-uniform (scalar) load; the partial
// answers meet in LDS.
constexpr int kSampleWaves = 16;
__global__ __launch_bounds__(64 * kSampleWaves) void sample_spheres_kernel(
    const double *__restrict__ p, int stride, long long np, const SampleSph *__restrict__ tab, int m,
    double r_bound, uint8_t *__restrict__ unsafe, int32_t *__restrict__ lists, int32_t *__restrict__ list_n) {
  __shared__ int s_bad[64];
  __shared__ int s_n[64];
  __shared__ int s_list[64][kSphListCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long i = (long long)blockIdx.x * 64 + lane;
  const bool act = i < np;
  if (wave == 0) { s_bad[lane] = 0; s_n[lane] = 0; }
  __syncthreads();
  double px = 0, py = 0, pz = 0;
  if (act) { px = p[i * stride + 0]; py = p[i * stride + 1]; pz = p[i * stride + 2]; }
  const double pmax = fmax(fmax(fabs(px), fabs(py)), fabs(pz));
  // slack for the rounding of the foot point and of this distance; NaN / inf sample: everything is a candidate
  const double base_b = r_bound + 1e-12 * (pmax + 1.0);
  const int per = (m + kSampleWaves - 1) / kSampleWaves;
  const int j0 = wave * per, j1 = min(j0 + per, m);
  bool bad = false;
  constexpr int kU = 8;                  // sphere records fetched together (wave-uniform loads in flight)
  for (int jb = j0; jb < j1; jb += kU) {
    SampleSph sp[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) sp[u] = tab[min(jb + u, j1 - 1)];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int j = jb + u;
      if (j >= j1) break;
      const double s = sq3(sp[u].cx, sp[u].cy, sp[u].cz, px, py, pz);
      bad = bad | !(s >= sp[u].thr_in) | (s < sp[u].thr_pt);
      const double B = base_b + sp[u].reach;
      if (act && !(s > B * B * (1.0 + 1e-12))) {
        const int at = atomicAdd(&s_n[lane], 1);
        if (at < kSphListCap) s_list[lane][at] = j;
      }
    }
  }
  if (bad) atomicOr(&s_bad[lane], 1);
  __syncthreads();
  if (wave == 0 && act) {
    if (unsafe) unsafe[i] = s_bad[lane] ? 1 : 0;
    const int n = s_n[lane];
    list_n[i] = n;
    for (int c = 0; c < min(n, kSphListCap); ++c) lists[i * kSphListCap + c] = s_list[lane][c];
  }
}

// ------------------------------------------------------------ polygons ------
// distanceSqrdPointToSegment, R/DRRT.jl:1060-1083
__device__ __forceinline__ double dist_sqrd_point_to_segment(double px, double py, double ax, double ay,
                                                             double bx, double by) {
  double vx = px - ax, vy = py - ay;
  double ux = bx - ax, uy = by - ay;
  double det = vx * ux + vy * uy;
  if (det <= 0) {
    return vx * vx + vy * vy;
  } else {
    double len = ux * ux + uy * uy;
    if (det >= len) {
      double ex = bx - px, ey = by - py;
      return ex * ex + ey * ey;
    } else {
      double cr = ux * vy - uy * vx;
      return (cr * cr) / len;
    }
  }
}

// segmentDistSqrd, R/DRRT.jl:1144-1202
__device__ double segment_dist_sqrd(double pax, double pay, double pbx, double pby, double qax, double qay,
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
  double r = dist_sqrd_point_to_segment(pax, pay, qax, qay, qbx, qby);
  r = jl_min(r, dist_sqrd_point_to_segment(pbx, pby, qax, qay, qbx, qby));
  r = jl_min(r, dist_sqrd_point_to_segment(qax, qay, pax, pay, pbx, pby));
  r = jl_min(r, dist_sqrd_point_to_segment(qbx, qby, pax, pay, pbx, pby));
  return r;
}

// segmentDistSqrd for the SAME edge taken in both directions -- rF = segmentDistSqrd(pa, pb, qa, qb), rR =
// segmentDistSqrd(pb, pa, qa, qb) -- as extend() needs them (newNode -> near and near -> newNode are both checked,
// R/DRRT_Q.jl:1951-1963, 2600-2602).  Each result is the reference's own sequence of operations on its own argument
// order; what the two sequences have in common is computed once: the "close to vertical" tests (|x - y| = |y - x|),
// the second line's side test (its two differences swap roles and the test is symmetric in them), the distances of
// the edge's end points to the other segment (the same two calls in the other order under a commutative min), and
// len = |u|^2 inside distanceSqrdPointToSegment.  Direction-dependent and therefore computed twice: the edge's slope
// ((pby - pay) / (pbx - pax) vs (pay - pby) / (pax - pbx): equal except for the sign of a zero), the first side test's
// differences (anchored at pa resp. pb) and the two distances TO the edge (their det / cross are taken from the other
// end).  One call replaces two at ~1.3x the cost of one.
__device__ void segment_dist_sqrd_both(double pax, double pay, double pbx, double pby, double qax, double qay, double qbx,
                                       double qby, double &rF, double &rR) {
  bool sepF, sepR;
  if (fabs(pbx - pax) < .000001) {
    sepF = (qax >= pax && qbx >= pax) || (qax <= pax && qbx <= pax);
    sepR = (qax >= pbx && qbx >= pbx) || (qax <= pbx && qbx <= pbx);
  } else {
    const double mF = (pby - pay) / (pbx - pax);
    const double mR = (pay - pby) / (pax - pbx);
    const double dAF = (mF * (qax - pax) + pay) - qay, dBF = (mF * (qbx - pax) + pay) - qby;
    const double dAR = (mR * (qax - pbx) + pby) - qay, dBR = (mR * (qbx - pbx) + pby) - qby;
    sepF = (dAF > 0.0 && dBF > 0.0) || (dAF < 0.0 && dBF < 0.0);
    sepR = (dAR > 0.0 && dBR > 0.0) || (dAR < 0.0 && dBR < 0.0);
  }
  bool sepQ = false;       // (the reference evaluates it only while `possible` holds; its value does not depend on that)
  if (!(sepF && sepR)) {
    if (fabs(qbx - qax) < .000001) {
      sepQ = (pax >= qax && pbx >= qax) || (pax <= qax && pbx <= qax);
    } else {
      const double m = (qby - qay) / (qbx - qax);
      const double diffA = (m * (pax - qax) + qay) - pay;
      const double diffB = (m * (pbx - qax) + qay) - pby;
      sepQ = (diffA > 0.0 && diffB > 0.0) || (diffA < 0.0 && diffB < 0.0);
    }
  }
  const bool possF = !sepF && !sepQ, possR = !sepR && !sepQ;
  rF = 0.0; rR = 0.0;
  if (possF && possR) return;
  const double d1 = dist_sqrd_point_to_segment(pax, pay, qax, qay, qbx, qby);
  const double d2 = dist_sqrd_point_to_segment(pbx, pby, qax, qay, qbx, qby);
  if (!possF) {
    double r = jl_min(d1, d2);
    r = jl_min(r, dist_sqrd_point_to_segment(qax, qay, pax, pay, pbx, pby));
    rF = jl_min(r, dist_sqrd_point_to_segment(qbx, qby, pax, pay, pbx, pby));
  }
  if (!possR) {
    double r = jl_min(d2, d1);
    r = jl_min(r, dist_sqrd_point_to_segment(qax, qay, pbx, pby, pax, pay));
    rR = jl_min(r, dist_sqrd_point_to_segment(qbx, qby, pbx, pby, pax, pay));
  }
}

// (explicitEdgeCheck2D for kinds 1 and 3, R/DRRT.jl:1523-1578: its bounding-circle test and its segment tests are
// stages A and B of edges_polygons_kernel below)

// (index_before_time, transform_obs_to_time, edge_hits_moving: collide_device.hpp, shared with kernels_dubins.hip)

constexpr int kPolyPairs = 256;        // (edge, obstacle) pairs a wave queues for the bounding-circle test
constexpr int kPolyQueue = 384;        // (edge, polygon segment) tests a wave queues (320 in the build without a time column)
constexpr int kPolyWaveCand = 256;     // candidate obstacles a wave lists (more: the whole list is walked)
constexpr int kPolyWaveSamples = 16;   // ... for at most this many samples per wave
constexpr int kPolyListCap = 64;       // obstacles the sample pass lists per sample (more: the wave lists for itself)
constexpr int kPolyBitWords = 16;      // ... merged per wave through a bit set of this many 64-bit words
// per-wave scratch of edges_polygons_kernel: the wave's 64 edges, the boxes of the current group of 32 obstacles,
// each edge's first hit (list position), the queues of the two test stages, the wave's candidate obstacles
template <bool MOVING, int QUEUE>
struct PolyWaveT {
  double e[MOVING ? 6 : 4][64];   // ax, ay, bx, by per lane; with obstacles that move in time also at, bt (startPoint[3])
  double em[64];       // the edge's slope (by - ay) / (bx - ax) as segmentDistSqrd divides it (R/DRRT.jl:1158)
  double emr[64];      // PAIRED: the slope as the reverse edge divides it, (ay - by) / (ax - bx)
  float4 box[32];      // current group: bounding boxes (xlo, xhi, ylo, yhi) rounded outward to fp32
  int first[128];      // first hit (list position) per edge; PAIRED: [2 lane] forward, [2 lane + 1] reverse
  int jidx[32];        // current group: list position of each of its obstacles
  double4 gmeta[32];   // current group: (cx, cy, radius, kind) of each of its obstacles
  int goff[32], gcnt[32];   // current group: first vertex and number of vertices
  unsigned short pairq[kPolyPairs];   // (edge lane | obstacle slot << 6) of the pairs the box test leaves
  unsigned pq[QUEUE];              // (edge lane | obstacle slot << 6 | segment << 11) of the segment tests to run
  short wc[kPolyWaveCand];   // CSR mode: the obstacles any edge of the wave can reach (list positions, ascending)
  unsigned long long bits[kPolyBitWords];   // CSR mode: union of the lists of the wave's samples
  unsigned head[64];         // stage A: (pair lane | chunk number << 6) where a pair's run of polygon sides starts in the chunk
};

// Second edge source of edges_polygons_kernel: the candidate edges of extend() straight from the CSR
// neighbour lists -- thread e holds sample -> neighbour of entry e and the reverse edge
// (R/DRRT_Q.jl:1951-1963, 2600-2602).  q == nullptr selects the p0 / p1 arrays.
struct PolyCsr {
  const double *q;
  const int64_t *offsets;
  const int32_t *idx;
  const int32_t *owner;
  const double4 *nodes_aos;
  uint8_t *hit_in;
  long long cap;
  int nq, n_nodes;
};

#ifdef RRTX_TILE_CLOCKS
// (one row per wave, plain stores: atomics on eight shared words slowed the measured kernels tenfold)
constexpr int kClkRows = 65536;
__device__ unsigned long long g_pe_clk[kClkRows * 8];
__device__ unsigned long long g_pp_clk[kClkRows * 8];
#define RRTX_PE_T(var) const unsigned long long var = clock64()
#define RRTX_PE_ACC(acc, a, b) acc += (b) - (a)
#else
#define RRTX_PE_T(var) do { } while (0)
#define RRTX_PE_ACC(acc, a, b) do { } while (0)
#endif

// PAIRED (the candidate edges of extend(), CSR mode): a lane holds BOTH directions of one CSR entry -- sample -> neighbour
// and the reverse edge (R/DRRT_Q.jl:1951-1963, 2600-2602).  The lane lists obstacles and hands out pairs once for the two;
// stage A runs the bounding-circle test for both directions in the pair's lane (it differs between them only in
// rounding: distanceSqrdPointToSegment measures from the other end), and stage B's segment_dist_sqrd_both gives both
// directions' answers at 1.3x the cost of one.  Every boolean is still the reference's own expression for that directed
// edge.  (Until late in round 3 the two directions sat in neighbouring lanes and the odd one idled through the listing, the
// boxes and the hand-out: 64 entries per wave instead of 32 halve those per-wave costs and fill the rounds of both stages.)
template <bool PAIRED, bool MOVING, int QUEUE>
__device__ __forceinline__ void edges_polygons_wave(long long i, PolyWaveT<MOVING, QUEUE> &w,
                                                    const double *__restrict__ p0,
                                                    const double *__restrict__ p1, int stride, long long ne,
                                                    const PolyCsr &csr, const double *__restrict__ meta,
                                                    const int32_t *__restrict__ off, const double *__restrict__ vxy,
                                                    const double *__restrict__ vslope,
                                                    const int32_t *__restrict__ path_off,
                                                    const double *__restrict__ path, int has_moving,
                                                    const int32_t *__restrict__ orig, int m_begin, int m_end,
                                                    double robot_radius, uint8_t *__restrict__ hit,
                                                    int32_t *__restrict__ first_hit,
                                                    const unsigned short *__restrict__ near_lists,
                                                    const unsigned short *__restrict__ near_cnt, double list_r,
                                                    int tail_eighths) {
  bool act;
  double ax = 0, ay = 0, at = 0, bx = 0, by = 0, bt = 0;
#ifdef RRTX_TILE_CLOCKS
  unsigned long long acc_box = 0ull, acc_hand = 0ull, acc_a1 = 0ull, acc_a2 = 0ull, acc_b = 0ull;
#endif
  RRTX_PE_T(t_start);
  int qi_mine = 0;
  if (csr.q) {
    // (the entry's owner and node are asked for before the list length is known: the arrays hold `cap` entries, and what
    // lies past the last written one is never looked at)
    int n = 0;
    bool lane_on = true;
    long long total;
    if (tail_eighths > 0) {
      // The last eighths of the entries go through in waves of 32: the kernel ends with the GPU draining (6 466 waves of
      // ~28 us for 5 120 slots), and what runs then should be short rather than full.  (The entry a lane holds then
      // depends on the list length, so these loads wait for it.)
      total = csr.offsets[csr.nq];
      if (total > csr.cap) return;
      const long long t1 = (total / 8 * (8 - tail_eighths)) & ~63ll;
      const long long wv = i >> 6;
      if (wv >= (t1 >> 6)) {
        i = t1 + 32 * (wv - (t1 >> 6)) + (threadIdx.x & 31);
        lane_on = (threadIdx.x & 63) < 32;
      }
      if (lane_on && i < csr.cap) { qi_mine = csr.owner[i]; n = csr.idx[i]; }
    } else {
      if (i < csr.cap) { qi_mine = csr.owner[i]; n = csr.idx[i]; }        // (CSR mode: i = entry, both directions)
      total = csr.offsets[csr.nq];
      if (total > csr.cap) return;          // capacity overflow: the CSR arrays are only partly written
    }
    act = lane_on && i < total && (unsigned)qi_mine < (unsigned)csr.nq && (unsigned)n < (unsigned)csr.n_nodes;   // defensive
    if (act) {
      const double *s = csr.q + (size_t)qi_mine * stride;
      const double4 g = csr.nodes_aos[n];
      ax = s[0]; ay = s[1]; at = s[2]; bx = g.x; by = g.y; bt = g.z;      // forward: sample -> neighbour
    } else qi_mine = 0;
    if (__ballot(act) == 0ull) return;    // the grid covers the caller's capacity
  } else {
    act = i < ne;
    if (act) {
      ax = p0[i * stride + 0]; ay = p0[i * stride + 1];
      bx = p1[i * stride + 0]; by = p1[i * stride + 1];
      if (has_moving) { at = p0[i * stride + 2]; bt = p1[i * stride + 2]; }   // startPoint[3] = time
    }
  }
  // The list is walked 32 obstacles at a time.  First every lane (= edge) drops, with a box test, the
  // obstacles whose bounding circle it cannot reach (explicitEdgeCheck2D's first test, :1536-1539,
  // fails for them for certain) and keeps the rest as a bit mask.  The
  // surviving (edge, obstacle) pairs of the wave are then queued and dealt out one per lane for the tests
  // (stages A and B below), whichever edge they belong to, so the wave does not wait on the one or two edges
  // that lie near several obstacles.  An edge's first
  // hit is the smallest list position among its hits; an edge that has hit takes no part in later
  // groups.  The set of tests that can decide a result and the arithmetic of each are unchanged.
  const int lane = threadIdx.x & 63;
  // CSR mode: the 64 edges of a wave belong to a few consecutive samples (the CSR is ordered by sample), and every
  // point of a candidate edge lies within the edge's length of its sample.  An obstacle whose bounding circle
  // (inflated by the robot radius, as in the box test below) stays farther than the wave's longest edge from
  // every one of those samples fails explicitEdgeCheck2D's first test for every edge of the wave: it is left
  // out of the walk.  What remains (typically a dozen of 256) is walked exactly as before, in list order.
  int n_walk = m_end - m_begin;
  bool listed = false;
  if (csr.q) {
    const unsigned long long amask = __ballot(act);
    const int lf = __ffsll((long long)amask) - 1, ll = 63 - __clzll((long long)amask);
    const int s0 = __builtin_amdgcn_readlane(qi_mine, lf), s1 = __builtin_amdgcn_readlane(qi_mine, ll);
    const bool span_ok = s1 >= s0 && s1 - s0 < kPolyWaveSamples;
    // The sample pass (points_polygons_flag_kernel) has listed, per sample, the obstacles within the same bound taken
    // with the radius of the search ball in place of the wave's longest edge -- a superset, since no candidate edge is
    // longer than that radius (checked below: lmax <= list_r).  The wave's list is the union of its samples' lists;
    // it is put together here, while the edges' coordinates are still on their way.
    const int mm = m_end - m_begin;
    bool merged = false;
    if (span_ok && near_cnt != nullptr && m_begin == 0 && mm <= 64 * kPolyBitWords) {
      if (lane < kPolyBitWords) w.bits[lane] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // (the counts in one load, the lists four samples at a time without waiting for the counts: a row of the list array
      // is kPolyListCap entries whatever its count, what lies past the count is dropped below)
      const int span = s1 - s0 + 1;
      const int c_mine = lane < span ? (int)near_cnt[s0 + lane] : 0;
      int jv[4];
      for (int k0 = 0; k0 < span; k0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) jv[u] = (k0 + u < span) ? (int)near_lists[(size_t)(s0 + k0 + u) * kPolyListCap + lane] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k0 + u >= span) break;
          const int c = __shfl(c_mine, k0 + u);                 // wave-uniform
          if (lane < c && jv[u] < mm) atomicOr(&w.bits[jv[u] >> 6], 1ull << (jv[u] & 63));
        }
      }
      merged = __ballot(c_mine > kPolyListCap) == 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double l2 = 0.0;
    if (act) { const double ex = bx - ax, ey = by - ay; l2 = ex * ex + ey * ey; }
    const bool fin = (l2 - l2 == 0.0);               // false for NaN / inf lengths
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l2 = fmax(l2, __shfl_xor(l2, off));
    double lmax = sqrt_rn(l2) * (1.0 + 1e-9);
    if (__ballot(!fin) != 0ull) lmax = __builtin_inf();          // a non-finite edge: keep everything
    if (span_ok) {
      int nc = 0;                                                  // wave-uniform
      const bool from_lists = merged && lmax <= list_r;
      if (from_lists) {
        for (int wd = 0; wd * 64 < mm; ++wd) {
          const unsigned long long km = w.bits[wd];
          const bool keep = ((km >> lane) & 1ull) != 0ull;
          const int at = nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
          if (keep && at < kPolyWaveCand) w.wc[at] = (short)(wd * 64 + lane);
          nc += __popcll(km);
        }
      }
      if (!from_lists)
      for (int j0 = m_begin; j0 < m_end; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < m_end) {
          const int kind = (int)meta[4 * j + 3];
          if (kind == 6 || kind == 7) keep = true;               // no bounding test for obstacles that move (:1532)
          else {
            const double cx = meta[4 * j + 0], cy = meta[4 * j + 1];
            const double R = fabs(robot_radius + meta[4 * j + 2]) * (1.0 + 1e-9) + 1e-9 * (1.0 + fabs(cx) + fabs(cy));
            for (int sidx = s0; sidx <= s1; ++sidx) {
              const double sx = csr.q[(size_t)sidx * stride], sy = csr.q[(size_t)sidx * stride + 1];   // wave-uniform
              const double ddx = cx - sx, ddy = cy - sy;
              const double bnd = (R + lmax) * (1.0 + 1e-9) + 1e-9 * (fabs(sx) + fabs(sy));
              keep = keep || !(ddx * ddx + ddy * ddy > bnd * bnd);   // NaN anywhere: kept
            }
          }
        }
        const unsigned long long km = __ballot(keep);
        const int at = nc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
        if (keep && at < kPolyWaveCand) w.wc[at] = (short)(j - m_begin);
        nc += __popcll(km);
      }
      if (nc <= kPolyWaveCand && m_end - m_begin < 32768) { listed = true; n_walk = nc; }
    }
  }
  RRTX_PE_T(t_listed);
  w.e[0][lane] = ax; w.e[1][lane] = ay; w.e[2][lane] = bx; w.e[3][lane] = by;
  if constexpr (MOVING) { w.e[4][lane] = at; w.e[5][lane] = bt; }
  w.em[lane] = (by - ay) / (bx - ax);          // (read only where the edge is not "close to vertical")
  if (PAIRED) w.emr[lane] = (ay - by) / (ax - bx);
  w.first[lane] = 0x7fffffff;
  w.first[lane + 64] = 0x7fffffff;
  w.head[lane] = 0u;                               // (chunk numbers start at 1)
  // The edge's own box, widened by far more than any rounding of the exact test (1e-9 relative against
  // ~1e-15) and rounded outward to fp32.  NaN-propagating min / max: an edge with a NaN coordinate keeps
  // every obstacle (all comparisons below are false); inf / overflow give an unbounded box.
  float exlo, exhi, eylo, eyhi;
  {
    const double eslack = 1e-9 * jl_max(jl_max(fabs(ax), fabs(bx)), jl_max(fabs(ay), fabs(by)));
    exlo = __double2float_rd(jl_min(ax, bx) - eslack); exhi = __double2float_ru(jl_max(ax, bx) + eslack);
    eylo = __double2float_rd(jl_min(ay, by) - eslack); eyhi = __double2float_ru(jl_max(ay, by) + eslack);
  }
  bool done = !act, done_r = PAIRED ? !act : true;
  int first = -1, first_r = -1;
  unsigned chunk_no = 0u;                          // stage A's chunks of polygon sides, numbered through the whole kernel
  for (int g0 = 0; g0 < n_walk; g0 += 32) {
    const int jn = min(32, n_walk - g0);
    RRTX_PE_T(t_g0);
    // boxes of the group's bounding circles (lane b = b-th obstacle of the group), widened the same way and
    // rounded outward to fp32: a pair whose boxes are disjoint fails the reference's first test for
    // certain.  Kinds 6 / 7 have no bounding test (:1532): unbounded box.
    if (lane < jn) {
      const int j = m_begin + (listed ? (int)w.wc[g0 + lane] : g0 + lane);
      w.jidx[lane] = j;
      // (the group's records stay in LDS for stages A and B: neither goes back to memory for them)
      const double4 mt = reinterpret_cast<const double4 *>(meta)[j];
      const int v0 = off[j];
      w.gmeta[lane] = mt;
      w.goff[lane] = v0;
      w.gcnt[lane] = off[j + 1] - v0;
      const int kind = (int)mt.w;
      const float inf = __builtin_inff();
      float4 o = {-inf, inf, -inf, inf};
      if (kind != 6 && kind != 7) {
        const double cx = mt.x, cy = mt.y;
        const double R = fabs(robot_radius + mt.z) * (1.0 + 1e-9) + 1e-9 * (1.0 + fabs(cx) + fabs(cy));
        o.x = __double2float_rd(cx - R); o.y = __double2float_ru(cx + R);
        o.z = __double2float_rd(cy - R); o.w = __double2float_ru(cy + R);
      }
      w.box[lane] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned cand = 0;
    if (!(done && done_r)) {         // (PAIRED: as long as either direction is undecided)
      for (int b = 0; b < jn; ++b) {
        const float4 o = w.box[b];
        const bool c = !(exhi < o.x || exlo > o.y || eyhi < o.z || eylo > o.w);
        cand |= (c ? 1u : 0u) << b;
      }
    }
    RRTX_PE_T(t_g1);
    RRTX_PE_ACC(acc_box, t_g0, t_g1);
    if (__ballot(cand != 0u) != 0ull) {
      // The pairs are decided in two stages so that the lanes stay together.  Stage A, lane = pair: the reference's
      // first test (bounding circle, explicitEdgeCheck2D :1536-1539), which settles balls and the obstacles that
      // move in time and drops about half of the polygons; for the others a polygon segment is queued unless its
      // box stays farther than the robot radius from the edge's box in x or in y AND one of segmentDistSqrd's two
      // side tests (R/DRRT.jl:1150-1188, evaluated here with the reference's own operations: the slopes -- the
      // polygon sides' are divided once when the list is set --, the two differences, the strict comparisons)
      // finds both ends of one segment on one side of the other's line.  For such
      // a segment the reference returns the smallest of four point-to-segment distances, each at least the gap,
      // never below robotRadius^2.  The side test is part of the condition because the reference answers 0.0 --
      // a hit -- whenever neither side test separates the two, however far apart they are: segments on one common
      // line (an edge running along the extension of a polygon side), NaN differences.
      // Stage B, lane = queued (edge, segment): the
      // reference's segment test; any segment within the robot radius is a hit (an OR over the segments, their
      // order does not matter).  Same tests on the pairs that can decide anything, same arithmetic; a hit enters
      // the edge's first-hit minimum as before.
      const double rr2 = robot_radius * robot_radius;
      const double gap_min = fabs(robot_radius) * (1.0 + 1e-9);
      int nqd = 0;                                          // wave-uniform: tests in the queue
      auto stage_b = [&]() {
        RRTX_PE_T(t_b0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q0 = 0; q0 < nqd; q0 += 64) {
          const int qi = q0 + lane;
          if (qi < nqd) {
            const unsigned ent = w.pq[qi];
            const int owner = (int)(ent & 63u), sg = (int)((ent >> 11) & 0x7ffffu);
            const int slot = (int)((ent >> 6) & 31u);
            const int j = w.jidx[slot];
            const int vb0 = w.goff[slot], P = w.gcnt[slot];
            const int va = vb0 + (sg == 0 ? P - 1 : sg - 1), vb = vb0 + sg;
            if (PAIRED) {
              double rF, rR;
              segment_dist_sqrd_both(w.e[0][owner], w.e[1][owner], w.e[2][owner], w.e[3][owner], vxy[2 * va], vxy[2 * va + 1],
                                     vxy[2 * vb], vxy[2 * vb + 1], rF, rR);
              if (((ent >> 30) & 1u) && rF < rr2) atomicMin(&w.first[2 * owner], j);
              if ((ent >> 31) && rR < rr2) atomicMin(&w.first[2 * owner + 1], j);
            } else if (segment_dist_sqrd(w.e[0][owner], w.e[1][owner], w.e[2][owner], w.e[3][owner], vxy[2 * va],
                                         vxy[2 * va + 1], vxy[2 * vb], vxy[2 * vb + 1]) < rr2)
              atomicMin(&w.first[owner], j);
          }
        }
        nqd = 0;
        RRTX_PE_T(t_b1);
        RRTX_PE_ACC(acc_b, t_b0, t_b1);
      };
      // stage A over the first `np` entries of the pair queue
      int npair = 0;                                        // wave-uniform: pairs queued
      auto stage_a = [&]() {
       __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
       __builtin_amdgcn_wave_barrier();
       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
       for (int p0 = 0; p0 < npair; p0 += 64) {
        RRTX_PE_T(t_a0);
        const int p = p0 + lane;
        int owner = 0, slot = 0, vb0 = 0, P = 0;           // P > 0: a polygon past the bounding circle
        bool pass_f = false, pass_r = false;               // which direction(s) got past the bounding circle
        if (p < npair) {
          const unsigned pe = w.pairq[p];
          owner = (int)(pe & 63u);
          slot = (int)(pe >> 6);
          const int j = w.jidx[slot];
          const double eax = w.e[0][owner], eay = w.e[1][owner], ebx = w.e[2][owner], eby = w.e[3][owner];
          double eat = 0.0, ebt = 0.0;
          if constexpr (MOVING) { eat = w.e[4][owner]; ebt = w.e[5][owner]; }
          const double4 mt = w.gmeta[slot];
          const int kind = (int)mt.w;
          if (kind == 6 || kind == 7) {
            if (edge_hits_moving(eax, eay, eat, ebx, eby, ebt, robot_radius, mt.x,
                                 mt.y, mt.z, path + 3 * (size_t)path_off[j],
                                 path_off[j + 1] - path_off[j]))
              atomicMin(&w.first[PAIRED ? 2 * owner : owner], j);
            if (PAIRED && edge_hits_moving(ebx, eby, ebt, eax, eay, eat, robot_radius, mt.x,
                                           mt.y, mt.z, path + 3 * (size_t)path_off[j],
                                           path_off[j + 1] - path_off[j]))
              atomicMin(&w.first[2 * owner + 1], j);
          } else {
            const double dsq = dist_sqrd_point_to_segment(mt.x, mt.y, eax, eay, ebx, eby);
            const double rr = robot_radius + mt.z;
            pass_f = !(dsq > rr * rr);
            if (PAIRED) {
              // The reverse edge measures the same distance from the other end: the two results differ by rounding only
              // (below 1e-14 of the squared lengths involved), so the reverse test is evaluated only where the forward
              // distance lies within 1e-7 of those lengths of the threshold -- or is not finite.
              const double scale = sq2(mt.x, mt.y, eax, eay) + sq2(ebx, eby, eax, eay) + rr * rr;
              if (fabs(dsq - rr * rr) > 1e-7 * scale) pass_r = pass_f;
              else pass_r = !(dist_sqrd_point_to_segment(mt.x, mt.y, ebx, eby, eax, eay) > rr * rr);
            }
            if (pass_f || pass_r) {
              if (kind == 1) {
                if (pass_f) atomicMin(&w.first[PAIRED ? 2 * owner : owner], j);
                if (pass_r) atomicMin(&w.first[2 * owner + 1], j);
              } else if (kind == 3) {
                vb0 = w.goff[slot];
                P = w.gcnt[slot];
                if (P < 2) P = 0;                          // (:1551: fewer than two vertices never collide)
              }
            }
          }
        }
        // The sides of the polygons that got past the bounding circle, one (pair, side) per lane (side 0 is (last vertex,
        // first vertex)): the pairs' side counts are summed along the wave, every pair marks where its run of sides
        // starts in the current chunk of 64, and lane t finds the last mark at or before it.  (Lane = pair with a loop
        // over the sides kept a third of the lanes busy: half the pairs stop at the bounding circle, the others have
        // three to six sides.)
        RRTX_PE_T(t_a1);
        RRTX_PE_ACC(acc_a1, t_a0, t_a1);
#ifdef RRTX_TILE_CLOCKS
        const unsigned long long b_before = acc_b;
#endif
        int pstart = P;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pstart, o); if (lane >= o) pstart += v; }
        const int total_sides = __builtin_amdgcn_readlane(pstart, 63);
        pstart -= P;                                       // the pair's first side is number pstart of the wave
        const int pk = owner | (slot << 6) | (pass_f ? 1 << 11 : 0) | (pass_r ? 1 << 12 : 0);
        for (int t0 = 0; t0 < total_sides; t0 += 64) {
          ++chunk_no;
          if (P > 0 && pstart < t0 + 64 && pstart + P > t0) w.head[max(pstart, t0) - t0] = (unsigned)lane | (chunk_no << 6);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const unsigned hv = w.head[lane];
          const unsigned long long heads = __ballot((hv >> 6) == chunk_no);       // (bit 0 is always set: some pair covers t0)
          const unsigned long long below = heads & ((2ull << lane) - 1ull);
          const int k = __shfl((int)hv, below ? 63 - __clzll((long long)below) : 0) & 63;
          const int pk_e = __shfl(pk, k), vb0_e = __shfl(vb0, k), P_e = __shfl(P, k), sg = t0 + lane - __shfl(pstart, k);
          bool push = false;
          unsigned ent = 0u;
          if (t0 + lane < total_sides) {
            const int owner_e = pk_e & 63;
            const bool pf = (pk_e >> 11) & 1, pr = (pk_e >> 12) & 1;
            const double pax = w.e[0][owner_e], pay = w.e[1][owner_e], pbx = w.e[2][owner_e], pby = w.e[3][owner_e];
            const double elx = fmin(pax, pbx), ehx = fmax(pax, pbx), ely = fmin(pay, pby), ehy = fmax(pay, pby);
            double slack = gap_min + 1e-9 * (fabs(elx) + fabs(ehx) + fabs(ely) + fabs(ehy));
            // (fmin / fmax drop a NaN operand: an edge with a non-finite coordinate keeps every segment)
            if (!(pax - pax == 0.0 && pay - pay == 0.0 && pbx - pbx == 0.0 && pby - pby == 0.0)) slack = __builtin_inf();
            const bool evert = fabs(pbx - pax) < .000001;  // the edge is "close to vertical" (:1151)
            double em = 0.0, em_r = 0.0;                   // the edge's slope (R/DRRT.jl:1158), and as the reverse edge divides it
            if (!evert) { em = w.em[owner_e]; if (PAIRED) em_r = w.emr[owner_e]; }
            const int va = vb0_e + (sg == 0 ? P_e - 1 : sg - 1), vb = vb0_e + sg;
            const double Ax = vxy[2 * va], Ay = vxy[2 * va + 1], Bx = vxy[2 * vb], By = vxy[2 * vb + 1];
            const bool apart = (fmin(Ax, Bx) - ehx > slack) || (elx - fmax(Ax, Bx) > slack) ||
                               (fmin(Ay, By) - ehy > slack) || (ely - fmax(Ay, By) > slack);
            const bool finite = (Ax - Ax == 0.0) && (Ay - Ay == 0.0) && (Bx - Bx == 0.0) && (By - By == 0.0);
            bool one_side, one_side_r = false;             // segmentDistSqrd's side tests, as the reference computes them
            if (evert) {
              one_side = (Ax >= pax && Bx >= pax) || (Ax <= pax && Bx <= pax);
              if (PAIRED) one_side_r = (Ax >= pbx && Bx >= pbx) || (Ax <= pbx && Bx <= pbx);
            } else {
              // (both differences strictly positive or both strictly negative <=> their product is positive, except that
              // the product of two tiny ones can round to zero: the side then just goes to the exact test)
              const double diff_a1 = (em * (Ax - pax) + pay) - Ay, diff_b1 = (em * (Bx - pax) + pay) - By;
              one_side = diff_a1 * diff_b1 > 0.0;
              if (PAIRED) {                                // the reverse edge's line is anchored at ITS first point
                const double diff_a1r = (em_r * (Ax - pbx) + pby) - Ay, diff_b1r = (em_r * (Bx - pbx) + pby) - By;
                one_side_r = diff_a1r * diff_b1r > 0.0;
              }
            }
            bool sep_q;                                    // the second side test is the same for both directions
            if (fabs(Bx - Ax) < .000001) sep_q = (pax >= Ax && pbx >= Ax) || (pax <= Ax && pbx <= Ax);
            else {
              const double qm = vslope[vb];                // (By - Ay) / (Bx - Ax), divided once when the list was set
              const double diff_a = (qm * (pax - Ax) + Ay) - pay;
              const double diff_b = (qm * (pbx - Ax) + Ay) - pby;
              sep_q = diff_a * diff_b > 0.0;
            }
            // (slack = +inf or NaN: never apart)
            const bool need_f = pf && (!(apart && (one_side || sep_q)) || !finite);
            const bool need_r = PAIRED && pr && (!(apart && (one_side_r || sep_q)) || !finite);
            push = need_f || need_r;
            ent = (unsigned)(pk_e & 0x7ff) | ((unsigned)sg << 11) | (need_f ? 1u << 30 : 0u) | (need_r ? 1u << 31 : 0u);
          }
          const unsigned long long sv = __ballot(push);
          if (push)
            w.pq[nqd + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sv, 0u))] = ent;
          nqd += __popcll(sv);
          if (nqd > QUEUE - 64) stage_b();
        }
        RRTX_PE_T(t_a2);
#ifdef RRTX_TILE_CLOCKS
        acc_a2 += (t_a2 - t_a1) - (acc_b - b_before);
#endif
       }
       npair = 0;
      };
      // the wave's (edge, obstacle) pairs: every edge hands out one set bit of its mask per round, in list order.  After
      // every second round the queued pairs are decided, and an edge that has its hit hands out nothing more: a later
      // obstacle of the list can neither change the boolean nor lower the first-hit position (55 % of C4's candidate edges
      // collide, typically with one of the first obstacles they come near).
      int rounds = 0;
      RRTX_PE_T(t_h0);
#ifdef RRTX_TILE_CLOCKS
      const unsigned long long in_before = acc_a1 + acc_a2 + acc_b;
#endif
      for (unsigned rem = cand;;) {
        const bool any = __ballot(rem != 0u) != 0ull;
        if (any) {
          const bool has = rem != 0u;
          const int slot = has ? __ffs((int)rem) - 1 : 0;
          rem &= rem - 1u;
          const unsigned long long pv = __ballot(has);
          if (has)
            w.pairq[npair + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pv, 0u))] =
                (unsigned short)(lane | (slot << 6));
          npair += __popcll(pv);
          ++rounds;
        }
        if (!any || npair > kPolyPairs - 64 || rounds == 2) {
          stage_a();
          stage_b();
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          bool fin = w.first[PAIRED ? 2 * lane : lane] != 0x7fffffff;
          if (PAIRED) fin = fin && w.first[2 * lane + 1] != 0x7fffffff;   // (the lane hands out for both directions)
          if (fin) rem = 0u;
        }
        if (!any) break;
      }
      RRTX_PE_T(t_h1);
#ifdef RRTX_TILE_CLOCKS
      acc_hand += (t_h1 - t_h0) - (acc_a1 + acc_a2 + acc_b - in_before);
#endif
      const int f = w.first[PAIRED ? 2 * lane : lane];
      if (!done && f != 0x7fffffff) { done = true; first = orig[f]; }
      if (PAIRED) {
        const int fr = w.first[2 * lane + 1];
        if (!done_r && fr != 0x7fffffff) { done_r = true; first_r = orig[fr]; }
      }
    }
    if (__ballot(!(done && done_r)) == 0ull) break;
  }
  if (act) {
    if (csr.q) {
      hit[i] = first >= 0 ? 1 : 0;
      csr.hit_in[i] = first_r >= 0 ? 1 : 0;
    } else {
      hit[i] = first >= 0 ? 1 : 0;
      if (first_hit) first_hit[i] = first;
    }
  }
#ifdef RRTX_TILE_CLOCKS
  {
    const unsigned long long t_end = clock64();
    if (lane == 0) {
      unsigned long long *row = g_pe_clk + (size_t)((i >> 6) & (kClkRows - 1)) * 8;
      row[0] = t_listed - t_start; row[1] = acc_box; row[2] = acc_hand; row[3] = acc_a1; row[4] = acc_a2; row[5] = acc_b;
      row[6] = t_end - t_start; row[7] = 1ull | (wall_clock64() << 8);          // (bits 8..: 100 MHz time the wave ended at)
    }
  }
#endif
}

// WAVES waves per workgroup, each with its own 64 edges and its own scratch (no workgroup barrier anywhere).  The fused
// extend preamble runs one wave per workgroup: the waves' running times differ by a factor of several, and a workgroup's
// registers are only handed on when its last wave is through (measured: 0.166 -> 0.161 ms per step; a resident grid
// striding over the edges instead costs registers and was slower).  OCC: waves per SIMD the register budget is cut for
// (the wave's 9.3 KB of LDS allow 16 waves per CU = 4 per SIMD, so nothing is gained by spilling down to 96 registers).
template <bool PAIRED, int WAVES, int OCC, bool MOVING>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(OCC))) void edges_polygons_kernel(const double *__restrict__ p0,
                                                                    const double *__restrict__ p1, int stride,
                                                                    long long ne, PolyCsr csr,
                                                                    const double *__restrict__ meta,
                                                                    const int32_t *__restrict__ off,
                                                                    const double *__restrict__ vxy,
                                                                    const double *__restrict__ vslope,
                                                                    const int32_t *__restrict__ path_off,
                                                                    const double *__restrict__ path, int has_moving,
                                                                    const int32_t *__restrict__ orig, int m_begin,
                                                                    int m_end, double robot_radius,
                                                                    uint8_t *__restrict__ hit,
                                                                    int32_t *__restrict__ first_hit,
                                                                    const unsigned short *__restrict__ near_lists,
                                                                    const unsigned short *__restrict__ near_cnt,
                                                                    double list_r, int tail_eighths) {
  constexpr int QUEUE = MOVING ? kPolyQueue : kPolyQueue - 64;
  __shared__ PolyWaveT<MOVING, QUEUE> s_w[WAVES];
  PolyWaveT<MOVING, QUEUE> &w = s_w[threadIdx.x >> 6];
  edges_polygons_wave<PAIRED, MOVING, QUEUE>((long long)blockIdx.x * (64 * WAVES) + threadIdx.x, w, p0, p1, stride, ne, csr, meta, off, vxy,
                              vslope, path_off, path, has_moving, orig, m_begin, m_end, robot_radius, hit, first_hit,
                              near_lists, near_cnt, list_r, tail_eighths);
}

// pointInPolygon (MacMartin crossings), R/DRRT.jl:1009-1056
// MOV: the vertices are originalPolygon .+ (ox, oy), the transformed copy kinds 6 / 7 test against
// (R/DRRT.jl:1301-1302)
template <bool MOV = false>
__device__ bool point_in_polygon(double px, double py, const double *__restrict__ vxy, int b, int e,
                                 double ox = 0.0, double oy = 0.0) {
  const int P = e - b;
  if (P < 2) return false;
  int crossings = 0;
  double sx = vxy[2 * (e - 1)], sy = vxy[2 * (e - 1) + 1];
  if (MOV) { sx = sx + ox; sy = sy + oy; }
  for (int v = b; v < e; ++v) {
    double ex = vxy[2 * v], ey = vxy[2 * v + 1];
    if (MOV) { ex = ex + ox; ey = ey + oy; }
    if ((sy > py && ey < py) || (sy < py && ey > py)) {
      if (sx > px && ex > px) {
        crossings += 1;
      } else if (sx < px && ex < px) {
      } else {
        double T = 2 * jl_max(sx, ex);
        double x = (-((sx * ey - sy * ex) * (px - T)) + ((sx - ex) * (px * py - py * T))) /
                   ((sy - ey) * (px - T));
        if (x > px) crossings += 1;
      }
    }
    sx = ex; sy = ey;
  }
  return (crossings & 1) != 0;
}

// distToPolygonSqrd, R/DRRT.jl:1087-1106
template <bool MOV = false>
__device__ double dist_to_polygon_sqrd(double px, double py, const double *__restrict__ vxy, int b, int e,
                                       double ox = 0.0, double oy = 0.0) {
  double best = __builtin_inf();
  double sx = vxy[2 * (e - 1)], sy = vxy[2 * (e - 1) + 1];
  if (MOV) { sx = sx + ox; sy = sy + oy; }
  for (int v = b; v < e; ++v) {
    double ex = vxy[2 * v], ey = vxy[2 * v + 1];
    if (MOV) { ex = ex + ox; ey = ey + oy; }
    double dd = dist_sqrd_point_to_segment(px, py, sx, sy, ex, ey);
    if (dd < best) best = dd;
    sx = ex; sy = ey;
  }
  return best;
}

// the value lane l (a constant) holds, as a wave-uniform value
__device__ __forceinline__ double lane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// One polygon against one point with the eight lanes of a lane group dealt over the polygon's edges (the eight
// hold the same point and the same polygon; other groups of the wave work on other polygons at the same time, and
// the whole wave must call this -- it uses ballots): pointInPolygon's crossing count (:1009-1056) and
// distToPolygonSqrd's minimum (:1087-1106).  Each edge's contribution is computed as the sequential loops do; a
// crossing count is an integer sum and the minimum (NaN never taken, as `dd < best` never takes it) does not depend
// on the order.  n_rounds: wave-uniform, ceil(longest polygon of the wave / 8).
__device__ void group8_point_vs_polygon(double px, double py, const double *__restrict__ vxy, int b, int e, bool mov,
                                        double ox, double oy, int n_rounds, bool &inside, double &dsq) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7, shift = lane & 56;
  int crossings = 0;
  double best = __builtin_inf();
  for (int k = 0; k < n_rounds; ++k) {
    const int v = b + 8 * k + sub;
    bool c = false;
    if (v < e) {
      const int sv = (v == b) ? e - 1 : v - 1;
      double sx = vxy[2 * sv], sy = vxy[2 * sv + 1], ex = vxy[2 * v], ey = vxy[2 * v + 1];
      if (mov) { sx = sx + ox; sy = sy + oy; ex = ex + ox; ey = ey + oy; }
      if ((sy > py && ey < py) || (sy < py && ey > py)) {
        if (sx > px && ex > px) {
          c = true;
        } else if (sx < px && ex < px) {
        } else {
          const double T = 2 * jl_max(sx, ex);
          const double x = (-((sx * ey - sy * ex) * (px - T)) + ((sx - ex) * (px * py - py * T))) /
                           ((sy - ey) * (px - T));
          c = x > px;
        }
      }
      const double dd = dist_sqrd_point_to_segment(px, py, sx, sy, ex, ey);
      if (dd < best) best = dd;
    }
    crossings += __popc((unsigned)((__ballot(c) >> shift) & 0xffull));
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    const double other = __shfl_xor(best, o);
    if (other < best) best = other;
  }
  inside = (e - b >= 2) && (crossings & 1) != 0;
  dsq = best;
}

// explicitPointCheck over polygons (R/DRRT.jl:1434-1470): quickCheck pass (:1258-1331) then the
// explicitPointCheck2D loop (:1343-1427); Wdist over the first two coordinates.  One wave per point.
//   quick pass: lane = obstacle, 64 at a time; "inside any obstacle" is an OR, so order is free.
//   explicit pass: the reference skips obstacle j when (centre distance - robotRadius) - radius exceeds
//   the running certificate, which only changes when an obstacle is evaluated.  The next eight obstacles of the
//   list that the current certificate does not skip -- found with a ballot over the 64 obstacles of the chunk --
//   are evaluated at once, eight lanes each (polygon containment + distance dealt over the polygon's edges), and
//   their results are then gone through in list order with the reference's skip test against the certificate as
//   it stands by then: an obstacle an earlier one of the eight makes the reference skip is dropped unlooked-at
//   (its evaluation was wasted work, not a result).  Same obstacles taken into account in the same order with
//   the same arithmetic as the sequential loop -- whether or not the caller wants the certificate: a far obstacle
//   cannot be left out of a flag-only call, since pointInPolygon answers "inside" for points far outside a polygon
//   when the ray meets a vertex whose two sides lie on opposite sides of it (the crossing tests are strict), and
//   whether the reference looks at such an obstacle is decided by the certificates of everything before it.
__device__ __forceinline__ void point_full_loop(const double *__restrict__ p, int stride, long long i,
                                                const double *__restrict__ meta, const int32_t *__restrict__ off,
                                                const double *__restrict__ vxy, const int32_t *__restrict__ path_off,
                                                const double *__restrict__ path, int has_moving, int m,
                                                double robot_radius, uint8_t *__restrict__ unsafe,
                                                double *__restrict__ clearance) {
  const int lane = threadIdx.x & 63;
  const double px = p[i * stride + 0], py = p[i * stride + 1];
  const double pt = has_moving ? p[i * stride + 2] : 0.0;      // point[3] = time (kinds 6 / 7)
  // One pass over the list, 64 obstacles at a time: the quickCheck of the group (an OR: "inside any obstacle"),
  // then the group's share of the explicitPointCheck2D loop.  The reference runs the whole quick pass first; both
  // end the call with unsafe = 1 and certificate 0 when they find something and neither influences the other
  // (the certificate comes from the explicit loop alone), so doing them group by group gives the same outputs
  // with every obstacle record read and every centre distance taken once.
  double ret_cert = __builtin_inf();
  for (int j0 = 0; j0 < m; j0 += 64) {
    const int j = j0 + lane;
    const bool valid = j < m;
    double rad = 0.0, dx = 0.0, dy = 0.0, tdc = 0.0;
    int kind = 0, vb_j = 0, ve_j = 0;
    bool in = false;
    if (valid) {
      vb_j = off[j]; ve_j = off[j + 1];                        // (asked for here: one round trip less per round below)
      double cx = meta[4 * j + 0], cy = meta[4 * j + 1];
      rad = meta[4 * j + 2];
      kind = (int)meta[4 * j + 3];
      const bool mov = (kind == 6 || kind == 7);
      if (mov) {                                               // R/DRRT.jl:1289-1305, 1395-1408
        transform_obs_to_time(path + 3 * (size_t)path_off[j], path_off[j + 1] - path_off[j], pt, dx, dy);
        cx = cx + dx; cy = cy + dy;
      }
      const double dc = sqrt_rn(sq2(cx, cy, px, py));
      // ---- quickCheck (:1258-1331) ----
      if (!(dc > rad))
        in = kind == 1 || (kind == 3 && point_in_polygon(px, py, vxy, vb_j, ve_j)) ||
             (mov && point_in_polygon<true>(px, py, vxy, vb_j, ve_j, dx, dy));
      tdc = dc - robot_radius;                                 // distance from the robot boundary to the centre
    }
    if (__ballot(in) != 0ull) {
      if (lane == 0) { unsafe[i] = 1; if (clearance) clearance[i] = 0.0; }
      return;
    }
    // ---- explicitPointCheck2D over the group (:1343-1427) ----
    unsigned long long todo = __ballot(valid);
    for (;;) {
      unsigned long long c = __ballot(valid && !(tdc - rad > ret_cert)) & todo;
      if (c == 0ull) break;
      // the next (up to) eight obstacles of the list that are not skipped: lane group g takes the g-th of them
      int my_l = -1, n_c = 0, last_l = 0;
      for (; n_c < 8 && c != 0ull; ++n_c) {
        last_l = __ffsll((long long)c) - 1;
        c &= c - 1ull;
        if ((lane >> 3) == n_c) my_l = last_l;
      }
      todo &= ~((2ull << last_l) - 1ull);
      const int src = my_l >= 0 ? my_l : 0;
      const int kind_l = __shfl(kind, src);
      const double tdc_l = __shfl(tdc, src), rad_l = __shfl(rad, src);
      const double dx_l = __shfl(dx, src), dy_l = __shfl(dy, src);
      const bool poly = my_l >= 0 && kind_l != 1;
      int vb = __shfl(vb_j, src), ve = __shfl(ve_j, src);
      if (!poly) { vb = 0; ve = 0; }
      int n_rounds = (ve - vb + 7) >> 3;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) n_rounds = max(n_rounds, __shfl_xor(n_rounds, o));
      bool inside;
      double dsq;
      group8_point_vs_polygon(px, py, vxy, vb, ve, kind_l != 3, dx_l, dy_l, n_rounds, inside, dsq);
      double this_dist;
      bool bad;
      if (kind_l == 1) {
        this_dist = tdc_l - rad_l;
        bad = this_dist < 0.0;
      } else if (inside) {
        this_dist = tdc_l;                                     // (not looked at: the call ends here)
        bad = true;
      } else {
        this_dist = sqrt_rn(dsq) - robot_radius;
        bad = this_dist < 0.0;
      }
      // the eight in list order, each behind the reference's skip test against the certificate as it stands now
      // (v_readlane from the first lane of each group: the values are wave-uniform from here on)
      const double gap_l = tdc_l - rad_l;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g >= n_c) break;
        if (lane_f64(gap_l, 8 * g) > ret_cert) continue;
        if (__builtin_amdgcn_readlane((int)bad, 8 * g) != 0) {
          if (lane == 0) { unsafe[i] = 1; if (clearance) clearance[i] = 0.0; }
          return;
        }
        const double this_cert = jl_min(ret_cert, lane_f64(this_dist, 8 * g));
        if (this_cert < ret_cert) ret_cert = this_cert;
      }
    }
  }
  if (lane == 0) { unsafe[i] = 0; if (clearance) clearance[i] = ret_cert; }
}

__global__ __launch_bounds__(256) void points_polygons_kernel(const double *__restrict__ p, int stride,
                                                              long long np, const double *__restrict__ meta,
                                                              const int32_t *__restrict__ off,
                                                              const double *__restrict__ vxy,
                                                              const int32_t *__restrict__ path_off,
                                                              const double *__restrict__ path, int has_moving,
                                                              int m, double robot_radius,
                                                              uint8_t *__restrict__ unsafe,
                                                              double *__restrict__ clearance) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= np) return;
  point_full_loop(p, stride, i, meta, off, vxy, path_off, path, has_moving, m, robot_radius, unsafe, clearance);
}

  // ---- flag-only calls (the fused extend preamble asks only whether the sample is in collision) ----
struct PolyGrid {        // sync_polygon_grid; g == 0: none
  double x0, y0, inv_wx, inv_wy;
  int g;
  const int32_t *start;
  const uint16_t *items;
};

// points_polygons_flag_kernel: explicitPointCheck when only the flag is wanted (the fused extend preamble).
// The flag is an OR over the obstacles the reference's loop EVALUATES of "inside or closer than the robot radius".
//  * an obstacle whose bound (Wdist - robotRadius) - radius is <= 0 is never skipped (the running certificate is
//    >= 0): its own answer counts whatever the list order -- evaluated here exactly as the reference does;
//  * an obstacle with bound > 0 lies beyond the robot's reach; evaluated or not, it can only say "in collision"
//    through pointInPolygon's strict crossing tests, which miscount when the ray meets a vertex (py equal to a
//    vertex's y) or when the point falls in the bounding box of one of its sides (the x-intercept formula of
//    R/DRRT.jl:1040-1046 is then evaluated, and it divides by px - 2 max(sx, ex)).  Outside the polygon's bounding
//    box and with py different from EVERY vertex y of the list (one look-up in the sorted table) the crossing
//    count is decided by comparisons alone and is even: such an obstacle cannot raise the flag, whether or not
//    the certificates of the obstacles before it let the reference look at it.  Inside the box the reference's crossing
//    count is taken: "outside" cannot raise the flag either; "inside" would, if the reference gets to look.
// Anything else -- a bound within 1e-9 of zero, a far polygon whose crossing count says "inside", a y that matches a
// vertex, non-finite input, obstacles that move in time -- takes the full loop above, which is the reference's sequence.
// Two points per wave, 32 lanes each (the list is gone through 32 obstacles at a time per point, the near polygons four
// at a time, eight lanes each): half the waves of the one-point form for the same instructions per wave.
__global__ __launch_bounds__(256) void points_polygons_flag_kernel(const double *__restrict__ p, int stride,
                                                                   long long np, const double *__restrict__ meta,
                                                                   const int32_t *__restrict__ off,
                                                                   const double *__restrict__ vxy, int m,
                                                                   double robot_radius, uint8_t *__restrict__ unsafe,
                                                                   const double *__restrict__ bbox,
                                                                   const double *__restrict__ ytab, int n_ytab,
                                                                   double list_r, unsigned short *__restrict__ lists,
                                                                   unsigned short *__restrict__ list_cnt, PolyGrid grid) {
  const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  const long long i0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  RRTX_PE_T(t_0);
  // (np >= 1; a wave or half past the last point repeats it and writes nothing: the workgroup stays whole for the barriers)
  const long long i = min(i0 + half, np - 1);
  const double px = p[i * stride + 0], py = p[i * stride + 1];
  bool slow = !((px - px == 0.0) && (py - py == 0.0)) || m > 32767;
  if (n_ytab > 0) {
    // two-level look-up of py in the sorted table (at most 64 x 64 entries; longer tables: the full loop), all 64 lanes
    // for one point, then for the other
    const int step = (n_ytab + 63) >> 6;
    if (step > 64) slow = true;
    else {
      const int k0 = lane * step;
      const double head = ytab[min(k0, n_ytab - 1)];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double qy = lane_f64(py, 32 * h);
        const unsigned long long le = __ballot(k0 < n_ytab && head <= qy);
        const int blk = le ? 63 - __clzll((long long)le) : 0;    // last block whose first entry is <= qy
        const int k = blk * step + lane;
        const bool eq = lane < step && k < n_ytab && ytab[k] == qy;
        if (__ballot(eq) != 0ull && half == h) slow = true;
      }
    }
  }
  bool bad = false;
  RRTX_PE_T(t_1);
  // the polygons within reach are collected over the whole list first (typically two or three of 256) and then
  // evaluated by eight lanes each -- not by the one lane that owns them
  __shared__ short s_near[4][2][32];
  short *near_list = s_near[threadIdx.x >> 6][half];
  int n_near = 0;                                                // uniform over the half
  int n_list = 0;
  const bool mine = i0 + half < np;
  // one group of (up to) 32 obstacles per half: lane hl of each half looks at obstacle j of ITS point's list
  auto visit = [&](bool valid, int j, const double4 &mt, const double4 &bb) {
    bool near3 = false, far3 = false, within = false;
    if (valid) {
      const double cx = mt.x, cy = mt.y, rad = mt.z;
      const int kind = (int)mt.w;
      // near / far by squares: reach = robotRadius + radius, the two sure cases leave a band of 1e-9 around it that
      // goes to the full loop; the reference's own bound (Wdist - robotRadius) - radius is only needed for balls
      const double s2 = sq2(cx, cy, px, py);
      const double reach = robot_radius + rad;
      if (lists) {
        // for edges_polygons_kernel: the obstacles a candidate edge of this sample (no longer than list_r) can reach --
        // its own bound (see there) with list_r for the wave's longest edge
        const double R = fabs(reach) * (1.0 + 1e-9) + 1e-9 * (1.0 + fabs(cx) + fabs(cy));
        const double bnd = (R + list_r) * (1.0 + 1e-9) + 1e-9 * (fabs(px) + fabs(py));
        within = !(s2 > bnd * bnd);                             // NaN anywhere: kept
      }
      const double sl = 1e-9 * (1.0 + fabs(cx) + fabs(cy) + fabs(px) + fabs(py) + fabs(reach));
      const double hi = (reach + sl) * (1.0 + 1e-9), lo = (reach - sl) * (1.0 - 1e-9);
      if (s2 > hi * hi && hi >= 0.0) {                          // beyond reach for certain: bound > 0
        // inside the polygon's box (its corners reach past the bounding circle: one point in twelve at C4's density) the
        // crossing count is taken below; only if it says "inside" does the reference's order of skipping matter
        const bool outside = px < bb.x || px > bb.y || py < bb.z || py > bb.w;
        if (kind == 3 && !outside) { near3 = true; far3 = true; }
        else if (kind != 1 && !outside) slow = true;
      } else if (lo > 0.0 && s2 < lo * lo && robot_radius >= 0.0) {   // within reach for certain: bound < 0, never skipped
        if (kind == 1) bad = true;                             // (Wdist - robotRadius) - radius < 0
        else if (kind == 3) near3 = true;
        else slow = true;
      } else slow = true;
    }
    const unsigned hm = (unsigned)(__ballot(near3) >> (32 * half));
    const int at = n_near + __popc(hm & ((1u << hl) - 1u));
    if (near3) { if (at < 32) near_list[at] = (short)(j | (far3 ? 0x8000 : 0)); else slow = true; }
    n_near += __popc(hm);
    if (lists) {
      const unsigned wm = (unsigned)(__ballot(within) >> (32 * half));
      const int wat = n_list + __popc(wm & ((1u << hl) - 1u));
      if (within && wat < kPolyListCap && mine) lists[(size_t)i * kPolyListCap + wat] = (unsigned short)j;
      n_list += __popc(wm);
    }
  };
  // With a grid over the obstacles (sync_polygon_grid) a point looks at the obstacles listed for its cell only: every
  // other one is beyond reach for certain, has the point outside its box and is outside the bound of the per-sample
  // lists.  A point outside the grid (or not finite), a negative robot radius, a list too short for a grid: the whole
  // list, as before.
  const double fx = (px - grid.x0) * grid.inv_wx, fy = (py - grid.y0) * grid.inv_wy;
  const bool in_grid = grid.g > 0 && robot_radius >= 0.0 && fx >= 0.0 && fx < (double)grid.g && fy >= 0.0 && fy < (double)grid.g;
  const bool wave_grid = __ballot(!in_grid) == 0ull;
  const int wg_walk = __syncthreads_or(wave_grid ? 0 : 1);
  if (wave_grid) {
    const int cellid = (int)fy * grid.g + (int)fx;
    const int cs = grid.start[cellid], cnt = grid.start[cellid + 1] - cs;
    const int cnt_max = max(cnt, __shfl_xor(cnt, 32));
    for (int t0 = 0; t0 < cnt_max; t0 += 32) {
      const int t = t0 + hl;
      const bool valid = t < cnt;
      int j = 0;
      double4 mt = {0.0, 0.0, 0.0, 0.0}, bb = {0.0, 0.0, 0.0, 0.0};
      if (valid) {
        j = grid.items[cs + t];
        mt = reinterpret_cast<const double4 *>(meta)[j];
        bb = reinterpret_cast<const double4 *>(bbox)[j];
      }
      visit(valid, j, mt, bb);
    }
  }
  if (wg_walk) {
    // the records (centre, radius, kind; box) of 256 obstacles at a time go through LDS: one trip to memory per 256 for
    // the workgroup instead of one per 32 for every wave
    __shared__ double4 s_tab[2][256];
    for (int c0 = 0; c0 < m; c0 += 256) {
      if (c0 > 0) __syncthreads();
      if (c0 + (int)threadIdx.x < m) {
        s_tab[0][threadIdx.x] = reinterpret_cast<const double4 *>(meta)[c0 + threadIdx.x];
        s_tab[1][threadIdx.x] = reinterpret_cast<const double4 *>(bbox)[c0 + threadIdx.x];
      }
      __syncthreads();
      const int c1 = min(m, c0 + 256);
      if (!wave_grid)
        for (int j0 = c0; j0 < c1; j0 += 32) {
          const int j = j0 + hl;
          const bool valid = j < c1;
          visit(valid, j, s_tab[0][valid ? j - c0 : 0], s_tab[1][valid ? j - c0 : 0]);
        }
    }
  }
  if (lists && mine && hl == 0) list_cnt[i] = (unsigned short)min(n_list, 0xffff);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  RRTX_PE_T(t_2);
  const int n_eval = min(n_near, 32);
  const int n_max = max(n_eval, __shfl_xor(n_eval, 32));
  for (int r0 = 0; r0 < n_max; r0 += 4) {
    const int g = r0 + (hl >> 3);
    int vb = 0, ve = 0;
    bool far = false;
    if (g < n_eval) { const int e = near_list[g]; far = (e & 0x8000) != 0; const int j = e & 0x7fff; vb = off[j]; ve = off[j + 1]; }
    if (g < n_eval && ve - vb < 1) slow = true;
    int n_rounds = (ve - vb + 7) >> 3;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_rounds = max(n_rounds, __shfl_xor(n_rounds, o));
    bool inside;
    double dsq;
    group8_point_vs_polygon(px, py, vxy, vb, ve, false, 0.0, 0.0, n_rounds, inside, dsq);
    if (g < n_eval) {
      if (far) slow = slow || inside;          // (farther than the robot radius from every side for certain)
      else bad = bad || inside || (sqrt_rn(dsq) - robot_radius < 0.0);
    }
  }
  const unsigned long long sb = __ballot(slow), bm = __ballot(bad);
  RRTX_PE_T(t_3);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (i0 + h >= np) break;
    const unsigned long long hmask = 0xffffffffull << (32 * h);
    if (unsafe == nullptr) break;                              // (a call for the lists alone)
    if (sb & hmask)
      point_full_loop(p, stride, i0 + h, meta, off, vxy, nullptr, nullptr, 0, m, robot_radius, unsafe, nullptr);
    else if (lane == 0) unsafe[i0 + h] = (bm & hmask) ? 1 : 0;
  }
#ifdef RRTX_TILE_CLOCKS
  {
    const unsigned long long t_end = clock64();
    if (lane == 0 && i0 < np) {
      unsigned long long *row = g_pp_clk + (size_t)((i0 >> 1) & (kClkRows - 1)) * 8;
      row[0] = t_1 - t_0; row[1] = t_2 - t_1; row[2] = t_3 - t_2; row[3] = t_end - t_3; row[4] = t_end - t_0; row[5] = 1ull;
      row[6] = t_0; row[7] = t_end;
    }
  }
#endif
}

// calculateTrajectory(S, ::SimpleEdge), R/DRRT_SimpleEdge_functions.jl:177-181
__global__ void simple_steer_kernel(const double *__restrict__ s, const double *__restrict__ g, int dim,
                                    long long ne, double *__restrict__ dist, double *__restrict__ wdist) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  const double *a = s + i * dim, *b = g + i * dim;
  double w = sq3(a[0], a[1], a[2], b[0], b[1], b[2]);
  double d = w;
  if (dim == 4) d = sq4(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
  if (dist) dist[i] = sqrt_rn(d);
  if (wdist) wdist[i] = sqrt_rn(w);
}

// one wave per 64-bit word: lane j supplies bit j
__global__ __launch_bounds__(256) void pack_hits_kernel(const uint8_t *__restrict__ hit_out,
                                                        const uint8_t *__restrict__ hit_in,
                                                        const int64_t *__restrict__ n_valid, long long cap,
                                                        long long n_words, unsigned long long *__restrict__ words) {
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= n_words) return;
  const int lane = threadIdx.x & 63;
  const long long g = w * 64 + lane;
  long long nv = *n_valid;
  if (nv > cap) nv = cap;
  bool bit = false;
  if (g < cap) bit = (g < nv) && hit_out[g] != 0;
  else if (g < 2 * cap) bit = (g - cap < nv) && hit_in[g - cap] != 0;
  unsigned long long m = __ballot(bit);
  if (lane == 0) words[w] = m;
}

}  // namespace

#ifdef RRTX_TILE_CLOCKS
// out: kClkRows x 8 words, one row per wave (rows of waves that did not run are zero after a reset)
extern "C" int rrtx_debug_polygon_point_clocks(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_clk), sizeof(unsigned long long) * kClkRows * 8);
  if (reset) { void *p = nullptr; rc |= (int)hipGetSymbolAddress(&p, HIP_SYMBOL(g_pp_clk)); rc |= (int)hipMemset(p, 0, sizeof(unsigned long long) * kClkRows * 8); }
  return rc;
}
extern "C" int rrtx_debug_polygon_edge_clocks(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pe_clk), sizeof(unsigned long long) * kClkRows * 8);
  if (reset) { void *p = nullptr; rc |= (int)hipGetSymbolAddress(&p, HIP_SYMBOL(g_pe_clk)); rc |= (int)hipMemset(p, 0, sizeof(unsigned long long) * kClkRows * 8); }
  return rc;
}
#endif

// map a list-position range [begin, end) onto the packed active table
void packed_range(const std::vector<int32_t> &orig, int begin, int end, int &pb, int &pe) {
  pb = 0;
  while (pb < (int)orig.size() && orig[pb] < begin) ++pb;
  pe = pb;
  while (pe < (int)orig.size() && orig[pe] < end) ++pe;
}

// ------------------------------------------------------------- host glue -----

int sync_spheres(rrtx_ctx *ctx, double robot_radius) {
  const bool same_origin = ctx->sph_packed_origin[0] == ctx->origin[0] && ctx->sph_packed_origin[1] == ctx->origin[1] &&
                           ctx->sph_packed_origin[2] == ctx->origin[2];
  if (!ctx->sph_dirty && ctx->sph_packed_rr == robot_radius && same_origin) return RRTX_OK;
  const int m = (int)ctx->sph_active.size();
  std::vector<SphRec> rec, reach;
  std::vector<double> radius, thr_in;
  std::vector<int32_t> orig;
  for (int i = 0; i < m; ++i) {
    if (!ctx->sph_active[i]) continue;
    const double *c = &ctx->sph[4 * (size_t)i];
    SphRec r;
    r.cx = c[0]; r.cy = c[1]; r.cz = c[2];
    r.thr = thr_first_gt(robot_radius + c[3]);
    rec.push_back(r);
    // inflated reach for the conservative midpoint test (kernel: may_touch)
    SphRec rb = r;
    const double R = robot_radius + c[3];
    const double cm = std::fmax(std::fmax(std::fabs(c[0]), std::fabs(c[1])), std::fabs(c[2]));
    const bool usable = std::isfinite(R) && std::isfinite(cm) && cm < 1e100 && std::fabs(R) < 1e100;
    rb.thr = usable ? (std::fmax(R, 0.0) * (1.0 + 1e-12) + 1e-12 * (cm + 1.0)) * (1.0 + 1e-12)
                    : std::numeric_limits<double>::infinity();
    reach.push_back(rb);
    radius.push_back(c[3]);
    thr_in.push_back(thr_first_gt(c[3]));
    orig.push_back(i);
  }
  const int na = (int)rec.size();
  ctx->sph_n_active = na;
  if (na > 0) {
    RRTX_HIP(ctx, ctx->d_sph.ensure(sizeof(SphRec) * na));
    RRTX_HIP(ctx, ctx->d_sph_aux.ensure((sizeof(double) * 2 + sizeof(int32_t)) * na + 64));
    // the packed tables may still be read by kernels in flight on the stream
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_sph.p, rec.data(), sizeof(SphRec) * na, hipMemcpyHostToDevice));
    RRTX_HIP(ctx, ctx->d_sph_reach.ensure(sizeof(SphRec) * na));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_sph_reach.p, reach.data(), sizeof(SphRec) * na, hipMemcpyHostToDevice));
    // fp32 reach table for the packed screen of candidate_edges_kernel: centres relative to the
    // context origin, pair-interleaved {cxA,cxB, cyA,cyB, czA,czB, RA,RB}, padded to groups of 8
    // obstacles with centres at +inf (never within reach).  R~ = RU[(R' + 3 eps |c|_1)(1 + 8 eps)].
    {
      const int ng = (na + 7) / 8;
      std::vector<float> tf((size_t)ng * 32, std::numeric_limits<float>::infinity());
      const double eps = 5.9604644775390625e-08;
      for (int k = 0; k < na; ++k) {
        const double cx = reach[k].cx - ctx->origin[0], cy = reach[k].cy - ctx->origin[1], cz = reach[k].cz - ctx->origin[2];
        const double l1 = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);
        const double Rr = reach[k].thr;   // already inflated fp64 reach (or +inf)
        double Rf = (Rr + 3.0 * eps * l1 + 1e-30) * (1.0 + 8.0 * eps);
        float rf = (float)Rf;
        if (!(rf >= Rf)) rf = std::nextafterf(rf, std::numeric_limits<float>::infinity());
        if (!std::isfinite(Rf) || !(l1 < 1e30)) rf = std::numeric_limits<float>::infinity();
        float *g = &tf[(size_t)(k / 8) * 32 + (size_t)((k % 8) / 2) * 8];
        const int h = k & 1;
        g[0 + h] = (float)cx; g[2 + h] = (float)cy; g[4 + h] = (float)cz; g[6 + h] = rf;
        if (!std::isfinite(rf)) { g[0 + h] = 0.f; g[2 + h] = 0.f; g[4 + h] = 0.f; }   // reach +inf: always evaluated
      }
      // padding entries keep centre = +inf and R = +inf would pass; give them R = 0
      for (int k = na; k < ng * 8; ++k) {
        float *g = &tf[(size_t)(k / 8) * 32 + (size_t)((k % 8) / 2) * 8];
        g[6 + (k & 1)] = 0.f;
      }
      RRTX_HIP(ctx, ctx->d_sph_reach_f.ensure(sizeof(float) * tf.size()));
      RRTX_HIP(ctx, hipMemcpy(ctx->d_sph_reach_f.p, tf.data(), sizeof(float) * tf.size(), hipMemcpyHostToDevice));
    }
    {
      std::vector<SampleSph> st((size_t)na);
      for (int k = 0; k < na; ++k) {
        st[k].cx = rec[k].cx; st[k].cy = rec[k].cy; st[k].cz = rec[k].cz;
        st[k].thr_in = thr_in[k];
        st[k].thr_pt = thr_point_clear(robot_radius, radius[k]);
        st[k].reach = reach[k].thr;
        st[k].pad0 = 0.0; st[k].pad1 = 0.0;
      }
      RRTX_HIP(ctx, ctx->d_sph_sample.ensure(sizeof(SampleSph) * na));
      RRTX_HIP(ctx, hipMemcpy(ctx->d_sph_sample.p, st.data(), sizeof(SampleSph) * na, hipMemcpyHostToDevice));
    }
    char *aux = ctx->d_sph_aux.as<char>();
    RRTX_HIP(ctx, hipMemcpy(aux, radius.data(), sizeof(double) * na, hipMemcpyHostToDevice));
    RRTX_HIP(ctx, hipMemcpy(aux + sizeof(double) * na, thr_in.data(), sizeof(double) * na,
                            hipMemcpyHostToDevice));
    RRTX_HIP(ctx, hipMemcpy(aux + sizeof(double) * 2 * na, orig.data(), sizeof(int32_t) * na,
                            hipMemcpyHostToDevice));
  }
  ctx->sph_dirty = false;
  ctx->sph_packed_rr = robot_radius;
  for (int k = 0; k < 3; ++k) ctx->sph_packed_origin[k] = ctx->origin[k];
  return RRTX_OK;
}

static const double *sph_radius_dev(rrtx_ctx *ctx) { return ctx->d_sph_aux.as<double>(); }
static const double *sph_thr_in_dev(rrtx_ctx *ctx) { return ctx->d_sph_aux.as<double>() + ctx->sph_n_active; }
static const int32_t *sph_orig_dev(rrtx_ctx *ctx) {
  return reinterpret_cast<const int32_t *>(ctx->d_sph_aux.as<double>() + 2 * (size_t)ctx->sph_n_active);
}

std::vector<int32_t> active_positions(const std::vector<uint8_t> &active) {
  std::vector<int32_t> o;
  for (int i = 0; i < (int)active.size(); ++i)
    if (active[i]) o.push_back(i);
  return o;
}

int sync_polygons(rrtx_ctx *ctx) {
  if (!ctx->poly_dirty) return RRTX_OK;
  const int m = (int)ctx->poly_active.size();
  std::vector<double> meta;
  std::vector<int32_t> off(1, 0), orig;
  std::vector<double> vxy, path, slope, bbox, ytab, pbox;
  std::vector<int32_t> path_off(1, 0);
  bool moving = false;
  for (int i = 0; i < m; ++i) {
    if (!ctx->poly_active[i]) continue;
    const int pr0 = ctx->poly_path_off[i], pr1 = ctx->poly_path_off[i + 1];
    if (ctx->poly_kind[i] == 6 || ctx->poly_kind[i] == 7) {
      // findTransformObsToTimeOfPoint reads path[1, :] unconditionally (R/DRRT_Q.jl:1373-1376)
      if (pr1 <= pr0) return fail(ctx, RRTX_E_STATE, "moving obstacle %d has no path (rrtx_polygon_paths_set)", i);
      moving = true;
    }
    path.insert(path.end(), ctx->poly_path.begin() + 3 * (size_t)pr0, ctx->poly_path.begin() + 3 * (size_t)pr1);
    {
      // where the obstacle's centre can be while it moves: the box of position + path[k, 1:2] over all rows, the very
      // sums the moving-obstacle edge test forms (R/DRRT.jl:1607-1608); a NaN makes that side NaN (nothing is disjoint
      // from such a box)
      const double pcx = ctx->poly_cr[3 * (size_t)i + 0], pcy = ctx->poly_cr[3 * (size_t)i + 1];
      double x0 = pcx, x1 = pcx, y0 = pcy, y1 = pcy;
      if (pr1 > pr0) { x0 = y0 = __builtin_inf(); x1 = y1 = -__builtin_inf(); }
      for (int k = pr0; k < pr1; ++k) {
        const double x = ctx->poly_path[3 * (size_t)k] + pcx, y = ctx->poly_path[3 * (size_t)k + 1] + pcy;
        if (!(x >= x0)) x0 = x;
        if (!(x <= x1)) x1 = x;
        if (!(y >= y0)) y0 = y;
        if (!(y <= y1)) y1 = y;
      }
      pbox.push_back(x0); pbox.push_back(x1); pbox.push_back(y0); pbox.push_back(y1);
    }
    path_off.push_back((int32_t)(path.size() / 3));
    meta.push_back(ctx->poly_cr[3 * (size_t)i + 0]);
    meta.push_back(ctx->poly_cr[3 * (size_t)i + 1]);
    meta.push_back(ctx->poly_cr[3 * (size_t)i + 2]);
    meta.push_back((double)ctx->poly_kind[i]);
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    for (int v = ctx->poly_off[i]; v < ctx->poly_off[i + 1]; ++v) {
      const double x_v = ctx->poly_vxy[2 * (size_t)v], y_v = ctx->poly_vxy[2 * (size_t)v + 1];
      // (a NaN vertex makes the box NaN-sided: every "outside the box" comparison is then false)
      if (!(x_v >= bx0)) bx0 = x_v;
      if (!(x_v <= bx1)) bx1 = x_v;
      if (!(y_v >= by0)) by0 = y_v;
      if (!(y_v <= by1)) by1 = y_v;
      if (ctx->poly_kind[i] == 3) ytab.push_back(y_v);
      vxy.push_back(ctx->poly_vxy[2 * (size_t)v]);
      vxy.push_back(ctx->poly_vxy[2 * (size_t)v + 1]);
      // slope of the side (previous vertex -> v) as segmentDistSqrd's second side test computes it (R/DRRT.jl:1178;
      // one correctly rounded division, the same on the host and on the device); read only where the side is not
      // "close to vertical", so a division by zero here is never looked at
      const int u = v > ctx->poly_off[i] ? v - 1 : ctx->poly_off[i + 1] - 1;
      slope.push_back((ctx->poly_vxy[2 * (size_t)v + 1] - ctx->poly_vxy[2 * (size_t)u + 1]) /
                      (ctx->poly_vxy[2 * (size_t)v] - ctx->poly_vxy[2 * (size_t)u]));
    }
    off.push_back((int32_t)(vxy.size() / 2));
    orig.push_back(i);
    bbox.push_back(bx0); bbox.push_back(bx1); bbox.push_back(by0); bbox.push_back(by1);
  }
  // (NaN coordinates sort nowhere: a list with one has no table, and the point kernel takes its full loop)
  bool y_ok = true;
  for (double y : ytab) y_ok = y_ok && (y == y);
  if (y_ok) std::sort(ytab.begin(), ytab.end()); else ytab.clear();
  ctx->poly_n_ytab = y_ok ? (int)ytab.size() : -1;
  const int na = (int)orig.size();
  ctx->poly_n_active = na;
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (na > 0) {
    RRTX_HIP(ctx, ctx->d_poly_meta.ensure(sizeof(double) * meta.size()));
    RRTX_HIP(ctx, ctx->d_poly_off.ensure(sizeof(int32_t) * off.size()));
    RRTX_HIP(ctx, ctx->d_poly_vxy.ensure(sizeof(double) * (vxy.size() + 2)));
    RRTX_HIP(ctx, ctx->d_poly_orig.ensure(sizeof(int32_t) * na));
    RRTX_HIP(ctx, ctx->d_poly_slope.ensure(sizeof(double) * (slope.size() + 1)));
    if (!slope.empty())
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_slope.p, slope.data(), sizeof(double) * slope.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_meta.p, meta.data(), sizeof(double) * meta.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_off.p, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
    if (!vxy.empty())
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_vxy.p, vxy.data(), sizeof(double) * vxy.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_orig.p, orig.data(), sizeof(int32_t) * na, hipMemcpyHostToDevice));
    RRTX_HIP(ctx, ctx->d_poly_bbox.ensure(sizeof(double) * bbox.size()));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_bbox.p, bbox.data(), sizeof(double) * bbox.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, ctx->d_poly_pbox.ensure(sizeof(double) * pbox.size()));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_pbox.p, pbox.data(), sizeof(double) * pbox.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, ctx->d_poly_ytab.ensure(sizeof(double) * (ytab.size() + 1)));
    if (!ytab.empty())
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_ytab.p, ytab.data(), sizeof(double) * ytab.size(), hipMemcpyHostToDevice));
    RRTX_HIP(ctx, ctx->d_poly_path_off.ensure(sizeof(int32_t) * path_off.size()));
    RRTX_HIP(ctx, ctx->d_poly_path.ensure(sizeof(double) * (path.size() + 3)));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_path_off.p, path_off.data(), sizeof(int32_t) * path_off.size(), hipMemcpyHostToDevice));
    if (!path.empty())
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_path.p, path.data(), sizeof(double) * path.size(), hipMemcpyHostToDevice));
  }
  ctx->poly_has_moving = moving;
  ctx->poly_dirty = false;
  ctx->poly_h_meta.swap(meta);
  ctx->poly_h_bbox.swap(bbox);
  ctx->poly_grid_pad = -1.0;
  return RRTX_OK;
}

// Grid for points_polygons_flag_kernel.  A point in cell (ix, iy) -- ix = (int)((px - x0) * inv_wx), the expression the
// kernel evaluates, monotone in px -- can only be concerned with the obstacles listed for the cell: an obstacle is listed in
// every cell of ix(lo) .. ix(hi) x iy(lo) .. iy(hi), where [lo, hi] is its bounding circle's box padded by more than
// |robot radius| + list_r (by 1e-6 relative, against the kernel's 1e-9 slacks) joined with its vertex box.  For any other
// obstacle the point lies outside that box in x or in y: beyond reach for certain, outside the vertex box, and outside
// the bound of the per-sample lists -- the three things the kernel's walk over the whole list would find out one by one.
int sync_polygon_grid(rrtx_ctx *ctx, double pad_needed) {
  const int na = ctx->poly_n_active;
  // (a wider pad than needed only lengthens the cell lists; calls with and without the per-sample lists share one grid)
  if (ctx->poly_grid_pad >= pad_needed && ctx->poly_grid_pad <= 4.0 * pad_needed + 8.0) return RRTX_OK;
  ctx->poly_grid_pad = -1.0;
  if (na < 64 || na > 65535 || !(pad_needed >= 0.0) || !(pad_needed < 1e300)) return RRTX_OK;   // short lists: walked whole
  const double pad = pad_needed * 1.25;
  const std::vector<double> &meta = ctx->poly_h_meta, &bbox = ctx->poly_h_bbox;
  std::vector<double> box(4 * (size_t)na);
  double cmax = 0.0, gx0 = __builtin_inf(), gx1 = -__builtin_inf(), gy0 = __builtin_inf(), gy1 = -__builtin_inf();
  for (int j = 0; j < na; ++j)
    for (int k = 0; k < 3; ++k) cmax = std::max(cmax, std::fabs(meta[4 * (size_t)j + k]));
  for (int j = 0; j < na; ++j) {
    const double cx = meta[4 * (size_t)j], cy = meta[4 * (size_t)j + 1], rad = std::fabs(meta[4 * (size_t)j + 2]);
    const double h = (rad + pad) * (1.0 + 1e-6) + 1e-6 * (1.0 + 4.0 * cmax + pad);
    double lox = std::min(cx - h, bbox[4 * (size_t)j]), hix = std::max(cx + h, bbox[4 * (size_t)j + 1]);
    double loy = std::min(cy - h, bbox[4 * (size_t)j + 2]), hiy = std::max(cy + h, bbox[4 * (size_t)j + 3]);
    if (!(lox - lox == 0.0 && hix - hix == 0.0 && loy - loy == 0.0 && hiy - hiy == 0.0)) return RRTX_OK;   // not finite: no grid
    box[4 * (size_t)j] = lox; box[4 * (size_t)j + 1] = hix; box[4 * (size_t)j + 2] = loy; box[4 * (size_t)j + 3] = hiy;
    gx0 = std::min(gx0, lox); gx1 = std::max(gx1, hix); gy0 = std::min(gy0, loy); gy1 = std::max(gy1, hiy);
  }
  const int G = 64;
  if (!(gx1 > gx0) || !(gy1 > gy0)) return RRTX_OK;
  const double inv_wx = (double)G / (gx1 - gx0), inv_wy = (double)G / (gy1 - gy0);
  if (!(inv_wx - inv_wx == 0.0) || !(inv_wy - inv_wy == 0.0)) return RRTX_OK;
  auto cell = [](double v, double v0, double inv_w) {     // the kernel's expression, clamped
    const double f = (v - v0) * inv_w;
    return f < 0.0 ? 0 : (f >= (double)G ? G - 1 : (int)f);
  };
  std::vector<int32_t> start((size_t)G * G + 1, 0);
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<int32_t> cursor;
    std::vector<uint16_t> items;
    if (pass == 1) {
      for (size_t c = 0; c < (size_t)G * G; ++c) start[c + 1] += start[c];
      cursor.assign(start.begin(), start.end() - 1);
      items.resize((size_t)start[(size_t)G * G] + 1);
    }
    for (int j = 0; j < na; ++j) {        // ascending j within every cell
      const int ix0 = cell(box[4 * (size_t)j], gx0, inv_wx), ix1 = cell(box[4 * (size_t)j + 1], gx0, inv_wx);
      const int iy0 = cell(box[4 * (size_t)j + 2], gy0, inv_wy), iy1 = cell(box[4 * (size_t)j + 3], gy0, inv_wy);
      for (int iy = iy0; iy <= iy1; ++iy)
        for (int ix = ix0; ix <= ix1; ++ix) {
          if (pass == 0) ++start[(size_t)iy * G + ix + 1];
          else items[(size_t)cursor[(size_t)iy * G + ix]++] = (uint16_t)j;
        }
    }
    if (pass == 1) {
      RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      RRTX_HIP(ctx, ctx->d_poly_grid_start.ensure(sizeof(int32_t) * start.size()));
      RRTX_HIP(ctx, ctx->d_poly_grid_items.ensure(sizeof(uint16_t) * items.size()));
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_grid_start.p, start.data(), sizeof(int32_t) * start.size(), hipMemcpyHostToDevice));
      RRTX_HIP(ctx, hipMemcpy(ctx->d_poly_grid_items.p, items.data(), sizeof(uint16_t) * items.size(), hipMemcpyHostToDevice));
    }
  }
  ctx->poly_grid_x0 = gx0; ctx->poly_grid_y0 = gy0; ctx->poly_grid_inv_wx = inv_wx; ctx->poly_grid_inv_wy = inv_wy;
  ctx->poly_grid_g = G;
  ctx->poly_grid_pad = pad;
  return RRTX_OK;
}

static int zero_outputs(rrtx_ctx *ctx, int64_t ne, uint8_t *hit_dev, int32_t *first_hit_dev) {
  RRTX_HIP(ctx, hipMemsetAsync(hit_dev, 0, (size_t)ne, ctx->stream));
  if (first_hit_dev) RRTX_HIP(ctx, hipMemsetAsync(first_hit_dev, 0xff, (size_t)ne * sizeof(int32_t), ctx->stream));
  return RRTX_OK;
}

int launch_edges_spheres(rrtx_ctx *ctx, const double *p0_dev, const double *p1_dev, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end,
                         uint8_t *hit_dev, int32_t *first_hit_dev, const int32_t *sidx_dev,
                         const int32_t *eidx_dev, const uint8_t *mask_host) {
  if (ne <= 0) return RRTX_OK;
  int rc = sync_spheres(ctx, robot_radius);
  if (rc) return rc;
  const uint8_t *maskp = nullptr;
  if (mask_host) {
    // list-order mask -> packed (active-only) order, uploaded per call
    std::vector<int32_t> pos = active_positions(ctx->sph_active);
    std::vector<uint8_t> packed(pos.size() + 1, 0);
    for (size_t k = 0; k < pos.size(); ++k) packed[k] = mask_host[pos[k]] ? 1 : 0;
    RRTX_HIP(ctx, ctx->ws_mask.ensure(packed.size()));
    RRTX_HIP(ctx, hipMemcpyAsync(ctx->ws_mask.p, packed.data(), packed.size(), hipMemcpyHostToDevice, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `packed` is a local
    maskp = ctx->ws_mask.as<uint8_t>();
  }
  const int m = (int)ctx->sph_active.size();
  if (obstacle_or_minus1 >= m) return fail(ctx, RRTX_E_INVALID, "obstacle index %d out of range (%d spheres)", obstacle_or_minus1, m);
  if (obstacle_or_minus1 >= 0) { obs_begin = obstacle_or_minus1; obs_end = obstacle_or_minus1 + 1; }
  if (obs_end < 0 || obs_end > m) obs_end = m;
  if (obs_begin < 0) obs_begin = 0;
  int pb, pe;
  packed_range(active_positions(ctx->sph_active), obs_begin, obs_end, pb, pe);
  if (pe <= pb) return zero_outputs(ctx, ne, hit_dev, first_hit_dev);
  span_begin(ctx, KF_EDGES);
  hipLaunchKernelGGL(edges_spheres_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, p0_dev,
                     p1_dev, ctx->dim, sidx_dev, eidx_dev, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2], (long long)ne,
                     ctx->d_sph.as<SphRec>(), ctx->d_sph_reach.as<SphRec>(), maskp, sph_orig_dev(ctx), pb, pe, hit_dev,
                     first_hit_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_candidate_edges(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                           const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap, double robot_radius,
                           uint8_t *hit_out_dev, uint8_t *hit_in_dev, double r, uint8_t *sample_unsafe_dev) {
  // r >= 0: radius of the ball the lists were built with -> per-sample sphere lists; r < 0: full loop
  if (nq <= 0) return RRTX_OK;
  int rc = sync_spheres(ctx, robot_radius);
  if (rc) return rc;
  const int m = ctx->sph_n_active;
  const bool use_lists = r >= 0.0 && m > 0;
  const double r_bound = r * (1.0 + 1e-12);
  const int32_t *lists = nullptr, *list_n = nullptr;
  if (use_lists || sample_unsafe_dev) {
    RRTX_HIP(ctx, ctx->ws_sph_lists.ensure(sizeof(int32_t) * (size_t)nq * (kSphListCap + 1)));
    int32_t *l = ctx->ws_sph_lists.as<int32_t>();
    int32_t *ln = l + (size_t)nq * kSphListCap;
    span_begin(ctx, KF_POINTS);
    if (m > 0) {
      hipLaunchKernelGGL(sample_spheres_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64 * kSampleWaves), 0, ctx->stream,
                         q_dev, ctx->dim, (long long)nq, ctx->d_sph_sample.as<SampleSph>(), m,
                         use_lists ? r_bound : 0.0, sample_unsafe_dev, l, ln);
    } else if (sample_unsafe_dev) {
      RRTX_HIP(ctx, hipMemsetAsync(sample_unsafe_dev, 0, (size_t)nq, ctx->stream));
    }
    span_end(ctx);
    if (use_lists) { lists = l; list_n = ln; }
  }
  if (cap <= 0) return RRTX_OK;
  span_begin(ctx, KF_EDGES);
  // the grid covers the caller's capacity; lanes past offsets[nq] idle
  hipLaunchKernelGGL(candidate_edges_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, ctx->stream, q_dev,
                     ctx->dim, offsets_dev, nq, idx_dev, owner_dev, reinterpret_cast<const double4 *>(ctx->nodes_aos),
                     (int)ctx->n_nodes, (long long)cap, ctx->d_sph.as<SphRec>(), ctx->d_sph_reach_f.as<float>(), ctx->origin[0],
                     ctx->origin[1], ctx->origin[2], m, lists, list_n, kSphListCap, r_bound, hit_out_dev, hit_in_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// candidate edges of extend() against the polygon list (SimpleEdge in a space whose obstacles are
// Obstacle polygons): both directed edges of every CSR entry, plus explicitPointCheck of the samples
int launch_candidate_edges_polygons(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                                    const int32_t *idx_dev, const int32_t *owner_dev, int64_t cap,
                                    double robot_radius, uint8_t *hit_out_dev, uint8_t *hit_in_dev,
                                    uint8_t *sample_unsafe_dev, double r) {
  // r >= 0: radius of the ball the lists were built with -> per-sample obstacle lists from the sample pass
  if (nq <= 0) return RRTX_OK;
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  const double list_r = (r >= 0.0 && r - r == 0.0) ? r * (1.0 + 1e-8) : -1.0;
  const unsigned short *near_lists = nullptr, *near_cnt = nullptr;
  if (sample_unsafe_dev || (list_r >= 0.0 && cap > 0)) {
    if (ctx->poly_n_active > 0) {
      rc = launch_points_polygons(ctx, q_dev, nq, robot_radius, sample_unsafe_dev, nullptr, list_r, &near_lists, &near_cnt);
      if (rc) return rc;
    } else if (sample_unsafe_dev) {
      RRTX_HIP(ctx, hipMemsetAsync(sample_unsafe_dev, 0, (size_t)nq, ctx->stream));
    }
  }
  if (cap <= 0) return RRTX_OK;
  if (ctx->poly_n_active == 0) {
    RRTX_HIP(ctx, hipMemsetAsync(hit_out_dev, 0, (size_t)cap, ctx->stream));
    RRTX_HIP(ctx, hipMemsetAsync(hit_in_dev, 0, (size_t)cap, ctx->stream));
    return RRTX_OK;
  }
  PolyCsr csr;
  csr.q = q_dev; csr.offsets = offsets_dev; csr.idx = idx_dev; csr.owner = owner_dev;
  csr.nodes_aos = reinterpret_cast<const double4 *>(ctx->nodes_aos);
  csr.hit_in = hit_in_dev; csr.cap = (long long)cap; csr.nq = nq; csr.n_nodes = (int)ctx->n_nodes;
  span_begin(ctx, KF_EDGES);
  // (without a time column a wave's scratch is 8 KB: five waves per SIMD, with the register budget cut to match)
  // the last eighth of the entries in waves of 32 (measured: 1/8 0.1314, 2/8 0.1319, 3/8 0.1326, 4/8 0.1347, none 0.1345 ms)
  const int tail_eighths = 1;
  hipLaunchKernelGGL((ctx->poly_has_moving ? edges_polygons_kernel<true, 1, 4, true> : edges_polygons_kernel<true, 1, 5, false>),
                     dim3((unsigned)((cap + 63) / 64 + (tail_eighths ? (cap + 31) / 32 * tail_eighths / 8 + 2 : 0))), dim3(64), 0, ctx->stream,
                     (const double *)nullptr, (const double *)nullptr, ctx->dim, 0ll, csr, ctx->d_poly_meta.as<double>(),
                     ctx->d_poly_off.as<int32_t>(), ctx->d_poly_vxy.as<double>(), ctx->d_poly_slope.as<double>(),
                     ctx->d_poly_path_off.as<int32_t>(),
                     ctx->d_poly_path.as<double>(), ctx->poly_has_moving ? 1 : 0, ctx->d_poly_orig.as<int32_t>(), 0,
                     ctx->poly_n_active, robot_radius, hit_out_dev, (int32_t *)nullptr, near_lists, near_cnt, list_r, tail_eighths);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_edges_polygons(rrtx_ctx *ctx, const double *p0_dev, const double *p1_dev, int64_t ne,
                          double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end,
                          uint8_t *hit_dev, int32_t *first_hit_dev) {
  if (ne <= 0) return RRTX_OK;
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  const int m = (int)ctx->poly_active.size();
  if (obstacle_or_minus1 >= m) return fail(ctx, RRTX_E_INVALID, "obstacle index %d out of range (%d polygons)", obstacle_or_minus1, m);
  if (obstacle_or_minus1 >= 0) { obs_begin = obstacle_or_minus1; obs_end = obstacle_or_minus1 + 1; }
  if (obs_end < 0 || obs_end > m) obs_end = m;
  if (obs_begin < 0) obs_begin = 0;
  int pb, pe;
  packed_range(active_positions(ctx->poly_active), obs_begin, obs_end, pb, pe);
  if (pe <= pb) return zero_outputs(ctx, ne, hit_dev, first_hit_dev);
  span_begin(ctx, KF_EDGES);
  hipLaunchKernelGGL((edges_polygons_kernel<false, 4, 4, true>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, p0_dev,
                     p1_dev, ctx->dim, (long long)ne, PolyCsr{}, ctx->d_poly_meta.as<double>(), ctx->d_poly_off.as<int32_t>(),
                     ctx->d_poly_vxy.as<double>(), ctx->d_poly_slope.as<double>(), ctx->d_poly_path_off.as<int32_t>(),
                     ctx->d_poly_path.as<double>(),
                     ctx->poly_has_moving ? 1 : 0, ctx->d_poly_orig.as<int32_t>(), pb, pe, robot_radius, hit_dev,
                     first_hit_dev, (const unsigned short *)nullptr, (const unsigned short *)nullptr, 0.0, 0);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_points_spheres(rrtx_ctx *ctx, const double *p_dev, int64_t np, double robot_radius, int quick,
                          uint8_t *unsafe_dev, double *clearance_dev) {
  if (np <= 0) return RRTX_OK;
  int rc = sync_spheres(ctx, robot_radius);
  if (rc) return rc;
  span_begin(ctx, KF_POINTS);
  hipLaunchKernelGGL(points_spheres_kernel, dim3((unsigned)((np * kPtLanes + 255) / 256)), dim3(256), 0, ctx->stream, p_dev,
                     ctx->dim, (long long)np, ctx->d_sph.as<SphRec>(), sph_radius_dev(ctx), sph_thr_in_dev(ctx),
                     ctx->sph_n_active, robot_radius, quick, unsafe_dev, clearance_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_points_polygons(rrtx_ctx *ctx, const double *p_dev, int64_t np, double robot_radius,
                           uint8_t *unsafe_dev, double *clearance_dev, double list_r,
                           const unsigned short **lists_out, const unsigned short **list_cnt_out) {
  if (lists_out) *lists_out = nullptr;
  if (list_cnt_out) *list_cnt_out = nullptr;
  if (np <= 0) return RRTX_OK;
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  unsigned short *lists = nullptr, *list_cnt = nullptr;
  // flag-only calls over obstacles that stand still: two points per wave with the reference's loop as the fall-back
  // inside the kernel; a wanted certificate or obstacles that move in time: the reference's loop, one point per wave
  const bool flag_only = clearance_dev == nullptr && !ctx->poly_has_moving && ctx->poly_n_ytab >= 0 &&
                         ctx->d_poly_bbox.as<double>() != nullptr;
  if (!flag_only && unsafe_dev == nullptr && clearance_dev == nullptr) return RRTX_OK;   // (a call for the lists alone)
  span_begin(ctx, KF_POINTS);
  if (flag_only) {
    // list_r >= 0: also list, per point, the obstacles an edge of at most that length from the point can reach (the
    // fused extend preamble hands them to edges_polygons_kernel)
    if (lists_out && list_cnt_out && list_r >= 0.0 && ctx->poly_n_active <= 65535) {
      RRTX_HIP(ctx, ctx->ws_poly_lists.ensure(sizeof(unsigned short) * (size_t)np * (kPolyListCap + 1)));
      lists = ctx->ws_poly_lists.as<unsigned short>();
      list_cnt = lists + (size_t)np * kPolyListCap;
    }
    PolyGrid grid = {0.0, 0.0, 0.0, 0.0, 0, nullptr, nullptr};
    if (np >= 256) {                            // (a handful of points: not worth a grid build)
      rc = sync_polygon_grid(ctx, std::fabs(robot_radius) + (lists ? list_r : 0.0));
      if (rc) { span_end(ctx); return rc; }
      if (ctx->poly_grid_pad >= 0.0)
        grid = PolyGrid{ctx->poly_grid_x0, ctx->poly_grid_y0, ctx->poly_grid_inv_wx, ctx->poly_grid_inv_wy, ctx->poly_grid_g,
                        ctx->d_poly_grid_start.as<int32_t>(), ctx->d_poly_grid_items.as<uint16_t>()};
    }
    hipLaunchKernelGGL(points_polygons_flag_kernel, dim3((unsigned)((np + 7) / 8)), dim3(256), 0, ctx->stream, p_dev,
                       ctx->dim, (long long)np, ctx->d_poly_meta.as<double>(), ctx->d_poly_off.as<int32_t>(),
                       ctx->d_poly_vxy.as<double>(), ctx->poly_n_active, robot_radius, unsafe_dev,
                       ctx->d_poly_bbox.as<double>(), ctx->d_poly_ytab.as<double>(), ctx->poly_n_ytab, list_r, lists,
                       list_cnt, grid);
    if (lists) { *lists_out = lists; *list_cnt_out = list_cnt; }
  } else
    hipLaunchKernelGGL(points_polygons_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, ctx->stream, p_dev,
                       ctx->dim, (long long)np, ctx->d_poly_meta.as<double>(), ctx->d_poly_off.as<int32_t>(),
                       ctx->d_poly_vxy.as<double>(), ctx->d_poly_path_off.as<int32_t>(), ctx->d_poly_path.as<double>(),
                       ctx->poly_has_moving ? 1 : 0, ctx->poly_n_active, robot_radius, unsafe_dev, clearance_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_pack_hits(rrtx_ctx *ctx, const uint8_t *hit_out, const uint8_t *hit_in, const int64_t *n_valid_dev,
                     int64_t cap, uint64_t *words) {
  if (cap <= 0) return RRTX_OK;
  const long long n_words = (2 * (long long)cap + 63) / 64;
  span_begin(ctx, KF_EDGES);
  hipLaunchKernelGGL(pack_hits_kernel, dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, ctx->stream, hit_out,
                     hit_in, n_valid_dev, (long long)cap, n_words, reinterpret_cast<unsigned long long *>(words));
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_simple_steer(rrtx_ctx *ctx, const double *s_dev, const double *g_dev, int64_t ne, double *dist_dev,
                        double *wdist_dev) {
  if (ne <= 0) return RRTX_OK;
  span_begin(ctx, KF_DUBINS);
  hipLaunchKernelGGL(simple_steer_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, s_dev,
                     g_dev, ctx->dim, (long long)ne, dist_dev, wdist_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

}  // namespace rrtx
