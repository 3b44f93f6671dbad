/*
 * rrtx_detmath.h -- ONE deterministic fp64 implementation of the transcendentals on the Dubins
 * decision paths (sin, cos, atan2, acos and the rows of a sampled arc), written once and compiled
 * into BOTH the HIP kernels (rrtqx_3d_amd/csrc/kernels_dubins.hip, hipcc --offload-arch=gfx950
 * -ffp-contract=off) and the CPU checker (oracle/rrtx_oracle.c, gcc -ffp-contract=off).
 *
 * Why: calculateTrajectory(S, ::DubinsEdge) (R/DRRT_DubinsEdge_functions.jl:348-501) and
 * rightTurnDist / leftTurnDist (R/DRRT_distance_functions.jl:62-80) decide on the LAST BIT of
 * sin / cos / atan / acos: `theta < 0 -> theta + 2 pi` turns an arc of length exactly 0 into a
 * full turn, the strict `bestDist > len` picks the first of tied words, and a polyline piece
 * grazes a polygon side or not.  Julia's libm, glibc's and ROCm's OCML differ in that bit, so a
 * device that calls OCML can never be compared exactly with a host that calls glibc.  With this
 * header the device and the checker evaluate the SAME sequence of IEEE operations and agree bit
 * for bit; the (un-pinnable) gap that remains is checker <-> Julia's libm, and it is bounded by
 * the accuracy of these routines (< 2 ulp against glibc, tests/test_detmath.py).
 *
 * Rules: only + - * / sqrt, comparisons and rint (all correctly rounded IEEE operations on both
 * targets), no fused multiply-add (both builds use -ffp-contract=off; the pragma below says it
 * again for clang), no table look-ups that depend on the target's memory model, no libm.
 * Polynomial coefficients and the Cody-Waite split of pi/2 are the classical fdlibm values
 * (public domain constants of Sun's freely distributable libm); the evaluation schemes here are
 * simpler than fdlibm's (one division per atan2, no bit manipulation).
 *
 * C and HIP compatible (C99 / C++17).
 */
#ifndef RRTX_DETMATH_H
#define RRTX_DETMATH_H

#include <math.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#define RRTX_DM_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define RRTX_DM_FN static inline
#endif

#define RRTX_DM_PI 3.141592653589793        /* Float64(pi) */
#define RRTX_DM_PI_LO 1.2246467991473531772e-16
#define RRTX_DM_PIO2 1.5707963267948966     /* Float64(pi / 2) */

/* ---- argument reduction: x = n * pi/2 + (r + rt), |r| <= pi/4 (+ a rounding), for |x| < 2^20 * pi/2.
 * pi/2 = P1 + P2 + P2t: P1 and P2 carry 33 bits each, so n * P1 and n * P2 are exact for |n| < 2^20.
 * Larger |x| are first folded by 2 pi in plain double arithmetic (the accuracy then degrades with the
 * magnitude -- headings are O(10)); inf / NaN give NaN through the arithmetic itself. */
RRTX_DM_FN int rrtx_dm_rem_pio2(double x, double *r, double *rt) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double p1 = 1.57079632673412561417e+00;
  const double p2 = 6.07710050630396597660e-11;
  const double p2t = 2.02226624879595063154e-21;
  if (fabs(x) >= 1.6e6) x = x - (2.0 * RRTX_DM_PI) * rint(x / (2.0 * RRTX_DM_PI));
  const double fn = rint(x * invpio2);
  const double r1 = x - fn * p1;
  const double w = fn * p2;
  const double r2 = r1 - w;
  const double wt = fn * p2t - ((r1 - r2) - w);
  const double y0 = r2 - wt;
  *r = y0;
  *rt = (r2 - y0) - wt;
  /* fn is integral and |fn| < 2^21 here (NaN: any quadrant, the result is NaN anyway) */
  return (fn == fn) ? ((int)fn & 3) : 0;
}

/* sin on [-pi/4, pi/4] with the tail of the reduced argument */
RRTX_DM_FN double rrtx_dm_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
/* cos on [-pi/4, pi/4] */
RRTX_DM_FN double rrtx_dm_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  return 1.0 - (0.5 * z - (z * r - x * y));
}

RRTX_DM_FN void rrtx_dm_sincos(double x, double *s, double *c) {
  double r, rt;
  const int n = rrtx_dm_rem_pio2(x, &r, &rt);
  const double ks = rrtx_dm_ksin(r, rt), kc = rrtx_dm_kcos(r, rt);
  const double ss = (n & 1) ? kc : ks;
  const double cc = (n & 1) ? ks : kc;
  *s = (n & 2) ? -ss : ss;
  *c = ((n + 1) & 2) ? -cc : cc;
}
RRTX_DM_FN double rrtx_dm_sin(double x) {
  double s, c;
  rrtx_dm_sincos(x, &s, &c);
  return s;
}
RRTX_DM_FN double rrtx_dm_cos(double x) {
  double s, c;
  rrtx_dm_sincos(x, &s, &c);
  return c;
}

/* atan(a / b) for a >= 0, b >= 0 finite, not both zero, with ONE division: the classical five ranges of
 * t = a / b (breakpoints 7/16, 11/16, 19/16, 39/16), the reduced argument formed from a and b directly.
 * Result in [0, pi/2]. */
RRTX_DM_FN double rrtx_dm_atan_ratio_core(double a, double b) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
               aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
               aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
  double num, den, hi, lo;
  int direct = 0;
  if (a < 0.4375 * b) { num = a; den = b; hi = 0.0; lo = 0.0; direct = 1; }
  else if (a < 0.6875 * b) { num = 2.0 * a - b; den = 2.0 * b + a; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (a < 1.1875 * b) { num = a - b; den = a + b; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (a < 2.4375 * b) { num = a - 1.5 * b; den = b + 1.5 * a; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else { num = -b; den = a; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  const double x = num / den;
  const double z = x * x;
  const double w = z * z;
  const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (direct) return x - x * (s1 + s2);
  return hi - ((x * (s1 + s2) - lo) - x);
}
RRTX_DM_FN double rrtx_dm_atan_ratio(double a, double b) {
  /* keep 2 b + a, a + 1.5 b ... away from overflow and the quotient's operands away from the subnormals */
  const double big = (a > b) ? a : b;
  if (big > 1e150) { a = a * 0x1p-600; b = b * 0x1p-600; }
  else if (big < 1e-150) { a = a * 0x1p600; b = b * 0x1p600; }
  return rrtx_dm_atan_ratio_core(a, b);
}

/* atan2(y, x) with the IEEE / C99 special cases (signed zeros, infinities, NaN) */
RRTX_DM_FN double rrtx_dm_atan2(double y, double x) {
  {
    /* the ordinary case first -- both operands non-zero, the larger magnitude between 1e-150 and 1e150 -- behind ONE
     * test (NaN fails it): exactly what the general sequence below computes for such operands (no special case
     * applies, no rescaling happens), without walking through its tests.  22 calls per steered Dubins edge. */
    const double fx = fabs(x), fy = fabs(y);
    const double mx = (fx > fy) ? fx : fy, mn = (fx > fy) ? fy : fx;
    if (mn > 0.0 && mx > 1e-150 && mx < 1e150) {
      double zf = rrtx_dm_atan_ratio_core(fy, fx);
      if (x < 0.0) zf = RRTX_DM_PI - (zf - RRTX_DM_PI_LO);
      return (y < 0.0) ? -zf : zf;
    }
  }
  if (x != x || y != y) return x + y;
  const int sy = __builtin_signbit(y) ? 1 : 0, sx = __builtin_signbit(x) ? 1 : 0;
  const double ax = fabs(x), ay = fabs(y);
  const double inf = (double)INFINITY;
  double z;
  if (ay == 0.0) {
    z = sx ? RRTX_DM_PI : 0.0;
    return sy ? -z : z;
  }
  if (ax == 0.0) return sy ? -RRTX_DM_PIO2 : RRTX_DM_PIO2;
  if (ax == inf) {
    if (ay == inf) z = sx ? 3.0 * (RRTX_DM_PI / 4.0) : RRTX_DM_PI / 4.0;
    else z = sx ? RRTX_DM_PI : 0.0;
    return sy ? -z : z;
  }
  if (ay == inf) return sy ? -RRTX_DM_PIO2 : RRTX_DM_PIO2;
  z = rrtx_dm_atan_ratio(ay, ax);
  if (sx) z = RRTX_DM_PI - (z - RRTX_DM_PI_LO);
  return sy ? -z : z;
}

/* acos(x) = atan2(sqrt((1 - x)(1 + x)), x): 1 - x and 1 + x carry the argument's distance from +-1 without
 * cancellation; |x| > 1 and NaN give NaN. */
RRTX_DM_FN double rrtx_dm_acos(double x) {
  if (!(fabs(x) <= 1.0)) return (x - x) / (x - x);
  return rrtx_dm_atan2(sqrt((1.0 - x) * (1.0 + x)), x);
}

/* ---- rows of a sampled arc.  Row k of collect(phi_start : -+0.1 : phi_end) is at phi_start -+ k * 0.1
 * (R/DRRT_DubinsEdge_functions.jl:529-532 and siblings: x = cx + r cos(phi), y = cy + r sin(phi)); its
 * cos / sin are ONE angle addition on the arc's own cos / sin of phi_start with cos(k * 0.1), sin(k * 0.1)
 * from this table (an arc spans less than 2 pi: at most 63 rows; beyond the table the same two numbers come
 * from rrtx_dm_sincos).  This IS the definition of an arc row on both targets; it differs from
 * sin(phi_start -+ k * 0.1) evaluated directly by an ulp or two of the coordinate -- as any two libms do. */
#define RRTX_DM_ARC_TAB 72
#define RRTX_DM_ARC_COS_INIT {1.0, 0.9950041652780258, 0.9800665778412416, 0.955336489125606, 0.9210609940028851, 0.8775825618903728, 0.8253356149096782, 0.7648421872844884, 0.6967067093471654, 0.6216099682706644, 0.5403023058681398, 0.4535961214255773, 0.3623577544766734, 0.26749882862458735, 0.16996714290024081, 0.0707372016677029, -0.029199522301288815, -0.12884449429552486, -0.2272020946930871, -0.3232895668635036, -0.4161468365471424, -0.5048461045998576, -0.5885011172553458, -0.6662760212798244, -0.7373937155412458, -0.8011436155469337, -0.8568887533689473, -0.9040721420170612, -0.9422223406686583, -0.9709581651495907, -0.9899924966004454, -0.9991351502732795, -0.9982947757947531, -0.9874797699088649, -0.9667981925794609, -0.9364566872907963, -0.896758416334147, -0.848100031710408, -0.7909677119144165, -0.7259323042001399, -0.6536436208636119, -0.5748239465332685, -0.4902608213406994, -0.40079917207997545, -0.30733286997841935, -0.2107957994307797, -0.11215252693505398, -0.01238866346289056, 0.08749898343944727, 0.18651236942257576, 0.28366218546322625, 0.37797774271298107, 0.4685166713003771, 0.5543743361791615, 0.6346928759426347, 0.70866977429126, 0.7755658785102502, 0.8347127848391598, 0.8855195169413194, 0.9274784307440359, 0.960170286650366, 0.9832684384425847, 0.9965420970232175, 0.9998586363834151, 0.9931849187581926, 0.9765876257280235, 0.9502325919585293, 0.9143831482353194, 0.8693974903498248, 0.8157251001253568, 0.7539022543433046, 0.6845466664428059}
#define RRTX_DM_ARC_SIN_INIT {0.0, 0.09983341664682815, 0.19866933079506122, 0.2955202066613396, 0.3894183423086505, 0.479425538604203, 0.5646424733950355, 0.6442176872376911, 0.7173560908995228, 0.7833269096274834, 0.8414709848078965, 0.8912073600614354, 0.9320390859672264, 0.963558185417193, 0.9854497299884603, 0.9974949866040544, 0.9995736030415051, 0.9916648104524686, 0.9738476308781951, 0.9463000876874145, 0.9092974268256817, 0.8632093666488737, 0.8084964038195901, 0.74570521217672, 0.6754631805511506, 0.5984721441039565, 0.5155013718214642, 0.4273798802338298, 0.33498815015590466, 0.23924932921398198, 0.1411200080598672, 0.04158066243329049, -0.058374143427580086, -0.15774569414324865, -0.25554110202683167, -0.35078322768961984, -0.44252044329485246, -0.5298361409084934, -0.6118578909427193, -0.6877661591839741, -0.7568024953079282, -0.8182771110644108, -0.8715757724135882, -0.9161659367494549, -0.951602073889516, -0.977530117665097, -0.9936910036334645, -0.9999232575641008, -0.9961646088358406, -0.9824526126243325, -0.9589242746631385, -0.9258146823277321, -0.8834546557201531, -0.8322674422239008, -0.7727644875559871, -0.7055403255703919, -0.6312666378723208, -0.5506855425976376, -0.4646021794137566, -0.373876664830236, -0.27941549819892586, -0.18216250427209502, -0.0830894028174964, 0.0168139004843506, 0.11654920485049364, 0.21511998808781552, 0.3115413635133787, 0.4048499206165983, 0.49411335113860894, 0.5784397643882001, 0.6569865987187891, 0.7289690401258765}

/* point of row k: (cx + r (a ck - b sk'), cy + r (b ck + a sk')), a = cos(phi_start), b = sin(phi_start),
 * (ck, sk) = cos / sin of k * 0.1, sk' = -sk for a right turn (step -0.1) */
RRTX_DM_FN void rrtx_dm_arc_row(double cx, double cy, double r, double a, double b, double ck, double sk, int right_turn,
                                double *x, double *y) {
  if (right_turn) sk = -sk;
  *x = cx + r * (a * ck - b * sk);
  *y = cy + r * (b * ck + a * sk);
}

#endif /* RRTX_DETMATH_H */
