// exact_math.hpp -- device arithmetic that must be bit-identical to the Julia
// reference: IEEE fp64, no FMA contraction, the reference's operation order.
// Every translation unit including this header is compiled with
// -ffp-contract=off; the pragma below makes that explicit for clang.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace rrtx {

// squared euclidianDist: sum((x-y).^2) as Julia's left fold
// (R/DRRT_distance_functions.jl:37; Base reduce for n < 16)
__device__ __forceinline__ double sq3(double ax, double ay, double az, double bx, double by, double bz) {
  double dx = ax - bx, dy = ay - by, dz = az - bz;
  double s = dx * dx;
  s = s + dy * dy;
  s = s + dz * dz;
  return s;
}
__device__ __forceinline__ double sq4(double ax, double ay, double az, double aw, double bx, double by,
                                      double bz, double bw) {
  double dx = ax - bx, dy = ay - by, dz = az - bz, dw = aw - bw;
  double s = dx * dx;
  s = s + dy * dy;
  s = s + dz * dz;
  s = s + dw * dw;
  return s;
}
__device__ __forceinline__ double sq2(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  double s = dx * dx;
  s = s + dy * dy;
  return s;
}

// Julia Base.min/max on Float64: NaN-propagating, -0.0 < +0.0 (base/math.jl)
__device__ __forceinline__ double jl_min(double x, double y) {
  bool sx = __builtin_signbit(x), sy = __builtin_signbit(y);
  if ((y < x) || (sy && !sx)) return (x != x) ? x : y;
  return (y != y) ? y : x;
}
__device__ __forceinline__ double jl_max(double x, double y) {
  bool sx = __builtin_signbit(x), sy = __builtin_signbit(y);
  if ((y > x) || (!sy && sx)) return (x != x) ? x : y;
  return (y != y) ? y : x;
}
// max(0, min(1, v)) with Julia semantics (R/DRRT_Q.jl:1208): NaN stays NaN,
// -0.0 becomes +0.0
__device__ __forceinline__ double jl_clamp01(double v) {
  double c = (v <= 0.0) ? 0.0 : ((v >= 1.0) ? 1.0 : v);
  return (v != v) ? v : c;
}

// correctly rounded fp64 sqrt / divide: hipcc's default lowering is IEEE for
// f64 (checked on hardware by tests/test_gpu_parity.py::test_device_sqrt_div)
__device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }

// Order-preserving map double -> uint64 (for atomicMin / atomicMax on coordinates) and back.
// An empty minimum is ~0 and an empty maximum is 0; both decode to NaN, which fails every
// comparison, so an empty range never overlaps anything.
__device__ __forceinline__ unsigned long long enc_ord(double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_ord(unsigned long long e) {
  const unsigned long long u = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)u);
}

// Equal-width slab of coordinate x: monotone non-decreasing in x (every step is a monotone
// rounded operation), NaN -> slab 0.  inv_w == 0 (degenerate or non-finite extent) puts
// everything in slab 0.
__device__ __forceinline__ int slab_of(double x, double x0, double inv_w, int K) {
  const double v = (x - x0) * inv_w;
  return (v >= 0.0) ? ((v < (double)K) ? (int)v : K - 1) : 0;
}
// x and y extent of a group of nodes (enc_ord values; empty: lo = ~0, hi = 0)
struct ChunkExt { unsigned long long xlo, xhi, ylo, yhi; };

// Cell of (x, y) in a Kx x Ky grid laid over [x0, ..] x [y0, ..], numbered row by row in
// boustrophedon order (odd rows run backwards) so that consecutive cells are neighbours.
// Only used to ORDER nodes and query copies; no result depends on it.
__device__ __forceinline__ int cell_of(double x, double y, double x0, double inv_wx, int Kx, double y0,
                                       double inv_wy, int Ky) {
  const int cx = slab_of(x, x0, inv_wx, Kx), cy = slab_of(y, y0, inv_wy, Ky);
  return cy * Kx + ((cy & 1) ? (Kx - 1 - cx) : cx);
}

__device__ __forceinline__ void slab_map(unsigned long long lo_enc, unsigned long long hi_enc, int K, double *x0,
                                         double *inv_w) {
  const double lo = dec_ord(lo_enc), hi = dec_ord(hi_enc);
  const double w = hi - lo;
  double inv = (w > 0.0 && w < 1e300) ? (double)K / w : 0.0;
  if (!(inv > 0.0 && inv < 1e300)) inv = 0.0;
  *x0 = (lo == lo && lo > -1e300 && lo < 1e300) ? lo : 0.0;
  *inv_w = inv;
}

}  // namespace rrtx
