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

}  // namespace rrtx
