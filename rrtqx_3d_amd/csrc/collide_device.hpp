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

// Per-sample sphere lists of the fused extend path (sample_spheres_kernel / nn_finish_kernel):
// everything the sample pass needs about one active sphere, one 64-byte record:
//   thr_in  : quickCheck, inside the sphere            <=> !(s >= thr_in)   (R/DRRT_Q.jl:1410)
//   thr_pt  : (sqrt(s) - robotRadius) - radius < 0     <=>  s < thr_pt      (thr_point_clear)
//   reach   : inflated robotRadius + radius (or +inf)
struct alignas(64) SampleSph { double cx, cy, cz, thr_in, thr_pt, reach, pad0, pad1; };
constexpr int kSphListCap = 8;   // spheres listed per sample; a longer list sends the sample's edges to the full loop

}  // namespace
}  // namespace rrtx
