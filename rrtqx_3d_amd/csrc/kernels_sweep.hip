// kernels_sweep.hip -- the edge loop of addNewObstacle (R/DRRT_Q.jl:3195-3290) against a device
// mirror of the planner's directed edges: nodes within the search range of the obstacle centre
// (findPointsInConflictWithObstacle = kdFindWithinRange with the root's <=), every mirrored edge
// that starts at such a node, explicitEdgeCheck(S, edge, ob) against that one obstacle.
// Returns the ids of the colliding edges in ascending order.  gfx950 only.
#include "collide_device.hpp"

namespace rrtx {

namespace {

// nodes in range of the obstacle centre: dist < range, the root with <= (thresholds on the squared
// distance, see rrtx_sq_thresholds)
__global__ void sweep_mark_kernel(const double *__restrict__ nx, const double *__restrict__ ny,
                                  const double *__restrict__ nz, int n, double cx, double cy, double cz,
                                  double thr_lt, double thr_gt, uint8_t *__restrict__ mark) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double s = sq3(cx, cy, cz, nx[i], ny[i], nz[i]);
  mark[i] = ((s < thr_lt) || (i == 0 && s < thr_gt)) ? 1 : 0;
}

constexpr int kSweepBlock = 1024;

// flag[e] = edge e starts at a marked node and collides with the obstacle; per-block counts
__global__ __launch_bounds__(kSweepBlock) void sweep_edges_kernel(
    const int32_t *__restrict__ e_start, const int32_t *__restrict__ e_end, long long ne, int n_nodes,
    const uint8_t *__restrict__ mark, const double *__restrict__ naos, SphRec ob, int active,
    uint8_t *__restrict__ flag, int *__restrict__ block_count) {
  __shared__ int wcnt[kSweepBlock / 64];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool hit = false;
  if (e < ne) {
    const int a = e_start[e], b = e_end[e];
    if ((unsigned)a < (unsigned)n_nodes && (unsigned)b < (unsigned)n_nodes && mark[a] && active) {
      const double4 p0 = reinterpret_cast<const double4 *>(naos)[a];
      const double4 p1 = reinterpret_cast<const double4 *>(naos)[b];
      const double len = sqrt_rn(sq3(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z));
      hit = edge_hits_sphere(p0.x, p0.y, p0.z, p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, len, ob);
    }
    flag[e] = hit ? 1 : 0;
  }
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < kSweepBlock / 64; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
}

// exclusive scan of the block counts by one workgroup of 1024; out[n] = total
__global__ __launch_bounds__(1024) void sweep_scan_kernel(const int *__restrict__ in, long long *__restrict__ out, int n) {
  __shared__ long long wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int b = min(t * per, n), e = min(b + per, n);
  long long local = 0;
  for (int i = b; i < e; ++i) local += in[i];
  long long v = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  long long prefix = v - local;
  for (int w = 0; w < wave; ++w) prefix += wsum[w];
  for (int i = b; i < e; ++i) {
    out[i] = prefix;
    prefix += in[i];
  }
  if (t == 1023) out[n] = prefix;
}

// flagged edge ids, ascending: block offset + rank inside the block
__global__ __launch_bounds__(kSweepBlock) void sweep_write_kernel(const uint8_t *__restrict__ flag, long long ne,
                                                                  const long long *__restrict__ block_start,
                                                                  int32_t *__restrict__ out, long long cap) {
  __shared__ int wcnt[kSweepBlock / 64];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool hit = e < ne && flag[e] != 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(hit);
  if (lane == 0) wcnt[wave] = __popcll(m);
  __syncthreads();
  if (!hit) return;
  long long pos = block_start[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wcnt[w];
  if (pos < cap) out[pos] = (int32_t)e;
}

}  // namespace

// device side of rrtx_obstacle_sweep; needed_dev[0] = colliding edges, needed_dev[1] = block count scratch
int launch_obstacle_sweep(rrtx_ctx *ctx, const double centre[3], double thr_lt, double thr_gt, const SphRec &ob,
                          int active, int32_t *out_dev, int64_t cap, long long **total_dev) {
  const int n = (int)ctx->n_nodes;
  const long long ne = ctx->ge_n;
  const int nb = (int)((ne + kSweepBlock - 1) / kSweepBlock);
  RRTX_HIP(ctx, ctx->ws_sweep_mark.ensure((size_t)n));
  RRTX_HIP(ctx, ctx->ws_sweep_flag.ensure((size_t)(ne > 0 ? ne : 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_cnt.ensure(sizeof(int) * (size_t)(nb + 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_start.ensure(sizeof(long long) * (size_t)(nb + 2)));
  hipStream_t st = ctx->stream;
  span_begin(ctx, KF_EDGES);
  hipLaunchKernelGGL(sweep_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                     n, centre[0], centre[1], centre[2], thr_lt, thr_gt, ctx->ws_sweep_mark.as<uint8_t>());
  if (nb > 0) {
    hipLaunchKernelGGL(sweep_edges_kernel, dim3(nb), dim3(kSweepBlock), 0, st, ctx->ge_start, ctx->ge_end, ne, n,
                       ctx->ws_sweep_mark.as<uint8_t>(), ctx->nodes_aos, ob, active, ctx->ws_sweep_flag.as<uint8_t>(),
                       ctx->ws_sweep_cnt.as<int>());
  }
  hipLaunchKernelGGL(sweep_scan_kernel, dim3(1), dim3(1024), 0, st, ctx->ws_sweep_cnt.as<int>(),
                     ctx->ws_sweep_start.as<long long>(), nb);
  if (nb > 0 && cap > 0) {
    hipLaunchKernelGGL(sweep_write_kernel, dim3(nb), dim3(kSweepBlock), 0, st, ctx->ws_sweep_flag.as<uint8_t>(), ne,
                       ctx->ws_sweep_start.as<long long>(), out_dev, (long long)cap);
  }
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  *total_dev = ctx->ws_sweep_start.as<long long>() + nb;
  return RRTX_OK;
}

}  // namespace rrtx
