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

// ---- polygon / Dubins space (R/DRRT.jl:3048-3290) ----
// findPointsInConflictWithObstacle there is one range query (static obstacle) or one per path segment (kinds 6 / 7,
// accumulated with kdFindMoreWithinRange), each with its ghosts in wrapped dimensions: a node is in the list when
// ANY of them finds it -- dist < range, the root with <= for the un-wrapped query points (kdTree_general.jl:896,
// 934) and < for the ghosts.  thr_root is the root's threshold (= thr_lt where the root gets no <=).
struct SweepQuery { double x, y, z, w, thr_lt, thr_root; };

__global__ void sweep_mark_multi_kernel(const double *__restrict__ nx, const double *__restrict__ ny,
                                        const double *__restrict__ nz, const double *__restrict__ nw, int n, int dim,
                                        const SweepQuery *__restrict__ qs, int nqs, uint8_t *__restrict__ mark) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = nx[i], y = ny[i], z = nz[i], w = (dim == 4) ? nw[i] : 0.0;
  bool in = false;
  for (int k = 0; k < nqs; ++k) {
    const SweepQuery q = qs[k];
    const double s = (dim == 4) ? sq4(q.x, q.y, q.z, q.w, x, y, z, w) : sq3(q.x, q.y, q.z, x, y, z);
    in = in || (s < q.thr_lt) || (i == 0 && s < q.thr_root);
  }
  mark[i] = in ? 1 : 0;
}

// flag[e] = edge e of the mirror starts at a marked node (and, blocked_only: edge.dist == Inf); per-block counts
__global__ __launch_bounds__(kSweepBlock) void sweep_select_kernel(const int32_t *__restrict__ e_start, long long ne,
                                                                   int n_nodes, const uint8_t *__restrict__ mark,
                                                                   const double *__restrict__ e_dist, int blocked_only,
                                                                   uint8_t *__restrict__ flag, int *__restrict__ block_count) {
  __shared__ int wcnt[kSweepBlock / 64];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool sel = false;
  if (e < ne) {
    const int a = e_start[e];
    sel = (unsigned)a < (unsigned)n_nodes && mark[a] != 0 && (!blocked_only || e_dist[e] == __builtin_inf());
    flag[e] = sel ? 1 : 0;
  }
  const unsigned long long m = __ballot(sel);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < kSweepBlock / 64; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
}

// flag[k] = hit[k] and none of the "other obstacle" results (may be null); per-block counts
__global__ __launch_bounds__(kSweepBlock) void sweep_combine_kernel(const uint8_t *__restrict__ hit, const uint8_t *__restrict__ o1,
                                                                    const uint8_t *__restrict__ o2, long long n,
                                                                    uint8_t *__restrict__ flag, int *__restrict__ block_count) {
  __shared__ int wcnt[kSweepBlock / 64];
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = false;
  if (k < n) {
    f = hit[k] != 0 && !(o1 && o1[k] != 0) && !(o2 && o2[k] != 0);
    flag[k] = f ? 1 : 0;
  }
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < kSweepBlock / 64; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
}

// start / end rows (dim doubles each) of the mirrored edges ids[k]
__global__ void sweep_gather_kernel(const int32_t *__restrict__ ids, long long n, const int32_t *__restrict__ es,
                                    const int32_t *__restrict__ ee, const double *__restrict__ naos, int dim,
                                    double *__restrict__ p0, double *__restrict__ p1) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int id = ids[k];
  const double4 a = reinterpret_cast<const double4 *>(naos)[es[id]], b = reinterpret_cast<const double4 *>(naos)[ee[id]];
  p0[dim * k] = a.x; p0[dim * k + 1] = a.y; p0[dim * k + 2] = a.z;
  p1[dim * k] = b.x; p1[dim * k + 1] = b.y; p1[dim * k + 2] = b.z;
  if (dim == 4) { p0[dim * k + 3] = a.w; p1[dim * k + 3] = b.w; }
}

// flagged positions, ascending: block offset + rank inside the block; ids (may be null) maps a position to what is written
__global__ __launch_bounds__(kSweepBlock) void sweep_write_kernel(const uint8_t *__restrict__ flag, long long ne,
                                                                  const long long *__restrict__ block_start,
                                                                  int32_t *__restrict__ out, long long cap,
                                                                  const int32_t *__restrict__ ids = nullptr) {
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
  if (pos < cap) out[pos] = ids ? ids[e] : (int32_t)e;
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

// ---- the polygon / Dubins sweep: compaction steps shared by rrtx_obstacle_sweep_polygon (rrtx_capi.hip) ----
int launch_sweep_mark_multi(rrtx_ctx *ctx, const void *queries_host, int nqs) {
  const int n = (int)ctx->n_nodes;
  RRTX_HIP(ctx, ctx->ws_sweep_mark.ensure((size_t)n));
  RRTX_HIP(ctx, ctx->ws_mask.ensure(sizeof(SweepQuery) * (size_t)nqs));
  RRTX_HIP(ctx, hipMemcpyAsync(ctx->ws_mask.p, queries_host, sizeof(SweepQuery) * (size_t)nqs, hipMemcpyHostToDevice, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (the caller's table is a local)
  hipLaunchKernelGGL(sweep_mark_multi_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->nodes[0], ctx->nodes[1],
                     ctx->nodes[2], ctx->nodes[3], n, ctx->dim, ctx->ws_mask.as<SweepQuery>(), nqs, ctx->ws_sweep_mark.as<uint8_t>());
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// positions with flag != 0 -> out (ids[position] when ids is given), ascending; *total_dev = their number
static int compact_flags(rrtx_ctx *ctx, long long n, const int32_t *ids_dev, int32_t *out_dev, long long cap, long long **total_dev) {
  const int nb = (int)((n + kSweepBlock - 1) / kSweepBlock);
  hipLaunchKernelGGL(sweep_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->ws_sweep_cnt.as<int>(),
                     ctx->ws_sweep_start.as<long long>(), nb);
  if (nb > 0 && cap > 0)
    hipLaunchKernelGGL(sweep_write_kernel, dim3(nb), dim3(kSweepBlock), 0, ctx->stream, ctx->ws_sweep_flag.as<uint8_t>(), n,
                       ctx->ws_sweep_start.as<long long>(), out_dev, cap, ids_dev);
  RRTX_HIP(ctx, hipGetLastError());
  *total_dev = ctx->ws_sweep_start.as<long long>() + nb;
  return RRTX_OK;
}

// mirrored edges that start at a marked node (blocked_only: and have dist == Inf) -> ascending ids in out_dev
int launch_sweep_select(rrtx_ctx *ctx, int blocked_only, int32_t *out_dev, long long cap, long long **total_dev) {
  const long long ne = ctx->ge_n;
  const int nb = (int)((ne + kSweepBlock - 1) / kSweepBlock);
  RRTX_HIP(ctx, ctx->ws_sweep_flag.ensure((size_t)(ne > 0 ? ne : 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_cnt.ensure(sizeof(int) * (size_t)(nb + 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_start.ensure(sizeof(long long) * (size_t)(nb + 2)));
  if (nb > 0)
    hipLaunchKernelGGL(sweep_select_kernel, dim3(nb), dim3(kSweepBlock), 0, ctx->stream, ctx->ge_start, ne, (int)ctx->n_nodes,
                       ctx->ws_sweep_mark.as<uint8_t>(), ctx->ge_dist, blocked_only, ctx->ws_sweep_flag.as<uint8_t>(),
                       ctx->ws_sweep_cnt.as<int>());
  return compact_flags(ctx, ne, nullptr, out_dev, cap, total_dev);
}

// of the n candidates ids_dev[k]: those with hit[k] and neither o1[k] nor o2[k] -> their ids, ascending
int launch_sweep_finish(rrtx_ctx *ctx, const int32_t *ids_dev, long long n, const uint8_t *hit, const uint8_t *o1,
                        const uint8_t *o2, int32_t *out_dev, long long cap, long long **total_dev) {
  const int nb = (int)((n + kSweepBlock - 1) / kSweepBlock);
  RRTX_HIP(ctx, ctx->ws_sweep_flag.ensure((size_t)(n > 0 ? n : 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_cnt.ensure(sizeof(int) * (size_t)(nb + 1)));
  RRTX_HIP(ctx, ctx->ws_sweep_start.ensure(sizeof(long long) * (size_t)(nb + 2)));
  if (nb > 0)
    hipLaunchKernelGGL(sweep_combine_kernel, dim3(nb), dim3(kSweepBlock), 0, ctx->stream, hit, o1, o2, n,
                       ctx->ws_sweep_flag.as<uint8_t>(), ctx->ws_sweep_cnt.as<int>());
  return compact_flags(ctx, n, ids_dev, out_dev, cap, total_dev);
}

int launch_sweep_gather(rrtx_ctx *ctx, const int32_t *ids_dev, long long n, double *p0_dev, double *p1_dev) {
  if (n <= 0) return RRTX_OK;
  hipLaunchKernelGGL(sweep_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ids_dev, n, ctx->ge_start,
                     ctx->ge_end, ctx->nodes_aos, ctx->dim, p0_dev, p1_dev);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

}  // namespace rrtx
