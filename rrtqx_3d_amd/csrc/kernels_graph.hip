// kernels_graph.hip -- device-resident cost propagation over the mirror of the planner's directed
// edges.  RRT^X keeps, for every node v, rrtLMC(v) = min over its out-edges v -> u
// of rrtLMC(u) + edge.dist, rooted at rrtLMC(root) = 0; rewire / reduceInconsistency /
// propogateDescendants (R/DRRT_Q.jl:2647-2817, with recalculateLMCMineVTwo :2490-2541) drive the values
// to that fixed point one heap pop at a time.  With changeThresh = 0 and the queue run dry the fixed
// point does not depend on the order of the pops: every value is the minimum over a node's out-edges
// of ONE floating-point addition on an already final value, a recurrence with a unique least solution
// for non-negative edge costs (Dijkstra's argument carries over to rounded sums: fl(a + w) >= a).  That
// solution is what this file computes, by label correcting: every pass relaxes every edge whose end
// node improved in the pass before (atomicMin on the bit pattern of the cost, which orders like the
// value for non-negative doubles), until a pass changes nothing.  Blocked edges (dist = Inf: what
// explicitEdgeCheck / validMove / addNewObstacle decided) never relax, so orphaned subtrees come out at
// Inf or re-attached through their best remaining neighbour -- the state propogateDescendants followed
// by reduceInconsistency reaches.  rrtParentEdge(v) = the lowest edge id that attains the minimum (the
// reference's choice among equal sums is its visiting order; equal sums are measure-zero).
// A positive changeThresh makes the reference's result depend on the order of its heap pops; that
// epsilon-consistent variant stays on the host.  gfx950 only.
#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {

namespace {

constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;

// SimpleEdge cost of mirrored edges [first, first + n): edge.dist = dist(start, end) over all coordinates
// (calculateTrajectory, R/DRRT_SimpleEdge_functions.jl:177-181)
__global__ void graph_edge_dist_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee, long long first,
                                       long long n, int dim, const double4 *__restrict__ naos, double *__restrict__ dist) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 a = naos[es[first + i]], b = naos[ee[first + i]];
  dist[first + i] = sqrt_rn(dim == 4 ? sq4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w) : sq3(a.x, a.y, a.z, b.x, b.y, b.z));
}

__global__ void graph_init_kernel(unsigned long long *__restrict__ lmc, int *__restrict__ stamp, int32_t *__restrict__ parent,
                                  int n, int root, int *__restrict__ changed, int n_flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    lmc[i] = (i == root) ? 0ull : kInfBits;
    stamp[i] = (i == root) ? 0 : -2;
    parent[i] = 0x7fffffff;
  }
  if (i < n_flags) changed[i] = 0;
}

// one pass: edge v -> u relaxes v when u improved in the previous pass (stamp[u] == pass - 1)
__global__ void graph_relax_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee,
                                   const double *__restrict__ dist, long long ne, int n_nodes,
                                   unsigned long long *__restrict__ lmc, int *__restrict__ stamp, int pass,
                                   int *__restrict__ changed /* this pass's flag */) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool any = false;
  if (e < ne) {
    const int v = es[e], u = ee[e];
    if ((unsigned)v < (unsigned)n_nodes && (unsigned)u < (unsigned)n_nodes && stamp[u] == pass - 1) {
      const double w = dist[e];
      const unsigned long long du = lmc[u];
      if (du < kInfBits && w >= 0.0 && w < __builtin_inf()) {          // (NaN and negative costs never relax)
        const double cand = __longlong_as_double((long long)du) + w;   // rrtLMC(u) + edge.dist, one rounded sum
        const unsigned long long cb = (unsigned long long)__double_as_longlong(cand);
        const unsigned long long old = atomicMin(&lmc[v], cb);
        if (cb < old) { stamp[v] = pass; any = true; }
      }
    }
  }
  if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) *changed = 1;
}

// rrtParentEdge: the lowest edge id whose sum attains the node's value
__global__ void graph_parent_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee,
                                    const double *__restrict__ dist, long long ne, int n_nodes, int root,
                                    const unsigned long long *__restrict__ lmc, int32_t *__restrict__ parent) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int v = es[e], u = ee[e];
  if ((unsigned)v >= (unsigned)n_nodes || (unsigned)u >= (unsigned)n_nodes || v == root) return;
  const unsigned long long dv = lmc[v], du = lmc[u];
  const double w = dist[e];
  if (dv >= kInfBits || du >= kInfBits || !(w >= 0.0 && w < __builtin_inf())) return;
  const double cand = __longlong_as_double((long long)du) + w;
  if ((unsigned long long)__double_as_longlong(cand) == dv) atomicMin(&parent[v], (int32_t)e);
}

__global__ void graph_out_kernel(const unsigned long long *__restrict__ lmc, const int32_t *__restrict__ parent, int n,
                                 double *__restrict__ lmc_out, int32_t *__restrict__ parent_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lmc_out[i] = __longlong_as_double((long long)lmc[i]);
  if (parent_out) parent_out[i] = parent[i] == 0x7fffffff ? -1 : parent[i];
}

}  // namespace

int launch_graph_edge_dist(rrtx_ctx *ctx, long long first, long long n) {
  if (n <= 0) return RRTX_OK;
  hipLaunchKernelGGL(graph_edge_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->ge_start,
                     ctx->ge_end, first, n, ctx->dim, reinterpret_cast<const double4 *>(ctx->nodes_aos), ctx->ge_dist);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// lmc_dev: n_nodes doubles, parent_dev: n_nodes int32 (may be null).  Runs passes in groups of kGroup between
// reads of the "changed" flags (one small D2H copy and stream sync per group).
int launch_graph_cost_to_root(rrtx_ctx *ctx, int root, double *lmc_dev, int32_t *parent_dev, int *passes_out) {
  const int n = (int)ctx->n_nodes;
  const long long ne = ctx->ge_n;
  constexpr int kGroup = 16, kMaxPass = 1 << 20;
  RRTX_HIP(ctx, ctx->ws_graph_lmc.ensure(sizeof(unsigned long long) * (size_t)n));
  RRTX_HIP(ctx, ctx->ws_graph_stamp.ensure(sizeof(int) * (size_t)n));
  RRTX_HIP(ctx, ctx->ws_graph_parent.ensure(sizeof(int32_t) * (size_t)n));
  RRTX_HIP(ctx, ctx->ws_graph_flags.ensure(sizeof(int) * (size_t)(kGroup + 2)));
  unsigned long long *lmc = ctx->ws_graph_lmc.as<unsigned long long>();
  int *stamp = ctx->ws_graph_stamp.as<int>();
  int32_t *parent = ctx->ws_graph_parent.as<int32_t>();
  int *flags = ctx->ws_graph_flags.as<int>();
  hipStream_t st = ctx->stream;
  span_begin(ctx, KF_EDGES);
  hipLaunchKernelGGL(graph_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lmc, stamp, parent, n, root,
                     flags, kGroup + 2);
  int pass = 1, total = 0;
  const dim3 egrid((unsigned)((ne + 255) / 256)), eblock(256);
  while (ne > 0 && pass < kMaxPass) {
    // flags[k] = pass (base + k) changed something; stamps carry absolute pass numbers
    RRTX_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(int) * (kGroup + 2), st));
    for (int k = 0; k < kGroup; ++k)
      hipLaunchKernelGGL(graph_relax_kernel, egrid, eblock, 0, st, ctx->ge_start, ctx->ge_end, ctx->ge_dist, ne, n, lmc,
                         stamp, pass + k, flags + k);
    int host_flags[kGroup + 2];
    RRTX_HIP(ctx, hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, st));
    RRTX_HIP(ctx, hipStreamSynchronize(st));
    pass += kGroup;
    total += kGroup;
    if (!host_flags[kGroup - 1]) break;           // the last pass of the group changed nothing: fixed point
  }
  if (ne > 0 && parent_dev)
    hipLaunchKernelGGL(graph_parent_kernel, egrid, eblock, 0, st, ctx->ge_start, ctx->ge_end, ctx->ge_dist, ne, n, root, lmc,
                       parent);
  hipLaunchKernelGGL(graph_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lmc, parent, n, lmc_dev,
                     parent_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  if (passes_out) *passes_out = total;
  return RRTX_OK;
}

}  // namespace rrtx
