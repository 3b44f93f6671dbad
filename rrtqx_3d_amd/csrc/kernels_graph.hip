// kernels_graph.hip -- device-resident cost propagation over the mirror of the planner's directed
// edges.  RRT^X keeps, for every node v, rrtLMC(v) = min over its out-edges v -> u
// of rrtLMC(u) + edge.dist, rooted at rrtLMC(root) = 0; rewire / reduceInconsistency /
// propogateDescendants (R/DRRT_Q.jl:2647-2817, with recalculateLMCMineVTwo :2490-2541) drive the values
// to that fixed point one heap pop at a time.  With changeThresh = 0 and the queue run dry the fixed
// point does not depend on the order of the pops: every value is the minimum over a node's out-edges
// of ONE floating-point addition on an already final value, a recurrence with a unique least solution
// for non-negative edge costs (Dijkstra's argument carries over to rounded sums: fl(a + w) >= a).  That
// solution is what this file computes, by label correcting: a pass relaxes the in-edges of every node
// whose value improved in the pass before (atomicMin on the bit pattern of the cost, which orders like
// the value for non-negative doubles), until a pass changes nothing.
//
//   full solve    every node at Inf, the root at 0, passes from the root outwards.
//   update        from the state of the previous solve (kept on the device): the reference's
//                 propogateDescendants becomes "every node whose path to the root along parent edges
//                 crosses an edge whose cost was touched is an orphan" (pointer jumping over the parent
//                 forest); orphans go back to Inf, then one pass over the touched / new edges and the
//                 out-edges of orphans, then passes over the frontier as above.  Nodes that are not
//                 orphans keep values that are still attained and still minimal, so the update ends in
//                 the same state as a full solve (tests compare the two bit for bit).
//
// Passes walk a CSR of in-edges (edge ids grouped by end node); edges appended since the CSR was built
// form a tail that every pass scans edge by edge, and the CSR is rebuilt once the tail is large.
// Blocked edges (dist = Inf: what explicitEdgeCheck / validMove / addNewObstacle decided) never relax.
// rrtParentEdge(v) = the lowest edge id that attains the minimum (the reference's choice among equal
// sums is its visiting order; equal sums are measure-zero).  A positive changeThresh makes the
// reference's result depend on the order of its heap pops; that variant stays on the host.  gfx950 only.
#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {

namespace {

constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;
constexpr int32_t kNoParent = 0x7fffffff;
constexpr int kGroupMax = 32;        // most passes launched between two reads of the "changed" flags
constexpr int kFlagInts = 64;        // [0, kGroupMax): pass flags, 40: orphan roots seen, 41: a parent edge of cost 0, 42: seed pass
constexpr int kFlagOrphans = 40, kFlagZeroCost = 41, kFlagSeed = 42;

__device__ __forceinline__ bool cost_ok(double w) { return w >= 0.0 && w < __builtin_inf(); }   // NaN, < 0, Inf: never relaxes

__device__ __forceinline__ bool relax(unsigned long long *lmc, int *stamp, int v, unsigned long long du, double w, int pass) {
  if (du >= kInfBits || !cost_ok(w)) return false;
  const double cand = __longlong_as_double((long long)du) + w;      // rrtLMC(u) + edge.dist, one rounded sum
  const unsigned long long cb = (unsigned long long)__double_as_longlong(cand);
  // values only fall during a solve, so a plain (possibly stale) read that is already <= the candidate settles it;
  // most candidates end here and never reach the atomic
  if (cb >= lmc[v]) return false;
  const unsigned long long old = atomicMin(&lmc[v], cb);
  if (cb < old) { stamp[v] = pass; return true; }
  return false;
}

// SimpleEdge cost of mirrored edges [first, first + n): edge.dist = dist(start, end) over all coordinates
// (calculateTrajectory, R/DRRT_SimpleEdge_functions.jl:177-181)
__global__ void graph_edge_dist_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee, long long first,
                                       long long n, int dim, const double4 *__restrict__ naos, double *__restrict__ dist,
                                       uint8_t *__restrict__ dirty) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double4 a = naos[es[first + i]], b = naos[ee[first + i]];
  dist[first + i] = sqrt_rn(dim == 4 ? sq4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w) : sq3(a.x, a.y, a.z, b.x, b.y, b.z));
  dirty[first + i] = 0;
}

__global__ void graph_block_kernel(const int32_t *__restrict__ ids, long long n, double *__restrict__ dist,
                                   uint8_t *__restrict__ dirty, const int32_t *__restrict__ in_pos, long long in_ne,
                                   double *__restrict__ in_w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t e = ids[i];
  dist[e] = __builtin_inf();
  dirty[e] = 1;
  if (e < in_ne) in_w[in_pos[e]] = __builtin_inf();
}

// costs of edges [first, first + n) were overwritten: mark them, refresh the copies the CSR holds
__global__ void graph_touch_kernel(long long first, long long n, const double *__restrict__ dist, uint8_t *__restrict__ dirty,
                                   const int32_t *__restrict__ in_pos, long long in_ne, double *__restrict__ in_w) {
  const long long e = first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= first + n) return;
  dirty[e] = 1;
  if (e < in_ne) in_w[in_pos[e]] = dist[e];
}

// ---- CSR of in-edges ------------------------------------------------------------------------------------------
__global__ void csr_count_kernel(const int32_t *__restrict__ ee, long long ne, int *__restrict__ cnt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) atomicAdd(&cnt[ee[e]], 1);
}

constexpr int kScanTile = 2048;      // counts per workgroup of 256 in the two tiled scan kernels

__global__ __launch_bounds__(256) void csr_tile_sum_kernel(const int *__restrict__ cnt, int n, int *__restrict__ tile_sum) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * kScanTile;
  int local = 0;
  for (int k = threadIdx.x; k < kScanTile; k += 256) local += (base + k < n) ? cnt[base + k] : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the tile sums by one workgroup of 1024 (in place); out[n] = total
__global__ __launch_bounds__(1024) void csr_scan_tiles_kernel(int *__restrict__ v, int n) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int b = min(t * per, n), e = min(b + per, n);
  int local = 0;
  for (int i = b; i < e; ++i) local += v[i];
  int x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(x, off);
    if (lane >= off) x += o;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int prefix = x - local;
  for (int w = 0; w < wave; ++w) prefix += wsum[w];
  for (int i = b; i < e; ++i) {
    const int c = v[i];
    v[i] = prefix;
    prefix += c;
  }
  if (t == 1023) v[n] = prefix;
}

// start = exclusive scan of cnt (tile offset + scan inside the tile, 8 consecutive counts per lane); cursor = start
__global__ __launch_bounds__(256) void csr_tile_scan_kernel(const int *__restrict__ cnt, int n, const int *__restrict__ tile_off,
                                                            int n_tiles, int *__restrict__ start, int *__restrict__ cursor) {
  __shared__ int wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int base = blockIdx.x * kScanTile + t * 8;
  int c[8], local = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k] = (base + k < n) ? cnt[base + k] : 0;
    local += c[k];
  }
  int x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(x, off);
    if (lane >= off) x += o;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int prefix = tile_off[blockIdx.x] + x - local;
  for (int w = 0; w < wave; ++w) prefix += wsum[w];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) { start[base + k] = prefix; cursor[base + k] = prefix; }
    prefix += c[k];
  }
  if (blockIdx.x == n_tiles - 1 && t == 255) start[n] = tile_off[n_tiles];
}

// edge ids grouped by end node, with copies of the start node and the cost in the same order (a pass then streams
// them instead of gathering by edge id) and the slot of every edge (to refresh the cost copy when a cost changes)
__global__ void csr_fill_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee, const double *__restrict__ dist,
                                long long ne, int *__restrict__ cursor, int32_t *__restrict__ in_src, double *__restrict__ in_w,
                                int32_t *__restrict__ in_pos) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int pos = atomicAdd(&cursor[ee[e]], 1);
  in_src[pos] = es[e];
  in_w[pos] = dist[e];
  in_pos[e] = pos;
}

// ---- node state -----------------------------------------------------------------------------------------------
struct NodeState {
  unsigned long long *lmc;
  int32_t *parent;
  int *stamp;
};

// full solve: every node at Inf but the root
__global__ void graph_init_kernel(NodeState s, int n, int root, int *__restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    s.lmc[i] = (i == root) ? 0ull : kInfBits;
    s.stamp[i] = (i == root) ? 0 : -2;
    s.parent[i] = kNoParent;
  }
  if (i < kFlagInts) flags[i] = 0;
}

// update: nodes added since the last solve start at Inf; with touched edges, set up the parent forest for the
// orphan search (anc = the node the parent edge ends at, or the node itself)
__global__ void graph_resume_kernel(NodeState s, int n, int solved_nodes, const int32_t *__restrict__ ee, uint8_t *__restrict__ orph,
                                    int32_t *__restrict__ anc, int *__restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    if (i >= solved_nodes) { s.lmc[i] = kInfBits; s.parent[i] = kNoParent; }
    s.stamp[i] = -2;
    if (orph) {
      const int32_t p = s.parent[i];
      orph[i] = 0;
      anc[i] = p == kNoParent ? i : ee[p];
    }
  }
  if (i < kFlagInts) flags[i] = 0;
}

// a touched edge that is its start node's parent edge makes that node the root of an orphaned subtree
// (addNewObstacle: verifyInOSQueue(edge.startNode), R/DRRT_Q.jl:3252-3262)
__global__ void graph_orphan_roots_kernel(const int32_t *__restrict__ es, const uint8_t *__restrict__ dirty, long long ne,
                                          const int32_t *__restrict__ parent, uint8_t *__restrict__ orph, int *__restrict__ any) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool hit = false;
  if (e < ne && dirty[e]) {
    const int v = es[e];
    if (parent[v] == (int32_t)e) { orph[v] = 1; hit = true; }
  }
  if (__ballot(hit) != 0ull && (threadIdx.x & 63) == 0) *any = 1;
}

// one round of pointer jumping over the parent forest (propogateDescendants' walk over successor lists,
// R/DRRT_Q.jl:2760-2817): orphan(i) |= orphan(anc(i)), anc(i) = anc(anc(i))
__global__ void graph_orphan_jump_kernel(const uint8_t *__restrict__ o_in, const int32_t *__restrict__ a_in,
                                         uint8_t *__restrict__ o_out, int32_t *__restrict__ a_out, int n, int *__restrict__ changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool ch = false;
  if (i < n) {
    const int32_t a = a_in[i];
    const uint8_t o = o_in[i] | o_in[a];
    const int32_t aa = a_in[a];
    o_out[i] = o;
    a_out[i] = aa;
    ch = o != o_in[i] || (aa != a && !o);            // an orphan's ancestors no longer matter
  }
  if (__ballot(ch) != 0ull && (threadIdx.x & 63) == 0) *changed = 1;
}

__global__ void graph_orphan_reset_kernel(NodeState s, const uint8_t *__restrict__ orph, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && orph[i]) { s.lmc[i] = kInfBits; s.parent[i] = kNoParent; }
}

// update, first pass: edges [lo, ne) that are new (>= solved_edges), touched, or leave an orphan
__global__ void graph_seed_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee, const double *__restrict__ dist,
                                  uint8_t *__restrict__ dirty, const uint8_t *__restrict__ orph, long long lo, long long ne,
                                  long long solved_edges, NodeState s, int *__restrict__ changed) {
  const long long e = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool any = false;
  if (e < ne) {
    const int v = es[e];
    const bool d = dirty[e] != 0;
    if (d) dirty[e] = 0;
    if (e >= solved_edges || d || (orph && orph[v])) any = relax(s.lmc, s.stamp, v, s.lmc[ee[e]], dist[e], 1);
  }
  if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) *changed = 1;
}

// one pass.  Blocks [0, node_blocks): 256 nodes each; the in-edges of the nodes that improved in the pass before are
// dealt out over the whole workgroup (prefix sum of their degrees, a lane per (node, in-edge) pair), so a pass costs
// what its frontier costs and no lane waits on another node's list.  Blocks beyond: the tail of edges the CSR does
// not hold, a lane per edge.
__global__ __launch_bounds__(256) void graph_pass_kernel(const int *__restrict__ in_start, const int32_t *__restrict__ in_src,
                                                         const double *__restrict__ in_w, int in_nn, long long in_ne,
                                                         const int32_t *__restrict__ es,
                                                         const int32_t *__restrict__ ee, const double *__restrict__ dist,
                                                         long long ne, int n, int node_blocks, NodeState s, int pass,
                                                         int *__restrict__ changed) {
  __shared__ int pre[257];
  __shared__ int sb0[256];
  __shared__ unsigned long long sdu[256];
  __shared__ int wsum[4];
  bool any = false;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if ((int)blockIdx.x < node_blocks) {
    const int i = blockIdx.x * 256 + t;
    const bool active = i < in_nn && s.stamp[i] == pass - 1;
    if (!__syncthreads_or(active)) return;
    int b0 = 0, deg = 0;
    unsigned long long du = kInfBits;
    if (active) {
      du = s.lmc[i];
      b0 = in_start[i];
      deg = in_start[i + 1] - b0;
    }
    int v = deg;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(v, off);
      if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    sb0[t] = b0;
    sdu[t] = du;
    __syncthreads();
    for (int w = 0; w < wave; ++w) v += wsum[w];
    pre[t + 1] = v;
    if (t == 0) pre[0] = 0;
    __syncthreads();
    const int total = pre[256];
    for (int j = t; j < total; j += 256) {
      int lo = 0, hi = 256;                        // pre[lo] <= j < pre[hi]
#pragma unroll
      for (int step = 0; step < 8; ++step) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= j) lo = mid; else hi = mid;
      }
      const int k = sb0[lo] + (j - pre[lo]);
      any |= relax(s.lmc, s.stamp, in_src[k], sdu[lo], in_w[k], pass);
    }
  } else {
    const long long e = in_ne + (long long)(blockIdx.x - node_blocks) * 256 + t;
    if (e < ne) {
      const int u = ee[e];
      if (s.stamp[u] == pass - 1) any = relax(s.lmc, s.stamp, es[e], s.lmc[u], dist[e], pass);
    }
  }
  if (__ballot(any) != 0ull && lane == 0) *changed = 1;
}

__global__ void graph_parent_clear_kernel(int32_t *__restrict__ parent, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = kNoParent;
}

// rrtParentEdge: the lowest edge id whose sum attains the node's value
__global__ void graph_parent_kernel(const int32_t *__restrict__ es, const int32_t *__restrict__ ee,
                                    const double *__restrict__ dist, long long ne, int root, NodeState s, int *__restrict__ zero_cost) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int v = es[e], u = ee[e];
  if (v == root) return;
  const unsigned long long dv = s.lmc[v], du = s.lmc[u];
  const double w = dist[e];
  if (dv >= kInfBits || du >= kInfBits || !cost_ok(w)) return;
  const double cand = __longlong_as_double((long long)du) + w;
  if ((unsigned long long)__double_as_longlong(cand) == dv) {
    atomicMin(&s.parent[v], (int32_t)e);
    if (du == dv) *zero_cost = 1;      // parents along edges that add nothing need not form a forest: the next update solves in full
  }
}

__global__ void graph_out_kernel(NodeState s, int n, double *__restrict__ lmc_out, int32_t *__restrict__ parent_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lmc_out[i] = __longlong_as_double((long long)s.lmc[i]);
  if (parent_out) parent_out[i] = s.parent[i] == kNoParent ? -1 : s.parent[i];
}

// grow a buffer whose first keep bytes must survive
hipError_t ensure_keep(DevBuf &b, size_t need, size_t keep, hipStream_t st) {
  if (need <= b.bytes) return hipSuccess;
  DevBuf nb;
  hipError_t e = nb.ensure(need);
  if (e != hipSuccess) return e;
  if (keep > 0 && b.p) {
    e = hipMemcpyAsync(nb.p, b.p, keep, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { nb.release(); return e; }
  }
  b.release();
  b = nb;
  return hipSuccess;
}

inline dim3 grid_for(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

int read_flags(rrtx_ctx *ctx, int *host) {
  RRTX_HIP(ctx, hipMemcpyAsync(host, ctx->gc.flags.p, sizeof(int) * kFlagInts, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int build_in_csr(rrtx_ctx *ctx, int n, long long ne) {
  GraphCost &gc = ctx->gc;
  hipStream_t st = ctx->stream;
  const int n_tiles = (n + kScanTile - 1) / kScanTile;
  RRTX_HIP(ctx, gc.in_cnt.ensure(sizeof(int) * (size_t)(n + 1)));
  RRTX_HIP(ctx, gc.in_start.ensure(sizeof(int) * (size_t)(n + 1)));
  RRTX_HIP(ctx, gc.in_cursor.ensure(sizeof(int) * (size_t)(n + 1)));
  RRTX_HIP(ctx, gc.in_tiles.ensure(sizeof(int) * (size_t)(n_tiles + 1)));
  RRTX_HIP(ctx, gc.in_src.ensure(sizeof(int32_t) * (size_t)(ne > 0 ? ne : 1)));
  RRTX_HIP(ctx, gc.in_w.ensure(sizeof(double) * (size_t)(ne > 0 ? ne : 1)));
  // in_pos is indexed by edge id and patched by set_dist / block between builds: sized like the edge arrays
  RRTX_HIP(ctx, gc.in_pos.ensure(sizeof(int32_t) * (size_t)(ctx->ge_cap > 0 ? ctx->ge_cap : 1)));
  RRTX_HIP(ctx, hipMemsetAsync(gc.in_cnt.p, 0, sizeof(int) * (size_t)(n + 1), st));
  if (ne > 0) hipLaunchKernelGGL(csr_count_kernel, grid_for(ne), dim3(256), 0, st, ctx->ge_end, ne, gc.in_cnt.as<int>());
  hipLaunchKernelGGL(csr_tile_sum_kernel, dim3(n_tiles), dim3(256), 0, st, gc.in_cnt.as<int>(), n, gc.in_tiles.as<int>());
  hipLaunchKernelGGL(csr_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, gc.in_tiles.as<int>(), n_tiles);
  hipLaunchKernelGGL(csr_tile_scan_kernel, dim3(n_tiles), dim3(256), 0, st, gc.in_cnt.as<int>(), n, gc.in_tiles.as<int>(), n_tiles,
                     gc.in_start.as<int>(), gc.in_cursor.as<int>());
  if (ne > 0)
    hipLaunchKernelGGL(csr_fill_kernel, grid_for(ne), dim3(256), 0, st, ctx->ge_start, ctx->ge_end, ctx->ge_dist, ne,
                       gc.in_cursor.as<int>(), gc.in_src.as<int32_t>(), gc.in_w.as<double>(), gc.in_pos.as<int32_t>());
  RRTX_HIP(ctx, hipGetLastError());
  gc.in_nn = n;
  gc.in_ne = ne;
  return RRTX_OK;
}

}  // namespace

int launch_graph_edge_dist(rrtx_ctx *ctx, long long first, long long n) {
  if (n <= 0) return RRTX_OK;
  hipLaunchKernelGGL(graph_edge_dist_kernel, grid_for(n), dim3(256), 0, ctx->stream, ctx->ge_start, ctx->ge_end, first, n,
                     ctx->dim, reinterpret_cast<const double4 *>(ctx->nodes_aos), ctx->ge_dist, ctx->ge_dirty);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// ids (host, validated by the caller): dist = Inf
int launch_graph_block(rrtx_ctx *ctx, const int32_t *ids_host, long long n) {
  if (n <= 0) return RRTX_OK;
  RRTX_HIP(ctx, ctx->gc.ids.ensure(sizeof(int32_t) * (size_t)n));
  RRTX_HIP(ctx, hipMemcpyAsync(ctx->gc.ids.p, ids_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(graph_block_kernel, grid_for(n), dim3(256), 0, ctx->stream, ctx->gc.ids.as<int32_t>(), n, ctx->ge_dist,
                     ctx->ge_dirty, ctx->gc.in_pos.as<int32_t>(), (long long)ctx->gc.in_ne, ctx->gc.in_w.as<double>());
  RRTX_HIP(ctx, hipGetLastError());
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));      // ids_host is the caller's
  ctx->gc.touched_old = true;
  return RRTX_OK;
}

// edge costs [first, first + n) were overwritten on the stream (rrtx_graph_edges_set_dist)
int launch_graph_touch(rrtx_ctx *ctx, long long first, long long n) {
  if (n <= 0) return RRTX_OK;
  hipLaunchKernelGGL(graph_touch_kernel, grid_for(n), dim3(256), 0, ctx->stream, first, n, ctx->ge_dist, ctx->ge_dirty,
                     ctx->gc.in_pos.as<int32_t>(), (long long)ctx->gc.in_ne, ctx->gc.in_w.as<double>());
  RRTX_HIP(ctx, hipGetLastError());
  ctx->gc.touched_old = true;
  return RRTX_OK;
}

// lmc_dev: n_nodes doubles, parent_dev: n_nodes int32 (may be null).  update = continue from the previous solve when
// there is one for this root.
static int graph_cost_impl(rrtx_ctx *ctx, int root, bool update, double *lmc_dev, int32_t *parent_dev, int *passes_out);

// A solve that fails part-way (a HIP error, a round or pass limit reached) has already rewritten the node state: the
// record of the last solve is dropped, so the next call -- update or not -- starts from scratch instead of resuming from
// a state no fixed point stands behind.
int launch_graph_cost(rrtx_ctx *ctx, int root, bool update, double *lmc_dev, int32_t *parent_dev, int *passes_out) {
  const int rc = graph_cost_impl(ctx, root, update, lmc_dev, parent_dev, passes_out);
  if (rc != RRTX_OK) {
    if (ctx->span_open) span_end(ctx);
    graph_cost_forget(ctx);
  }
  return rc;
}

static int graph_cost_impl(rrtx_ctx *ctx, int root, bool update, double *lmc_dev, int32_t *parent_dev, int *passes_out) {
  GraphCost &gc = ctx->gc;
  const int n = (int)ctx->n_nodes;
  const long long ne = ctx->ge_n;
  hipStream_t st = ctx->stream;
  const size_t keep = (size_t)gc.solved_nodes;
  RRTX_HIP(ctx, ensure_keep(gc.lmc, sizeof(unsigned long long) * (size_t)n, sizeof(unsigned long long) * keep, st));
  RRTX_HIP(ctx, ensure_keep(gc.parent, sizeof(int32_t) * (size_t)n, sizeof(int32_t) * keep, st));
  RRTX_HIP(ctx, gc.stamp.ensure(sizeof(int) * (size_t)n));
  RRTX_HIP(ctx, gc.flags.ensure(sizeof(int) * kFlagInts));
  NodeState s{gc.lmc.as<unsigned long long>(), gc.parent.as<int32_t>(), gc.stamp.as<int>()};
  int *flags = gc.flags.as<int>();
  int host_flags[kFlagInts];

  bool resume = update && gc.solved_root == root && gc.solved_nodes > 0 && gc.solved_edges <= ne;
  if (resume) {
    // the previous solve's "a parent edge of cost 0" flag decides whether its forest can be trusted
    int rc = read_flags(ctx, host_flags);
    if (rc) return rc;
    if (host_flags[kFlagZeroCost]) resume = false;
  }
  span_begin(ctx, KF_EDGES);
  // the CSR of in-edges: rebuilt for a full solve, or when the tail of newer edges has grown large
  const long long tail = ne - gc.in_ne;
  if (gc.in_ne > ne || (!resume && (tail > 0 || gc.in_nn == 0)) || tail > 262144 + gc.in_ne / 8) {
    int rc = build_in_csr(ctx, n, ne);
    if (rc) return rc;
  }
  int total = 0;
  if (!resume) {
    hipLaunchKernelGGL(graph_init_kernel, grid_for(n > kFlagInts ? n : kFlagInts), dim3(256), 0, st, s, n, root, flags);
  } else {
    const bool touched = gc.touched_old;
    uint8_t *orph = nullptr;
    if (touched) {
      RRTX_HIP(ctx, gc.orph.ensure(2 * (size_t)n));
      RRTX_HIP(ctx, gc.anc.ensure(2 * sizeof(int32_t) * (size_t)n));
      orph = gc.orph.as<uint8_t>();
    }
    int32_t *anc = touched ? gc.anc.as<int32_t>() : nullptr;
    hipLaunchKernelGGL(graph_resume_kernel, grid_for(n > kFlagInts ? n : kFlagInts), dim3(256), 0, st, s, n, (int)gc.solved_nodes,
                       ctx->ge_end, orph, anc, flags);
    if (touched) {
      if (gc.solved_edges > 0)
        hipLaunchKernelGGL(graph_orphan_roots_kernel, grid_for(gc.solved_edges), dim3(256), 0, st, ctx->ge_start, ctx->ge_dirty,
                           (long long)gc.solved_edges, s.parent, orph, flags + kFlagOrphans);
      int rc = read_flags(ctx, host_flags);
      if (rc) return rc;
      if (host_flags[kFlagOrphans]) {
        // pointer jumping doubles the reach every round; 40 rounds bound the work whatever the forest looks like
        int cur = 0;
        bool settled = false;
        for (int round = 0; round < 40;) {
          RRTX_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(int) * 4, st));
          for (int k = 0; k < 4; ++k, ++round) {
            hipLaunchKernelGGL(graph_orphan_jump_kernel, grid_for(n), dim3(256), 0, st, orph + (size_t)cur * n, anc + (size_t)cur * n,
                               orph + (size_t)(1 - cur) * n, anc + (size_t)(1 - cur) * n, n, flags + k);
            cur = 1 - cur;
          }
          rc = read_flags(ctx, host_flags);
          if (rc) return rc;
          if (!host_flags[3]) { settled = true; break; }
        }
        if (!settled)
          return fail(ctx, RRTX_E_STATE, "graph_cost_update: the orphan marking did not settle in 40 rounds of pointer jumping "
                                         "(a parent forest deeper than 2^40 cannot exist: the parent edges do not form a forest)");
        orph += (size_t)cur * n;
        hipLaunchKernelGGL(graph_orphan_reset_kernel, grid_for(n), dim3(256), 0, st, s, orph, n);
      } else {
        orph = nullptr;                      // touched edges, none of them a parent edge
      }
    }
    const long long lo = touched ? 0 : gc.solved_edges;
    if (ne > lo)
      hipLaunchKernelGGL(graph_seed_kernel, grid_for(ne - lo), dim3(256), 0, st, ctx->ge_start, ctx->ge_end, ctx->ge_dist,
                         ctx->ge_dirty, orph, lo, ne, (long long)gc.solved_edges, s, flags + kFlagSeed);
    total = 1;
  }
  // passes over the frontier
  const int node_blocks = (gc.in_nn + 255) / 256;
  const long long tail_now = ne - gc.in_ne;
  const dim3 pgrid((unsigned)(node_blocks + (tail_now + 255) / 256));
  int pass = resume ? 2 : 1;
  int group = 8;                                 // 8, 16, 32, 32, ...: a pass over an empty frontier costs little
  bool fixed_point = !(ne > 0 && pgrid.x > 0);
  while (!fixed_point && total < (1 << 22)) {
    RRTX_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(int) * kGroupMax, st));
    for (int k = 0; k < group; ++k)
      hipLaunchKernelGGL(graph_pass_kernel, pgrid, dim3(256), 0, st, gc.in_start.as<int>(), gc.in_src.as<int32_t>(), gc.in_w.as<double>(),
                         gc.in_nn, (long long)gc.in_ne, ctx->ge_start, ctx->ge_end, ctx->ge_dist, ne, n, node_blocks, s, pass + k, flags + k);
    int rc = read_flags(ctx, host_flags);
    if (rc) return rc;
    pass += group;
    total += group;
    if (!host_flags[group - 1]) { fixed_point = true; break; }   // the last pass of the group changed nothing
    if (group < kGroupMax) group *= 2;
  }
  if (!fixed_point)
    return fail(ctx, RRTX_E_STATE, "graph_cost: no fixed point after %d relaxation passes (negative or NaN edge costs?)", total);
  // parent edges, from scratch (so that an update and a full solve agree on ties too)
  RRTX_HIP(ctx, hipMemsetAsync(flags + kFlagZeroCost, 0, sizeof(int), st));
  hipLaunchKernelGGL(graph_parent_clear_kernel, grid_for(n), dim3(256), 0, st, s.parent, n);
  if (ne > 0)
    hipLaunchKernelGGL(graph_parent_kernel, grid_for(ne), dim3(256), 0, st, ctx->ge_start, ctx->ge_end, ctx->ge_dist, ne, root, s,
                       flags + kFlagZeroCost);
  hipLaunchKernelGGL(graph_out_kernel, grid_for(n), dim3(256), 0, st, s, n, lmc_dev, parent_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  if (!resume || gc.touched_old) {
    // a full solve does not look at the touched flags: clear them for the next update
    if (!resume && gc.touched_old && ne > 0) RRTX_HIP(ctx, hipMemsetAsync(ctx->ge_dirty, 0, (size_t)ne, st));
  }
  gc.touched_old = false;
  gc.solved_root = root;
  gc.solved_nodes = n;
  gc.solved_edges = ne;
  if (passes_out) *passes_out = total;
  return RRTX_OK;
}

void graph_cost_forget(rrtx_ctx *ctx) {
  ctx->gc.solved_root = -1;
  ctx->gc.solved_nodes = 0;
  ctx->gc.solved_edges = 0;
  ctx->gc.in_ne = 0;
  ctx->gc.in_nn = 0;
  ctx->gc.touched_old = false;
}

}  // namespace rrtx
