// kernels_slab.hip -- the slab index of the culled range search: a copy of the
// screen arrays ordered by (x, y) grid cell and bin of the third coordinate, with the first position of
// every (cell, bin) and per-chunk extents, rebuilt on the stream when enough nodes have been appended.
// gfx950 only.
#include "nn_device.hpp"

namespace rrtx {

namespace {

// ------------------------------------------------------------ slab index ------
// Rebuild of the slab-ordered shadow (rare: when enough nodes were appended since the last
// one).  A counting sort of the nodes by (cell, bin); the order inside a bin is the order in
// which the atomics happened to land, which no result depends on.
static_assert(kSlabChunk == kChunkF, "the culled scan visits one slab-index chunk per work unit");

// (SlabParams: rrtx_internal.hpp)

__global__ void slab_params_kernel(const unsigned long long *__restrict__ xrange, int Kx, int Ky, int Kz,
                                   SlabParams *__restrict__ sp, int *__restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    double x0, ix, y0, iy, z0, iz;
    slab_map(xrange[0], xrange[1], Kx, &x0, &ix);
    slab_map(xrange[2], xrange[3], Ky, &y0, &iy);
    slab_map(xrange[4], xrange[5], Kz, &z0, &iz);
    sp->x0 = x0; sp->inv_wx = ix; sp->y0 = y0; sp->inv_wy = iy; sp->z0 = z0; sp->inv_wz = iz;
    sp->Kx = Kx; sp->Ky = Ky; sp->Kz = Kz; sp->pad = 0;
  }
  for (int k = i; k <= Kx * Ky * Kz; k += gridDim.x * blockDim.x) hist[k] = 0;
}

// sort key of a node: (x, y) cell, then the bin of the third coordinate inside the cell
__global__ void slab_rank_kernel(const double *__restrict__ nx, const double *__restrict__ ny,
                                 const double *__restrict__ nz, int n, const SlabParams *__restrict__ sp,
                                 int *__restrict__ hist, int2 *__restrict__ sr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int Kz = sp->Kz;
  const int b = cell_of(nx[i], ny[i], sp->x0, sp->inv_wx, sp->Kx, sp->y0, sp->inv_wy, sp->Ky) * Kz +
                slab_of(nz[i], sp->z0, sp->inv_wz, Kz);
  sr[i] = make_int2(b, atomicAdd(&hist[b], 1));
}

// exclusive scan of n ints by one workgroup of 1024; out[n] = total
__global__ __launch_bounds__(1024) void excl_scan_kernel(const int *__restrict__ in, int *__restrict__ out, int n) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int b = min(t * per, n), e = min(b + per, n);
  int local = 0;
  for (int i = b; i < e; ++i) local += in[i];
  int v = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  int prefix = v - local;
  for (int w = 0; w < wave; ++w) prefix += wsum[w];
  for (int i = b; i < e; ++i) {
    const int c = in[i];
    out[i] = prefix;
    prefix += c;
  }
  if (t == 1023) out[n] = prefix;
}

__global__ void slab_scatter_kernel(int n, const int2 *__restrict__ sr, const int *__restrict__ start,
                                    const float *__restrict__ fx, const float *__restrict__ fy,
                                    const float *__restrict__ fz, const float *__restrict__ fw,
                                    const float *__restrict__ fpp, int dim, float *__restrict__ sx,
                                    float *__restrict__ sy, float *__restrict__ sz, float *__restrict__ sw,
                                    float *__restrict__ spp, int32_t *__restrict__ sid,
                                    const double *__restrict__ nx, const double *__restrict__ ny,
                                    const double *__restrict__ nz, const double *__restrict__ nw,
                                    double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ dz,
                                    double *__restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int2 r = sr[i];
  const int p = start[r.x] + r.y;
  sx[p] = fx[i]; sy[p] = fy[i]; sz[p] = fz[i];
  dx[p] = nx[i]; dy[p] = ny[i]; dz[p] = nz[i];
  if (dim == 4) { sw[p] = fw[i]; dw[p] = nw[i]; }
  spp[p] = fpp[i];
  sid[p] = i;
}

// exact fp64 x and y extent of every chunk of kSlabChunk positions (one wave per chunk)
__global__ __launch_bounds__(256) void chunk_range_kernel(const double *__restrict__ nx, const double *__restrict__ ny,
                                                          const int32_t *__restrict__ sid, int n, int n_chunks,
                                                          ChunkExt *__restrict__ chunk_ext) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  unsigned long long lo = ~0ull, hi = 0ull, ylo = ~0ull, yhi = 0ull;
  for (int u = 0; u < kSlabChunk / 64; ++u) {
    const int p = c * kSlabChunk + u * 64 + lane;
    if (p < n) {
      const int id = sid[p];
      const double x = nx[id], y = ny[id];
      if (x == x) { const unsigned long long e = enc_ord(x); lo = min(lo, e); hi = max(hi, e); }
      if (y == y) { const unsigned long long e = enc_ord(y); ylo = min(ylo, e); yhi = max(yhi, e); }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, (unsigned long long)__shfl_xor(lo, off));
    hi = max(hi, (unsigned long long)__shfl_xor(hi, off));
    ylo = min(ylo, (unsigned long long)__shfl_xor(ylo, off));
    yhi = max(yhi, (unsigned long long)__shfl_xor(yhi, off));
  }
  if (lane == 0) {
    ChunkExt ce;
    ce.xlo = lo; ce.xhi = hi; ce.ylo = ylo; ce.yhi = yhi;
    chunk_ext[c] = ce;
  }
}

// ---- sorted runs: a batch appended between two rebuilds is laid out in cell order inside its own range of
// positions, so that its chunks are short strips of the grid instead of samples of the whole world and a
// tile's extent test skips most of them.  (Position -> node index stays sl_id; only the order inside the
// batch's range changes.)
constexpr int kRunMaxCells = 4096;
constexpr int kRunLdsChunks = 512;

// every workgroup scans the cell histogram itself (at most kRunMaxCells counters: cheaper than a launch)
__global__ __launch_bounds__(256) void run_place_kernel(
    long long base, int n, int K, const int2 *__restrict__ sr, const int *__restrict__ hist, const float *__restrict__ fx,
    const float *__restrict__ fy, const float *__restrict__ fz, const float *__restrict__ fw, const float *__restrict__ fpp,
    int dim, float *__restrict__ sx, float *__restrict__ sy, float *__restrict__ sz, float *__restrict__ sw,
    float *__restrict__ spp, int32_t *__restrict__ sid, const double *__restrict__ nx, const double *__restrict__ ny,
    const double *__restrict__ nz, const double *__restrict__ nw, double *__restrict__ dx, double *__restrict__ dy,
    double *__restrict__ dz, double *__restrict__ dw, ChunkExt *__restrict__ chunk_ext, int *__restrict__ hist_next) {
  __shared__ int start[kRunMaxCells];
  // the histogram the NEXT run will count in (last read by the run before this one) is cleared on the way
  for (int k = blockIdx.x * 256 + threadIdx.x; k <= kRunMaxCells; k += gridDim.x * 256) hist_next[k] = 0;
  __shared__ int wsum[4];
  __shared__ unsigned long long ext[kRunLdsChunks][4];      // the workgroup's share of the run's chunk extents
  const long long ch0 = base / kSlabChunk;
  const int n_run_chunks = (int)((base + n - 1) / kSlabChunk - ch0 + 1);
  const bool lds_ext = n_run_chunks <= kRunLdsChunks;
  if (lds_ext)
    for (int k = threadIdx.x; k < n_run_chunks; k += 256) { ext[k][0] = ~0ull; ext[k][1] = 0ull; ext[k][2] = ~0ull; ext[k][3] = 0ull; }
  {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (K + 255) / 256;
    const int b0 = min(t * per, K), b1 = min(b0 + per, K);
    int local = 0;
    for (int k = b0; k < b1; ++k) local += hist[k];
    int v = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(v, off);
      if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int prefix = v - local;
    for (int w = 0; w < wave; ++w) prefix += wsum[w];
    for (int k = b0; k < b1; ++k) { start[k] = prefix; prefix += hist[k]; }
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int2 r = sr[i];
    const int off = start[r.x] + r.y;
    if ((unsigned)off < (unsigned)n) {                 // always, with a consistent histogram; never write outside
      const long long p = base + off, src = base + i;
      const double x = nx[src], y = ny[src];
      sx[p] = fx[src]; sy[p] = fy[src]; sz[p] = fz[src];
      dx[p] = x; dy[p] = y; dz[p] = nz[src];
      if (dim == 4) { sw[p] = fw[src]; dw[p] = nw[src]; }
      spp[p] = fpp[src];
      sid[p] = (int32_t)src;
      // chunk extents: through LDS first (the 512 nodes of a chunk would otherwise queue up on four addresses)
      const long long ch = p / kSlabChunk;
      unsigned long long *e4 = lds_ext ? ext[ch - ch0] : &chunk_ext[ch].xlo;
      if (x == x) { const unsigned long long e = enc_ord(x); atomicMin(&e4[0], e); atomicMax(&e4[1], e); }
      if (y == y) { const unsigned long long e = enc_ord(y); atomicMin(&e4[2], e); atomicMax(&e4[3], e); }
    }
  }
  if (lds_ext) {
    __syncthreads();
    for (int k = threadIdx.x; k < n_run_chunks; k += 256) {
      ChunkExt *g = &chunk_ext[ch0 + k];
      if (ext[k][0] != ~0ull) { atomicMin(&g->xlo, ext[k][0]); atomicMax(&g->xhi, ext[k][1]); }
      if (ext[k][2] != ~0ull) { atomicMin(&g->ylo, ext[k][2]); atomicMax(&g->yhi, ext[k][3]); }
    }
  }
}

}  // namespace

// can the batch [base, base + n) be appended as a sorted run?  (an index exists, its grid fits the place kernel,
// the batch is worth two more launches)
bool slab_run_wanted(const rrtx_ctx *ctx, int64_t n) {
  return ctx->sl_n_sorted > 0 && ctx->ws_slab_params.p && ctx->sl_cells > 0 && ctx->sl_cells <= kRunMaxCells && n >= 1024;
}

// before the append kernel of a batch that becomes a sorted run: where that kernel leaves the nodes' (cell, rank)
int slab_run_prepare(rrtx_ctx *ctx, int64_t n, RunRank *rr) {
  if (!ctx->ws_run_hist.p) {
    // two histograms used alternately (a run's place kernel clears the other one), zero from the start
    RRTX_HIP(ctx, ctx->ws_run_hist.ensure(sizeof(int) * 2 * (size_t)(kRunMaxCells + 1)));
    RRTX_HIP(ctx, hipMemsetAsync(ctx->ws_run_hist.p, 0, sizeof(int) * 2 * (size_t)(kRunMaxCells + 1), ctx->stream));
  }
  RRTX_HIP(ctx, ctx->ws_run_sr.ensure(sizeof(int2) * (size_t)n));
  rr->sp = ctx->ws_slab_params.as<SlabParams>();
  rr->hist = ctx->ws_run_hist.as<int>() + (size_t)ctx->run_hist_flip * (kRunMaxCells + 1);
  rr->sr = ctx->ws_run_sr.as<int2>();
  return RRTX_OK;
}

// the slab arrays of nodes [base, base + n) (index-order arrays and (cell, rank) already written by the append
// kernel) in cell order
int slab_append_run(rrtx_ctx *ctx, int64_t base, int64_t n) {
  hipStream_t st = ctx->stream;
  const int K = ctx->sl_cells;
  const int wi = ctx->dim == 4 ? 3 : 2;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  int *hist = ctx->ws_run_hist.as<int>() + (size_t)ctx->run_hist_flip * (kRunMaxCells + 1);
  int *hist_next = ctx->ws_run_hist.as<int>() + (size_t)(ctx->run_hist_flip ^ 1) * (kRunMaxCells + 1);
  ctx->run_hist_flip ^= 1;
  hipLaunchKernelGGL(run_place_kernel, grid, block, 0, st, (long long)base, (int)n, K, ctx->ws_run_sr.as<int2>(),
                     hist, ctx->nodes_f[0], ctx->nodes_f[1], ctx->nodes_f[2], ctx->nodes_f[wi],
                     ctx->nodes_pp, ctx->dim, ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2], ctx->sl_f[wi], ctx->sl_pp, ctx->sl_id,
                     ctx->nodes[0], ctx->nodes[1], ctx->nodes[2], ctx->nodes[wi], ctx->sl_d[0], ctx->sl_d[1], ctx->sl_d[2],
                     ctx->sl_d[wi], reinterpret_cast<ChunkExt *>(ctx->chunk_ext), hist_next);
  RRTX_HIP(ctx, hipGetLastError());
  // chunks of the run (a chunk shared with the batch before counts once)
  ctx->sl_run_chunks += (double)((base + n + kSlabChunk - 1) / kSlabChunk - (base + kSlabChunk - 1) / kSlabChunk);
  return RRTX_OK;
}

// When to rebuild: a search pays for every chunk of the appended tail.  An unsorted tail chunk (nodes in arrival
// order) spans the world, so every tile screens it: one more 512-node unit per tile, ~1 us per search of a
// 16384-sample batch (1024 tiles in flight).  A chunk of a sorted run (slab_append_run) is a strip of the grid,
// of which a tile's reach touches about one in six: ~5 units per tile and run of 32 chunks, ~0.16 us per run
// chunk (tools/steady_probe.py).  A rebuild costs ~40 us of launches + ~0.12 ns per node (the sort keys carry
// the bin of the third coordinate, so the rank atomics spread over 32 x more counters than in round 1: 83 -> 15 us
// at 400 k nodes).  The searches since the last rebuild run up a debt of what the tail has cost them; the index
// is rebuilt when the debt exceeds the price of a rebuild (and always before the tail outgrows half the tree).
// Single appends between single queries thus rebuild rarely, whole batches appended between batched searches
// every five or six calls.  Only the moment changes; no result depends on it.
int slab_refresh(rrtx_ctx *ctx, long long n_tiles) {
  const int64_t n = ctx->n_nodes;
  const int64_t tail = n - ctx->sl_n_sorted;
  const int64_t floor_tail = n / 128 > 1024 ? n / 128 : 1024;
  if (tail <= floor_tail) return RRTX_OK;
  const double tail_chunks = (double)((tail + kSlabChunk - 1) / kSlabChunk);
  const double run_chunks = ctx->sl_run_chunks < tail_chunks ? ctx->sl_run_chunks : tail_chunks;
  // (tiles in flight: 1024; a small batch spreads a tile's units over several workgroups)
  const double rounds = n_tiles > 64 ? (double)n_tiles / 1024.0 : 1.0 / 16.0;
  ctx->sl_debt_us += (1.0 * (tail_chunks - run_chunks) + 0.16 * run_chunks) * rounds;
  const double rebuild_us = 40.0 + 0.12e-3 * (double)n;
  if (ctx->sl_n_sorted > 0 && tail <= n / 2 && ctx->sl_debt_us < rebuild_us) return RRTX_OK;
  ctx->sl_debt_us = 0.0;
  ctx->sl_run_chunks = 0.0;
  hipStream_t st = ctx->stream;
  // about one cell per chunk: cells of ~512 nodes, laid out as a square grid over (x, y)
  // (experiment switches, RRTX_OPT_TUNE bits 24-25: smaller cells; bits 8-15: bins of the third coordinate)
  const int cell_nodes = kSlabChunk >> ((ctx->opt_tune >> 24) & 3);
  int side = (int)std::sqrt((double)n / (double)cell_nodes);
  if (side < 2) side = 2;
  if (side > 256) side = 256;
  const int K = side * side;
  const int Kz = ((ctx->opt_tune >> 8) & 0xff) ? ((ctx->opt_tune >> 8) & 0xff) : kSlabKz;
  const int KK = K * Kz;                  // sort keys: (cell, bin of the third coordinate)
  // Workspaces sized for the node CAPACITY, not the current count: a tree that grows towards its capacity then
  // rebuilds without a hipFree / hipMalloc pair (each a device synchronisation, occasionally milliseconds) in
  // the middle of the caller's stream of searches and appends.
  const int64_t n_cap = ctx->cap_nodes > n ? ctx->cap_nodes : n;
  int side_cap = (int)std::sqrt((double)n_cap / (double)cell_nodes);
  if (side_cap < side) side_cap = side;
  if (side_cap > 256) side_cap = 256;
  const size_t kk_cap = (size_t)side_cap * side_cap * Kz + 1;
  RRTX_HIP(ctx, ctx->ws_slab_params.ensure(sizeof(SlabParams)));
  RRTX_HIP(ctx, ctx->ws_slab_hist.ensure(sizeof(int) * kk_cap));
  RRTX_HIP(ctx, ctx->ws_slab_start.ensure(sizeof(int) * kk_cap));
  RRTX_HIP(ctx, ctx->ws_slab_sr.ensure(sizeof(int2) * (size_t)n_cap));
  SlabParams *sp = ctx->ws_slab_params.as<SlabParams>();
  int *hist = ctx->ws_slab_hist.as<int>();
  int *start = ctx->ws_slab_start.as<int>();
  int2 *sr = ctx->ws_slab_sr.as<int2>();
  const int nb = (int)((n + 255) / 256);
  const int n_chunks = (int)((n + kSlabChunk - 1) / kSlabChunk);
  span_begin(ctx, KF_NN_FINISH);
  hipLaunchKernelGGL(slab_params_kernel, dim3((KK + 256) / 256), dim3(256), 0, st,
                     ctx->d_xrange.as<unsigned long long>(), side, side, Kz, sp, hist);
  hipLaunchKernelGGL(slab_rank_kernel, dim3(nb), dim3(256), 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2], (int)n, sp,
                     hist, sr);
  hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, st, hist, start, KK);
  hipLaunchKernelGGL(slab_scatter_kernel, dim3(nb), dim3(256), 0, st, (int)n, sr, start, ctx->nodes_f[0],
                     ctx->nodes_f[1], ctx->nodes_f[2], ctx->nodes_f[ctx->dim == 4 ? 3 : 2], ctx->nodes_pp, ctx->dim,
                     ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2], ctx->sl_f[ctx->dim == 4 ? 3 : 2], ctx->sl_pp,
                     ctx->sl_id, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2], ctx->nodes[ctx->dim == 4 ? 3 : 2],
                     ctx->sl_d[0], ctx->sl_d[1], ctx->sl_d[2], ctx->sl_d[ctx->dim == 4 ? 3 : 2]);
  hipLaunchKernelGGL(chunk_range_kernel, dim3((n_chunks + 3) / 4), dim3(256), 0, st, ctx->nodes[0], ctx->nodes[1],
                     ctx->sl_id, (int)n, n_chunks, reinterpret_cast<ChunkExt *>(ctx->chunk_ext));
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  ctx->sl_n_sorted = n;
  ctx->sl_cells = K;
  ctx->sl_kz = Kz;
  return RRTX_OK;
}

}  // namespace rrtx
