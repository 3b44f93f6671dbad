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
// exclusive prefix sum of n counters by ONE workgroup, 4096 at a time: every thread takes four neighbouring counters (one
// 16-byte load, the wave reads 1 KB in a row), the waves' sums meet in LDS, the running total carries over.  (Round 2 gave
// every thread a contiguous stretch of n / 1024 counters: 64 cache lines per load instruction and two dependent loads per
// counter -- 22 us for the 15 000 counters of a 500 k-node index, a quarter of a rebuild.)
__global__ __launch_bounds__(1024) void excl_scan_kernel(const int *__restrict__ in, int *__restrict__ out, int n) {
  __shared__ int wsum[2][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int carry = 0, flip = 0;
  for (int base = 0; base < n; base += 4096, flip ^= 1) {
    const int i = base + 4 * t;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (i + 3 < n) {
      const int4 v4 = *reinterpret_cast<const int4 *>(in + i);       // (in is 256-byte aligned, i a multiple of four)
      c0 = v4.x; c1 = v4.y; c2 = v4.z; c3 = v4.w;
    } else {
      if (i < n) c0 = in[i];
      if (i + 1 < n) c1 = in[i + 1];
      if (i + 2 < n) c2 = in[i + 2];
    }
    const int local = c0 + c1 + c2 + c3;
    int v = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(v, off);
      if (lane >= off) v += o;
    }
    if (lane == 63) wsum[flip][wave] = v;
    __syncthreads();                      // (two sets of sums: the next round's writes cannot overtake this round's reads)
    int prefix = carry + v - local, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int x = wsum[flip][w];
      if (w < wave) prefix += x;
      total += x;
    }
    if (i + 3 < n) {
      *reinterpret_cast<int4 *>(out + i) = make_int4(prefix, prefix + c0, prefix + c0 + c1, prefix + c0 + c1 + c2);
    } else {
      if (i < n) out[i] = prefix;
      if (i + 1 < n) out[i + 1] = prefix + c0;
      if (i + 2 < n) out[i + 2] = prefix + c0 + c1;
    }
    carry += total;
  }
  if (t == 0) out[n] = carry;
}

// Rebuild, last two steps.  (Round 2 scattered every node's ten values to its place -- ten partial-line writes to
// random addresses per node, 33 us for 400 k nodes -- and read them back in a separate launch for the chunks' extents, 11
// us.)  Now only the node's index goes to its place (4 bytes), and a second kernel walks the places in order, fetches
// the values of the node that sits there (random READS of arrays that fit the L2) and writes rows; a workgroup is one
// chunk, so the chunk's exact fp64 extent is a reduction over the workgroup on the way.
__global__ void slab_sid_kernel(int n, const int2 *__restrict__ sr, const int *__restrict__ start,
                                int32_t *__restrict__ sid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int2 r = sr[i];
  sid[start[r.x] + r.y] = i;
}

__global__ __launch_bounds__(kSlabChunk) void slab_gather_kernel(
    int n, const int32_t *__restrict__ sid, const double4 *__restrict__ aos, double ox, double oy, double oz, double ow,
    int dim, float *__restrict__ sx, float *__restrict__ sy, float *__restrict__ sz, float *__restrict__ sw,
    float *__restrict__ spp, double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ dz,
    double *__restrict__ dw, ChunkExt *__restrict__ chunk_ext) {
  const int p = blockIdx.x * kSlabChunk + threadIdx.x;
  unsigned long long lo = ~0ull, hi = 0ull, ylo = ~0ull, yhi = 0ull;
  if (p < n) {
    // ONE random read per node: the (x, y, z, w) record; the fp32 shadow values are worked out again from it exactly as
    // the append kernel did (aos_to_soa_kernel: same expressions, no contraction), instead of five more random reads
    const double4 v = aos[sid[p]];
    const double x = v.x, y = v.y;
    const float fa = (float)(v.x - ox), fb = (float)(v.y - oy), fc = (float)(v.z - oz);
    double pp = (double)fa * (double)fa + (double)fb * (double)fb + (double)fc * (double)fc;
    sx[p] = fa; sy[p] = fb; sz[p] = fc;
    dx[p] = x; dy[p] = y; dz[p] = v.z;
    if (dim == 4) {
      const float fd = (float)(v.w - ow);
      sw[p] = fd; dw[p] = v.w;
      pp += (double)fd * (double)fd;
    }
    spp[p] = (float)pp;
    if (x == x) { lo = hi = enc_ord(x); }
    if (y == y) { ylo = yhi = enc_ord(y); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, (unsigned long long)__shfl_xor(lo, off));
    hi = max(hi, (unsigned long long)__shfl_xor(hi, off));
    ylo = min(ylo, (unsigned long long)__shfl_xor(ylo, off));
    yhi = max(yhi, (unsigned long long)__shfl_xor(yhi, off));
  }
  __shared__ unsigned long long red[kSlabChunk / 64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = lo; red[wave][1] = hi; red[wave][2] = ylo; red[wave][3] = yhi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ChunkExt ce;
    ce.xlo = ~0ull; ce.xhi = 0ull; ce.ylo = ~0ull; ce.yhi = 0ull;
    for (int w = 0; w < kSlabChunk / 64; ++w) {
      ce.xlo = min(ce.xlo, red[w][0]); ce.xhi = max(ce.xhi, red[w][1]);
      ce.ylo = min(ce.ylo, red[w][2]); ce.yhi = max(ce.yhi, red[w][3]);
    }
    chunk_ext[blockIdx.x] = ce;
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
    // (four neighbouring counters per thread, 1024 per round: the wave reads 1 KB in a row -- a contiguous stretch of
    //  K / 256 counters per thread was 2 x 16 dependent, scattered loads)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int carry = 0;
    for (int kb = 0; kb < K; kb += 1024) {
      const int k = kb + 4 * t;
      const int c0 = k < K ? hist[k] : 0, c1 = k + 1 < K ? hist[k + 1] : 0, c2 = k + 2 < K ? hist[k + 2] : 0,
                c3 = k + 3 < K ? hist[k + 3] : 0;
      const int local = c0 + c1 + c2 + c3;
      int v = local;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
      }
      if (lane == 63) wsum[wave] = v;
      __syncthreads();
      int prefix = carry + v - local, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int x = wsum[w];
        if (w < wave) prefix += x;
        total += x;
      }
      if (k < K) start[k] = prefix;
      if (k + 1 < K) start[k + 1] = prefix + c0;
      if (k + 2 < K) start[k + 2] = prefix + c0 + c1;
      if (k + 3 < K) start[k + 3] = prefix + c0 + c1 + c2;
      carry += total;
      __syncthreads();
    }
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
  // (round 3: a rebuild is 47 us at 400 k nodes; with 15 + 0.08e-3 n here the index is rebuilt every four batches instead of
  // six and the steady step is the same 0.102-0.103 ms -- tools/steady_trace.py -- so the estimate stays)
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
  hipLaunchKernelGGL(slab_sid_kernel, dim3(nb), dim3(256), 0, st, (int)n, sr, start, ctx->sl_id);
  hipLaunchKernelGGL(slab_gather_kernel, dim3(n_chunks), dim3(kSlabChunk), 0, st, (int)n, ctx->sl_id,
                     reinterpret_cast<const double4 *>(ctx->nodes_aos), ctx->origin[0], ctx->origin[1], ctx->origin[2],
                     ctx->origin[3], ctx->dim, ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2], ctx->sl_f[ctx->dim == 4 ? 3 : 2],
                     ctx->sl_pp, ctx->sl_d[0], ctx->sl_d[1], ctx->sl_d[2], ctx->sl_d[ctx->dim == 4 ? 3 : 2],
                     reinterpret_cast<ChunkExt *>(ctx->chunk_ext));
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  ctx->sl_n_sorted = n;
  ctx->sl_cells = K;
  ctx->sl_kz = Kz;
  return RRTX_OK;
}

}  // namespace rrtx
