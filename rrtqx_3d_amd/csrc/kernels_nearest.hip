// kernels_nearest.hip -- batched kdFindNearest (R/kdTree_general.jl:254-385) over the node SoA:
// exact fp64 scan and the screened fast path.  gfx950 only.
#include "nn_device.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rrtx {

namespace {

// -------------------------------------------------------------- nearest -----
// lane = query; nodes are streamed through SGPRs (wave-uniform loads).  Each
// block handles 256 queries against one node segment and writes the segment's
// best (d2, idx); nn_nearest_reduce picks the lexicographic minimum.
template <int D>
__global__ __launch_bounds__(256) void nn_nearest_partial_kernel(
    const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz,
    const double *__restrict__ nw, int n_nodes, const double *__restrict__ q, int nq, int n_wraps, int wd0,
    int wd1, int wd2, double wp0, double wp1, double wp2, int seg_len, double *__restrict__ part_d2,
    int32_t *__restrict__ part_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int seg = blockIdx.y;
  const int node_begin = seg * seg_len;
  const int node_end = min(n_nodes, node_begin + seg_len);
  double p[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < nq) {
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = q[(size_t)i * D + k];
  }
  const int wd[3] = {wd0, wd1, wd2};
  const double wp[3] = {wp0, wp1, wp2};
  double best = __builtin_inf();
  int best_i = 0x7fffffff;
  const int n_slots = 1 << n_wraps;
  for (int k = 0; k < n_slots; ++k) {
    double g[4] = {p[0], p[1], p[2], p[3]};
    for (int w = 0; w < n_wraps; ++w) {
      if (!((k >> (n_wraps - 1 - w)) & 1)) continue;
      int dimi = wd[w];
      g[dimi] = (p[dimi] < wp[w] / 2.0) ? (p[dimi] + wp[w]) : (p[dimi] - wp[w]);
    }
#pragma unroll 4
    for (int n = node_begin; n < node_end; ++n) {
      double s;
      if constexpr (D == 4) s = sq4(g[0], g[1], g[2], g[3], nx[n], ny[n], nz[n], nw[n]);
      else s = sq3(g[0], g[1], g[2], nx[n], ny[n], nz[n]);
      bool better = (s < best) || (s == best && n < best_i);
      best = better ? s : best;
      best_i = better ? n : best_i;
    }
  }
  if (i < nq) {
    part_d2[(size_t)seg * nq + i] = best;
    part_idx[(size_t)seg * nq + i] = best_i;
  }
}

__global__ void nn_nearest_reduce_kernel(const double *__restrict__ part_d2,
                                         const int32_t *__restrict__ part_idx, int nq, int n_seg,
                                         int32_t *__restrict__ idx, double *__restrict__ dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  double best = __builtin_inf();
  int best_i = 0x7fffffff;
  for (int s = 0; s < n_seg; ++s) {
    double d2 = part_d2[(size_t)s * nq + i];
    int id = part_idx[(size_t)s * nq + i];
    bool better = (d2 < best) || (d2 == best && id < best_i);
    best = better ? d2 : best;
    best_i = better ? id : best_i;
  }
  idx[i] = best_i;
  dist[i] = sqrt_rn(best);
}

// nearest from the radius lists (valid when the list is non-empty): the first
// minimum of the stored keys, ties to the lowest index (lists are index-sorted)
__global__ void nn_nearest_from_lists_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ idx,
                                             const double *__restrict__ dist, int nq,
                                             int32_t *__restrict__ nearest_idx,
                                             double *__restrict__ nearest_dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  double best = __builtin_inf();
  int best_i = -1;
  for (int64_t k = offsets[i]; k < offsets[i + 1]; ++k) {
    double d = dist[k];
    if (d < best) { best = d; best_i = idx[k]; }
  }
  nearest_idx[i] = best_i;   // -1: empty list, caller falls back to the full scan
  nearest_dist[i] = best;
}

// ------------------------------------------------ nearest, screened (fast path) ------
// kdFindNearest for a batch: lane = query copy, nodes streamed through SGPRs 8 at a time.
// Each lane keeps the running minimum of the fp32 screen value t (see "fp32 prefilter");
// for two nodes i, j of the same copy, "i is at least as close as j" in exact arithmetic
// implies  t_i <= t_j + M  with  M = (2K + 16 D + 4) eps C^2  (both roundings of t, plus the
// effect of the fp32 conversions on the two distances).  So every node that can be the exact
// nearest satisfies t <= running_min + M when it is visited; those few nodes (about ln(n) per
// segment) are queued and confirmed with the exact unfused fp64 distance:
//   atomicMin on the bit pattern of d2 (monotone for non-negative doubles), then the lowest
//   index among the records that attain it (nn_nearest_tie_kernel).
template <int D>
__device__ __forceinline__ void drain_nearest(const int2 *cand, int &wn, const double *__restrict__ nx,
                                              const double *__restrict__ ny, const double *__restrict__ nz,
                                              const double *__restrict__ nw,
                                              const typename QRecT<D>::type *__restrict__ copies,
                                              const int2 *__restrict__ meta, HitRec *__restrict__ recs,
                                              long long cap, Scalars *__restrict__ sc,
                                              unsigned long long *__restrict__ best_bits,
                                              int *__restrict__ redo) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  for (int i0 = 0; i0 < wn; i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < wn;
    double s = 0.0;
    int owner = 0, id = 0;
    if (v) {
      const int2 c = cand[i];
      id = c.y;
      const typename QRecT<D>::type ce = copies[c.x];
      owner = meta[c.x].x;
      if constexpr (D == 4) s = sq4(ce.x, ce.y, ce.z, ce.w, nx[id], ny[id], nz[id], nw[id]);
      else s = sq3(ce.x, ce.y, ce.z, nx[id], ny[id], nz[id]);
      atomicMin(&best_bits[owner], (unsigned long long)__double_as_longlong(s));
    }
    const unsigned long long mask = __ballot(v);
    const int n = __popcll(mask);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&sc->total, (unsigned long long)n);
    base = __shfl(base, 0);
    if (v) {
      const long long pos = (long long)base + lane;
      if (pos < cap) {
        HitRec r;
        r.owner = owner; r.idx = id; r.d2 = s;
        recs[pos] = r;
      } else {
        redo[owner] = 1;      // record dropped: this query is answered by nn_nearest_fixup_kernel
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  wn = 0;
}

template <int D>
__global__ __launch_bounds__(256) void nn_nearest_f32_kernel(
    const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz,
    const double *__restrict__ nw, const float *__restrict__ fx, const float *__restrict__ fy,
    const float *__restrict__ fz, const float *__restrict__ fw, const float *__restrict__ fpp, int n_nodes,
    const typename QRecT<D>::type *__restrict__ copies, const typename QRecFT<D>::type *__restrict__ copies_f,
    const int2 *__restrict__ meta, const unsigned long long *__restrict__ node_absmax, int n_seg, int seg_len,
    HitRec *__restrict__ recs, long long cap, Scalars *__restrict__ sc,
    unsigned long long *__restrict__ best_bits, int *__restrict__ redo) {
  __shared__ int2 cand_all[4][kCandCap];
  const int seg = blockIdx.x % n_seg;        // XCD-affine node segment
  const int cb = blockIdx.x / n_seg;
  const int n_copies = sc->n_copies;
  if (cb * 256 >= n_copies) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int2 *cand = cand_all[wave];
  int wn = 0;
  const int copy = cb * 256 + threadIdx.x;
  typename QRecFT<D>::type c;
  c.x = 0.f; c.y = 0.f; c.z = 0.f;
  if constexpr (D == 4) c.w = 0.f;
  bool valid = copy < n_copies;
  if (valid) c = copies_f[copy];
  // a NaN copy can never be ordered: it produces no candidates (result: idx INT_MAX, dist inf)
  valid = valid && (c.x == c.x) && (c.y == c.y) && (c.z == c.z);
  if constexpr (D == 4) valid = valid && (c.w == c.w);
  unsigned long long cbits = max(*node_absmax, sc->q_absmax);
  const double C = __longlong_as_double((long long)cbits);
  const double eps = 5.9604644775390625e-08;
  const double Kc = (D == 4) ? (2.0 * 42.0 + 16.0 * 4.0 + 4.0) : (2.0 * 26.0 + 16.0 * 3.0 + 4.0);
  const float M = (C <= 1e15) ? __double2float_ru(Kc * eps * C * C + 1e-30) : __builtin_inff();
  const f32x2 cx2 = {c.x, c.x}, cy2 = {c.y, c.y}, cz2 = {c.z, c.z};
  float runmin = __builtin_inff();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int node_begin = seg * seg_len;
  const int node_end = min(n_nodes, node_begin + seg_len);

  // t of eight wave-uniform nodes starting at j (unconditional scalar loads, straight-line VALU):
  // t = fma(cx, px, fma(cy, py, cz * pz)) + pp  (same rounding budget K as the range screen)
  auto screen_group = [&](int j, float *t) -> float {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = j + 2 * v;
      const f32x2 pz = {fz[a], fz[a + 1]}, py = {fy[a], fy[a + 1]}, px = {fx[a], fx[a + 1]};
      const f32x2 pp = {fpp[a], fpp[a + 1]};
      f32x2 acc = cz2 * pz;
      if constexpr (D == 4) {
        const f32x2 cw2 = {c.w, c.w}, pw = {fw[a], fw[a + 1]};
        acc = __builtin_elementwise_fma(cw2, pw, acc);
      }
      acc = __builtin_elementwise_fma(cy2, py, acc);
      acc = __builtin_elementwise_fma(cx2, px, acc);
      acc = acc + pp;
      t[2 * v] = acc.x; t[2 * v + 1] = acc.y;
    }
    const float m1 = fminf(fminf(t[0], t[1]), t[2]);
    const float m2 = fminf(fminf(t[3], t[4]), t[5]);
    const float m3 = fminf(fminf(t[6], t[7]), m1);
    return fminf(m2, m3);
  };
  // warm start: the running minimum over the first nodes of the tree (no candidates are taken
  // here; these nodes are visited again by the segment that owns them).  Starting from the best
  // of m samples cuts the expected number of running-minimum updates per segment from ln(n) to
  // ln((n + m) / m).
  {
    const int warm = min(kNearestWarm, n_nodes / 8 * 8);
    for (int j = 0; j < warm; j += 8) {
      float t[8];
      runmin = fminf(runmin, screen_group(j, t));
    }
  }
  const int full_end = node_begin + (node_end - node_begin) / 8 * 8;
  for (int j = node_begin; j < full_end; j += 8) {
    float t[8];
    const float tmin = screen_group(j, t);
    const float bound = runmin + M;
    if (__builtin_expect(__ballot(valid && !(tmin > bound)) != 0ull, 0)) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long m = __ballot(valid && !(t[u] > bound));
        if (m == 0ull) continue;
        const int n = __popcll(m);
        if (wn + n > kCandCap) drain_nearest<D>(cand, wn, nx, ny, nz, nw, copies, meta, recs, cap, sc, best_bits, redo);
        if ((m >> lane) & 1ull) cand[wn + __popcll(m & lt_mask)] = make_int2(copy, j + u);
        wn += n;
      }
    }
    runmin = fminf(runmin, tmin);
  }
  // ragged tail (< 8 nodes, only the last segment has one): every node is a candidate
  for (int j = full_end; j < node_end; ++j) {
    const unsigned long long m = __ballot(valid);
    if (m == 0ull) break;
    const int n = __popcll(m);
    if (wn + n > kCandCap) drain_nearest<D>(cand, wn, nx, ny, nz, nw, copies, meta, recs, cap, sc, best_bits, redo);
    if (valid) cand[wn + __popcll(m & lt_mask)] = make_int2(copy, j);
    wn += n;
  }
  drain_nearest<D>(cand, wn, nx, ny, nz, nw, copies, meta, recs, cap, sc, best_bits, redo);
}

// lowest node index among the confirmed candidates that attain the minimum d2
__global__ void nn_nearest_tie_kernel(const HitRec *__restrict__ recs, long long cap,
                                      const Scalars *__restrict__ sc,
                                      const unsigned long long *__restrict__ best_bits,
                                      int *__restrict__ best_idx) {
  long long total = (long long)sc->total;
  if (total > cap) total = cap;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const HitRec r = recs[i];
    if ((unsigned long long)__double_as_longlong(r.d2) == best_bits[r.owner]) atomicMin(&best_idx[r.owner], r.idx);
  }
}

__global__ void nn_nearest_out_kernel(const unsigned long long *__restrict__ best_bits,
                                      const int *__restrict__ best_idx, int nq, int32_t *__restrict__ idx,
                                      double *__restrict__ dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const unsigned long long b = best_bits[i];
  idx[i] = (b == ~0ull) ? 0x7fffffff : best_idx[i];     // no candidate (NaN query): same as the exact scan
  dist[i] = (b == ~0ull) ? __builtin_inf() : sqrt_rn(__longlong_as_double((long long)b));
}

// The screened scan keeps about ln(n) candidates per query and segment; an adversarial visiting order
// (nodes sorted by decreasing distance) can produce more than the record buffer holds.  Queries that
// lost a record are answered again here, exactly and on the device (no host round trip, so the
// device-pointer entry points cannot come back wrong): one workgroup per such query, expanding
// search over the slab index for the query and each of its ghosts (the same 2^w copies the exact
// scan walks), lexicographic minimum of (d2, index).
constexpr int kFixThreads = 512;
template <int D>
__global__ __launch_bounds__(kFixThreads) void nn_nearest_fixup_kernel(
    const Scalars *__restrict__ sc, long long cap, const int *__restrict__ redo, const double *__restrict__ q, int nq,
    int n_wraps, int wd0, int wd1, int wd2, double wp0, double wp1, double wp2, NearestIndex ni,
    int32_t *__restrict__ idx, double *__restrict__ dist) {
  __shared__ NearestScratch ns;
  if ((long long)sc->total <= cap) return;           // nothing was dropped (the normal case)
  const int wd[3] = {wd0, wd1, wd2};
  const double wp[3] = {wp0, wp1, wp2};
  for (int i = blockIdx.x; i < nq; i += gridDim.x) {
    if (!redo[i]) continue;
    double p[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = q[(size_t)i * D + k];
    double best = __builtin_inf();
    int best_i = 0x7fffffff;
    const int n_slots = 1 << n_wraps;
    for (int k = 0; k < n_slots; ++k) {
      double g[4] = {p[0], p[1], p[2], p[3]};
      for (int w = 0; w < n_wraps; ++w) {
        if (!((k >> (n_wraps - 1 - w)) & 1)) continue;
        const int dimi = wd[w];
        g[dimi] = (p[dimi] < wp[w] / 2.0) ? (p[dimi] + wp[w]) : (p[dimi] - wp[w]);
      }
      double bs;
      int bi;
      block_nearest<D, kFixThreads>(ni, g[0], g[1], g[2], g[3], 1.0, ns, bs, bi);
      if ((bs < best) || (bs == best && bi < best_i)) { best = bs; best_i = bi; }
    }
    if (threadIdx.x == 0) {
      idx[i] = best_i;
      dist[i] = sqrt_rn(best);
    }
  }
}

}  // namespace

static int launch_nn_nearest_exact(rrtx_ctx *ctx, const double *q_dev, int nq, int32_t *idx_dev, double *dist_dev) {
  const int n_nodes = (int)ctx->n_nodes;
  const int qblocks = (nq + 255) / 256;
  int want_seg = (2048 + qblocks - 1) / qblocks;
  int max_seg = (n_nodes + 255) / 256;
  int n_seg = want_seg < max_seg ? want_seg : max_seg;
  if (n_seg < 1) n_seg = 1;
  int seg_len = (n_nodes + n_seg - 1) / n_seg;
  n_seg = (n_nodes + seg_len - 1) / seg_len;
  RRTX_HIP(ctx, ctx->ws_partial.ensure((size_t)n_seg * nq * (sizeof(double) + sizeof(int32_t))));
  double *pd2 = ctx->ws_partial.as<double>();
  int32_t *pidx = reinterpret_cast<int32_t *>(pd2 + (size_t)n_seg * nq);
  hipStream_t st = ctx->stream;
  span_begin(ctx, KF_NN_NEAREST);
  dim3 grid(qblocks, n_seg), block(256);
  if (ctx->dim == 4)
    hipLaunchKernelGGL(nn_nearest_partial_kernel<4>, grid, block, 0, st, ctx->nodes[0], ctx->nodes[1],
                       ctx->nodes[2], ctx->nodes[3], n_nodes, q_dev, nq, ctx->n_wraps, ctx->wrap_dim[0],
                       ctx->wrap_dim[1], ctx->wrap_dim[2], ctx->wrap_period[0], ctx->wrap_period[1],
                       ctx->wrap_period[2], seg_len, pd2, pidx);
  else
    hipLaunchKernelGGL(nn_nearest_partial_kernel<3>, grid, block, 0, st, ctx->nodes[0], ctx->nodes[1],
                       ctx->nodes[2], ctx->nodes[2], n_nodes, q_dev, nq, ctx->n_wraps, ctx->wrap_dim[0],
                       ctx->wrap_dim[1], ctx->wrap_dim[2], ctx->wrap_period[0], ctx->wrap_period[1],
                       ctx->wrap_period[2], seg_len, pd2, pidx);
  hipLaunchKernelGGL(nn_nearest_reduce_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, pd2, pidx, nq, n_seg,
                     idx_dev, dist_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// Screened nearest (see nn_nearest_f32_kernel); queries whose candidates did not fit the record buffer
// are answered by nn_nearest_fixup_kernel on the device.
static int launch_nn_nearest_screened(rrtx_ctx *ctx, const double *q_dev, int nq, int32_t *idx_dev,
                                      double *dist_dev) {
  const int D = ctx->dim;
  const int n_slots = 1 << ctx->n_wraps;
  const int n_nodes = (int)ctx->n_nodes;
  hipStream_t st = ctx->stream;
  const size_t n_copies_max = (size_t)nq * n_slots;
  const size_t qrec_bytes = (D == 4) ? sizeof(QRec4) : sizeof(QRec3);
  const size_t qf_bytes = (D == 4) ? sizeof(QRecF4) : sizeof(QRecF3);
  const long long rec_cap = ctx->opt_nearest_rec_cap > 0 ? ctx->opt_nearest_rec_cap : (long long)n_copies_max * 1024 + 4096;
  RRTX_HIP(ctx, ctx->ws_slots.ensure(n_copies_max * sizeof(SlotRec)));
  RRTX_HIP(ctx, ctx->ws_copies.ensure(n_copies_max * qrec_bytes));
  RRTX_HIP(ctx, ctx->ws_copies_f.ensure((n_copies_max + kQPI) * qf_bytes));
  RRTX_HIP(ctx, ctx->ws_copy_meta.ensure(n_copies_max * sizeof(int2)));
  RRTX_HIP(ctx, ctx->ws_scalars_nn.ensure(sizeof(Scalars)));
  RRTX_HIP(ctx, ctx->ws_recs.ensure((size_t)rec_cap * sizeof(HitRec)));
  RRTX_HIP(ctx, ctx->ws_partial.ensure((size_t)nq * (sizeof(unsigned long long) + 2 * sizeof(int))));
  unsigned long long *best_bits = ctx->ws_partial.as<unsigned long long>();
  int *best_idx = reinterpret_cast<int *>(best_bits + nq);
  int *redo = best_idx + nq;              // queries that lost a candidate record
  Scalars *sc = ctx->ws_scalars_nn.as<Scalars>();
  hipLaunchKernelGGL(nn_init_kernel, dim3((nq + 255) / 256 < 64 ? (nq + 255) / 256 : 64), dim3(256), 0, st, sc,
                     (ctx->n_wraps == 0) ? nq : 0, redo, nq, best_bits, nq, ~0ull, best_idx, nq, 0x7fffffff,
                     (ConfirmArgs *)nullptr, ConfirmArgs{});
  const double inf = std::numeric_limits<double>::infinity();
  const double nan = std::numeric_limits<double>::quiet_NaN();

  // launch geometry: blocks of 256 copies x node segments (segment fast-varying -> XCD affine)
  const int cblocks = (int)((n_copies_max + 255) / 256);
  int want_seg = (2048 + cblocks - 1) / cblocks;
  int max_seg = (n_nodes + 1023) / 1024;
  int n_seg = want_seg < max_seg ? want_seg : max_seg;
  if (n_seg < 1) n_seg = 1;
  if (n_seg >= 8) n_seg = n_seg / 8 * 8;
  int seg_len = round_up((n_nodes + n_seg - 1) / n_seg, 8);
  n_seg = (n_nodes + seg_len - 1) / seg_len;

  span_begin(ctx, KF_NN_NEAREST);
  {
    dim3 grid((nq + 255) / 256), block(256);
    dim3 pgrid((unsigned)((n_copies_max + kQPI + 255) / 256));
    dim3 sgrid((unsigned)cblocks * (unsigned)n_seg);
    if (D == 4) {
      hipLaunchKernelGGL(nn_pack_kernel<4>, grid, block, 0, st, q_dev, nq, (const double *)nullptr,
                         (const double *)nullptr, inf, nan, ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1],
                         ctx->wrap_dim[2], ctx->wrap_period[0], ctx->wrap_period[1], ctx->wrap_period[2],
                         ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_slots.as<SlotRec>(), ctx->ws_copies.as<QRec4>(), ctx->ws_copy_meta.as<int2>(), sc,
                         (const unsigned long long *)nullptr, 1, 1, (int *)nullptr, (int2 *)nullptr, PackFused{},
                         ConfirmArgs{});
      hipLaunchKernelGGL(nn_filter_prep_kernel<4>, pgrid, block, 0, st, ctx->ws_copies.as<QRec4>(), sc,
                         ctx->d_absmax.as<unsigned long long>(), (int)n_copies_max, ctx->origin[0], ctx->origin[1],
                         ctx->origin[2], ctx->origin[3], ctx->ws_copies_f.as<QRecF4>());
      hipLaunchKernelGGL(nn_nearest_f32_kernel<4>, sgrid, block, 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                         ctx->nodes[3], ctx->nodes_f[0], ctx->nodes_f[1], ctx->nodes_f[2], ctx->nodes_f[3],
                         ctx->nodes_pp, n_nodes, ctx->ws_copies.as<QRec4>(), ctx->ws_copies_f.as<QRecF4>(),
                         ctx->ws_copy_meta.as<int2>(), ctx->d_absmax.as<unsigned long long>(), n_seg, seg_len,
                         ctx->ws_recs.as<HitRec>(), rec_cap, sc, best_bits, redo);
    } else {
      hipLaunchKernelGGL(nn_pack_kernel<3>, grid, block, 0, st, q_dev, nq, (const double *)nullptr,
                         (const double *)nullptr, inf, nan, ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1],
                         ctx->wrap_dim[2], ctx->wrap_period[0], ctx->wrap_period[1], ctx->wrap_period[2],
                         ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_slots.as<SlotRec>(), ctx->ws_copies.as<QRec3>(), ctx->ws_copy_meta.as<int2>(), sc,
                         (const unsigned long long *)nullptr, 1, 1, (int *)nullptr, (int2 *)nullptr, PackFused{},
                         ConfirmArgs{});
      hipLaunchKernelGGL(nn_filter_prep_kernel<3>, pgrid, block, 0, st, ctx->ws_copies.as<QRec3>(), sc,
                         ctx->d_absmax.as<unsigned long long>(), (int)n_copies_max, ctx->origin[0], ctx->origin[1],
                         ctx->origin[2], ctx->origin[3], ctx->ws_copies_f.as<QRecF3>());
      hipLaunchKernelGGL(nn_nearest_f32_kernel<3>, sgrid, block, 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                         ctx->nodes[2], ctx->nodes_f[0], ctx->nodes_f[1], ctx->nodes_f[2], ctx->nodes_f[2],
                         ctx->nodes_pp, n_nodes, ctx->ws_copies.as<QRec3>(), ctx->ws_copies_f.as<QRecF3>(),
                         ctx->ws_copy_meta.as<int2>(), ctx->d_absmax.as<unsigned long long>(), n_seg, seg_len,
                         ctx->ws_recs.as<HitRec>(), rec_cap, sc, best_bits, redo);
    }
    hipLaunchKernelGGL(nn_nearest_tie_kernel, dim3(1024), dim3(256), 0, st, ctx->ws_recs.as<HitRec>(), rec_cap, sc,
                       best_bits, best_idx);
    hipLaunchKernelGGL(nn_nearest_out_kernel, grid, block, 0, st, best_bits, best_idx, nq, idx_dev, dist_dev);
    // queries that lost a candidate record (sc->total > rec_cap): answered again, exactly
    NearestIndex ni;
    ni.sx = ctx->sl_d[0]; ni.sy = ctx->sl_d[1]; ni.sz = ctx->sl_d[2]; ni.sw = ctx->sl_d[D == 4 ? 3 : 2];
    ni.sid = ctx->sl_id;
    ni.chunk_ext = reinterpret_cast<const ChunkExt *>(ctx->chunk_ext);
    ni.n_nodes = n_nodes;
    ni.n_chunks = (n_nodes + kSlabChunk - 1) / kSlabChunk;
    const dim3 fgrid((unsigned)(nq < 2048 ? nq : 2048));
    if (D == 4)
      hipLaunchKernelGGL(nn_nearest_fixup_kernel<4>, fgrid, dim3(kFixThreads), 0, st, sc, rec_cap, redo, q_dev, nq,
                         ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1], ctx->wrap_dim[2], ctx->wrap_period[0],
                         ctx->wrap_period[1], ctx->wrap_period[2], ni, idx_dev, dist_dev);
    else
      hipLaunchKernelGGL(nn_nearest_fixup_kernel<3>, fgrid, dim3(kFixThreads), 0, st, sc, rec_cap, redo, q_dev, nq,
                         ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1], ctx->wrap_dim[2], ctx->wrap_period[0],
                         ctx->wrap_period[1], ctx->wrap_period[2], ni, idx_dev, dist_dev);
  }
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

// ------------------------------------------------------------- k nearest -----
// kdFindKNearest (R/kdTree_general.jl:696-723) for a batch, by exact selection: one workgroup
// per query finds the kk-th smallest squared distance with a 6-pass radix select over the bit
// pattern of the unfused fp64 sum (monotone for non-negative doubles; NaN sorts last), gathers
// the kk selected nodes and sorts them by (distance, index).  Where several nodes tie at the
// kk-th distance the lowest indices are taken (the reference's choice there is its tree-visit
// order).  The reference has no caller for this search, so the kernel is exact and
// deterministic rather than tuned: it streams the node arrays seven times per query.
namespace {
constexpr int kKnnMax = 2048;
constexpr int kKnnU = 4;
constexpr int kKnnBins = 2048;
constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;

template <int D>
__device__ __forceinline__ unsigned long long knn_key(const double (&g)[4], const double *__restrict__ nx,
                                                      const double *__restrict__ ny,
                                                      const double *__restrict__ nz,
                                                      const double *__restrict__ nw, int n) {
  double s;
  if constexpr (D == 4) s = sq4(g[0], g[1], g[2], g[3], nx[n], ny[n], nz[n], nw[n]);
  else s = sq3(g[0], g[1], g[2], nx[n], ny[n], nz[n]);
  return (s != s) ? ~0ull : (unsigned long long)__double_as_longlong(s);
}

// sorts the n gathered (key, index) pairs by (distance, index) and writes the first kk as row qi
__device__ void knn_sort_emit(unsigned long long *skey, int *sidx, int n, int kk, int stride, int qi,
                              int32_t *__restrict__ idx_out, double *__restrict__ dist_out,
                              int32_t *__restrict__ count_out) {
  const int tid = threadIdx.x, nt = blockDim.x;
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = n + tid; i < P; i += nt) { skey[i] = ~0ull; sidx[i] = 0x7fffffff; }
  __syncthreads();
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += nt) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long ka = skey[i], kb = skey[l];
          const int ia = sidx[i], ib = sidx[l];
          const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
          const bool up = (i & k2) == 0;
          if (up ? a_gt_b : !a_gt_b) { skey[i] = kb; skey[l] = ka; sidx[i] = ib; sidx[l] = ia; }
        }
      }
      __syncthreads();
    }
  }
  // nodes at a non-finite distance never pass the reference's `newDist < worst` test (:633,:665)
  for (int j = tid; j < stride; j += nt) {
    const bool ok = j < kk && skey[j] < kInfBits;
    idx_out[(size_t)qi * stride + j] = ok ? sidx[j] : -1;
    dist_out[(size_t)qi * stride + j] = ok ? sqrt_rn(__longlong_as_double((long long)skey[j])) : __builtin_inf();
    if (ok && (j + 1 == kk || skey[j + 1] >= kInfBits)) count_out[qi] = j + 1;
  }
  if (tid == 0 && !(skey[0] < kInfBits)) count_out[qi] = 0;
}

// Launched with 256 threads per workgroup for whole batches (throughput) and with 1024 for the few
// queries of a list (one workgroup's speed is set by how many node loads it keeps in flight).
template <int D, int NT>
__global__ __launch_bounds__(NT) void nn_knearest_kernel(
    const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz,
    const double *__restrict__ nw, int n_nodes, const double *__restrict__ q, int kk, int stride,
    int32_t *__restrict__ idx_out, double *__restrict__ dist_out, int32_t *__restrict__ count_out,
    const int *__restrict__ qlist, const int *__restrict__ n_list) {
  __shared__ unsigned hist[kKnnBins];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_rank, s_less, s_ties, s_cnt;
  __shared__ unsigned long long skey[kKnnMax];
  __shared__ int sidx[kKnnMax];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nt = NT;
  int qi = blockIdx.x;
  if (qlist) {                       // only the queries the list names (workgroups past its end leave)
    if (qi >= *n_list) return;
    qi = qlist[qi];
  }
  double g[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < D; ++k) g[k] = q[(size_t)qi * D + k];

  unsigned long long prefix = 0;
  unsigned rank = (unsigned)kk - 1u, less = 0, ties = 0;
  // digits from the top: five of 11 bits, then one of 9 (64 bits in six passes)
  for (int pass = 0; pass < 6; ++pass) {
    const int width = pass < 5 ? 11 : 9;
    const int shift = pass < 5 ? 53 - 11 * pass : 0;
    const unsigned n_bins = 1u << width;
    for (unsigned b = tid; b < n_bins; b += nt) hist[b] = 0;
    __syncthreads();
    for (int base = 0; base < n_nodes; base += nt * kKnnU) {
      unsigned long long key[kKnnU];
      bool valid[kKnnU];
#pragma unroll
      for (int u = 0; u < kKnnU; ++u) {          // kKnnU independent node loads in flight per thread
        const int n = base + u * nt + tid;
        valid[u] = n < n_nodes;
        key[u] = valid[u] ? knn_key<D>(g, nx, ny, nz, nw, n) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < kKnnU; ++u) {
        bool m = valid[u];
        if (pass > 0) m = m && ((key[u] >> (shift + width)) == prefix);
        const unsigned digit = (unsigned)(key[u] >> shift) & (n_bins - 1u);
        // most keys share the leading exponent bits: one add per wave when they agree
        const unsigned long long mask = __ballot(m);
        if (mask == 0) continue;
        const int lead = __ffsll((long long)mask) - 1;
        const unsigned d0 = (unsigned)__shfl((int)digit, lead);
        const unsigned long long same = __ballot(m && digit == d0);
        if (same == mask) {
          if (lane == lead) atomicAdd(&hist[d0], (unsigned)__popcll(mask));
        } else if (m) {
          atomicAdd(&hist[digit], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {                               // wave 0: the bin that holds rank `rank`
      const unsigned per = n_bins / 64u;
      unsigned tot = 0;
      for (unsigned j = 0; j < per; ++j) tot += hist[per * tid + j];
      unsigned inc = tot;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)inc, o);
        if (tid >= o) inc += v;
      }
      unsigned run = inc - tot;
      for (unsigned j = 0; j < per; ++j) {
        const unsigned c = hist[per * tid + j];
        if (rank >= run && rank < run + c) {
          s_prefix = (prefix << width) | (unsigned long long)(per * tid + j);
          s_rank = rank - run;
          s_less = less + run;
          s_ties = c;
        }
        run += c;
      }
    }
    __syncthreads();
    prefix = s_prefix; rank = s_rank; less = s_less; ties = s_ties;
  }
  // prefix = the kk-th smallest key T; `less` keys lie below it, `ties` equal it
  const unsigned long long T = prefix;
  const unsigned need = (unsigned)kk - less;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const bool all_ties = (ties == need);
  for (int base = 0; base < n_nodes; base += nt * kKnnU) {
    unsigned long long key[kKnnU];
#pragma unroll
    for (int u = 0; u < kKnnU; ++u) {
      const int n = base + u * nt + tid;
      key[u] = n < n_nodes ? knn_key<D>(g, nx, ny, nz, nw, n) : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < kKnnU; ++u) {
      const int n = base + u * nt + tid;
      if (n < n_nodes && (key[u] < T || (all_ties && key[u] == T))) {
        const unsigned at = atomicAdd(&s_cnt, 1u);
        skey[at] = key[u]; sidx[at] = n;
      }
    }
  }
  if (!all_ties && tid < 64) {     // more ties than places: lowest indices first
    unsigned got = 0;
    for (int base = 0; base < n_nodes && got < need; base += 64) {
      const int n = base + tid;
      const bool tie = n < n_nodes && knn_key<D>(g, nx, ny, nz, nw, n) == T;
      const unsigned long long mask = __ballot(tie);
      const unsigned r = got + (unsigned)__popcll(mask & ((1ull << tid) - 1ull));
      if (tie && r < need) { skey[less + r] = T; sidx[less + r] = n; }
      got += (unsigned)__popcll(mask);
    }
  }
  knn_sort_emit(skey, sidx, kk, kk, stride, qi, idx_out, dist_out, count_out);
}
// Fast path: the k nearest of a query are the k smallest of its range-search list whenever that list
// (every node with distance < r0, found by the culled search) holds at least kk nodes.  One workgroup
// per query recomputes the exact squared distances of its list entries and sorts them; queries whose
// list is too short (or too long for LDS) are appended to fb_list for the exhaustive kernel above.
template <int D>
__global__ __launch_bounds__(256) void nn_knearest_lists_kernel(
    const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz,
    const double *__restrict__ nw, const double *__restrict__ q, int q_first, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ lidx, int kk, int stride, int32_t *__restrict__ idx_out,
    double *__restrict__ dist_out, int32_t *__restrict__ count_out, int *__restrict__ fb_list,
    int *__restrict__ n_fb) {
  __shared__ unsigned long long skey[kKnnMax];
  __shared__ int sidx[kKnnMax];
  const int tid = threadIdx.x;
  const int qi = q_first + blockIdx.x;
  const long long b = offsets[blockIdx.x], e = offsets[blockIdx.x + 1];
  const int L = (int)(e - b);
  if (L < kk || e - b > kKnnMax) {
    if (tid == 0) fb_list[atomicAdd(n_fb, 1)] = qi;
    return;
  }
  double g[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < D; ++k) g[k] = q[(size_t)qi * D + k];
  for (int j = tid; j < L; j += 256) {
    const int id = lidx[b + j];
    skey[j] = knn_key<D>(g, nx, ny, nz, nw, id);
    sidx[j] = id;
  }
  knn_sort_emit(skey, sidx, L, kk, stride, qi, idx_out, dist_out, count_out);
}

// the kk-th distances of the sampled queries (row qlist[i], column kk - 1)
__global__ void knn_gather_kth_kernel(const double *__restrict__ dist_out, int stride, int kk,
                                      const int *__restrict__ qlist, int n, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dist_out[(size_t)qlist[i] * stride + kk - 1];
}
}  // namespace

// row width of rrtx_nn_knearest: the reference's heap starts with the root and a dummy, so a
// search for k = 1 comes back with two nodes (R/kdTree_general.jl:699-706, 580-593)
int knearest_row(int k, int64_t n_nodes) {
  int64_t kk = k < 2 ? 2 : k;
  return (int)(kk < n_nodes ? kk : n_nodes);
}

static void knearest_exhaustive(rrtx_ctx *ctx, const double *q_dev, int n_blocks, int kk, int stride,
                                int32_t *idx_dev, double *dist_dev, int32_t *count_dev, const int *qlist,
                                const int *n_list) {
  const int n_nodes = (int)ctx->n_nodes;
  const double *nw = ctx->nodes[ctx->dim == 4 ? 3 : 2];
#define RRTX_KNN_LAUNCH(D, NT)                                                                                  \
  hipLaunchKernelGGL((nn_knearest_kernel<D, NT>), dim3(n_blocks), dim3(NT), 0, ctx->stream, ctx->nodes[0],      \
                     ctx->nodes[1], ctx->nodes[2], nw, n_nodes, q_dev, kk, stride, idx_dev, dist_dev, count_dev, \
                     qlist, n_list)
  if (ctx->dim == 4) { if (qlist) RRTX_KNN_LAUNCH(4, 1024); else RRTX_KNN_LAUNCH(4, 256); }
  else { if (qlist) RRTX_KNN_LAUNCH(3, 1024); else RRTX_KNN_LAUNCH(3, 256); }
#undef RRTX_KNN_LAUNCH
}

int launch_nn_knearest(rrtx_ctx *ctx, const double *q_dev, int nq, int k, int32_t *idx_dev, double *dist_dev,
                       int32_t *count_dev) {
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "k-nearest search on an empty tree");
  if (ctx->n_wraps > 0) return fail(ctx, RRTX_E_STATE, "knn search has not been implimented for wrapped space");
  if (k < 1 || k > kKnnMax) return fail(ctx, RRTX_E_INVALID, "nn_knearest: k must be in 1..2048");
  if (nq <= 0) return RRTX_OK;
  const int kk = knearest_row(k, ctx->n_nodes);
  const int stride = k < 2 ? 2 : k;
  const int n_nodes = (int)ctx->n_nodes;
  hipStream_t st = ctx->stream;

  // The range search does the heavy lifting when it pays: big tree (so it culls), a batch worth the
  // extra launches, and lists that fit the LDS sort.  Otherwise every query runs the exhaustive kernel.
  constexpr int kSample = 32;
  const bool culls = ctx->opt_nn_filter && (ctx->opt_nn_cull == 2 || (ctx->opt_nn_cull == 1 && n_nodes >= 8192));
  const bool fast = culls && ctx->opt_knn_lists && nq >= 8 * kSample && kk <= 512 && kk * 16 <= n_nodes;
  if (!fast) {
    span_begin(ctx, KF_NN_NEAREST);
    knearest_exhaustive(ctx, q_dev, nq, kk, stride, idx_dev, dist_dev, count_dev, nullptr, nullptr);
    span_end(ctx);
    RRTX_HIP(ctx, hipGetLastError());
    return RRTX_OK;
  }

  // 1. radius guess: exhaustive search on a strided sample; r0 = twice the median kk-th distance, so a
  //    query in uniform surroundings gets a list of about 8 kk nodes and one at a face of the cloud 4 kk
  RRTX_HIP(ctx, ctx->ws_knn_misc.ensure(sizeof(int) * ((size_t)nq + kSample + 4) + sizeof(double) * kSample));
  double *kth_dev = ctx->ws_knn_misc.as<double>();
  int *sample_dev = reinterpret_cast<int *>(kth_dev + kSample);
  int *n_fb = sample_dev + kSample;          // [0] fallback count, [1] sample count
  int *fb_list = n_fb + 4;
  int sample[kSample];
  for (int i = 0; i < kSample; ++i) sample[i] = (int)((long long)i * nq / kSample);
  const int head[2] = {0, kSample};
  RRTX_HIP(ctx, hipMemcpyAsync(sample_dev, sample, sizeof(sample), hipMemcpyHostToDevice, st));
  RRTX_HIP(ctx, hipMemcpyAsync(n_fb, head, sizeof(head), hipMemcpyHostToDevice, st));
  knearest_exhaustive(ctx, q_dev, kSample, kk, stride, idx_dev, dist_dev, count_dev, sample_dev, n_fb + 1);
  hipLaunchKernelGGL(knn_gather_kth_kernel, dim3(1), dim3(64), 0, st, dist_dev, stride, kk, sample_dev, kSample, kth_dev);
  double kth[kSample];
  RRTX_HIP(ctx, hipMemcpyAsync(kth, kth_dev, sizeof(kth), hipMemcpyDeviceToHost, st));
  RRTX_HIP(ctx, hipStreamSynchronize(st));   // also keeps `sample` / `head` alive long enough
  int n_fin = 0;
  for (int i = 0; i < kSample; ++i)
    if (std::isfinite(kth[i])) kth[n_fin++] = kth[i];
  if (n_fin < kSample / 2) {                 // mostly non-finite queries or nodes: nothing to guess from
    knearest_exhaustive(ctx, q_dev, nq, kk, stride, idx_dev, dist_dev, count_dev, nullptr, nullptr);
    RRTX_HIP(ctx, hipGetLastError());
    return RRTX_OK;
  }
  std::sort(kth, kth + n_fin);
  // (larger k: 1.6 x -> about 4 kk nodes, so that the lists still fit the 2048-entry LDS sort)
  const double r0 = (kk <= 128 ? 2.0 : 1.6) * kth[n_fin / 2];

  // 2. per batch of queries: range search with r0 -> select from the lists -> exhaustive for the rest
  const long long cap_max = 16ll << 20;
  long long per_q = (long long)kk * (kk <= 128 ? 12 : 7);
  int batch = (int)std::min<long long>(nq, std::max<long long>(256, cap_max / per_q));
  for (int first = 0; first < nq; first += batch) {
    const int nb = std::min(batch, nq - first);
    const int64_t cap = (int64_t)nb * per_q + 4096;
    RRTX_HIP(ctx, ctx->ws_knn_off.ensure(sizeof(int64_t) * ((size_t)nb + 2)));
    RRTX_HIP(ctx, ctx->ws_knn_idx.ensure(sizeof(int32_t) * (size_t)cap));
    RRTX_HIP(ctx, ctx->ws_knn_dist.ensure(sizeof(double) * (size_t)cap));
    int64_t *off = ctx->ws_knn_off.as<int64_t>();
    int64_t *needed_dev = off + nb + 1;
    const double *qb = q_dev + (size_t)first * ctx->dim;
    int rc = launch_nn_radius(ctx, qb, nullptr, r0, nb, off, ctx->ws_knn_idx.as<int32_t>(),
                              ctx->ws_knn_dist.as<double>(), cap, needed_dev);
    if (rc) return rc;
    int64_t needed = 0;
    RRTX_HIP(ctx, hipMemcpyAsync(&needed, needed_dev, sizeof(needed), hipMemcpyDeviceToHost, st));
    RRTX_HIP(ctx, hipStreamSynchronize(st));
    span_begin(ctx, KF_NN_NEAREST);
    if (needed > cap) {
      // far denser around these queries than around the sample: the lists were cut off, do not use them
      knearest_exhaustive(ctx, qb, nb, kk, stride, idx_dev + (size_t)first * stride, dist_dev + (size_t)first * stride,
                          count_dev + first, nullptr, nullptr);
    } else {
      RRTX_HIP(ctx, hipMemsetAsync(n_fb, 0, sizeof(int), st));
      if (ctx->dim == 4)
        hipLaunchKernelGGL(nn_knearest_lists_kernel<4>, dim3(nb), dim3(256), 0, st, ctx->nodes[0], ctx->nodes[1],
                           ctx->nodes[2], ctx->nodes[3], q_dev, first, off, ctx->ws_knn_idx.as<int32_t>(), kk, stride,
                           idx_dev, dist_dev, count_dev, fb_list, n_fb);
      else
        hipLaunchKernelGGL(nn_knearest_lists_kernel<3>, dim3(nb), dim3(256), 0, st, ctx->nodes[0], ctx->nodes[1],
                           ctx->nodes[2], ctx->nodes[2], q_dev, first, off, ctx->ws_knn_idx.as<int32_t>(), kk, stride,
                           idx_dev, dist_dev, count_dev, fb_list, n_fb);
      knearest_exhaustive(ctx, q_dev, nb, kk, stride, idx_dev, dist_dev, count_dev, fb_list, n_fb);
    }
    span_end(ctx);
  }
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

int launch_nn_nearest(rrtx_ctx *ctx, const double *q_dev, int nq, int32_t *idx_dev, double *dist_dev, bool exact) {
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "nearest search on an empty tree");
  if (nq <= 0) return RRTX_OK;
  if (exact || !ctx->opt_nn_filter) return launch_nn_nearest_exact(ctx, q_dev, nq, idx_dev, dist_dev);
  return launch_nn_nearest_screened(ctx, q_dev, nq, idx_dev, dist_dev);
}

int launch_nearest_from_lists(rrtx_ctx *ctx, const double *q_dev, int nq, const int64_t *offsets_dev,
                              const int32_t *idx_dev, const double *dist_dev, int32_t *nearest_idx_dev,
                              double *nearest_dist_dev) {
  (void)q_dev;
  if (nq <= 0) return RRTX_OK;
  span_begin(ctx, KF_NN_FINISH);
  hipLaunchKernelGGL(nn_nearest_from_lists_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream,
                     offsets_dev, idx_dev, dist_dev, nq, nearest_idx_dev, nearest_dist_dev);
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

}  // namespace rrtx
