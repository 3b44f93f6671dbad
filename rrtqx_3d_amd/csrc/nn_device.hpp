// nn_device.hpp -- device-side pieces shared by the neighbour-search translation units
// (kernels_nn.hip: range search, kernels_nearest.hip: nearest search, kernels_slab.hip: slab
// index): per-call scalars, hit sink, copy records, the pack / prefilter-prep kernels both
// searches launch, the packed fp32 screen.  Everything here has internal linkage.
#pragma once
#include "collide_device.hpp"
#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {
namespace {


constexpr int kScanU = 4;        // nodes per lane
constexpr int kScanThreads = 256;
constexpr int kChunk = 64 * kScanU;  // nodes per wave per chunk

// device scalars in ws_scalars
struct Scalars {
  unsigned long long total;        // records produced by scan + rootfix
  int n_copies;                    // valid query copies
  int n_units;                     // unused (the culled search reports its chunk visits per wave)
  unsigned long long q_absmax;     // bit pattern of max |coordinate| over the query copies
};

// Where confirmed neighbours go.  count[q] hands out slots of query q's bucket (a returning
// atomic per hit, but on one address per query, so they spread over the L2 channels; a single
// shared counter sustains only ~88 returning atomics/us on MI355X).  A hit beyond the bucket
// capacity goes to the shared overflow list, one atomic per wave.  count[q] ends as the exact
// list length either way.
// one confirmed neighbour in a query's bucket: node index + squared distance in ONE 16-byte store
// (separate 4- and 8-byte arrays cost a partial cache line write each)
// (pad: bit 0 / 1 = the edge sample -> node / node -> sample collides, fused extend() path)
struct alignas(16) BktRec { int32_t idx; int32_t pad; double d2; };

struct HitSink {
  int *count;
  BktRec *bkt;          // [nq][bcap]
  int bcap;
  int pad;
  HitRec *recs;         // overflow list
  long long cap;
  Scalars *sc;          // ->total: entries of the overflow list
};

// arguments of the exact confirmation (see "Rare path of the range scan")
struct ConfirmArgs {
  const double *nx, *ny, *nz, *nw;   // node coordinates by shadow POSITION (fp64)
  const void *copies;                // QRec3 / QRec4, in the order the scan indexes them
  const int2 *meta;
  const SlotRec *slots;
  const int32_t *pos_id;             // node index of a shadow position, null = identity
  int n_slots;
  int pad;
  HitSink hs;
};

// fp32 copy record for the prefilter: 16 B (D=3) / 32 B (D=4), one scalar load
struct alignas(16) QRecF3 { float x, y, z, thr; };
struct alignas(32) QRecF4 { float x, y, z, w, thr, pad0, pad1, pad2; };
template <int D> struct QRecFT;
template <> struct QRecFT<3> { using type = QRecF3; };
template <> struct QRecFT<4> { using type = QRecF4; };

constexpr int kTileQExact = 32;            // query copies per workgroup tile, exact fp64 scan
constexpr int kTileQFilter = 64;           // ... fp32-prefilter scan
constexpr int kScanFU = 8;                 // nodes per lane in the fp32-prefilter scan
constexpr int kQPI = 4;                    // query copies per inner iteration (amortises loop/branch/SMEM overhead)
constexpr int kChunkF = 64 * kScanFU;

// ---------------------------------------------------------------- init ------
// per-call device state in one launch (a 24-byte H2D copy from pageable memory plus a memset were
// three runtime kernels and ~20 us)
__global__ void nn_init_kernel(Scalars *__restrict__ sc, int n_copies_init, int *__restrict__ zero_i32, int n_i32,
                               unsigned long long *__restrict__ fill_u64, int n_u64, unsigned long long v_u64,
                               int *__restrict__ fill_i32, int n_fill_i32, int v_i32,
                               ConfirmArgs *__restrict__ ca_dst, ConfirmArgs ca) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = gridDim.x * blockDim.x;
  if (i == 0) {
    sc->total = 0ull; sc->n_copies = n_copies_init; sc->n_units = 0; sc->q_absmax = 0ull;
    if (ca_dst) *ca_dst = ca;
  }
  for (int k = i; k < n_i32; k += stride) zero_i32[k] = 0;
  for (int k = i; k < n_u64; k += stride) fill_u64[k] = v_u64;
  for (int k = i; k < n_fill_i32; k += stride) fill_i32[k] = v_i32;
}

// ---------------------------------------------------------------- pack ------
// What the radius search folds into the pack pass (all null / zero for the nearest search):
//   * per-call state: count[i] (with the root rule below), the long-list counter, the device
//     copy of the confirmation arguments, and the reset of the scalars the NEXT call will use
//     (the calls alternate between two Scalars records, so no separate init launch is needed);
//   * the root rule: kdFindWithinRange adds the root when distToRoot <= range
//     (R/kdTree_general.jl:896) while every other node needs < range (:830): the scan finds the
//     root only when <, so a root at exactly the range is entered here as the first list entry.
struct PackFused {
  int *count;                      // [nq] list lengths, [nq] = long-list counter
  Scalars *sc_next;
  ConfirmArgs *ca_dst;
  const double *nx, *ny, *nz, *nw; // node arrays (the root is node 0)
  const SphRec *sph;               // fused extend() path: the root entry's two edge flags are decided here
  int m_sph;                       // active spheres, -1: not the fused path
  int root_rule;                   // 0: node 0 of this context is not the tree's root (node-range shard)
};

// Bucket of a query copy.  It only ORDERS the copies (no result depends on it): the tiles of the culled
// search are runs of 16 consecutive copies in bucket order, and a tile's cost is the volume its copies'
// balls span, so a run should be compact in every coordinate the slab index can cull by -- (x, y) cells
// and bins of the third coordinate.  g3 = side of the cubic grid, used when the tree has an extent in the
// third coordinate worth cutting (layers of cells, rows inside a layer and cells inside a row all run
// alternately forwards and backwards, so consecutive buckets are neighbours); g2 = side of the square
// (x, y) grid otherwise (planar trees, time = 0).  At most kMaxQBuckets buckets either way.
struct QGrid { double x0, ix, y0, iy, z0, iz; int sx, sy, sz; };
__device__ __forceinline__ QGrid query_grid(const unsigned long long *__restrict__ xrange, int g2, int g3) {
  QGrid g;
  const double xw = dec_ord(xrange[1]) - dec_ord(xrange[0]), yw = dec_ord(xrange[3]) - dec_ord(xrange[2]);
  const double zw = dec_ord(xrange[5]) - dec_ord(xrange[4]);
  const bool use_z = g3 > 1 && zw > 0.125 * fmax(xw, yw) && zw < 1e300;      // NaN extents: false
  g.sx = use_z ? g3 : g2; g.sy = g.sx; g.sz = use_z ? g3 : 1;
  slab_map(xrange[0], xrange[1], g.sx, &g.x0, &g.ix);
  slab_map(xrange[2], xrange[3], g.sy, &g.y0, &g.iy);
  slab_map(xrange[4], xrange[5], g.sz, &g.z0, &g.iz);
  return g;
}
__device__ __forceinline__ int query_bucket(const QGrid &g, double x, double y, double z) {
  const int cx = slab_of(x, g.x0, g.ix, g.sx), cy = slab_of(y, g.y0, g.iy, g.sy), cz = slab_of(z, g.z0, g.iz, g.sz);
  const int row = cz * g.sy + ((cz & 1) ? (g.sy - 1 - cy) : cy);
  return row * g.sx + ((row & 1) ? (g.sx - 1 - cx) : cx);
}

template <int D>
__global__ void nn_pack_kernel(const double *__restrict__ q, int nq, const double *__restrict__ thr_lt_arr,
                               const double *__restrict__ thr_gt_arr, double thr_lt_s, double thr_gt_s,
                               int n_wraps, int wd0, int wd1, int wd2, double wp0, double wp1, double wp2,
                               double ox, double oy, double oz, double ow,
                               SlotRec *__restrict__ slots, typename QRecT<D>::type *__restrict__ copies,
                               int2 *__restrict__ meta, Scalars *__restrict__ sc,
                               const unsigned long long *__restrict__ xrange, int g2, int g3,
                               int *__restrict__ qhist, int2 *__restrict__ cb, PackFused pf, ConfirmArgs ca) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = i < nq;
  if (i == 0 && pf.count) {
    pf.count[nq] = 0;
    *pf.ca_dst = ca;
    Scalars *nx_sc = pf.sc_next;
    nx_sc->total = 0ull; nx_sc->n_copies = 0; nx_sc->n_units = 0; nx_sc->q_absmax = 0ull;
    if (n_wraps == 0) sc->n_copies = nq;       // nobody counts copies then: one per query
  }
  unsigned long long am = 0ull;                // max |copy - origin| feeds the prefilter's rounding bound
  if (act) {
    // culled scan: copies are bucketed by grid cell over the extent of the node coordinates
    QGrid qg;
    if (qhist) qg = query_grid(xrange, g2, g3);
    double p[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = q[(size_t)i * D + k];
    const double tlt = thr_lt_arr ? thr_lt_arr[i] : thr_lt_s;
    const double tgt = thr_gt_arr ? thr_gt_arr[i] : thr_gt_s;
    const int n_slots = 1 << n_wraps;
    const int wd[3] = {wd0, wd1, wd2};
    const double wp[3] = {wp0, wp1, wp2};
    if (pf.count) {
      const double s = (D == 4) ? sq4(p[0], p[1], p[2], p[3], pf.nx[0], pf.ny[0], pf.nz[0], pf.nw[0])
                                : sq3(p[0], p[1], p[2], pf.nx[0], pf.ny[0], pf.nz[0]);
      const bool add = pf.root_rule && s >= tlt && s < tgt;
      pf.count[i] = add ? 1 : 0;
      if (add) {                               // bcap >= 8: entry 0 of the bucket always exists
        BktRec br;
        br.idx = 0; br.pad = 0; br.d2 = s;
        if (pf.m_sph > 0) {
          // explicitEdgeCheck of sample -> root and root -> sample over the whole list (measure-zero case)
          const double rx = pf.nx[0], ry = pf.ny[0], rz = pf.nz[0], len = sqrt_rn(s);
          bool ho = false, hi = false;
          for (int j = 0; j < pf.m_sph; ++j) {
            const SphRec ob = pf.sph[j];
            ho = ho || edge_hits_sphere(p[0], p[1], p[2], rx - p[0], ry - p[1], rz - p[2], len, ob);
            hi = hi || edge_hits_sphere(rx, ry, rz, p[0] - rx, p[1] - ry, p[2] - rz, len, ob);
          }
          br.pad = (ho ? 1 : 0) | (hi ? 2 : 0);
        }
        ca.hs.bkt[(size_t)i * (size_t)ca.hs.bcap] = br;
      }
    }
    for (int k = 0; k < n_slots; ++k) {
      // ghost k: bit pattern of k, the LAST wrapped dimension is the least
      // significant bit (iteration order of getNextGhostPoint)
      double g[4] = {p[0], p[1], p[2], p[3]};
      double c[4] = {p[0], p[1], p[2], p[3]};
      for (int w = 0; w < n_wraps; ++w) {
        int bit = (k >> (n_wraps - 1 - w)) & 1;
        if (!bit) continue;
        int dimi = wd[w];
        double dim_val = p[dimi];
        double dim_closest = 0.0;
        if (p[dimi] < wp[w] / 2.0) { dim_val += wp[w]; dim_closest += wp[w]; }
        else { dim_val -= wp[w]; }
        g[dimi] = dim_val;
        c[dimi] = dim_closest;
      }
      bool valid = true;
      if (k > 0) {
        // skip when dist(closestUnwrappedPoint, ghost) > range  (R/ghostPoint.jl:104)
        double s = (D == 4) ? sq4(c[0], c[1], c[2], c[3], g[0], g[1], g[2], g[3])
                            : sq3(c[0], c[1], c[2], g[0], g[1], g[2]);
        valid = !(s >= tgt);
      }
      SlotRec sr;
      sr.x = g[0]; sr.y = g[1]; sr.z = g[2]; sr.w = g[3];
      sr.thr_lt = valid ? tlt : -1.0;
      sr.thr_gt = (pf.count && !pf.root_rule) ? tlt : tgt;   // (the root's <= in seen_by_earlier_slot)
      sr.pad0 = 0.0; sr.pad1 = 0.0;
      if (n_wraps > 0) slots[(size_t)i * n_slots + k] = sr;   // only the ghost rules read the table
      if (valid) {
        const double og[4] = {ox, oy, oz, ow};
        for (int c2 = 0; c2 < D; ++c2) am = max(am, (unsigned long long)__double_as_longlong(fabs(g[c2] - og[c2])));
        int pos = (n_wraps == 0) ? i : atomicAdd(&sc->n_copies, 1);
        typename QRecT<D>::type qr;
        qr.x = g[0]; qr.y = g[1]; qr.z = g[2];
        if constexpr (D == 4) { qr.w = g[3]; qr.pad0 = 0.0; qr.pad1 = 0.0; qr.pad2 = 0.0; }
        qr.thr = tlt;
        copies[pos] = qr;
        meta[pos] = make_int2(i, k);
        if (qhist) {
          const int b = query_bucket(qg, g[0], g[1], g[2]);
          cb[pos] = make_int2(b, atomicAdd(&qhist[b], 1));
        }
      }
    }
  }
  // one atomic per wave on the shared maximum
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(am, off);
    am = max(am, o);
  }
  if ((threadIdx.x & 63) == 0 && am != 0ull) atomicMax(&sc->q_absmax, am);
}

// ---------------------------------------------------------------- scan ------
// was this node already discovered by an earlier copy (slot < my slot) of the
// same query?  (addToRangeList keeps the first discovery, R/kdTree_general.jl:765)
template <int D>
__device__ __noinline__ bool seen_by_earlier_slot(const SlotRec *__restrict__ slots, int n_slots, int owner,
                                                  int slot, int node_idx, double x, double y, double z,
                                                  double w) {
  for (int j = 0; j < slot; ++j) {
    SlotRec sr = slots[(size_t)owner * n_slots + j];
    if (sr.thr_lt < 0.0) continue;
    double s = (D == 4) ? sq4(sr.x, sr.y, sr.z, sr.w, x, y, z, w) : sq3(sr.x, sr.y, sr.z, x, y, z);
    double thr = (j == 0 && node_idx == 0) ? sr.thr_gt : sr.thr_lt;  // root uses <=
    if (s < thr) return true;
  }
  return false;
}

// Store a hit whose slot in its query's list is already known: bucket if the slot fits, shared
// overflow list otherwise.  Every lane of the wave calls this together (the overflow branch
// uses a ballot).
__device__ __forceinline__ void place_hit(const HitSink &hs, bool hit, int owner, int slot, int id, double d2,
                                          int flags = 0) {
  const bool inb = hit && slot < hs.bcap;
  if (inb) {
    const size_t at = (size_t)owner * (size_t)hs.bcap + (size_t)slot;
    BktRec br;
    br.idx = id; br.pad = flags; br.d2 = d2;
    hs.bkt[at] = br;
  }
  const bool ov = hit && !inb;
  const unsigned long long m = __ballot(ov);
  if (m != 0ull) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&hs.sc->total, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (ov) {
      const long long pos = (long long)base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < hs.cap) {
        HitRec r;
        r.owner = owner | (flags << 30);     // nq < 2^30 (checked by the launcher): flags ride in the top bits
        r.idx = id; r.d2 = d2;
        hs.recs[pos] = r;
      }
    }
  }
}

// one hit per call: the slot comes from the query's counter.  Whole wave together.
__device__ __forceinline__ void emit_hit(const HitSink &hs, bool hit, int owner, int id, double d2) {
  int slot = 0;
  if (hit) slot = atomicAdd(&hs.count[owner], 1);
  place_hit(hs, hit, owner, slot, id, d2);
}

// Many hits of few queries in one wave (dense balls): one counter update per distinct query.
// Whole wave together.
__device__ __forceinline__ void emit_hits_grouped(const HitSink &hs, bool hit, int owner, int id, double d2,
                                                  int flags = 0) {
  const int lane = threadIdx.x & 63;
  unsigned long long rem = __ballot(hit);
  while (rem != 0ull) {
    const int L = __ffsll((long long)rem) - 1;
    const int o = __builtin_amdgcn_readlane(owner, L);
    const bool mine = hit && owner == o;
    const unsigned long long m = __ballot(mine);
    int base = 0;
    if (lane == L) base = atomicAdd(&hs.count[o], __popcll(m));
    base = __builtin_amdgcn_readlane(base, L);
    place_hit(hs, mine, o, base + __popcll(m & ((1ull << lane) - 1ull)), id, d2, flags);
    rem &= ~m;
  }
}

// emitters for confirm_entry: what to do with a confirmed neighbour
struct GlobalEmit {
  static constexpr bool kNeedsOwner = true;
  static constexpr bool kBatch = false;
  __device__ __forceinline__ void batch(int, int, unsigned, const int *, const double *) const {}
  const HitSink &hs;
  __device__ __forceinline__ void operator()(bool h, int /*q*/, int owner, int id, double d2) const {
    emit_hit(hs, h, owner, id, d2);
  }
};

// ------------------------------------------------------ fp32 prefilter ------
// Conservative screen: a pair may only be DROPPED when it provably fails the
// exact test; pairs that survive are re-tested with the exact unfused fp64
// arithmetic, which alone decides membership ("fp32 prefilter").
//
// Norm expansion on coordinates shifted by the context origin o (P = p - o,
// Q = q - o, p~ = fl32(P), q~ = fl32(Q), eps = 2^-24, C >= max |P_i|, |Q_i|):
//     t = fma(ax, p~x, fma(ay, p~y, fma(az, p~z, pp))),  a = -2 q~,  pp = fl32(|p~|^2)
//   |t - (|p~|^2 - 2 q~.p~)| <= K eps C^2      K = D + sum_{k<=D} (D + 2k)  (24 / 40), used: 26 / 42
//   |q~ - p~| <= |Q - P| + 2 sqrt(D) eps C
// so whenever the exact fp64 s < thr (=> |Q - P| <= R = sqrt(thr)(1 + 1e-15)):
//     t <= (R + 2 sqrt(D) eps C)^2 + K eps C^2 - |q~|^2  =: thr'   (rounded UP to fp32)
// and "t > thr'" proves s >= thr.  Non-finite or huge C disables the screen.
template <int D>
__device__ __forceinline__ typename QRecFT<D>::type make_qrecf(const typename QRecT<D>::type &c, double C, double ox,
                                                               double oy, double oz, double ow) {
  const float qx = (float)(c.x - ox), qy = (float)(c.y - oy), qz = (float)(c.z - oz);
  float qw = 0.f;
  double qq = (double)qx * (double)qx + (double)qy * (double)qy + (double)qz * (double)qz;
  if constexpr (D == 4) { qw = (float)(c.w - ow); qq += (double)qw * (double)qw; }
  float thr_f;
  if (!(C <= 1e15)) {
    thr_f = __builtin_inff();              // non-finite or huge coordinates: screen nothing
  } else if (!(c.thr > 0.0)) {
    thr_f = -__builtin_inff();              // exact test can never pass (s >= 0 >= thr, or thr NaN)
  } else {
    const double eps = 5.9604644775390625e-08;   // 2^-24
    const double K = (D == 4) ? 42.0 : 26.0;
    const double two_sqrt_d = (D == 4) ? 4.0 : 3.4641016151377544;
    const double R = sqrt_rn(c.thr) * (1.0 + 1e-15);
    const double b = R + two_sqrt_d * eps * C * (1.0 + 1e-6);
    const double T = b * b * (1.0 + 1e-12) + K * eps * C * C + 1e-30 - qq * (1.0 - 1e-14);
    thr_f = __double2float_ru(T);
    if (thr_f != thr_f) thr_f = __builtin_inff();
  }
  typename QRecFT<D>::type f;
  f.x = -2.0f * qx; f.y = -2.0f * qy; f.z = -2.0f * qz;
  if constexpr (D == 4) { f.w = -2.0f * qw; f.pad0 = 0.f; f.pad1 = 0.f; f.pad2 = 0.f; }
  f.thr = thr_f;
  return f;
}

template <int D>
__device__ __forceinline__ typename QRecFT<D>::type never_pass_qrecf() {
  typename QRecFT<D>::type f;
  f.x = 0.f; f.y = 0.f; f.z = 0.f;
  if constexpr (D == 4) { f.w = 0.f; f.pad0 = 0.f; f.pad1 = 0.f; f.pad2 = 0.f; }
  f.thr = -__builtin_inff();
  return f;
}

template <int D>
__global__ void nn_filter_prep_kernel(const typename QRecT<D>::type *__restrict__ copies,
                                      const Scalars *__restrict__ sc,
                                      const unsigned long long *__restrict__ node_absmax, int /*n_copies_max*/,
                                      double ox, double oy, double oz, double ow,
                                      typename QRecFT<D>::type *__restrict__ copies_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_copies = sc->n_copies;
  if (i >= n_copies) {
    // pad to a multiple of kQPI with records that never pass (the scan reads kQPI at a time)
    if (i < ((n_copies + kQPI - 1) / kQPI) * kQPI) copies_f[i] = never_pass_qrecf<D>();
    return;
  }
  unsigned long long cb = max(*node_absmax, sc->q_absmax);
  copies_f[i] = make_qrecf<D>(copies[i], __longlong_as_double((long long)cb), ox, oy, oz, ow);
}

constexpr int kCandCap = 192;   // (copy, node) candidates queued in LDS per wave
constexpr int kNearestWarm = 256;   // nodes sampled for the initial bound of the screened nearest scan
typedef float f32x2 __attribute__((ext_vector_type(2)));

// t values of one copy against the lane's 8 nodes: D packed fp32 FMAs per PAIR of
// nodes (v_pk_fma_f32).  Plain v_fma_f32 issues once per 4 cycles per SIMD like
// the fp64 ops; only the packed form reaches the fp32 vector rate.
template <int D>
__device__ __forceinline__ void screen8(const typename QRecFT<D>::type &c, const float *x, const float *y,
                                        const float *z, const float *w, const float *pp, float *t) {
  const f32x2 cx2 = {c.x, c.x}, cy2 = {c.y, c.y}, cz2 = {c.z, c.z};
#pragma unroll
  for (int v = 0; v < kScanFU / 2; ++v) {
    f32x2 a = {pp[2 * v], pp[2 * v + 1]};
    const f32x2 xz = {z[2 * v], z[2 * v + 1]}, xy = {y[2 * v], y[2 * v + 1]}, xx = {x[2 * v], x[2 * v + 1]};
    a = __builtin_elementwise_fma(cz2, xz, a);
    if constexpr (D == 4) {
      const f32x2 cw2 = {c.w, c.w}, xw = {w[2 * v], w[2 * v + 1]};
      a = __builtin_elementwise_fma(cw2, xw, a);
    }
    a = __builtin_elementwise_fma(cy2, xy, a);
    a = __builtin_elementwise_fma(cx2, xx, a);
    t[2 * v] = a.x; t[2 * v + 1] = a.y;
  }
}

// ---------------------------------------------------------- slab index geometry ------
// written by slab_params_kernel at every rebuild: the (x, y) grid the sorted part of the slab index
// is ordered by (cell_of), and inside every cell the Kz equal-width bins of the third coordinate;
// cell_start[c * Kz + b] = first position of bin b of cell c, cell_start[Kx * Ky * Kz] = sl_n_sorted
// (struct SlabParams: rrtx_internal.hpp)
constexpr int kSlabKz = 32;      // bins of the third coordinate inside a cell

// ------------------------------------------------ exact nearest for one point ------
// kdFindNearest for ONE point by a whole workgroup (NT threads, all call it together): expanding
// search over the slab index.  Every chunk whose exact x / y extent overlaps [p - R, p + R] is
// examined node by node with the unfused fp64 distance; a node outside every examined chunk has
// |dx| >= R or |dy| >= R, hence s >= R^2 (1 - 2^-51), so the best examined node is the global
// lexicographic minimum of (s, index) as soon as its s < R^2 (1 - 1e-12).  Otherwise R doubles; with
// R = +inf every non-empty chunk is examined.  Non-finite points skip the culling.  Result for a
// point that orders against nothing (NaN coordinate): (inf, INT_MAX), as the exhaustive scan gives.
// The slab arrays are kept up to date by every append (position = index for the tail), so this
// works whether or not the range search culls.
struct NearestIndex {
  const double *sx, *sy, *sz, *sw;   // fp64 node coordinates by slab POSITION
  const int32_t *sid;                // node index of a position
  const ChunkExt *chunk_ext;
  int n_chunks, n_nodes;
};
struct NearestScratch { double s[16]; int i[16]; int skipped; };

template <int D, int NT>
__device__ __forceinline__ void block_nearest(const NearestIndex &ni, const double px, const double py, const double pz,
                                           const double pw, double r_start, NearestScratch &sm, double &best_s_out,
                                           int &best_i_out) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int NW = NT / 64;
  const bool finite_p = (px - px == 0.0) && (py - py == 0.0) && (pz - pz == 0.0) && (D == 3 || (pw - pw == 0.0));
  double R = (r_start > 0.0 && r_start < 1e300) ? r_start : 1.0;
  if (!finite_p) R = __builtin_inf();
  double best = __builtin_inf();
  int best_i = 0x7fffffff;
  for (int it = 0;; ++it) {
    if (it >= 48) R = __builtin_inf();
    double lo = -__builtin_inf(), hi = __builtin_inf(), ylo = lo, yhi = hi;
    const bool cull = finite_p && R < 1e300;
    if (cull) {
      const double a = px - R, b = px + R, ya = py - R, yb = py + R;
      lo = a - (fabs(a) * 4.5e-16 + 1e-300);
      hi = b + (fabs(b) * 4.5e-16 + 1e-300);
      ylo = ya - (fabs(ya) * 4.5e-16 + 1e-300);
      yhi = yb + (fabs(yb) * 4.5e-16 + 1e-300);
    }
    if (t == 0) sm.skipped = 0;
    __syncthreads();
    best = __builtin_inf();
    best_i = 0x7fffffff;
    bool skipped = false;
    for (int c = wave; c < ni.n_chunks; c += NW) {
      if (cull) {
        const ChunkExt ce = ni.chunk_ext[c];          // wave-uniform
        if (!(dec_ord(ce.xhi) >= lo && dec_ord(ce.xlo) <= hi && dec_ord(ce.yhi) >= ylo && dec_ord(ce.ylo) <= yhi)) {
          skipped = true;
          continue;
        }
      }
#pragma unroll 2
      for (int u = 0; u < kChunkF / 64; ++u) {
        const int pos = c * kChunkF + u * 64 + lane;
        if (pos < ni.n_nodes) {
          double s;
          if constexpr (D == 4) s = sq4(px, py, pz, pw, ni.sx[pos], ni.sy[pos], ni.sz[pos], ni.sw[pos]);
          else s = sq3(px, py, pz, ni.sx[pos], ni.sy[pos], ni.sz[pos]);
          const int id = ni.sid[pos];
          const bool better = (s < best) || (s == best && id < best_i);
          best = better ? s : best;
          best_i = better ? id : best_i;
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(best_i, off);
      if ((ob < best) || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if (lane == 0) {
      sm.s[wave] = best; sm.i[wave] = best_i;
      if (skipped) sm.skipped = 1;
    }
    __syncthreads();
    best = sm.s[0]; best_i = sm.i[0];
    for (int w = 1; w < NW; ++w) {
      const double ob = sm.s[w];
      const int oi = sm.i[w];
      if ((ob < best) || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    const bool any_skipped = sm.skipped != 0;
    __syncthreads();                       // sm is reused by the next round / the caller
    if (!cull || !any_skipped) break;      // everything that can order was examined
    if (best < R * R * (1.0 - 1e-12)) break;
    R = R * 2.0;
  }
  best_s_out = best;
  best_i_out = best_i;
}

// small host helpers shared by the launchers
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

inline int pow2_ceil(long long v) {
  int p = 1;
  while (p < v && p < (1 << 30)) p <<= 1;
  return p;
}

// bring the slab-ordered shadow up to date when the appended tail has grown too long
}  // namespace
}  // namespace rrtx
