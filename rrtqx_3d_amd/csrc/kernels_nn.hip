// kernels_nn.hip -- batched radius search over the node SoA.
// Replaces kdFindWithinRange (R/kdTree_general.jl:800-955) and the ghost-point
// handling for wrapped dimensions (R/ghostPoint.jl:60-111); kdFindNearest lives in
// kernels_nearest.hip, the slab index in kernels_slab.hip, shared device code in
// nn_device.hpp.  gfx950 only: 64-lane waves, scalar (SMEM) query stream, packed
// fp32 screen, exact fp64 confirmation.
//
// Radius search pipeline (all on ctx->stream, no host sync):
//   pack    : queries -> slot table (query + ghosts), copy records, grid bucket and rank of
//             every copy, the root rule (<= instead of <, R/kdTree_general.jl:896 vs :830),
//             per-call state
//   place   : copies in bucket order + their fp32 screen records      (culled search)
//   tile    : per tile of 16 copies: groups of eight nodes within reach (cells of the slab index cut
//             by bins of the third coordinate) -> fp32 screen -> exact confirmation -> hits into the
//             queries' buckets; on the fused extend() path also the sample check and both edge flags
//             of every neighbour                                          (culled search)
//   [ prep, scan, confirm : the same three steps for the brute-force search, every tile of 64
//             copies streams every node ]
//   finish  : offsets, order by node index, dist = sqrt(d2), owner, nearest, empty balls
//             (kernels_finish.hip; offsets + scatter below only after a call that overflowed by much)
#include "collide_device.hpp"
#include "nn_device.hpp"

#include <limits>

namespace rrtx {

namespace {

// ------------------------------------------------------- exact fp64 scan ------
// RRTX_OPT_NN_FILTER = 0: every (copy, node) pair with the reference's arithmetic
template <int D>
__global__ __launch_bounds__(kScanThreads) void nn_scan_kernel(
    const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz,
    const double *__restrict__ nw, int n_nodes, const typename QRecT<D>::type *__restrict__ copies,
    const int2 *__restrict__ meta, const SlotRec *__restrict__ slots, int n_slots, int tile_q, int n_seg,
    int seg_len, const Scalars *__restrict__ sc, HitSink hs) {
  // blocks b and b+8 share an XCD: the node segment is the fast-varying index so
  // each XCD's L2 keeps re-serving the same 1/8th of the node arrays.
  const int seg = blockIdx.x % n_seg;
  const int tile = blockIdx.x / n_seg;
  const int n_copies = sc->n_copies;
  const int q0 = tile * tile_q;
  if (q0 >= n_copies) return;
  const int q1 = min(q0 + tile_q, n_copies);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int node_begin = seg * seg_len;
  const int node_end = min(n_nodes, node_begin + seg_len);
  const double kNaN = __builtin_nan("");

  for (int base = node_begin + wave * kChunk; base < node_end; base += (kScanThreads / 64) * kChunk) {
    double x[kScanU], y[kScanU], z[kScanU], w[kScanU];
    int id[kScanU];
#pragma unroll
    for (int u = 0; u < kScanU; ++u) {
      id[u] = base + u * 64 + lane;
      bool ok = id[u] < node_end;
      x[u] = ok ? nx[id[u]] : kNaN;   // NaN coordinates never compare < thr
      y[u] = ok ? ny[id[u]] : kNaN;
      z[u] = ok ? nz[id[u]] : kNaN;
      if constexpr (D == 4) w[u] = ok ? nw[id[u]] : kNaN; else w[u] = 0.0;
    }
    for (int q = q0; q < q1; ++q) {
      // wave-uniform address: s_load of the 32/64-byte copy record
      const typename QRecT<D>::type c = copies[q];
      double s[kScanU];
      bool h[kScanU];
      bool any = false;
#pragma unroll
      for (int u = 0; u < kScanU; ++u) {
        if constexpr (D == 4) s[u] = sq4(c.x, c.y, c.z, c.w, x[u], y[u], z[u], w[u]);
        else s[u] = sq3(c.x, c.y, c.z, x[u], y[u], z[u]);
        h[u] = s[u] < c.thr;
        any = any || h[u];
      }
      if (__builtin_expect(__ballot(any) != 0ull, 0)) {
        // rare path (k/N of the pairs)
        const int2 m = meta[q];
#pragma unroll
        for (int u = 0; u < kScanU; ++u) {
          bool hu = h[u];
          if (m.y > 0 && hu)
            hu = !seen_by_earlier_slot<D>(slots, n_slots, m.x, m.y, id[u], x[u], y[u], z[u], w[u]);
          emit_hit(hs, hu, m.x, id[u], s[u]);
        }
      }
    }
  }
  (void)lane;
}

// Culled scan: the same records, written in bucket order (query_grid; bucket start + the rank the pack
// kernel drew), together with the fp64 copy and its (query, slot) tag.  Every workgroup first
// scans the whole bucket histogram (<= 4096 counters) into LDS; that is cheaper than a launch.
constexpr int kMaxQBuckets = 4096;
template <int D>
__global__ __launch_bounds__(256) void nn_place_kernel(
    const typename QRecT<D>::type *__restrict__ copies, const int2 *__restrict__ meta, const int2 *__restrict__ cb,
    const int *__restrict__ qhist, int n_buckets, const Scalars *__restrict__ sc,
    const unsigned long long *__restrict__ node_absmax, double ox, double oy, double oz, double ow,
    typename QRecT<D>::type *__restrict__ copies_s, int2 *__restrict__ meta_s,
    typename QRecFT<D>::type *__restrict__ copies_f) {
  __shared__ int start[kMaxQBuckets];
  __shared__ int wsum[4];
  {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // a thread's share of the counters in registers (the buffer holds kMaxQBuckets + 1 ints, zero beyond the
    // buckets in use): four 16-byte loads in flight instead of up to sixteen dependent 4-byte ones, read once
    static_assert(kMaxQBuckets == 256 * 16, "sixteen counters per thread");
    const int per = (n_buckets + 255) / 256;            // <= 16
    const int b0 = min(t * per, n_buckets), b1 = min(b0 + per, n_buckets);
    int h[16];
    if (per == 16) {
      const int4 *h4 = reinterpret_cast<const int4 *>(qhist + b0);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int4 v4 = h4[k]; h[4 * k] = v4.x; h[4 * k + 1] = v4.y; h[4 * k + 2] = v4.z; h[4 * k + 3] = v4.w; }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) h[k] = (b0 + k < b1) ? qhist[b0 + k] : 0;
    }
    int local = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) local += h[k];
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
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (b0 + k < b1) { start[b0 + k] = prefix; prefix += h[k]; }
    __syncthreads();
  }
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_copies = sc->n_copies;
  if (i >= n_copies) {
    if (i < ((n_copies + kQPI - 1) / kQPI) * kQPI) copies_f[i] = never_pass_qrecf<D>();
    return;
  }
  const int2 b = cb[i];
  const int dst = start[b.x] + b.y;
  if ((unsigned)dst >= (unsigned)n_copies) return;   // cannot happen with a consistent histogram; never write outside
  const typename QRecT<D>::type c = copies[i];
  unsigned long long am = max(*node_absmax, sc->q_absmax);
  copies_s[dst] = c;
  meta_s[dst] = meta[i];
  copies_f[dst] = make_qrecf<D>(c, __longlong_as_double((long long)am), ox, oy, oz, ow);
}

// Rare path of the range scan.  The screen ends, per (copy, node chunk), in a 64-bit mask of
// the lanes whose eight nodes were not all screened out.  Each flagged lane records that as an
// ENTRY (copy q, first position p0 of its eight nodes) in its wave's private slice of a global
// buffer: no atomics, no recomputation, the position in the slice is a prefix count over the
// mask.  nn_confirm_kernel then runs one lane per entry: it re-tests the eight nodes with the
// exact unfused fp64 distance and the reference's strict compare (first-discovery rule for
// ghosts).  Nothing of the fp32 screen is reused, so the screen only has to be conservative.
// A slice has room for the worst case of one chunk (every lane of every copy of the tile) on
// top of kEvSlack entries; a wave that cannot guarantee that before its next chunk confirms its
// own entries on the spot (same routine), so nothing is ever dropped.
constexpr int kEvSlack = 2048;

// All lanes of the wave call this together; a lane without an entry passes has = false.
// LM (lane-major chunk layout, slab-ordered shadow): the eight nodes sit at p0 + u, eight
// consecutive doubles per coordinate array; otherwise at p0 + 64 u.
template <int D, bool LM, class Emit>
__device__ __forceinline__ void confirm_entry(bool has, int2 en, int n_nodes, const ConfirmArgs &a, const Emit &emit,
                                              const typename QRecT<D>::type *tile_copies = nullptr, int tile_q0 = 0) {
  const typename QRecT<D>::type *__restrict__ copies = static_cast<const typename QRecT<D>::type *>(a.copies);
  typename QRecT<D>::type ce;
  ce.x = 0.0; ce.y = 0.0; ce.z = 0.0; ce.thr = 0.0;
  if constexpr (D == 4) ce.w = 0.0;
  int2 m = make_int2(0, 0);
  const int p0 = has ? en.y : 0;
  if (has) {
    // the tile kernel keeps its 16 copies in LDS: one dependent global load less per entry
    ce = tile_copies ? tile_copies[en.x - tile_q0] : copies[en.x];
    // (owner, slot): the slot matters only with ghosts; emitters that want the owner for every hit say so
    if (Emit::kNeedsOwner || a.n_slots > 1) m = a.meta[en.x];
  }
  // the lane's eight nodes in two halves of four: half the registers, and the exact part is a fraction
  // of the kernel's instructions anyway
  constexpr int kH = kScanFU / 2;
#pragma unroll 1
  for (int hv = 0; hv < 2; ++hv) {
    double ex[kH], ey[kH], ez[kH], ew[kH];
    int nid[kH];                      // node index of each position, requested with the rows (not per confirmed hit)
    if constexpr (LM) {
      if (a.pos_id) {
        const int4 vi = *reinterpret_cast<const int4 *>(a.pos_id + p0 + kH * hv);
        nid[0] = vi.x; nid[1] = vi.y; nid[2] = vi.z; nid[3] = vi.w;
      } else {
#pragma unroll
        for (int u = 0; u < kH; ++u) nid[u] = p0 + kH * hv + u;
      }
      // node arrays are allocated in whole chunks: the 32-byte rows are always readable
      const double2 *rx = reinterpret_cast<const double2 *>(a.nx + p0 + kH * hv);
      const double2 *ry = reinterpret_cast<const double2 *>(a.ny + p0 + kH * hv);
      const double2 *rz = reinterpret_cast<const double2 *>(a.nz + p0 + kH * hv);
      const double2 *rw = reinterpret_cast<const double2 *>(a.nw + p0 + kH * hv);
#pragma unroll
      for (int v = 0; v < kH / 2; ++v) {
        const double2 vx = rx[v], vy = ry[v], vz = rz[v];
        ex[2 * v] = vx.x; ex[2 * v + 1] = vx.y;
        ey[2 * v] = vy.x; ey[2 * v + 1] = vy.y;
        ez[2 * v] = vz.x; ez[2 * v + 1] = vz.y;
        if constexpr (D == 4) { const double2 vw = rw[v]; ew[2 * v] = vw.x; ew[2 * v + 1] = vw.y; }
        else { ew[2 * v] = 0.0; ew[2 * v + 1] = 0.0; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < kH; ++u) {
        const int pos = p0 + 64 * (kH * hv + u);
        const int pc = (pos < n_nodes) ? pos : 0;
        ex[u] = a.nx[pc]; ey[u] = a.ny[pc]; ez[u] = a.nz[pc];
        if constexpr (D == 4) ew[u] = a.nw[pc]; else ew[u] = 0.0;
        nid[u] = pos;                 // (index-order arrays: position = node index)
      }
    }
    double s[kH];
    unsigned hm = 0u;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const int pos = LM ? p0 + kH * hv + u : p0 + 64 * (kH * hv + u);
      if constexpr (D == 4) s[u] = sq4(ce.x, ce.y, ce.z, ce.w, ex[u], ey[u], ez[u], ew[u]);
      else s[u] = sq3(ce.x, ce.y, ce.z, ex[u], ey[u], ez[u]);
      bool hu = has && pos < n_nodes && (s[u] < ce.thr);
      if (m.y > 0 && hu)
        hu = !seen_by_earlier_slot<D>(a.slots, a.n_slots, m.x, m.y, nid[u], ex[u], ey[u], ez[u], ew[u]);
      hm |= (hu ? 1u : 0u) << u;
    }
    if constexpr (Emit::kBatch) {
      // the lane's hits of this half go out together (one slot request per lane)
      static_assert(kH == 4, "batch emitters take four nodes at a time");
      emit.batch(en.x, m.x, hm, nid, s);
    } else {
      // normally one of the eight is a neighbour; emit them one per round
      while (__ballot(hm != 0u) != 0ull) {
        const bool h = hm != 0u;
        const int uu = h ? (__ffs((int)hm) - 1) : 0;
        hm &= hm - 1u;
        double hs2 = s[0];
        int hid = nid[0];
#pragma unroll
        for (int u = 1; u < kH; ++u) { hs2 = (uu == u) ? s[u] : hs2; hid = (uu == u) ? nid[u] : hid; }
        emit(h, en.x, m.x, hid, hs2);
      }
    }
  }
}

// One lane per entry.  kConfirmParts waves share a slice (rounds of 64 entries dealt round-robin)
// so that the few slices holding several times the average do not set the kernel's duration.
constexpr int kConfirmParts = 4;
template <int D, bool LM>
__global__ __launch_bounds__(256) void nn_confirm_kernel(ConfirmArgs a, const int2 *__restrict__ ev,
                                                         const int *__restrict__ ev_cnt, int n_slices,
                                                         int slice_cap, int n_nodes) {
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
  const int w = gw / kConfirmParts, part = gw % kConfirmParts;
  if (w >= n_slices) return;
  const int cnt = ev_cnt[w];
  const int2 *__restrict__ mine = ev + (size_t)w * (size_t)slice_cap;
  for (int e0 = 64 * part; e0 < cnt; e0 += 64 * kConfirmParts) {
    const int e = e0 + lane;
    const bool has = e < cnt;
    int2 en = make_int2(0, 0);
    if (has) en = mine[e];
    confirm_entry<D, LM>(has, en, n_nodes, a, GlobalEmit{a.hs});
  }
}


// One chunk of kChunkF nodes (8 per lane, held in VGPRs) against the copies [q0, q1):
// the hot loop of the range search.  LM selects the lane-major chunk layout of the slab-ordered
// shadow.  Flagged lanes file (copy, first position) entries at mine[wn...] (see "Rare path").
// MODE 0: the lane's nodes are base + lane + 64 u (index-order arrays); 1: lane-major chunk of the slab
// index, positions base + 8 lane .. + 7; 2: lane-major GROUPS, every lane its own eight consecutive
// positions p_lane .. + 7 (lanes outside the wave mask vm hold nothing and file nothing).
template <int D, int MODE>
__device__ __forceinline__ void scan_chunk_f32(const float *__restrict__ fx, const float *__restrict__ fy,
                                               const float *__restrict__ fz, const float *__restrict__ fw,
                                               const float *__restrict__ fpp, const int base, const int node_end,
                                               const typename QRecFT<D>::type *__restrict__ copies_f, const int q0,
                                               const int q1, int2 *__restrict__ mine, int &wn,
                                               const unsigned p_lane = 0u, const unsigned long long vm = ~0ull) {
  const int lane = threadIdx.x & 63;
  const float kInf = __builtin_inff();
  constexpr bool LM = MODE != 0;
  float x[kScanFU], y[kScanFU], z[kScanFU], w[kScanFU], pp[kScanFU];
  const unsigned p0 = (MODE == 2) ? p_lane : (unsigned)(base + 8 * lane);
  if constexpr (LM) {
    // lane-major: two 16-byte loads per array (the arrays are allocated in whole chunks, so the
    // rows of the last chunk are readable; what lies beyond node_end gets pp = +inf)
    const float4 *rx = reinterpret_cast<const float4 *>(fx + p0);
    const float4 *ry = reinterpret_cast<const float4 *>(fy + p0);
    const float4 *rz = reinterpret_cast<const float4 *>(fz + p0);
    const float4 *rw = reinterpret_cast<const float4 *>(fw + p0);
    const float4 *rp = reinterpret_cast<const float4 *>(fpp + p0);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float4 vx = rx[v], vy = ry[v], vz = rz[v], vp = rp[v];
      x[4 * v] = vx.x; x[4 * v + 1] = vx.y; x[4 * v + 2] = vx.z; x[4 * v + 3] = vx.w;
      y[4 * v] = vy.x; y[4 * v + 1] = vy.y; y[4 * v + 2] = vy.z; y[4 * v + 3] = vy.w;
      z[4 * v] = vz.x; z[4 * v + 1] = vz.y; z[4 * v + 2] = vz.z; z[4 * v + 3] = vz.w;
      pp[4 * v] = vp.x; pp[4 * v + 1] = vp.y; pp[4 * v + 2] = vp.z; pp[4 * v + 3] = vp.w;
      if constexpr (D == 4) {
        const float4 vw = rw[v];
        w[4 * v] = vw.x; w[4 * v + 1] = vw.y; w[4 * v + 2] = vw.z; w[4 * v + 3] = vw.w;
      } else {
        w[4 * v] = 0.f; w[4 * v + 1] = 0.f; w[4 * v + 2] = 0.f; w[4 * v + 3] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kScanFU; ++u)
      if ((int)p0 + u >= node_end) { x[u] = 0.f; y[u] = 0.f; z[u] = 0.f; w[u] = 0.f; pp[u] = kInf; }
  } else {
#pragma unroll
    for (int u = 0; u < kScanFU; ++u) {
      const int id = base + u * 64 + lane;
      const bool ok = id < node_end;
      const unsigned idc = (unsigned)(ok ? id : node_end - 1);   // clamped (no divergent load), unsigned
                                                                  // 32-bit index -> global_load saddr form
      x[u] = fx[idc];
      y[u] = fy[idc];
      z[u] = fz[idc];
      if constexpr (D == 4) w[u] = fw[idc]; else w[u] = 0.f;
      pp[u] = ok ? fpp[idc] : kInf;        // +inf padding: t = +inf never survives a finite bound
    }
  }
  for (int q = q0; q < q1; q += kQPI) {
    // kQPI wave-uniform copy records per iteration (scalar loads); copies_f is padded to a
    // multiple of kQPI records that never pass
    typename QRecFT<D>::type c[kQPI];
#pragma unroll
    for (int k = 0; k < kQPI; ++k) c[k] = copies_f[q + k];
    unsigned long long mk[kQPI];      // wave masks stay on the SALU (no short-circuit control flow)
    unsigned long long anym = 0ull;
#pragma unroll
    for (int k = 0; k < kQPI; ++k) {
      float t[kScanFU];
      screen8<D>(c[k], x, y, z, w, pp, t);
      const float m1 = fminf(fminf(t[0], t[1]), t[2]);          // v_min3_f32 x3 + v_min_f32
      const float m2 = fminf(fminf(t[3], t[4]), t[5]);
      const float m3 = fminf(fminf(t[6], t[7]), m1);
      const float tmin = fminf(m2, m3);
      mk[k] = __ballot(!(tmin > c[k].thr));
      if constexpr (MODE == 2) mk[k] &= vm;
      anym |= mk[k];
    }
    if (anym != 0ull) {
      // some lane of some copy was not screened out: every flagged lane files its own entry
#pragma unroll
      for (int k = 0; k < kQPI; ++k) {
        if (mk[k] == 0ull || q + k >= q1) continue;
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(mk[k] >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)mk[k], 0u));
        if ((mk[k] >> lane) & 1ull)
          mine[wn + (int)before] = make_int2(q + k, LM ? (int)p0 : base + lane);
        wn += __popcll(mk[k]);
      }
    }
  }
}

// a wave confirms the entries of its own slice (slice nearly full; same routine as nn_confirm_kernel)
template <int D, bool LM, class Emit>
__device__ __forceinline__ void drain_slice_to(const int2 *__restrict__ mine, int &wn, int n_nodes, const ConfirmArgs &a,
                                               const Emit &emit, const typename QRecT<D>::type *tile_copies, int tile_q0) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // see drain_slice
  for (int e0 = 0; e0 < wn; e0 += 64) {
    const int e = e0 + lane;
    const bool has = e < wn;
    int2 en = make_int2(0, 0);
    if (has) en = mine[e];
    confirm_entry<D, LM>(has, en, n_nodes, a, emit, tile_copies, tile_q0);
  }
  wn = 0;
}

template <int D, bool LM>
__device__ __forceinline__ void drain_slice(const int2 *__restrict__ mine, int &wn, int n_nodes,
                                            const ConfirmArgs *__restrict__ ca) {
  const int lane = threadIdx.x & 63;
  // the wave's entry stores -> visible to its own loads (workgroup scope: no cache maintenance,
  // an agent-scope fence would write back / invalidate L2 across the XCDs)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const ConfirmArgs a = *ca;
  for (int e0 = 0; e0 < wn; e0 += 64) {
    const int e = e0 + lane;
    const bool has = e < wn;
    int2 en = make_int2(0, 0);
    if (has) en = mine[e];
    confirm_entry<D, LM>(has, en, n_nodes, a, GlobalEmit{a.hs});
  }
  wn = 0;
}

// Unculled range scan: every tile of copies streams every node.
template <int D>
__global__ __launch_bounds__(kScanThreads, 5) void nn_scan_f32_kernel(
    const float *__restrict__ fx, const float *__restrict__ fy, const float *__restrict__ fz,
    const float *__restrict__ fw, const float *__restrict__ fpp, int n_nodes,
    const typename QRecFT<D>::type *__restrict__ copies_f, int tile_q, int n_seg, int seg_len,
    const Scalars *__restrict__ sc, int2 *__restrict__ ev, int *__restrict__ ev_cnt, int slice_cap,
    const ConfirmArgs *__restrict__ ca) {
  // Persistent workgroups; waves never synchronise.  Block b walks items b, b + gridDim.x, ...
  // where item = tile * n_seg + seg.  gridDim.x and n_seg are multiples of 8, so a block keeps
  // the same (item % 8) class: blocks b and b+8 share an XCD, hence each XCD's L2 keeps serving
  // the same node segments.  One chunk can raise at most 64 tile_q entries.
  const int n_copies = sc->n_copies;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (kScanThreads / 64) + wave);
  int2 *__restrict__ mine = ev + (size_t)slice * (size_t)slice_cap;
  int wn = 0;                               // wave-uniform: entries in the slice
  const int n_tiles = (n_copies + tile_q - 1) / tile_q;
  const int n_items = n_tiles * n_seg;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int seg = item % n_seg;
    const int tile = item / n_seg;
    const int q0 = tile * tile_q;
    const int q1 = min(q0 + tile_q, n_copies);
    const int node_begin = seg * seg_len;
    const int node_end = min(n_nodes, node_begin + seg_len);
    for (int base = node_begin + wave * kChunkF; base < node_end; base += (kScanThreads / 64) * kChunkF) {
      if (wn + 64 * tile_q > slice_cap) drain_slice<D, false>(mine, wn, n_nodes, ca);
      scan_chunk_f32<D, 0>(fx, fy, fz, fw, fpp, base, node_end, copies_f, q0, q1, mine, wn);
    }
  }
  if (lane == 0) ev_cnt[slice] = wn;
}

// ---------------------------------------------------------- tile kernel ------
// Culled range search, one workgroup per (tile of kTileB bucket-ordered copies, part):
//   1. the tile's reach in x, y and the third coordinate: every copy's coordinate -+ its search radius,
//      rounded outwards;
//   2. the nodes that can hold a neighbour.  The sorted part of the slab index is ordered by (x, y) grid cell
//      (rows of cells running alternately left and right) and inside a cell by bin of the third coordinate,
//      so the positions of a cell within reach that can matter are ONE run: wave 0 (lane = cell) reads two
//      entries of the cell-start table and lists the GROUPS of eight positions (one scan lane's nodes) the
//      runs touch, each group once (step 2a below).  A tile whose groups do not fit its list falls back to
//      whole chunks: one contiguous run of positions per row of cells.  Chunks of the appended tail are few and
//      are tested by extent.  A listed chunk is still skipped when its own exact x / y extent misses the reach
//      (a node outside the reach in one coordinate is out of range of every copy of the tile: |x_q - x_n| > R
//      implies that square alone is >= thr after rounding, and the other squares only add, exact_math.hpp sq3);
//      part p of a tile takes units u = p mod n_parts;
//   3. the screen of the list, 64 groups (or one chunk) per wave at a time (scan_chunk_f32), entries into
//      the waves' slices;
//   4. exact confirmation of the workgroup's entries, hits collected per copy in LDS;
//   5. one counter update per copy, then the copy's hits go to its query's bucket in one piece
//      (16-byte records, consecutive slots).
constexpr int kTileB = 16;            // copies per tile
constexpr int kTbLcap = 96;           // hits per copy collected in LDS (more: straight to the bucket)
constexpr int kTbList = 1024;         // chunk ids per pass of step 2 (tail chunks; whole-chunk fallback)
constexpr int kTbSlack = 512;

constexpr int kTbGroups = 1024;       // 8-position groups a tile can list (more: whole chunks, step 2)
template <int D>
struct TileLds {
  int list[kTbList];
  int n_list;
  int glist[kTbGroups];
  int n_groups, gmode;
  double zlo, zhi;
  int wcnt[kScanThreads / 64];
  double lo, hi, ylo, yhi;
  int lcnt[kTileB];
  typename QRecT<D>::type cp[kTileB];
  BktRec hrec[kTileB][kTbLcap];
  // fused extend() path: per copy (= sample) the spheres its candidate edges can touch, and whether
  // the sample itself is in collision
  int snl[kTileB];
  int sbad[kTileB];
  int sok[kTileB];             // the sample's coordinates are finite and moderate (screens apply)
  float4 srp[kTileB];          // the sample's probe of the fp32 reach table (centre, inflated ball radius)
  double sbase[kTileB];        // ball radius + slack of the exact list test
  int sq[2][64];               // (sample, sphere) pairs the screen left over, one queue per sample-pass wave
  int ssl[kTileB][kSphListCap];
  SphRec ssr[kTileB][kSphListCap];   // ... and their records, fetched by the sample pass (the hand-out's edge tests
                                     //     then wait for the neighbour's coordinates only)
};

// the (x, y) cell structure of the sorted part of the slab index
struct TileGrid {
  const SlabParams *sp;
  const int *cell_start;       // [Kx * Ky * Kz + 1]
  int n_sorted_chunks;         // chunks [0, n_sorted_chunks) hold sorted positions only
  int kz;                      // bins of the third coordinate per cell (stride of cell_start)
  int groups;                  // list 8-position groups (cut by the third coordinate too) instead of chunks
};

// EXT: every confirmed neighbour is a candidate edge of extend().  Both directed edges are checked
// against the sample's sphere list when the tile hands its hits to the buckets (tile_edge_flags) and
// the two booleans travel in the record's spare word.
// Most samples have no sphere within reach of their ball (sok: finite, moderate coordinates): an
// edge inside the ball then collides with nothing and needs neither the node's coordinates nor a
// test.  (d2 finite and the sample finite => the node is finite; len <= r_bound => the list covers it.)
template <int D>
__device__ __forceinline__ bool tile_edge_needed(const ExtendDev &x, const TileLds<D> &sm, int cl, double d2) {
  const double len = sqrt_rn(d2);
  const bool quick = sm.snl[cl] == 0 && sm.sok[cl] != 0 && len > 0.0 && len <= x.r_bound && len < 1e29;
  return !quick && x.m > 0;
}
// the two flags of one neighbour whose node record nd has been fetched; all lanes of the wave together
template <int D>
__device__ __forceinline__ int tile_edge_eval(const ExtendDev &x, const TileLds<D> &sm, bool need, int cl,
                                              const double4 nd, double d2) {
  if (__ballot(need) == 0ull) return 0;
  const typename QRecT<D>::type c = sm.cp[cl];
  bool ho, hi;
  edge_flags(x, need, c.x, c.y, c.z, nd.x, nd.y, nd.z, sqrt_rn(d2), sm.snl[cl], sm.ssl[cl], ho, hi, sm.ssr[cl]);
  return (ho ? 1 : 0) | (hi ? 2 : 0);
}
template <int D>
__device__ __forceinline__ int tile_edge_flags(const ExtendDev &x, const TileLds<D> &sm, bool h, int cl, int id,
                                               double d2) {
  const bool need = h && tile_edge_needed<D>(x, sm, cl, d2);
  if (__ballot(need) == 0ull) return 0;
  double4 nd = make_double4(0.0, 0.0, 0.0, 0.0);
  if (need) nd = x.naos[id];
  return tile_edge_eval<D>(x, sm, need, cl, nd, d2);
}

template <int D, bool EXT>
struct TileEmit {
  static constexpr bool kNeedsOwner = false;
  static constexpr bool kBatch = true;
  // up to four hits of one lane (bit u of hm: node id[u] at squared distance d2[u]): ONE slot request for all of
  // them, then each goes to its slot of the copy's LDS list -- no loop over the busiest lane's hits, no selects
  __device__ __forceinline__ void batch(int q, int /*owner*/, unsigned hm, const int *id, const double *d2) const {
    const int cl = hm ? q - q0 : 0;
    int base = 0;
    if (hm) base = atomicAdd(&sm.lcnt[cl], __popc(hm));
    unsigned spill = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool b = (hm >> u) & 1u;
      if (__ballot(b) == 0ull) continue;
      const int slot = base + __popc(hm & ((1u << u) - 1u));
      if (b && slot < kTbLcap) {
        BktRec br;
        br.idx = id[u]; br.pad = 0; br.d2 = d2[u];
        sm.hrec[cl][slot] = br;
      }
      spill |= (b && slot >= kTbLcap) ? (1u << u) : 0u;
    }
    if (__ballot(spill != 0u) != 0ull) {
      // a dense ball (more than kTbLcap hits of one copy in one tile): straight to the bucket
#pragma unroll 1
      for (int u = 0; u < 4; ++u) {
        const bool sp = (spill >> u) & 1u;
        if (__ballot(sp) == 0ull) continue;
        int owner = 0, flags = 0;
        if (sp) owner = meta[q].x;
        if constexpr (EXT) flags = tile_edge_flags<D>(x, sm, sp, cl, id[u], d2[u]);
        emit_hits_grouped(hs, sp, owner, id[u], d2[u], flags);   // at most kTileB queries per wave
      }
    }
  }
  TileLds<D> &sm;
  const HitSink &hs;
  const int2 *meta;
  const ExtendDev &x;
  int q0;
  __device__ __forceinline__ void operator()(bool h, int q, int /*owner*/, int id, double d2) const {
    const int cl = h ? q - q0 : 0;
    int slot = 0;
    if (h) slot = atomicAdd(&sm.lcnt[cl], 1);
    const bool in_lds = h && slot < kTbLcap;
    if (in_lds) {
      BktRec br;
      br.idx = id; br.pad = 0; br.d2 = d2;
      sm.hrec[cl][slot] = br;
    }
    const bool spill = h && !in_lds;
    if (__ballot(spill) != 0ull) {
      // a dense ball (more than kTbLcap hits of one copy in one tile): straight to the bucket
      int owner = 0, flags = 0;
      if (spill) owner = meta[q].x;
      if constexpr (EXT) flags = tile_edge_flags<D>(x, sm, spill, cl, id, d2);
      emit_hits_grouped(hs, spill, owner, id, d2, flags);   // at most kTileB queries per wave
    }
  }
};

// Wave shuffles whose source-lane arithmetic is redone where it is used: `l` comes from lane_here(), which the
// compiler cannot see through, so the twelve bpermute addresses of the xor / up patterns are not hoisted out of
// the tile loop and kept (or spilled: 13 dwords per lane, written by every wave at kernel start) for the whole kernel.
__device__ __forceinline__ int lane_here() {
  int l = (int)(threadIdx.x & 63);
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ int wshfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
// Inclusive scans over the 64 lanes of a wave with DPP moves (row shifts 1, 2, 4, 8, then the row broadcasts of
// GFX9): a few cycles per step where a ds_bpermute step is a trip through the LDS pipe.  All lanes active.
template <bool MAX>
__device__ __forceinline__ int wave_scan_incl(int v, const int identity) {
#define RRTX_SCAN_STEP(ctrl, rows)                                                              \
  {                                                                                             \
    const int o = __builtin_amdgcn_update_dpp(identity, v, ctrl, rows, 0xf, false);             \
    v = MAX ? max(v, o) : v + o;                                                                \
  }
  RRTX_SCAN_STEP(0x111, 0xf) RRTX_SCAN_STEP(0x112, 0xf) RRTX_SCAN_STEP(0x114, 0xf) RRTX_SCAN_STEP(0x118, 0xf)
  RRTX_SCAN_STEP(0x142, 0xa) RRTX_SCAN_STEP(0x143, 0xc)
#undef RRTX_SCAN_STEP
  return v;
}
// a double from the lane N places up in the same row of 16 lanes (row_shl:N; lanes without such a lane keep their own)
template <int N>
__device__ __forceinline__ double row_up_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)(b & 0xffffffffll), (int)(b & 0xffffffffll), 0x100 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0x100 + N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// the value of the lane before (lane 0: `first`); wave_shr:1
__device__ __forceinline__ int wave_prev(int v, const int first) {
  return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ double wshfl_d(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// Measuring build (python -m rrtqx_3d_amd.build --clocks -> librrtx_hip_clk.so, tools/tile_clocks.py): thread 0
// of every workgroup (and the first lane of waves 1, 2 where they finish their part of the list phase) leaves
// the 100 MHz wall clock at the phase boundaries of its (first) tile.
#ifdef RRTX_TILE_CLOCKS
__device__ unsigned long long g_tile_clk[4096 * 16];
#define RRTX_TILE_CLK_AT(thread, k) do { if (t == (thread)) g_tile_clk[(blockIdx.x & 4095) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define RRTX_TILE_CLK_AT(thread, k) do { } while (0)
#endif
#define RRTX_TILE_CLK(k) RRTX_TILE_CLK_AT(0, k)

template <int D, bool EXT>
__global__ __launch_bounds__(kScanThreads, 4) void nn_tile_kernel(
    const float *__restrict__ fx, const float *__restrict__ fy, const float *__restrict__ fz,
    const float *__restrict__ fw, const float *__restrict__ fpp, int n_nodes, int n_chunks,
    const ChunkExt *__restrict__ chunk_ext, const typename QRecT<D>::type *__restrict__ copies_s, const typename QRecFT<D>::type *__restrict__ copies_f,
    const Scalars *__restrict__ sc, int n_parts, int2 *__restrict__ ev, int slice_cap,
    const ConfirmArgs a, const ConfirmArgs *__restrict__ ca, const TileGrid tg, const ExtendDev x,
    int *__restrict__ visits, const int n_copies_known) {
  __shared__ TileLds<D> sm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int part = blockIdx.x % n_parts;
  // (without ghosts the host knows the number of copies: no device word to wait for before the first tile)
  const int n_copies = n_copies_known >= 0 ? n_copies_known : sc->n_copies;
  const int n_tiles = (n_copies + kTileB - 1) / kTileB;
  const int slice = (int)blockIdx.x * (kScanThreads / 64) + wave;
  int2 *__restrict__ mine = ev + (size_t)slice * (size_t)slice_cap;
  const int2 *__restrict__ blk = ev + (size_t)blockIdx.x * (kScanThreads / 64) * (size_t)slice_cap;
  int visited = 0;                          // groups of eight nodes this wave screened (statistics)

  // Workgroups are dealt to the 8 XCDs round-robin (b and b + 8 share an XCD and its L2).  Tiles are in
  // cell order, so neighbouring tiles stream the same chunks: give each XCD one contiguous eighth of the
  // tile range instead of every eighth tile, and its L2 keeps serving the chunks of that eighth.
  const int n_slots = (int)gridDim.x / n_parts;            // tiles in flight per sweep of the grid
  const int slot = (int)blockIdx.x / n_parts;
  const int my = (n_slots % 8 == 0) ? (slot % 8) * (n_slots / 8) + slot / 8 : slot;   // a permutation of the slots
  for (int tile = my; tile < n_tiles; tile += n_slots) {
    const int q0 = tile * kTileB;
    const int q1 = min(q0 + kTileB, n_copies);
    int wn = 0;                             // wave-uniform: entries in the slice

    // ---- 1. reach ----
    RRTX_TILE_CLK(0);
    if (t < kTileB) sm.lcnt[t] = 0;
    if (wave == 0) {
      double lo = __builtin_inf(), hi = -__builtin_inf(), ylo = __builtin_inf(), yhi = -__builtin_inf();
      double zlo = __builtin_inf(), zhi = -__builtin_inf();
      if (q0 + lane < q1) {
        const typename QRecT<D>::type c = copies_s[q0 + lane];
        sm.cp[lane] = c;
        if constexpr (EXT) {
          const double rb = x.r_bound >= 0.0 ? x.r_bound : 0.0;
          const double pmax = fmax(fmax(fabs(c.x), fabs(c.y)), fabs(c.z));
          // slack for the rounding of the foot point and of this distance; NaN / inf sample: everything is a candidate
          const double base_b = rb + 1e-12 * (pmax + 1.0);
          const bool usable = (pmax < 1e29) && (c.x == c.x) && (c.y == c.y) && (c.z == c.z) && (base_b < 1e29);
          const ReachProbe rp = reach_probe(x, c.x, c.y, c.z, base_b, usable);
          sm.srp[lane] = make_float4(rp.mx, rp.my, rp.mz, rp.h);
          sm.sbase[lane] = base_b;
          sm.sok[lane] = (usable && x.r_bound >= 0.0) ? 1 : 0;
          sm.snl[lane] = 0; sm.sbad[lane] = 0;
        }
        // thr NaN / <= 0, x or y NaN or +-inf: the copy can never have a neighbour
        // (a non-finite third coordinate gives s = NaN or inf, which is never < thr)
        const bool can_hit = (c.thr > 0.0) && (c.x - c.x == 0.0) && (c.y - c.y == 0.0) && (c.z - c.z == 0.0);
        if (can_hit) {
          const double R = sqrt_rn(c.thr) * (1.0 + 1e-15);         // thr = +inf -> R = +inf
          const double xa = c.x - R, xb = c.x + R, ya = c.y - R, yb = c.y + R, za = c.z - R, zb = c.z + R;
          lo = xa - (fabs(xa) * 4.5e-16 + 1e-300);
          hi = xb + (fabs(xb) * 4.5e-16 + 1e-300);
          ylo = ya - (fabs(ya) * 4.5e-16 + 1e-300);
          yhi = yb + (fabs(yb) * 4.5e-16 + 1e-300);
          zlo = za - (fabs(za) * 4.5e-16 + 1e-300);
          zhi = zb + (fabs(zb) * 4.5e-16 + 1e-300);
        }
      }
      {
        // the copies sit in lanes 0 .. kTileB - 1 = one DPP row: four row shifts bring their extremes to lane 0
        static_assert(kTileB == 16, "reach reduction covers lanes 0..15");
#define RRTX_REACH_STEP(N)                                                                   \
        lo = fmin(lo, row_up_d<N>(lo)); hi = fmax(hi, row_up_d<N>(hi));                     \
        ylo = fmin(ylo, row_up_d<N>(ylo)); yhi = fmax(yhi, row_up_d<N>(yhi));               \
        zlo = fmin(zlo, row_up_d<N>(zlo)); zhi = fmax(zhi, row_up_d<N>(zhi));
        RRTX_REACH_STEP(8) RRTX_REACH_STEP(4) RRTX_REACH_STEP(2) RRTX_REACH_STEP(1)
#undef RRTX_REACH_STEP
      }
      if (lane == 0) {
        sm.lo = lo; sm.hi = hi; sm.ylo = ylo; sm.yhi = yhi; sm.zlo = zlo; sm.zhi = zhi;
        sm.n_list = 0; sm.n_groups = 0; sm.gmode = 0;
      }
    }
    __syncthreads();
    RRTX_TILE_CLK(1);
    const double lo = sm.lo, hi = sm.hi, ylo = sm.ylo, yhi = sm.yhi;
    bool gmode_w0 = false;                  // (wave 0) this tile's sorted part is listed as groups
    // (no copy of the tile can have a neighbour: nothing to list or screen -- but the fused path's sample pass,
    // which sits in the list phase, runs all the same: explicitPointCheck of a sample does not depend on its ball;
    // with lo > hi every range below is empty)
    if (lo <= hi || EXT) {
      for (int cb = 0; cb < n_chunks; cb += kTbList) {
        // ---- 2. chunk list of this pass: chunk ids in [cb, ce) ----
        const int ce = min(cb + kTbList, n_chunks);
        if (wave == 0) {
          const int w_end = min(ce, tg.n_sorted_chunks);
          if (cb == 0 && tg.groups && w_end > 0) {
            // ---- 2a. the sorted part as GROUPS of eight positions (one lane's nodes).  Inside an (x, y)
            //      cell the positions are ordered by bin of the third coordinate, so the positions of a
            //      cell that can hold a neighbour are ONE run: bins [cz0, cz1] (a node in another bin has
            //      |z_q - z_n| > R for every copy of the tile, hence fl(dz dz) >= thr and s >= thr, as for
            //      x and y: slab_of is monotone).  Lane = cell within reach, in position order; a group
            //      shared by two runs is listed once (prefix maximum of the runs' last groups).  Groups
            //      of whole sorted chunks only: the chunk the sorted part ends in goes by extent below.
            const SlabParams sp = *tg.sp;
            const int Kz = tg.kz;
            const int cx0 = slab_of(lo, sp.x0, sp.inv_wx, sp.Kx), cx1 = slab_of(hi, sp.x0, sp.inv_wx, sp.Kx);
            const int cy0 = slab_of(ylo, sp.y0, sp.inv_wy, sp.Ky), cy1 = slab_of(yhi, sp.y0, sp.inv_wy, sp.Ky);
            const int cz0 = slab_of(sm.zlo, sp.z0, sp.inv_wz, Kz), cz1 = slab_of(sm.zhi, sp.z0, sp.inv_wz, Kz);
            const int wdt = cx1 - cx0 + 1, nc = (lo <= hi) ? wdt * (cy1 - cy0 + 1) : 0;
            const int g_end = tg.n_sorted_chunks * (kChunkF / 8);
            int listed_to = -1, G = 0;                     // wave-uniform
            for (int i0 = 0; i0 < nc && G <= kTbGroups; i0 += 64) {
              const int i = i0 + lane;
              int g0 = 0, g1 = -1;                         // this cell's groups (empty)
              if (i < nc) {
                const int row = i / wdt, col = i - row * wdt;
                const int cy = cy0 + row;
                const int c = cy * sp.Kx + ((cy & 1) ? (sp.Kx - 1 - cx1) : cx0) + col;
                const int p0 = tg.cell_start[c * Kz + cz0], p1 = tg.cell_start[c * Kz + cz1 + 1];
                if (p1 > p0) { g0 = p0 >> 3; g1 = min((p1 - 1) >> 3, g_end - 1); }
              }
              const int run = wave_scan_incl<true>(g1, -1);          // last group listed up to and including this lane
              const int before = max(wave_prev(run, listed_to), listed_to);
              g0 = max(g0, before + 1);
              const int cnt = max(g1 - g0 + 1, 0);
              const int inc = wave_scan_incl<false>(cnt, 0);
              const int at = G + inc - cnt;
              for (int j = 0; j < cnt; ++j)
                if (at + j < kTbGroups) sm.glist[at + j] = g0 + j;
              G += __builtin_amdgcn_readlane(inc, 63);
              listed_to = max(listed_to, __builtin_amdgcn_readlane(run, 63));
            }
            gmode_w0 = G <= kTbGroups;                     // more (a dense or spread-out tile): whole chunks
            if (lane == 0) { sm.n_groups = gmode_w0 ? G : 0; sm.gmode = gmode_w0 ? 1 : 0; }
          }
          if (cb < w_end && !gmode_w0 && lo <= hi) {
            const SlabParams sp = *tg.sp;
            const int Kz = tg.kz;
            const int cx0 = slab_of(lo, sp.x0, sp.inv_wx, sp.Kx), cx1 = slab_of(hi, sp.x0, sp.inv_wx, sp.Kx);
            const int cy0 = slab_of(ylo, sp.y0, sp.inv_wy, sp.Ky), cy1 = slab_of(yhi, sp.y0, sp.inv_wy, sp.Ky);
            int listed_to = cb - 1;                        // wave-uniform: highest chunk id listed so far
            for (int r0 = cy0; r0 <= cy1; r0 += 64) {
              const int cy = r0 + lane;
              int ca_ = 0, cb_ = -1;                       // this row's chunk range (empty)
              if (cy <= cy1) {
                const int base = cy * sp.Kx;
                const int c_first = base + ((cy & 1) ? (sp.Kx - 1 - cx1) : cx0);
                const int c_last = base + ((cy & 1) ? (sp.Kx - 1 - cx0) : cx1);
                const int p0 = tg.cell_start[c_first * Kz], p1 = tg.cell_start[(c_last + 1) * Kz];
                if (p1 > p0) { ca_ = max(p0 / kChunkF, cb); cb_ = min((p1 - 1) / kChunkF, w_end - 1); }
              }
              // rows come in increasing position order: a row starts no earlier than where the rows
              // before it (in this wave instruction and in earlier ones) have already listed
              int run = cb_;
              const int l = lane_here();
#pragma unroll
              for (int off = 1; off < 64; off <<= 1) {
                const int o = wshfl_i(run, l >= off ? l - off : l);
                if (lane >= off) run = max(run, o);
              }
              int before = wshfl_i(run, l >= 1 ? l - 1 : l);
              if (lane == 0) before = listed_to;
              before = max(before, listed_to);
              ca_ = max(ca_, before + 1);
              const int cnt = cb_ - ca_ + 1;
              if (cnt > 0) {
                const int at = atomicAdd(&sm.n_list, cnt);
                for (int j = 0; j < cnt; ++j) sm.list[at + j] = ca_ + j;
              }
              listed_to = max(listed_to, __builtin_amdgcn_readlane(run, 63));
            }
          }
        } else if (EXT && wave >= 2) {
          // ---- 1b. sample pass beside the list building (waves 2 and 3): a lane owns a pair of
          //      spheres (one table read) and walks the tile's samples, whose probes are in LDS ----
          if (cb == 0) {
            const int n_pairs = (x.m + 1) / 2;
            const float4 *tp = reinterpret_cast<const float4 *>(x.reach_f);
            const int qw = wave - 2;
            // exact part for one left-over pair
            auto exact = [&](int cl, int j) {
              const typename QRecT<D>::type c = sm.cp[cl];
              bool listed;
              const SphRec rec = x.sph[j];          // (requested beside the sample record, used if listed)
              if (sample_exact(x, j, c.x, c.y, c.z, sm.sbase[cl], &listed)) sm.sbad[cl] = 1;
              if (listed) {
                const int at = atomicAdd(&sm.snl[cl], 1);
                if (at < kSphListCap) { sm.ssl[cl][at] = j; sm.ssr[cl][at] = rec; }
              }
            };
            // the tile's probes: lane cl < 16 of this wave holds sample cl's, the loop below takes them from
            // there with v_readlane (scalar operands of the packed arithmetic: no LDS read and wait per sample)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            float4 mypf = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < q1 - q0) mypf = sm.srp[lane];
            const int ipx = __float_as_int(mypf.x), ipy = __float_as_int(mypf.y), ipz = __float_as_int(mypf.z),
                      ipw = __float_as_int(mypf.w);
            const int nsamp = q1 - q0;
            int nqueued = 0;                            // wave-uniform: near pairs this wave has queued
            for (int pr0 = 0; pr0 < n_pairs; pr0 += 128) {
              const int pr = pr0 + (t - 128);
              float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = make_float4(0.f, 0.f, 0.f, 0.f);
              const bool pv = pr < n_pairs;
              if (pv) { u = tp[2 * pr]; v = tp[2 * pr + 1]; }
              const f32x2 ux = {u.x, u.y}, uy = {u.z, u.w}, uz = {v.x, v.y}, ur = {v.z, v.w};
              const bool va = pv && 2 * pr < x.m, vb = pv && 2 * pr + 1 < x.m;
              for (int cl = 0; cl < nsamp; ++cl) {    // (wave-uniform: the probes come by lane number)
                const float px = __int_as_float(__builtin_amdgcn_readlane(ipx, cl));
                const float py = __int_as_float(__builtin_amdgcn_readlane(ipy, cl));
                const float pz = __int_as_float(__builtin_amdgcn_readlane(ipz, cl));
                const float pw = __int_as_float(__builtin_amdgcn_readlane(ipw, cl));
                // both spheres of the pair at once (v_pk_*): the same operations in the same order as one at a time
                const f32x2 dx = ux - px, dy = uy - py, dz = uz - pz;
                f32x2 d = dx * dx;
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                const f32x2 bnd = ur + pw;
                const f32x2 b2 = bnd * bnd;
                // the left-over pairs are queued (no load in this loop) and evaluated one per lane below
                const bool na = va && !(d.x > b2.x), nb = vb && !(d.y > b2.y);
                const unsigned long long ma = __ballot(na), mb = __ballot(nb);
                if ((ma | mb) == 0ull) continue;
                // queue places from the wave's masks (a samples' worth of near pairs used to line up on one LDS counter)
                const int at_a = nqueued + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u));
                const int at_b = nqueued + __popcll(ma) +
                                 (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
                nqueued += __popcll(ma) + __popcll(mb);
                if (na) {
                  if (at_a < 64) sm.sq[qw][at_a] = cl | ((2 * pr) << 4);
                  else exact(cl, 2 * pr);            // queue full (dense obstacle field): right away
                }
                if (nb) {
                  if (at_b < 64) sm.sq[qw][at_b] = cl | ((2 * pr + 1) << 4);
                  else exact(cl, 2 * pr + 1);
                }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int nq_ev = min(nqueued, 64);
            if (lane < nq_ev) {
              const int ev = sm.sq[qw][lane];
              exact(ev & 15, ev >> 4);
            }
          }
        } else {
          // appended since the last rebuild (and the chunk the sorted part ends in): by extent
          const int nt = EXT ? 64 : 192;         // wave 1 (and waves 2, 3 unless they run the sample pass)
          for (int c = max(cb, tg.n_sorted_chunks) + (t - 64); c < ce; c += nt) {
            const ChunkExt cx = chunk_ext[c];
            if (dec_ord(cx.xhi) >= lo && dec_ord(cx.xlo) <= hi && dec_ord(cx.yhi) >= ylo && dec_ord(cx.ylo) <= yhi)
              sm.list[atomicAdd(&sm.n_list, 1)] = c;
          }
        }
        RRTX_TILE_CLK_AT(0, 8);        // wave 0: list written
        RRTX_TILE_CLK_AT(64, 9);       // wave 1: tail chunks by extent
        RRTX_TILE_CLK_AT(128, 10);     // wave 2: sample pass
        __syncthreads();
        RRTX_TILE_CLK(2);
        const int nl = sm.n_list;
        // ---- 3. screen ----
        if (cb == 0 && sm.gmode) {
          // units of 64 listed groups; part p of a tile takes units p, p + n_parts, ..., dealt to its waves in turn
          const int G = sm.n_groups, n_gu = (G + 63) >> 6;
          for (int k = wave;; k += kScanThreads / 64) {
            const int u = part + n_parts * k;
            if (u >= n_gu) break;
            const int gi = 64 * u + lane;
            const bool valid = gi < G;
            const unsigned p_lane = valid ? 8u * (unsigned)sm.glist[gi] : 0u;
            if (wn + 64 * kTileB > slice_cap) {
              const TileEmit<D, EXT> emit_now{sm, a.hs, a.meta, x, q0};
              drain_slice_to<D, true>(mine, wn, n_nodes, a, emit_now, sm.cp, q0);
            }
            scan_chunk_f32<D, 2>(fx, fy, fz, fw, fpp, 0, n_nodes, copies_f, q0, q1, mine, wn, p_lane, __ballot(valid));
            visited += min(64, G - 64 * u);
          }
        }
        for (int i = wave; i < nl; i += kScanThreads / 64) {
          const int chunk = __builtin_amdgcn_readfirstlane(sm.list[i]);
          if (n_parts > 1 && chunk % n_parts != part) continue;
          {
            const ChunkExt cx = chunk_ext[chunk];          // wave-uniform
            if (!(dec_ord(cx.xhi) >= lo && dec_ord(cx.xlo) <= hi && dec_ord(cx.yhi) >= ylo && dec_ord(cx.ylo) <= yhi))
              continue;
          }
          if (wn + 64 * kTileB > slice_cap) {
            // the slice is nearly full (a screen that passes almost everything: a non-finite or far-away node makes
            // the fp32 bounds useless): confirm it now, through the tile's own emitter -- the hits must carry the edge flags
            const TileEmit<D, EXT> emit_now{sm, a.hs, a.meta, x, q0};
            drain_slice_to<D, true>(mine, wn, n_nodes, a, emit_now, sm.cp, q0);
          }
          scan_chunk_f32<D, 1>(fx, fy, fz, fw, fpp, chunk * kChunkF, n_nodes, copies_f, q0, q1, mine, wn);
          visited += kChunkF / 8;
        }
        if (ce < n_chunks) {  // (trees beyond kTbList chunks) the list is rebuilt by the next pass
          __syncthreads();
          if (t == 0) sm.n_list = 0;
          __syncthreads();
        }
      }
    }
    // ---- 4. confirm the workgroup's entries ----
    RRTX_TILE_CLK(3);
    if (lane == 0) sm.wcnt[wave] = wn;
    // entries were written through to L2 by waves of this workgroup (same CU, same L1, which does
    // not keep written lines): a workgroup-scope fence orders them before the barrier
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    RRTX_TILE_CLK(4);
    const TileEmit<D, EXT> emit{sm, a.hs, a.meta, x, q0};
    const int c0 = sm.wcnt[0], c1 = c0 + sm.wcnt[1], c2 = c1 + sm.wcnt[2], total = c2 + sm.wcnt[3];
    for (int e0 = 0; e0 < total; e0 += kScanThreads) {
      const int e = e0 + t;
      const bool has = e < total;
      int2 en = make_int2(0, 0);
      if (has) {
        const int w = (e >= c0) + (e >= c1) + (e >= c2);
        const int local = e - (w == 0 ? 0 : (w == 1 ? c0 : (w == 2 ? c1 : c2)));
        en = blk[(size_t)w * (size_t)slice_cap + local];
      }
      if (__ballot(has) != 0ull) confirm_entry<D, true>(has, en, n_nodes, a, emit, sm.cp, q0);
    }
    __syncthreads();
    RRTX_TILE_CLK(5);
    // ---- 5. hand the collected hits to the buckets: half a wave per copy, two copies per half ----
    {
      constexpr int kPerWave = kTileB / (kScanThreads / 64);     // 4 copies per wave
      const int nc = q1 - q0;
      // the counter updates of this wave's copies fly together: lane s < 4 owns copy slot s
      int my_n = 0, my_owner = 0, my_base = 0;
      if (lane < kPerWave) {
        const int cl = (lane >> 1) * (kTileB / 2) + wave * 2 + (lane & 1);
        if (cl < nc) {
          my_n = min(sm.lcnt[cl], kTbLcap);
          if (my_n > 0) {
            my_owner = a.meta[q0 + cl].x;
            my_base = atomicAdd(&a.hs.count[my_owner], my_n);
          }
        }
      }
      const int half = lane >> 5, hl = lane & 31;
      for (int i = 0; i < kPerWave / 2; ++i) {
        const int s = i * 2 + half;                     // this half's copy slot
        const int cl = i * (kTileB / 2) + wave * 2 + half;
        const int n = __shfl(my_n, s), owner = __shfl(my_owner, s), base = __shfl(my_base, s);
        const int nmax = max(n, wshfl_i(n, lane_here() ^ 32));
        for (int j0 = 0; j0 < nmax; j0 += 32) {
          const int j = j0 + hl;
          const bool h = j < n;
          BktRec br;
          br.idx = 0; br.pad = 0; br.d2 = 0.0;
          if (h) br = sm.hrec[cl][j];
          int flags = 0;
          if constexpr (EXT) flags = tile_edge_flags<D>(x, sm, h, h ? cl : 0, br.idx, br.d2);
          place_hit(a.hs, h, owner, base + j, br.idx, br.d2, flags);
        }
      }
    }
    if constexpr (EXT) {
      // explicitPointCheck of the samples (one part of a tile reports them)
      if (part == 0 && t < q1 - q0 && x.sample_unsafe) x.sample_unsafe[a.meta[q0 + t].x] = sm.sbad[t] ? 1 : 0;
    }
    RRTX_TILE_CLK(6);
    __syncthreads();          // LDS is reused by the next tile
    RRTX_TILE_CLK(7);
  }
  if (lane == 0) visits[slice] = visited;
}

// ------------------------------------------------------------- offsets ------
// Exclusive scan of the list lengths.  block_sum (per-256 sums, nn_blocksum_kernel) is used for
// very large batches; otherwise workgroup b adds up the counts of the workgroups before it
// itself, which is cheaper than another launch.  Also re-zeroes the bucket histogram of the
// culled scan for the next call.
__global__ __launch_bounds__(256) void nn_blocksum_kernel(const int *__restrict__ count, int nq,
                                                          long long *__restrict__ block_sum) {
  __shared__ long long wsum[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  long long v = (i < nq) ? (long long)count[i] : 0ll;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void nn_offsets_kernel(const int *__restrict__ count, int nq,
                                                         const long long *__restrict__ block_sum,
                                                         int64_t *__restrict__ offsets,
                                                         int *__restrict__ cursor,
                                                         int64_t *__restrict__ needed, int *__restrict__ qhist,
                                                         int n_qhist) {
  __shared__ long long red[4];
  __shared__ long long wave_tot[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int k = blockIdx.x * 256 + t; k < n_qhist; k += gridDim.x * 256) qhist[k] = 0;
  long long p = 0;
  if (block_sum) {
    for (int j = t; j < (int)blockIdx.x; j += 256) p += block_sum[j];
  } else {
    // 256 b counts precede workgroup b: 16-byte loads, four in flight per thread
    const int4 *c4 = reinterpret_cast<const int4 *>(count);
    const int n4 = (int)blockIdx.x * 64;
#pragma unroll 4
    for (int j = t; j < n4; j += 256) {
      const int4 v = c4[j];
      p += (long long)v.x + v.y + v.z + v.w;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
  if (lane == 0) red[wave] = p;
  const int i = blockIdx.x * 256 + t;
  const long long c = (i < nq) ? (long long)count[i] : 0ll;
  long long v = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    long long o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  if (lane == 63) wave_tot[wave] = v;
  __syncthreads();
  long long prefix = red[0] + red[1] + red[2] + red[3];
  for (int w = 0; w < wave; ++w) prefix += wave_tot[w];
  if (i < nq) {
    offsets[i] = prefix + v - c;
    cursor[i] = 0;
  }
  if (i == nq - 1) {
    offsets[nq] = prefix + v;
    if (needed) *needed = (int64_t)(prefix + v);
  }
}

// ------------------------------------------------------------- scatter ------
// Entry j of query q's list lives in its bucket for j < bcap and at tmp[offsets[q] + j]
// otherwise; this kernel moves the overflow list to those places.  Only launched (with
// nn_offsets_kernel in front) after a call whose lists outgrew their buckets by much; normally
// the finish kernel computes the offsets itself and lets an overflowed query collect its own
// records (kernels_finish.hip).
__global__ void nn_scatter_kernel(const HitRec *__restrict__ recs, long long cap,
                                  const Scalars *__restrict__ sc, const int64_t *__restrict__ offsets,
                                  int *__restrict__ cursor, int bcap, BktRec *__restrict__ tmp, long long out_cap) {
  long long total = (long long)sc->total;
  if (total > cap) total = cap;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    HitRec r = recs[i];
    const int owner = r.owner & 0x3fffffff;            // top bits: edge flags of the fused extend() path
    long long dst = offsets[owner] + bcap + atomicAdd(&cursor[owner], 1);
    if (dst < out_cap) {
      BktRec br;
      br.idx = r.idx; br.pad = (int)((unsigned)r.owner >> 30); br.d2 = r.d2;
      tmp[dst] = br;
    }
  }
}

}  // namespace

// ------------------------------------------------------------- launchers ----
int launch_nn_radius(rrtx_ctx *ctx, const double *q_dev, const double *r_dev_thr_lt, double r_scalar, int nq,
                     int64_t *offsets_dev, int32_t *idx_dev, double *dist_dev, int64_t cap,
                     int64_t *needed_dev, int32_t *owner_dev, int32_t *nearest_idx_dev,
                     double *nearest_dist_dev, ExtendFuse *ext) {
  // r_dev_thr_lt: optional device array of 2*nq thresholds (thr_lt[nq] then thr_gt[nq])
  if (ext) ext->fused = false;
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "radius search on an empty tree");
  if (nq <= 0) return RRTX_OK;
  if (nq >= (1 << 30)) return fail(ctx, RRTX_E_INVALID, "radius search: at most 2^30 - 1 queries per call");
  const int D = ctx->dim;
  const int n_slots = 1 << ctx->n_wraps;
  hipStream_t st = ctx->stream;

  double tlt = 0.0, tgt = 0.0;
  if (!r_dev_thr_lt) {
    if (ctx->thr_cache_r != r_scalar || std::isnan(r_scalar)) {
      ctx->thr_cache_r = r_scalar;
      ctx->thr_cache_lt = thr_first_ge(r_scalar);
      ctx->thr_cache_gt = thr_first_gt(r_scalar);
    }
    tlt = ctx->thr_cache_lt;
    tgt = ctx->thr_cache_gt;
  }
  const size_t n_copies_max = (size_t)nq * n_slots;
  const size_t qrec_bytes = (D == 4) ? sizeof(QRec4) : sizeof(QRec3);
  RRTX_HIP(ctx, ctx->ws_slots.ensure(n_copies_max * sizeof(SlotRec)));
  RRTX_HIP(ctx, ctx->ws_copies.ensure(n_copies_max * qrec_bytes));
  RRTX_HIP(ctx, ctx->ws_copy_meta.ensure(n_copies_max * sizeof(int2)));
  RRTX_HIP(ctx, ctx->ws_counts.ensure(((size_t)nq * 3 + 2) * sizeof(int)));
  // two Scalars records used alternately: each call's pack kernel resets the other one
  if (!ctx->ws_scalars.p) {
    RRTX_HIP(ctx, ctx->ws_scalars.ensure(2 * sizeof(Scalars)));
    RRTX_HIP(ctx, hipMemsetAsync(ctx->ws_scalars.p, 0, 2 * sizeof(Scalars), st));
  }
  const long long rec_cap = (long long)(cap > 0 ? cap : 1);
  RRTX_HIP(ctx, ctx->ws_recs.ensure((size_t)rec_cap * sizeof(HitRec)));
  RRTX_HIP(ctx, ctx->ws_tmp.ensure((size_t)rec_cap * sizeof(BktRec)));
  // What the previous calls reported (host-mapped words written by the finish kernel; possibly one call
  // stale, which is fine: this only tunes, every setting gives the same lists).  Lists that outgrew
  // their buckets: wider buckets from now on; many such records: offsets + scatter as launches of their
  // own in front of the finish kernel instead of every overflowed query collecting its own records.
  if (!ctx->mailbox) {
    RRTX_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->mailbox), 64, hipHostMallocMapped));
    std::memset(ctx->mailbox, 0, 64);
  }
  const unsigned prev_overflow = *reinterpret_cast<volatile unsigned *>(ctx->mailbox);
  if (prev_overflow > 0 && ctx->bkt_mult < 16) ctx->bkt_mult *= 2;

  // (the switch is committed once the pack kernel, which resets the other record, is enqueued:
  // an error return before that leaves the pristine record for the next call)
  const int flip = ctx->scalars_flip ^ 1;
  Scalars *sc = ctx->ws_scalars.as<Scalars>() + flip;
  Scalars *sc_next = ctx->ws_scalars.as<Scalars>() + (flip ^ 1);
  int *count = ctx->ws_counts.as<int>();
  int *cursor = count + nq + 1;         // (count[nq] is scratch of the pack kernel)

  // ---- slab culling: on for large trees unless the worst-case unit list would be huge ----
  const int n_nodes = (int)ctx->n_nodes;
  const bool use_filter = ctx->opt_nn_filter != 0;
  int tile_q = use_filter ? kTileQFilter : kTileQExact;
  if (ctx->opt_tile_q > 0) tile_q = (ctx->opt_tile_q + kQPI - 1) / kQPI * kQPI;
  if (use_filter && tile_q > 128) tile_q = 128;   // bounds the worst case of one chunk (64 tile_q entries)
  const int n_tiles = (int)((n_copies_max + tile_q - 1) / tile_q);
  const int n_chunks = (n_nodes + kSlabChunk - 1) / kSlabChunk;
  const bool use_cull = use_filter && (ctx->opt_nn_cull == 2 || (ctx->opt_nn_cull == 1 && n_nodes >= 8192));
  ctx->last_culled = use_cull;
  int n_buckets = 1, g2 = 1, g3 = 1;
  int *qhist = nullptr;
  int2 *cbk = nullptr;
  if (use_cull) {
    int rc = slab_refresh(ctx, (long long)((n_copies_max + 15) / 16));
    if (rc) return rc;
    // square (x, y) grid of about 16 copies per bucket, or (trees with an extent in the third coordinate,
    // decided by the pack kernel from the node bounds) a cubic grid of about 4: query_grid, nn_device.hpp
    while (g2 * g2 < (long long)(n_copies_max / 16) && g2 * g2 < kMaxQBuckets) g2 *= 2;
    g3 = (int)std::lround(std::cbrt((double)n_copies_max / 4.0));
    if (g3 < 1) g3 = 1;
    if (g3 > 16) g3 = 16;
    if ((ctx->opt_tune >> 16) & 0x1f) g3 = (ctx->opt_tune >> 16) & 0x1f;   // experiment switch: side of the cubic grid
    if (g3 > 16) g3 = 16;
    if (ctx->opt_tune & 1) g3 = 1;          // experiment switch: (x, y) order of the copies
    n_buckets = g2 * g2 > g3 * g3 * g3 ? g2 * g2 : g3 * g3 * g3;
    if (!ctx->ws_qhist.p) {               // stays all zero between calls (nn_offsets_kernel re-zeroes it)
      RRTX_HIP(ctx, ctx->ws_qhist.ensure(sizeof(int) * (size_t)(kMaxQBuckets + 1)));
      RRTX_HIP(ctx, hipMemsetAsync(ctx->ws_qhist.p, 0, sizeof(int) * (size_t)(kMaxQBuckets + 1), st));
    }
    RRTX_HIP(ctx, ctx->ws_cb.ensure(sizeof(int2) * n_copies_max));
    RRTX_HIP(ctx, ctx->ws_copies_s.ensure(n_copies_max * qrec_bytes));
    RRTX_HIP(ctx, ctx->ws_meta_s.ensure(n_copies_max * sizeof(int2)));
    qhist = ctx->ws_qhist.as<int>();
    cbk = ctx->ws_cb.as<int2>();
  }
  // ---- hit sink: per-query buckets (bkt_mult x the average the caller made room for, at most 16 GiB
  //      in all -- there are 288) + overflow list ----
  long long bcap_ll = (rec_cap + nq - 1) / nq * ctx->bkt_mult;
  while (bcap_ll > 16 && bcap_ll * nq > (1ll << 30)) bcap_ll /= 2;
  bcap_ll = (bcap_ll + 7) / 8 * 8;
  if (bcap_ll < 8) bcap_ll = 8;
  if (bcap_ll > (1ll << 24)) bcap_ll = 1ll << 24;
  const int bcap = (int)bcap_ll;
  RRTX_HIP(ctx, ctx->ws_bkt.ensure((size_t)nq * (size_t)bcap * sizeof(BktRec)));
  HitSink hs;
  hs.count = count;
  hs.bkt = ctx->ws_bkt.as<BktRec>();
  hs.bcap = bcap;
  hs.pad = 0;
  hs.recs = ctx->ws_recs.as<HitRec>();
  hs.cap = rec_cap;
  hs.sc = sc;
  ConfirmArgs ca;
  double *const *pd = use_cull ? ctx->sl_d : ctx->nodes;
  ca.nx = pd[0]; ca.ny = pd[1]; ca.nz = pd[2]; ca.nw = pd[D == 4 ? 3 : 2];
  ca.copies = use_cull ? ctx->ws_copies_s.p : ctx->ws_copies.p;
  ca.meta = use_cull ? ctx->ws_meta_s.as<int2>() : ctx->ws_copy_meta.as<int2>();
  ca.slots = ctx->ws_slots.as<SlotRec>();
  ca.pos_id = use_cull ? ctx->sl_id : nullptr;
  ca.n_slots = n_slots;
  ca.pad = 0;
  ca.hs = hs;
  RRTX_HIP(ctx, ctx->ws_confirm_args.ensure(sizeof(ConfirmArgs)));
  ConfirmArgs *ca_dev = ctx->ws_confirm_args.as<ConfirmArgs>();
  PackFused pf;
  pf.count = count;
  pf.sc_next = sc_next;
  pf.ca_dst = ca_dev;
  pf.nx = ctx->nodes[0]; pf.ny = ctx->nodes[1]; pf.nz = ctx->nodes[2]; pf.nw = ctx->nodes[D == 4 ? 3 : 2];
  // fused extend() path (sphere list; the caller has synced the tables): only the culled search carries it
  const bool fuse = ext && use_cull && D == 3 && n_slots == 1;
  ExtendDev xd;
  std::memset(&xd, 0, sizeof(xd));
  xd.r_bound = -1.0;
  pf.sph = nullptr; pf.m_sph = -1;
  pf.root_rule = ctx->opt_root_rule;
  if (fuse) {
    xd.sph = ctx->d_sph.as<SphRec>();
    xd.stab = ctx->d_sph_sample.as<SampleSph>();
    xd.reach_f = ctx->d_sph_reach_f.as<float>();
    xd.naos = reinterpret_cast<const double4 *>(ctx->nodes_aos);
    xd.ox = ctx->origin[0]; xd.oy = ctx->origin[1]; xd.oz = ctx->origin[2];
    xd.m = ctx->sph_n_active;
    xd.r_bound = (ext->r >= 0.0 && xd.m > 0) ? ext->r * (1.0 + 1e-12) : -1.0;
    xd.sample_unsafe = ext->sample_unsafe;
    pf.sph = xd.sph; pf.m_sph = xd.m;
    ext->fused = true;
  }

  const double *thr_lt_arr = r_dev_thr_lt;
  const double *thr_gt_arr = r_dev_thr_lt ? r_dev_thr_lt + nq : nullptr;

  span_begin(ctx, KF_NN_FINISH);
  {
    dim3 grid((nq + 255) / 256), block(256);
    if (D == 4)
      hipLaunchKernelGGL(nn_pack_kernel<4>, grid, block, 0, st, q_dev, nq, thr_lt_arr, thr_gt_arr, tlt, tgt,
                         ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1], ctx->wrap_dim[2],
                         ctx->wrap_period[0], ctx->wrap_period[1], ctx->wrap_period[2],
                         ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_slots.as<SlotRec>(), ctx->ws_copies.as<QRec4>(),
                         ctx->ws_copy_meta.as<int2>(), sc, ctx->d_xrange.as<unsigned long long>(), g2, g3, qhist,
                         cbk, pf, ca);
    else
      hipLaunchKernelGGL(nn_pack_kernel<3>, grid, block, 0, st, q_dev, nq, thr_lt_arr, thr_gt_arr, tlt, tgt,
                         ctx->n_wraps, ctx->wrap_dim[0], ctx->wrap_dim[1], ctx->wrap_dim[2],
                         ctx->wrap_period[0], ctx->wrap_period[1], ctx->wrap_period[2],
                         ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_slots.as<SlotRec>(), ctx->ws_copies.as<QRec3>(),
                         ctx->ws_copy_meta.as<int2>(), sc, ctx->d_xrange.as<unsigned long long>(), g2, g3, qhist,
                         cbk, pf, ca);
  }
  span_end(ctx);
  ctx->scalars_flip = flip;

  // ---- scan geometry: tiles of copies x node segments (segment = XCD-affine) ----
  const int chunk = use_filter ? kChunkF : kChunk;
  const int wg_nodes = (kScanThreads / 64) * chunk;  // nodes one workgroup covers per pass
  int max_seg = (n_nodes + wg_nodes - 1) / wg_nodes;
  int want_seg = ((use_filter ? ctx->opt_scan_items : 4096) + n_tiles - 1) / n_tiles;
  int n_seg = want_seg < max_seg ? want_seg : max_seg;
  if (n_seg < 1) n_seg = 1;
  if (n_seg >= 8) n_seg = n_seg / 8 * 8;  // segment index == blockIdx % 8 class == XCD
  int seg_len = round_up((n_nodes + n_seg - 1) / n_seg, chunk);
  n_seg = (n_nodes + seg_len - 1) / seg_len;
  ctx->last_tile_q = use_cull ? kTileB : tile_q;

  if (use_filter) {
    const size_t qf_bytes = (D == 4) ? sizeof(QRecF4) : sizeof(QRecF3);
    RRTX_HIP(ctx, ctx->ws_copies_f.ensure((n_copies_max + kQPI) * qf_bytes));
    span_begin(ctx, KF_NN_FINISH);
    dim3 grid((unsigned)((n_copies_max + kQPI + 255) / 256)), block(256);
    const unsigned long long *absmax = ctx->d_absmax.as<unsigned long long>();
    if (use_cull) {
      if (D == 4)
        hipLaunchKernelGGL(nn_place_kernel<4>, grid, block, 0, st, ctx->ws_copies.as<QRec4>(),
                           ctx->ws_copy_meta.as<int2>(), cbk, qhist, n_buckets, sc, absmax, ctx->origin[0],
                           ctx->origin[1], ctx->origin[2], ctx->origin[3], ctx->ws_copies_s.as<QRec4>(),
                           ctx->ws_meta_s.as<int2>(), ctx->ws_copies_f.as<QRecF4>());
      else
        hipLaunchKernelGGL(nn_place_kernel<3>, grid, block, 0, st, ctx->ws_copies.as<QRec3>(),
                           ctx->ws_copy_meta.as<int2>(), cbk, qhist, n_buckets, sc, absmax, ctx->origin[0],
                           ctx->origin[1], ctx->origin[2], ctx->origin[3], ctx->ws_copies_s.as<QRec3>(),
                           ctx->ws_meta_s.as<int2>(), ctx->ws_copies_f.as<QRecF3>());
    } else if (D == 4)
      hipLaunchKernelGGL(nn_filter_prep_kernel<4>, grid, block, 0, st, ctx->ws_copies.as<QRec4>(), sc, absmax,
                         (int)n_copies_max, ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_copies_f.as<QRecF4>());
    else
      hipLaunchKernelGGL(nn_filter_prep_kernel<3>, grid, block, 0, st, ctx->ws_copies.as<QRec3>(), sc, absmax,
                         (int)n_copies_max, ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->origin[3],
                         ctx->ws_copies_f.as<QRecF3>());
    span_end(ctx);
  }

  span_begin(ctx, KF_NN_SCAN);
  {
    dim3 grid((unsigned)n_tiles * (unsigned)n_seg), block(kScanThreads);
    const int wi = D == 4 ? 3 : 2;
    if (use_cull) {
      // one workgroup per (tile of kTileB copies, part); few tiles: several parts share a tile's chunks;
      // many tiles: a bounded grid walks them
      const int n_tiles_b = (int)((n_copies_max + kTileB - 1) / kTileB);
      int n_parts = (1024 + n_tiles_b - 1) / n_tiles_b;
      if (n_parts > 64) n_parts = 64;
      long long nb = (long long)n_tiles_b * n_parts;
      if (nb > 4096) nb = 4096;             // n_parts == 1 here
      const int n_slices = (int)nb * (kScanThreads / 64);
      const int slice_cap = 64 * kTileB + kTbSlack;
      RRTX_HIP(ctx, ctx->ws_ev_a.ensure((size_t)n_slices * (size_t)slice_cap * sizeof(int2)));
      RRTX_HIP(ctx, ctx->ws_ev_cnt.ensure((size_t)n_slices * sizeof(int)));
      ctx->last_visit_slices = n_slices;
      TileGrid tg;
      tg.sp = ctx->ws_slab_params.as<SlabParams>();
      tg.cell_start = ctx->ws_slab_start.as<int>();
      tg.n_sorted_chunks = (ctx->ws_slab_params.p && ctx->ws_slab_start.p) ? (int)(ctx->sl_n_sorted / kSlabChunk) : 0;
      tg.kz = ctx->sl_kz > 0 ? ctx->sl_kz : 1;
      tg.groups = (tg.kz > 1 && !(ctx->opt_tune & 2)) ? 1 : 0;     // (experiment switch 2: whole chunks)
      if (fuse)
        hipLaunchKernelGGL((nn_tile_kernel<3, true>), dim3((unsigned)nb), block, 0, st, ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2],
                           ctx->sl_f[wi], ctx->sl_pp, n_nodes, n_chunks, reinterpret_cast<const ChunkExt *>(ctx->chunk_ext),
                           ctx->ws_copies_s.as<QRec3>(), ctx->ws_copies_f.as<QRecF3>(), sc, n_parts,
                           ctx->ws_ev_a.as<int2>(), slice_cap, ca, ca_dev, tg, xd, ctx->ws_ev_cnt.as<int>(), n_slots == 1 ? nq : -1);
      else if (D == 4)
        hipLaunchKernelGGL((nn_tile_kernel<4, false>), dim3((unsigned)nb), block, 0, st, ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2],
                           ctx->sl_f[wi], ctx->sl_pp, n_nodes, n_chunks, reinterpret_cast<const ChunkExt *>(ctx->chunk_ext),
                           ctx->ws_copies_s.as<QRec4>(), ctx->ws_copies_f.as<QRecF4>(), sc, n_parts,
                           ctx->ws_ev_a.as<int2>(), slice_cap, ca, ca_dev, tg, xd, ctx->ws_ev_cnt.as<int>(), n_slots == 1 ? nq : -1);
      else
        hipLaunchKernelGGL((nn_tile_kernel<3, false>), dim3((unsigned)nb), block, 0, st, ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2],
                           ctx->sl_f[wi], ctx->sl_pp, n_nodes, n_chunks, reinterpret_cast<const ChunkExt *>(ctx->chunk_ext),
                           ctx->ws_copies_s.as<QRec3>(), ctx->ws_copies_f.as<QRecF3>(), sc, n_parts,
                           ctx->ws_ev_a.as<int2>(), slice_cap, ca, ca_dev, tg, xd, ctx->ws_ev_cnt.as<int>(), n_slots == 1 ? nq : -1);
    } else if (use_filter) {
      // persistent grid: opt_scan_blocks workgroups (multiple of 8) striding over the work
      unsigned pg = (unsigned)ctx->opt_scan_blocks / 8u * 8u;
      if (pg < 8u) pg = 8u;
      if (pg < grid.x) grid.x = pg;
      const int n_slices = (int)grid.x * (kScanThreads / 64);
      const int slice_cap = 64 * tile_q + kEvSlack;
      RRTX_HIP(ctx, ctx->ws_ev_a.ensure((size_t)n_slices * (size_t)slice_cap * sizeof(int2)));
      RRTX_HIP(ctx, ctx->ws_ev_cnt.ensure((size_t)n_slices * sizeof(int)));
      int2 *ev = ctx->ws_ev_a.as<int2>();
      int *ev_cnt = ctx->ws_ev_cnt.as<int>();
      float *const *nf = ctx->nodes_f;
      if (D == 4)
        hipLaunchKernelGGL(nn_scan_f32_kernel<4>, grid, block, 0, st, nf[0], nf[1], nf[2], nf[wi], ctx->nodes_pp,
                           n_nodes, ctx->ws_copies_f.as<QRecF4>(), tile_q, n_seg, seg_len, sc, ev, ev_cnt, slice_cap,
                           ca_dev);
      else
        hipLaunchKernelGGL(nn_scan_f32_kernel<3>, grid, block, 0, st, nf[0], nf[1], nf[2], nf[wi], ctx->nodes_pp,
                           n_nodes, ctx->ws_copies_f.as<QRecF3>(), tile_q, n_seg, seg_len, sc, ev, ev_cnt, slice_cap,
                           ca_dev);
      // exact confirmation of the queued entries: one lane per entry
      dim3 cgrid((unsigned)(((long long)n_slices * kConfirmParts + 3) / 4));
      if (D == 4)
        hipLaunchKernelGGL((nn_confirm_kernel<4, false>), cgrid, dim3(256), 0, st, ca, ev, ev_cnt, n_slices, slice_cap,
                           n_nodes);
      else
        hipLaunchKernelGGL((nn_confirm_kernel<3, false>), cgrid, dim3(256), 0, st, ca, ev, ev_cnt, n_slices, slice_cap,
                           n_nodes);
    } else if (D == 4)
      hipLaunchKernelGGL(nn_scan_kernel<4>, grid, block, 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                         ctx->nodes[3], n_nodes, ctx->ws_copies.as<QRec4>(), ctx->ws_copy_meta.as<int2>(),
                         ctx->ws_slots.as<SlotRec>(), n_slots, tile_q, n_seg, seg_len, sc, hs);
    else
      hipLaunchKernelGGL(nn_scan_kernel<3>, grid, block, 0, st, ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                         ctx->nodes[2], n_nodes, ctx->ws_copies.as<QRec3>(), ctx->ws_copy_meta.as<int2>(),
                         ctx->ws_slots.as<SlotRec>(), n_slots, tile_q, n_seg, seg_len, sc, hs);
  }
  span_end(ctx);
  ctx->last_pairs = (int64_t)n_copies_max * n_nodes;

  // ---- finish: offsets, order, (extend work), one launch (kernels_finish.hip) ----
  const bool prescatter = nq > 65536 || prev_overflow > 8192u;
  span_begin(ctx, KF_NN_FINISH);
  if (prescatter) {
    RRTX_HIP(ctx, ctx->ws_bsum.ensure(sizeof(long long) * (size_t)((nq + 255) / 256)));
    long long *bsum = ctx->ws_bsum.as<long long>();
    const int nblk = (nq + 255) / 256;
    const long long *bsum_arg = nullptr;
    if (nq > 65536) {                     // large batch: per-256 sums first
      hipLaunchKernelGGL(nn_blocksum_kernel, dim3(nblk), dim3(256), 0, st, count, nq, bsum);
      bsum_arg = bsum;
    }
    hipLaunchKernelGGL(nn_offsets_kernel, dim3(nblk), dim3(256), 0, st, count, nq, bsum_arg, offsets_dev, cursor,
                       needed_dev, (int *)nullptr, 0);
    hipLaunchKernelGGL(nn_scatter_kernel, dim3(1024), dim3(256), 0, st, ctx->ws_recs.as<HitRec>(), rec_cap, sc,
                       offsets_dev, cursor, bcap, ctx->ws_tmp.as<BktRec>(), (long long)cap);
  }
  {
    FinishLaunch f;
    f.count = count; f.nq = nq; f.bcap = bcap;
    f.bkt = hs.bkt; f.tmp = ctx->ws_tmp.p; f.ovf = ctx->ws_recs.p; f.ovf_cap = rec_cap; f.scalars = sc;
    f.prescattered = prescatter;
    f.offsets = offsets_dev; f.needed = needed_dev; f.idx = idx_dev; f.dist = dist_dev; f.out_cap = (long long)cap;
    f.owner = owner_dev; f.nearest_idx = nearest_idx_dev; f.nearest_dist = nearest_dist_dev;
    f.qhist = qhist; f.n_qhist = use_cull ? n_buckets + 1 : 0;
    f.mailbox = ctx->mailbox;
    f.q = q_dev;
    f.r_start = (!r_dev_thr_lt && r_scalar > 0.0) ? 2.0 * r_scalar : 1.0;
    f.hit_out = fuse ? ext->hit_out : nullptr; f.hit_in = fuse ? ext->hit_in : nullptr;
    int rc = launch_nn_finish(ctx, f);
    if (rc) return rc;
  }
  span_end(ctx);
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

#ifdef RRTX_TILE_CLOCKS
extern "C" int rrtx_debug_tile_clocks(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_clk), sizeof(unsigned long long) * (size_t)n_words);
}
#endif

int scan_units(rrtx_ctx *ctx, int *units) {
  // node chunks screened by the last culled range search: per-wave counts written by nn_tile_kernel
  std::vector<int> v((size_t)ctx->last_visit_slices);
  if (!v.empty()) {
    RRTX_HIP(ctx, hipMemcpyAsync(v.data(), ctx->ws_ev_cnt.p, sizeof(int) * v.size(), hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  long long tot = 0;
  for (int x : v) tot += x;               // groups of eight node visits; a unit = one wave's worth = 512 visits
  tot = (tot + kChunkF / 8 - 1) / (kChunkF / 8);
  *units = (int)(tot > 0x7fffffffll ? 0x7fffffffll : tot);
  return RRTX_OK;
}

}  // namespace rrtx
