// kernels_finish.hip -- the last launch of the range search: one kernel turns the per-query hit
// buckets into the caller's CSR lists.
//
// Replaces, per workgroup of 16 queries (half a wave each):
//   offsets : exclusive scan of the list lengths -- workgroup b adds up the 16 b counts before it
//             (16-byte loads; cheaper than a launch of its own for batches up to 65536 queries)
//   scatter : a query whose list outgrew its bucket gathers its records from the shared overflow
//             list itself (rare; after a call that overflowed, the next one widens the buckets and
//             may pre-scatter with nn_offsets + nn_scatter, kernels_nn.hip)
//   order   : sort by node index, dist = sqrt(d2) (the key the reference stores,
//             R/kdTree_general.jl:829-831), owner, nearest of the list; lists longer than 64 entries
//             are sorted by the whole workgroup (bitonic network in LDS up to 2048 entries)
//   nearest : a sample whose ball is empty gets kdFindNearest's answer from an expanding search
//             over the slab index (block_nearest, nn_device.hpp) -- no -1 leaves the device
//   extend  : the two collision flags every record of the fused extend() path carries (decided where
//             the neighbour was found, kernels_nn.hip TileEmit) go to hit_out / hit_in
// gfx950 only.
#include "nn_device.hpp"

namespace rrtx {

namespace {

constexpr int kFinThreads = 512;
constexpr int kFinQ = kFinThreads / 32;   // queries per workgroup
constexpr int kBigSort = 2048;      // LDS sort of the normal build of the kernel
constexpr int kHugeSort = 8192;     // ... of the build for calls with long lists (dense balls: Dubins spaces, sweeps)
constexpr int kFinBins = 2048;      // index bins of the long-list build's placement
constexpr int kFinCrowd = 128;      // ... and the most entries a bin may hold before the list goes to the sort network

struct FinishArgs {
  const int *count;
  int nq, bcap;
  const BktRec *bkt;           // [nq][bcap]
  BktRec *tmp;                 // entry j >= bcap of query q lives at tmp[offsets[q] + j]
  const HitRec *ovf;           // shared overflow list
  long long ovf_cap;
  const Scalars *sc;           // ->total: records in the overflow list
  int prescattered;            // 1: nn_offsets + nn_scatter already ran (offsets valid, tmp filled)
  int want_nearest_fix;        // 1: resolve empty balls with block_nearest
  int64_t *offsets;
  int64_t *needed;
  int32_t *idx;
  double *dist;
  long long out_cap;
  int32_t *owner;
  int32_t *nearest_idx;
  double *nearest_dist;
  int *qhist;                  // bucket histogram of the culled search, re-zeroed for the next call
  int n_qhist;
  unsigned *mailbox;           // host-mapped: [0] = overflow records of this call
  const double *q;             // query points (nearest of empty balls)
  uint8_t *hit_out, *hit_in;   // fused extend(): flags of the records (null otherwise)
  double r_start;
  NearestIndex ni;
};

template <int KSORT>
struct FinLds {
  long long red[kFinThreads / 64];
  long long off[kFinQ + 1];
  int kfull[kFinQ];            // list length as counted
  int todo[kFinQ];             // the list needs the whole workgroup
  unsigned todo_mask;          // ... as bits (the two loops over the workgroup's queries below only visit set bits:
  unsigned empty_mask;         //     sixteen flag reads per wave and loop were a quarter of the kernel's instructions)
  int gcnt;
  int s_idx[KSORT];
  double s_d2[KSORT];
  unsigned char s_fl[KSORT];   // the records' edge flags travel with the sort
  int bins[kFinBins + 1];      // long-list build: counts, then first places, of the index bins (see below)
  int mm[2 * (kFinThreads / 64)];
  double r_best[kFinThreads / 64];
  int r_besti[kFinThreads / 64];
  NearestScratch ns;
};

// entry j of the (unordered) list of query q; the part behind the bucket is read with agent-scope
// loads (it may have been written by this workgroup a moment ago)
__device__ __forceinline__ BktRec load_rec(const FinishArgs &a, int q, long long b, long long j) {
  if (j < a.bcap) return a.bkt[(size_t)q * (size_t)a.bcap + (size_t)j];
  const unsigned long long *p = reinterpret_cast<const unsigned long long *>(a.tmp + b + j);
  BktRec r;
  const unsigned long long w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.idx = (int)(unsigned)(w0 & 0xffffffffull);
  r.pad = (int)(unsigned)(w0 >> 32);
  r.d2 = __longlong_as_double((long long)w1);
  return r;
}

// D: coordinates per query point.  Eight waves per SIMD = four workgroups per CU: the 1024 workgroups of
// a 16384-query batch are then resident at once and the kernel is one pass of their dependency chain.
template <int D, int KSORT>
__global__ __launch_bounds__(kFinThreads, KSORT > kBigSort ? 2 : 8) void nn_finish_kernel(FinishArgs a) {
  __shared__ FinLds<KSORT> sm;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int hl = t & 31, hw = t >> 5;
  const int q0 = (int)blockIdx.x * kFinQ;
  const int q = q0 + hw;
  const bool qv = q < a.nq;

  // ---- housekeeping for the next call ----
  for (int k = (int)blockIdx.x * kFinThreads + t; k < a.n_qhist; k += (int)gridDim.x * kFinThreads) a.qhist[k] = 0;
  if (blockIdx.x == 0 && t == 0 && a.mailbox) {
    const unsigned long long tot = a.sc->total;
    a.mailbox[0] = tot > 0xffffffffull ? 0xffffffffu : (unsigned)tot;
  }

  // Everything below up to the barrier depends only on the query, not on where its list goes: the
  // loads (count, bucket records, node records, sphere table) are issued before the workgroup meets
  // for the offsets, so the prefix over the preceding counts travels beside them.
  // (No workgroup barrier on the way: a barrier waits for every outstanding load.  Each half wave
  // resets its own LDS words; LDS operations of one wave execute in order.)
  if (hl == 0) sm.todo[hw] = 0;
  if (t == 0) sm.empty_mask = 0u;
  int kfull = 0;
  if (qv) kfull = a.count[q];
  const int kInt = 0x7fffffff;
  BktRec e0, e1;
  e0.idx = kInt; e0.pad = 0; e0.d2 = __builtin_inf();
  e1 = e0;
  // the first 32 slots of the bucket are requested before the count has arrived (bcap >= 8; slots past
  // the count hold stale records and are masked below)
  if (qv && hl < a.bcap) e0 = a.bkt[(size_t)q * (size_t)a.bcap + (size_t)hl];
  long long psum = 0;
  if (!a.prescattered) {
    // 16 b counts precede workgroup b
    const int4 *c4 = reinterpret_cast<const int4 *>(a.count);
    const int n4 = (int)blockIdx.x * (kFinQ / 4);
#pragma unroll 4
    for (int j = t; j < n4; j += kFinThreads) {
      const int4 v = c4[j];
      psum += (long long)v.x + v.y + v.z + v.w;
    }
  }
  // a list the half wave cannot finish alone: longer than 64 entries, or overflowed its bucket
  const bool small = kfull <= 64 && kfull <= a.bcap;
  const int kk = (qv && small) ? kfull : 0;
  if (!(hl < kk)) { e0.idx = kInt; e0.d2 = __builtin_inf(); }
  if (hl + 32 < kk) e1 = a.bkt[(size_t)q * (size_t)a.bcap + (size_t)(hl + 32)];
  __builtin_amdgcn_wave_barrier();

  // ---- short lists: rank by node index inside the half wave ----
  const int m0 = e0.idx, m1 = e1.idx;
  const double d0 = e0.d2, d1 = e1.d2;
  int r0 = 0, r1 = 0;
  {
    const int kmax = max(kk, __shfl_xor(kk, 32));      // the two halves of a wave run the same loops
    if (kmax <= 32) {
      // the usual case, no second entry in either half of the wave: half the compares, four entries per round
      // (slots past a list hold index INT_MAX, which ranks below nothing)
      const int kr = (kmax + 3) & ~3;
      for (int j = 0; j < kr; j += 4) {
        const int o0 = __shfl(m0, j, 32), o1 = __shfl(m0, j + 1, 32), o2 = __shfl(m0, j + 2, 32), o3 = __shfl(m0, j + 3, 32);
        r0 += ((o0 < m0) ? 1 : 0) + ((o1 < m0) ? 1 : 0) + ((o2 < m0) ? 1 : 0) + ((o3 < m0) ? 1 : 0);
      }
    } else {
      for (int j = 0; j < 32; ++j) {
        const int o = __shfl(m0, j, 32);
        r0 += (o < m0) ? 1 : 0;
        r1 += (o < m1) ? 1 : 0;
      }
      for (int j = 32; j < kmax; ++j) {
        const int o = __shfl(m1, j - 32, 32);
        r0 += (o < m0) ? 1 : 0;
        r1 += (o < m1) ? 1 : 0;
      }
    }
  }
  double nbest = d0;
  int nbest_i = m0;
  if (a.nearest_idx) {
    // lexicographic minimum of (d2, index) over the half wave: the smallest d2 first (no NaN among confirmed
    // hits; empty slots hold +inf), then the lowest index among the lanes that attain it
    if ((d1 < nbest) || (d1 == nbest && m1 < nbest_i)) { nbest = d1; nbest_i = m1; }
    double dm = nbest;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) dm = fmin(dm, __shfl_xor(dm, off));
    int ci = (nbest == dm) ? nbest_i : kInt;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) ci = min(ci, __shfl_xor(ci, off));
    nbest = dm; nbest_i = ci;
  }

  // ---- offsets of this workgroup's queries ----
  if (!a.prescattered) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
    if (lane == 0) sm.red[wave] = psum;
  }
  if (qv && hl == 0) {
    sm.kfull[hw] = kfull;
    if (!small) sm.todo[hw] = 1;
  }
  if (!qv && hl == 0) sm.kfull[hw] = 0;
  __syncthreads();
  if (wave == 0) {
    // the sixteen list starts: a scan over lanes 0 .. 15 (one thread walking the sixteen lengths kept the other
    // fifteen half waves waiting at the next barrier)
    long long p = 0;
    if (a.prescattered) {
      p = a.offsets[q0];
    } else {
      for (int w = 0; w < kFinThreads / 64; ++w) p += sm.red[w];
    }
    const int kf = lane < kFinQ ? sm.kfull[lane] : 0;
    const int td = lane < kFinQ ? sm.todo[lane] : 0;
    long long v = kf;
#pragma unroll
    for (int off = 1; off < kFinQ; off <<= 1) {
      const long long o = __shfl_up(v, off);
      if (lane >= off) v += o;
    }
    if (lane < kFinQ) sm.off[lane] = p + v - kf;
    if (lane == kFinQ - 1) sm.off[kFinQ] = p + v;
    const unsigned long long tmb = __ballot(td != 0);
    if (lane == 0) sm.todo_mask = (unsigned)tmb;
  }
  __syncthreads();
  if (!a.prescattered && t <= kFinQ) {
    const int qq = q0 + t;
    if (qq < a.nq) a.offsets[qq] = sm.off[t];
    if (qq == a.nq) {
      a.offsets[qq] = sm.off[t];
      if (a.needed) *a.needed = sm.off[t];
    }
  }

  // ---- results of the short lists at their sorted places (never past the caller's capacity) ----
  {
    const long long b = qv ? sm.off[hw] : 0;
    const bool w0 = hl < kk && b + r0 < a.out_cap, w1 = hl + 32 < kk && b + r1 < a.out_cap;
    if (w0) {
      a.idx[b + r0] = m0;
      a.dist[b + r0] = sqrt_rn(d0);
      if (a.owner) a.owner[b + r0] = q;
      if (a.hit_out) { a.hit_out[b + r0] = e0.pad & 1; a.hit_in[b + r0] = (e0.pad >> 1) & 1; }
    }
    if (w1) {
      a.idx[b + r1] = m1;
      a.dist[b + r1] = sqrt_rn(d1);
      if (a.owner) a.owner[b + r1] = q;
      if (a.hit_out) { a.hit_out[b + r1] = e1.pad & 1; a.hit_in[b + r1] = (e1.pad >> 1) & 1; }
    }
    if (a.nearest_idx && qv && small && hl == 0) {
      if (kk > 0) { a.nearest_idx[q] = nbest_i; a.nearest_dist[q] = sqrt_rn(nbest); }
      else if (a.want_nearest_fix) atomicOr(&sm.empty_mask, 1u << hw);
      else { a.nearest_idx[q] = -1; a.nearest_dist[q] = __builtin_inf(); }
    }
  }
  __syncthreads();

  // ---- long or overflowed lists: the whole workgroup, one list at a time ----
  for (unsigned tm = sm.todo_mask; tm != 0u; tm &= tm - 1u) {     // workgroup-uniform
    const int g = __ffs((int)tm) - 1;
    const int qq = q0 + g;
    const long long gb = sm.off[g];
    const int gfull = sm.kfull[g];
    long long ge = gb + gfull;
    if (ge > a.out_cap) ge = a.out_cap;
    const int gk = ge > gb ? (int)(ge - gb) : 0;
    if (!a.prescattered && gfull > a.bcap) {
      // gather this query's records from the shared overflow list behind its bucket part
      if (t == 0) sm.gcnt = 0;
      __syncthreads();
      long long total = (long long)a.sc->total;
      if (total > a.ovf_cap) total = a.ovf_cap;
      for (long long i = t; i < total; i += kFinThreads) {
        const HitRec r = a.ovf[i];
        if ((r.owner & 0x3fffffff) == qq) {
          const long long dst = gb + a.bcap + atomicAdd(&sm.gcnt, 1);
          if (dst < a.out_cap) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(a.tmp + dst);
            __hip_atomic_store(p, (unsigned long long)(unsigned)r.idx | ((unsigned long long)((unsigned)r.owner >> 30) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p + 1, (unsigned long long)__double_as_longlong(r.d2), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    double best = __builtin_inf();
    int best_i = 0x7fffffff;
    bool placed = false;                                   // workgroup-uniform: the list was written by counting
    if (KSORT > kBigSort && gk <= KSORT) {
      // Long-list build: no sort network.  The node indices of a list are distinct and spread over the tree, so
      // every entry finds its place by counting: the entries are dealt into kFinBins equal-width bins of
      // [smallest index, largest index] (a monotone function of the index; one LDS atomic each), a scan of the bin
      // counts gives every bin its first place, and an entry's rank inside its bin -- a handful of entries -- is
      // the number of bin mates with a smaller index (ties, which a list does not have, by arrival).  Every thread
      // keeps its entries in registers and writes them straight to their places of the caller's arrays: 8
      // barriers per list instead of the 66 of a 2048-entry bitonic network (C3, lists of ~1400: 60 -> ~10 us).
      // A list whose indices crowd into few bins (a block of consecutive indices next to a lone far one in a
      // large tree) would make the in-bin count long: more than kFinCrowd entries in one bin send the list to the
      // sort network below instead (nothing has been written by then).
      constexpr int kEpt = KSORT / kFinThreads;
      int my_idx[kEpt], my_fl[kEpt], my_bin[kEpt], my_in[kEpt];
      double my_d2[kEpt];
      int lo = 0x7fffffff, hi = (int)0x80000000;
#pragma unroll
      for (int e = 0; e < kEpt; ++e) {
        const int i = e * kFinThreads + t;
        my_idx[e] = 0x7fffffff; my_fl[e] = 0; my_d2[e] = __builtin_inf(); my_bin[e] = 0; my_in[e] = 0;
        if (i < gk) {
          const BktRec r = load_rec(a, qq, gb, i);
          my_idx[e] = r.idx; my_d2[e] = r.d2; my_fl[e] = r.pad;
          lo = min(lo, r.idx); hi = max(hi, r.idx);
        }
      }
      for (int k = t; k <= kFinBins; k += kFinThreads) sm.bins[k] = 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
      if (lane == 0) { sm.mm[2 * wave] = lo; sm.mm[2 * wave + 1] = hi; }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kFinThreads / 64; ++w) { lo = min(lo, sm.mm[2 * w]); hi = max(hi, sm.mm[2 * w + 1]); }
      const double scale = (double)kFinBins / ((double)hi - (double)lo + 1.0);      // gk >= 1 here: hi >= lo
#pragma unroll
      for (int e = 0; e < kEpt; ++e) {
        if (e * kFinThreads + t < gk) {
          int b = (int)(((double)my_idx[e] - (double)lo) * scale);                   // monotone in the index
          b = b < 0 ? 0 : (b > kFinBins - 1 ? kFinBins - 1 : b);
          my_bin[e] = b;
          my_in[e] = atomicAdd(&sm.bins[b], 1);
        }
      }
      __syncthreads();
      {
        // exclusive scan of the bin counts, four bins per thread
        static_assert(kFinBins == 4 * kFinThreads, "four bins per thread");
        const int c0 = sm.bins[4 * t], c1 = sm.bins[4 * t + 1], c2 = sm.bins[4 * t + 2], c3 = sm.bins[4 * t + 3];
        int v = c0 + c1 + c2 + c3;
        const int mine = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_up(v, off);
          if (lane >= off) v += o;
        }
        int cmax = max(max(c0, c1), max(c2, c3));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        if (lane == 63) { sm.mm[wave] = v; sm.mm[kFinThreads / 64 + wave] = cmax; }
        __syncthreads();
        int pre = v - mine;
        for (int w = 0; w < wave; ++w) pre += sm.mm[w];
        sm.bins[4 * t] = pre; sm.bins[4 * t + 1] = pre + c0; sm.bins[4 * t + 2] = pre + c0 + c1; sm.bins[4 * t + 3] = pre + c0 + c1 + c2;
        if (t == kFinThreads - 1) sm.bins[kFinBins] = pre + mine;
      }
      int crowd = 0;
#pragma unroll
      for (int w = 0; w < kFinThreads / 64; ++w) crowd = max(crowd, sm.mm[kFinThreads / 64 + w]);
      placed = crowd <= kFinCrowd;
      __syncthreads();
     if (placed) {
      int *t_idx = sm.s_idx;                                          // the indices in bin order ...
      int *t_pos = reinterpret_cast<int *>(sm.s_d2);                  // ... and where each came from
#pragma unroll
      for (int e = 0; e < kEpt; ++e) {
        if (e * kFinThreads + t < gk) {
          const int p = sm.bins[my_bin[e]] + my_in[e];
          t_idx[p] = my_idx[e];
          t_pos[p] = e * kFinThreads + t;
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < kEpt; ++e) {
        const int i = e * kFinThreads + t;
        if (i < gk) {
          const int my = my_idx[e];
          const int s0 = sm.bins[my_bin[e]], s1 = sm.bins[my_bin[e] + 1];
          int rank = s0;
          for (int k = s0; k < s1; ++k) {
            const int v = t_idx[k];
            rank += ((v < my) || (v == my && t_pos[k] < i)) ? 1 : 0;
          }
          const double d2 = my_d2[e];
          a.idx[gb + rank] = my;
          a.dist[gb + rank] = sqrt_rn(d2);
          if (a.owner) a.owner[gb + rank] = qq;
          if ((d2 < best) || (d2 == best && my < best_i)) { best = d2; best_i = my; }
          if (a.hit_out) { a.hit_out[gb + rank] = my_fl[e] & 1; a.hit_in[gb + rank] = (my_fl[e] >> 1) & 1; }
        }
      }
     }
    }
    if (placed) {
    } else if (gk <= KSORT) {
      int n2 = 64;
      while (n2 < gk) n2 <<= 1;
      for (int i = t; i < n2; i += kFinThreads) {
        BktRec r;
        r.idx = 0x7fffffff; r.pad = 0; r.d2 = __builtin_inf();
        if (i < gk) r = load_rec(a, qq, gb, i);
        sm.s_idx[i] = r.idx;
        sm.s_d2[i] = r.d2;
        sm.s_fl[i] = (unsigned char)r.pad;
      }
      __syncthreads();
      for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int i = t; i < n2; i += kFinThreads) {
            const int j = i ^ stride;
            if (j > i) {
              const bool up = (i & size) == 0;
              const int aa = sm.s_idx[i], cc = sm.s_idx[j];
              if ((aa > cc) == up) {
                sm.s_idx[i] = cc; sm.s_idx[j] = aa;
                const double da = sm.s_d2[i];
                sm.s_d2[i] = sm.s_d2[j]; sm.s_d2[j] = da;
                const unsigned char fa = sm.s_fl[i];
                sm.s_fl[i] = sm.s_fl[j]; sm.s_fl[j] = fa;
              }
            }
          }
          __syncthreads();
        }
      }
      for (int i0 = 0; i0 < gk; i0 += kFinThreads) {
        const int i = i0 + t;
        const bool act = i < gk;
        int my = 0;
        double d2 = 0.0;
        if (act) {
          my = sm.s_idx[i];
          d2 = sm.s_d2[i];
          a.idx[gb + i] = my;
          a.dist[gb + i] = sqrt_rn(d2);
          if (a.owner) a.owner[gb + i] = qq;
          if ((d2 < best) || (d2 == best && my < best_i)) { best = d2; best_i = my; }
          if (a.hit_out) { const int fl = sm.s_fl[i]; a.hit_out[gb + i] = fl & 1; a.hit_in[gb + i] = (fl >> 1) & 1; }
        }
      }
    } else {
      // longer than the LDS sort: rank counting over the unordered list
      for (int i0 = 0; i0 < gk; i0 += kFinThreads) {
        const int i = i0 + t;
        const bool act = i < gk;
        int my = 0, rank = 0;
        double d2 = 0.0;
        if (act) {
          const BktRec r = load_rec(a, qq, gb, i);
          my = r.idx; d2 = r.d2;
          for (int j = 0; j < gk; ++j) rank += (load_rec(a, qq, gb, j).idx < my) ? 1 : 0;
          a.idx[gb + rank] = my;
          a.dist[gb + rank] = sqrt_rn(d2);
          if (a.owner) a.owner[gb + rank] = qq;
          if ((d2 < best) || (d2 == best && my < best_i)) { best = d2; best_i = my; }
          if (a.hit_out) { a.hit_out[gb + rank] = r.pad & 1; a.hit_in[gb + rank] = (r.pad >> 1) & 1; }
        }
      }
    }
    if (a.nearest_idx) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(best_i, off);
        if ((ob < best) || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
      }
      if (lane == 0) { sm.r_best[wave] = best; sm.r_besti[wave] = best_i; }
      __syncthreads();
      if (t == 0) {
        for (int w = 1; w < kFinThreads / 64; ++w)
          if ((sm.r_best[w] < best) || (sm.r_best[w] == best && sm.r_besti[w] < best_i)) {
            best = sm.r_best[w]; best_i = sm.r_besti[w];
          }
        if (gk > 0) { a.nearest_idx[qq] = best_i; a.nearest_dist[qq] = sqrt_rn(best); }
        else if (a.want_nearest_fix) atomicOr(&sm.empty_mask, 1u << g);       // list cut off by the caller's capacity
        else { a.nearest_idx[qq] = -1; a.nearest_dist[qq] = __builtin_inf(); }
      }
    }
    __syncthreads();   // LDS is reused by the next list
  }

  // ---- empty balls: kdFindNearest by expanding search (R/kdTree_general.jl:357-385) ----
  if (a.nearest_idx && a.want_nearest_fix) {
    for (unsigned em = sm.empty_mask; em != 0u; em &= em - 1u) {   // workgroup-uniform
      const int g = __ffs((int)em) - 1;
      const int qq = q0 + g;
      const double ex = a.q[(size_t)qq * D + 0], ey = a.q[(size_t)qq * D + 1], ez = a.q[(size_t)qq * D + 2];
      double ew = 0.0;
      if constexpr (D == 4) ew = a.q[(size_t)qq * D + 3];
      double bs;
      int bi;
      block_nearest<D, kFinThreads>(a.ni, ex, ey, ez, ew, a.r_start, sm.ns, bs, bi);
      if (t == 0) { a.nearest_idx[qq] = bi; a.nearest_dist[qq] = sqrt_rn(bs); }
    }
  }
}

}  // namespace

int launch_nn_finish(rrtx_ctx *ctx, const FinishLaunch &f) {
  const int D = ctx->dim;
  FinishArgs a;
  a.count = f.count; a.nq = f.nq; a.bcap = f.bcap;
  a.bkt = reinterpret_cast<const BktRec *>(f.bkt);
  a.tmp = reinterpret_cast<BktRec *>(f.tmp);
  a.ovf = reinterpret_cast<const HitRec *>(f.ovf);
  a.ovf_cap = f.ovf_cap;
  a.sc = reinterpret_cast<const Scalars *>(f.scalars);
  a.prescattered = f.prescattered ? 1 : 0;
  a.offsets = f.offsets; a.needed = f.needed;
  a.idx = f.idx; a.dist = f.dist; a.out_cap = f.out_cap;
  a.owner = f.owner; a.nearest_idx = f.nearest_idx; a.nearest_dist = f.nearest_dist;
  a.qhist = f.qhist; a.n_qhist = f.n_qhist;
  a.mailbox = f.mailbox;
  a.q = f.q;
  a.r_start = f.r_start;
  // empty balls are resolved on the device only where the nearest comes off the lists at all
  // (no wrapped dimensions: with ghosts a list key is not the distance to the query itself)
  a.want_nearest_fix = (f.nearest_idx && ctx->n_wraps == 0) ? 1 : 0;
  a.ni.sx = ctx->sl_d[0]; a.ni.sy = ctx->sl_d[1]; a.ni.sz = ctx->sl_d[2]; a.ni.sw = ctx->sl_d[D == 4 ? 3 : 2];
  a.ni.sid = ctx->sl_id;
  a.ni.chunk_ext = reinterpret_cast<const ChunkExt *>(ctx->chunk_ext);
  a.ni.n_nodes = (int)ctx->n_nodes;
  a.ni.n_chunks = (int)((ctx->n_nodes + kSlabChunk - 1) / kSlabChunk);
  a.hit_out = f.hit_out; a.hit_in = f.hit_in;
  const dim3 grid((unsigned)((f.nq + kFinQ - 1) / kFinQ)), block(kFinThreads);
  // lists of hundreds of entries and more (the caller made room for them): the build that sorts up to
  // kHugeSort entries in LDS (one workgroup per CU) instead of counting ranks over global memory
  // (measured at C3, lists of ~1400: 1.0 ms in that build against 1.7 ms in the 8-waves-per-SIMD build, whose
  //  64 registers the sort network spills -- so any call whose capacity allows for lists in the hundreds takes it)
  const bool huge = f.out_cap / (f.nq > 0 ? f.nq : 1) > 512;
  if (D == 4) {
    if (huge) hipLaunchKernelGGL((nn_finish_kernel<4, kHugeSort>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((nn_finish_kernel<4, kBigSort>), grid, block, 0, ctx->stream, a);
  } else {
    if (huge) hipLaunchKernelGGL((nn_finish_kernel<3, kHugeSort>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((nn_finish_kernel<3, kBigSort>), grid, block, 0, ctx->stream, a);
  }
  RRTX_HIP(ctx, hipGetLastError());
  return RRTX_OK;
}

}  // namespace rrtx
