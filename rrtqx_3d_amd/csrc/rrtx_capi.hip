// rrtx_capi.hip -- the extern "C" surface of librrtx_hip.so (include/rrtx.h):
// context lifetime, node/obstacle tables, host-buffer staging around the kernel
// launchers.  gfx950 only; no CPU fallback: every compute entry point runs HIP
// kernels or fails with RRTX_E_DEVICE.
#include <dlfcn.h>

#include <cstdarg>
#include <mutex>

#include "exact_math.hpp"
#include "rrtx_internal.hpp"

namespace rrtx {

int fail(rrtx_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

// ---- roctx ranges around the C-ABI calls (tracing hook) ---------------------------------
// A rocprofv3 --marker-trace of a Julia-hosted run then shows which planner call every kernel belongs
// to.  The marker library is looked up at run time (rocprofiler-sdk's roctx, else the legacy one) so the
// library carries no link dependency on it; without it the ranges are no-ops.
namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve() {
  const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
  for (const char *n : names) {
    void *h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) continue;
    roctx_push_fn push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
    roctx_pop_fn pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
    if (push && pop) { g_roctx_push = push; g_roctx_pop = pop; return; }
  }
}
}  // namespace

ApiRange::ApiRange(const char *name) {
  std::call_once(g_roctx_once, roctx_resolve);
  active = g_roctx_push != nullptr;
  if (active) (void)g_roctx_push(name);
}
ApiRange::~ApiRange() {
  if (active) (void)g_roctx_pop();
}

// ---- exact thresholds on squared distances ------------------------------------
// sqrt is monotone non-decreasing under round-to-nearest, so {s : sqrt(s) >= r}
// is an upper set of the non-negative doubles; its least element is found by a
// few nextafter steps around r*r (bisection over the bit pattern as fallback).
namespace {
template <class Pred>
double first_true(double guess, Pred pred) {
  // pred is monotone (false ... false true ... true) over s in [0, +inf]
  if (!(guess >= 0.0)) guess = 0.0;
  double s = guess;
  for (int it = 0; it < 16; ++it) {
    if (pred(s)) {
      if (s == 0.0) return 0.0;
      double p = std::nextafter(s, -INFINITY);
      if (!pred(p)) return s;
      s = p;
    } else {
      double n = std::nextafter(s, INFINITY);
      if (n == s) return std::nan("");  // s == +inf and pred false: never true
      if (pred(n)) return n;
      s = n;
    }
  }
  // bisection on the IEEE bit pattern (monotone for non-negative doubles)
  uint64_t lo = 0, hi = 0x7ff0000000000000ull;  // 0 .. +inf
  auto val = [](uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; };
  if (pred(val(lo))) return 0.0;
  if (!pred(val(hi))) return std::nan("");
  while (hi - lo > 1) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (pred(val(mid))) hi = mid; else lo = mid;
  }
  return val(hi);
}
}  // namespace

double thr_first_ge(double r) {
  if (std::isnan(r)) return std::nan("");  // sqrt(s) < NaN is never true
  if (r <= 0.0) return 0.0;                // sqrt(s) >= r for every s >= 0
  return first_true(r * r, [r](double s) { return std::sqrt(s) >= r; });
}
double thr_first_gt(double r) {
  if (std::isnan(r)) return 0.0;           // sqrt(s) <= NaN never true: s < 0 never
  if (r < 0.0) return 0.0;
  return first_true(r * r, [r](double s) { return std::sqrt(s) > r; });
}

// first s with !((sqrt(s) - robotRadius) - radius < 0): explicitPointCheck's per-sphere test
// (R/DRRT_Q.jl:1555-1570) as a threshold on the squared distance, "collision <=> s < thr".
// The expression is monotone in s (every step is a monotone rounded operation).
double thr_point_clear(double robot_radius, double radius) {
  const double g = robot_radius + radius;
  double t = first_true(g * g, [=](double s) { return !((std::sqrt(s) - robot_radius) - radius < 0.0); });
  return std::isnan(t) ? INFINITY : t;      // never clear (e.g. infinite radius): s < +inf
}

// ---- profiling spans -------------------------------------------------------------
static hipEvent_t take_event(rrtx_ctx *ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void span_begin(rrtx_ctx *ctx, int family) {
  ctx->fam_launches[family] += 1;
  // level 1 times only the dominant kernel family: every event record costs ~10 us of pipeline
  // drain between two kernels that would otherwise run back to back
  ctx->span_open = ctx->profiling == 2;
  if (ctx->profiling == 1 && family == KF_NN_SCAN) {
    // level 1 may sample: only every opt_profile_every-th launch of the family carries events
    ctx->span_tick += 1;
    ctx->span_open = ctx->span_tick % ctx->opt_profile_every == 0;
  }
  if (!ctx->span_open) { ctx->fam_launches[family] -= 1; return; }   // launches = timed launches
  TimedSpan s;
  s.a = take_event(ctx);
  s.b = take_event(ctx);
  s.family = family;
  (void)hipEventRecord(s.a, ctx->stream);
  ctx->spans.push_back(s);
}

void span_end(rrtx_ctx *ctx) {
  if (!ctx->span_open || ctx->spans.empty()) return;
  (void)hipEventRecord(ctx->spans.back().b, ctx->stream);
  ctx->span_open = false;
}

static void drain_spans(rrtx_ctx *ctx) {
  for (auto &s : ctx->spans) {
    float ms = 0.f;
    if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess)
      ctx->fam_ms[s.family] += ms;
    ctx->event_pool.push_back(s.a);
    ctx->event_pool.push_back(s.b);
  }
  ctx->spans.clear();
}

namespace {

// row-major points -> SoA.  fp64 arrays hold the coordinates as given; the fp32
// shadow (prefilter only) holds coordinates relative to the context origin plus
// |p~|^2, and absmax tracks max |p - origin| (bit pattern of a non-negative double
// orders like the value; NaN sorts above inf).
__global__ __launch_bounds__(256) void aos_to_soa_kernel(const double *__restrict__ pos, int dim, long long n,
                                                         long long base, double ox, double oy, double oz,
                                                         double ow, double *__restrict__ x,
                                                         double *__restrict__ y, double *__restrict__ z,
                                                         double *__restrict__ w, float *__restrict__ xf,
                                                         float *__restrict__ yf, float *__restrict__ zf,
                                                         float *__restrict__ wf, float *__restrict__ ppf,
                                                         unsigned long long *__restrict__ absmax,
                                                         float *__restrict__ sx, float *__restrict__ sy,
                                                         float *__restrict__ sz, float *__restrict__ sw,
                                                         float *__restrict__ spp, int32_t *__restrict__ sid,
                                                         double *__restrict__ dx, double *__restrict__ dy,
                                                         double *__restrict__ dz, double *__restrict__ dw,
                                                         double *__restrict__ aos,
                                                         ChunkExt *__restrict__ chunk_ext,
                                                         unsigned long long *__restrict__ xrange, int slab_later,
                                                         const RunRank rr) {
  // slab_later: the batch's slab entries and chunk extents are written in cell order by slab_append_run, for
  // which this kernel draws every node's (cell, rank in the cell) on the way (rr; one launch less per batch)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long m = 0ull;
  unsigned long long xlo = ~0ull, xhi = 0ull, ylo = ~0ull, yhi = 0ull, zlo = ~0ull, zhi = 0ull;
  if (i < n) {
    const double a = pos[i * dim + 0], b = pos[i * dim + 1], c = pos[i * dim + 2];
    x[base + i] = a; y[base + i] = b; z[base + i] = c;
    const double sa = a - ox, sb = b - oy, sc = c - oz;
    const float fa = (float)sa, fb = (float)sb, fc = (float)sc;
    xf[base + i] = fa; yf[base + i] = fb; zf[base + i] = fc;
    // slab index: a new node sits at position == index until the next rebuild; its chunk's
    // x extent grows accordingly (a NaN x can never be within range of anything: not tracked)
    if (!slab_later) {
      sx[base + i] = fa; sy[base + i] = fb; sz[base + i] = fc;
      dx[base + i] = a; dy[base + i] = b; dz[base + i] = c;
      sid[base + i] = (int32_t)(base + i);
    }
    if (slab_later) {
      const int cell = cell_of(a, b, rr.sp->x0, rr.sp->inv_wx, rr.sp->Kx, rr.sp->y0, rr.sp->inv_wy, rr.sp->Ky);
      rr.sr[i] = make_int2(cell, atomicAdd(&rr.hist[cell], 1));
    }
    if (a == a) xlo = xhi = enc_ord(a);
    if (b == b) ylo = yhi = enc_ord(b);
    if (c == c) zlo = zhi = enc_ord(c);
    double pp = (double)fa * (double)fa + (double)fb * (double)fb + (double)fc * (double)fc;
    m = max(max((unsigned long long)__double_as_longlong(fabs(sa)), (unsigned long long)__double_as_longlong(fabs(sb))),
            (unsigned long long)__double_as_longlong(fabs(sc)));
    if (dim == 4) {
      const double d = pos[i * dim + 3];
      w[base + i] = d;
      const double sd = d - ow;
      const float fd = (float)sd;
      wf[base + i] = fd;
      if (!slab_later) { sw[base + i] = fd; dw[base + i] = d; }
      pp += (double)fd * (double)fd;
      m = max(m, (unsigned long long)__double_as_longlong(fabs(sd)));
    }
    ppf[base + i] = (float)pp;
    if (!slab_later) spp[base + i] = (float)pp;
    reinterpret_cast<double4 *>(aos)[base + i] = make_double4(a, b, c, dim == 4 ? pos[i * dim + 3] : 0.0);
  }
  // chunk extents: the 64 positions of a wave lie in one chunk unless they straddle a chunk
  // boundary; then one lane updates the chunk with the wave's extent, otherwise every lane its own
  const long long ch = (base + i) / kSlabChunk;
  const long long ch_first = (base + ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63))) / kSlabChunk;
  const bool one_chunk = __ballot(i < n && ch != ch_first) == 0ull;
  if (!one_chunk && i < n && !slab_later) {
    if (xlo != ~0ull) { atomicMin(&chunk_ext[ch].xlo, xlo); atomicMax(&chunk_ext[ch].xhi, xhi); }
    if (ylo != ~0ull) { atomicMin(&chunk_ext[ch].ylo, ylo); atomicMax(&chunk_ext[ch].yhi, yhi); }
  }
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(m, off);
    m = max(m, o);
    o = __shfl_xor(xlo, off);
    xlo = min(xlo, o);
    o = __shfl_xor(xhi, off);
    xhi = max(xhi, o);
    o = __shfl_xor(ylo, off);
    ylo = min(ylo, o);
    o = __shfl_xor(yhi, off);
    yhi = max(yhi, o);
    o = __shfl_xor(zlo, off);
    zlo = min(zlo, o);
    o = __shfl_xor(zhi, off);
    zhi = max(zhi, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (one_chunk && !slab_later) {
      if (xlo != ~0ull) { atomicMin(&chunk_ext[ch_first].xlo, xlo); atomicMax(&chunk_ext[ch_first].xhi, xhi); }
      if (ylo != ~0ull) { atomicMin(&chunk_ext[ch_first].ylo, ylo); atomicMax(&chunk_ext[ch_first].yhi, yhi); }
    }
    // the bounds of the whole tree rarely move: look first (a stale look only costs an atomic that changes nothing)
    if (m != 0ull && m > *absmax) atomicMax(absmax, m);
    if (xlo != ~0ull) {
      if (xlo < xrange[0]) atomicMin(&xrange[0], xlo);
      if (xhi > xrange[1]) atomicMax(&xrange[1], xhi);
    }
    if (ylo != ~0ull) {
      if (ylo < xrange[2]) atomicMin(&xrange[2], ylo);
      if (yhi > xrange[3]) atomicMax(&xrange[3], yhi);
    }
    if (zlo != ~0ull) {
      if (zlo < xrange[4]) atomicMin(&xrange[4], zlo);
      if (zhi > xrange[5]) atomicMax(&xrange[5], zhi);
    }
  }
}

// reallocate one device array of the node store, keeping the first `keep` elements
template <class T>
int regrow(rrtx_ctx *ctx, T *&ptr, int64_t new_count, int64_t keep) {
  T *nb = nullptr;
  hipError_t e = hipMalloc(&nb, sizeof(T) * (size_t)new_count);
  if (e != hipSuccess)
    return fail(ctx, RRTX_E_NOMEM, "hipMalloc of %lld node slots failed: %s", (long long)new_count, hipGetErrorString(e));
  if (keep > 0 && ptr) RRTX_HIP(ctx, hipMemcpy(nb, ptr, sizeof(T) * (size_t)keep, hipMemcpyDeviceToDevice));
  if (ptr) RRTX_HIP(ctx, hipFree(ptr));
  ptr = nb;
  return RRTX_OK;
}

int grow_nodes(rrtx_ctx *ctx, int64_t need) {
  if (need <= ctx->cap_nodes) return RRTX_OK;
  int64_t nc = ctx->cap_nodes > 0 ? ctx->cap_nodes : 1024;
  while (nc < need) nc *= 2;
  if (nc > 0x7fffffffll) return fail(ctx, RRTX_E_CAPACITY, "node index space is int32 (%lld requested)", (long long)need);
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc;
  for (int k = 0; k < ctx->dim; ++k) {
    if ((rc = regrow(ctx, ctx->nodes[k], nc, ctx->n_nodes))) return rc;
    if ((rc = regrow(ctx, ctx->nodes_f[k], nc, ctx->n_nodes))) return rc;
    if ((rc = regrow(ctx, ctx->sl_f[k], nc, ctx->n_nodes))) return rc;
    if ((rc = regrow(ctx, ctx->sl_d[k], nc, ctx->n_nodes))) return rc;
  }
  if ((rc = regrow(ctx, ctx->nodes_pp, nc, ctx->n_nodes))) return rc;
  {
    double *old_aos = ctx->nodes_aos;
    double *nb = nullptr;
    hipError_t e = hipMalloc(&nb, sizeof(double) * 4 * (size_t)nc);
    if (e != hipSuccess) return fail(ctx, RRTX_E_NOMEM, "hipMalloc of %lld node slots failed: %s", (long long)nc, hipGetErrorString(e));
    if (ctx->n_nodes > 0 && old_aos)
      RRTX_HIP(ctx, hipMemcpy(nb, old_aos, sizeof(double) * 4 * (size_t)ctx->n_nodes, hipMemcpyDeviceToDevice));
    if (old_aos) RRTX_HIP(ctx, hipFree(old_aos));
    ctx->nodes_aos = nb;
  }
  if ((rc = regrow(ctx, ctx->sl_pp, nc, ctx->n_nodes))) return rc;
  if ((rc = regrow(ctx, ctx->sl_id, nc, ctx->n_nodes))) return rc;
  {
    // chunk extents: new chunks start empty (min = ~0, max = 0)
    const int64_t nch = nc / kSlabChunk + 1;
    const int64_t old = ctx->cap_chunks;
    if ((rc = regrow(ctx, ctx->chunk_ext, nch, old))) return rc;
    std::vector<ChunkExtHost> empty((size_t)(nch - old), ChunkExtHost{~0ull, 0ull, ~0ull, 0ull});
    RRTX_HIP(ctx, hipMemcpy(ctx->chunk_ext + old, empty.data(), sizeof(ChunkExtHost) * empty.size(), hipMemcpyHostToDevice));
    ctx->cap_chunks = nch;
  }
  if (!ctx->d_absmax.p) {
    RRTX_HIP(ctx, ctx->d_absmax.ensure(sizeof(unsigned long long)));
    RRTX_HIP(ctx, hipMemset(ctx->d_absmax.p, 0, sizeof(unsigned long long)));
    const unsigned long long none[6] = {~0ull, 0ull, ~0ull, 0ull, ~0ull, 0ull};
    RRTX_HIP(ctx, ctx->d_xrange.ensure(sizeof(none)));
    RRTX_HIP(ctx, hipMemcpy(ctx->d_xrange.p, none, sizeof(none), hipMemcpyHostToDevice));
  }
  ctx->cap_nodes = nc;
  return RRTX_OK;
}

// ---- host <-> device transfers of the host-pointer entry points ----
static bool host_registered(const rrtx_ctx *ctx, const void *p, size_t bytes) {
  const char *c = static_cast<const char *>(p);
  for (const auto &r : ctx->h_registered)
    if (c >= r.first && c + bytes <= r.first + r.second) return true;
  return false;
}
// room for `bytes` more in the pinned arena; a call reserves its whole need up front (arena_begin), so the arena never
// moves while copies into it are in flight
static int arena_begin(rrtx_ctx *ctx, size_t need) {
  ctx->h_arena_used = 0;
  ctx->h_pending.clear();
  if (need <= ctx->h_arena_bytes) return RRTX_OK;
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_arena) (void)hipHostFree(ctx->h_arena);
  ctx->h_arena = nullptr; ctx->h_arena_bytes = 0;
  size_t nb = 1 << 20;
  while (nb < need) nb *= 2;
  RRTX_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_arena), nb, hipHostMallocDefault));
  ctx->h_arena_bytes = nb;
  return RRTX_OK;
}
static char *arena_take(rrtx_ctx *ctx, size_t bytes) {
  const size_t at = (ctx->h_arena_used + 63) & ~(size_t)63;
  if (at + bytes > ctx->h_arena_bytes) return nullptr;
  ctx->h_arena_used = at + bytes;
  return ctx->h_arena + at;
}
// device -> caller array: straight DMA when the array is registered, else through the arena (copied out by
// arena_flush after the stream has been synchronised)
// Only SMALL results are staged (a 16 KB pageable destination costs a driver-side bounce of its own, ~35 us against
// ~20 us pinned): measured on the MI355X box (tools/pcie_probe.py), a pageable destination of a megabyte or more
// receives its DMA at the same ~50 GB/s as a pinned one, while a staged copy has to be read back from DRAM by one host
// thread at a third of that.
static int d2h(rrtx_ctx *ctx, void *dst, const void *src_dev, size_t bytes) {
  if (!bytes) return RRTX_OK;
  void *to = dst;
  if (bytes <= (256u << 10) && !host_registered(ctx, dst, bytes)) {
    char *a = arena_take(ctx, bytes);
    if (a) { to = a; ctx->h_pending.push_back({dst, a, bytes}); }
  }
  RRTX_HIP(ctx, hipMemcpyAsync(to, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return RRTX_OK;
}
static void arena_flush(rrtx_ctx *ctx) {
  for (const auto &c : ctx->h_pending) std::memcpy(c.dst, c.src, c.bytes);
  ctx->h_pending.clear();
}

// stage a host array on the device workspace `buf`
int stage_in(rrtx_ctx *ctx, DevBuf &buf, const void *host, size_t bytes) {
  RRTX_HIP(ctx, buf.ensure(bytes ? bytes : 8));
  if (bytes) RRTX_HIP(ctx, hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return RRTX_OK;
}
// the samples of a batched call: through the pinned arena when they are small (one host memcpy of cache-resident data,
// then a pinned H2D), so the driver does not bounce them itself.  Call after arena_begin.
static int stage_in_small(rrtx_ctx *ctx, DevBuf &buf, const void *host, size_t bytes) {
  RRTX_HIP(ctx, buf.ensure(bytes ? bytes : 8));
  if (!bytes) return RRTX_OK;
  const void *from = host;
  if (bytes <= (1u << 20) && !host_registered(ctx, host, bytes)) {
    char *a = arena_take(ctx, bytes);
    if (a) { std::memcpy(a, host, bytes); from = a; }
  }
  RRTX_HIP(ctx, hipMemcpyAsync(buf.p, from, bytes, hipMemcpyHostToDevice, ctx->stream));
  return RRTX_OK;
}

std::string g_create_err;
std::mutex g_create_mu;

}  // namespace
}  // namespace rrtx

using namespace rrtx;

#define CHECK_CTX(ctx)                 \
  ::rrtx::ApiRange _rrtx_api_range(__func__); \
  do {                                 \
    if (!(ctx)) return RRTX_E_INVALID; \
    hipError_t _sd = hipSetDevice((ctx)->device); \
    if (_sd != hipSuccess) return fail((ctx), RRTX_E_DEVICE, "hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(_sd)); \
  } while (0)

extern "C" {

const char *rrtx_create_error(void) { return g_create_err.c_str(); }

int rrtx_sq_thresholds(double r, double *first_ge, double *first_gt) {
  if (!first_ge || !first_gt) return RRTX_E_INVALID;
  *first_ge = thr_first_ge(r);
  *first_gt = thr_first_gt(r);
  return RRTX_OK;
}

int rrtx_create(rrtx_ctx **out, int dim, int device, int64_t node_capacity) {
  std::lock_guard<std::mutex> lk(g_create_mu);
  if (!out) { g_create_err = "out is NULL"; return RRTX_E_INVALID; }
  *out = nullptr;
  if (dim != 3 && dim != 4) { g_create_err = "dim must be 3 or 4"; return RRTX_E_INVALID; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("no HIP device available: ") + hipGetErrorString(e);
    return RRTX_E_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_create_err = "device ordinal out of range"; return RRTX_E_INVALID; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return RRTX_E_DEVICE; }
  rrtx_ctx *ctx = new (std::nothrow) rrtx_ctx();
  if (!ctx) { g_create_err = "out of host memory"; return RRTX_E_NOMEM; }
  ctx->dim = dim;
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    g_create_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
    delete ctx;
    return RRTX_E_DEVICE;
  }
  ctx->stream = ctx->own_stream;
  int rc = grow_nodes(ctx, node_capacity > 0 ? node_capacity : 1024);
  if (rc != RRTX_OK) {
    g_create_err = ctx->err;
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return rc;
  }
  *out = ctx;
  return RRTX_OK;
}

int rrtx_destroy(rrtx_ctx *ctx) {
  if (!ctx) return RRTX_E_INVALID;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  drain_spans(ctx);
  for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
  for (int k = 0; k < 4; ++k)
    if (ctx->nodes[k]) (void)hipFree(ctx->nodes[k]);
  for (int k = 0; k < 4; ++k)
    if (ctx->nodes_f[k]) (void)hipFree(ctx->nodes_f[k]);
  if (ctx->nodes_pp) (void)hipFree(ctx->nodes_pp);
  if (ctx->nodes_aos) (void)hipFree(ctx->nodes_aos);
  for (int k = 0; k < 4; ++k)
    if (ctx->sl_f[k]) (void)hipFree(ctx->sl_f[k]);
  for (int k = 0; k < 4; ++k)
    if (ctx->sl_d[k]) (void)hipFree(ctx->sl_d[k]);
  if (ctx->sl_pp) (void)hipFree(ctx->sl_pp);
  if (ctx->sl_id) (void)hipFree(ctx->sl_id);
  if (ctx->chunk_ext) (void)hipFree(ctx->chunk_ext);
  ctx->d_absmax.release();
  ctx->d_xrange.release();
  DevBuf *bufs[] = {&ctx->d_sph, &ctx->d_sph_reach, &ctx->d_sph_reach_f, &ctx->d_sph_aux, &ctx->d_poly_off, &ctx->d_poly_vxy, &ctx->d_poly_slope, &ctx->d_poly_meta,
                    &ctx->d_poly_orig, &ctx->d_poly_bbox, &ctx->d_poly_ytab, &ctx->d_poly_pbox, &ctx->d_poly_grid_start, &ctx->d_poly_grid_items, &ctx->d_poly_path_off, &ctx->d_poly_path, &ctx->ws_knn_off, &ctx->ws_knn_idx,
                    &ctx->ws_knn_dist, &ctx->ws_knn_misc, &ctx->ws_q, &ctx->ws_q2, &ctx->ws_slots, &ctx->ws_copies,
                    &ctx->ws_copy_meta, &ctx->ws_copies_f, &ctx->ws_recs, &ctx->ws_counts, &ctx->ws_bsum, &ctx->ws_scalars, &ctx->ws_scalars_nn, &ctx->ws_tmp,
                    &ctx->ws_owner, &ctx->ws_dub_rec, &ctx->ws_out_off, &ctx->ws_out_idx, &ctx->ws_out_dist, &ctx->ws_out_u8a,
                    &ctx->ws_out_u8b, &ctx->ws_out_i32, &ctx->ws_out_f64, &ctx->ws_partial, &ctx->ws_thr, &ctx->ws_mask, &ctx->ws_i32a, &ctx->ws_i32b,
                    &ctx->ws_slab_hist, &ctx->ws_slab_start, &ctx->ws_slab_sr, &ctx->ws_slab_params, &ctx->ws_run_hist, &ctx->ws_run_sr,
                    &ctx->ws_copies_s,
                    &ctx->ws_meta_s, &ctx->ws_cb, &ctx->ws_qhist, &ctx->ws_bkt,
                    &ctx->ws_ev_a, &ctx->ws_ev_cnt, &ctx->ws_confirm_args, &ctx->ws_sph_lists, &ctx->ws_poly_lists, &ctx->d_sph_sample,
                    &ctx->ws_sweep_mark, &ctx->ws_sweep_flag, &ctx->ws_sweep_cnt, &ctx->ws_sweep_start,
                    &ctx->gc.lmc, &ctx->gc.parent, &ctx->gc.stamp, &ctx->gc.flags, &ctx->gc.orph, &ctx->gc.anc, &ctx->gc.ids,
                    &ctx->gc.in_cnt, &ctx->gc.in_start, &ctx->gc.in_cursor, &ctx->gc.in_tiles, &ctx->gc.in_src, &ctx->gc.in_w,
                    &ctx->gc.in_pos};
  for (auto b : bufs) b->release();
  if (ctx->ge_start) (void)hipFree(ctx->ge_start);
  if (ctx->ge_end) (void)hipFree(ctx->ge_end);
  if (ctx->ge_dist) (void)hipFree(ctx->ge_dist);
  if (ctx->ge_dirty) (void)hipFree(ctx->ge_dirty);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->h_arena) (void)hipHostFree(ctx->h_arena);
  for (const auto &r : ctx->h_registered) (void)hipHostUnregister(const_cast<char *>(r.first));
  (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return RRTX_OK;
}

const char *rrtx_last_error(rrtx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rrtx_set_stream(rrtx_ctx *ctx, void *hip_stream) {
  CHECK_CTX(ctx);
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return RRTX_OK;
}

void *rrtx_get_stream(rrtx_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int rrtx_sync(rrtx_ctx *ctx) {
  CHECK_CTX(ctx);
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_profile(rrtx_ctx *ctx, int enable) {
  CHECK_CTX(ctx);
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  drain_spans(ctx);
  for (int k = 0; k < KF_COUNT; ++k) { ctx->fam_ms[k] = 0.0; ctx->fam_launches[k] = 0; }
  ctx->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return RRTX_OK;
}

int rrtx_set_option(rrtx_ctx *ctx, int option, int64_t value) {
  CHECK_CTX(ctx);
  switch (option) {
    case RRTX_OPT_NN_FILTER: ctx->opt_nn_filter = value != 0; return RRTX_OK;
    case RRTX_OPT_SCAN_BLOCKS: ctx->opt_scan_blocks = value > 0 ? (int)value : 1280; return RRTX_OK;
    case RRTX_OPT_SCAN_ITEMS: ctx->opt_scan_items = value > 0 ? (int)value : 2048; return RRTX_OK;
    case RRTX_OPT_SCAN_TILE_Q: ctx->opt_tile_q = value > 0 ? (int)value : 0; return RRTX_OK;
    case RRTX_OPT_NN_CULL: ctx->opt_nn_cull = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return RRTX_OK;
    case RRTX_OPT_PROFILE_EVERY: ctx->opt_profile_every = value > 1 ? (int)value : 1; return RRTX_OK;
    case RRTX_OPT_KNN_LISTS: ctx->opt_knn_lists = value != 0; return RRTX_OK;
    case RRTX_OPT_EXTEND_OBSTACLES: ctx->opt_extend_polygons = value == 1; return RRTX_OK;
    case RRTX_OPT_TUNE: ctx->opt_tune = (int)value; return RRTX_OK;
    case RRTX_OPT_ROOT_RULE: ctx->opt_root_rule = value != 0; return RRTX_OK;
    case RRTX_OPT_SPACE_HAS_TIME:
      if (value != 0 && ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "a space with time is [x y t theta]: dim = 4");
      ctx->opt_space_has_time = value != 0;
      return RRTX_OK;
    case RRTX_OPT_NEAREST_REC_CAP: ctx->opt_nearest_rec_cap = value > 0 ? (long long)value : 0; return RRTX_OK;
    case RRTX_OPT_BUCKET_MULT: {
      int m = 2;
      while (m < value && m < 16) m *= 2;
      ctx->bkt_mult = m;
      return RRTX_OK;
    }
    default: return fail(ctx, RRTX_E_INVALID, "set_option: unknown option %d", option);
  }
}

int rrtx_host_register(rrtx_ctx *ctx, void *ptr, size_t bytes) {
  CHECK_CTX(ctx);
  if (!ptr || bytes == 0) return fail(ctx, RRTX_E_INVALID, "host_register: bad arguments");
  if (host_registered(ctx, ptr, bytes)) return RRTX_OK;
  RRTX_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  ctx->h_registered.push_back({static_cast<const char *>(ptr), bytes});
  return RRTX_OK;
}

int rrtx_host_unregister(rrtx_ctx *ctx, void *ptr) {
  CHECK_CTX(ctx);
  for (size_t k = 0; k < ctx->h_registered.size(); ++k)
    if (ctx->h_registered[k].first == static_cast<const char *>(ptr)) {
      RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      RRTX_HIP(ctx, hipHostUnregister(ptr));
      ctx->h_registered.erase(ctx->h_registered.begin() + (long)k);
      return RRTX_OK;
    }
  return fail(ctx, RRTX_E_INVALID, "host_unregister: %p was not registered with this context", ptr);
}

int rrtx_get_option(rrtx_ctx *ctx, int option, int64_t *value) {
  CHECK_CTX(ctx);
  if (!value) return fail(ctx, RRTX_E_INVALID, "get_option: null output");
  switch (option) {
    case RRTX_OPT_NN_FILTER: *value = ctx->opt_nn_filter ? 1 : 0; return RRTX_OK;
    case RRTX_OPT_SCAN_BLOCKS: *value = ctx->opt_scan_blocks; return RRTX_OK;
    case RRTX_OPT_SCAN_ITEMS: *value = ctx->opt_scan_items; return RRTX_OK;
    case RRTX_OPT_SCAN_TILE_Q: *value = ctx->opt_tile_q; return RRTX_OK;
    case RRTX_OPT_NN_CULL: *value = ctx->opt_nn_cull; return RRTX_OK;
    case RRTX_OPT_PROFILE_EVERY: *value = ctx->opt_profile_every; return RRTX_OK;
    case RRTX_OPT_KNN_LISTS: *value = ctx->opt_knn_lists ? 1 : 0; return RRTX_OK;
    case RRTX_OPT_EXTEND_OBSTACLES: *value = ctx->opt_extend_polygons ? 1 : 0; return RRTX_OK;
    case RRTX_OPT_TUNE: *value = ctx->opt_tune; return RRTX_OK;
    case RRTX_OPT_ROOT_RULE: *value = ctx->opt_root_rule ? 1 : 0; return RRTX_OK;
    case RRTX_OPT_SPACE_HAS_TIME: *value = ctx->opt_space_has_time ? 1 : 0; return RRTX_OK;
    case RRTX_OPT_NEAREST_REC_CAP: *value = (int64_t)ctx->opt_nearest_rec_cap; return RRTX_OK;
    case RRTX_OPT_BUCKET_MULT: *value = ctx->bkt_mult; return RRTX_OK;
    default: return fail(ctx, RRTX_E_INVALID, "get_option: unknown option %d", option);
  }
}

int rrtx_stats(rrtx_ctx *ctx, rrtx_stats_t *out) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, RRTX_E_INVALID, "stats: out is NULL");
  drain_spans(ctx);
  out->n_nodes = ctx->n_nodes;
  out->dim = ctx->dim;
  out->n_spheres = (int32_t)ctx->sph_active.size();
  out->n_polygons = (int32_t)ctx->poly_active.size();
  out->n_wraps = ctx->n_wraps;
  out->ms_nn_scan = ctx->fam_ms[KF_NN_SCAN];       out->launches_nn_scan = ctx->fam_launches[KF_NN_SCAN];
  out->ms_nn_finish = ctx->fam_ms[KF_NN_FINISH];   out->launches_nn_finish = ctx->fam_launches[KF_NN_FINISH];
  out->ms_nn_nearest = ctx->fam_ms[KF_NN_NEAREST]; out->launches_nn_nearest = ctx->fam_launches[KF_NN_NEAREST];
  out->ms_edges = ctx->fam_ms[KF_EDGES];           out->launches_edges = ctx->fam_launches[KF_EDGES];
  out->ms_points = ctx->fam_ms[KF_POINTS];         out->launches_points = ctx->fam_launches[KF_POINTS];
  out->ms_dubins = ctx->fam_ms[KF_DUBINS];         out->launches_dubins = ctx->fam_launches[KF_DUBINS];
  out->ms_dubins_steer = ctx->fam_ms[KF_DUBINS_STEER]; out->launches_dubins_steer = ctx->fam_launches[KF_DUBINS_STEER];
  out->last_sweep_candidates = ctx->last_sweep_candidates;
  out->last_pairs = ctx->last_pairs;
  out->last_neighbors = ctx->last_neighbors;
  out->last_tile_q = ctx->last_tile_q;
  out->last_scan_units = 0;
  if (ctx->last_culled) {
    int u = 0;
    int rc = scan_units(ctx, &u);
    if (rc) return rc;
    out->last_scan_units = u;
  }
  return RRTX_OK;
}

// ---- tree ---------------------------------------------------------------------------
int64_t rrtx_nodes_count(rrtx_ctx *ctx) { return ctx ? ctx->n_nodes : 0; }

int rrtx_nodes_append_dev(rrtx_ctx *ctx, const double *pos_dev, int64_t n) {
  CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && !pos_dev)) return fail(ctx, RRTX_E_INVALID, "nodes_append: bad arguments");
  if (n == 0) return RRTX_OK;
  int rc = grow_nodes(ctx, ctx->n_nodes + n);
  if (rc) return rc;
  if (!ctx->origin_set) {
    // origin of the fp32 shadow = the first node (the tree root); one small read-back, once per ctx
    double first[4] = {0, 0, 0, 0};
    RRTX_HIP(ctx, hipMemcpyAsync(first, pos_dev, sizeof(double) * ctx->dim, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < ctx->dim; ++k) ctx->origin[k] = std::isfinite(first[k]) ? first[k] : 0.0;
    ctx->origin_set = true;
  }
  const bool as_run = slab_run_wanted(ctx, n);
  RunRank rr;
  rr.sp = nullptr; rr.hist = nullptr; rr.sr = nullptr;
  if (as_run) {
    rc = slab_run_prepare(ctx, n, &rr);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, pos_dev,
                     ctx->dim, (long long)n, (long long)ctx->n_nodes, ctx->origin[0], ctx->origin[1], ctx->origin[2],
                     ctx->origin[3], ctx->nodes[0], ctx->nodes[1], ctx->nodes[2],
                     ctx->nodes[ctx->dim == 4 ? 3 : 2], ctx->nodes_f[0], ctx->nodes_f[1], ctx->nodes_f[2],
                     ctx->nodes_f[ctx->dim == 4 ? 3 : 2], ctx->nodes_pp, ctx->d_absmax.as<unsigned long long>(),
                     ctx->sl_f[0], ctx->sl_f[1], ctx->sl_f[2], ctx->sl_f[ctx->dim == 4 ? 3 : 2], ctx->sl_pp, ctx->sl_id,
                     ctx->sl_d[0], ctx->sl_d[1], ctx->sl_d[2], ctx->sl_d[ctx->dim == 4 ? 3 : 2], ctx->nodes_aos,
                     reinterpret_cast<ChunkExt *>(ctx->chunk_ext), ctx->d_xrange.as<unsigned long long>(), as_run ? 1 : 0, rr);
  RRTX_HIP(ctx, hipGetLastError());
  if (as_run) {
    rc = slab_append_run(ctx, ctx->n_nodes, n);
    if (rc) return rc;
  }
  ctx->n_nodes += n;
  return RRTX_OK;
}

int rrtx_nodes_append(rrtx_ctx *ctx, const double *pos, int64_t n, int64_t *first_index) {
  CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && !pos)) return fail(ctx, RRTX_E_INVALID, "nodes_append: bad arguments");
  if (first_index) *first_index = ctx->n_nodes;
  if (n == 0) return RRTX_OK;
  int rc = stage_in(ctx, ctx->ws_q, pos, sizeof(double) * (size_t)n * ctx->dim);
  if (rc) return rc;
  rc = rrtx_nodes_append_dev(ctx, ctx->ws_q.as<double>(), n);
  if (rc) return rc;
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_set_wrap(rrtx_ctx *ctx, int dim_index, double period) {
  CHECK_CTX(ctx);
  if (dim_index < 0 || dim_index >= ctx->dim || !(period > 0.0))
    return fail(ctx, RRTX_E_INVALID, "set_wrap: dimension %d / period %g invalid", dim_index, period);
  for (int w = 0; w < ctx->n_wraps; ++w)
    if (ctx->wrap_dim[w] == dim_index) { ctx->wrap_period[w] = period; return RRTX_OK; }
  if (ctx->n_wraps >= kMaxWraps) return fail(ctx, RRTX_E_CAPACITY, "at most %d wrapped dimensions", kMaxWraps);
  ctx->wrap_dim[ctx->n_wraps] = dim_index;
  ctx->wrap_period[ctx->n_wraps] = period;
  ctx->n_wraps += 1;
  return RRTX_OK;
}

// ---- obstacles -----------------------------------------------------------------------
int rrtx_spheres_set(rrtx_ctx *ctx, const double *cxyzr, const uint8_t *active, int m) {
  CHECK_CTX(ctx);
  if (m < 0 || (m > 0 && !cxyzr)) return fail(ctx, RRTX_E_INVALID, "spheres_set: bad arguments");
  ctx->sph.assign(cxyzr, cxyzr + 4 * (size_t)m);
  ctx->sph_active.assign((size_t)m, 1);
  if (active) for (int i = 0; i < m; ++i) ctx->sph_active[i] = active[i] ? 1 : 0;
  ctx->sph_dirty = true;
  return RRTX_OK;
}

int rrtx_obstacle_update(rrtx_ctx *ctx, int which, double radius, uint8_t active) {
  CHECK_CTX(ctx);
  if (which < 0 || which >= (int)ctx->sph_active.size())
    return fail(ctx, RRTX_E_INVALID, "obstacle_update: index %d out of range", which);
  ctx->sph[4 * (size_t)which + 3] = radius;
  ctx->sph_active[which] = active ? 1 : 0;
  ctx->sph_dirty = true;
  return RRTX_OK;
}

int rrtx_polygons_set(rrtx_ctx *ctx, const int32_t *vert_off, const double *vxy, const double *centre_radius,
                      const uint8_t *kind, const uint8_t *active, int m) {
  CHECK_CTX(ctx);
  if (m < 0 || (m > 0 && (!vert_off || !vxy))) return fail(ctx, RRTX_E_INVALID, "polygons_set: bad arguments");
  for (int i = 0; i < m; ++i)
    if (vert_off[i + 1] < vert_off[i]) return fail(ctx, RRTX_E_INVALID, "polygons_set: vert_off not monotone");
  ctx->poly_off.assign(vert_off, vert_off + (m > 0 ? m + 1 : 0));
  if (m == 0) ctx->poly_off.assign(1, 0);
  const int nv = m > 0 ? vert_off[m] : 0;
  ctx->poly_vxy.assign(vxy, vxy + 2 * (size_t)nv);
  ctx->poly_kind.assign((size_t)m, 3);
  ctx->poly_active.assign((size_t)m, 1);
  if (kind) for (int i = 0; i < m; ++i) ctx->poly_kind[i] = kind[i];
  if (active) for (int i = 0; i < m; ++i) ctx->poly_active[i] = active[i] ? 1 : 0;
  ctx->poly_cr.assign(3 * (size_t)m, 0.0);
  for (int i = 0; i < m; ++i) {
    const int kd = ctx->poly_kind[i];
    if (kd != 1 && kd != 3 && kd != 6 && kd != 7)
      return fail(ctx, RRTX_E_INVALID, "polygons_set: obstacle kind %d not supported (1, 3, 6 or 7)", kd);
    if (centre_radius) {
      for (int k = 0; k < 3; ++k) ctx->poly_cr[3 * (size_t)i + k] = centre_radius[3 * (size_t)i + k];
      continue;
    }
    // Obstacle(kind, polygon) ctor, R/DRRT_data_structures.jl:229-241
    const int b = vert_off[i], e = vert_off[i + 1];
    if (e <= b) continue;
    double maxx = vxy[2 * b], minx = maxx, maxy = vxy[2 * b + 1], miny = maxy;
    for (int v = b + 1; v < e; ++v) {
      maxx = std::fmax(maxx, vxy[2 * v]); minx = std::fmin(minx, vxy[2 * v]);
      maxy = std::fmax(maxy, vxy[2 * v + 1]); miny = std::fmin(miny, vxy[2 * v + 1]);
    }
    const double px = (maxx + minx) / 2.0, py = (maxy + miny) / 2.0;
    double best = 0.0;
    for (int v = b; v < e; ++v) {
      double dx = vxy[2 * v] - px, dy = vxy[2 * v + 1] - py;
      double s = dx * dx + dy * dy;
      if (v == b || s > best) best = s;
    }
    ctx->poly_cr[3 * (size_t)i + 0] = px;
    ctx->poly_cr[3 * (size_t)i + 1] = py;
    ctx->poly_cr[3 * (size_t)i + 2] = std::sqrt(best);
  }
  ctx->poly_path_off.assign((size_t)m + 1, 0);
  ctx->poly_path.clear();
  ctx->poly_dirty = true;
  return RRTX_OK;
}

int rrtx_polygon_paths_set(rrtx_ctx *ctx, const int32_t *path_off, const double *path_xyt, int m) {
  CHECK_CTX(ctx);
  if (m != (int)ctx->poly_kind.size())
    return fail(ctx, RRTX_E_INVALID, "polygon_paths_set: %d paths for %d polygons", m, (int)ctx->poly_kind.size());
  if (m > 0 && !path_off) return fail(ctx, RRTX_E_INVALID, "polygon_paths_set: bad arguments");
  if (m == 0) return RRTX_OK;
  if (path_off[0] != 0) return fail(ctx, RRTX_E_INVALID, "polygon_paths_set: path_off must start at 0");
  for (int i = 0; i < m; ++i)
    if (path_off[i + 1] < path_off[i]) return fail(ctx, RRTX_E_INVALID, "polygon_paths_set: path_off not monotone");
  if (path_off[m] > 0 && !path_xyt) return fail(ctx, RRTX_E_INVALID, "polygon_paths_set: bad arguments");
  ctx->poly_path_off.assign(path_off, path_off + m + 1);
  ctx->poly_path.assign(path_xyt, path_xyt + 3 * (size_t)path_off[m]);
  ctx->poly_dirty = true;
  return RRTX_OK;
}

// ---- nearest neighbours ----------------------------------------------------------------
int rrtx_nn_nearest_dev(rrtx_ctx *ctx, const double *q, int nq, int32_t *idx, double *dist) {
  CHECK_CTX(ctx);
  if (nq < 0 || (nq > 0 && (!q || !idx || !dist))) return fail(ctx, RRTX_E_INVALID, "nn_nearest: bad arguments");
  return launch_nn_nearest(ctx, q, nq, idx, dist);
}

int rrtx_nn_nearest(rrtx_ctx *ctx, const double *q, int nq, int32_t *idx, double *dist) {
  CHECK_CTX(ctx);
  if (nq < 0 || (nq > 0 && (!q || !idx || !dist))) return fail(ctx, RRTX_E_INVALID, "nn_nearest: bad arguments");
  if (nq == 0) return RRTX_OK;
  int rc = stage_in(ctx, ctx->ws_q, q, sizeof(double) * (size_t)nq * ctx->dim);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_idx.ensure(sizeof(int32_t) * (size_t)nq));
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)nq));
  rc = launch_nn_nearest(ctx, ctx->ws_q.as<double>(), nq, ctx->ws_out_idx.as<int32_t>(), ctx->ws_out_dist.as<double>());
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(idx, ctx->ws_out_idx.p, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(dist, ctx->ws_out_dist.p, sizeof(double) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_nn_knearest_dev(rrtx_ctx *ctx, const double *q, int nq, int k, int32_t *idx, double *dist, int32_t *count) {
  CHECK_CTX(ctx);
  if (nq < 0 || (nq > 0 && (!q || !idx || !dist || !count)))
    return fail(ctx, RRTX_E_INVALID, "nn_knearest: bad arguments");
  return launch_nn_knearest(ctx, q, nq, k, idx, dist, count);
}

int rrtx_nn_knearest(rrtx_ctx *ctx, const double *q, int nq, int k, int32_t *idx, double *dist, int32_t *count) {
  CHECK_CTX(ctx);
  if (nq < 0 || (nq > 0 && (!q || !idx || !dist || !count)))
    return fail(ctx, RRTX_E_INVALID, "nn_knearest: bad arguments");
  if (nq == 0) return RRTX_OK;
  const size_t stride = (size_t)(k < 2 ? 2 : k);
  int rc = stage_in(ctx, ctx->ws_q, q, sizeof(double) * (size_t)nq * ctx->dim);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_idx.ensure(sizeof(int32_t) * ((size_t)nq * stride + (size_t)nq)));
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)nq * stride));
  int32_t *idx_dev = ctx->ws_out_idx.as<int32_t>();
  int32_t *count_dev = idx_dev + (size_t)nq * stride;
  rc = launch_nn_knearest(ctx, ctx->ws_q.as<double>(), nq, k, idx_dev, ctx->ws_out_dist.as<double>(), count_dev);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(idx, idx_dev, sizeof(int32_t) * (size_t)nq * stride, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(dist, ctx->ws_out_dist.p, sizeof(double) * (size_t)nq * stride, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(count, count_dev, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_nn_radius_dev(rrtx_ctx *ctx, const double *q, double r, int nq, int64_t *offsets, int32_t *idx,
                       double *dist, int64_t cap, int64_t *needed_dev) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !offsets)) || (cap > 0 && (!idx || !dist)))
    return fail(ctx, RRTX_E_INVALID, "nn_radius: bad arguments");
  return launch_nn_radius(ctx, q, nullptr, r, nq, offsets, idx, dist, cap, needed_dev);
}

int rrtx_nn_radius(rrtx_ctx *ctx, const double *q, const double *r, int r_stride, int nq, int64_t *offsets,
                   int32_t *idx, double *dist, int64_t cap, int64_t *needed) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !r || !offsets)) || (cap > 0 && (!idx || !dist)) ||
      (r_stride != 0 && r_stride != 1))
    return fail(ctx, RRTX_E_INVALID, "nn_radius: bad arguments");
  if (nq == 0) { if (needed) *needed = 0; if (offsets) offsets[0] = 0; return RRTX_OK; }
  int rc = stage_in(ctx, ctx->ws_q, q, sizeof(double) * (size_t)nq * ctx->dim);
  if (rc) return rc;
  const double *thr_dev = nullptr;
  if (r_stride == 1) {
    std::vector<double> thr(2 * (size_t)nq);
    double last_r = std::nan(""), lt = 0, gt = 0;
    for (int i = 0; i < nq; ++i) {
      if (!(r[i] == last_r)) { last_r = r[i]; lt = thr_first_ge(r[i]); gt = thr_first_gt(r[i]); }
      thr[i] = lt; thr[(size_t)nq + i] = gt;
    }
    rc = stage_in(ctx, ctx->ws_thr, thr.data(), sizeof(double) * thr.size());
    if (rc) return rc;
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // thr goes out of scope
    thr_dev = ctx->ws_thr.as<double>();
  }
  const int64_t dcap = cap > 0 ? cap : 1;
  RRTX_HIP(ctx, ctx->ws_out_off.ensure(sizeof(int64_t) * ((size_t)nq + 2)));
  RRTX_HIP(ctx, ctx->ws_out_idx.ensure(sizeof(int32_t) * (size_t)dcap));
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)dcap));
  int64_t *off_dev = ctx->ws_out_off.as<int64_t>();
  int64_t *needed_dev = off_dev + nq + 1;
  rc = launch_nn_radius(ctx, ctx->ws_q.as<double>(), thr_dev, r[0], nq, off_dev, ctx->ws_out_idx.as<int32_t>(),
                        ctx->ws_out_dist.as<double>(), cap, needed_dev);
  if (rc) return rc;
  int64_t total = 0;
  RRTX_HIP(ctx, hipMemcpyAsync(offsets, off_dev, sizeof(int64_t) * ((size_t)nq + 1), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(&total, needed_dev, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (needed) *needed = total;
  ctx->last_neighbors = total;
  if (total > cap) return fail(ctx, RRTX_E_CAPACITY, "nn_radius: %lld neighbours, capacity %lld", (long long)total, (long long)cap);
  if (total > 0) {
    RRTX_HIP(ctx, hipMemcpyAsync(idx, ctx->ws_out_idx.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(dist, ctx->ws_out_dist.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return RRTX_OK;
}

// ---- collision ----------------------------------------------------------------------------
int rrtx_edges_check_dev(rrtx_ctx *ctx, int kind, const double *p0, const double *p1, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, int obs_begin, int obs_end, uint8_t *hit,
                         int32_t *first_hit) {
  CHECK_CTX(ctx);
  if (ne < 0 || (ne > 0 && (!p0 || !p1 || !hit)) || (kind != 0 && kind != 1))
    return fail(ctx, RRTX_E_INVALID, "edges_check: bad arguments");
  if (kind == 0)
    return launch_edges_spheres(ctx, p0, p1, ne, robot_radius, obstacle_or_minus1, obs_begin, obs_end, hit, first_hit);
  return launch_edges_polygons(ctx, p0, p1, ne, robot_radius, obstacle_or_minus1, obs_begin, obs_end, hit, first_hit);
}

int rrtx_edges_check(rrtx_ctx *ctx, int kind, const double *p0, const double *p1, int64_t ne, double robot_radius,
                     int obstacle_or_minus1, uint8_t *hit, int32_t *first_hit) {
  CHECK_CTX(ctx);
  if (ne < 0 || (ne > 0 && (!p0 || !p1 || !hit)) || (kind != 0 && kind != 1))
    return fail(ctx, RRTX_E_INVALID, "edges_check: bad arguments");
  if (ne == 0) return RRTX_OK;
  const size_t pb = sizeof(double) * (size_t)ne * ctx->dim;
  int rc = stage_in(ctx, ctx->ws_q, p0, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, p1, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure((size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)ne));
  rc = rrtx_edges_check_dev(ctx, kind, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, robot_radius,
                            obstacle_or_minus1, -1, -1, ctx->ws_out_u8a.as<uint8_t>(),
                            first_hit ? ctx->ws_out_i32.as<int32_t>() : nullptr);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(hit, ctx->ws_out_u8a.p, (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (first_hit)
    RRTX_HIP(ctx, hipMemcpyAsync(first_hit, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_edges_check_idx(rrtx_ctx *ctx, const int32_t *start_idx, const int32_t *end_idx, int64_t ne,
                         double robot_radius, int obstacle_or_minus1, const uint8_t *obstacle_mask, uint8_t *hit,
                         int32_t *first_hit) {
  CHECK_CTX(ctx);
  if (ne < 0 || (ne > 0 && (!start_idx || !end_idx || !hit)))
    return fail(ctx, RRTX_E_INVALID, "edges_check_idx: bad arguments");
  if (ne == 0) return RRTX_OK;
  for (int64_t i = 0; i < ne; ++i)
    if (start_idx[i] < 0 || start_idx[i] >= ctx->n_nodes || end_idx[i] < 0 || end_idx[i] >= ctx->n_nodes)
      return fail(ctx, RRTX_E_INVALID, "edges_check_idx: edge %lld references a node outside [0, %lld)",
                  (long long)i, (long long)ctx->n_nodes);
  int rc = stage_in(ctx, ctx->ws_i32a, start_idx, sizeof(int32_t) * (size_t)ne);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_i32b, end_idx, sizeof(int32_t) * (size_t)ne);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure((size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)ne));
  rc = launch_edges_spheres(ctx, nullptr, nullptr, ne, robot_radius, obstacle_or_minus1, -1, -1,
                            ctx->ws_out_u8a.as<uint8_t>(), first_hit ? ctx->ws_out_i32.as<int32_t>() : nullptr,
                            ctx->ws_i32a.as<int32_t>(), ctx->ws_i32b.as<int32_t>(), obstacle_mask);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(hit, ctx->ws_out_u8a.p, (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (first_hit)
    RRTX_HIP(ctx, hipMemcpyAsync(first_hit, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

// ---- obstacle sweeps over a device mirror of the planner's edges -----------------------
int64_t rrtx_graph_edges_count(rrtx_ctx *ctx) { return ctx ? ctx->ge_n : 0; }

int rrtx_graph_edges_clear(rrtx_ctx *ctx) {
  CHECK_CTX(ctx);
  ctx->ge_n = 0;
  graph_cost_forget(ctx);
  return RRTX_OK;
}

int rrtx_graph_edges_append(rrtx_ctx *ctx, const int32_t *start_idx, const int32_t *end_idx, int64_t n,
                            int64_t *first_id) {
  CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && (!start_idx || !end_idx))) return fail(ctx, RRTX_E_INVALID, "graph_edges_append: bad arguments");
  if (first_id) *first_id = ctx->ge_n;
  if (n == 0) return RRTX_OK;
  for (int64_t i = 0; i < n; ++i)
    if (start_idx[i] < 0 || start_idx[i] >= ctx->n_nodes || end_idx[i] < 0 || end_idx[i] >= ctx->n_nodes)
      return fail(ctx, RRTX_E_INVALID, "graph_edges_append: edge %lld references a node outside [0, %lld)",
                  (long long)i, (long long)ctx->n_nodes);
  if (ctx->ge_n + n > 0x7fffffffll) return fail(ctx, RRTX_E_CAPACITY, "edge ids are int32");
  if (ctx->ge_n + n > ctx->ge_cap) {
    int64_t nc = ctx->ge_cap > 0 ? ctx->ge_cap : 4096;
    while (nc < ctx->ge_n + n) nc *= 2;
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = regrow(ctx, ctx->ge_start, nc, ctx->ge_n))) return rc;
    if ((rc = regrow(ctx, ctx->ge_end, nc, ctx->ge_n))) return rc;
    if ((rc = regrow(ctx, ctx->ge_dist, nc, ctx->ge_n))) return rc;
    if ((rc = regrow(ctx, ctx->ge_dirty, nc, ctx->ge_n))) return rc;
    ctx->ge_cap = nc;
  }
  RRTX_HIP(ctx, hipMemcpyAsync(ctx->ge_start + ctx->ge_n, start_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(ctx->ge_end + ctx->ge_n, end_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  // edge.dist defaults to the SimpleEdge cost of the edge (rrtx_graph_edges_set_dist overrides it)
  int rc2 = launch_graph_edge_dist(ctx, ctx->ge_n, n);
  if (rc2) return rc2;
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->ge_n += n;
  return RRTX_OK;
}

int rrtx_graph_edges_set_dist(rrtx_ctx *ctx, int64_t first_id, const double *dist, int64_t n) {
  CHECK_CTX(ctx);
  if (n < 0 || first_id < 0 || first_id + n > ctx->ge_n || (n > 0 && !dist))
    return fail(ctx, RRTX_E_INVALID, "graph_edges_set_dist: edges [%lld, %lld) of %lld", (long long)first_id,
                (long long)(first_id + n), (long long)ctx->ge_n);
  if (n == 0) return RRTX_OK;
  RRTX_HIP(ctx, hipMemcpyAsync(ctx->ge_dist + first_id, dist, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_graph_touch(ctx, first_id, n);
  if (rc) return rc;
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_graph_edges_block(rrtx_ctx *ctx, const int32_t *edge_ids, int64_t n) {
  CHECK_CTX(ctx);
  if (n < 0 || (n > 0 && !edge_ids)) return fail(ctx, RRTX_E_INVALID, "graph_edges_block: bad arguments");
  for (int64_t i = 0; i < n; ++i)
    if (edge_ids[i] < 0 || edge_ids[i] >= ctx->ge_n) return fail(ctx, RRTX_E_INVALID, "graph_edges_block: edge id %d out of range", edge_ids[i]);
  return launch_graph_block(ctx, edge_ids, n);
}

namespace {
int graph_cost_host(rrtx_ctx *ctx, const char *fn, int root_idx, bool update, double *lmc, int32_t *parent_edge, int32_t *passes) {
  if (!lmc) return fail(ctx, RRTX_E_INVALID, "%s: lmc is NULL", fn);
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "%s on an empty tree", fn);
  if (root_idx < 0 || root_idx >= ctx->n_nodes) return fail(ctx, RRTX_E_INVALID, "%s: root %d out of range", fn, root_idx);
  const size_t n = (size_t)ctx->n_nodes;
  RRTX_HIP(ctx, ctx->ws_out_f64.ensure(sizeof(double) * n));
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * n));
  int np = 0;
  int rc = launch_graph_cost(ctx, root_idx, update, ctx->ws_out_f64.as<double>(), parent_edge ? ctx->ws_out_i32.as<int32_t>() : nullptr, &np);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(lmc, ctx->ws_out_f64.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (parent_edge) RRTX_HIP(ctx, hipMemcpyAsync(parent_edge, ctx->ws_out_i32.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (passes) *passes = np;
  return RRTX_OK;
}

int graph_cost_dev(rrtx_ctx *ctx, const char *fn, int root_idx, bool update, double *lmc_dev, int32_t *parent_edge_dev) {
  if (!lmc_dev) return fail(ctx, RRTX_E_INVALID, "%s: lmc is NULL", fn);
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "%s on an empty tree", fn);
  if (root_idx < 0 || root_idx >= ctx->n_nodes) return fail(ctx, RRTX_E_INVALID, "%s: root %d out of range", fn, root_idx);
  return launch_graph_cost(ctx, root_idx, update, lmc_dev, parent_edge_dev, nullptr);
}
}  // namespace

int rrtx_graph_cost_to_root(rrtx_ctx *ctx, int root_idx, double *lmc, int32_t *parent_edge, int32_t *passes) {
  CHECK_CTX(ctx);
  return graph_cost_host(ctx, "graph_cost_to_root", root_idx, false, lmc, parent_edge, passes);
}

int rrtx_graph_cost_update(rrtx_ctx *ctx, int root_idx, double *lmc, int32_t *parent_edge, int32_t *passes) {
  CHECK_CTX(ctx);
  return graph_cost_host(ctx, "graph_cost_update", root_idx, true, lmc, parent_edge, passes);
}

int rrtx_graph_cost_to_root_dev(rrtx_ctx *ctx, int root_idx, double *lmc_dev, int32_t *parent_edge_dev) {
  CHECK_CTX(ctx);
  return graph_cost_dev(ctx, "graph_cost_to_root", root_idx, false, lmc_dev, parent_edge_dev);
}

int rrtx_graph_cost_update_dev(rrtx_ctx *ctx, int root_idx, double *lmc_dev, int32_t *parent_edge_dev) {
  CHECK_CTX(ctx);
  return graph_cost_dev(ctx, "graph_cost_update", root_idx, true, lmc_dev, parent_edge_dev);
}

// ---- obstacle sweeps of the polygon / Dubins space (R/DRRT.jl:3048-3290) ----------------------------------------
namespace {
struct SweepQ { double x, y, z, w, thr_lt, thr_root; };    // layout of kernels_sweep.hip's SweepQuery

// one query of findPointsInConflictWithObstacle: the point itself (the root is taken with <=) and its ghosts in the
// wrapped dimensions, in getNextGhostPoint's order and under its skip rule (R/ghostPoint.jl:60-111)
void push_query_with_ghosts(rrtx_ctx *ctx, const double q[4], double range, std::vector<SweepQ> &out) {
  const double tlt = rrtx::thr_first_ge(range), tgt = rrtx::thr_first_gt(range);
  SweepQ o = {q[0], q[1], q[2], q[3], tlt, ctx->opt_root_rule ? tgt : tlt};
  out.push_back(o);
  const int nw = ctx->n_wraps;
  for (int k = 1; k < (1 << nw); ++k) {
    double g[4] = {q[0], q[1], q[2], q[3]}, c[4] = {q[0], q[1], q[2], q[3]};
    for (int w = 0; w < nw; ++w) {
      if (!((k >> (nw - 1 - w)) & 1)) continue;
      const int d = ctx->wrap_dim[w];
      double dim_val = q[d], dim_closest = 0.0;
      if (q[d] < ctx->wrap_period[w] / 2.0) { dim_val += ctx->wrap_period[w]; dim_closest += ctx->wrap_period[w]; }
      else dim_val -= ctx->wrap_period[w];
      g[d] = dim_val; c[d] = dim_closest;
    }
    double s = 0.0;                          // KDdist(closestUnwrappedPoint, ghost)^2, left fold over the d coordinates
    for (int d = 0; d < ctx->dim; ++d) { const double t = c[d] - g[d]; s = (d == 0) ? t * t : s + t * t; }
    if (s >= tgt) continue;                  // > bestDist: this ghost is not searched (:104)
    SweepQ gq = {g[0], g[1], g[2], g[3], tlt, tlt};
    out.push_back(gq);
  }
}
}  // namespace

int rrtx_obstacle_sweep_polygon(rrtx_ctx *ctx, int obstacle, double robot_radius, double delta, double r_min, int mode,
                                int32_t *edge_ids, int64_t cap, int64_t *needed) {
  CHECK_CTX(ctx);
  const int m = (int)ctx->poly_active.size();
  if (obstacle < 0 || obstacle >= m) return fail(ctx, RRTX_E_INVALID, "obstacle_sweep_polygon: obstacle %d out of range (%d polygons)", obstacle, m);
  if (cap < 0 || (cap > 0 && !edge_ids) || (mode != 0 && mode != 1)) return fail(ctx, RRTX_E_INVALID, "obstacle_sweep_polygon: bad arguments");
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "obstacle_sweep_polygon on an empty tree");
  const bool dubins = ctx->dim == 4, has_time = ctx->opt_space_has_time;
  const int kind = ctx->poly_kind[obstacle];
  const double cx = ctx->poly_cr[3 * (size_t)obstacle], cy = ctx->poly_cr[3 * (size_t)obstacle + 1],
               rad = ctx->poly_cr[3 * (size_t)obstacle + 2];
  // ---- findPointsInConflictWithObstacle (R/DRRT.jl:3048-3125) ----
  std::vector<SweepQ> qs;
  if (kind >= 1 && kind <= 5) {
    if (has_time) return fail(ctx, RRTX_E_STATE, "this type of obstacle not coded for this type of space (a static obstacle in a "
                              "space with time, R/DRRT.jl:3067)");
    // Euclidean space: range robotRadius + delta + radius around ob.position (a dim = 3 tree is the legacy 2-D space
    // embedded at z = 0); Dubins space: [x y 0.0 pi], range + pi
    const double q[4] = {cx, cy, 0.0, dubins ? 3.141592653589793 : 0.0};
    double range = robot_radius + delta + rad;
    if (dubins) range = range + 3.141592653589793;
    push_query_with_ghosts(ctx, q, range, qs);
  } else if (kind == 6 || kind == 7) {
    const int p0 = ctx->poly_path_off[obstacle], np = ctx->poly_path_off[obstacle + 1] - p0;
    if (np < 1) return fail(ctx, RRTX_E_STATE, "moving obstacle %d has no path (rrtx_polygon_paths_set)", obstacle);
    const double base = robot_radius + delta + rad;
    for (int i = 0; i < np; ++i) {
      const int j = (np == 1) ? 0 : i + 1;
      const double *pi = &ctx->poly_path[3 * (size_t)(p0 + i)], *pj = &ctx->poly_path[3 * (size_t)(p0 + j)];
      const double q[4] = {cx + (pi[0] + pj[0]) / 2.0, cy + (pi[1] + pj[1]) / 2.0, 0.0 + (pi[2] + pj[2]) / 2.0,
                           dubins ? 3.141592653589793 : 0.0};
      const double dx = pi[0] - pj[0], dy = pi[1] - pj[1], dt = pi[2] - pj[2];
      double range = base + std::sqrt((dx * dx + dy * dy) + dt * dt) / 2.0;
      if (dubins) range += 3.141592653589793;
      push_query_with_ghosts(ctx, q, range, qs);
      if (j == np - 1) break;
    }
  } else {
    return fail(ctx, RRTX_E_STATE, "this case not coded yet (obstacle kind %d, R/DRRT.jl:3121)", kind);
  }
  if (needed) *needed = 0;
  ctx->last_sweep_candidates = 0;
  const long long ne = ctx->ge_n;
  if (ne == 0) return RRTX_OK;
  int rc = sync_polygons(ctx);
  if (rc) return rc;
  // the obstacle in the packed (in-use only) table: an unused obstacle collides with nothing (R/DRRT.jl:1525)
  const std::vector<int32_t> pos = active_positions(ctx->poly_active);
  int pb, pe;
  packed_range(pos, obstacle, obstacle + 1, pb, pe);
  if (pe <= pb) return RRTX_OK;
  rc = launch_sweep_mark_multi(ctx, qs.data(), (int)qs.size());
  if (rc) return rc;
  // ---- the out-edges (and parent edges) of those nodes; removeObstacle looks at blocked ones only ----
  RRTX_HIP(ctx, ctx->ws_i32a.ensure(sizeof(int32_t) * (size_t)ne));
  long long *total_dev = nullptr, n_c = 0;
  span_begin(ctx, KF_EDGES);
  rc = launch_sweep_select(ctx, mode == 1 ? 1 : 0, ctx->ws_i32a.as<int32_t>(), ne, &total_dev);
  span_end(ctx);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(&n_c, total_dev, sizeof(n_c), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->last_sweep_candidates = n_c;
  if (n_c == 0) return RRTX_OK;
  // ---- explicitEdgeCheck(S, edge, ob) of every candidate; removeObstacle: and against every OTHER obstacle in use ----
  const int n_pass = mode == 1 ? 3 : 1;
  RRTX_HIP(ctx, ctx->ws_out_u8b.ensure((size_t)n_c * 3));
  uint8_t *hit = ctx->ws_out_u8b.as<uint8_t>(), *o1 = hit + n_c, *o2 = o1 + n_c;
  const int32_t *cand = ctx->ws_i32a.as<int32_t>();
  const int na = (int)pos.size();
  const int rb[3] = {pb, 0, pe}, re[3] = {pe, pb, na};       // ob | the packed obstacles before it | after it
  if (dubins) {
    for (int k = 0; k < n_pass; ++k) {
      rc = launch_dubins_edges_idx(ctx, cand, n_c, r_min, robot_radius, rb[k], re[k], hit + (size_t)k * n_c);
      if (rc) return rc;
    }
  } else {
    RRTX_HIP(ctx, ctx->ws_q.ensure(sizeof(double) * 3 * (size_t)n_c));
    RRTX_HIP(ctx, ctx->ws_q2.ensure(sizeof(double) * 3 * (size_t)n_c));
    rc = launch_sweep_gather(ctx, cand, n_c, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>());
    if (rc) return rc;
    const int lb[3] = {obstacle, 0, obstacle + 1}, le[3] = {obstacle + 1, obstacle, m};   // the same three ranges in list positions
    for (int k = 0; k < n_pass; ++k) {
      rc = launch_edges_polygons(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), n_c, robot_radius, -1, lb[k], le[k],
                                 hit + (size_t)k * n_c, nullptr);
      if (rc) return rc;
    }
  }
  const int64_t dcap = cap > 0 ? cap : 1;
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)dcap));
  span_begin(ctx, KF_EDGES);
  rc = launch_sweep_finish(ctx, cand, n_c, hit, mode == 1 ? o1 : nullptr, mode == 1 ? o2 : nullptr,
                           ctx->ws_out_i32.as<int32_t>(), cap, &total_dev);
  span_end(ctx);
  if (rc) return rc;
  long long total = 0;
  RRTX_HIP(ctx, hipMemcpyAsync(&total, total_dev, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (needed) *needed = total;
  if (total > cap) return fail(ctx, RRTX_E_CAPACITY, "obstacle_sweep_polygon: %lld edges, capacity %lld", total, (long long)cap);
  if (total > 0) {
    RRTX_HIP(ctx, hipMemcpyAsync(edge_ids, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return RRTX_OK;
}

int rrtx_dubins_edges_check_obstacle(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                                     double robot_radius, int obstacle, uint8_t *hit) {
  CHECK_CTX(ctx);
  const int m = (int)ctx->poly_active.size();
  if (ne < 0 || (ne > 0 && (!s || !g || !hit))) return fail(ctx, RRTX_E_INVALID, "dubins_edges_check_obstacle: bad arguments");
  if (obstacle < 0 || obstacle >= m) return fail(ctx, RRTX_E_INVALID, "dubins_edges_check_obstacle: obstacle %d out of range (%d polygons)", obstacle, m);
  if (ne == 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  int pb, pe;
  packed_range(active_positions(ctx->poly_active), obstacle, obstacle + 1, pb, pe);
  if (pe <= pb) { std::memset(hit, 0, (size_t)ne); return RRTX_OK; }          // not in use: collides with nothing
  const size_t pbytes = sizeof(double) * (size_t)ne * 4;
  int rc = stage_in(ctx, ctx->ws_q, s, pbytes);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, g, pbytes);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_u8b.ensure((size_t)ne));
  rc = launch_dubins_edges_check(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min, robot_radius, nullptr, nullptr,
                                 ctx->ws_out_u8b.as<uint8_t>(), nullptr, pb, pe);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(hit, ctx->ws_out_u8b.p, (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_obstacle_sweep(rrtx_ctx *ctx, int obstacle, double search_range, double robot_radius, int32_t *edge_ids,
                        int64_t cap, int64_t *needed) {
  CHECK_CTX(ctx);
  const int m = (int)ctx->sph_active.size();
  if (obstacle < 0 || obstacle >= m) return fail(ctx, RRTX_E_INVALID, "obstacle_sweep: obstacle %d out of range (%d spheres)", obstacle, m);
  if (cap < 0 || (cap > 0 && !edge_ids)) return fail(ctx, RRTX_E_INVALID, "obstacle_sweep: bad arguments");
  if (ctx->n_nodes <= 0) return fail(ctx, RRTX_E_STATE, "obstacle_sweep on an empty tree");
  if (ctx->dim != 3) return fail(ctx, RRTX_E_STATE, "obstacle_sweep is the SimpleEdge (dim=3) path");
  const double *c = &ctx->sph[4 * (size_t)obstacle];
  SphRec ob;
  ob.cx = c[0]; ob.cy = c[1]; ob.cz = c[2];
  ob.thr = thr_first_gt(robot_radius + c[3]);
  const int64_t dcap = cap > 0 ? cap : 1;
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)dcap));
  long long *total_dev = nullptr;
  int rc = launch_obstacle_sweep(ctx, c, thr_first_ge(search_range), thr_first_gt(search_range), ob,
                                 ctx->sph_active[obstacle] ? 1 : 0, ctx->ws_out_i32.as<int32_t>(), cap, &total_dev);
  if (rc) return rc;
  long long total = 0;
  RRTX_HIP(ctx, hipMemcpyAsync(&total, total_dev, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (needed) *needed = total;
  if (total > cap) return fail(ctx, RRTX_E_CAPACITY, "obstacle_sweep: %lld colliding edges, capacity %lld", total, (long long)cap);
  if (total > 0) {
    RRTX_HIP(ctx, hipMemcpyAsync(edge_ids, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return RRTX_OK;
}

int rrtx_points_check_dev(rrtx_ctx *ctx, int kind, const double *p, int64_t np, double robot_radius, int quick,
                          uint8_t *unsafe, double *clearance) {
  CHECK_CTX(ctx);
  if (np < 0 || (np > 0 && (!p || !unsafe)) || (kind != 0 && kind != 1))
    return fail(ctx, RRTX_E_INVALID, "points_check: bad arguments");
  if (kind == 0) return launch_points_spheres(ctx, p, np, robot_radius, quick, unsafe, clearance);
  return launch_points_polygons(ctx, p, np, robot_radius, unsafe, clearance);
}

int rrtx_points_check(rrtx_ctx *ctx, int kind, const double *p, int64_t np, double robot_radius, int quick,
                      uint8_t *unsafe, double *clearance) {
  CHECK_CTX(ctx);
  if (np < 0 || (np > 0 && (!p || !unsafe)) || (kind != 0 && kind != 1))
    return fail(ctx, RRTX_E_INVALID, "points_check: bad arguments");
  if (np == 0) return RRTX_OK;
  int rc = stage_in(ctx, ctx->ws_q, p, sizeof(double) * (size_t)np * ctx->dim);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure((size_t)np));
  RRTX_HIP(ctx, ctx->ws_out_f64.ensure(sizeof(double) * (size_t)np));
  // (no certificate wanted: the polygon check then only walks the obstacles near each point)
  rc = rrtx_points_check_dev(ctx, kind, ctx->ws_q.as<double>(), np, robot_radius, quick,
                             ctx->ws_out_u8a.as<uint8_t>(), clearance ? ctx->ws_out_f64.as<double>() : nullptr);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(unsafe, ctx->ws_out_u8a.p, (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
  if (clearance)
    RRTX_HIP(ctx, hipMemcpyAsync(clearance, ctx->ws_out_f64.p, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

// ---- steering -------------------------------------------------------------------------------
int rrtx_simple_steer(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double *dist, double *wdist) {
  CHECK_CTX(ctx);
  if (ne < 0 || (ne > 0 && (!s || !g))) return fail(ctx, RRTX_E_INVALID, "simple_steer: bad arguments");
  if (ne == 0) return RRTX_OK;
  const size_t pb = sizeof(double) * (size_t)ne * ctx->dim;
  int rc = stage_in(ctx, ctx->ws_q, s, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, g, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_f64.ensure(sizeof(double) * (size_t)ne));
  rc = launch_simple_steer(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, ctx->ws_out_dist.as<double>(),
                           ctx->ws_out_f64.as<double>());
  if (rc) return rc;
  if (dist) RRTX_HIP(ctx, hipMemcpyAsync(dist, ctx->ws_out_dist.p, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (wdist) RRTX_HIP(ctx, hipMemcpyAsync(wdist, ctx->ws_out_f64.p, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

static int dubins_common(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                         double robot_radius, bool check, double *cost, uint8_t *word, uint8_t *hit,
                         int32_t *traj_len) {
  if (ne < 0 || (ne > 0 && (!s || !g || !cost)) || (check && ne > 0 && !hit))
    return fail(ctx, RRTX_E_INVALID, "dubins: bad arguments");
  if (ne == 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  const size_t pb = sizeof(double) * (size_t)ne * 4;
  int rc = stage_in(ctx, ctx->ws_q, s, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, g, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure(3 * (size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_u8b.ensure((size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)ne));
  if (check)
    rc = launch_dubins_edges_check(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min, robot_radius,
                                   ctx->ws_out_dist.as<double>(), ctx->ws_out_u8a.as<uint8_t>(),
                                   ctx->ws_out_u8b.as<uint8_t>(), ctx->ws_out_i32.as<int32_t>());
  else
    rc = launch_dubins_steer(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min,
                             ctx->ws_out_dist.as<double>(), ctx->ws_out_u8a.as<uint8_t>());
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(cost, ctx->ws_out_dist.p, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (word) RRTX_HIP(ctx, hipMemcpyAsync(word, ctx->ws_out_u8a.p, 3 * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (check) {
    RRTX_HIP(ctx, hipMemcpyAsync(hit, ctx->ws_out_u8b.p, (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
    if (traj_len)
      RRTX_HIP(ctx, hipMemcpyAsync(traj_len, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  }
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_dubins_steer(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min, double *cost,
                      uint8_t *word) {
  CHECK_CTX(ctx);
  return dubins_common(ctx, s, g, ne, r_min, 0.0, false, cost, word, nullptr, nullptr);
}

int rrtx_set_dubins_velocity(rrtx_ctx *ctx, double v_min, double v_max) {
  CHECK_CTX(ctx);
  ctx->dubins_vmin = v_min;
  ctx->dubins_vmax = v_max;
  return RRTX_OK;
}

int rrtx_dubins_steer_full(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min, double *dist,
                           double *wdist, double *velocity, uint8_t *word, uint8_t *valid_move) {
  CHECK_CTX(ctx);
  if (ne < 0 || (ne > 0 && (!s || !g))) return fail(ctx, RRTX_E_INVALID, "dubins_steer_full: bad arguments");
  if (ne == 0) return RRTX_OK;
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  const size_t pb = sizeof(double) * (size_t)ne * 4;
  int rc = stage_in(ctx, ctx->ws_q, s, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, g, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * 3 * (size_t)ne));
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure(4 * (size_t)ne));
  double *d_dist = ctx->ws_out_dist.as<double>(), *d_w = d_dist + ne, *d_v = d_w + ne;
  uint8_t *d_word = ctx->ws_out_u8a.as<uint8_t>(), *d_valid = d_word + 3 * (size_t)ne;
  rc = launch_dubins_steer(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min, d_dist, d_word, d_w, d_v, d_valid);
  if (rc) return rc;
  if (dist) RRTX_HIP(ctx, hipMemcpyAsync(dist, d_dist, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (wdist) RRTX_HIP(ctx, hipMemcpyAsync(wdist, d_w, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (velocity) RRTX_HIP(ctx, hipMemcpyAsync(velocity, d_v, sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (word) RRTX_HIP(ctx, hipMemcpyAsync(word, d_word, 3 * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  if (valid_move) RRTX_HIP(ctx, hipMemcpyAsync(valid_move, d_valid, (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_dubins_edges_check(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                            double robot_radius, double *cost, uint8_t *word, uint8_t *hit, int32_t *traj_len) {
  CHECK_CTX(ctx);
  return dubins_common(ctx, s, g, ne, r_min, robot_radius, true, cost, word, hit, traj_len);
}

int rrtx_dubins_trajectory(rrtx_ctx *ctx, const double *s, const double *g, int64_t ne, double r_min,
                           int64_t *traj_off, double *traj_xy, int cols_in, int64_t cap_rows, int64_t *needed_rows) {
  CHECK_CTX(ctx);
  if (ne < 0 || cap_rows < 0 || (ne > 0 && (!s || !g || !traj_off)) || (cap_rows > 0 && !traj_xy))
    return fail(ctx, RRTX_E_INVALID, "dubins_trajectory: bad arguments");
  if (cols_in != (ctx->opt_space_has_time ? 3 : 2))
    return fail(ctx, RRTX_E_INVALID, "dubins_trajectory: rows are %d doubles wide in this context (RRTX_OPT_SPACE_HAS_TIME = %d), "
                "the caller's buffer was sized for %d", ctx->opt_space_has_time ? 3 : 2, ctx->opt_space_has_time ? 1 : 0, cols_in);
  if (ne == 0) { if (needed_rows) *needed_rows = 0; if (traj_off) traj_off[0] = 0; return RRTX_OK; }
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "Dubins steering needs a dim=4 [x y t theta] context");
  const size_t pb = sizeof(double) * (size_t)ne * 4;
  int rc = stage_in(ctx, ctx->ws_q, s, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, g, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)ne));
  // pass 1: rows per edge
  rc = launch_dubins_trajectory(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min, nullptr, nullptr, 0,
                                ctx->ws_out_i32.as<int32_t>());
  if (rc) return rc;
  std::vector<int32_t> len((size_t)ne);
  RRTX_HIP(ctx, hipMemcpyAsync(len.data(), ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)ne, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  traj_off[0] = 0;
  for (int64_t i = 0; i < ne; ++i) traj_off[i + 1] = traj_off[i] + len[(size_t)i];
  const int64_t total = traj_off[ne];
  if (needed_rows) *needed_rows = total;
  if (total > cap_rows) return fail(ctx, RRTX_E_CAPACITY, "dubins_trajectory: %lld rows, capacity %lld", (long long)total, (long long)cap_rows);
  if (total == 0) return RRTX_OK;
  // pass 2: write the polylines
  rc = stage_in(ctx, ctx->ws_out_off, traj_off, sizeof(int64_t) * ((size_t)ne + 1));
  if (rc) return rc;
  const size_t cols = ctx->opt_space_has_time ? 3 : 2;      // (x, y) or, in a space with time, (x, y, t)
  RRTX_HIP(ctx, ctx->ws_out_f64.ensure(sizeof(double) * cols * (size_t)total));
  rc = launch_dubins_trajectory(ctx, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), ne, r_min,
                                ctx->ws_out_off.as<int64_t>(), ctx->ws_out_f64.as<double>(), total, nullptr);
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(traj_xy, ctx->ws_out_f64.p, sizeof(double) * cols * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

int rrtx_detmath_eval(rrtx_ctx *ctx, int op, const double *x, const double *y, int64_t n, double *out) {
  CHECK_CTX(ctx);
  if (n < 0 || op < 0 || op > 3 || (n > 0 && (!x || !out)) || (op == 2 && n > 0 && !y))
    return fail(ctx, RRTX_E_INVALID, "detmath_eval: bad arguments");
  if (n == 0) return RRTX_OK;
  const size_t pb = sizeof(double) * (size_t)n;
  int rc = stage_in(ctx, ctx->ws_q, x, pb);
  if (rc) return rc;
  rc = stage_in(ctx, ctx->ws_q2, y ? y : x, pb);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(pb));
  rc = launch_detmath_eval(ctx, op, ctx->ws_q.as<double>(), ctx->ws_q2.as<double>(), n, ctx->ws_out_dist.as<double>());
  if (rc) return rc;
  RRTX_HIP(ctx, hipMemcpyAsync(out, ctx->ws_out_dist.p, pb, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RRTX_OK;
}

// ---- fused extend() preamble ------------------------------------------------------------------
int rrtx_extend_candidates_dev(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                               int64_t *offsets, int32_t *idx, double *cost, uint8_t *hit_out, uint8_t *hit_in,
                               int64_t cap, int64_t *needed_dev, int32_t *nearest_idx, double *nearest_dist,
                               uint8_t *sample_unsafe) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !offsets)) || (cap > 0 && (!idx || !cost || !hit_out || !hit_in)))
    return fail(ctx, RRTX_E_INVALID, "extend_candidates: bad arguments");
  if (ctx->dim != 3) return fail(ctx, RRTX_E_STATE, "extend_candidates is the SimpleEdge (dim=3) path");
  if (ctx->n_wraps != 0)
    return fail(ctx, RRTX_E_STATE, "extend_candidates reads kdFindNearest off the range lists, which is only valid "
                                   "without wrapped dimensions (use rrtx_extend_candidates_dubins / the separate calls)");
  if (nq == 0) return RRTX_OK;
  const bool want_nearest = nearest_idx && nearest_dist;
  // nearest falls out of the radius lists (kdFindNearest's answer lies inside the ball whenever the
  // ball is non-empty); a sample with an empty ball is answered by an expanding search on the device
  // (kernels_finish.hip), so no -1 reaches the caller.
  if (ctx->opt_extend_polygons) {
    RRTX_HIP(ctx, ctx->ws_owner.ensure(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1)));
    int rc = launch_nn_radius(ctx, q, nullptr, r, nq, offsets, idx, cost, cap, needed_dev,
                              ctx->ws_owner.as<int32_t>(), want_nearest ? nearest_idx : nullptr,
                              want_nearest ? nearest_dist : nullptr);
    if (rc) return rc;
    return launch_candidate_edges_polygons(ctx, q, nq, offsets, idx, ctx->ws_owner.as<int32_t>(), cap, robot_radius,
                                           hit_out, hit_in, sample_unsafe, (r >= 0.0) ? r : -1.0);
  }
  // sphere list: in the culled search the sample pass and both directed edges of every neighbour are
  // decided where the neighbour is found (kernels_nn.hip, TileEmit) and the finish kernel hands out
  // the flags; the other search paths (small trees, culling off) run the two kernels of their own
  int rc = sync_spheres(ctx, robot_radius);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_owner.ensure(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1)));
  ExtendFuse ef;
  ef.r = (r >= 0.0) ? r : -1.0;
  ef.hit_out = hit_out; ef.hit_in = hit_in; ef.sample_unsafe = sample_unsafe;
  ef.fused = false;
  rc = launch_nn_radius(ctx, q, nullptr, r, nq, offsets, idx, cost, cap, needed_dev, ctx->ws_owner.as<int32_t>(),
                        want_nearest ? nearest_idx : nullptr, want_nearest ? nearest_dist : nullptr, &ef);
  if (rc || ef.fused) return rc;
  return launch_candidate_edges(ctx, q, nq, offsets, idx, ctx->ws_owner.as<int32_t>(), cap, robot_radius, hit_out,
                                hit_in, (r >= 0.0) ? r : -1.0, sample_unsafe);
}

int rrtx_extend_candidates(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius, int64_t *offsets,
                           int32_t *idx, double *cost, uint8_t *hit_out, uint8_t *hit_in, int64_t cap,
                           int64_t *needed, int32_t *nearest_idx, double *nearest_dist, uint8_t *sample_unsafe) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !offsets)) || (cap > 0 && (!idx || !cost || !hit_out || !hit_in)))
    return fail(ctx, RRTX_E_INVALID, "extend_candidates: bad arguments");
  if (nq == 0) { if (needed) *needed = 0; if (offsets) offsets[0] = 0; return RRTX_OK; }
  // ONE small device block for everything that is per sample -- [offsets (nq + 1) | count | nearest_dist (nq) |
  // nearest_idx (nq) | sample_unsafe (nq)] -- so that it leaves in one transfer; the per-entry arrays follow once the
  // count is known.  Small transfers go through the context's pinned arena, large ones straight to the caller.
  const int64_t dcap = cap > 0 ? cap : 1;
  const size_t o_cnt = sizeof(int64_t) * ((size_t)nq + 1), o_nd = o_cnt + sizeof(int64_t), o_ni = o_nd + sizeof(double) * (size_t)nq,
               o_un = o_ni + sizeof(int32_t) * (size_t)nq, small_bytes = o_un + (size_t)nq;
  int rc = arena_begin(ctx, small_bytes + sizeof(double) * (size_t)nq * ctx->dim + 256);
  if (rc) return rc;
  rc = stage_in_small(ctx, ctx->ws_q, q, sizeof(double) * (size_t)nq * ctx->dim);
  if (rc) return rc;
  RRTX_HIP(ctx, ctx->ws_out_off.ensure(small_bytes + 64));
  RRTX_HIP(ctx, ctx->ws_out_idx.ensure(sizeof(int32_t) * (size_t)dcap));
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * (size_t)dcap));
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure((size_t)dcap));
  RRTX_HIP(ctx, ctx->ws_out_u8b.ensure((size_t)dcap));
  char *blk = ctx->ws_out_off.as<char>();
  int64_t *off_dev = reinterpret_cast<int64_t *>(blk);
  int64_t *needed_dev = off_dev + nq + 1;
  double *nd_dev = reinterpret_cast<double *>(blk + o_nd);
  int32_t *ni_dev = reinterpret_cast<int32_t *>(blk + o_ni);
  uint8_t *unsafe_dev = reinterpret_cast<uint8_t *>(blk + o_un);
  const bool want_nearest = nearest_idx && nearest_dist;
  rc = rrtx_extend_candidates_dev(ctx, ctx->ws_q.as<double>(), nq, r, robot_radius, off_dev,
                                  ctx->ws_out_idx.as<int32_t>(), ctx->ws_out_dist.as<double>(),
                                  ctx->ws_out_u8a.as<uint8_t>(), ctx->ws_out_u8b.as<uint8_t>(), cap, needed_dev,
                                  want_nearest ? ni_dev : nullptr, want_nearest ? nd_dev : nullptr,
                                  sample_unsafe ? unsafe_dev : nullptr);
  if (rc) return rc;
  char *host_blk = arena_take(ctx, small_bytes);
  if (!host_blk) return fail(ctx, RRTX_E_NOMEM, "extend_candidates: staging arena");
  RRTX_HIP(ctx, hipMemcpyAsync(host_blk, blk, small_bytes, hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int64_t total = 0;
  std::memcpy(offsets, host_blk, o_cnt);
  std::memcpy(&total, host_blk + o_cnt, sizeof(int64_t));
  if (needed) *needed = total;
  ctx->last_neighbors = total;
  if (total > cap) return fail(ctx, RRTX_E_CAPACITY, "extend_candidates: %lld neighbours, capacity %lld", (long long)total, (long long)cap);
  if (total > 0) {
    if ((rc = d2h(ctx, cost, ctx->ws_out_dist.p, sizeof(double) * (size_t)total))) return rc;
    if ((rc = d2h(ctx, idx, ctx->ws_out_idx.p, sizeof(int32_t) * (size_t)total))) return rc;
    if ((rc = d2h(ctx, hit_out, ctx->ws_out_u8a.p, (size_t)total))) return rc;
    if ((rc = d2h(ctx, hit_in, ctx->ws_out_u8b.p, (size_t)total))) return rc;
  }
  // (the per-sample results are copied out of the arena while the per-entry transfers are in flight)
  if (sample_unsafe) std::memcpy(sample_unsafe, host_blk + o_un, (size_t)nq);
  if (want_nearest) {
    std::memcpy(nearest_dist, host_blk + o_nd, sizeof(double) * (size_t)nq);
    std::memcpy(nearest_idx, host_blk + o_ni, sizeof(int32_t) * (size_t)nq);
  }
  if (total > 0) {
    RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    arena_flush(ctx);
  }
  if (nearest_idx && nearest_dist) {
    // samples with an empty ball: resolve with the full nearest scan
    std::vector<int> miss;
    for (int i = 0; i < nq; ++i) if (nearest_idx[i] < 0) miss.push_back(i);
    if (!miss.empty()) {
      std::vector<double> mq(miss.size() * (size_t)ctx->dim);
      for (size_t k = 0; k < miss.size(); ++k)
        std::memcpy(&mq[k * ctx->dim], q + (size_t)miss[k] * ctx->dim, sizeof(double) * ctx->dim);
      std::vector<int32_t> mi(miss.size());
      std::vector<double> md(miss.size());
      rc = rrtx_nn_nearest(ctx, mq.data(), (int)miss.size(), mi.data(), md.data());
      if (rc) return rc;
      for (size_t k = 0; k < miss.size(); ++k) { nearest_idx[miss[k]] = mi[k]; nearest_dist[miss[k]] = md[k]; }
    }
  }
  return RRTX_OK;
}

int rrtx_extend_candidates_dubins_dev(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                                      double r_min, int64_t *offsets, int32_t *idx, double *key, double *cost_out,
                                      double *cost_in, uint8_t *word_out, uint8_t *word_in, uint8_t *hit_out,
                                      uint8_t *hit_in, int64_t cap, int64_t *needed_dev, int32_t *nearest_idx,
                                      double *nearest_dist, uint8_t *sample_unsafe) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !offsets)) ||
      (cap > 0 && (!idx || !key || !cost_out || !cost_in || !hit_out || !hit_in)))
    return fail(ctx, RRTX_E_INVALID, "extend_candidates_dubins: bad arguments");
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "extend_candidates_dubins needs a dim=4 [x y t theta] context");
  if (nq == 0) return RRTX_OK;
  RRTX_HIP(ctx, ctx->ws_owner.ensure(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1)));
  const bool want_nearest = nearest_idx && nearest_dist;
  // with wrapped dimensions a list key is the distance to the copy that found the node FIRST
  // (addToRangeList), not the minimum over the copies, so kdFindNearest cannot be read off the list
  const bool nearest_from_list = want_nearest && ctx->n_wraps == 0;
  int rc = launch_nn_radius(ctx, q, nullptr, r, nq, offsets, idx, key, cap, needed_dev, ctx->ws_owner.as<int32_t>(),
                            nearest_from_list ? nearest_idx : nullptr, nearest_from_list ? nearest_dist : nullptr);
  if (rc) return rc;
  rc = launch_candidate_dubins(ctx, q, nq, offsets, idx, ctx->ws_owner.as<int32_t>(), cap, r_min, robot_radius, cost_out,
                               cost_in, word_out, word_in, hit_out, hit_in);
  if (rc) return rc;
  if (sample_unsafe) {
    rc = launch_points_polygons(ctx, q, nq, robot_radius, sample_unsafe, nullptr);
    if (rc) return rc;
  }
  if (want_nearest && !nearest_from_list) {
    rc = launch_nn_nearest(ctx, q, nq, nearest_idx, nearest_dist);
    if (rc) return rc;
  }
  return RRTX_OK;
}

int rrtx_extend_candidates_dubins(rrtx_ctx *ctx, const double *q, int nq, double r, double robot_radius,
                                  double r_min, int64_t *offsets, int32_t *idx, double *key, double *cost_out,
                                  double *cost_in, uint8_t *word_out, uint8_t *word_in, uint8_t *hit_out,
                                  uint8_t *hit_in, int64_t cap, int64_t *needed, int32_t *nearest_idx,
                                  double *nearest_dist, uint8_t *sample_unsafe) {
  CHECK_CTX(ctx);
  if (nq < 0 || cap < 0 || (nq > 0 && (!q || !offsets)) ||
      (cap > 0 && (!idx || !key || !cost_out || !cost_in || !hit_out || !hit_in)))
    return fail(ctx, RRTX_E_INVALID, "extend_candidates_dubins: bad arguments");
  if (ctx->dim != 4) return fail(ctx, RRTX_E_STATE, "extend_candidates_dubins needs a dim=4 [x y t theta] context");
  if (nq == 0) { if (needed) *needed = 0; if (offsets) offsets[0] = 0; return RRTX_OK; }
  int rc = stage_in(ctx, ctx->ws_q, q, sizeof(double) * (size_t)nq * 4);
  if (rc) return rc;
  const int64_t dcap = cap > 0 ? cap : 1;
  const bool want_nearest = nearest_idx && nearest_dist;
  RRTX_HIP(ctx, ctx->ws_out_off.ensure(sizeof(int64_t) * ((size_t)nq + 2)));
  RRTX_HIP(ctx, ctx->ws_out_idx.ensure(sizeof(int32_t) * (size_t)dcap));
  RRTX_HIP(ctx, ctx->ws_out_dist.ensure(sizeof(double) * 3 * (size_t)dcap));          // key, cost_out, cost_in
  RRTX_HIP(ctx, ctx->ws_out_u8a.ensure(8 * (size_t)dcap + (size_t)nq));                // words (3+3), hits (1+1), unsafe
  RRTX_HIP(ctx, ctx->ws_out_i32.ensure(sizeof(int32_t) * (size_t)nq));
  RRTX_HIP(ctx, ctx->ws_out_f64.ensure(sizeof(double) * (size_t)nq));
  RRTX_HIP(ctx, ctx->ws_owner.ensure(sizeof(int32_t) * (size_t)dcap));
  int64_t *off_dev = ctx->ws_out_off.as<int64_t>();
  int64_t *needed_dev = off_dev + nq + 1;
  double *key_dev = ctx->ws_out_dist.as<double>(), *co_dev = key_dev + dcap, *ci_dev = co_dev + dcap;
  uint8_t *wo_dev = ctx->ws_out_u8a.as<uint8_t>(), *wi_dev = wo_dev + 3 * dcap, *ho_dev = wi_dev + 3 * dcap,
          *hi_dev = ho_dev + dcap, *unsafe_dev = hi_dev + dcap;
  rc = rrtx_extend_candidates_dubins_dev(ctx, ctx->ws_q.as<double>(), nq, r, robot_radius, r_min, off_dev,
                                         ctx->ws_out_idx.as<int32_t>(), key_dev, co_dev, ci_dev, wo_dev, wi_dev, ho_dev,
                                         hi_dev, cap, needed_dev, want_nearest ? ctx->ws_out_i32.as<int32_t>() : nullptr,
                                         want_nearest ? ctx->ws_out_f64.as<double>() : nullptr,
                                         sample_unsafe ? unsafe_dev : nullptr);
  if (rc) return rc;
  int64_t total = 0;
  RRTX_HIP(ctx, hipMemcpyAsync(offsets, off_dev, sizeof(int64_t) * ((size_t)nq + 1), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipMemcpyAsync(&total, needed_dev, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (needed) *needed = total;
  ctx->last_neighbors = total;
  if (total > cap) return fail(ctx, RRTX_E_CAPACITY, "extend_candidates_dubins: %lld neighbours, capacity %lld", (long long)total, (long long)cap);
  const size_t n = (size_t)total;
  if (n > 0) {
    RRTX_HIP(ctx, hipMemcpyAsync(idx, ctx->ws_out_idx.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(key, key_dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(cost_out, co_dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(cost_in, ci_dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (word_out) RRTX_HIP(ctx, hipMemcpyAsync(word_out, wo_dev, 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (word_in) RRTX_HIP(ctx, hipMemcpyAsync(word_in, wi_dev, 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(hit_out, ho_dev, n, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(hit_in, hi_dev, n, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (sample_unsafe) RRTX_HIP(ctx, hipMemcpyAsync(sample_unsafe, unsafe_dev, (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  if (want_nearest) {
    RRTX_HIP(ctx, hipMemcpyAsync(nearest_idx, ctx->ws_out_i32.p, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
    RRTX_HIP(ctx, hipMemcpyAsync(nearest_dist, ctx->ws_out_f64.p, sizeof(double) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  }
  RRTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (want_nearest) {
    std::vector<int> miss;
    for (int i = 0; i < nq; ++i) if (nearest_idx[i] < 0) miss.push_back(i);
    if (!miss.empty()) {
      std::vector<double> mq(miss.size() * 4);
      for (size_t k = 0; k < miss.size(); ++k) std::memcpy(&mq[k * 4], q + (size_t)miss[k] * 4, sizeof(double) * 4);
      std::vector<int32_t> mi(miss.size());
      std::vector<double> md(miss.size());
      rc = rrtx_nn_nearest(ctx, mq.data(), (int)miss.size(), mi.data(), md.data());
      if (rc) return rc;
      for (size_t k = 0; k < miss.size(); ++k) { nearest_idx[miss[k]] = mi[k]; nearest_dist[miss[k]] = md[k]; }
    }
  }
  return RRTX_OK;
}

int rrtx_pack_hits_dev(rrtx_ctx *ctx, const uint8_t *hit_out, const uint8_t *hit_in, const int64_t *n_valid_dev,
                       int64_t cap, uint64_t *words) {
  CHECK_CTX(ctx);
  if (cap < 0 || (cap > 0 && (!hit_out || !hit_in || !n_valid_dev || !words)))
    return fail(ctx, RRTX_E_INVALID, "pack_hits: bad arguments");
  return launch_pack_hits(ctx, hit_out, hit_in, n_valid_dev, cap, words);
}

}  // extern "C"
