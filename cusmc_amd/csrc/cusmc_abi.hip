// libcusmc_hip.so -- implementation of the C ABI declared in include/cusmc_hip.h.
// Host side only: contexts, the distribution objects (factor once, upload once), argument
// checking, dispatch to the gfx950 kernels under kernels/.  No CPU compute fallback exists: a
// call without a usable device fails with CUSMC_ENODEVICE / CUSMC_EHIP.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/cusmc_hip.h"
#include "hostla.h"
#include "launch.h"

#define CUSMC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// The reference's CUDA_CALL prints and Rcpp::stop()s (inst/include/support.cuh:9-14,26);
// here a failing HIP call becomes a status code + message and the glue raises the R error.
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(CUSMC_EHIP, "%s in %s at line %d", hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return CUSMC_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return CUSMC_OK;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Small host->device parameter uploads (factor fragments, shift / bias vectors, proposal
// matrices) go through a ring of pinned host slots: the caller's buffer is consumed by a memcpy
// into the slot before the call returns, the DMA runs stream-ordered from the slot, and a slot
// is reused only after the event recorded behind its last DMA has completed.  No stream sync,
// no dependence on how the runtime treats pageable sources.
struct StagingRing {
  static const int kSlots = 8;
  void *host[kSlots] = {};
  size_t cap[kSlots] = {};
  hipEvent_t done[kSlots] = {};
  bool used[kSlots] = {};
  int next = 0;
  int upload(void *dst_dev, const void *src, size_t bytes, hipStream_t stream)
  {
    const int i = next;
    next = (next + 1) % kSlots;
    if (used[i]) HIP_TRY(hipEventSynchronize(done[i]));
    if (!done[i]) HIP_TRY(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    if (bytes > cap[i]) {
      if (host[i]) (void)hipHostFree(host[i]);
      host[i] = nullptr;
      cap[i] = 0;
      const size_t want = bytes < 65536 ? 65536 : bytes;
      HIP_TRY(hipHostMalloc(&host[i], want, hipHostMallocDefault));
      cap[i] = want;
    }
    memcpy(host[i], src, bytes);
    HIP_TRY(hipMemcpyAsync(dst_dev, host[i], bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(done[i], stream));
    used[i] = true;
    return CUSMC_OK;
  }
  void release()
  {
    for (int i = 0; i < kSlots; ++i) {
      if (used[i]) (void)hipEventSynchronize(done[i]);
      if (done[i]) (void)hipEventDestroy(done[i]);
      if (host[i]) (void)hipHostFree(host[i]);
      host[i] = nullptr; cap[i] = 0; done[i] = nullptr; used[i] = false;
    }
  }
};

}  // namespace

struct cusmc_dist;

struct cusmc_ctx {
  int device = 0;
  int num_cus = 0;
  bool propagate_rows = false;  // CUSMC_PROPAGATE_ROWS at creation: round 1's one-workgroup-per-particle proposal kernel above d = 128 (A/B timing)
  hipStream_t stream = nullptr;
  DevBuf scratch[6];  // host-pointer entry points: X, out, w, a, small matrices
  StagingRing ring;  // pinned staging for small parameter uploads
  DevBuf whi;        // high words of the weight vector (resampler, large N)
  DevBuf nb4_pool;   // tail-pool counters of the assembly log-pdf kernel (kernels/logpdf_nb4_gfx950.s)
  cusmc::Nb4Pool nb4_state;
  DevBuf step_mats;  // [Q | G] of the fused filter step, re-uploaded only when they change
  std::vector<double> step_mats_host;
  // proposal-draw parameter image (packed fragments / transposed factors / diagonals + m0) of the last
  // draws() call and the host values it was built from: a filter calls with the same G and Q at every
  // time step, and packing + uploading 2 d^2 doubles per step is what a small filter's step would cost
  DevBuf draw_img;
  std::vector<double> draw_key;
  int draw_layout = 0;
  // the distributions created on this context: destroying the context orphans them (their device
  // buffers go, their handles stay destroyable), because callers' finalizers -- R's at session end,
  // Python's at interpreter exit -- run in no particular order
  std::vector<cusmc_dist *> dists;
};

struct cusmc_dist {
  cusmc_ctx *ctx = nullptr;
  int kind = CUSMC_MVN;
  int d = 0;
  float nu = 0.f;
  double logdet = 0.0, lognorm = 0.0;
  std::vector<double> mu, W;  // W = L^-1, row-major lower triangular
  std::vector<double> sigma;  // as given: what a replica on another device is created from
  std::vector<cusmc_dist *> replicas;  // slot r of a device list -> this distribution on that slot's context (multi-device host paths)
  // device images of the current (M, shift, bias) plan
  DevBuf frags, Mdev, shift, bias;
  // what is currently uploaded, to skip redundant uploads in time loops
  int plan = 0;  // 0 none, 1 centred, 2 affine
  bool plan_tri = true;
  std::vector<double> plan_F, plan_shift, plan_bias;
  std::vector<double> plan_Qt;  // affine plan, d >= 16: Q^T of -W F = Q L (hostM holds L)
  bool frags_valid = false, M_valid = false;
  int frags_kind = 0;  // which kernel family the device fragments are packed for (1 tile, 2 wide)
  std::vector<double> hostM;
};

namespace {

using cusmc::Epilogue;

int activate(cusmc_ctx *ctx)
{
  if (!ctx) return fail(CUSMC_EINVAL, "null context (or the context this handle was created on has been destroyed)");
  HIP_TRY(hipSetDevice(ctx->device));
  return CUSMC_OK;
}

Epilogue make_epilogue(const cusmc_dist *dist, int flags)
{
  Epilogue ep;
  ep.lognorm = dist->lognorm;
  const float nu_plus_d = dist->nu + (float)dist->d;  // float arithmetic, as the reference
  ep.half_nu_plus_d = 0.5 * (double)nu_plus_d;
  ep.inv_nu = dist->kind == CUSMC_MVT ? 1.0 / (double)dist->nu : 0.0;
  ep.kind = dist->kind;
  ep.out_density = (flags & CUSMC_OUT_DENSITY) ? 1 : 0;
  return ep;
}

// Install the plan z = bias + M (x - shift) on the device.  Re-uploads only what changed.
int install_plan(cusmc_dist *dist, int plan, bool tri, const std::vector<double> &M,
                 const double *F_key, const std::vector<double> &shift,
                 const std::vector<double> &bias)
{
  cusmc_ctx *ctx = dist->ctx;
  const int d = dist->d;
  const size_t dd = (size_t)d * d;
  bool same_M = dist->plan == plan && dist->plan_tri == tri;
  if (same_M) {
    if (F_key)
      same_M = dist->plan_F.size() == dd && !memcmp(dist->plan_F.data(), F_key, dd * 8);
    else
      same_M = dist->plan_F.empty();
  }
  if (!same_M) {
    dist->hostM = M;
    dist->frags_valid = false;
    dist->M_valid = false;
    dist->plan = plan;
    dist->plan_tri = tri;
    if (F_key) dist->plan_F.assign(F_key, F_key + dd); else dist->plan_F.clear();
  }
  // shift / bias: padded to a multiple of 16 entries for the MFMA kernel's LDS image
  const size_t padded = (size_t)((d + 63) / 64) * 64;  // covers every kernel's 16*NB (the wide kernel runs d = 144 as 192)
  if (dist->plan_shift != shift || dist->shift.p == nullptr) {
    std::vector<double> tmp(padded, 0.0);
    std::copy(shift.begin(), shift.end(), tmp.begin());
    if (int rc = dist->shift.reserve(padded * 8)) return rc;
    if (int rc = ctx->ring.upload(dist->shift.p, tmp.data(), padded * 8, ctx->stream)) return rc;
    dist->plan_shift = shift;
  }
  if (dist->plan_bias != bias || dist->bias.p == nullptr) {
    std::vector<double> tmp(padded, 0.0);
    std::copy(bias.begin(), bias.end(), tmp.begin());
    if (int rc = dist->bias.reserve(padded * 8)) return rc;
    if (int rc = ctx->ring.upload(dist->bias.p, tmp.data(), padded * 8, ctx->stream)) return rc;
    dist->plan_bias = bias;
  }
  return CUSMC_OK;
}

int ensure_frags(cusmc_dist *dist, int kind)
{
  if (dist->frags_valid && dist->frags_kind == kind) return CUSMC_OK;
  const int d = dist->d, nb = kind == 2 ? cusmc::mfma_wide_nb(d) : (d + 15) / 16, dp = 16 * nb;
  // (the matrix-core kernels take lower triangular factors only: plan_affine() rotates)
  if (!dist->plan_tri) return fail(CUSMC_EINVAL, "internal: dense factor on the matrix-core path");
  const size_t n = kind == 2 ? cusmc::mfma_wide_frag_doubles(nb)
                             : (size_t)cusmc::mfma_num_frags(nb, true) * 64;
  std::vector<double> frags(n, 0.0);
  // zero-pad M to 16*nb when d is not that already: the extra output rows give z = 0, the extra
  // columns only ever meet zeros (the kernels mask what they load there)
  std::vector<double> Mp;
  const double *M = dist->hostM.data();
  if (dp != d) {
    Mp.assign((size_t)dp * dp, 0.0);
    for (int i = 0; i < d; ++i) std::copy(dist->hostM.begin() + (size_t)i * d, dist->hostM.begin() + (size_t)(i + 1) * d, Mp.begin() + (size_t)i * dp);
    M = Mp.data();
  }
  if (kind == 2)
    cusmc::mfma_wide_pack_frags(M, dp, frags.data());
  else
    cusmc::mfma_pack_frags(M, dp, true, frags.data());
  if (int rc = dist->frags.reserve(n * 8)) return rc;
  if (int rc = dist->ctx->ring.upload(dist->frags.p, frags.data(), n * 8, dist->ctx->stream)) return rc;
  dist->frags_valid = true;
  dist->frags_kind = kind;
  return CUSMC_OK;
}

int ensure_M(cusmc_dist *dist)
{
  if (dist->M_valid) return CUSMC_OK;
  const size_t n = (size_t)dist->d * dist->d;
  if (int rc = dist->Mdev.reserve(n * 8)) return rc;
  if (int rc = dist->ctx->ring.upload(dist->Mdev.p, dist->hostM.data(), n * 8, dist->ctx->stream)) return rc;
  dist->M_valid = true;
  return CUSMC_OK;
}

// shift_dev / bias_dev: the plan's vectors, or (a time loop that uploaded every step's observation
// vector in one table ahead of its launches) this step's rows of that table
int run_logpdf(cusmc_dist *dist, const double *X_dev, int64_t N, int64_t ldx, int flags,
               double *out_dev, const double *shift_dev = nullptr, const double *bias_dev = nullptr)
{
  cusmc_ctx *ctx = dist->ctx;
  const Epilogue ep = make_epilogue(dist, flags);
  const int d = dist->d;
  bool has_shift = shift_dev != nullptr;
  for (double v : dist->plan_shift) has_shift |= (v != 0.0);
  const double *shift = shift_dev ? shift_dev : (const double *)dist->shift.p;
  const double *bias = bias_dev ? bias_dev : (const double *)dist->bias.p;
  if (cusmc::mfma_wide_supported(d, X_dev, ldx)) {
    if (int rc = ensure_frags(dist, 2)) return rc;
    HIP_TRY(cusmc::launch_logpdf_mfma_wide(X_dev, N, ldx, d, dist->plan == 1, has_shift,
                                           (const double *)dist->frags.p, shift, bias, ep, out_dev, ctx->num_cus,
                                           ctx->stream));
    return CUSMC_OK;
  }
  if (cusmc::mfma_supported(d, X_dev, ldx)) {
    if (int rc = ensure_frags(dist, 1)) return rc;
    if (!ctx->nb4_pool.p) {  // the assembly kernel's tail-pool counters: zeroed once, the kernel leaves them zeroed
      if (int rc = ctx->nb4_pool.reserve(cusmc::nb4_pool_bytes())) return rc;
      HIP_TRY(hipMemsetAsync(ctx->nb4_pool.p, 0, cusmc::nb4_pool_bytes(), ctx->stream));
    }
    ctx->nb4_state.dev = (unsigned *)ctx->nb4_pool.p;
    HIP_TRY(cusmc::launch_logpdf_mfma(X_dev, N, ldx, d, dist->plan == 1, has_shift, (const double *)dist->frags.p,
                                      shift, bias, ep, out_dev, ctx->num_cus, ctx->stream, &ctx->nb4_state));
    return CUSMC_OK;
  }
  if (!cusmc::generic_supported(d))
    return fail(CUSMC_ERANGE, "d = %d is beyond the generic log-pdf kernel (and not a multiple of 16)", d);
  if (int rc = ensure_M(dist)) return rc;
  HIP_TRY(cusmc::launch_logpdf_generic(X_dev, N, ldx, d, dist->plan_tri, (const double *)dist->Mdev.p, shift, bias,
                                       ep, out_dev, ctx->num_cus, ctx->stream));
  return CUSMC_OK;
}

int check_batch(const cusmc_dist *dist, const void *X, int64_t N, int64_t ldx, const void *out)
{
  if (!dist) return fail(CUSMC_EINVAL, "null distribution");
  if (N < 0) return fail(CUSMC_EINVAL, "N = %lld is negative", (long long)N);
  if (N > 0 && (!X || !out)) return fail(CUSMC_EINVAL, "null batch or output pointer");
  if (ldx < dist->d) return fail(CUSMC_EINVAL, "ldx = %lld < d = %d", (long long)ldx, dist->d);
  return CUSMC_OK;
}

// centred form:  r = x - F mu   (pdf(y, F); src/statistics.cc.cpp:192, :305)
int plan_centred(cusmc_dist *dist, const double *F)
{
  const int d = dist->d;
  std::vector<double> shift(dist->mu);
  if (F && !cusmc::la::is_identity(F, d)) cusmc::la::matvec(F, dist->mu.data(), d, shift);
  const std::vector<double> bias(d, 0.0);
  return install_plan(dist, 1, true, dist->W, nullptr, shift, bias);
}

// affine form:  r = y - F x   (reweight_G; src/mcmc.cpp:208).  With F = I this is the centred
// form with shift = y (q is even in r), which keeps the factor triangular.
int plan_affine(cusmc_dist *dist, const double *y, const double *F)
{
  const int d = dist->d;
  if (!y) return fail(CUSMC_EINVAL, "null y");
  if (!F || cusmc::la::is_identity(F, d)) {
    const std::vector<double> shift(y, y + d), bias(d, 0.0);
    return install_plan(dist, 1, true, dist->W, nullptr, shift, bias);
  }
  // z = W y - (W F) x
  std::vector<double> M, bias;
  const bool cached = dist->plan == 2 && dist->plan_F.size() == (size_t)d * d &&
                      !memcmp(dist->plan_F.data(), F, (size_t)d * d * 8);
  // d >= 16 (matrix-core kernels): only |z|^2 is wanted, so rotate z by the Q^T of -W F = Q L.  The
  // kernels then multiply by a lower triangular L -- d(d+1)/2 products per particle instead of
  // d^2 -- and the bias Q^T W y enters as the accumulators' initial value.
  const bool rotate = d >= 16;
  if (!cached) {
    cusmc::la::matmul(dist->W.data(), F, d, M);
    for (double &v : M) v = -v;
    if (rotate && !cusmc::la::is_lower_triangular(M.data(), d)) {
      std::vector<double> L;
      cusmc::la::ql_factor(M.data(), d, L, dist->plan_Qt);
      M.swap(L);
    } else {
      dist->plan_Qt.clear();
    }
  }
  cusmc::la::matvec(dist->W.data(), y, d, bias);
  if (!dist->plan_Qt.empty()) {
    std::vector<double> rb;
    cusmc::la::matvec(dist->plan_Qt.data(), bias.data(), d, rb);
    bias.swap(rb);
  }
  const std::vector<double> shift(d, 0.0);
  return install_plan(dist, 2, rotate, cached ? dist->hostM : M, F, shift, bias);
}


// ---- several GPUs below the C ABI: shared pieces ----------------------------------------------------------------
//
// The host-pointer entry points (what the R glue binds) shard over the devices named by CUSMC_DEVICES="0,1,.." or by
// the *_multi_host entry points: contiguous shards, one host thread per shard -- a pageable hipMemcpyAsync holds
// its calling thread, so only threads put several PCIe links to work at once --, each on a context and stream the
// LIBRARY owns (slot r of the list; created on first use and kept for the life of the process: nothing to
// finalize in the wrong order at exit).  A device may be listed more than once (how a one-GPU box rehearses the
// path).  One multi-device call at a time (the pool is locked for the duration of a call).
struct PoolSlot {
  cusmc_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
};
std::mutex g_pool_mutex;
std::vector<PoolSlot> g_pool;

// "0,1,2" -> {0,1,2}; unset or empty -> {}
int env_devices(std::vector<int> &devs)
{
  devs.clear();
  const char *env = getenv("CUSMC_DEVICES");
  if (!env) return CUSMC_OK;
  for (const char *c = env; *c;) {
    char *end = nullptr;
    const long v = strtol(c, &end, 10);
    if (end == c || (*end && *end != ','))
      return fail(CUSMC_EINVAL, "CUSMC_DEVICES='%s' is not a comma-separated list of device numbers", env);
    devs.push_back((int)v);
    c = *end == ',' ? end + 1 : end;
  }
  return CUSMC_OK;
}

int check_devices(const int *devices, int ndev)
{
  if (!devices || ndev < 1) return fail(CUSMC_EINVAL, "empty device list");
  if (ndev > cusmc::kMaxShards) return fail(CUSMC_ERANGE, "%d devices exceed the %d supported", ndev, cusmc::kMaxShards);
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0)
    return fail(CUSMC_ENODEVICE, "no HIP device visible: libcusmc_hip has no CPU fallback");
  for (int r = 0; r < ndev; ++r)
    if (devices[r] < 0 || devices[r] >= visible)
      return fail(CUSMC_EINVAL, "device %d out of range (%d visible)", devices[r], visible);
  return CUSMC_OK;
}

// slot r's context on `device` (g_pool_mutex held by the caller)
int pool_ctx(int slot, int device, cusmc_ctx **out)
{
  if ((int)g_pool.size() <= slot) g_pool.resize(slot + 1);
  PoolSlot &ps = g_pool[slot];
  if (ps.ctx && ps.ctx->device != device) {  // the list changed between calls
    ps.ctx->stream = nullptr;
    cusmc_ctx_destroy(ps.ctx);
    if (ps.stream) (void)hipStreamDestroy(ps.stream);
    ps = PoolSlot();
  }
  if (!ps.ctx) {
    if (int rc = cusmc_ctx_create(device, &ps.ctx)) return rc;
    const hipError_t e = hipStreamCreateWithFlags(&ps.stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      cusmc_ctx_destroy(ps.ctx);
      ps = PoolSlot();
      return fail(CUSMC_EHIP, "%s creating a stream on device %d", hipGetErrorString(e), device);
    }
    ps.ctx->stream = ps.stream;
  }
  *out = ps.ctx;
  return CUSMC_OK;
}

// the calling thread's current device, put back when a multi-device entry point returns
struct DeviceRestore {
  int saved = -1;
  DeviceRestore() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
  ~DeviceRestore() { if (saved >= 0) (void)hipSetDevice(saved); }
};

// contiguous shards of n items over `parts`: the first n % parts shards take one more
void shard_range(uint64_t n, int parts, int r, uint64_t *first, uint64_t *count)
{
  const uint64_t base = n / (uint64_t)parts, extra = n % (uint64_t)parts;
  *first = base * (uint64_t)r + ((uint64_t)r < extra ? (uint64_t)r : extra);
  *count = base + ((uint64_t)r < extra ? 1u : 0u);
}

// fn(r) -> status for r < n, shards that do NOT wait for each other: rank 0 on the calling thread, the others on
// threads of their own; a rank whose thread cannot be started runs on the caller afterwards; nothing thrown
// inside a rank leaves it (the ABI promises that no exception crosses it).  Returns the first failure and its text.
template <class F>
int run_sharded(int n, F &&fn)
{
  std::vector<int> rc(n, CUSMC_OK);
  std::vector<std::string> err(n);
  auto body = [&](int r) {
    try {
      rc[r] = fn(r);
    } catch (const std::exception &e) {
      rc[r] = fail(CUSMC_EINVAL, "exception in shard %d: %s", r, e.what());
    } catch (...) {
      rc[r] = fail(CUSMC_EINVAL, "exception in shard %d", r);
    }
    if (rc[r]) err[r] = g_last_error;
  };
  std::vector<std::thread> threads;
  std::vector<int> inline_ranks;
  for (int r = 1; r < n; ++r) {
    try {
      threads.emplace_back(body, r);
    } catch (...) {
      inline_ranks.push_back(r);
    }
  }
  body(0);
  for (int r : inline_ranks) body(r);
  for (auto &th : threads) th.join();
  for (int r = 0; r < n; ++r)
    if (rc[r]) {
      g_last_error = err[r];
      return rc[r];
    }
  return CUSMC_OK;
}

}  // namespace

// ---- library / context ----------------------------------------------------------------------

CUSMC_EXPORT const char *cusmc_version(void) { return "cusmc-hip 0.3 (gfx950; rng contract 4)"; }
CUSMC_EXPORT int cusmc_rng_contract(void) { return 4; }

// One independent Philox key per R-level call: the (call + 1)-th output of SplitMix64 seeded with
// `seed`.  The R-level draw / resample / run() entry points have no seed argument (the reference
// reseeds from std::random_device on every call: src/samplers.cpp:10-11, src/statistics.cc.cpp:231-232);
// the host layers (cusmc_amd/api.py, rcpp/src/glue.hpp) keep a session seed and a call counter and hand
// cusmc_stream_key(seed, counter) to the entry points below as their 64-bit seed, so that successive
// calls never share a (key, counter) pair while a seeded session stays reproducible.
CUSMC_EXPORT uint64_t cusmc_stream_key(uint64_t seed, uint64_t call)
{
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (call + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
CUSMC_EXPORT const char *cusmc_last_error(void) { return g_last_error.c_str(); }

CUSMC_EXPORT int cusmc_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {
// CUSMC_TRACE_TERMINATE=1: print the call stack of whatever reaches std::terminate (diagnostic for
// aborts at process exit: whose destructor threw).  Off by default.
void trace_terminate()
{
  static const char msg[] = "[cusmc] std::terminate reached; call stack:\n";
  (void)!write(2, msg, sizeof msg - 1);
  void *frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  abort();
}
}  // namespace

CUSMC_EXPORT int cusmc_ctx_create(int device, cusmc_ctx **out)
{
  if (!out) return fail(CUSMC_EINVAL, "null output pointer");
  *out = nullptr;
  static const bool traced = [] {
    if (getenv("CUSMC_TRACE_TERMINATE")) std::set_terminate(trace_terminate);
    return true;
  }();
  (void)traced;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(CUSMC_ENODEVICE, "no HIP device visible: libcusmc_hip has no CPU fallback");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= n) return fail(CUSMC_EINVAL, "device %d out of range (%d visible)", device, n);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CUSMC_ENODEVICE, "device %d is %s; this library carries gfx950 code only", device,
                prop.gcnArchName);
  cusmc_ctx *ctx = new (std::nothrow) cusmc_ctx;
  if (!ctx) return fail(CUSMC_EINVAL, "out of host memory");
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  ctx->propagate_rows = getenv("CUSMC_PROPAGATE_ROWS") != nullptr;
  *out = ctx;
  return CUSMC_OK;
}

// Destroy functions run from finalizers -- R's at session end, Python's at interpreter exit, in any order
// and possibly while the HIP runtime is itself shutting down.  Two rules keep that safe:
//   (1) a distribution never dereferences a dead context: destroying the context orphans its
//       distributions (dist->ctx = nullptr) and releases their device buffers itself.  Round 1's abort at
//       pytest exit (`std::bad_variant_access`, thrown inside libamdhip64 -- the only library in the
//       process besides torch that carries that type) was this: a distribution finalized AFTER its context
//       read ctx->device / ctx->stream out of freed memory and handed the garbage to hipSetDevice /
//       hipStreamSynchronize / hipFree, whose handle lookup threw.  DESIGN.md section 9.
//   (2) nothing the runtime throws or returns during teardown leaves these functions: HIP status codes
//       (hipErrorDeinitialized included) are ignored here, and a C++ exception from inside the runtime is
//       swallowed -- the ABI promises that no exception crosses it.
CUSMC_EXPORT int cusmc_ctx_destroy(cusmc_ctx *ctx)
{
  if (!ctx) return CUSMC_OK;
  try {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (cusmc_dist *dist : ctx->dists) {
      dist->frags.release();
      dist->Mdev.release();
      dist->shift.release();
      dist->bias.release();
      dist->ctx = nullptr;
    }
    ctx->dists.clear();
    for (auto &b : ctx->scratch) b.release();
    ctx->draw_img.release();
    ctx->whi.release();
    ctx->nb4_pool.release();
    ctx->step_mats.release();
    ctx->ring.release();
  } catch (...) {
    for (cusmc_dist *dist : ctx->dists) dist->ctx = nullptr;  // (still orphan them: rule 1)
  }
  delete ctx;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_ctx_set_stream(cusmc_ctx *ctx, void *hip_stream)
{
  if (!ctx) return fail(CUSMC_EINVAL, "null context");
  hipStream_t next = reinterpret_cast<hipStream_t>(hip_stream);
  if (next != ctx->stream && ctx->nb4_pool.p) {
    // the assembly log-pdf kernel's counter blocks alternate between CONSECUTIVE launches of one stream: work still
    // running on the old stream must be through with them before launches on the new one take their turn
    if (int rc = activate(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  ctx->stream = next;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_ctx_synchronize(cusmc_ctx *ctx)
{
  if (int rc = activate(ctx)) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // diagnostic (scripts/calib/nb4_stamps.py): CUSMC_NB4_STAMPS=<file> makes the assembly log-pdf kernel record every
  // wave's entry / exit time behind its pool counters; a synchronize writes the last launch's records to the file
  if (const char *path = getenv("CUSMC_NB4_STAMPS")) {
    if (ctx->nb4_pool.p && path[0] && path[0] != '0' && path[0] != '1') {
      std::vector<char> buf(cusmc::nb4_pool_bytes() - 8192);
      HIP_TRY(hipMemcpy(buf.data(), (const char *)ctx->nb4_pool.p + 8192, buf.size(), hipMemcpyDeviceToHost));
      if (FILE *f = fopen(path, "wb")) {
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
      }
    }
  }
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_ctx_num_cus(cusmc_ctx *ctx, int *out)
{
  if (!ctx || !out) return fail(CUSMC_EINVAL, "null argument");
  *out = ctx->num_cus;
  return CUSMC_OK;
}

// ---- distributions --------------------------------------------------------------------------

CUSMC_EXPORT int cusmc_dist_create(cusmc_ctx *ctx, int kind, const double *mu, const double *sigma,
                                   int d, float nu, cusmc_dist **out)
{
  if (!out) return fail(CUSMC_EINVAL, "null output pointer");
  *out = nullptr;
  if (int rc = activate(ctx)) return rc;
  if (kind != CUSMC_MVN && kind != CUSMC_MVT) return fail(CUSMC_EINVAL, "unknown distribution kind %d", kind);
  if (!sigma) return fail(CUSMC_EINVAL, "null sigma");
  if (d < 1) return fail(CUSMC_EINVAL, "d = %d must be positive", d);
  if (d > CUSMC_MAX_DIM) return fail(CUSMC_ERANGE, "d = %d exceeds CUSMC_MAX_DIM = %d", d, CUSMC_MAX_DIM);
  if (kind == CUSMC_MVT && !(nu > 0.f)) return fail(CUSMC_EINVAL, "nu = %g must be positive", (double)nu);

  std::vector<double> L;
  const int rc = cusmc::la::cholesky(sigma, d, L);
  if (rc == -1) return fail(CUSMC_ENOTSPD, "sigma is not symmetric");
  if (rc) return fail(CUSMC_ENOTSPD, "sigma is not positive definite (pivot %d)", rc - 1);

  cusmc_dist *dist = new (std::nothrow) cusmc_dist;
  if (!dist) return fail(CUSMC_EINVAL, "out of host memory");
  dist->ctx = ctx;
  dist->kind = kind;
  dist->d = d;
  dist->nu = kind == CUSMC_MVT ? nu : 0.f;
  if (mu) dist->mu.assign(mu, mu + d); else dist->mu.assign(d, 0.0);
  dist->sigma.assign(sigma, sigma + (size_t)d * d);
  cusmc::la::lower_inverse(L, d, dist->W);
  double logdet = 0.0;
  for (int i = 0; i < d; ++i) logdet += 2.0 * std::log(L[(size_t)i * d + i]);
  dist->logdet = logdet;
  const double pi = 3.14159265358979323846;
  if (kind == CUSMC_MVN) {
    // log of 1 / ((sqrt 2pi)^n det^1/2)                       src/statistics.cc.cpp:205-211
    dist->lognorm = -0.5 * ((double)d * std::log(2.0 * pi) + logdet);
  } else {
    // log of (pi nu)^(-n/2) det^(-1/2) Gamma((nu+n)/2) / Gamma(nu/2)   :332-340; lgamma, so no
    // tgamma overflow at d >~ 340; nu + n in float as the reference computes it
    const float nu_plus_d = nu + (float)d;
    dist->lognorm = std::lgamma(0.5 * (double)nu_plus_d) - std::lgamma(0.5 * (double)nu) -
                    0.5 * (double)d * std::log(pi * (double)nu) - 0.5 * logdet;
  }
  ctx->dists.push_back(dist);
  *out = dist;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_dist_destroy(cusmc_dist *dist)
{
  if (!dist) return CUSMC_OK;
  for (cusmc_dist *rep : dist->replicas) cusmc_dist_destroy(rep);
  dist->replicas.clear();
  if (cusmc_ctx *ctx = dist->ctx) {  // (null: the context went first and took the device buffers with it)
    try {
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      dist->frags.release();
      dist->Mdev.release();
      dist->shift.release();
      dist->bias.release();
    } catch (...) {  // (rule 2 above)
    }
    ctx->dists.erase(std::remove(ctx->dists.begin(), ctx->dists.end(), dist), ctx->dists.end());
  }
  delete dist;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_dist_lognorm(const cusmc_dist *dist, double *out)
{
  if (!dist || !out) return fail(CUSMC_EINVAL, "null argument");
  *out = dist->lognorm;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_dist_logdet(const cusmc_dist *dist, double *out)
{
  if (!dist || !out) return fail(CUSMC_EINVAL, "null argument");
  *out = dist->logdet;
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_dist_pdf_dev(cusmc_dist *dist, const double *X_dev, int64_t N, int64_t ldx,
                                    const double *F, int flags, double *out_dev)
{
  if (int rc = check_batch(dist, X_dev, N, ldx, out_dev)) return rc;
  if (int rc = activate(dist->ctx)) return rc;
  if (N == 0) return CUSMC_OK;
  if (int rc = plan_centred(dist, F)) return rc;
  return run_logpdf(dist, X_dev, N, ldx, flags, out_dev);
}

CUSMC_EXPORT int cusmc_dist_reweight_dev(cusmc_dist *dist, const double *X_dev, int64_t N,
                                         int64_t ldx, const double *y, const double *F, int flags,
                                         double *out_dev)
{
  if (int rc = check_batch(dist, X_dev, N, ldx, out_dev)) return rc;
  if (int rc = activate(dist->ctx)) return rc;
  if (N == 0) return CUSMC_OK;
  if (int rc = plan_affine(dist, y, F)) return rc;
  return run_logpdf(dist, X_dev, N, ldx, flags, out_dev);
}

namespace {

// Shared body of the two host-pointer density entry points.
int host_density(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx, const double *y,
                 const double *F, int flags, double *out, bool affine)
{
  if (int rc = check_batch(dist, X, N, ldx, out)) return rc;
  cusmc_ctx *ctx = dist->ctx;
  if (int rc = activate(ctx)) return rc;
  if (N == 0) return CUSMC_OK;
  const size_t xbytes = ((size_t)(N - 1) * ldx + dist->d) * 8;
  if (int rc = ctx->scratch[0].reserve(xbytes)) return rc;
  if (int rc = ctx->scratch[1].reserve((size_t)N * 8)) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->scratch[0].p, X, xbytes, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = affine ? plan_affine(dist, y, F) : plan_centred(dist, F)) return rc;
  if (int rc = run_logpdf(dist, (const double *)ctx->scratch[0].p, N, ldx, flags, (double *)ctx->scratch[1].p))
    return rc;
  HIP_TRY(hipMemcpyAsync(out, ctx->scratch[1].p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSMC_OK;
}

// Rows sharded over a device list: shard r = rows [first_r, first_r + count_r) on slot r's context, through a
// replica of the distribution there (factored and uploaded once per slot, kept with the handle).  Every row's
// value is the single-device value bit for bit (a particle's result does not depend on its tile mates).
int multi_density(cusmc_dist *dist, const int *devices, int ndev, const double *X, int64_t N, int64_t ldx,
                  const double *y, const double *F, int flags, double *out, bool affine)
{
  if (int rc = check_batch(dist, X, N, ldx, out)) return rc;
  if (int rc = check_devices(devices, ndev)) return rc;
  if (N == 0) return CUSMC_OK;
  if (affine && !y) return fail(CUSMC_EINVAL, "null y");
  // (a shard below ~1000 rows costs more in threads and launches than it saves)
  const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(ndev, N / 1024));
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  DeviceRestore restore;
  if ((int)dist->replicas.size() < parts) dist->replicas.resize(parts, nullptr);
  for (int r = 0; r < parts; ++r) {
    cusmc_ctx *ctx = nullptr;
    if (int rc = pool_ctx(r, devices[r], &ctx)) return rc;
    cusmc_dist *&rep = dist->replicas[r];
    if (rep && rep->ctx != ctx) {  // the slot's context was rebuilt (another device list): start over
      cusmc_dist_destroy(rep);
      rep = nullptr;
    }
    if (!rep)
      if (int rc = cusmc_dist_create(ctx, dist->kind, dist->mu.data(), dist->sigma.data(), dist->d, dist->nu, &rep)) return rc;
  }
  return run_sharded(parts, [&](int r) {
    uint64_t first, count;
    shard_range((uint64_t)N, parts, r, &first, &count);
    return host_density(dist->replicas[r], X + first * (uint64_t)ldx, (int64_t)count, ldx, y, F, flags, out + first, affine);
  });
}

// CUSMC_DEVICES in the environment: the host-pointer call goes to that device list (one entry: that device)
int routed_density(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx, const double *y, const double *F, int flags,
                   double *out, bool affine)
{
  std::vector<int> devs;
  if (int rc = env_devices(devs)) return rc;
  if (!devs.empty() && dist && dist->ctx && !(devs.size() == 1 && devs[0] == dist->ctx->device))
    return multi_density(dist, devs.data(), (int)devs.size(), X, N, ldx, y, F, flags, out, affine);
  return host_density(dist, X, N, ldx, y, F, flags, out, affine);
}

}  // namespace

CUSMC_EXPORT int cusmc_dist_pdf_host(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx,
                                     const double *F, int flags, double *out)
{
  return routed_density(dist, X, N, ldx, nullptr, F, flags, out, false);
}

CUSMC_EXPORT int cusmc_dist_reweight_host(cusmc_dist *dist, const double *X, int64_t N, int64_t ldx,
                                          const double *y, const double *F, int flags, double *out)
{
  return routed_density(dist, X, N, ldx, y, F, flags, out, true);
}

CUSMC_EXPORT int cusmc_dist_pdf_multi_host(cusmc_dist *dist, const int *devices, int ndev, const double *X, int64_t N,
                                           int64_t ldx, const double *F, int flags, double *out)
{
  return multi_density(dist, devices, ndev, X, N, ldx, nullptr, F, flags, out, false);
}

CUSMC_EXPORT int cusmc_dist_reweight_multi_host(cusmc_dist *dist, const int *devices, int ndev, const double *X,
                                                int64_t N, int64_t ldx, const double *y, const double *F, int flags,
                                                double *out)
{
  return multi_density(dist, devices, ndev, X, N, ldx, y, F, flags, out, true);
}

// ---- resampler ------------------------------------------------------------------------------

CUSMC_EXPORT int cusmc_metropolis_dev(cusmc_ctx *ctx, const double *w_dev, uint32_t N, uint32_t B,
                                      uint64_t seed, uint32_t step, uint32_t first, uint32_t count,
                                      uint32_t *a_dev)
{
  if (int rc = activate(ctx)) return rc;
  if ((uint64_t)first + count > N) return fail(CUSMC_EINVAL, "shard [%u, %u) exceeds N = %u", first, first + count, N);
  if (count == 0) return CUSMC_OK;
  if (!w_dev || !a_dev) return fail(CUSMC_EINVAL, "null weight or ancestor pointer");
  const uint32_t *whi = nullptr;
  if ((cusmc::metropolis_wants_hiwords(N) && B > 1) || cusmc::metropolis_wants_lds_table(N, B, count)) {
    if (int rc = ctx->whi.reserve((size_t)N * 4)) return rc;
    HIP_TRY(cusmc::launch_hiwords(w_dev, N, (uint32_t *)ctx->whi.p, ctx->num_cus, ctx->stream));
    whi = (const uint32_t *)ctx->whi.p;
  }
  HIP_TRY(cusmc::launch_metropolis(w_dev, whi, N, B, seed, step, first, count, a_dev, ctx->num_cus, ctx->stream));
  return CUSMC_OK;
}

namespace {

// one device: upload w, run chains [first, first + count), bring their ancestors back
int host_metropolis(cusmc_ctx *ctx, const double *w, uint32_t N, uint32_t B, uint64_t seed, uint32_t step, bool logw,
                    uint32_t first, uint32_t count, uint32_t *a)
{
  if (int rc = activate(ctx)) return rc;
  if (count == 0) return CUSMC_OK;
  if (int rc = ctx->scratch[2].reserve((size_t)N * 8)) return rc;
  if (int rc = ctx->scratch[3].reserve((size_t)count * 4)) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->scratch[2].p, w, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = logw ? cusmc_metropolis_log_dev(ctx, (const double *)ctx->scratch[2].p, N, B, seed, step, first, count,
                                               (uint32_t *)ctx->scratch[3].p)
                    : cusmc_metropolis_dev(ctx, (const double *)ctx->scratch[2].p, N, B, seed, step, first, count,
                                           (uint32_t *)ctx->scratch[3].p))
    return rc;
  HIP_TRY(hipMemcpyAsync(a, ctx->scratch[3].p, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSMC_OK;
}

// Chains sharded over a device list (BASELINE configs[3]: "chains sharded over 8 x MI355X"): every device takes the
// whole weight vector (8 N bytes over its own PCIe link) and runs chains [first_r, first_r + count_r); the draws
// are keyed by the global chain index, so the ancestors are the single-device ones bit for bit.
int multi_metropolis(const int *devices, int ndev, const double *w, uint32_t N, uint32_t B, uint64_t seed, uint32_t step,
                     bool logw, uint32_t *a)
{
  if (int rc = check_devices(devices, ndev)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!w || !a) return fail(CUSMC_EINVAL, "null weight or ancestor pointer");
  // (a shard is worth its upload of the whole weight vector only with enough chain steps behind it)
  const uint64_t work = (uint64_t)N * B;
  const int parts = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)ndev, (uint64_t)N, work / 100000u}));
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  DeviceRestore restore;
  std::vector<cusmc_ctx *> ctxs(parts, nullptr);
  for (int r = 0; r < parts; ++r)
    if (int rc = pool_ctx(r, devices[r], &ctxs[r])) return rc;
  return run_sharded(parts, [&](int r) {
    uint64_t first, count;
    shard_range(N, parts, r, &first, &count);
    return host_metropolis(ctxs[r], w, N, B, seed, step, logw, (uint32_t)first, (uint32_t)count, a + first);
  });
}

int routed_metropolis(cusmc_ctx *ctx, const double *w, uint32_t N, uint32_t B, uint64_t seed, uint32_t step, bool logw,
                      uint32_t *a)
{
  if (int rc = activate(ctx)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!w || !a) return fail(CUSMC_EINVAL, "null weight or ancestor pointer");
  std::vector<int> devs;
  if (int rc = env_devices(devs)) return rc;
  if (!devs.empty() && !(devs.size() == 1 && devs[0] == ctx->device))
    return multi_metropolis(devs.data(), (int)devs.size(), w, N, B, seed, step, logw, a);
  return host_metropolis(ctx, w, N, B, seed, step, logw, 0, N, a);
}

}  // namespace

CUSMC_EXPORT int cusmc_metropolis_host(cusmc_ctx *ctx, const double *w, uint32_t N, uint32_t B,
                                       uint64_t seed, uint32_t step, uint32_t *a)
{
  return routed_metropolis(ctx, w, N, B, seed, step, false, a);
}

CUSMC_EXPORT int cusmc_metropolis_multi_host(const int *devices, int ndev, const double *w, uint32_t N, uint32_t B,
                                             uint64_t seed, uint32_t step, int log_weights, uint32_t *a)
{
  return multi_metropolis(devices, ndev, w, N, B, seed, step, log_weights != 0, a);
}

CUSMC_EXPORT int cusmc_metropolis_log_dev(cusmc_ctx *ctx, const double *logw_dev, uint32_t N, uint32_t B,
                                          uint64_t seed, uint32_t step, uint32_t first, uint32_t count,
                                          uint32_t *a_dev)
{
  if (int rc = activate(ctx)) return rc;
  if ((uint64_t)first + count > N) return fail(CUSMC_EINVAL, "shard [%u, %u) exceeds N = %u", first, first + count, N);
  if (count == 0) return CUSMC_OK;
  if (!logw_dev || !a_dev) return fail(CUSMC_EINVAL, "null weight or ancestor pointer");
  HIP_TRY(cusmc::launch_metropolis_log(logw_dev, N, B, seed, step, first, count, a_dev, ctx->num_cus, ctx->stream));
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_metropolis_log_host(cusmc_ctx *ctx, const double *logw, uint32_t N, uint32_t B,
                                           uint64_t seed, uint32_t step, uint32_t *a)
{
  return routed_metropolis(ctx, logw, N, B, seed, step, true, a);
}

// ---- proposal draws -------------------------------------------------------------------------

namespace {

int draws(cusmc_ctx *ctx, int kind, float nu, const double *X_prev_dev, const uint32_t *a_dev,
          const double *G, const double *Q, const double *m0, int d, double scale, uint64_t seed,
          uint32_t step, uint32_t domain, uint32_t first, uint32_t count, double *X_out_dev)
{
  if (int rc = activate(ctx)) return rc;
  if (kind != CUSMC_MVN && kind != CUSMC_MVT) return fail(CUSMC_EINVAL, "unknown distribution kind %d", kind);
  if (kind == CUSMC_MVT && !(nu > 0.f)) return fail(CUSMC_EINVAL, "nu = %g must be positive", (double)nu);
  if (d < 1 || !Q) return fail(CUSMC_EINVAL, "bad d or null Q");
  if (count == 0) return CUSMC_OK;
  if (!X_out_dev) return fail(CUSMC_EINVAL, "null output pointer");
  const size_t dd = (size_t)d * d;
  // The device image of (Q, G, m0) in the layout the chosen kernel reads, built and uploaded only when
  // the layout or the host values differ from the previous call's.
  std::vector<double> key;
  key.reserve(2 * dd + d + 2);
  key.insert(key.end(), Q, Q + dd);
  key.push_back(G ? 1.0 : 0.0);
  if (G) key.insert(key.end(), G, G + dd);
  key.push_back(m0 ? 1.0 : 0.0);
  if (m0) key.insert(key.end(), m0, m0 + d);
  auto image = [&](int layout, size_t doubles, auto &&build) -> int {
    if (ctx->draw_layout == layout && ctx->draw_img.p && ctx->draw_key.size() == key.size() &&
        !memcmp(ctx->draw_key.data(), key.data(), key.size() * 8))
      return CUSMC_OK;
    std::vector<double> img(doubles, 0.0);
    build(img);
    ctx->draw_layout = 0;
    if (int rc = ctx->draw_img.reserve(doubles * 8)) return rc;
    if (int rc = ctx->ring.upload(ctx->draw_img.p, img.data(), doubles * 8, ctx->stream)) return rc;
    ctx->draw_layout = layout;
    ctx->draw_key.swap(key);
    return CUSMC_OK;
  };
  if (cusmc::la::is_diagonal(Q, d) && (!G || cusmc::la::is_diagonal(G, d))) {
    // device image: [diag(Q) | diag(G) | m0]
    if (int rc = image(1, 3 * (size_t)d, [&](std::vector<double> &img) {
          for (int j = 0; j < d; ++j) {
            img[j] = Q[(size_t)j * d + j];
            if (G) img[d + j] = G[(size_t)j * d + j];
            if (m0) img[2 * d + j] = m0[j];
          }
        }))
      return rc;
    const double *base = (const double *)ctx->draw_img.p;
    HIP_TRY(cusmc::launch_propagate_diag(kind, nu, X_prev_dev, a_dev, G ? base + d : nullptr, base,
                                         m0 ? base + 2 * d : nullptr, d, scale, seed, step, domain, first, count,
                                         X_out_dev, ctx->num_cus, ctx->stream));
    return CUSMC_OK;
  }
  if (cusmc::propagate_mfma_supported(d, X_prev_dev, X_out_dev)) {
    // device image: [frags(Q) | frags(G) | m0], fragments in the MFMA operand order; factors
    // zero-padded to 16*ceil(d/16)
    const int nb = (d + 15) / 16, dp = 16 * nb;
    // a lower triangular Q (a Cholesky factor: what cusmc_pf_run_* passes) travels and multiplies as a triangle
    const bool triq = nb >= 2 && cusmc::la::is_lower_triangular(Q, d);
    const size_t nf = (size_t)cusmc::mfma_num_frags(nb, false) * 64, nfq = (size_t)cusmc::mfma_num_frags(nb, triq) * 64;
    // a diagonal G (random walk, AR(1) per component) under a dense Q needs no second matrix product:
    // its d entries travel instead of its fragments
    // (from d = 17 up: at one 16-block the 8-byte gathers cost more than the block's 16 MFMAs save,
    // 81 against 69 us per 1e6 particles)
    const bool g_diag = G && nb >= 2 && cusmc::la::is_diagonal(G, d);
    if (int rc = image((g_diag ? 5 : 2) + (triq ? 16 : 0), nfq + nf + d, [&](std::vector<double> &img) {
          std::vector<double> Mp;
          auto pack = [&](const double *M, bool tri, double *dst) {
            if (dp == d) { cusmc::mfma_pack_frags(M, d, tri, dst); return; }
            Mp.assign((size_t)dp * dp, 0.0);
            for (int i = 0; i < d; ++i) std::copy(M + (size_t)i * d, M + (size_t)(i + 1) * d, Mp.begin() + (size_t)i * dp);
            cusmc::mfma_pack_frags(Mp.data(), dp, tri, dst);
          };
          pack(Q, triq, img.data());
          if (g_diag) {
            for (int j = 0; j < d; ++j) img[nfq + j] = G[(size_t)j * d + j];
          } else if (G) {
            pack(G, false, img.data() + nfq);
          }
          if (m0) std::copy(m0, m0 + d, img.begin() + nfq + nf);
        }))
      return rc;
    const double *base = (const double *)ctx->draw_img.p;
    HIP_TRY((triq ? cusmc::launch_propagate_mfma_tri : cusmc::launch_propagate_mfma)(
        kind, nu, X_prev_dev, a_dev, base, G ? base + nfq : nullptr, g_diag, m0 ? base + nfq + nf : nullptr, d, scale, seed,
        step, domain, first, count, X_out_dev, ctx->num_cus, ctx->stream));
    return CUSMC_OK;
  }
  // (CUSMC_PROPAGATE_ROWS=1 when the context was created: round 1's one-workgroup-per-particle kernel instead, for A/B timing)
  if (cusmc::propagate_mfma_wide_supported(d, X_prev_dev, X_out_dev) && !ctx->propagate_rows) {
    // 128 < d <= 256: matrix cores with the output blocks split over the waves (kernels/propagate_mfma_wide.hip).
    // device image: [frags(Q) | frags(G) or diag(G) or m0], factors zero-padded to 16*ceil(d/16)
    const int nb = (d + 15) / 16, dp = 16 * nb;
    const bool triq = cusmc::la::is_lower_triangular(Q, d);
    const size_t nf = (size_t)cusmc::mfma_num_frags(nb, false) * 64, nfq = (size_t)cusmc::mfma_num_frags(nb, triq) * 64;
    const bool g_diag = G && cusmc::la::is_diagonal(G, d);
    const int mode = !G ? 0 : g_diag ? 4 : 1;
    if (int rc = image((mode == 1 ? 6 : mode == 4 ? 7 : 8) + (triq ? 16 : 0), nfq + (mode == 1 ? nf : (size_t)dp), [&](std::vector<double> &img) {
          std::vector<double> Mp;
          auto pack = [&](const double *M, bool tri, double *dst) {
            if (dp == d) { cusmc::mfma_pack_frags(M, d, tri, dst); return; }
            Mp.assign((size_t)dp * dp, 0.0);
            for (int i = 0; i < d; ++i) std::copy(M + (size_t)i * d, M + (size_t)(i + 1) * d, Mp.begin() + (size_t)i * dp);
            cusmc::mfma_pack_frags(Mp.data(), dp, tri, dst);
          };
          pack(Q, triq, img.data());
          if (mode == 1) pack(G, false, img.data() + nfq);
          else if (mode == 4) for (int j = 0; j < d; ++j) img[nfq + j] = G[(size_t)j * d + j];
          else if (m0) std::copy(m0, m0 + d, img.begin() + nfq);
        }))
      return rc;
    const double *base = (const double *)ctx->draw_img.p;
    HIP_TRY((triq ? cusmc::launch_propagate_mfma_wide_tri : cusmc::launch_propagate_mfma_wide)(
        kind, nu, X_prev_dev, a_dev, base, base + nfq, mode, d, scale, seed, step, domain, first, count, X_out_dev,
        ctx->num_cus, ctx->stream));
    return CUSMC_OK;
  }
  if (d > 128) {
    // device image: [Q^T | G^T | m0] (one workgroup per particle, kernels/propagate.hip)
    if (int rc = image(3, 2 * dd + d, [&](std::vector<double> &img) {
          for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
              img[(size_t)j * d + i] = Q[(size_t)i * d + j];
              if (G) img[dd + (size_t)j * d + i] = G[(size_t)i * d + j];
            }
          if (m0) std::copy(m0, m0 + d, img.begin() + 2 * dd);
        }))
      return rc;
    const double *base = (const double *)ctx->draw_img.p;
    HIP_TRY(cusmc::launch_propagate_rows(kind, nu, X_prev_dev, a_dev, G ? base + dd : nullptr, base,
                                         m0 ? base + 2 * dd : nullptr, d, scale, seed, step, domain, first, count,
                                         X_out_dev, ctx->num_cus, ctx->stream));
    return CUSMC_OK;
  }
  // device image: [Q | G | m0]
  if (int rc = image(4, 2 * dd + d, [&](std::vector<double> &img) {
        std::copy(Q, Q + dd, img.begin());
        if (G) std::copy(G, G + dd, img.begin() + dd);
        if (m0) std::copy(m0, m0 + d, img.begin() + 2 * dd);
      }))
    return rc;
  const double *base = (const double *)ctx->draw_img.p;
  HIP_TRY(cusmc::launch_propagate(kind, nu, X_prev_dev, a_dev, G ? base + dd : nullptr, base,
                                  m0 ? base + 2 * dd : nullptr, d, scale, seed, step, domain, first,
                                  count, X_out_dev, ctx->num_cus, ctx->stream));
  return CUSMC_OK;
}

}  // namespace

CUSMC_EXPORT int cusmc_propagate_dev(cusmc_ctx *ctx, int kind, float nu, const double *X_prev_dev,
                                     const uint32_t *a_dev, uint32_t N, int d, const double *G,
                                     const double *Q, double scale, uint64_t seed, uint32_t step,
                                     uint32_t first, uint32_t count, double *X_out_dev)
{
  if (!ctx) return fail(CUSMC_EINVAL, "null context");
  if (!G || !X_prev_dev) return fail(CUSMC_EINVAL, "null G or X_prev");
  // a == NULL reads X_prev[i] for the global i itself; with an ancestor array the rows of X_prev are
  // named by a[] alone and [first, first+count) only keys the draws (a sharded caller passes the
  // rows it fetched for its own particles: N = count, a = 0 .. count-1, first = its global offset)
  if (!a_dev && (uint64_t)first + count > N)
    return fail(CUSMC_EINVAL, "shard [%u, %u) exceeds N = %u", first, first + count, N);
  if ((uint64_t)first + count > 0xffffffffull) return fail(CUSMC_EINVAL, "first + count exceeds 2^32");
  return draws(ctx, kind, nu, X_prev_dev, a_dev, G, Q, nullptr, d, scale, seed, step, 2u, first, count, X_out_dev);
}

CUSMC_EXPORT int cusmc_initialize_dev(cusmc_ctx *ctx, int kind, float nu, const double *m0,
                                      const double *Q, int d, double scale, uint64_t seed,
                                      uint32_t first, uint32_t count, double *X_out_dev)
{
  if (!ctx) return fail(CUSMC_EINVAL, "null context");
  if (!m0) return fail(CUSMC_EINVAL, "null m0");
  return draws(ctx, kind, nu, nullptr, nullptr, nullptr, Q, m0, d, scale, seed, 0u, 4u, first, count, X_out_dev);
}

CUSMC_EXPORT int cusmc_sample_host(cusmc_ctx *ctx, int kind, float nu, const double *mu,
                                   const double *Q, int d, double scale, uint64_t seed,
                                   uint32_t step, uint32_t count, double *X_out)
{
  if (!ctx) return fail(CUSMC_EINVAL, "null context");
  if (!mu || !X_out) return fail(CUSMC_EINVAL, "null mu or output pointer");
  if (d < 1) return fail(CUSMC_EINVAL, "d = %d must be positive", d);
  if (count == 0) return CUSMC_OK;
  if (int rc = activate(ctx)) return rc;
  if (int rc = ctx->scratch[5].reserve((size_t)count * d * 8)) return rc;
  if (int rc = draws(ctx, kind, nu, nullptr, nullptr, nullptr, Q, mu, d, scale, seed, step, 4u, 0, count,
                     (double *)ctx->scratch[5].p))
    return rc;
  HIP_TRY(hipMemcpyAsync(X_out, ctx->scratch[5].p, (size_t)count * d * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_eigen_sqrt(const double *sigma, int d, double *Q)
{
  if (!sigma || !Q || d < 1) return fail(CUSMC_EINVAL, "bad argument");
  cusmc::la::eigen_sqrt(sigma, d, Q);
  return CUSMC_OK;
}

// ---- one filter time step -------------------------------------------------------------------

// ---- per-particle covariances ------------------------------------------------------------------

namespace {

int percov_epilogue(int kind, float nu, int d, int flags, Epilogue *ep)
{
  if (kind != CUSMC_MVN && kind != CUSMC_MVT) return fail(CUSMC_EINVAL, "unknown distribution kind %d", kind);
  if (kind == CUSMC_MVT && !(nu > 0.f)) return fail(CUSMC_EINVAL, "nu = %g must be positive", (double)nu);
  const double pi = 3.14159265358979323846;
  const float nu_plus_d = nu + (float)d;  // float arithmetic, as the reference (statistics.cc.cpp:332-340)
  ep->kind = kind;
  ep->out_density = (flags & CUSMC_OUT_DENSITY) ? 1 : 0;
  ep->half_nu_plus_d = 0.5 * (double)nu_plus_d;
  ep->inv_nu = kind == CUSMC_MVT ? 1.0 / (double)nu : 0.0;
  // the normalising constants of cusmc_dist_create() without their determinant term
  ep->lognorm = kind == CUSMC_MVN ? -0.5 * (double)d * std::log(2.0 * pi)
                                  : std::lgamma(0.5 * (double)nu_plus_d) - std::lgamma(0.5 * (double)nu) -
                                        0.5 * (double)d * std::log(pi * (double)nu);
  return CUSMC_OK;
}

int percov_check(int64_t N, int d)
{
  if (N < 0) return fail(CUSMC_EINVAL, "N = %lld is negative", (long long)N);
  if (!cusmc::percov_supported(d))
    return fail(CUSMC_ERANGE, "per-particle covariances are served for 1 <= d <= 16 (d = %d)", d);
  return CUSMC_OK;
}

}  // namespace

CUSMC_EXPORT int cusmc_chol_batched_dev(cusmc_ctx *ctx, const double *sigma_dev, int64_t N, int d,
                                        double *L_dev, double *logdet_dev, int32_t *info_dev)
{
  if (int rc = activate(ctx)) return rc;
  if (int rc = percov_check(N, d)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!sigma_dev || !L_dev) return fail(CUSMC_EINVAL, "null sigma or L pointer");
  HIP_TRY(cusmc::launch_cholesky_batched(sigma_dev, N, d, L_dev, logdet_dev, info_dev, ctx->num_cus, ctx->stream));
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_chol_batched_host(cusmc_ctx *ctx, const double *sigma, int64_t N, int d, double *L,
                                         double *logdet, int32_t *info)
{
  if (int rc = activate(ctx)) return rc;
  if (int rc = percov_check(N, d)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!sigma || !L) return fail(CUSMC_EINVAL, "null sigma or L pointer");
  const size_t mbytes = (size_t)N * d * d * 8;
  if (int rc = ctx->scratch[0].reserve(mbytes)) return rc;
  if (int rc = ctx->scratch[4].reserve(mbytes)) return rc;
  if (int rc = ctx->scratch[1].reserve((size_t)N * 8)) return rc;
  if (int rc = ctx->scratch[3].reserve((size_t)N * 4)) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->scratch[0].p, sigma, mbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(cusmc::launch_cholesky_batched((const double *)ctx->scratch[0].p, N, d, (double *)ctx->scratch[4].p,
                                         (double *)ctx->scratch[1].p, (int *)ctx->scratch[3].p, ctx->num_cus,
                                         ctx->stream));
  HIP_TRY(hipMemcpyAsync(L, ctx->scratch[4].p, mbytes, hipMemcpyDeviceToHost, ctx->stream));
  if (logdet) HIP_TRY(hipMemcpyAsync(logdet, ctx->scratch[1].p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (info) HIP_TRY(hipMemcpyAsync(info, ctx->scratch[3].p, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_logpdf_percov_dev(cusmc_ctx *ctx, int kind, float nu, const double *X_dev, int64_t N,
                                         int64_t ldx, const double *mu_dev, int64_t ldmu,
                                         const double *sigma_dev, int d, int flags, double *out_dev,
                                         int32_t *info_dev)
{
  if (int rc = activate(ctx)) return rc;
  if (int rc = percov_check(N, d)) return rc;
  Epilogue ep;
  if (int rc = percov_epilogue(kind, nu, d, flags, &ep)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!X_dev || !sigma_dev || !out_dev) return fail(CUSMC_EINVAL, "null batch, sigma or output pointer");
  if (ldx < d) return fail(CUSMC_EINVAL, "ldx = %lld < d = %d", (long long)ldx, d);
  if (mu_dev && ldmu != 0 && ldmu < d) return fail(CUSMC_EINVAL, "ldmu = %lld is neither 0 nor >= d = %d", (long long)ldmu, d);
  HIP_TRY(cusmc::launch_logpdf_percov(X_dev, N, ldx, mu_dev, ldmu, sigma_dev, d, ep, out_dev, info_dev, ctx->num_cus,
                                      ctx->stream));
  return CUSMC_OK;
}

CUSMC_EXPORT int cusmc_logpdf_percov_host(cusmc_ctx *ctx, int kind, float nu, const double *X, int64_t N,
                                          int64_t ldx, const double *mu, int64_t ldmu, const double *sigma,
                                          int d, int flags, double *out, int32_t *info)
{
  if (int rc = activate(ctx)) return rc;
  if (int rc = percov_check(N, d)) return rc;
  if (N == 0) return CUSMC_OK;
  if (!X || !sigma || !out) return fail(CUSMC_EINVAL, "null batch, sigma or output pointer");
  if (ldx < d) return fail(CUSMC_EINVAL, "ldx = %lld < d = %d", (long long)ldx, d);
  if (mu && ldmu != 0 && ldmu < d) return fail(CUSMC_EINVAL, "ldmu = %lld is neither 0 nor >= d = %d", (long long)ldmu, d);
  const size_t xbytes = ((size_t)(N - 1) * ldx + d) * 8, mbytes = (size_t)N * d * d * 8;
  const size_t mubytes = !mu ? 0 : ldmu == 0 ? (size_t)d * 8 : ((size_t)(N - 1) * ldmu + d) * 8;
  if (int rc = ctx->scratch[0].reserve(xbytes)) return rc;
  if (int rc = ctx->scratch[4].reserve(mbytes)) return rc;
  if (int rc = ctx->scratch[1].reserve((size_t)N * 8)) return rc;
  if (int rc = ctx->scratch[3].reserve((size_t)N * 4)) return rc;
  if (mubytes) {
    if (int rc = ctx->scratch[5].reserve(mubytes)) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->scratch[5].p, mu, mubytes, hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(hipMemcpyAsync(ctx->scratch[0].p, X, xbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->scratch[4].p, sigma, mbytes, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = cusmc_logpdf_percov_dev(ctx, kind, nu, (const double *)ctx->scratch[0].p, N, ldx,
                                       mubytes ? (const double *)ctx->scratch[5].p : nullptr, ldmu,
                                       (const double *)ctx->scratch[4].p, d, flags, (double *)ctx->scratch[1].p,
                                       (int32_t *)ctx->scratch[3].p))
    return rc;
  HIP_TRY(hipMemcpyAsync(out, ctx->scratch[1].p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (info) HIP_TRY(hipMemcpyAsync(info, ctx->scratch[3].p, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return CUSMC_OK;
}

// ---- fused filter step (d <= 8): host-side pieces --------------------------------------------------
namespace {

bool pf_step_is_fused(int d, uint32_t count)
{
  // One launch instead of three pays where launches are what a step costs (small shards) or where the
  // state is a register pair (d <= 2); a big shard at d = 8 is quicker through the three specialised
  // kernels (105 vs 120 us for 1e6 particles): the fused kernel's register footprint halves occupancy.
  return cusmc::pf_step_supported(d) && (d <= 2 || count <= 200000u);
}

// observation plan for y, and the device image [Q | G]: the small matrices are read through
// wave-uniform (scalar) loads.  A filter calls this with the same matrices at every time step:
// only what changed is uploaded.
int pf_step_prepare(cusmc_dist *obs, const double *G, const double *Q, const double *y, const double *F)
{
  cusmc_ctx *ctx = obs->ctx;
  const int d = obs->d;
  if (int rc = plan_affine(obs, y, F)) return rc;
  if (int rc = ensure_M(obs)) return rc;
  const size_t dd = (size_t)d * d;
  std::vector<double> img(2 * dd);
  std::copy(Q, Q + dd, img.begin());
  std::copy(G, G + dd, img.begin() + dd);
  if (img != ctx->step_mats_host || !ctx->step_mats.p) {
    if (int rc = ctx->step_mats.reserve(2 * dd * 8)) return rc;
    if (int rc = ctx->ring.upload(ctx->step_mats.p, img.data(), 2 * dd * 8, ctx->stream)) return rc;
    ctx->step_mats_host.swap(img);
  }
  return CUSMC_OK;
}

// the launch(es) of one fused step; shift_dev / bias_dev: the observation plan's vectors for this
// step's y (obs->shift / obs->bias, or rows of a table uploaded ahead of a whole time loop)
int pf_step_launch(cusmc_dist *obs, int kind, float nu, const double *w_prev_dev, const double *X_prev_dev,
                   uint32_t N, uint32_t B, double scale, uint64_t seed, uint32_t step, uint32_t first,
                   uint32_t count, uint32_t *a_out_dev, double *X_out_dev, double *w_out_dev, int flags,
                   const double *shift_dev, const double *bias_dev, const cusmc::ShardStep *sharded = nullptr,
                   const uint32_t *whi_prev_dev = nullptr)
{
  cusmc_ctx *ctx = obs->ctx;
  const int d = obs->d;
  const size_t dd = (size_t)d * d;
  const double *base = (const double *)ctx->step_mats.p;
  const uint32_t *whi = nullptr;
  if (cusmc::metropolis_wants_hiwords(N) && B > 1) {
    if (sharded) {  // (the sharded step keeps the table itself: every shard's kernel writes its part of it)
      whi = whi_prev_dev;
    } else {
      if (int rc = ctx->whi.reserve((size_t)N * 4)) return rc;
      HIP_TRY(cusmc::launch_hiwords(w_prev_dev, N, (uint32_t *)ctx->whi.p, ctx->num_cus, ctx->stream));
      whi = (const uint32_t *)ctx->whi.p;
    }
  }
  HIP_TRY(cusmc::launch_pf_step(kind, nu, w_prev_dev, whi, X_prev_dev, N, d, B, base + dd, base, scale, obs->plan_tri,
                                (const double *)obs->Mdev.p, shift_dev, bias_dev, make_epilogue(obs, flags), seed,
                                step, first, count, a_out_dev, X_out_dev, w_out_dev, ctx->num_cus, ctx->stream, sharded));
  return CUSMC_OK;
}

}  // namespace

CUSMC_EXPORT int cusmc_pf_step_dev(cusmc_dist *obs, int kind, float nu, const double *w_prev_dev,
                                   const double *X_prev_dev, uint32_t N, const double *G,
                                   const double *Q, const double *y, const double *F, uint32_t B,
                                   double scale, uint64_t seed, uint32_t step, uint32_t first,
                                   uint32_t count, uint32_t *a_out_dev, double *X_out_dev,
                                   double *w_out_dev, int flags)
{
  if (!obs) return fail(CUSMC_EINVAL, "null observation distribution");
  cusmc_ctx *ctx = obs->ctx;
  if (int rc = activate(ctx)) return rc;
  if (kind != CUSMC_MVN && kind != CUSMC_MVT) return fail(CUSMC_EINVAL, "unknown distribution kind %d", kind);
  if (kind == CUSMC_MVT && !(nu > 0.f)) return fail(CUSMC_EINVAL, "nu = %g must be positive", (double)nu);
  if (!G || !Q || !y) return fail(CUSMC_EINVAL, "null G, Q or y");
  if (N == 0) return fail(CUSMC_EINVAL, "N = 0");
  if ((uint64_t)first + count > N) return fail(CUSMC_EINVAL, "shard [%u, %u) exceeds N = %u", first, first + count, N);
  if (count == 0) return CUSMC_OK;
  if (!w_prev_dev || !X_prev_dev || !a_out_dev || !X_out_dev || !w_out_dev)
    return fail(CUSMC_EINVAL, "null device pointer");
  const int d = obs->d;
  if (!pf_step_is_fused(d, count)) {
    if (int rc = cusmc_metropolis_dev(ctx, w_prev_dev, N, B, seed, step, first, count, a_out_dev)) return rc;
    if (int rc = draws(ctx, kind, nu, X_prev_dev, a_out_dev, G, Q, nullptr, d, scale, seed, step, 2u, first, count,
                       X_out_dev))
      return rc;
    return cusmc_dist_reweight_dev(obs, X_out_dev, count, d, y, F, flags, w_out_dev);
  }
  if (int rc = pf_step_prepare(obs, G, Q, y, F)) return rc;
  return pf_step_launch(obs, kind, nu, w_prev_dev, X_prev_dev, N, B, scale, seed, step, first, count, a_out_dev,
                        X_out_dev, w_out_dev, flags, (const double *)obs->shift.p, (const double *)obs->bias.p);
}

// First touch of freshly allocated host pages costs ~0.3 us per 4 KB page on one thread -- 0.2 s for
// the 2.4 GB history of a 1e6-particle, 100-step filter, twenty times the GPU time of the filter
// itself -- and a device-to-host copy into untouched pageable memory pays it serially.  The output
// buffers are write-only for us, so worker threads fault them in (one byte per page) while the GPU
// runs the time loop; the copies then go at PCIe speed.  Plain memory writes only: nothing of
// the caller's runtime (R, Python) is touched from these threads.
namespace {
void prefault_async(std::vector<std::thread> &pool, void *p, size_t bytes, unsigned threads)
{
  if (!p || bytes < (8u << 20) || threads == 0) return;  // small outputs: not worth a thread
  threads = (unsigned)std::min<size_t>(threads, bytes / (4u << 20));  // >= 4 MB per thread
  const size_t page = 4096;
  const size_t pages = (bytes + page - 1) / page, per = (pages + threads - 1) / threads;
  for (unsigned t = 0; t < threads; ++t) {
    const size_t lo = (size_t)t * per, hi = lo + per < pages ? lo + per : pages;
    if (lo >= hi) break;
    pool.emplace_back([=] {
      volatile char *c = static_cast<volatile char *>(p);
      for (size_t g = lo; g < hi; ++g) c[g * page] = 0;
    });
  }
}
}  // namespace

// ---- the filter -----------------------------------------------------------------------------

namespace {

// All of Y is known when a filter starts, so every step's observation vector -- the shift y_t (F = I) or the bias
// W y_t, rotated by Q^T when the general-F plan is (plan_affine) -- goes up in ONE table and the time loop is
// launches only: for a small filter the per-step uploads (two 16..2048-byte copies in the stream) would cost
// more than the launches.  Same values as the per-step plan, bit for bit.  Installs the plan for y_1 (and, for
// the fused step, the [Q | G] image); `host` must stay alive until the upload has run.
// The filter's square roots of C0 and W.  Default: eigenSolver's V sqrt(Lambda), as MCMC() computes them
// (src/mcmc.cpp:70-71, 280; src/linear_algebra.cpp:10-23).  CUSMC_PROPOSAL_FACTOR=cholesky: the lower Cholesky factor
// instead where the matrix is positive definite -- the same law (Q Q^T is what the proposal's covariance is), other
// realisations, and from d = 32 up the proposal's Q xi costs half the matrix-core work (kernels/propagate_mfma*.hip:
// TRIQ); a matrix that is only semi-definite keeps the eigen form.
int filter_factors(const double *C0, const double *W, int d, std::vector<double> &Q0, std::vector<double> &Qw)
{
  bool chol = false;
  if (const char *env = getenv("CUSMC_PROPOSAL_FACTOR")) {
    if (!strcmp(env, "cholesky")) chol = true;
    else if (strcmp(env, "eigen") != 0)
      return fail(CUSMC_EINVAL, "CUSMC_PROPOSAL_FACTOR=\"%s\": expected \"eigen\" or \"cholesky\"", env);
  }
  auto root = [&](const double *S, std::vector<double> &Q) {
    Q.assign((size_t)d * d, 0.0);
    std::vector<double> L;
    if (chol && cusmc::la::cholesky(S, d, L) == 0) Q = L;
    else cusmc::la::eigen_sqrt(S, d, Q.data());
  };
  root(C0, Q0);
  root(W, Qw);
  return CUSMC_OK;
}

struct ObsTable {
  std::vector<double> host;
  DevBuf dev;
  bool centred = true;
  size_t row = 0;
  const double *shift(const cusmc_dist *obs, uint32_t t) const { return centred ? (const double *)dev.p + (size_t)t * row : (const double *)obs->shift.p; }
  const double *bias(const cusmc_dist *obs, uint32_t t) const { return centred ? (const double *)obs->bias.p : (const double *)dev.p + (size_t)t * row; }
};
int pf_obs_table(cusmc_dist *obs, bool fused, const double *G, const double *Qw, const double *Y, uint32_t T, const double *F,
                 ObsTable &tab)
{
  cusmc_ctx *ctx = obs->ctx;
  const int d = obs->d;
  if (int rc = fused ? pf_step_prepare(obs, G, Qw, Y + d, F) : plan_affine(obs, Y + d, F)) return rc;
  tab.centred = obs->plan == 1;
  tab.row = (size_t)((d + 63) / 64) * 64;  // the matrix-core kernels stage 16*NB entries
  tab.host.assign((size_t)T * tab.row, 0.0);
  std::vector<double> b, rb;
  for (uint32_t t = 1; t < T; ++t) {
    const double *y = Y + (size_t)t * d;
    double *dst = tab.host.data() + (size_t)t * tab.row;
    if (tab.centred) {
      std::copy(y, y + d, dst);
    } else {
      cusmc::la::matvec(obs->W.data(), y, d, b);
      if (!obs->plan_Qt.empty()) {
        cusmc::la::matvec(obs->plan_Qt.data(), b.data(), d, rb);
        b.swap(rb);
      }
      std::copy(b.begin(), b.end(), dst);
    }
  }
  if (int rc = tab.dev.reserve((size_t)T * tab.row * 8)) return rc;
  HIP_TRY(hipMemcpyAsync(tab.dev.p, tab.host.data(), (size_t)T * tab.row * 8, hipMemcpyHostToDevice, ctx->stream));
  return CUSMC_OK;
}

// the whole filter on ONE device (the environment is not consulted here)
int pf_run_single(cusmc_ctx *ctx, const double *Y, uint32_t N, int d, uint32_t T, const double *m0, const double *C0,
                  const double *F, const double *G, const double *V, const double *W, float df, const char *resampler,
                  const char *distribution, uint32_t B, double scale, uint64_t seed, double *X_out, double *w_out,
                  uint32_t *a_out)
{
  if (int rc = activate(ctx)) return rc;
  // validate the option strings BEFORE any work: the reference default-constructs an empty
  // std::function for an unknown key and throws bad_function_call mid-run (mcmc.cpp:269-272)
  if (!resampler || strcmp(resampler, "metropolis") != 0)
    return fail(CUSMC_EINVAL, "unknown resampler '%s' (known: metropolis)", resampler ? resampler : "(null)");
  int kind;
  if (distribution && !strcmp(distribution, "mvn")) kind = CUSMC_MVN;
  else if (distribution && !strcmp(distribution, "mvt")) kind = CUSMC_MVT;
  else return fail(CUSMC_EINVAL, "unknown distribution '%s' (known: mvn, mvt)", distribution ? distribution : "(null)");
  if (!Y || !m0 || !C0 || !F || !G || !V || !W) return fail(CUSMC_EINVAL, "null model argument");
  if (N == 0 || T == 0 || d < 1) return fail(CUSMC_EINVAL, "N, T and d must be positive");

  const bool trace = getenv("CUSMC_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = now();
  auto phase = [&](const char *name) {
    if (!trace) return;
    const double t = now();
    fprintf(stderr, "[cusmc_pf_run_host] %-28s %8.2f ms\n", name, (t - t_prev) * 1e3);
    t_prev = t;
  };
  std::vector<double> Q0, Qw;
  if (int rc = filter_factors(C0, W, d, Q0, Qw)) return rc;

  cusmc_dist *obs = nullptr;  // pdf_{0,V}: reweight_G sets mu = 0, sigma = V (mcmc.cpp:188-189)
  if (int rc = cusmc_dist_create(ctx, kind, nullptr, V, d, df, &obs)) return rc;

  const size_t slice = (size_t)N * d;
  DevBuf dX, dw, da, whi2;
  ObsTable ytab;  // (alive until the stream is drained)
  // The history goes back to the host in chunks of whole time steps WHILE the loop runs (the copy
  // engine is idle otherwise and the 2.4 GB of BASELINE configs[2] take four times longer to
  // cross PCIe than to compute): an event after the last step of each chunk, a second stream for
  // the copies.  Small histories are one chunk.
  const size_t step_bytes = slice * 8 + (size_t)N * 12;
  size_t chunk_bytes = 128u << 20;
  if (const char *env = getenv("CUSMC_PF_CHUNK_BYTES")) {  // (test switch: lets a small filter take the chunked path)
    const long long v = atoll(env);
    if (v > 0) chunk_bytes = (size_t)v;
  }
  const uint32_t chunk_steps = (size_t)T * step_bytes <= chunk_bytes ? T : (uint32_t)std::max<size_t>(1, chunk_bytes / step_bytes);
  const uint32_t nchunks = (T + chunk_steps - 1) / chunk_steps;
  std::vector<hipEvent_t> chunk_done(nchunks, nullptr);
  hipStream_t copy_stream = nullptr;
  int rc = dX.reserve(slice * T * 8);
  if (!rc) rc = dw.reserve((size_t)N * T * 8);
  if (!rc) rc = da.reserve((size_t)N * T * 4);
  auto cleanup = [&](int code) {
    (void)hipStreamSynchronize(ctx->stream);
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    for (hipEvent_t ev : chunk_done) if (ev) (void)hipEventDestroy(ev);
    dX.release(); dw.release(); da.release(); whi2.release(); ytab.dev.release();
    cusmc_dist_destroy(obs);
    return code;
  };
  // called when step t has been enqueued: closes a chunk after its last step
  auto mark = [&](uint32_t t) -> int {
    if (nchunks == 1 || ((t + 1) % chunk_steps != 0 && t + 1 != T)) return CUSMC_OK;
    const uint32_t c = t / chunk_steps;
    HIP_TRY(hipEventCreateWithFlags(&chunk_done[c], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(chunk_done[c], ctx->stream));
    return CUSMC_OK;
  };
  if (rc) return cleanup(rc);
  phase("device allocation");
  double *X = (double *)dX.p, *w = (double *)dw.p;
  uint32_t *a = (uint32_t *)da.p;

  // initialize(): x_0 ~ dist(m0).sample(Q0); w_0 = 1/N        src/mcmc.cpp:76-85
  rc = cusmc_initialize_dev(ctx, kind, df, m0, Q0.data(), d, scale, seed, 0, N, X);
  if (rc) return cleanup(rc);
  {
    std::vector<double> w0(N, 1.0 / (double)N);
    if (hipMemcpyAsync(w, w0.data(), (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(a, 0, (size_t)N * 4, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
      return cleanup(fail(CUSMC_EHIP, "initial weight upload failed"));
  }
  rc = mark(0);
  if (rc) return cleanup(rc);
  // MCMC(): for t = 1..T-1: resample -> propagate -> reweight   src/mcmc.cpp:292-308
  if (T > 1) {
    const bool fused = pf_step_is_fused(d, N);
    rc = pf_obs_table(obs, fused, G, Qw.data(), Y, T, F, ytab);
    if (rc) return cleanup(rc);
    // The fused step carries the resampler's truncated weight table along: the kernel that computes w_t stores its
    // high words as well (the one-shard form of the sharded step kernel), so the loop has no 6 us table pre-pass per
    // step.  Two tables in turn; the first from w_0.
    const bool carry_hi = fused && cusmc::metropolis_wants_hiwords(N) && B > 1;
    if (carry_hi) {
      rc = whi2.reserve((size_t)N * 2 * 4);
      if (rc) return cleanup(rc);
      if (hipError_t e = cusmc::launch_hiwords(w, N, (uint32_t *)whi2.p, ctx->num_cus, ctx->stream); e != hipSuccess)
        return cleanup(fail(CUSMC_EHIP, "%s building the weight table", hipGetErrorString(e)));
    }
    for (uint32_t t = 1; t < T; ++t) {
      const double *w_prev = w + (size_t)(t - 1) * N, *X_prev = X + (size_t)(t - 1) * slice;
      uint32_t *a_t = a + (size_t)t * N;
      double *X_t = X + (size_t)t * slice, *w_t = w + (size_t)t * N;
      const double *shift_t = ytab.shift(obs, t), *bias_t = ytab.bias(obs, t);
      if (carry_hi) {
        cusmc::ShardStep st;
        for (int s2 = 0; s2 < cusmc::kMaxShards; ++s2) {
          st.x.base[s2] = nullptr;
          st.x.first[s2] = 0xffffffffu;
          st.w_dst[s2] = nullptr;
          st.whi_dst[s2] = nullptr;
        }
        st.x.first[cusmc::kMaxShards] = 0xffffffffu;
        st.x.n = 1;
        st.x.base[0] = X_prev;
        st.x.first[0] = 0;
        st.w_dst[0] = w_t;  // (the same store as w_out's)
        st.whi_dst[0] = (uint32_t *)whi2.p + (size_t)(t & 1) * N;
        rc = pf_step_launch(obs, kind, df, w_prev, nullptr, N, B, scale, seed, t, 0, N, a_t, X_t, w_t, CUSMC_OUT_DENSITY,
                            shift_t, bias_t, &st, (const uint32_t *)whi2.p + (size_t)((t - 1) & 1) * N);
      } else if (fused) {
        rc = pf_step_launch(obs, kind, df, w_prev, X_prev, N, B, scale, seed, t, 0, N, a_t, X_t, w_t,
                            CUSMC_OUT_DENSITY, shift_t, bias_t);
      } else {
        rc = cusmc_metropolis_dev(ctx, w_prev, N, B, seed, t, 0, N, a_t);
        if (!rc) rc = draws(ctx, kind, df, X_prev, a_t, G, Qw.data(), nullptr, d, scale, seed, t, 2u, 0, N, X_t);
        if (!rc) rc = run_logpdf(obs, X_t, N, d, CUSMC_OUT_DENSITY, w_t, shift_t, bias_t);
      }
      if (!rc) rc = mark(t);
      if (rc) return cleanup(rc);
    }
  }
  phase("enqueue of the time loop");
  // The loop above is only enqueued: fault the output pages in with worker threads while it runs
  // (the copies then go at PCIe speed), chunk by chunk, and copy each chunk out behind its event
  // as soon as its pages are there.
  {
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned threads = getenv("CUSMC_NO_PREFAULT") ? 0 : hw == 0 ? 4 : (hw > 16 ? 16 : hw);  // (switch: for A/B timing)
    hipError_t e = hipSuccess;
    if (nchunks == 1) {
      std::vector<std::thread> pool;
      prefault_async(pool, X_out, slice * T * 8, threads);
      prefault_async(pool, w_out, (size_t)N * T * 8, threads);
      prefault_async(pool, a_out, (size_t)N * T * 4, threads);
      for (auto &th : pool) th.join();
      if (X_out) e = hipMemcpyAsync(X_out, X, slice * T * 8, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && w_out) e = hipMemcpyAsync(w_out, w, (size_t)N * T * 8, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && a_out) e = hipMemcpyAsync(a_out, a, (size_t)N * T * 4, hipMemcpyDeviceToHost, ctx->stream);
    } else {
      // worker k touches its share of every chunk, in chunk order, and counts the chunk done
      std::vector<std::atomic<unsigned>> ready(nchunks);
      for (auto &r : ready) r.store(0, std::memory_order_relaxed);
      auto touch = [](void *p, size_t bytes, unsigned k, unsigned n) {
        if (!p) return;
        const size_t page = 4096, pages = (bytes + page - 1) / page, per = (pages + n - 1) / n;
        const size_t lo = (size_t)k * per, hi = std::min(pages, lo + per);
        volatile char *c = static_cast<volatile char *>(p);
        for (size_t g = lo; g < hi; ++g) c[g * page] = 0;
      };
      std::vector<std::thread> pool;
      for (unsigned k = 0; k < threads; ++k)
        pool.emplace_back([&, k] {
          for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t t0 = c * chunk_steps, t1 = std::min<uint32_t>(T, t0 + chunk_steps);
            touch(X_out ? X_out + (size_t)t0 * slice : nullptr, (size_t)(t1 - t0) * slice * 8, k, threads);
            touch(w_out ? w_out + (size_t)t0 * N : nullptr, (size_t)(t1 - t0) * N * 8, k, threads);
            touch(a_out ? a_out + (size_t)t0 * N : nullptr, (size_t)(t1 - t0) * N * 4, k, threads);
            ready[c].fetch_add(1, std::memory_order_release);
          }
        });
      e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
      for (uint32_t c = 0; c < nchunks && e == hipSuccess; ++c) {
        while (ready[c].load(std::memory_order_acquire) < threads) std::this_thread::yield();
        const uint32_t t0 = c * chunk_steps, t1 = std::min<uint32_t>(T, t0 + chunk_steps);
        const size_t steps = t1 - t0;
        e = hipStreamWaitEvent(copy_stream, chunk_done[c], 0);
        if (e == hipSuccess && X_out)
          e = hipMemcpyAsync(X_out + (size_t)t0 * slice, X + (size_t)t0 * slice, steps * slice * 8, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess && w_out)
          e = hipMemcpyAsync(w_out + (size_t)t0 * N, w + (size_t)t0 * N, steps * N * 8, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess && a_out)
          e = hipMemcpyAsync(a_out + (size_t)t0 * N, a + (size_t)t0 * N, steps * N * 4, hipMemcpyDeviceToHost, copy_stream);
      }
      for (auto &th : pool) th.join();
      if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return cleanup(fail(CUSMC_EHIP, "%s copying filter outputs", hipGetErrorString(e)));
  }
  phase("page faults + device-to-host copies (overlapping the loop)");
  const int done = cleanup(CUSMC_OK);
  phase("device memory released");
  return done;
}

}  // namespace

CUSMC_EXPORT int cusmc_pf_run_host(cusmc_ctx *ctx, const double *Y, uint32_t N, int d, uint32_t T,
                                   const double *m0, const double *C0, const double *F,
                                   const double *G, const double *V, const double *W, float df,
                                   const char *resampler, const char *distribution, uint32_t B,
                                   double scale, uint64_t seed, double *X_out, double *w_out,
                                   uint32_t *a_out)
{
  if (int rc = activate(ctx)) return rc;
  // CUSMC_DEVICES="0,1,2,3": the same call shards the particles over those GPUs, so
  // that callers bound to this entry point -- the R package's run() -- use the node without a new argument;
  // a one-entry list names the device to run on
  std::vector<int> devs;
  if (int rc = env_devices(devs)) return rc;
  if (!devs.empty() && !(devs.size() == 1 && devs[0] == ctx->device))
    return cusmc_pf_run_multi_host(devs.data(), (int)devs.size(), Y, N, d, T, m0, C0, F, G, V, W, df, resampler,
                                   distribution, B, scale, seed, X_out, w_out, a_out);
  return pf_run_single(ctx, Y, N, d, T, m0, C0, F, G, V, W, df, resampler, distribution, B, scale, seed, X_out, w_out,
                       a_out);
}

// ---- the filter on several GPUs ---------------------------------------------------------------------
//
// MCMC()'s time loop (src/mcmc.cpp:292-308) with the particles sharded contiguously over `ndev` devices of
// one node, one host thread and one (library-owned) context per shard -- the exact algorithm, not an island
// filter, and bit for bit the single-device result, because every draw is keyed by the GLOBAL particle index.
// Per step and shard:
//     resample its own chains over the full weight vector w_{t-1}        (cusmc_metropolis_dev)
//     fetch the rows x_{t-1}[a_i] its ancestors name from their owners   (kernels/gather.hip: peer reads)
//     propagate and reweight its own particles                           (the single-device kernels)
//     copy its w_t shard into every shard's copy of w_t                  (hipMemcpyPeerAsync)
//     record the event "step t published"
// Bytes into a device per step: 8 (N - N/R) of weights and at most 8 d N/R of rows (the ancestors that live
// elsewhere) -- never the whole of x_{t-1}.  The reference has no counterpart: its loop runs on one host (zero
// collective call sites).
//
// ORDERING IS ON THE DEVICES (round 3; rounds 1-2 ended every step with hipStreamSynchronize + a barrier of all
// host threads, 20-50 us against ~10 us of compute per GPU at d = 2).  Before step t a shard's stream waits for
// every peer's "step t-1 published" event (hipStreamWaitEvent); the host threads enqueue all T steps without
// blocking on the GPU.  The one host-side handshake left is that an event must have been RECORDED before a peer
// can wait for it: `published[r]` = the last step whose event shard r has recorded, read by its peers in a
// yield loop -- which only ever waits for a peer's ENQUEUE, never for its GPU.  The double-buffered weight vector
// needs nothing more: w_{t+1} goes into the buffer step t read, and a shard writes it at the END of its step
// t + 1, i.e. after it has waited for every peer's step-t event.
namespace {

struct ThreadBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0;
  unsigned gen = 0;
  explicit ThreadBarrier(int n_) : n(n_) {}
  void wait()
  {
    std::unique_lock<std::mutex> lock(m);
    const unsigned g = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return g != gen; });
    }
  }
};

struct FilterShard {
  int device = 0;
  uint32_t first = 0, count = 0;
  cusmc_ctx *ctx = nullptr;
  cusmc_dist *obs = nullptr;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  DevBuf X, w, a, wfull, whifull, anc, ident;  // history [T][count][d], [T][count], [T][count]; 2 x N weights and their high words; count x d; count
  ObsTable ytab;
  std::vector<hipEvent_t> step_done;  // [t]: this shard's step t is complete and its w_t shard is on every device
  std::atomic<uint32_t> published{0}; // last t with step_done[t] recorded (0xffffffff: gave up)
  int rc = CUSMC_OK;
  std::string error;
};

// one byte per page of [p, p + bytes): first touch of the caller's fresh output pages (see prefault_async)
void touch_pages(void *p, size_t bytes)
{
  if (!p || !bytes) return;
  volatile char *c = static_cast<volatile char *>(p);
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  for (uintptr_t g = (lo + 4095) & ~(uintptr_t)4095; g < hi; g += 4096) c[g - lo] = 0;
  c[0] = 0;
}

}  // namespace

CUSMC_EXPORT int cusmc_pf_run_multi_host(const int *devices, int ndev, const double *Y, uint32_t N, int d,
                                         uint32_t T, const double *m0, const double *C0, const double *F,
                                         const double *G, const double *V, const double *W, float df,
                                         const char *resampler, const char *distribution, uint32_t B,
                                         double scale, uint64_t seed, double *X_out, double *w_out,
                                         uint32_t *a_out)
{
  if (!resampler || strcmp(resampler, "metropolis") != 0)
    return fail(CUSMC_EINVAL, "unknown resampler '%s' (known: metropolis)", resampler ? resampler : "(null)");
  int kind;
  if (distribution && !strcmp(distribution, "mvn")) kind = CUSMC_MVN;
  else if (distribution && !strcmp(distribution, "mvt")) kind = CUSMC_MVT;
  else return fail(CUSMC_EINVAL, "unknown distribution '%s' (known: mvn, mvt)", distribution ? distribution : "(null)");
  if (!Y || !m0 || !C0 || !F || !G || !V || !W) return fail(CUSMC_EINVAL, "null model argument");
  if (N == 0 || T == 0 || d < 1) return fail(CUSMC_EINVAL, "N, T and d must be positive");
  if (int rc = check_devices(devices, ndev)) return rc;  // (after the option strings: they are wrong on any box)
  if ((uint32_t)ndev > N) return fail(CUSMC_EINVAL, "%d devices for N = %u particles", ndev, N);

  std::lock_guard<std::mutex> lock(g_pool_mutex);
  DeviceRestore restore;
  if (ndev == 1) {  // one device: the plain loop on that device's library-owned context
    cusmc_ctx *ctx = nullptr;
    if (int rc = pool_ctx(0, devices[0], &ctx)) return rc;
    return pf_run_single(ctx, Y, N, d, T, m0, C0, F, G, V, W, df, resampler, distribution, B, scale, seed, X_out, w_out,
                         a_out);
  }

  // peers must be able to read each other's memory (several shards may also share one device: the
  // rehearsal of this path on a one-GPU box)
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < ndev; ++j) {
      if (devices[i] == devices[j]) continue;
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, devices[i], devices[j]));
      if (!can) return fail(CUSMC_EHIP, "device %d cannot access device %d's memory", devices[i], devices[j]);
      HIP_TRY(hipSetDevice(devices[i]));
      const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(CUSMC_EHIP, "%s enabling peer access %d -> %d", hipGetErrorString(e), devices[i], devices[j]);
      (void)hipGetLastError();
    }

  std::vector<double> Q0, Qw;
  if (int rc = filter_factors(C0, W, d, Q0, Qw)) return rc;

  std::vector<FilterShard> sh(ndev);
  for (int r = 0; r < ndev; ++r) {
    uint64_t first, count;
    shard_range(N, ndev, r, &first, &count);
    sh[r].device = devices[r];
    sh[r].first = (uint32_t)first;
    sh[r].count = (uint32_t)count;
    if (int rc = pool_ctx(r, devices[r], &sh[r].ctx)) return rc;
    sh[r].stream = sh[r].ctx->stream;
  }
  ThreadBarrier barrier(ndev);
  std::atomic<bool> abort_run{false};
  const bool trace = getenv("CUSMC_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  // history copy-out in chunks of whole time steps, as pf_run_single: an event after each chunk's last step
  const size_t step_bytes_all = (size_t)N * d * 8 + (size_t)N * 12;
  size_t chunk_bytes = 128u << 20;
  if (const char *env = getenv("CUSMC_PF_CHUNK_BYTES")) {
    const long long v = atoll(env);
    if (v > 0) chunk_bytes = (size_t)v;
  }
  const uint32_t chunk_steps = (size_t)T * step_bytes_all <= chunk_bytes ? T : (uint32_t)std::max<size_t>(1, chunk_bytes / step_bytes_all);
  const bool fused = cusmc::pf_step_supported(d);  // (every d <= 8: a shard's step is launch-bound long before a device's is)
  // "step t published" must reach the other DEVICES: a system-scope release (the default).  When every shard sits
  // on one device (the rehearsal) agent scope orders them just as well and leaves the device's L2 alone.
  bool one_device = true;
  for (int r = 1; r < ndev; ++r) one_device &= devices[r] == devices[0];
  unsigned event_flags = hipEventDisableTiming;
  if (one_device && !getenv("CUSMC_SYSTEM_FENCE")) event_flags |= hipEventDisableSystemFence;

  auto worker_body = [&](int r, int &passed) {
    FilterShard &me = sh[r];
    const size_t rows = me.count, slice = rows * d;
    auto guard = [&](int rc) {  // records the first failure of this thread and releases the peers
      if (rc && !me.rc) {
        me.rc = rc;
        me.error = g_last_error;
        abort_run.store(true);
        me.published.store(0xffffffffu, std::memory_order_release);
      }
      return rc;
    };
    auto hip = [&](hipError_t e, const char *what) {
      return e == hipSuccess ? CUSMC_OK : fail(CUSMC_EHIP, "%s: %s (device %d)", what, hipGetErrorString(e), me.device);
    };
    // ---- set-up
    int rc = activate(me.ctx);
    if (!rc) rc = hip(hipStreamCreateWithFlags(&me.copy_stream, hipStreamNonBlocking), "stream");
    if (!rc) rc = cusmc_dist_create(me.ctx, kind, nullptr, V, d, df, &me.obs);
    if (!rc) rc = me.X.reserve(slice * T * 8);
    if (!rc) rc = me.w.reserve(rows * T * 8);
    if (!rc) rc = me.a.reserve(rows * T * 4);
    if (!rc) rc = me.wfull.reserve((size_t)N * 2 * 8);
    if (!rc) rc = me.whifull.reserve((size_t)N * 2 * 4);
    if (!rc) rc = me.anc.reserve(slice * 8);
    if (!rc) rc = me.ident.reserve(rows * 4);
    if (!rc) {
      me.step_done.assign(T, nullptr);
      for (uint32_t t = 0; t < T && !rc; ++t)
        rc = hip(hipEventCreateWithFlags(&me.step_done[t], event_flags), "event");
    }
    double *X = (double *)me.X.p, *w = (double *)me.w.p, *wfull = (double *)me.wfull.p;
    uint32_t *a = (uint32_t *)me.a.p;
    if (!rc) {
      std::vector<uint32_t> id(rows);
      for (size_t i = 0; i < rows; ++i) id[i] = (uint32_t)i;
      const std::vector<double> w0(N, 1.0 / (double)N);  // initialize(): w_0 = 1/N  (src/mcmc.cpp:85)
      rc = hip(hipMemcpyAsync(me.ident.p, id.data(), rows * 4, hipMemcpyHostToDevice, me.stream), "upload");
      if (!rc) rc = hip(hipMemcpyAsync(wfull, w0.data(), (size_t)N * 8, hipMemcpyHostToDevice, me.stream), "upload");
      if (!rc) rc = hip(hipMemcpyAsync(w, w0.data(), rows * 8, hipMemcpyHostToDevice, me.stream), "upload");
      if (!rc) rc = hip(hipMemsetAsync(a, 0, rows * 4, me.stream), "memset");
      if (!rc) rc = hip(cusmc::launch_hiwords(wfull, N, (uint32_t *)me.whifull.p, me.ctx->num_cus, me.stream), "high words");
      if (!rc) rc = cusmc_initialize_dev(me.ctx, kind, df, m0, Q0.data(), d, scale, seed, me.first, me.count, X);
      if (!rc && T > 1) rc = pf_obs_table(me.obs, fused, G, Qw.data(), Y, T, F, me.ytab);
      if (!rc) rc = hip(hipStreamSynchronize(me.stream), "initial state");
    }
    guard(rc);
    barrier.wait();  // every shard's buffers exist and hold step 0 (the only step-0 ordering: once per run)
    passed = 1;
    // ---- MCMC(): for t = 1..T-1: resample -> (fetch ancestors) -> propagate -> reweight, enqueued without blocking
    for (uint32_t t = 1; t < T && !abort_run.load(std::memory_order_relaxed); ++t) {
      rc = CUSMC_OK;
      if (t > 1) {
        for (int s2 = 0; s2 < ndev && !rc; ++s2) {
          if (s2 == r) continue;
          while (sh[s2].published.load(std::memory_order_acquire) < t - 1 && !abort_run.load(std::memory_order_relaxed))
            std::this_thread::yield();
          if (abort_run.load(std::memory_order_relaxed)) break;
          rc = hip(hipStreamWaitEvent(me.stream, sh[s2].step_done[t - 1], 0), "wait for a peer's step");
        }
        if (abort_run.load(std::memory_order_relaxed)) break;
      }
      // several shards on ONE device (a rehearsal, or more shards than GPUs): in rank order, not side by side -- two
      // step kernels from two queues of one device take 110 us together where they take 2 x 32 us one after
      // the other (N = 1e6, d = 2; profiles/r03_multi_filter.md)
      for (int s2 = 0; s2 < r && !rc; ++s2) {
        if (sh[s2].device != me.device) continue;
        while (sh[s2].published.load(std::memory_order_acquire) < t && !abort_run.load(std::memory_order_relaxed))
          std::this_thread::yield();
        if (abort_run.load(std::memory_order_relaxed)) break;
        rc = hip(hipStreamWaitEvent(me.stream, sh[s2].step_done[t], 0), "wait for a co-located shard");
      }
      if (abort_run.load(std::memory_order_relaxed)) break;
      const double *w_prev = wfull + (size_t)((t - 1) & 1) * N;
      uint32_t *a_t = a + (size_t)t * rows;
      double *X_t = X + (size_t)t * slice, *w_t = w + (size_t)t * rows;
      const double *shift_t = me.ytab.shift(me.obs, t), *bias_t = me.ytab.bias(me.obs, t);
      cusmc::ShardStep st;
      st.x.n = ndev;
      for (int s2 = ndev; s2 < cusmc::kMaxShards; ++s2) {  // (pf_step_kernel's select chain runs over every entry)
        st.x.base[s2] = nullptr;
        st.x.first[s2] = 0xffffffffu;
        st.w_dst[s2] = nullptr;
        st.whi_dst[s2] = nullptr;
      }
      st.x.first[cusmc::kMaxShards] = 0xffffffffu;
      for (int s2 = 0; s2 < ndev; ++s2) {
        st.x.base[s2] = (const double *)sh[s2].X.p + (size_t)(t - 1) * sh[s2].count * d;
        st.x.first[s2] = sh[s2].first;
        st.w_dst[s2] = (double *)sh[s2].wfull.p + (size_t)(t & 1) * N;
        st.whi_dst[s2] = (uint32_t *)sh[s2].whifull.p + (size_t)(t & 1) * N;
      }
      st.x.first[ndev] = N;
      if (fused) {
        // d <= 8: ONE launch per shard and step -- chain, the ancestor's row from its owner, proposal, weight, and the
        // weight (+ its high word) stored into every shard's copy of w_t
        if (!rc)
          rc = pf_step_launch(me.obs, kind, df, w_prev, nullptr, N, B, scale, seed, t, me.first, me.count, a_t, X_t, w_t,
                              CUSMC_OUT_DENSITY, shift_t, bias_t, &st, (const uint32_t *)me.whifull.p + (size_t)((t - 1) & 1) * N);
      } else {
        if (!rc) rc = cusmc_metropolis_dev(me.ctx, w_prev, N, B, seed, t, me.first, me.count, a_t);
        if (!rc)
          rc = hip(cusmc::launch_gather_rows_sharded(st.x, a_t, me.count, d, (double *)me.anc.p, me.ctx->num_cus, me.stream),
                   "ancestor gather");
        if (!rc)
          rc = draws(me.ctx, kind, df, (const double *)me.anc.p, (const uint32_t *)me.ident.p, G, Qw.data(), nullptr, d, scale,
                     seed, t, 2u, me.first, me.count, X_t);
        if (!rc) rc = run_logpdf(me.obs, X_t, me.count, d, CUSMC_OUT_DENSITY, w_t, shift_t, bias_t);
        for (int s2 = 0; s2 < ndev && !rc; ++s2) {
          double *dst = st.w_dst[s2] + me.first;
          rc = hip(sh[s2].device == me.device
                       ? hipMemcpyAsync(dst, w_t, rows * 8, hipMemcpyDeviceToDevice, me.stream)
                       : hipMemcpyPeerAsync(dst, sh[s2].device, w_t, me.device, rows * 8, me.stream),
                   "weight exchange");
        }
      }
      if (!rc) rc = hip(hipEventRecord(me.step_done[t], me.stream), "step event");
      if (guard(rc)) break;
      me.published.store(t, std::memory_order_release);
    }
    if (trace && r == 0) fprintf(stderr, "[cusmc_pf_run_multi_host] time loop enqueued        %8.2f ms\n", (now() - t_start) * 1e3);
    // ---- this shard's columns of the history, in the reference's packing (src/run.rcpp.cpp:110-125): chunk by
    // chunk behind the chunk's last step on a second stream, the caller's fresh pages touched by this thread
    // first (the GPU is still busy with the loop), one contiguous copy per step and array
    if (!abort_run.load()) {
      rc = CUSMC_OK;
      for (uint32_t t0 = 0; t0 < T && !rc; t0 += chunk_steps) {
        const uint32_t t1 = std::min<uint32_t>(T, t0 + chunk_steps);
        for (uint32_t t = t0; t < t1; ++t) {
          if (X_out) touch_pages(X_out + ((size_t)t * N + me.first) * d, slice * 8);
          if (w_out) touch_pages(w_out + (size_t)t * N + me.first, rows * 8);
          if (a_out) touch_pages(a_out + (size_t)t * N + me.first, rows * 4);
        }
        if (t1 - 1 >= 1) rc = hip(hipStreamWaitEvent(me.copy_stream, me.step_done[t1 - 1], 0), "history copy");
        for (uint32_t t = t0; t < t1 && !rc; ++t) {
          if (X_out)
            rc = hip(hipMemcpyAsync(X_out + ((size_t)t * N + me.first) * d, X + (size_t)t * slice, slice * 8, hipMemcpyDeviceToHost, me.copy_stream), "history copy");
          if (!rc && w_out)
            rc = hip(hipMemcpyAsync(w_out + (size_t)t * N + me.first, w + (size_t)t * rows, rows * 8, hipMemcpyDeviceToHost, me.copy_stream), "history copy");
          if (!rc && a_out)
            rc = hip(hipMemcpyAsync(a_out + (size_t)t * N + me.first, a + (size_t)t * rows, rows * 4, hipMemcpyDeviceToHost, me.copy_stream), "history copy");
        }
      }
      if (!rc) rc = hip(hipStreamSynchronize(me.copy_stream), "history copy");
      guard(rc);
    }
    (void)hipStreamSynchronize(me.stream);
    barrier.wait();  // nobody frees a buffer a peer may still be reading
    passed = 2;
    me.X.release(); me.w.release(); me.a.release(); me.wfull.release(); me.whifull.release(); me.anc.release(); me.ident.release();
    me.ytab.dev.release();
    for (hipEvent_t ev : me.step_done) if (ev) (void)hipEventDestroy(ev);
    me.step_done.clear();
    if (me.obs) cusmc_dist_destroy(me.obs);
    me.obs = nullptr;
    if (me.copy_stream) (void)hipStreamDestroy(me.copy_stream);
    me.copy_stream = nullptr;
  };
  // nothing thrown inside a shard leaves it (a bad_alloc in a std::thread would be std::terminate), and a shard
  // that dies still meets the barriers it owes, so that its peers are not left waiting
  std::atomic<int> go{0};  // 0: hold, 1: run, -1: a thread could not be started, nobody runs
  auto worker = [&](int r) {
    while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
    if (go.load(std::memory_order_acquire) < 0) return;
    int passed = 0;
    try {
      worker_body(r, passed);
    } catch (...) {
      FilterShard &me = sh[r];
      if (!me.rc) {
        me.rc = CUSMC_EINVAL;
        me.error = "exception in a filter shard (out of host memory?)";
      }
      abort_run.store(true);
      me.published.store(0xffffffffu, std::memory_order_release);
      for (; passed < 2; ++passed) barrier.wait();
    }
  };
  std::vector<std::thread> threads;
  try {
    for (int r = 1; r < ndev; ++r) threads.emplace_back(worker, r);
  } catch (...) {
    go.store(-1, std::memory_order_release);
    for (auto &th : threads) th.join();
    return fail(CUSMC_EINVAL, "could not start %d host threads", ndev - 1);
  }
  go.store(1, std::memory_order_release);
  worker(0);
  for (auto &th : threads) th.join();
  if (trace) fprintf(stderr, "[cusmc_pf_run_multi_host] done                        %8.2f ms\n", (now() - t_start) * 1e3);
  for (int r = 0; r < ndev; ++r)
    if (sh[r].rc) {
      g_last_error = sh[r].error;
      return sh[r].rc;
    }
  return CUSMC_OK;
}
