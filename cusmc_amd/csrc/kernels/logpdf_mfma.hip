// Launchers and host-side fragment packing for kernels/logpdf_mfma_kernel.h (the kernel itself
// and the design notes live there).
#include <atomic>
#include <cstddef>
#include <cstdlib>

#include "logpdf_mfma_kernel.h"

namespace cusmc {

int mfma_num_frags(int nb, bool tri) { return tri ? 4 * nb * (nb + 1) / 2 : 4 * nb * nb; }

// d = 16, 32, ..., 176 with 16-byte aligned rows take the plain kernel; every other d in (16, 176]
// and every other alignment the padded variant (PAD).  d < 16 stays with the generic kernel: a
// single, mostly empty k-block would cost more loads than it saves.  A tile's per-lane byte offset
// is kept in 32 bits.
bool mfma_supported(int d, const void *X, int64_t ldx)
{
  return d >= 16 && d <= kTileKernelMaxDim && ldx < (1L << 24);
}
static bool mfma_needs_pad(int d, const void *X, int64_t ldx)
{
  return d % 16 != 0 || (uintptr_t)X % 16 != 0 || ldx % 2 != 0;
}

// Fragment f (kernel loop order: kb, then s, then cb) holds, for lane l = (j, h):
//   M[16*cb + j][16*kb + pi(s,h)]
void mfma_pack_frags(const double *M, int d, bool tri, double *frags)
{
  const int nb = d / 16;
  int f = 0;
  for (int kb = 0; kb < nb; ++kb)
    for (int s = 0; s < 4; ++s)
      for (int cb = tri ? kb : 0; cb < nb; ++cb, ++f)
        for (int l = 0; l < 64; ++l) {
          const int j = l & 15, h = l >> 4;
          frags[(size_t)f * 64 + l] = M[(size_t)(16 * cb + j) * d + 16 * kb + pi_k(s, h)];
        }
}

template <int NB, bool CENTRED, bool SHIFT, int EPI, bool PAD>
static hipError_t launch_nb(const double *X, int64_t N, int64_t ldx, int d, const double *frags,
                            const double *shift, const double *bias, const Epilogue &ep,
                            double *out, int num_cus, hipStream_t stream)
{
  constexpr int NFRAG = 4 * NB * (NB + 1) / 2;  // lower triangular in both forms
  constexpr int THREADS = mfma_threads<NB>();
  const size_t lds_bytes = (size_t)(32 * NB + 4 + NFRAG * 64) * sizeof(double);  // (the factor passes through LDS in every variant)
  const long num_tiles = (N + 15) / 16;
  auto kern = logpdf_mfma_kernel<NB, CENTRED, SHIFT, 0, EPI, PAD>;
  static LdsConfig lds_configured;
  if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_configured); e != hipSuccess) return e;
  // one persistent workgroup per CU (its waves share the round counter), fewer when there is
  // less work than that
  long blocks = num_cus;
  if (blocks > (num_tiles + 15) / 16) blocks = (num_tiles + 15) / 16;  // >= 16 tiles each
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds_bytes, stream, X, (long)N,
                     (long)ldx, frags, shift, bias, ep, out, num_tiles, d);
  return hipGetLastError();
}

// ---- the hand-written assembly kernel (kernels/logpdf_nb4_gfx950.s) ------------------------------------------------
// d = 64 (NB = 4), centred, no shift, MVN log-density, 16-byte aligned rows: the headline shape (BASELINE.json: 1e6 x 64).
// A code object of its own, embedded here as bytes and loaded once per device.
namespace {
const unsigned char kNb4CodeObject[] = {
#include "logpdf_nb4_co.inc"
};
const unsigned char kNb4CodeObjectB0[] = {  // assembled with BIRTH = 0 (A/B runs: CUSMC_NB4_ASM=2)
#include "logpdf_nb4_b0_co.inc"
};
struct AsmKernel {
  std::atomic<int> state{0};  // 0 not tried, 1 ready, 2 unavailable
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};
AsmKernel g_nb4[64];
std::atomic_flag g_nb4_lock = ATOMIC_FLAG_INIT;
int nb4_mode()
{
  static const int mode = [] { const char *e = getenv("CUSMC_NB4_ASM"); return e ? atoi(e) : 1; }();  // 0: compiled kernel, 2: BIRTH = 0
  return mode;
}

hipFunction_t nb4_function()
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  AsmKernel &k = g_nb4[dev];
  int st = k.state.load(std::memory_order_acquire);
  if (st == 0) {
    while (g_nb4_lock.test_and_set(std::memory_order_acquire)) {}
    st = k.state.load(std::memory_order_acquire);
    if (st == 0) {
      st = 2;
      if (hipModuleLoadData(&k.mod, nb4_mode() == 2 ? kNb4CodeObjectB0 : kNb4CodeObject) == hipSuccess &&
          hipModuleGetFunction(&k.fn, k.mod, "cusmc_logpdf_nb4_asm") == hipSuccess)
        st = 1;
      (void)hipGetLastError();
      k.state.store(st, std::memory_order_release);
    }
    g_nb4_lock.clear(std::memory_order_release);
  }
  return st == 1 ? k.fn : nullptr;
}

// the kernel's explicit arguments, at the offsets its s_load instructions use
struct Nb4Args {
  const double *X;          // 0x00
  long N;                   // 0x08
  long ldx;                 // 0x10
  const double *frags;      // 0x18
  unsigned *pool;           // 0x20  this launch's tail-pool counters (kNb4PoolCounters x 128 bytes, zero)
  unsigned rounds_dealt;    // 0x28  rounds of the round-robin deal; the tiles from rounds_dealt * G on are the pool
  unsigned stamps;          // 0x2c  diagnostic: 1, 2 = record per-wave (entry, exit) s_memrealtime pairs in area 0, 1
  double lognorm;           // 0x30
  unsigned *pool_other;     // 0x38  the other launch parity's counters: zeroed by this launch for the next one
  char *stamp_records;      // 0x40
  long unused;              // 0x48
  double *out;              // 0x50
  long num_tiles;           // 0x58
  int d;                    // 0x60
};
static_assert(offsetof(Nb4Args, out) == 0x50 && offsetof(Nb4Args, d) == 0x60, "kernarg layout of logpdf_nb4_gfx950.s");
}  // namespace

size_t nb4_pool_bytes() { return 8192 + 2 * 32768; }  // two counter blocks of 32 x 128 bytes | 8192: two stamp areas of 256 workgroups x 8 waves x 16 bytes

// true: launched.  false: not this kernel's shape (or the code object is unavailable): the caller takes the compiled one.
static bool launch_nb4_asm(const double *X, int64_t N, int64_t ldx, const double *frags, const Epilogue &ep, double *out,
                           int num_cus, Nb4Pool *pool, hipStream_t stream, hipError_t *err)
{
  if (nb4_mode() == 0 || !pool || !pool->dev) return false;
  const long num_tiles = (N + 15) / 16;
  const long blocks = num_cus;
  static const int min_rounds = [] { const char *e = getenv("CUSMC_NB4_MIN_ROUNDS"); return e ? atoi(e) : 16; }();
  // (a tail pool needs a body -- and the first tile by birth needs 8 rounds --, and every one of the 32 counters a wave:
  // b / 8 + 4 w covers them from 64 workgroups up)
  if (blocks < 64 || num_tiles < (long)(min_rounds < 16 ? 16 : min_rounds) * blocks || num_tiles >= (1L << 31)) return false;
  hipFunction_t fn = nb4_function();
  if (!fn) return false;
  static const int pool_rounds = [] { const char *e = getenv("CUSMC_NB4_POOL_ROUNDS"); return e ? atoi(e) : 48; }();
  const long rounds = (num_tiles + blocks - 1) / blocks;
  Nb4Args a{};
  // two counter blocks, alternating from launch to launch on this stream (the kernel zeroes the one it does not use)
  const unsigned parity = pool->launches++ & 1u;
  char *base = reinterpret_cast<char *>(pool->dev);
  a.X = X; a.N = N; a.ldx = ldx; a.frags = frags;
  a.pool = reinterpret_cast<unsigned *>(base + parity * 4096);
  a.pool_other = reinterpret_cast<unsigned *>(base + (parity ^ 1u) * 4096);
  a.stamp_records = base + 8192;
  const long pooled = pool_rounds < rounds / 3 ? pool_rounds : rounds / 3;  // (at least two thirds of the tiles are dealt)
  a.rounds_dealt = (unsigned)(rounds - pooled);
  static const bool stamps = [] { const char *e = getenv("CUSMC_NB4_STAMPS"); return e && e[0] && e[0] != '0'; }();
  static std::atomic<unsigned> launches{0};
  a.stamps = stamps && blocks <= 256 ? 1u + (launches.fetch_add(1) & 1u) : 0u;
  a.lognorm = ep.lognorm; a.out = out; a.num_tiles = num_tiles; a.d = 64;
  size_t size = sizeof a;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const size_t lds_bytes = (size_t)(32 * 4 + 4 + 40 * 64) * sizeof(double);
  *err = hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 512, 1, 1, (unsigned)lds_bytes, stream, nullptr, config);
  return true;
}

// frags: mfma_pack_frags(., ., tri = true, .) of the lower triangular factor.  centred: z = L (x -
// shift); otherwise z = bias + L x (the QL-rotated affine form, cusmc_abi.hip plan_affine()).
// pool: the stream's tail-pool counters for the assembly kernel (nb4_pool_bytes() of device memory, zeroed once) and
// its launch count, or NULL.
hipError_t launch_logpdf_mfma(const double *X, int64_t N, int64_t ldx, int d, bool centred,
                              bool has_shift, const double *frags, const double *shift,
                              const double *bias, const Epilogue &ep, double *out, int num_cus,
                              hipStream_t stream, Nb4Pool *pool)
{
  if (N <= 0) return hipSuccess;
  const int epi = ep.out_density ? 0 : ep.kind == CUSMC_MVN ? 1 : 2;
  const bool pad = mfma_needs_pad(d, X, ldx);
  if (d == 64 && centred && !has_shift && epi == 1 && !pad) {
    hipError_t e = hipSuccess;
    if (launch_nb4_asm(X, N, ldx, frags, ep, out, num_cus, pool, stream, &e)) return e;
  }
#define CUSMC_PADV(nb, t, s, l)                                                                   \
  (pad ? launch_nb<nb, t, s, l, true>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream) \
       : launch_nb<nb, t, s, l, false>(X, N, ldx, d, frags, shift, bias, ep, out, num_cus, stream))
#define CUSMC_EPI(nb, t, s) (epi == 1 ? CUSMC_PADV(nb, t, s, 1) : epi == 2 ? CUSMC_PADV(nb, t, s, 2) : CUSMC_PADV(nb, t, s, 0))
#define CUSMC_CASE(nb)                                                                            \
  case nb:                                                                                        \
    if (!centred) return CUSMC_EPI(nb, false, false);                                               \
    return has_shift ? CUSMC_EPI(nb, true, true) : CUSMC_EPI(nb, true, false);
  switch ((d + 15) / 16) {
    CUSMC_CASE(1)
    CUSMC_CASE(2)
    CUSMC_CASE(3)
    CUSMC_CASE(4)
    CUSMC_CASE(5)
    CUSMC_CASE(6)
    CUSMC_CASE(7)
    CUSMC_CASE(8)
#if CUSMC_TILE_MAX_NB >= 9
    CUSMC_CASE(9)
#endif
#if CUSMC_TILE_MAX_NB >= 10
    CUSMC_CASE(10)
#endif
#if CUSMC_TILE_MAX_NB >= 11
    CUSMC_CASE(11)
#endif
#if CUSMC_TILE_MAX_NB >= 12
    CUSMC_CASE(12)
#endif
  }
#undef CUSMC_CASE
#undef CUSMC_EPI
#undef CUSMC_PADV
  return hipErrorInvalidValue;
}

}  // namespace cusmc
