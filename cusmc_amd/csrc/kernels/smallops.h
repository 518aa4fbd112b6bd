// Per-lane building blocks shared by the lane = particle kernels (resample.hip, propagate.hip,
// logpdf_generic.hip) and by the fused filter step (pf_step.hip), so that the fused and the
// unfused paths execute the same arithmetic in the same order.  Contract: philox.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../launch.h"
#include "../philox.h"
#include "../../../include/cusmc_hip.h"

namespace cusmc {

// A polynomial coefficient as a SCALAR operand.  Left alone, hipcc evaluates Horner steps as
// v_fmac_f64 (dst += a * b), which needs the coefficient copied into dst first -- one v_mov_b64 of
// VALU issue per step, ~30 per Box-Muller pair.  An f64 literal cannot be an inline operand, but an
// SGPR pair can: through this no-op the coefficient lives in SGPRs and each step is one v_fma_f64.
// Same operations on the same values: results are unchanged bit for bit.
#ifndef CUSMC_NO_SCALAR_COEFFS
static __device__ __forceinline__ double sc(double c)
{
  asm volatile("" : "+s"(c));  // (volatile: materialised at the point of use -- hoisted out of the loops, two dozen
                                  // coefficients exceed the SGPR file and come back through v_readlane)
  return c;
}
#else
static __device__ __forceinline__ double sc(double c) { return c; }
#endif

// ln(x) for finite x > 0, < 1 ulp: the classic reduction x = 2^e m, m in [sqrt(1/2), sqrt(2)),
// ln(m) = 2 atanh(s), s = (m - 1) / (m + 1), with the degree-7 minimax polynomial in s^2 and the
// hi/lo split of ln 2 of Sun's fdlibm e_log.c (constants from there).  The library log (ocml) costs
// several hundred instructions here because it also serves denormals, infinities and negative
// arguments; a uniform variate needs none of that, and every draw pays for it.
static __device__ __forceinline__ double ln_pos(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double f = m - 1.0;
  // s = f / (2 + f): reciprocal seed + two Newton steps + one residual correction (< 1 ulp)
  const double den = 2.0 + f;
  double r = __builtin_amdgcn_rcp(den);
  r = r * fma(-den, r, 2.0);
  r = r * fma(-den, r, 2.0);
  double sq = f * r;
  sq = fma(fma(-den, sq, f), r, sq);
  const double z = sq * sq, w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
  return dk * ln2_hi - ((hfsq - (sq * (hfsq + R) + dk * ln2_lo)) - f);
}

// cos(2 pi u), sin(2 pi u) for u in [0, 1): exact reduction to a quadrant (2u = q/2 + r,
// |r| <= 1/4), then fdlibm's k_sin.c / k_cos.c kernels on x = pi r, |x| <= pi/4 (constants from
// there).  No Payne-Hanek path, no double-double: < 1.5 ulp on both.
static __device__ __forceinline__ void sincos_2pi(double u, double &c, double &s)
{
  const double t = 2.0 * u;                // [0, 2), exact
  const double q = __builtin_rint(2.0 * t);  // 0..4
  const double r = fma(-0.5, q, t);        // exact, |r| <= 1/4
  const double pi_hi = 3.14159265358979311600e+00, pi_lo = 1.22464679914735317723e-16;
  const double x = fma(r, pi_hi, r * pi_lo);
  const double z = x * x;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double v = z * x;
  const double ps = fma(z, fma(z, fma(z, fma(z, sc(S6), sc(S5)), sc(S4)), sc(S3)), sc(S2));
  const double sx = fma(v, fma(z, ps, sc(S1)), x);
  const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, sc(C6), sc(C5)), sc(C4)), sc(C3)), sc(C2)), sc(C1));
  const double hz = 0.5 * z;
  const double w1 = 1.0 - hz;
  const double cx = w1 + (((1.0 - w1) - hz) + z * pc);
  const int qi = (int)q & 3;
  const double a = (qi & 1) ? sx : cx;   // |cos| takes sx in odd quadrants
  const double b = (qi & 1) ? cx : sx;
  c = (qi == 1 || qi == 2) ? -a : a;
  s = (qi >= 2) ? -b : b;
}

// Box-Muller on one Philox block: u1 in (0,1], u2 in [0,1):
//     z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2)
// (oracle/cusmc_oracle.c evaluates the same expressions with libm; the two agree to a few ulp).
static __device__ __forceinline__ void normal_pair(const u32x4 r, double &z0, double &z1)
{
#ifdef CUSMC_ABL_NO_BOXMULLER  // ablation builds only (scripts/calib/prop_time.py): what ln, sqrt and sincos cost a kernel
  z0 = (double)r.x, z1 = (double)r.z;
  return;
#endif
  const double u1 = 1.0 - u01_53(r.x, r.y);  // (0,1]
  const double u2 = u01_53(r.z, r.w);        // [0,1)
  const double rad = sqrt(-2.0 * ln_pos(u1));
  double c, s;
  sincos_2pi(u2, c, s);
  z0 = rad * c;
  z1 = rad * s;
}

// chi^2_nu = 2 Gamma(nu/2, 1) by Marsaglia-Tsang, the reference's device sampler (curand_gamma,
// src/mvt_dist.cu.cpp:20-61) including its squeeze test (:45); counter layout as
// oracle/cusmc_oracle.c:chi_square_for -- attempt m < 63 of (particle, component j) takes its normal from
// Philox block (particle, 64 j + m, step, 3) and its uniform from block (particle, 64 j + m, step, 5); the
// a < 1 boost uniform is block (particle, 64 j + 63, step, 5).  Accept attempt m iff
//     v = 1 + c z0 > 0   and   ( u < 1 - 0.0331 z0^4   or   ln u < z0^2/2 + dd - dd v^3 + dd ln v^3 ).
//
// Cost model (f64 MFMA kernels: every VALU instruction is on the critical path; elsewhere: the proposal
// kernels are RNG-bound).  One attempt is two Philox blocks and a Box-Muller pair; the log test adds two
// ln.  The squeeze settles ~92 % of the attempts without a logarithm, ~96 % of the attempts accept -- but a
// wave runs a branch as long as ONE of its 64 lanes needs it, so a per-draw loop executes its slow path on
// nearly every draw (1 - 0.92^64) and its second attempt on most (1 - 0.96^64): ~2 x (attempt + 2 ln) per
// draw, which is what round 1's version cost (29 x the Normal draw in the d = 2 filter).  chi_square_batch
// therefore takes the K draws of a lane TOGETHER: first attempt and squeeze for all K with no branch, then
// the few draws still open (8 % per lane: the wave makes max-over-lanes trips, ~4 of 16) go through the full
// loop one per trip.  Same counters, same accept rule, same results as the draw-by-draw loop.
struct ChiSquare {
  double dd, c, inv_a;
  bool boost;
};
static __device__ __forceinline__ ChiSquare chi_setup(float nu)
{
  ChiSquare cs;
  double a = 0.5 * (double)nu;
  cs.boost = a < 1.0;
  cs.inv_a = 1.0 / a;
  if (cs.boost) a += 1.0;
  cs.dd = a - 1.0 / 3.0;
  cs.c = 1.0 / sqrt(9.0 * cs.dd);
  return cs;
}
// attempt (particle, jm = 64 j + m): its normal, v = 1 + c z0, and its uniform in (0, 1]
static __device__ __forceinline__ void chi_attempt(const ChiSquare &cs, uint32_t particle, uint32_t jm, uint32_t step,
                                                   uint32_t k0, uint32_t k1, double &z0, double &v, double &u)
{
  double z1;
  normal_pair(philox4x32_10(particle, jm, step, 3u, k0, k1), z0, z1);
  v = 1.0 + cs.c * z0;
  const u32x4 r = philox4x32_10(particle, jm, step, 5u, k0, k1);
  u = 1.0 - u01_53(r.x, r.y);
}
// u < 1 - 0.0331 z0^4, as ONE fixed sequence of roundings (the oracle evaluates the same fma)
static __device__ __forceinline__ bool chi_squeeze(double z0, double u)
{
  const double z2 = z0 * z0;
  return u < fma(-(0.0331 * z2), z2, 1.0);
}
// the draw-by-draw loop: every attempt from m = 0 with the full rule; returns Gamma(a, 1) / boost
static __device__ __forceinline__ double chi_loop(const ChiSquare &cs, uint32_t particle, uint32_t j, uint32_t step,
                                                  uint32_t k0, uint32_t k1)
{
  double g = cs.dd;  // value if all 63 attempts reject (probability < 1e-60)
  for (uint32_t m = 0; m < 63u; ++m) {
    double z0, v, u;
    chi_attempt(cs, particle, j * 64u + m, step, k0, k1, z0, v, u);
    if (v <= 0.0) continue;
    v = v * v * v;
    if (chi_squeeze(z0, u) || ln_pos(u) < 0.5 * z0 * z0 + cs.dd - cs.dd * v + cs.dd * ln_pos(v)) {
      g = cs.dd * v;
      break;
    }
  }
  return g;
}
static __device__ __forceinline__ double chi_boost(const ChiSquare &cs, uint32_t particle, uint32_t j, uint32_t step,
                                                   uint32_t k0, uint32_t k1)
{
  const u32x4 r = philox4x32_10(particle, j * 64u + 63u, step, 5u, k0, k1);
  return pow(1.0 - u01_53(r.x, r.y), cs.inv_a);
}
// chi[c] = chi^2_nu draw of component jof(c) for the c < K with live(c); jof / live are evaluated for
// run-time c as well (closed forms, not tables: no dynamically indexed registers).
template <int K, typename JOf, typename Live>
static __device__ __forceinline__ void chi_square_batch(const ChiSquare &cs, uint32_t particle, uint32_t step, uint32_t k0,
                                                        uint32_t k1, JOf jof, Live live, double (&chi)[K])
{
  static_assert(K <= 32, "one pending bit per draw");
  uint32_t pend = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) chi[c] = 1.0;
  // A real loop, not K unrolled copies: unrolled, the K independent attempts are interleaved by the
  // scheduler and their ~50 live registers each add up (100-200 VGPRs spilled in the matrix-core proposal);
  // the price is the select chain that files the result under a run-time c.
#pragma unroll 1
  for (int c = 0; c < K; ++c) {
    if (!live(c)) continue;
    double z0, v, u;
    chi_attempt(cs, particle, (uint32_t)jof(c) * 64u, step, k0, k1, z0, v, u);
    const bool ok = (v > 0.0) & chi_squeeze(z0, u);
    const double g = cs.dd * (v * v * v);
    pend |= ok ? 0u : 1u << c;
#pragma unroll
    for (int cc = 0; cc < K; ++cc) chi[cc] = cc == c ? g : chi[cc];
  }
  while (pend) {  // one open draw per lane per trip
    const int c = __builtin_ctz(pend);
    pend &= pend - 1u;
    const double g = chi_loop(cs, particle, (uint32_t)jof(c), step, k0, k1);
#pragma unroll
    for (int cc = 0; cc < K; ++cc) chi[cc] = cc == c ? g : chi[cc];
  }
#pragma unroll
  for (int c = 0; c < K; ++c) {
    chi[c] *= 2.0;
    if (cs.boost) {  // (wave-uniform: nu is a launch parameter)
      if (live(c)) chi[c] *= chi_boost(cs, particle, (uint32_t)jof(c), step, k0, k1);
    }
  }
}
// a single draw (callers that own one component per lane)
static __device__ __forceinline__ double chi_square_for(uint32_t particle, uint32_t j, uint32_t step, uint32_t k0,
                                                        uint32_t k1, float nu)
{
  const ChiSquare cs = chi_setup(nu);
  double chi[1];
  chi_square_batch<1>(cs, particle, step, k0, k1, [&](int) { return j; }, [](int) { return true; }, chi);
  return chi[0];
}

// One Metropolis chain (Sampler::metropolis_hastings, src/samplers.cpp:21-35):
//     k = i;  B times { u ~ U[0,1); j ~ UnifInt[0,N); if (u <= w[j] / w[k]) k = j; }
// The draw order (u, then j), the division and the `<=` are the reference's, so a NaN ratio never
// accepts.  The random numbers and the gather of step n do not depend on the chain state, only
// the compare does, so the loop is unrolled by four: four Philox blocks and four gathers are in
// flight before the four dependent accept tests.  The chain state is updated by SELECTS, not inside
// `if (accept) { k = j; ... }`: with an accept condition of the form `a || expensive(b)` hipcc 7.2
// (gfx950) dropped the `k = j` of the second disjunct -- the register that held j was reused for
// expensive()'s result and `k` got its old value back while its companion (the weight) was updated --
// caught by the bit-exact comparison with the oracle, twice.
static __device__ __forceinline__ uint32_t metropolis_chain(const double *__restrict__ w, uint32_t N,
                                                            uint32_t B, uint32_t i, uint32_t step,
                                                            uint32_t k0, uint32_t k1)
{
  uint32_t k = i;
  double wk = w[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    double u[4], wj[4];
    uint32_t j[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 r = philox4x32_10(i, n + c, step, 1u, k0, k1);
      u[c] = u01_53(r.x, r.y);
      j[c] = uint_below(r.z, r.w, N);
#ifdef CUSMC_ABL_MH_COALESCED  // ablation builds only (scripts/calib/mh_time.py): the chain without its random gather
      wj[c] = w[(i + n + c) % N] + (double)(j[c] & 1);
#else
      wj[c] = w[j[c]];
#endif
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool acc = u[c] <= wj[c] / wk;
      k = acc ? j[c] : k;
      wk = acc ? wj[c] : wk;
    }
  }
  for (; n < B; ++n) {
    const u32x4 r = philox4x32_10(i, n, step, 1u, k0, k1);
    const double u = u01_53(r.x, r.y);
    const uint32_t j = uint_below(r.z, r.w, N);
    const double wj = w[j];
    const bool acc = u <= wj / wk;
    k = acc ? j : k;
    wk = acc ? wj : wk;
  }
  return k;
}

// The same chain with its gathers served from a table HALF the size: whi[j] = the high 32 bits of
// w[j] (sign, exponent, 20 mantissa bits: a lower bound of w[j] within 2^-20).  At N = 1e6 the
// doubles are 8 MB -- twice one XCD's 4 MB L2 -- and the chain is bound by L2 misses (85 us for
// B = 10); the 4 MB table fits and the same gathers take 45 us (scripts/calib/gather_probe.hip).
// The truncated values decide every step whose outcome they CAN decide:
//     wj in [a, a(1+2^-20)), wk in [b, b(1+2^-20))   =>   wj/wk in ( (a/b)(1-2^-19), (a/b)(1+2^-19) )
//     u b <= a (1 - 2^-18)  =>  u <= fl(wj / wk): accept;    u b > a (1 + 2^-18)  =>  reject
// (margins 2x wider than needed cover every rounding in sight), and only the sliver in between --
// about 2^-17 of the steps -- or operands outside the comfortable range (zero, denormal, tiny, huge,
// negative, NaN) fetch the two doubles and run the reference's own test, u <= w[j] / w[k].  The
// index sequence is therefore IDENTICAL to metropolis_chain's, and the common step saves the fp64
// division as well.
static __device__ __forceinline__ double hi_to_double(uint32_t hi)
{
  return __builtin_bit_cast(double, (uint64_t)hi << 32);
}
// positive, normal, and far enough from both ends of the exponent range that neither u * b nor the
// margins can overflow or lose bits: biased exponent in [123, 1923], sign clear
static __device__ __forceinline__ bool hi_comfortable(uint32_t hi) { return ((hi >> 20) - 123u) <= 1800u; }

// One step of the chain on the truncated table: (k, bh) = current index and the high word of its
// weight; returns with them updated.
static __device__ __forceinline__ void metropolis_step_hi(const double *__restrict__ w, double u, uint32_t j,
                                                          uint32_t ah, uint32_t &k, uint32_t &bh)
{
  bool acc, decided = false;
  if (hi_comfortable(ah) && hi_comfortable(bh)) {
    const double a = hi_to_double(ah), p = u * hi_to_double(bh);
    if (p <= a * (1.0 - 0x1p-18)) {
      acc = true;
      decided = true;
    } else if (p > a * (1.0 + 0x1p-18)) {
      acc = false;
      decided = true;
    }
  }
  if (!decided) acc = u <= w[j] / w[k];  // the reference's own test, on the full doubles
  k = acc ? j : k;
  bh = acc ? ah : bh;
}

static __device__ __forceinline__ uint32_t metropolis_chain_hi(const double *__restrict__ w,
                                                               const uint32_t *__restrict__ whi, uint32_t N,
                                                               uint32_t B, uint32_t i, uint32_t step,
                                                               uint32_t k0, uint32_t k1)
{
  uint32_t k = i, bh = whi[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    double u[4];
    uint32_t j[4], ah[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 r = philox4x32_10(i, n + c, step, 1u, k0, k1);
      u[c] = u01_53(r.x, r.y);
      j[c] = uint_below(r.z, r.w, N);
      ah[c] = whi[j[c]];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) metropolis_step_hi(w, u[c], j[c], ah[c], k, bh);
  }
  for (; n < B; ++n) {
    const u32x4 r = philox4x32_10(i, n, step, 1u, k0, k1);
    const uint32_t j = uint_below(r.z, r.w, N);
    metropolis_step_hi(w, u01_53(r.x, r.y), j, whi[j], k, bh);
  }
  return k;
}

// exp(t) for t <= 0, written so that oracle/cusmc_oracle.c:exp_nonpos evaluates the SAME sequence
// of correctly rounded operations (explicit fma where fused, nothing else for the compiler to
// contract): the log-weight resampler's accept test u <= exp(lw[j] - lw[k]) then gives bit-identical
// index sequences on the GPU and in the oracle, which no pair of library exp() would.  Reduction
// t = k ln2 + r and the degree-5 rational form of fdlibm's e_exp.c (constants from there).
static __device__ __forceinline__ double exp_nonpos(double t)
{
  // branch-free: the core runs on a clamped argument and the out-of-range answers are selected in
  // afterwards (underflow and -inf -> 0, NaN -> NaN).  Same operations as the oracle on (-746, 0].
  const bool in_range = t > -746.0;
  const double tc = in_range ? (t < 0.0 ? t : 0.0) : -746.0;
  const double kf = __builtin_rint(tc * 1.44269504088896338700e+00);
  double r = fma(-kf, 6.93147180369123816490e-01, tc);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  const double r2 = r * r;
  const double pp = fma(r2, fma(r2, fma(r2, fma(r2, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                        6.61375632143793436117e-05), -2.77777777770155933842e-03),
                        1.66666666666666019037e-01);
  const double c = fma(-r2, pp, r);
  const double num = r * c;
  const double den = c - 2.0;
  const double quo = num / den;
  const double e = ldexp(1.0 - (quo - r), (int)kf);
  return in_range ? e : (t != t ? t : 0.0);
}

// The chain over LOG-weights: accept iff u <= exp(lw[j] - lw[k]); a non-negative difference accepts
// without evaluating anything (u < 1).  -inf is a zero weight: it is never entered except from
// another zero weight's NaN ... which rejects, as w = 0 does in the density form.
static __device__ __forceinline__ uint32_t metropolis_chain_log(const double *__restrict__ lw, uint32_t N,
                                                                uint32_t B, uint32_t i, uint32_t step,
                                                                uint32_t k0, uint32_t k1)
{
  uint32_t k = i;
  double lk = lw[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    double u[4], lj[4];
    uint32_t j[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 r = philox4x32_10(i, n + c, step, 1u, k0, k1);
      u[c] = u01_53(r.x, r.y);
      j[c] = uint_below(r.z, r.w, N);
      lj[c] = lw[j[c]];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double t = lj[c] - lk;
      const bool acc = (t >= 0.0) | (u[c] <= exp_nonpos(t));
      k = acc ? j[c] : k;  // (selects, not branches: see the note on metropolis_chain)
      lk = acc ? lj[c] : lk;
    }
  }
  for (; n < B; ++n) {
    const u32x4 r = philox4x32_10(i, n, step, 1u, k0, k1);
    const double u = u01_53(r.x, r.y);
    const uint32_t j = uint_below(r.z, r.w, N);
    const double lj = lw[j], t = lj - lk;
    const bool acc = (t >= 0.0) | (u <= exp_nonpos(t));
    k = acc ? j : k;
    lk = acc ? lj : lk;
  }
  return k;
}

static __device__ __forceinline__ double finish_generic(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

}  // namespace cusmc
