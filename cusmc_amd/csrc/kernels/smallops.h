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

// log1p(t) for t >= 0 (the Student-t epilogue's q / nu): ln(1 + t) plus the first-order correction for the rounding
// of 1 + t; Inf and NaN pass through.  (The library log1p also serves t in (-1, 0) and costs four times the
// instructions.)
static __device__ __forceinline__ double log1p_nonneg(double t)
{
  const double u = 1.0 + t;
  const double c = (t - (u - 1.0)) * __builtin_amdgcn_rcp(u);
  const double r = ln_pos(u) + c;
  return u < __builtin_inf() ? r : u;
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

// chi^2_nu draws, RNG CONTRACT 4 (restated in oracle/cusmc_oracle.c:chi_square_for).
// Law: chi[j] ~ chi^2_nu per component, independent of each other and of the normals
// (src/statistics.cc.cpp:366, 383-386; device helper src/mvt_dist.cu.cpp:20-61).  Keyed by the component PAIR
// p = j / 2, half e = j % 2 -- the unit the proposal normals are keyed by:
//   integer nu <= 16   closed form, m = nu / 2 (rounded down):  chi^2_nu = -2 ln(u_1 .. u_m) [+ z^2 if nu is odd].
//                  Block b < ceil(m / 2) of the pair: (particle, p + (b << 16), step, 6); half e takes its words
//                  (2e, 2e+1) as u_{2b+1}, u_{2b+2} -- m = 1: ONE 52-bit uniform from both words, m >= 2: 32-bit
//                  uniforms, all in (0, 1), multiplied in that order.  Odd nu: (z_0, z_1) = Box-Muller of block
//                  (particle, p, step, 8), half e adds z_e^2 by fma.  No rejection, no divergent branch.
//                  (Contract 2 / 3 had this for nu = 2 and 4 only; their draws are unchanged.)
//   other nu       Marsaglia-Tsang, squeeze and a < 1 boost as the reference's helper.  Attempt m < 63 of pair p:
//                  block (particle, 64 p + m, step, 3) -> Box-Muller -> (z0, z1) = the normals of halves (0, 1);
//                  block (particle, 64 p + m, step, 5) -> words (2e, 2e+1) -> half e's uniform in (0, 1].  Accept iff
//                      v = 1 + c z > 0   and   ( u < 1 - 0.0331 z^4   or   ln u < z^2/2 + dd - dd v^3 + dd ln v^3 ).
//                  Boost uniform: block (particle, 64 p + 63, step, 5), words (2e, 2e+1).
// Contract 1 spent two blocks and a Box-Muller pair per attempt AND COMPONENT (second normal and half of the
// uniforms discarded): a Student-t component cost 3 - 4.6 x a Normal one (profiles/r02_summary.md).
//
// Cost model (f64 MFMA kernels: every VALU instruction is on the critical path; elsewhere: the proposal
// kernels are RNG-bound).  The squeeze settles ~92 % of the attempts without a logarithm, ~96 % accept -- but a
// wave runs a branch as long as ONE of its 64 lanes needs it, so a per-draw loop executes its slow path on
// nearly every draw.  chi_pair_batch therefore takes the KP pairs of a lane TOGETHER: first attempt and squeeze
// for all of them with no branch, then the few halves still open go through the full loop one per trip.
struct ChiSquare {
  double dd, c, inv_a;
  bool boost;
  int closed;  // 0: Marsaglia-Tsang; 1: closed form with m uniforms per component, plus a squared normal if odd
  int m;
  bool odd;
};
// which nu take the closed form (the launchers pick kernel variants by it)
__host__ __device__ __forceinline__ bool chi_nu_closed(float nu) { return nu >= 1.0f && nu <= 16.0f && nu == (float)(int)nu; }
static __device__ __forceinline__ ChiSquare chi_setup(float nu)
{
  ChiSquare cs;
  double a = 0.5 * (double)nu;
  cs.closed = chi_nu_closed(nu) ? 1 : 0;
  cs.m = cs.closed ? (int)nu / 2 : 0;
  cs.odd = cs.closed && ((int)nu & 1);
  cs.boost = a < 1.0;
  cs.inv_a = 1.0 / a;
  if (cs.boost) a += 1.0;
  cs.dd = a - 1.0 / 3.0;
  cs.c = 1.0 / sqrt(9.0 * cs.dd);
  return cs;
}
// closed form: both halves of pair p from one block
template <int M>
static __device__ __forceinline__ void chi_closed_pair(uint32_t particle, uint32_t p, uint32_t step, uint32_t k0,
                                                       uint32_t k1, double &chi0, double &chi1)
{
  const u32x4 r = philox4x32_10(particle, p, step, 6u, k0, k1);
  double P0, P1;
  if (M == 1) {
    P0 = ((double)(((((uint64_t)r.x << 32) | r.y)) >> 12) + 0.5) * 0x1.0p-52;
    P1 = ((double)(((((uint64_t)r.z << 32) | r.w)) >> 12) + 0.5) * 0x1.0p-52;
  } else {
    P0 = fma((double)r.x, 0x1.0p-32, 0x1.0p-33) * fma((double)r.y, 0x1.0p-32, 0x1.0p-33);
    P1 = fma((double)r.z, 0x1.0p-32, 0x1.0p-33) * fma((double)r.w, 0x1.0p-32, 0x1.0p-33);
  }
  chi0 = -2.0 * ln_pos(P0);
  chi1 = -2.0 * ln_pos(P1);
}
// any integer nu <= 16 (cs.m, cs.odd wave-uniform: nu is a launch parameter)
static __device__ __forceinline__ void chi_closed_pair_any(const ChiSquare &cs, uint32_t particle, uint32_t p, uint32_t step,
                                                           uint32_t k0, uint32_t k1, double &chi0, double &chi1)
{
  double P0 = 1.0, P1 = 1.0;
  if (cs.m == 1) {
    const u32x4 r = philox4x32_10(particle, p, step, 6u, k0, k1);
    P0 = ((double)(((((uint64_t)r.x << 32) | r.y)) >> 12) + 0.5) * 0x1.0p-52;
    P1 = ((double)(((((uint64_t)r.z << 32) | r.w)) >> 12) + 0.5) * 0x1.0p-52;
  } else {
#pragma unroll 1
    for (int b = 0; 2 * b < cs.m; ++b) {
      const u32x4 r = philox4x32_10(particle, p + ((uint32_t)b << 16), step, 6u, k0, k1);
      P0 *= fma((double)r.x, 0x1.0p-32, 0x1.0p-33);
      P1 *= fma((double)r.z, 0x1.0p-32, 0x1.0p-33);
      if (2 * b + 1 < cs.m) {
        P0 *= fma((double)r.y, 0x1.0p-32, 0x1.0p-33);
        P1 *= fma((double)r.w, 0x1.0p-32, 0x1.0p-33);
      }
    }
  }
  chi0 = cs.m ? -2.0 * ln_pos(P0) : 0.0;
  chi1 = cs.m ? -2.0 * ln_pos(P1) : 0.0;
  if (cs.odd) {
    double z0, z1;
    normal_pair(philox4x32_10(particle, p, step, 8u, k0, k1), z0, z1);
    chi0 = fma(z0, z0, chi0);
    chi1 = fma(z1, z1, chi1);
  }
}
// attempt m of pair p: the two normals, v = 1 + c z and the two uniforms in (0, 1]
static __device__ __forceinline__ void chi_attempt_pair(const ChiSquare &cs, uint32_t particle, uint32_t pm, uint32_t step,
                                                        uint32_t k0, uint32_t k1, double (&z)[2], double (&v)[2],
                                                        double (&u)[2])
{
  normal_pair(philox4x32_10(particle, pm, step, 3u, k0, k1), z[0], z[1]);
  v[0] = 1.0 + cs.c * z[0];
  v[1] = 1.0 + cs.c * z[1];
  const u32x4 r = philox4x32_10(particle, pm, step, 5u, k0, k1);
  u[0] = 1.0 - u01_53(r.x, r.y);
  u[1] = 1.0 - u01_53(r.z, r.w);
}
// u < 1 - 0.0331 z^4, as ONE fixed sequence of roundings (the oracle evaluates the same fma)
static __device__ __forceinline__ bool chi_squeeze(double z0, double u)
{
  const double z2 = z0 * z0;
  return u < fma(-(0.0331 * z2), z2, 1.0);
}
// one half's walk through its attempts from m = 0 with the full rule; returns Gamma(a, 1) / boost
static __device__ __forceinline__ double chi_loop(const ChiSquare &cs, uint32_t particle, uint32_t p, uint32_t e,
                                                  uint32_t step, uint32_t k0, uint32_t k1)
{
  double g = cs.dd;  // value if all 63 attempts reject (probability < 1e-60)
  for (uint32_t m = 0; m < 63u; ++m) {
    double z[2], v[2], u[2];
    chi_attempt_pair(cs, particle, p * 64u + m, step, k0, k1, z, v, u);
    const double ze = e ? z[1] : z[0], ue = e ? u[1] : u[0];
    double ve = e ? v[1] : v[0];
    if (ve <= 0.0) continue;
    ve = ve * ve * ve;
    if (chi_squeeze(ze, ue) || ln_pos(ue) < 0.5 * ze * ze + cs.dd - cs.dd * ve + cs.dd * ln_pos(ve)) {
      g = cs.dd * ve;
      break;
    }
  }
  return g;
}
static __device__ __forceinline__ void chi_boost_pair(const ChiSquare &cs, uint32_t particle, uint32_t p, uint32_t step,
                                                      uint32_t k0, uint32_t k1, double &b0, double &b1)
{
  const u32x4 r = philox4x32_10(particle, p * 64u + 63u, step, 5u, k0, k1);
  b0 = pow(1.0 - u01_53(r.x, r.y), cs.inv_a);
  b1 = pow(1.0 - u01_53(r.z, r.w), cs.inv_a);
}
// The open draws of a WAVE, resolved together.  After the branch-free first pass ~8 % of the draws are still open:
// with 8 .. 16 draws per lane that is one lane in two, and a wave that lets every lane walk its own takes as many
// trips through the full loop as its unluckiest lane has open draws (3 - 4 of the ~41 a wave of 512 draws leaves) --
// more time than the first pass itself.  Instead the lanes pool them in a per-wave LDS queue and share them out
// again, one per ACTIVE lane and trip: 41 open draws are one trip.  (Draws that do not fit the queue stay with
// their lane.)  Same counters, same results: who computes a draw does not enter it.
struct ChiQueue {
  static constexpr int CAP = 64;
  uint32_t count, pad;
  uint32_t item[CAP][2];  // (particle, 2 p + e)
  double result[CAP];
};
static __device__ __forceinline__ void chi_wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (LDS is in order per wave: this is s_waitcnt lgkmcnt(0) and a compiler barrier)
  __builtin_amdgcn_wave_barrier();
}

// chi[2 c + e] = chi^2_nu draw of component 2 pof(c) + e for the c < KP with live(c); pof / live are evaluated
// for run-time c as well (closed forms, not tables: no dynamically indexed registers).  A dead pair gets 1.
// q: this WAVE's queue in LDS (count zeroed before the first call), or NULL.
template <int KP, typename POf, typename Live>
static __device__ __forceinline__ void chi_pair_batch(const ChiSquare &cs, uint32_t particle, uint32_t step, uint32_t k0,
                                                      uint32_t k1, POf pof, Live live, double (&chi)[2 * KP],
                                                      ChiQueue *q = nullptr)
{
  static_assert(KP <= 16, "one pending bit per draw");
#pragma unroll
  for (int c = 0; c < 2 * KP; ++c) chi[c] = 1.0;
  if (cs.closed) {  // (wave-uniform: nu is a launch parameter)
    // a real loop for the same reason as below: KP unrolled copies of Philox + two ln keep ~40 registers each
#pragma unroll 1
    for (int c = 0; c < KP; ++c) {
      if (!live(c)) continue;
      double g0, g1;
      if (cs.m == 1 && !cs.odd) chi_closed_pair<1>(particle, (uint32_t)pof(c), step, k0, k1, g0, g1);
      else if (cs.m == 2 && !cs.odd) chi_closed_pair<2>(particle, (uint32_t)pof(c), step, k0, k1, g0, g1);
      else chi_closed_pair_any(cs, particle, (uint32_t)pof(c), step, k0, k1, g0, g1);
#pragma unroll
      for (int cc = 0; cc < KP; ++cc) {
        chi[2 * cc] = cc == c ? g0 : chi[2 * cc];
        chi[2 * cc + 1] = cc == c ? g1 : chi[2 * cc + 1];
      }
    }
    return;
  }
  uint32_t pend = 0;
  // A real loop, not KP unrolled copies: unrolled, the independent attempts are interleaved by the
  // scheduler and their ~50 live registers each add up (100-200 VGPRs spilled in the matrix-core proposal);
  // the price is the select chain that files the result under a run-time c.
#pragma unroll 1
  for (int c = 0; c < KP; ++c) {
    if (!live(c)) continue;
    double z[2], v[2], u[2];
    chi_attempt_pair(cs, particle, (uint32_t)pof(c) * 64u, step, k0, k1, z, v, u);
    const bool ok0 = (v[0] > 0.0) & chi_squeeze(z[0], u[0]), ok1 = (v[1] > 0.0) & chi_squeeze(z[1], u[1]);
    const double g0 = cs.dd * (v[0] * v[0] * v[0]), g1 = cs.dd * (v[1] * v[1] * v[1]);
    pend |= (ok0 ? 0u : 1u << (2 * c)) | (ok1 ? 0u : 2u << (2 * c));
#pragma unroll
    for (int cc = 0; cc < KP; ++cc) {
      chi[2 * cc] = cc == c ? g0 : chi[2 * cc];
      chi[2 * cc + 1] = cc == c ? g1 : chi[2 * cc + 1];
    }
  }
  if (q) {
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    const uint32_t nact = (uint32_t)__builtin_popcountll(act);
    const uint32_t n_mine = (uint32_t)__builtin_popcount(pend);
    uint32_t base = 0;
    if (n_mine) base = __hip_atomic_fetch_add(&q->count, n_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t rest = pend, k = 0;
    while (rest) {  // publish what fits
      const int c = __builtin_ctz(rest);
      rest &= rest - 1u;
      const uint32_t slot = base + k++;
      if (slot < (uint32_t)ChiQueue::CAP) {
        q->item[slot][0] = particle;
        q->item[slot][1] = ((uint32_t)pof(c >> 1) << 1) | (uint32_t)(c & 1);
      }
    }
    chi_wave_fence();
    uint32_t total = __hip_atomic_load(&q->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    total = total < (uint32_t)ChiQueue::CAP ? total : (uint32_t)ChiQueue::CAP;
    for (uint32_t idx = rank; idx < total; idx += nact) {  // one open draw per ACTIVE lane per trip
      const uint32_t ip = q->item[idx][0], pe = q->item[idx][1];
      q->result[idx] = chi_loop(cs, ip, pe >> 1, pe & 1u, step, k0, k1);
    }
    chi_wave_fence();
    rest = pend, k = 0;
    while (rest) {  // collect
      const int c = __builtin_ctz(rest);
      rest &= rest - 1u;
      const uint32_t slot = base + k++;
      if (slot < (uint32_t)ChiQueue::CAP) {
        const double g = q->result[slot];
        pend &= ~(1u << c);
#pragma unroll
        for (int cc = 0; cc < 2 * KP; ++cc) chi[cc] = cc == c ? g : chi[cc];
      }
    }
    chi_wave_fence();
    if (rank == 0) __hip_atomic_store(&q->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    chi_wave_fence();
  }
  while (pend) {  // one open draw per lane per trip (no queue, or its overflow)
    const int c = __builtin_ctz(pend);
    pend &= pend - 1u;
    const double g = chi_loop(cs, particle, (uint32_t)pof(c >> 1), (uint32_t)(c & 1), step, k0, k1);
#pragma unroll
    for (int cc = 0; cc < 2 * KP; ++cc) chi[cc] = cc == c ? g : chi[cc];
  }
#pragma unroll
  for (int c = 0; c < KP; ++c) {
    chi[2 * c] *= 2.0;
    chi[2 * c + 1] *= 2.0;
    if (cs.boost) {  // (wave-uniform)
      if (live(c)) {
        double b0, b1;
        chi_boost_pair(cs, particle, (uint32_t)pof(c), step, k0, k1, b0, b1);
        chi[2 * c] *= b0;
        chi[2 * c + 1] *= b1;
      }
    }
  }
}
// the D draws of one particle held by one lane (lane = particle kernels): chi[j], j < D
template <int D>
static __device__ __forceinline__ void chi_square_all(const ChiSquare &cs, uint32_t particle, uint32_t step, uint32_t k0,
                                                      uint32_t k1, double (&chi)[D], ChiQueue *q = nullptr)
{
  constexpr int KP = (D + 1) / 2;
  double tmp[2 * KP];
  chi_pair_batch<KP>(cs, particle, step, k0, k1, [](int c) { return c; }, [](int) { return true; }, tmp, q);
#pragma unroll
  for (int j = 0; j < D; ++j) chi[j] = tmp[j];
}
// a single draw (callers that own one component per lane)
static __device__ __forceinline__ double chi_square_for(uint32_t particle, uint32_t j, uint32_t step, uint32_t k0,
                                                        uint32_t k1, float nu)
{
  const ChiSquare cs = chi_setup(nu);
  double chi[2];
  chi_pair_batch<1>(cs, particle, step, k0, k1, [&](int) { return j >> 1; }, [](int) { return true; }, chi);
  return (j & 1u) ? chi[1] : chi[0];
}
// The lane's draws in the matrix cores' C layout: lane (p, h) owns components blk(b) + h + 4 r, r < 4, of NBLK
// 16-blocks, out[4 b + r].  Pairs (j, j + 1), j even, straddle the lanes h and h ^ 1 (16 lanes apart, the same
// particle): the even-h lane draws the pairs of r = 0, 1, the odd-h lane those of r = 2, 3, each keeps the half
// of its own parity and hands the other one over -- 2 pair draws per lane and block instead of 4 single ones.
template <int NBLK, typename BlkOf, typename LiveJ>
static __device__ __forceinline__ void chi_square_clayout(const ChiSquare &cs, uint32_t particle, uint32_t step,
                                                          uint32_t k0, uint32_t k1, int h, BlkOf blk, LiveJ livej,
                                                          double (&out)[4 * NBLK], ChiQueue *q = nullptr)
{
  const int e = h & 1, hb = h - e;
  auto jof = [&](int c) { return blk(c >> 1) + hb + 4 * (2 * e + (c & 1)); };  // the even component of pair c
  double mine[4 * NBLK];
  chi_pair_batch<2 * NBLK>(cs, particle, step, k0, k1, [&](int c) { return jof(c) >> 1; }, [&](int c) { return livej(c >> 1, jof(c)); }, mine, q);
#pragma unroll
  for (int c = 0; c < 2 * NBLK; ++c) {
    const double keep = e ? mine[2 * c + 1] : mine[2 * c], give = e ? mine[2 * c] : mine[2 * c + 1];
    const double got = __shfl_xor(give, 16);
    const int b = c >> 1, k = c & 1;
    out[4 * b + k] = e ? got : keep;      // r = k: drawn by the even-h lane
    out[4 * b + 2 + k] = e ? keep : got;  // r = 2 + k: drawn by the odd-h lane
  }
}

// One Metropolis chain (Sampler::metropolis_hastings, src/samplers.cpp:21-35):
//     k = i;  B times { u ~ U[0,1); j ~ UnifInt[0,N); if (u <= w[j] / w[k]) k = j; }
// The draw order (u, then j), the division and the `<=` are the reference's, so a NaN ratio never
// accepts.  Draws: RNG contract 3 (philox.h) -- ONE Philox block per TWO steps (the ten rounds were ~50 of the
// ~75 VALU instructions of a step).  The random numbers and the gather of step n do not depend on the chain
// state, only the compare does, so the loop is unrolled by four: two Philox blocks and four gathers are in
// flight before the four dependent accept tests.  The chain state is updated by SELECTS, not inside
// `if (accept) { k = j; ... }`: with an accept condition of the form `a || expensive(b)` hipcc 7.2
// (gfx950) dropped the `k = j` of the second disjunct -- the register that held j was reused for
// expensive()'s result and `k` got its old value back while its companion (the weight) was updated --
// caught by the bit-exact comparison with the oracle, twice.
struct MhDraw4 {
  uint32_t a[4], j[4];
};
// the draws of steps n .. n+3 (n a multiple of 4) of chain i
static __device__ __forceinline__ MhDraw4 mh_draw4(uint32_t i, uint32_t n, uint32_t step, uint32_t N, uint32_t tN,
                                                   uint32_t k0, uint32_t k1)
{
  MhDraw4 d;
  const u32x4 r0 = philox4x32_10(i, n >> 1, step, 1u, k0, k1), r1 = philox4x32_10(i, (n >> 1) + 1u, step, 1u, k0, k1);
  d.a[0] = r0.x, d.a[1] = r0.z, d.a[2] = r1.x, d.a[3] = r1.z;
  d.j[0] = mh_index(r0.y, i, n, step, N, tN, k0, k1);
  d.j[1] = mh_index(r0.w, i, n + 1u, step, N, tN, k0, k1);
  d.j[2] = mh_index(r1.y, i, n + 2u, step, N, tN, k0, k1);
  d.j[3] = mh_index(r1.w, i, n + 3u, step, N, tN, k0, k1);
  return d;
}
// one step's draws (the loop's remainder)
static __device__ __forceinline__ void mh_draw1(uint32_t i, uint32_t n, uint32_t step, uint32_t N, uint32_t tN, uint32_t k0,
                                                uint32_t k1, uint32_t &a, uint32_t &j)
{
  const u32x4 r = philox4x32_10(i, n >> 1, step, 1u, k0, k1);
  a = (n & 1u) ? r.z : r.x;
  j = mh_index((n & 1u) ? r.w : r.y, i, n, step, N, tN, k0, k1);
}

static __device__ __forceinline__ uint32_t metropolis_chain(const double *__restrict__ w, uint32_t N,
                                                            uint32_t B, uint32_t i, uint32_t step,
                                                            uint32_t k0, uint32_t k1)
{
  const uint32_t tN = mh_tn(N);
  uint32_t k = i;
  double wk = w[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    const MhDraw4 d = mh_draw4(i, n, step, N, tN, k0, k1);
    double wj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#ifdef CUSMC_ABL_MH_COALESCED  // ablation builds only (scripts/calib/mh_time.py): the chain without its random gather
      wj[c] = w[(i + n + c) % N] + (double)(d.j[c] & 1);
#else
      wj[c] = w[d.j[c]];
#endif
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool acc = mh_accept(d.a[c], wj[c] / wk, i, n + c, step, k0, k1);
      k = acc ? d.j[c] : k;
      wk = acc ? wj[c] : wk;
    }
  }
  for (; n < B; ++n) {
    uint32_t a, j;
    mh_draw1(i, n, step, N, tN, k0, k1, a, j);
    const double wj = w[j];
    const bool acc = mh_accept(a, wj / wk, i, n, step, k0, k1);
    k = acc ? j : k;
    wk = acc ? wj : wk;
  }
  return k;
}

// The same chain with its gathers served from a table HALF the size: whi[j] = the high 32 bits of
// w[j] (sign, exponent, 20 mantissa bits: a lower bound of w[j] within 2^-20).  At N = 1e6 the
// doubles are 8 MB -- twice one XCD's 4 MB L2 -- and the chain is bound by L2 misses (85 us for
// B = 10); the 4 MB table fits and the same gathers take 45 us (scripts/calib/gather_probe.hip).
// The truncated values decide every step whose outcome they CAN decide.  With u in [lo, hi) (its 32 leading
// bits: philox.h), wj in [a, a(1+2^-20)), wk in [b, b(1+2^-20)):
//     wj/wk in ( (a/b)(1-2^-19), (a/b)(1+2^-19) )
//     hi b <= a (1 - 2^-18)  =>  hi <= fl(wj / wk): the reference's test accepts;    lo b > a (1 + 2^-18)  =>  rejects
// (margins 2x wider than needed cover every rounding in sight), and only the sliver in between --
// about 2^-17 of the steps -- or operands outside the comfortable range (zero, denormal, tiny, huge,
// negative, NaN) fetch the two doubles and run the reference's own test, u <= w[j] / w[k] (mh_accept).  The
// index sequence is therefore IDENTICAL to metropolis_chain's, and the common step also saves the fp64
// division.
static __device__ __forceinline__ double hi_to_double(uint32_t hi)
{
  return __builtin_bit_cast(double, (uint64_t)hi << 32);
}
// positive, normal, and far enough from both ends of the exponent range that neither u * b nor the
// margins can overflow or lose bits: biased exponent in [123, 1923], sign clear
static __device__ __forceinline__ bool hi_comfortable(uint32_t hi) { return ((hi >> 20) - 123u) <= 1800u; }

// One step of the chain on the truncated table: (k, bh) = current index and the high word of its
// weight; returns with them updated.  ua = the 32 leading bits of u.
static __device__ __forceinline__ void metropolis_step_hi(const double *__restrict__ w, uint32_t ua, uint32_t j,
                                                          uint32_t ah, uint32_t &k, uint32_t &bh, uint32_t i, uint32_t n,
                                                          uint32_t step, uint32_t k0, uint32_t k1)
{
  bool acc, decided = false;
  if (hi_comfortable(ah) && hi_comfortable(bh)) {
    const double lo = (double)ua * 0x1.0p-32, hi = lo + 0x1.0p-32;
    const double a = hi_to_double(ah), b = hi_to_double(bh);
    if (hi * b <= a * (1.0 - 0x1p-18)) {
      acc = true;
      decided = true;
    } else if (lo * b > a * (1.0 + 0x1p-18)) {
      acc = false;
      decided = true;
    }
  }
  if (!decided) acc = mh_accept(ua, w[j] / w[k], i, n, step, k0, k1);  // the reference's own test, on the full doubles
  k = acc ? j : k;
  bh = acc ? ah : bh;
}

// lds_tab / L: the first L entries of whi held in LDS by the calling workgroup (L = 0: none).  The chain is bound by
// the request rate of its random gathers, one L2 request per lane and step: every gather that lands
// below L is an LDS read instead -- the same word, so the same index sequence.
static __device__ __forceinline__ uint32_t metropolis_chain_hi(const double *__restrict__ w,
                                                               const uint32_t *__restrict__ whi, uint32_t N,
                                                               uint32_t B, uint32_t i, uint32_t step,
                                                               uint32_t k0, uint32_t k1, const uint32_t *lds_tab = nullptr,
                                                               uint32_t L = 0)
{
  const uint32_t tN = mh_tn(N);
  // (two masked loads under a branch each.  Branch-free -- an LDS read at a clamped index plus a global read that
  // sends the LDS-resident lanes to word 0 -- measured SLOWER, 422 against 344 us at N = 1e5: the dummy lanes still
  // cost the vector-memory path their slots)
  auto word = [&](uint32_t j) -> uint32_t {
    uint32_t v;
    if (j < L) v = lds_tab[j];
    else v = whi[j];
    return v;
  };
  uint32_t k = i, bh = whi[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    const MhDraw4 d = mh_draw4(i, n, step, N, tN, k0, k1);
    uint32_t ah[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ah[c] = word(d.j[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) metropolis_step_hi(w, d.a[c], d.j[c], ah[c], k, bh, i, n + c, step, k0, k1);
  }
  for (; n < B; ++n) {
    uint32_t a, j;
    mh_draw1(i, n, step, N, tN, k0, k1, a, j);
    metropolis_step_hi(w, a, j, word(j), k, bh, i, n, step, k0, k1);
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
  const uint32_t tN = mh_tn(N);
  uint32_t k = i;
  double lk = lw[i];
  uint32_t n = 0;
  for (; n + 4 <= B; n += 4) {
    const MhDraw4 d = mh_draw4(i, n, step, N, tN, k0, k1);
    double lj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lj[c] = lw[d.j[c]];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double t = lj[c] - lk;
      const bool acc = (t >= 0.0) | mh_accept(d.a[c], exp_nonpos(t), i, n + c, step, k0, k1);
      k = acc ? d.j[c] : k;  // (selects, not branches: see the note on metropolis_chain)
      lk = acc ? lj[c] : lk;
    }
  }
  for (; n < B; ++n) {
    uint32_t a, j;
    mh_draw1(i, n, step, N, tN, k0, k1, a, j);
    const double lj = lw[j], t = lj - lk;
    const bool acc = (t >= 0.0) | mh_accept(a, exp_nonpos(t), i, n, step, k0, k1);
    k = acc ? j : k;
    lk = acc ? lj : lk;
  }
  return k;
}

static __device__ __forceinline__ double finish_generic(double q, const Epilogue &ep)
{
  double lp = (ep.kind == CUSMC_MVT) ? ep.lognorm - ep.half_nu_plus_d * log1p_nonneg(q * ep.inv_nu)
                                     : ep.lognorm - 0.5 * q;
  return ep.out_density ? exp(lp) : lp;
}

}  // namespace cusmc
