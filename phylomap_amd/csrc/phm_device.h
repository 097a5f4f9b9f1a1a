// phm_device.h -- device-side scalar building blocks shared by every kernel.
//
// These restate, independently of oracle/phm_oracle.c, the arithmetic spec of DESIGN.md:
// Philox4x32-7 counter streams, the (0,1) map of 32 random bits, and the deterministic
// log/exp (basic IEEE-754 binary64 operations only, no FMA contraction: the library is built
// with -ffp-contract=off) so that the CPU oracle and the GPU agree bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHM_LOGTAB_QUAL __device__
#include "phm_logtab.h"

namespace phm {

// device error bits -> phm_status (phm_internal.h)
constexpr uint32_t DERR_ZERO_PROB = 1u;   // all-zero / non-finite probability vector
constexpr uint32_t DERR_CAPACITY  = 2u;   // branch outgrew its slot capacity
constexpr uint32_t DERR_UNIF_CAP  = 4u;   // newunifSample > 300 jumps (src/phylomap.cpp:120)
constexpr uint32_t DERR_SAMPLEONCE = 8u;  // sampleOnce ran off the end (src/phylomap.cpp:85-89)

// entity tags: top two bits of Philox counter word 1
constexpr uint32_t ENT_NODE   = 0u;
constexpr uint32_t ENT_BSTATE = 1u << 30;
constexpr uint32_t ENT_BEXP   = 2u << 30;
constexpr uint32_t ENT_BUNIF  = 3u << 30;

// Philox4x32 with SEVEN rounds (Random123; Salmon et al. 2011): the fewest rounds at which the generator passes BigCrush
// ("Crush-resistant"), ten being Random123's default safety margin.  The round function and key schedule are pinned by the
// published known-answer vectors of both philox4x32-7 and philox4x32-10 (tests/test_oracle_cpu.py); a Philox block is a
// quarter of the VALU work of a sweep step, so three rounds fewer are 8 % fewer instructions.
#ifndef PHM_EXPERIMENT_PHILOX_ROUNDS      // measurement builds only (docs/EXPERIMENTS.md: what a cheaper generator could buy); never shipped
#define PHM_EXPERIMENT_PHILOX_ROUNDS 7
#endif
constexpr int PHILOX_ROUNDS = PHM_EXPERIMENT_PHILOX_ROUNDS;
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// 32 random bits -> double in the open interval (0,1): (x + 0.5) * 2^-32, exact.  R's own unif_rand (which the reference
// draws from through runif / rexp / sample) has the same 32-bit resolution; four uniforms come out of one Philox block.
__device__ __forceinline__ double u01(uint32_t x) {
  return ((double)x + 0.5) * 2.3283064365386962890625e-10;
}

// Per-tile arrays are addressed as a wave-uniform base (scalar registers) plus a 32-bit byte offset: one address register
// per access instead of a 64-bit vector address computed for each (the host keeps every array addressed this way below 4 GB).
template <class T>
__device__ __forceinline__ T& at(T* base, uint32_t byte_off) {
  return *reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ const T& at(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(base) + byte_off);
}

// Sequential draws of one stream (replica, iteration, entity): draw d is word d & 3 of Philox block d >> 2.
struct Stream {
  uint32_t ent, iter, rep, k0, k1;
  uint32_t blk;
  uint32_t w0, w1, w2, w3;
  __device__ __forceinline__ void open(uint32_t entity, uint32_t iteration, uint32_t replica, uint32_t seed_lo,
                                       uint32_t seed_hi) {
    ent = entity; iter = iteration; rep = replica; k0 = seed_lo; k1 = seed_hi; blk = 0xFFFFFFFFu; w0 = w1 = w2 = w3 = 0u;
  }
  __device__ __forceinline__ uint32_t draw_word(uint32_t d) {      // the 32 random bits of draw d
    uint32_t b = d >> 2;
    if (b != blk) {
      uint32_t o[4];
      philox4x32(b, ent, iter, rep, k0, k1, o);
      w0 = o[0]; w1 = o[1]; w2 = o[2]; w3 = o[3];
      blk = b;
    }
    const uint32_t lo = (d & 1u) ? w1 : w0, hi = (d & 1u) ? w3 : w2;
    return (d & 2u) ? hi : lo;
  }
  __device__ __forceinline__ double draw(uint32_t d) { return u01(draw_word(d)); }
};

// the 32 random bits of draw d of a stream, computed afresh (no block kept): lanes of a wave may ask for different draws
__device__ __forceinline__ uint32_t stream_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t rep, uint32_t iter,
                                                uint32_t ent, uint32_t d) {
  uint32_t o[4];
  philox4x32(d >> 2, ent, iter, rep, seed_lo, seed_hi, o);
  const uint32_t lo = (d & 1u) ? o[1] : o[0], hi = (d & 1u) ? o[3] : o[2];
  return (d & 2u) ? hi : lo;
}

// one-off draw (node states)
__device__ __forceinline__ double stream_u(uint32_t seed_lo, uint32_t seed_hi, uint32_t rep, uint32_t iter,
                                           uint32_t ent, uint32_t d) {
  uint32_t o[4];
  philox4x32(d >> 2, ent, iter, rep, seed_lo, seed_hi, o);
  const uint32_t lo = (d & 1u) ? o[1] : o[0], hi = (d & 1u) ? o[3] : o[2];
  return u01((d & 2u) ? hi : lo);
}

// Standard exponential variate from 32 random bits: -log(U), U = (k + 0.5) 2^-32 (what Rcpp::rexp's exp_rand delivers in
// distribution).  With y = 2k + 1 = 2^e f, f in [0.5, 1):  -log(U) = -((e - 33) ln 2 + log(c_j) + log1p(r)),
// c_j = 0.5 + (j + 0.5)/256 the table point below f, r = (f - c_j) / c_j (|r| < 2^-8; f - c_j is exact), log1p by its
// series to r^7; for U within 2^-8 of 1 the series is applied to r = f - 1 directly (no cancellation).  About one ulp
// (tools/gen_log_table.py, tests/test_oracle_cpu.py); 40 operations instead of the 65 of a general log.  `tab` holds the
// PHM_LOGTAB_N pairs (1/c_j, log c_j); the oracle and the Python restatement evaluate the same expression on the same table.
__device__ __forceinline__ double neglog_u32(uint32_t k, const double* __restrict__ tab) {
  const double y = __builtin_fma((double)k, 2.0, 1.0);            // 2k + 1 < 2^33: exact, fused or not
  int e;
  const double f = frexp(y, &e);
  const bool top = (e == 33) && (f >= 0.99609375);
  const int j = (int)((f - 0.5) * 256.0);                         // 0 .. 127 for every f in [0.5, 1)
  const double c = __builtin_fma((double)j, 0.00390625, 0.501953125);   // (2j + 257)/512: exact
  const double2 t = *reinterpret_cast<const double2*>(tab + 2 * j);     // one 16-byte read, no branch around it (tab 16-byte aligned)
  const double r = top ? f - 1.0 : (f - c) * t.x;
  const double c0 = top ? 0.0 : t.y;
  const double ee = top ? 0.0 : (double)(e - 33);
  double p = 1.0 / 7.0;
  p = p * r - 1.0 / 6.0;
  p = p * r + 0.2;
  p = p * r - 0.25;
  p = p * r + 1.0 / 3.0;
  p = p * r - 0.5;
  p = p * r * r + r;
  return -(ee * 6.93147180369123816490e-01 + (c0 + (p + ee * 1.90821492927058770002e-10)));
}

// the table as doubles, interleaved (1/c_j, log c_j), for kernels that stage it in LDS or read it through L1
__device__ __forceinline__ double logtab_entry(int i) {
  return __longlong_as_double((long long)((i & 1) ? PHM_LOGTAB_LOG_BITS[i >> 1] : PHM_LOGTAB_INV_BITS[i >> 1]));
}

// natural log for normal positive finite x (all callers pass u in (0,1) or validated positives)
__device__ __forceinline__ double phm_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int k = 0;
  if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; k -= 54; }
  uint64_t ux = (uint64_t)__double_as_longlong(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
  k += (int)(i >> 20);
  double f = __longlong_as_double((long long)ux) - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __forceinline__ double phm_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
               P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return __longlong_as_double(0x7ff0000000000000ll);
  if (x < -7.45133219101941108420e+02) return 0.0;
  double hi = x, lo = 0.0;
  int k = 0;
  if (fabs(x) > 0.34657359027997264) {
    k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    double t = (double)k;
    hi = x - t * ln2HI;
    lo = t * ln2LO;
  }
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return ldexp(y, k);
}

// first j with u*sum(p) <= p_0+..+p_j, index order (categorical draw)
template <int NS>
__device__ __forceinline__ int sample_cat(const double (&p)[NS], double u, uint32_t& err) {
  double total = p[0];
#pragma unroll
  for (int j = 1; j < NS; ++j) total += p[j];
  if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
  // first j with u * total <= p_0 + .. + p_j.  u < 1, so the threshold never exceeds the last partial sum (= total) and the
  // last comparison is always true; NS - 1 comparisons give the same index (and NS - 1 for a NaN total, flagged above).
  double thr = u * total;
  double cum = p[0];
  int idx = (thr <= cum) ? 0 : 1;
#pragma unroll
  for (int j = 1; j < NS - 1; ++j) { cum += p[j]; idx += (thr <= cum) ? 0 : 1; }
  return idx;
}

template <int NS>
__device__ __forceinline__ void matvec_u(const double* __restrict__ M, double (&v)[NS]) {
  // v <- M v, left-to-right, unfused; M is wave-uniform (kernel argument -> SGPRs)
  double y[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    double acc = M[i * NS] * v[0];
#pragma unroll
    for (int j = 1; j < NS; ++j) acc += M[i * NS + j] * v[j];
    y[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) v[i] = y[i];
}

// Maximum of a non-negative per-lane count over the 64 lanes of a wave (all lanes active at the call sites), as a wave-uniform
// value: row-shift / row-broadcast DPP moves (4 within the 16-lane rows, 2 across rows) leave it in lane 63 -- a third of
// the instructions of the shuffle butterfly, which goes through the LDS crossbar.  Lanes a move does not reach read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_step(int v) {
  const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ int wave_max_count(int v) {
  v = dpp_max_step<0x111, 0xf>(v);      // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);      // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);      // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);      // row_shr:8  -> lane 15 of every row holds its row's maximum
  v = dpp_max_step<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
  return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace phm
