// phm_coop.h -- wave-cooperative n-vector primitives of the 5..64-state kernels that keep one STATE per lane (phm_wbranch.hip,
// phm_wide.hip), scalars broadcast with v_readlane.  Arithmetic = the n > 4 spec of DESIGN.md section 2: a chain product is
// one FUSED multiply-add per term with j ascending from +0 (what the matrix cores compute in phm_wtiles.hip), the
// normalisation sum is four interleaved partial sums combined pairwise, prefix sums of a probability vector are formed in
// index order, unfused -- so results equal the oracle bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phm_device.h"

namespace phm {

__device__ __forceinline__ double readlane_f64(double v, int l) {   // l must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_max_w(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

// y <- M y with lanes over rows: M row-major in LDS with an ODD row stride ldn, so the 64 lanes reading M[c][j]
// (stride ldn doubles) fall on distinct banks, and the transposed access M[q][c] is contiguous anyway
__device__ __forceinline__ double coop_matvec(const double* __restrict__ M, double v, int n, int ldn, int c) {
  // The chain is one dependent FMA per term; what must not sit between two of them is the LDS latency of the matrix entry
  // (a read + s_waitcnt per term is ~100 cycles against ~10 for readlane + FMA).  Eight entries of the row are requested while
  // the previous eight are consumed; the order of the terms (j ascending, fused) is unchanged.
  const double* row = M + c * ldn;
  double acc = 0.0;
  double cur[8], nxt[8];
  const int n8 = n & ~7;
#pragma unroll
  for (int t = 0; t < 8; ++t) cur[t] = row[min(t, n - 1)];
  for (int j = 0; j < n8; j += 8) {
#pragma unroll
    for (int t = 0; t < 8; ++t) nxt[t] = row[min(j + 8 + t, n - 1)];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = __builtin_fma(cur[t], readlane_f64(v, j + t), acc);
#pragma unroll
    for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
  }
#pragma unroll
  for (int t = 0; t < 7; ++t)
    if (n8 + t < n) acc = __builtin_fma(cur[t], readlane_f64(v, n8 + t), acc);
  return acc;
}

// The same product with the vector passed through LDS as well (svec: 64 doubles of the calling wave, 16-byte aligned): v_readlane
// into an SGPR pair and its hazards cost ~3x the FMA they feed; a same-address LDS read is a broadcast and rides in the same
// batches of eight as the matrix entries.  Same terms, same order.
__device__ __forceinline__ double coop_matvec_lds(const double* __restrict__ M, double* __restrict__ svec, double v, int n, int ldn,
                                                  int c, int lane) {
  const double* row = M + c * ldn;
  svec[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  double acc = 0.0;
  double cur[8], nxt[8], vc[8], vn[8];
  const int n8 = n & ~7;
#pragma unroll
  for (int t = 0; t < 8; ++t) { cur[t] = row[min(t, n - 1)]; vc[t] = svec[t]; }
  for (int j = 0; j < n8; j += 8) {
#pragma unroll
    for (int t = 0; t < 8; ++t) { nxt[t] = row[min(j + 8 + t, n - 1)]; vn[t] = svec[(j + 8 + t) & 63]; }
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = __builtin_fma(cur[t], vc[t], acc);
#pragma unroll
    for (int t = 0; t < 8; ++t) { cur[t] = nxt[t]; vc[t] = vn[t]; }
  }
#pragma unroll
  for (int t = 0; t < 7; ++t)
    if (n8 + t < n) acc = __builtin_fma(cur[t], vc[t], acc);
  __builtin_amdgcn_wave_barrier();                   // every lane has read the vector before the next product overwrites it
  return acc;
}

// The same product with the lane's matrix row already in REGISTERS (brow[j] = M[c][j], zero beyond n): a kernel that applies the
// same matrix many times (the chains of a pruning level) loads its row once; a step is then 32 broadcast reads of the vector
// (two entries each) and the FMA chain.  Same terms, same order; the zero entries beyond n add fma(0, x, acc) = acc.
__device__ __forceinline__ double coop_matvec_regs(const double (&brow)[64], double* __restrict__ svec, double v, int n, int lane) {
  svec[lane] = (lane < n) ? v : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  const double2* sv2 = reinterpret_cast<const double2*>(svec);
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 64; j += 8) {
    if (j < n) {                                     // wave-uniform: whole batches of eight beyond n are skipped
      const double2 a = sv2[j / 2], b = sv2[j / 2 + 1], cc = sv2[j / 2 + 2], d = sv2[j / 2 + 3];
      acc = __builtin_fma(brow[j], a.x, acc);     acc = __builtin_fma(brow[j + 1], a.y, acc);
      acc = __builtin_fma(brow[j + 2], b.x, acc); acc = __builtin_fma(brow[j + 3], b.y, acc);
      acc = __builtin_fma(brow[j + 4], cc.x, acc); acc = __builtin_fma(brow[j + 5], cc.y, acc);
      acc = __builtin_fma(brow[j + 6], d.x, acc); acc = __builtin_fma(brow[j + 7], d.y, acc);
    }
  }
  __builtin_amdgcn_wave_barrier();                   // every lane has read the vector before the next product overwrites it
  return acc;
}

// the same product for a matrix with at most w non-zeros per row, kept in ELLPACK form (columns ascending; padding entries
// have value 0): lane c gathers v[col] of its own non-zeros.  Skipping the exact zeros of a row leaves its fused chain
// unchanged bit for bit -- fma(0, x, acc) = acc for finite x.
__device__ __forceinline__ double coop_matvec_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval, double v,
                                                  int w, int c) {
  double acc = 0.0;
  for (int t = 0; t < w; ++t) acc = __builtin_fma(eval[c * w + t], __shfl(v, ecol[c * w + t], 64), acc);
  return acc;
}

// the value of the lane below / above.  gfx950 has no whole-wave DPP shifts (wave_shr / wave_shl ended with gfx9.0): a row shift
// inside each row of 16 lanes, and the three lanes at a row boundary take their neighbour by v_readlane (uniform lane numbers).
// The lane at the end of the wave reads 0.
__device__ __forceinline__ double wave_from_below(double x, int lane) {        // lane c <- lane c - 1
  int lo = __double2loint(x), hi = __double2hiint(x);
  int slo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false);        // row_shr:1
  int shi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false);
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const int blo = __builtin_amdgcn_readlane(lo, 16 * r - 1), bhi = __builtin_amdgcn_readlane(hi, 16 * r - 1);
    slo = (lane == 16 * r) ? blo : slo; shi = (lane == 16 * r) ? bhi : shi;
  }
  return __hiloint2double(shi, slo);
}
__device__ __forceinline__ double wave_from_above(double x, int lane) {        // lane c <- lane c + 1
  int lo = __double2loint(x), hi = __double2hiint(x);
  int slo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, false);        // row_shl:1
  int shi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, false);
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const int blo = __builtin_amdgcn_readlane(lo, 16 * r), bhi = __builtin_amdgcn_readlane(hi, 16 * r);
    slo = (lane == 16 * r - 1) ? blo : slo; shi = (lane == 16 * r - 1) ? bhi : shi;
  }
  return __hiloint2double(shi, slo);
}

// The ELLPACK product for a BANDED matrix (half-bandwidth HB <= 2: the tutorial's tridiagonal Q): coef[d] = M[c][c + d - HB] (0 where
// the matrix has no entry), neighbours through DPP row shifts (+ readlane at the row boundaries) instead of ds_bpermute -- a chain step is
// 2 HB shifts and 2 HB + 1 fused multiply-adds.  Same terms in the same (ascending column) order; a zero coefficient leaves the fused chain unchanged.
template <int HB>
__device__ __forceinline__ double coop_matvec_band(const double (&coef)[5], double v, int lane) {
  double acc = 0.0;
  if (HB == 2) {
    const double m1 = wave_from_below(v, lane), p1 = wave_from_above(v, lane);
    const double m2 = wave_from_below(m1, lane), p2 = wave_from_above(p1, lane);
    acc = __builtin_fma(coef[0], m2, acc);
    acc = __builtin_fma(coef[1], m1, acc);
    acc = __builtin_fma(coef[2], v, acc);
    acc = __builtin_fma(coef[3], p1, acc);
    acc = __builtin_fma(coef[4], p2, acc);
  } else {
    const double m1 = wave_from_below(v, lane), p1 = wave_from_above(v, lane);
    acc = __builtin_fma(coef[0], m1, acc);
    acc = __builtin_fma(coef[1], v, acc);
    acc = __builtin_fma(coef[2], p1, acc);
  }
  return acc;
}

// first j with u*sum(p) <= p_0+..+p_j (index order); lanes >= n carry p = 0 and never count
__device__ __forceinline__ int coop_sample(double p, double u, int n, int lane, uint32_t& err) {
  double run = readlane_f64(p, 0);
  double mycum = run;
  for (int j = 1; j < n; ++j) {
    run += readlane_f64(p, j);
    if (lane == j) mycum = run;
  }
  if (!(run > 0.0) || isinf(run)) err |= DERR_ZERO_PROB;
  const double thr = u * run;
  const bool fail = (lane < n) && !(thr <= mycum);
  int idx = (int)__popcll(__ballot(fail));
  return idx < n ? idx : n - 1;
}

// The same draw with the probabilities passed through LDS (svec: 64 doubles of the calling wave, 16-byte aligned): every lane adds
// the running sums itself from broadcast reads (eight entries in flight) instead of 2 n v_readlane steps -- the same sums in
// the same order (entries beyond n are +0 and leave the running sum unchanged); about half the instructions of coop_sample.
__device__ __forceinline__ int coop_sample_lds(double p, double u, int n, int lane, double* __restrict__ svec, uint32_t& err) {
  svec[lane] = (lane < n) ? p : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  const double2* sv2 = reinterpret_cast<const double2*>(svec);
  double run = 0.0, mycum = 0.0;
#pragma unroll
  for (int j = 0; j < 64; j += 8) {
    if (j < n) {                                     // wave-uniform
      const double2 a = sv2[j / 2], b = sv2[j / 2 + 1], c = sv2[j / 2 + 2], d = sv2[j / 2 + 3];
      const double v[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        run = (j + t == 0) ? v[t] : run + v[t];
        mycum = (lane == j + t) ? run : mycum;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();                   // every lane has read the vector before it is overwritten
  if (!(run > 0.0) || isinf(run)) err |= DERR_ZERO_PROB;
  const double thr = u * run;
  const bool fail = (lane < n) && !(thr <= mycum);
  int idx = (int)__popcll(__ballot(fail));
  return idx < n ? idx : n - 1;
}

// normalisation sum of a partial-likelihood row (:525): t_g = x_g + x_{g+4} + ... (ascending), then (t_0 + t_1) + (t_2 + t_3)
__device__ __forceinline__ double coop_sum(double x, int n) {
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  for (int j = 0; j < n; j += 4) {
    t0 += readlane_f64(x, j);
    if (j + 1 < n) t1 += readlane_f64(x, j + 1);
    if (j + 2 < n) t2 += readlane_f64(x, j + 2);
    if (j + 3 < n) t3 += readlane_f64(x, j + 3);
  }
  return (t0 + t1) + (t2 + t3);
}

}  // namespace phm
