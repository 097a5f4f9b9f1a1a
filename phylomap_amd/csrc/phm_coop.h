// phm_coop.h -- wave-cooperative n-vector primitives of the 5..64-state kernels: one STATE per lane, scalars broadcast
// with v_readlane.  Summation orders are the spec's left-to-right ones (lane c accumulates while x_j is broadcast with j
// ascending; prefix sums of a probability vector are formed in index order), so results equal the oracle bit for bit.
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
  const double* row = M + c * ldn;
  double acc = row[0] * readlane_f64(v, 0);
  for (int j = 1; j < n; ++j) acc += row[j] * readlane_f64(v, j);
  return acc;
}

// the same product for a matrix with at most w non-zeros per row, kept in ELLPACK form (columns ascending; padding entries
// have value 0): lane c gathers v[col] of its own non-zeros.  Skipping the exact zeros of a row leaves its left-to-right sum
// unchanged bit for bit -- every term is a product of non-negative finite numbers, so the skipped terms are +0.
__device__ __forceinline__ double coop_matvec_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval, double v,
                                                  int w, int c) {
  double acc = eval[c * w] * __shfl(v, ecol[c * w], 64);
  for (int t = 1; t < w; ++t) acc += eval[c * w + t] * __shfl(v, ecol[c * w + t], 64);
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

__device__ __forceinline__ double coop_sum(double x, int n) {
  double run = readlane_f64(x, 0);
  for (int j = 1; j < n; ++j) run += readlane_f64(x, j);
  return run;
}

}  // namespace phm
