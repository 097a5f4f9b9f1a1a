// phm_wtiles.hip -- 5..64 states, one lane per replica, a wavefront per (tile of 64 replicas, item); see phm_wtiles.h.
#include "phm_wtiles.h"

#include <mutex>
#include <utility>



namespace phm {

namespace {

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function object of the CURRENT device: remembered per
// (function, device) under a mutex (one process may drive several GPUs from several threads), errors handed back.
hipError_t allow_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> guard(mu);
  for (const auto& d : done) if (d.first == fn && d.second == dev) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.emplace_back(fn, dev);
  return e;
}

using d4_t = __attribute__((ext_vector_type(4))) double;

// Categorical draw of one lane from a probability vector given term by term (term(c) = p_c, c = 0 .. n-1):
// first j with u * sum(p) <= p_0 + .. + p_j, index order, unfused left-to-right sums (categorical draw).
// Two passes over the terms -- the total, then the running sum against the threshold -- so nothing is kept per state.
template <class Term>
__device__ __forceinline__ int sample_terms(int n, double u, Term term, uint32_t& err) {
  double total = 0.0;                                  // 0 + p_0 = p_0 exactly
  for (int c = 0; c < n; ++c) total += term(c);
  if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
  const double thr = u * total;
  double cum = 0.0;
  int idx = 0;
  // u < 1: the last partial sum (= total) always passes, so n - 1 comparisons decide (sample_cat of phm_device.h)
  for (int c = 0; c < n - 1; ++c) {
    cum += term(c);
    const bool past = !(thr <= cum);
    idx += past ? 1 : 0;
    if ((c & 7) == 7 && !__any(past)) break;           // every lane of the wave has found its state
  }
  return idx;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pruning, one HEIGHT level: PL[parent] = (Bc^(m1-1) PL[c1]) (.) (Bc^(m0-1) PL[c0]), optionally / sum   (:503-529).
// A wave owns 16 replicas of a tile at one node.  MFMA fragment maps (cdna_hip_programming.md, f64 16x16x4):
//   A: lane l holds A[row = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][col = l & 15];
//   C/D: col = l & 15, row = (l >> 4) + 4 * reg.
// With M = output state, K = input state, N = replica, register q of row block i of an accumulator holds state
// 16 i + (l >> 4) + 4 q = 4 (4 i + q) + (l >> 4): exactly the B-operand slice of k-step 4 i + q.  So X <- Bc X is
// MT * 4 MT MFMAs and the result feeds the next step as it stands; each output entry is the fused chain
// fma(Bc[r][j], x[j], acc) with j ascending from +0 (k-steps in order, four k per MFMA in order).
// ---------------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(WT_BLOCK) void wt_up_kernel(WtParams p, int begin, int end) {
  constexpr int KS = 4 * MT;
  extern __shared__ __align__(16) double s_scr_all[];  // [wave][n][64]: the first child's vectors while the second child's chains run
  __shared__ uint8_t s_perm_all[WT_BLOCK / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int n = p.n_states, ldt = p.ldt;
  double* s_scr = s_scr_all + (size_t)wave * n * 64;      // n rows, not 16 MT (C5: pruning 21.0 -> 20.4 ms per sweep)
  uint8_t* s_perm = s_perm_all[wave];
  double Af[MT][KS];                                   // the chain matrix as A-operand fragments, for the whole launch
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int row = 16 * i + lr, k = 4 * s + lk;
      Af[i][s] = (row < n && k < n) ? p.Bc[row * n + k] : 0.0;
    }
  const int n_lvl = end - begin;
  const int64_t items = (int64_t)n_lvl * p.n_tiles;
  const int ks_used = (n + 3) >> 2;                    // wave-uniform: fma(0, 0, acc) = acc, so the skipped steps change nothing
  uint32_t err = 0;
  for (int64_t item = (int64_t)blockIdx.x * (WT_BLOCK / 64) + wave; item < items; item += (int64_t)gridDim.x * (WT_BLOCK / 64)) {
    const int tile = (int)(item % p.n_tiles), li = (int)(item / p.n_tiles);
    const UpStep st = p.up[p.up_order[begin + li]];
    double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
    const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
    const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;

    // The 64 replicas of the tile are dealt to the four 16-column MFMA blocks in the order of their chain lengths
    // (a block runs to the longest chain of its columns, so equal lengths side by side waste the fewest steps):
    // s_perm[r] = the lane (replica) of rank r; returns this lane's own chain length.
    auto sort_by_chain = [&](int edge) -> int {
      const int k = (int)mct[edge * 64 + lane] - 1;
      int rank = 0;
      for (int t = 0; t < 64; ++t) {
        const int kt = __builtin_amdgcn_readlane(k, t);
        rank += (kt < k || (kt == k && t < lane)) ? 1 : 0;
      }
      s_perm[rank] = (uint8_t)lane;
      return k;
    };
    // the child's vectors of replica column j as B-operand tiles
    auto load_x = [&](int child, int j, d4_t (&X)[MT]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          X[i][q] = (row < n) ? PLt[((size_t)child * n + row) * 64 + j] : 0.0;
        }
    };
    // Bc^kj applied to them: kmax steps on the matrix cores, column j keeps the product of step kj
    auto run_chain = [&](d4_t (&X)[MT], int kj, d4_t (&R)[MT]) {
#pragma unroll
      for (int i = 0; i < MT; ++i) R[i] = X[i];
      const int kmax = wave_max_count(kj);
      for (int step = 1; step <= kmax; ++step) {
        d4_t Y[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KS; ++s)                 // k-steps whose four input states are all >= n multiply zeros: skipped
            if (s < ks_used) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[i][s], X[s >> 2][s & 3], acc, 0, 0, 0);
          Y[i] = acc;
        }
        const bool mine = (kj == step);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          X[i] = Y[i];
#pragma unroll
          for (int q = 0; q < 4; ++q) R[i][q] = mine ? Y[i][q] : R[i][q];
        }
      }
    };
    // a tip child: a row of the chain table (the chain run from a unit vector / the parity mask)
    auto tipvec = [&](int child, int edge, int j, d4_t (&R)[MT]) {
      const int tip = ~child;
      const int ts = p.tips_per_replica ? tips_t[tip * 64 + j] : p.tips[tip];
      int k = (int)mct[edge * 64 + j] - 1;
      if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
      const double* __restrict__ src = p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * n + ts) * ldt;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          R[i][q] = (row < n) ? src[row] : 0.0;
        }
    };
    // PL[parent] = first (.) second (:510), / sum (:525), for replica column j
    auto finish = [&](int j, const d4_t (&R0)[MT], const d4_t (&R1)[MT]) {
      d4_t P[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) P[i] = R0[i] * R1[i];
      if (p.normalise) {
        double t = 0.0;                                // states lk, lk + 4, lk + 8, ... ascending: partial sum t_lk
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) t += P[i][q];
        t = t + __shfl_xor(t, 16, 64);                 // t_0 + t_1 | t_2 + t_3
        t = t + __shfl_xor(t, 32, 64);                 // (t_0 + t_1) + (t_2 + t_3)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) P[i][q] = P[i][q] / t;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          if (row < n) PLt[((size_t)st.parent * n + row) * 64 + j] = P[i][q];
        }
    };

    // "first" = child[1] (:508), "second" = child[0] (:509)
    const bool int_first = st.child[1] >= 0, int_second = st.child[0] >= 0;
    if (!int_first && !int_second) {
      for (int nt = 0; nt < 4; ++nt) {
        const int j = 16 * nt + lr;
        d4_t R0[MT], R1[MT];
        tipvec(st.child[1], st.edge[1], j, R0);
        tipvec(st.child[0], st.edge[0], j, R1);
        finish(j, R0, R1);
      }
    } else if (int_first != int_second) {
      // One wave per SIMD (the fragments of Bc take 128 registers): nothing else hides a load, so the next block's vectors and
      // this block's tip rows are requested BEFORE the chain of this block starts and arrive while the matrix cores run.
      const int ci = int_first ? 1 : 0;                // the internal child
      const int k = sort_by_chain(st.edge[ci]);
      d4_t X[MT], Xn[MT];
      int j = s_perm[lr];
      load_x(st.child[ci], j, X);
      for (int nt = 0; nt < 4; ++nt) {
        const int kj = __shfl(k, j, 64);
        const int jn = (nt < 3) ? s_perm[16 * (nt + 1) + lr] : j;
        if (nt < 3) load_x(st.child[ci], jn, Xn);
        d4_t Rc[MT], Rt[MT];
        tipvec(st.child[1 - ci], st.edge[1 - ci], j, Rt);
        run_chain(X, kj, Rc);
        if (int_first) finish(j, Rc, Rt); else finish(j, Rt, Rc);
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = Xn[i];
        j = jn;
      }
    } else {
      const int k1 = sort_by_chain(st.edge[1]);
      d4_t X[MT], Xn[MT];
      int j = s_perm[lr];
      load_x(st.child[1], j, X);
      for (int nt = 0; nt < 4; ++nt) {
        const int kj = __shfl(k1, j, 64);
        const int jn = (nt < 3) ? s_perm[16 * (nt + 1) + lr] : j;
        if (nt < 3) load_x(st.child[1], jn, Xn);
        d4_t R0[MT];
        run_chain(X, kj, R0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = 16 * i + lk + 4 * q; if (row < n) s_scr[row * 64 + j] = R0[i][q]; }
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = Xn[i];
        j = jn;
      }
      const int k0 = sort_by_chain(st.edge[0]);
      j = s_perm[lr];
      load_x(st.child[0], j, X);
      for (int nt = 0; nt < 4; ++nt) {
        const int kj = __shfl(k0, j, 64);
        const int jn = (nt < 3) ? s_perm[16 * (nt + 1) + lr] : j;
        if (nt < 3) load_x(st.child[0], jn, Xn);
        d4_t R0[MT], R1[MT];
        run_chain(X, kj, R1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = 16 * i + lk + 4 * q; R0[i][q] = (row < n) ? s_scr[row * 64 + j] : 0.0; }
        finish(j, R0, R1);
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = Xn[i];
        j = jn;
      }
    }
  }
  if (err) atomicOr(p.err, err);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel for 33 .. 64 states with TWO waves per SIMD.  wt_up_kernel keeps the chain matrix in registers as A-operand
// fragments -- 128 registers at 61 states, which with the vectors leaves room for one wave per SIMD: while that wave divides
// (:525), sorts its columns or waits for a child's vectors the matrix core of its SIMD idles (C4: 51 % of the f64 matrix
// peak over the pruning launches, profiles/r04_pmc_C4bf_summary.json).  Here the fragments sit in LDS (32 KB per workgroup,
// lane-contiguous: one conflict-free ds_read_b64 per MFMA, 256 LDS cycles per chain step against 4 096 on the matrix core), and the
// vectors of the first child of a node with two internal children wait in the PARENT's row of PL (written, fenced, read back by
// the same wave: it is about to be overwritten by the product anyway) instead of 31 KB of LDS per wave.  Same MFMAs on the same
// operands in the same order: same bits.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef PHM_UP2_WAVES
#define PHM_UP2_WAVES 2
#endif
constexpr int WT_UP2_WAVES = PHM_UP2_WAVES;         // waves per SIMD the kernel is compiled for beyond 32 states (17 .. 32 states: four)
#ifndef PHM_UP2_WAVES_SMALL
#define PHM_UP2_WAVES_SMALL 3
#endif
constexpr int wt_up2_waves(int mt) { return mt <= 2 ? PHM_UP2_WAVES_SMALL : WT_UP2_WAVES; }
template <int MT, int KSU>                          // KSU = ceil(n / 4): the k-steps that hold a state (the others multiply zeros: skipped)
__global__ __launch_bounds__(WT_BLOCK, wt_up2_waves(MT)) void wt_up2_kernel(WtParams p, int begin, int end) {
  constexpr int KS = 4 * MT;
  __shared__ double s_A[MT * KS * 64];               // fragment (i, s): lane l holds Bc[16 i + (l & 15)][4 s + (l >> 4)]
  __shared__ uint8_t s_perm_all[WT_BLOCK / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int n = p.n_states, ldt = p.ldt;
  uint8_t* s_perm = s_perm_all[wave];
  for (int f = wave; f < MT * KS; f += WT_BLOCK / 64) {
    const int row = 16 * (f / KS) + lr, k = 4 * (f % KS) + lk;
    s_A[f * 64 + lane] = (row < n && k < n) ? p.Bc[row * n + k] : 0.0;
  }
  __syncthreads();
  const int n_lvl = end - begin;
  const int64_t items = (int64_t)n_lvl * p.n_tiles;
  uint32_t err = 0;
  for (int64_t item = (int64_t)blockIdx.x * (WT_BLOCK / 64) + wave; item < items; item += (int64_t)gridDim.x * (WT_BLOCK / 64)) {
    const int tile = (int)(item % p.n_tiles), li = (int)(item / p.n_tiles);
    const UpStep st = p.up[p.up_order[begin + li]];
    double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
    const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
    const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;

    // rank of (chain length, lane) as in wt_up_kernel; the lengths of a tile span a few values, so the ranks come from one ballot
    // per VALUE (a counting sort: lanes of equal length keep their order) instead of one comparison per LANE
    auto sort_by_chain = [&](int edge) -> int {
      const int k = (int)mct[edge * 64 + lane] - 1;
      const int khi = wave_max_count(k), klo = 65535 - wave_max_count(65535 - k);
      int rank = 0;
      if (khi - klo < 40) {
        int base = 0;
        for (int v = klo; v <= khi; ++v) {
          const unsigned long long mask = __ballot(k == v);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          rank = (k == v) ? base + below : rank;
          base += (int)__popcll(mask);
        }
      } else {
        for (int t = 0; t < 64; ++t) {
          const int kt = __builtin_amdgcn_readlane(k, t);
          rank += (kt < k || (kt == k && t < lane)) ? 1 : 0;
        }
      }
      s_perm[rank] = (uint8_t)lane;
      return k;
    };
    auto load_x = [&](int node, int j, d4_t (&X)[MT]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          X[i][q] = (row < n) ? PLt[((size_t)node * n + row) * 64 + j] : 0.0;
        }
    };
    // Bc^kj applied to the vectors in X, in place: kmax steps on the matrix cores, the lanes of column j stop taking the products
    // after step kj (a lane's four-state slices all belong to ONE column, l & 15: the finished column rides along unchanged)
    auto run_chain = [&](d4_t (&X)[MT], int kj) {
      const int kmax = wave_max_count(kj);
      // the next EIGHT fragments are requested before the eight MFMAs at hand are issued (a dependent MFMA holds the wave's
      // instruction stream until its predecessor is done: a read issued behind it would wait out its LDS latency with the matrix
      // core idle); fragment f of the flattened order belongs to row block f / KSU, k-step f % KSU
      constexpr int NF = MT * KSU, CH = 8, NCH = (NF + CH - 1) / CH;
      double af[2][CH];
      auto fetch = [&](int c, double (&a)[CH]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int f = c * CH + u;
          if (f < NF) a[u] = s_A[((f / KSU) * KS + f % KSU) * 64 + lane];
        }
      };
      for (int step = 1; step <= kmax; ++step) {
        __asm__ volatile("" ::: "memory");             // the fragments are read from LDS in every step (hoisted, they would take the registers back)
        d4_t Y[MT];
        fetch(0, af[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          __builtin_amdgcn_sched_barrier(0);
          if (c + 1 < NCH) fetch(c + 1, af[(c + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int f = c * CH + u;
            if (f < NF) {
              const int i = f / KSU, ks = f % KSU;
              if (ks == 0) Y[i] = d4_t{0.0, 0.0, 0.0, 0.0};
              Y[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[c & 1][u], X[ks >> 2][ks & 3], Y[i], 0, 0, 0);
            }
          }
        }
        const bool live = step <= kj;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) X[i][q] = live ? Y[i][q] : X[i][q];
      }
    };
    auto tipvec = [&](int child, int edge, int j, d4_t (&R)[MT]) {
      const int tip = ~child;
      const int ts = p.tips_per_replica ? tips_t[tip * 64 + j] : p.tips[tip];
      int k = (int)mct[edge * 64 + j] - 1;
      if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
      const double* __restrict__ src = p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * n + ts) * ldt;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          R[i][q] = (row < n) ? src[row] : 0.0;
        }
    };
    auto store = [&](int j, const d4_t (&P)[MT]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          if (row < n) PLt[((size_t)st.parent * n + row) * 64 + j] = P[i][q];
        }
    };
    auto finish = [&](int j, const d4_t (&R0)[MT], const d4_t (&R1)[MT]) {      // first (.) second (:510), / sum (:525)
      d4_t P[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) P[i] = R0[i] * R1[i];
      if (p.normalise) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) t += P[i][q];
        t = t + __shfl_xor(t, 16, 64);
        t = t + __shfl_xor(t, 32, 64);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) P[i][q] = P[i][q] / t;
      }
      store(j, P);
    };

    const bool int_first = st.child[1] >= 0, int_second = st.child[0] >= 0;
    if (!int_first && !int_second) {
      for (int nt = 0; nt < 4; ++nt) {
        const int j = 16 * nt + lr;
        d4_t R0[MT], R1[MT];
        tipvec(st.child[1], st.edge[1], j, R0);
        tipvec(st.child[0], st.edge[0], j, R1);
        finish(j, R0, R1);
      }
    } else if (int_first != int_second) {
      const int ci = int_first ? 1 : 0;                // the internal child
      const int k = sort_by_chain(st.edge[ci]);
      for (int nt = 0; nt < 4; ++nt) {
        const int j = s_perm[16 * nt + lr];
        const int kj = __shfl(k, j, 64);
        d4_t X[MT], Rt[MT];
        load_x(st.child[ci], j, X);
        run_chain(X, kj);
        tipvec(st.child[1 - ci], st.edge[1 - ci], j, Rt);
        if (int_first) finish(j, X, Rt); else finish(j, Rt, X);
      }
    } else {
      const int k1 = sort_by_chain(st.edge[1]);
      for (int nt = 0; nt < 4; ++nt) {
        const int j = s_perm[16 * nt + lr];
        const int kj = __shfl(k1, j, 64);
        d4_t X[MT];
        load_x(st.child[1], j, X);
        run_chain(X, kj);
        store(j, X);                                   // waits in the parent's row
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      const int k0 = sort_by_chain(st.edge[0]);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int nt = 0; nt < 4; ++nt) {
        const int j = s_perm[16 * nt + lr];
        const int kj = __shfl(k0, j, 64);
        d4_t X[MT], R0[MT];
        load_x(st.child[0], j, X);
        run_chain(X, kj);
        load_x(st.parent, j, R0);
        finish(j, R0, X);
      }
    }
  }
  if (err) atomicOr(p.err, err);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pruning through a BANDED chain matrix (tridiagonal amino-acid-style or count-valued Q: half-bandwidth 1; the hidden-rates
// Q of make2sQ: 2) -- what SPARSEmakePLrcpp / spmmmmvFORpl (src/phylomap.cpp:490-501, :451-457) exploit through sp_mat.
// One LANE per replica, a wave per (node, tile): the lane keeps its child vector in registers and applies
//   y_i = fma(Bc[i][j], x_j, acc),  j = i - HB .. i + HB ascending, from +0
// in place.  The dense specification adds, in the same order, terms whose coefficient is an exact zero: fma(0, x_j, acc) =
// acc for the finite non-negative x of a partial likelihood, so the two give the same bits (and so do in-band zeros, which
// are multiplied).  The coefficients are kernel-argument constants (scalar loads).  A lane stops after its own m - 1 steps
// (exec mask), the wave runs to its longest chain.  At 20 states a chain step is 58 FMAs per replica where the matrix-core
// kernel issues 40 padded 16 x 16 x 4 MFMAs per 64 replicas = 16x the flops, and on this chip vector and matrix FP64 peak
// are the same number.
// ---------------------------------------------------------------------------------------------------------------------
template <int NP, int HB>
__device__ __forceinline__ void wt_up_band_node(const WtParams& p, const WtBand& bd, int tile, const UpStep& st, int lane, uint32_t& err) {
  constexpr int W = 2 * HB + 1;
  const int n = p.n_states, ldt = p.ldt;
  double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
  const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  double R[2][NP];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {                     // ch 0: "first" = child[1] (:508); ch 1: "second" = child[0] (:509)
    const int child = st.child[1 - ch], edge = st.edge[1 - ch];
    int k = (int)mct[edge * 64 + lane] - 1;
    if (child < 0) {                                   // tip: a row of the chain table (the chain run from a unit vector / the parity mask)
      const int tip = ~child;
      const int ts = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];
      if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
      const double2* __restrict__ src = reinterpret_cast<const double2*>(p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * n + ts) * ldt);
#pragma unroll
      for (int i = 0; i < NP; i += 2) {                // rows are 16-byte aligned and padded to an even length with zeros
        double2 v = {0.0, 0.0};
        if (i < n) v = src[i >> 1];
        R[ch][i] = v.x; R[ch][i + 1] = v.y;
      }
    } else {
      double (&x)[NP] = R[ch];
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = (i < n) ? PLt[((size_t)child * n + i) * 64 + lane] : 0.0;
      const int kmax = wave_max_count(k);
      for (int step = 1; step <= kmax; ++step) {
        if (step <= k) {                               // x <- Bc x for the lanes still inside their chain
          double prev[HB];
#pragma unroll
          for (int d = 0; d < HB; ++d) prev[d] = 0.0;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < HB; ++d) if (i - HB + d >= 0) acc = __builtin_fma(bd.c[i * W + d], prev[d], acc);
            acc = __builtin_fma(bd.c[i * W + HB], x[i], acc);
#pragma unroll
            for (int d = 1; d <= HB; ++d) if (i + d < NP) acc = __builtin_fma(bd.c[i * W + HB + d], x[i + d], acc);
#pragma unroll
            for (int d = 0; d + 1 < HB; ++d) prev[d] = prev[d + 1];
            prev[HB - 1] = x[i];
            x[i] = acc;
          }
        }
      }
    }
  }
  double P[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) P[i] = R[0][i] * R[1][i];                      // :510
  if (p.normalise) {                                                          // :525; four interleaved partial sums
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i & 3] += P[i];
    const double tt = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
    for (int i = 0; i < NP; ++i) P[i] = P[i] / tt;
  }
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (i < n) PLt[((size_t)st.parent * n + i) * 64 + lane] = P[i];
}

template <int NP, int HB>
__global__ __launch_bounds__(WT_BLOCK) void wt_up_band_kernel(WtParams p, WtBand bd, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int n_lvl = end - begin;
  const int64_t item = (int64_t)blockIdx.x * (WT_BLOCK / 64) + (threadIdx.x >> 6);
  if (item >= (int64_t)n_lvl * p.n_tiles) return;
  const int tile = (int)(item % p.n_tiles), li = (int)(item / p.n_tiles);
  const UpStep st = p.up[p.up_order[begin + li]];
  uint32_t err = 0;
  wt_up_band_node<NP, HB>(p, bd, tile, st, lane, err);
  if (err) atomicOr(p.err, err);
}

// DEEP trees (a ladder-like phylogeny: thousands of height levels, a launch per level and pass would be all the sweep does --
// 2 000-tip ladder, 8 states, 4 096 replicas: 19.9 of 21.9 ms): the same step over one TIER of subtree clusters
// (phm_sched.h ClusterPlan), a workgroup per (cluster, tile) walking the cluster's levels with a workgroup barrier between them,
// as tiles_up_cluster_kernel does for 2 .. 4 states.
constexpr int WT_CL_BLOCK = 512;
template <int NP, int HB>
__global__ __launch_bounds__(WT_CL_BLOCK) void wt_up_band_cluster_kernel(WtParams p, WtBand bd, int cl_begin, int n_cl) {
  constexpr int NW = WT_CL_BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = cl_begin + (int)(blockIdx.x % (unsigned)n_cl);
  const int tile = (int)(blockIdx.x / (unsigned)n_cl);
  const int l0 = p.cl_lvl_ptr[cl], l1 = p.cl_lvl_ptr[cl + 1] - 1;
  uint32_t err = 0;
  for (int l = l0; l < l1; ++l) {
    const int i1 = p.cl_lvl_off[l + 1];
    for (int i = p.cl_lvl_off[l] + wave; i < i1; i += NW) {
      const ClusterNode nd = p.cl_nodes[i];
      UpStep st;
      st.parent = nd.parent; st.child[0] = nd.child[0]; st.child[1] = nd.child[1]; st.edge[0] = nd.edge[0]; st.edge[1] = nd.edge[1];
      wt_up_band_node<NP, HB>(p, bd, tile, st, lane, err);
    }
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// The same pruning step for FEW tiles: a wave per (node, tile, 16-replica block), blocks in replica order, no LDS.  With one
// tile a height level of the kernel above is a single wave working through four blocks and two children (C4: 168 us per level,
// 84 % of a 4.6 ms sweep); here the four blocks run on four waves.  With many tiles the sorted blocks of the kernel above win
// (a block runs to the longest of its 16 chains: 31 against 38 ms per sweep on C4 at 65 536 replicas); measured crossover:
// 128 tiles on C4, about 80 on C5 (profiles/r02_probe_few_tiles.log).  Used for n <= 16 (one row block); with two or more
// row blocks the row-split workgroups of wt_up_msplit_kernel are faster still.
template <int MT>
__global__ __launch_bounds__(WT_BLOCK) void wt_up_blocks_kernel(WtParams p, int begin, int end) {
  constexpr int KS = 4 * MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int n = p.n_states, ldt = p.ldt;
  double Af[MT][KS];                                   // the chain matrix as A-operand fragments, for the whole launch
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int row = 16 * i + lr, k = 4 * s + lk;
      Af[i][s] = (row < n && k < n) ? p.Bc[row * n + k] : 0.0;
    }
  const int n_lvl = end - begin;
  const int64_t items = (int64_t)n_lvl * p.n_tiles * 4;
  const int ks_used = (n + 3) >> 2;                    // k-steps whose four input states are all >= n multiply zeros: skipped
  uint32_t err = 0;
  for (int64_t item = (int64_t)blockIdx.x * (WT_BLOCK / 64) + wave; item < items; item += (int64_t)gridDim.x * (WT_BLOCK / 64)) {
    const int nt = (int)(item & 3);
    const int64_t q4 = item >> 2;
    const int tile = (int)(q4 % p.n_tiles), li = (int)(q4 / p.n_tiles);
    const UpStep st = p.up[p.up_order[begin + li]];
    const int j = 16 * nt + lr;                        // this lane's replica within the tile
    double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
    const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
    const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
    d4_t R[2][MT];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {                   // ch 0: "first" = child[1] (:508); ch 1: "second" = child[0] (:509)
      const int child = st.child[1 - ch], edge = st.edge[1 - ch];
      int k = (int)mct[edge * 64 + j] - 1;
      if (child < 0) {                                 // tip: a row of the chain table (the chain run from a unit vector)
        const int tip = ~child;
        const int ts = p.tips_per_replica ? tips_t[tip * 64 + j] : p.tips[tip];
        if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
        const double* __restrict__ src = p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * n + ts) * ldt;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * i + lk + 4 * q;
            R[ch][i][q] = (row < n) ? src[row] : 0.0;
          }
      } else {                                         // internal child: the chain itself, on the matrix cores
        d4_t X[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * i + lk + 4 * q;
            X[i][q] = (row < n) ? PLt[((size_t)child * n + row) * 64 + j] : 0.0;
          }
#pragma unroll
        for (int i = 0; i < MT; ++i) R[ch][i] = X[i];
        const int kmax = wave_max_count(k);            // segment counts differ per replica: run to the longest, keep step k
        for (int step = 1; step <= kmax; ++step) {
          d4_t Y[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
              if (s < ks_used) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[i][s], X[s >> 2][s & 3], acc, 0, 0, 0);
            Y[i] = acc;
          }
          const bool mine = (k == step);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            X[i] = Y[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) R[ch][i][q] = mine ? Y[i][q] : R[ch][i][q];
          }
        }
      }
    }
    d4_t P[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) P[i] = R[0][i] * R[1][i];                     // :510
    if (p.normalise) {                                                         // :525
      double t = 0.0;                                  // states lk, lk + 4, lk + 8, ... ascending: partial sum t_lk
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) t += P[i][q];
      t = t + __shfl_xor(t, 16, 64);                   // t_0 + t_1 | t_2 + t_3
      t = t + __shfl_xor(t, 32, 64);                   // (t_0 + t_1) + (t_2 + t_3)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) P[i][q] = P[i][q] / t;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * i + lk + 4 * q;
        if (row < n) PLt[((size_t)st.parent * n + row) * 64 + j] = P[i][q];
      }
  }
  if (err) atomicOr(p.err, err);
}

// The pruning step for FEW tiles, latency form: a WORKGROUP per (node, tile, 16-replica block), wave w of it owning row block
// w (16 output states) of every product.  A chain step is then 4 MT MFMAs per wave instead of 4 MT^2 -- a third of the
// latency at 61 states -- at the price of passing the vectors through LDS between steps (the B-operand slices of a step are
// the rows all waves wrote in the step before) and of a barrier per step; the fragments of Bc a wave keeps shrink to its own
// row block (16 registers pairs at 61 states).  Same products, same order of additions, same bits as the kernels above.
template <int MT>
__global__ __launch_bounds__(64 * MT) void wt_up_msplit_kernel(WtParams p, int begin, int end) {
  constexpr int KS = 4 * MT, NP = 16 * MT;
  __shared__ double sX[2][NP * 16];                     // [buffer][state][replica column]
  __shared__ double sP[NP * 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int n = p.n_states, ldt = p.ldt;
  double Af[KS];                                        // row block w of the chain matrix as A-operand fragments
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int row = 16 * w + lr, k = 4 * s + lk;
    Af[s] = (row < n && k < n) ? p.Bc[row * n + k] : 0.0;
  }
  const int n_lvl = end - begin;
  const int64_t items = (int64_t)n_lvl * p.n_tiles * 4;
  const int ks_used = (n + 3) >> 2;
  uint32_t err = 0;
  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int nt = (int)(item & 3);
    const int64_t q4 = item >> 2;
    const int tile = (int)(q4 % p.n_tiles), li = (int)(q4 / p.n_tiles);
    const UpStep st = p.up[p.up_order[begin + li]];
    double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
    const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
    const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
    // (blocks in replica order: dealing the replicas to the blocks by chain length, as the per-tile kernel does, was measured here
    //  too -- the rank computation and the 8-byte gathers it forces cost more than the shorter chains return: 0.66 vs 0.77 G/s on C4
    //  at 4 096 replicas)
    const int j = 16 * nt + lr;                        // this lane's replica within the tile (the same in every wave)
    d4_t R[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {                   // ch 0: "first" = child[1] (:508); ch 1: "second" = child[0] (:509)
      const int child = st.child[1 - ch], edge = st.edge[1 - ch];
      int k = (int)mct[edge * 64 + j] - 1;
      if (child < 0) {
        const int tip = ~child;
        const int ts = p.tips_per_replica ? tips_t[tip * 64 + j] : p.tips[tip];
        if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
        const double* __restrict__ src = p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * n + ts) * ldt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * w + lk + 4 * q;
          R[ch][q] = (row < n) ? src[row] : 0.0;
        }
      } else {
        d4_t x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * w + lk + 4 * q;
          x[q] = (row < n) ? PLt[((size_t)child * n + row) * 64 + j] : 0.0;
          sX[0][row * 16 + lr] = x[q];
        }
        R[ch] = x;
        const int kmax = wave_max_count(k);            // the same 16 chain lengths in every wave of the workgroup
        __syncthreads();
        for (int step = 1; step <= kmax; ++step) {
          const int cur = (step - 1) & 1;
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KS; ++s)
            if (s < ks_used) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[s], sX[cur][(4 * s + lk) * 16 + lr], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) sX[cur ^ 1][(16 * w + lk + 4 * q) * 16 + lr] = acc[q];
          if (k == step) R[ch] = acc;
          __syncthreads();
        }
      }
    }
    d4_t P = R[0] * R[1];                                                       // :510
    if (p.normalise) {                                                         // :525
#pragma unroll
      for (int q = 0; q < 4; ++q) sP[(16 * w + lk + 4 * q) * 16 + lr] = P[q];
      __syncthreads();
      double t = 0.0;                                  // states lk, lk + 4, lk + 8, ... ascending: partial sum t_lk
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) t += sP[(16 * i + lk + 4 * q) * 16 + lr];
      t = t + __shfl_xor(t, 16, 64);
      t = t + __shfl_xor(t, 32, 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) P[q] = P[q] / t;
      __syncthreads();                                 // sP is rewritten by the next item
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * w + lk + 4 * q;
      if (row < n) PLt[((size_t)st.parent * n + row) * 64 + j] = P[q];
    }
  }
  if (err) atomicOr(p.err, err);
}

__global__ __launch_bounds__(WT_BLOCK) void wt_root_kernel(WtParams p, int it) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (WT_BLOCK / 64) + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;
  const int n = p.n_states;
  const double* __restrict__ PLr = p.PL + ((size_t)tile * p.n_node + p.root) * n * 64 + lane;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  uint32_t err = 0;
  const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
  const int rs = sample_terms(n, u, [&](int c) { return p.pid[c] * PLr[(size_t)c * 64]; }, err);   // :618, :627
  p.nstate[((size_t)tile * p.n_node + p.root) * 64 + lane] = (uint8_t)rs;
  if (err) atomicOr(p.err, err);
}

// child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their parity mask (:1384-1397).
// The draw needs the total of its probability vector before it can walk the running sum.  Through round 3 that was two passes over
// the child's partial likelihoods and the lane's row of the chain table (the total, then the walk); both kernels below make ONE
// pass and keep the running sums cum_c (the sampler's own left-to-right sums: cum_c = cum_{c-1} + p_c from +0) for every state; the
// state is then the number of c <= n - 2 with !(thr <= cum_c) -- the comparisons of the two-pass walk on the same numbers.
// n > 32 (a two-pass form reads 2 x 31 KB per wave and gathers 2 x 30 sixteen-byte pieces per lane from the row table -- one
// cache-line look-up per lane and piece: the gathers, not HBM, set its pace): the running sums are kept for every state -- states
// 0 .. 31 in LDS ([state][lane], 16 KB per wave), the rest in registers, so that nothing
// of the first half occupies registers while the second half's loads are in flight (144 registers: three waves per SIMD, ten per
// CU by LDS) -- and the state is the number of c <= n - 2 with !(thr <= cum_c): the comparisons of the two-pass walk on the same
// numbers.  A wave is a workgroup of its own.  C4 at 65 536 replicas: two passes (32 states per round of loads) 6.7 ms per sweep;
// two passes with the second one confined to the 16-state block the threshold falls into 5.5; one pass with all 122 values in
// registers (one wave per SIMD) 6.2; this form 4.5 (profiles/r04_probe_c4_variants.log).
template <int MT>
__global__ __launch_bounds__(64) void wt_down1_kernel(WtParams p, int it, int begin, int end) {
  constexpr int NP = 16 * MT;
  static_assert(MT >= 3, "one-pass node draws: n > 32");
  extern __shared__ double s_cum[];                    // [32][64]: running sums after states 0 .. 31
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int n_lvl = end - begin;
  const int n = p.n_states, ldt = p.ldt;
  const int tile = item / n_lvl;
  const DownStep ds = p.down[p.down_order[begin + item % n_lvl]];
  const int b = ds.edge;
  const double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  const int m = p.mcount[((size_t)tile * p.n_edge + b) * 64 + lane];
  const int ps = nst[ds.parent * 64 + lane];
  uint32_t err = 0;
  int cs;
  if (ds.child >= 0 || p.tip_masks) {
    int kk = m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double2* __restrict__ src = reinterpret_cast<const double2*>(p.rowL + ((size_t)kk * n + ps) * ldt);
    uint32_t node_id;
    const bool internal = ds.child >= 0;
    const double* __restrict__ PLc = PLt + (size_t)(internal ? ds.child : 0) * n * 64 + lane;
    int par = 0;
    if (internal) node_id = (uint32_t)(ds.child + p.n_tips);
    else { const int tip = ~ds.child; par = (p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip]) & 1; node_id = (uint32_t)tip; }
    const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | node_id, 0);
    // First half (states 0 .. 31): loads, products, running sums -> LDS ([state][lane], 16 KB); second half: the same with the sums
    // left in registers.  Nothing of the first half stays in registers while the second half's loads are in flight.
    double cum = 0.0;
    {
      double2 r[16];
      double pl[32];
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = src[j];                                   // n > 32: all inside the row
#pragma unroll
      for (int j = 0; j < 32; ++j) pl[j] = internal ? PLc[(size_t)j * 64] : (((j & 1) == par) ? 1.0 : 0.0);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        cum += ((j & 1) ? r[j >> 1].y : r[j >> 1].x) * pl[j];
        s_cum[j * 64 + lane] = cum;
      }
    }
    double hi[NP - 32];
    {
      double2 r[(NP - 32) / 2];
#pragma unroll
      for (int j = 0; j < (NP - 32) / 2; ++j) { r[j].x = 0.0; r[j].y = 0.0; if (32 + 2 * j < n) r[j] = src[16 + j]; }      // rows are padded to ldt (even) with zeros
#pragma unroll
      for (int j = 0; j < NP - 32; ++j) { const int c = 32 + j; hi[j] = internal ? ((c < n) ? PLc[(size_t)c * 64] : 0.0) : (((c & 1) == par) ? 1.0 : 0.0); }
#pragma unroll
      for (int j = 0; j < NP - 32; ++j) {
        const double pr = (32 + j < n) ? ((j & 1) ? r[j >> 1].y : r[j >> 1].x) * hi[j] : 0.0;
        cum += pr;
        hi[j] = cum;
      }
    }
    const double total = cum;
    if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
    const double thr = u * total;
    cs = 0;
#pragma unroll
    for (int j = 0; j < NP - 32; ++j) cs += (32 + j < n - 1 && !(thr <= hi[j])) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) cs += !(thr <= s_cum[j * 64 + lane]) ? 1 : 0;     // n > 32: states 0 .. 31 are all <= n - 2
    if (internal) nst[ds.child * 64 + lane] = (uint8_t)cs;                                   // :655
  } else {
    const int tip = ~ds.child;
    cs = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];            // :612
  }
  p.estate[((size_t)tile * p.n_edge + b) * 64 + lane] = (uint16_t)(ps | (cs << 8));   // updatenodestates :460-475
  if (err) atomicOr(p.err, err);
}

// n <= 32: every running sum in registers (NP = n rounded up to a multiple of four; 98 registers at 20 states), no LDS, a wave per
// (tile, edge), four waves per workgroup.  C5: 4.05 -> 3.49 ms per sweep against the two-pass form (55 registers, eight waves per SIMD).
template <int NP>
__device__ __forceinline__ void wt_down1r_edge(const WtParams& p, int it, int tile, const DownStep& ds, int lane, uint32_t& err) {
  const int n = p.n_states, ldt = p.ldt;
  const int b = ds.edge;
  const double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * n * 64;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  const int m = p.mcount[((size_t)tile * p.n_edge + b) * 64 + lane];
  const int ps = nst[ds.parent * 64 + lane];
  int cs;
  if (ds.child >= 0 || p.tip_masks) {
    int kk = m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double2* __restrict__ src = reinterpret_cast<const double2*>(p.rowL + ((size_t)kk * n + ps) * ldt);
    uint32_t node_id;
    const bool internal = ds.child >= 0;
    const double* __restrict__ PLc = PLt + (size_t)(internal ? ds.child : 0) * n * 64 + lane;
    int par = 0;
    if (internal) node_id = (uint32_t)(ds.child + p.n_tips);
    else { const int tip = ~ds.child; par = (p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip]) & 1; node_id = (uint32_t)tip; }
    const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | node_id, 0);
    double2 r[NP / 2];
    double pl[NP];
#pragma unroll
    for (int j = 0; j < NP / 2; ++j) { r[j].x = 0.0; r[j].y = 0.0; if (2 * j < n) r[j] = src[j]; }      // rows are padded to ldt (even) with zeros
#pragma unroll
    for (int c = 0; c < NP; ++c) pl[c] = internal ? ((c < n) ? PLc[(size_t)c * 64] : 0.0) : (((c & 1) == par) ? 1.0 : 0.0);
    double cum = 0.0;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      cum += (c < n) ? ((c & 1) ? r[c >> 1].y : r[c >> 1].x) * pl[c] : 0.0;
      pl[c] = cum;
    }
    if (!(cum > 0.0) || isinf(cum)) err |= DERR_ZERO_PROB;
    const double thr = u * cum;
    cs = 0;
#pragma unroll
    for (int c = 0; c < NP; ++c) cs += (c < n - 1 && !(thr <= pl[c])) ? 1 : 0;
    if (internal) nst[ds.child * 64 + lane] = (uint8_t)cs;                                   // :655
  } else {
    const int tip = ~ds.child;
    cs = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];            // :612
  }
  p.estate[((size_t)tile * p.n_edge + b) * 64 + lane] = (uint16_t)(ps | (cs << 8));   // updatenodestates :460-475
}

template <int NP>
__global__ __launch_bounds__(WT_BLOCK) void wt_down1r_kernel(WtParams p, int it, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WT_BLOCK / 64) + (threadIdx.x >> 6);
  const int n_lvl = end - begin;
  if (item >= n_lvl * p.n_tiles) return;
  const int tile = item / n_lvl;
  const DownStep ds = p.down[p.down_order[begin + item % n_lvl]];
  uint32_t err = 0;
  wt_down1r_edge<NP>(p, it, tile, ds, lane, err);
  if (err) atomicOr(p.err, err);
}

// deep trees: the node draws over one tier of subtree clusters, top level first (a wave per (node of the level, child side));
// the root has been drawn by wt_root_kernel
template <int NP>
__global__ __launch_bounds__(WT_CL_BLOCK) void wt_down1r_cluster_kernel(WtParams p, int it, int cl_begin, int n_cl) {
  constexpr int NW = WT_CL_BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = cl_begin + (int)(blockIdx.x % (unsigned)n_cl);
  const int tile = (int)(blockIdx.x / (unsigned)n_cl);
  const int l0 = p.cl_lvl_ptr[cl], l1 = p.cl_lvl_ptr[cl + 1] - 1;
  uint32_t err = 0;
  for (int l = l1 - 1; l >= l0; --l) {
    const int i0 = p.cl_lvl_off[l], cnt = 2 * (p.cl_lvl_off[l + 1] - i0);
    for (int j = wave; j < cnt; j += NW) {
      const ClusterNode nd = p.cl_nodes[i0 + (j >> 1)];
      DownStep ds;
      ds.edge = nd.edge[j & 1]; ds.parent = nd.parent; ds.child = nd.child[j & 1];
      wt_down1r_edge<NP>(p, it, tile, ds, lane, err);
    }
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// Dynamic LDS of the branch kernel, in this order (offsets in bytes, every piece 16-byte aligned):
//   b2   : rows of B [n][ldt] (SMALL, or B2L), or its band [n][2 BAND + 1]
//   dw   : SMALL: dwell sums of the workgroup [n][64] u64
//   ct   : SMALL with few possible transitions: counts [n_slots][64] u32;  slot: pair -> slot [n*n] i16
//   dwt  : reduced output, not SMALL: dwell sums of the workgroup summed over lanes, [n][16] u64 (lane & 15 spreads the atomics)
//   pc   : reduced output, no slots: transition counts of the workgroup summed over lanes, [ncnt] u32
struct BranchLds { uint32_t b2, dw, ct, slot, dwt, pc, total; };
__host__ __device__ inline BranchLds branch_lds(int n, int ldt, int n_slots, bool small, bool b2l, int band, bool red, bool ks) {
  auto up16 = [](uint32_t v) { return (v + 15u) & ~15u; };
  BranchLds L;
  uint32_t o = 0;
  L.b2 = o; o += up16(8u * (band > 0 ? (uint32_t)n * (2 * band + 1) : (small || b2l) ? (uint32_t)n * ldt : 0u));
  L.dw = o; o += small ? 8u * n * 64 : 0u;
  L.ct = o; o += small ? up16(4u * n_slots * 64) : 0u;
  L.slot = o; o += (small && n_slots > 0) ? up16(2u * n * n) : 0u;
  const bool red_dw = red && !small, red_pc = red && n_slots == 0;
  L.dwt = o; o += red_dw ? 8u * n * 16 : 0u;
  L.pc = o; o += red_pc ? up16(4u * (ks ? n * n : n * (n - 1))) : 0u;
  L.total = o;
  return L;
}

// One branch for the 64 replicas of a tile: resamplebranchstates :264-308, shortener :44-73 (shortenerbf :997-1030),
// virtual jumps sampleabranch :391-410, dwell sums updatedwelltimes :745-757.  The two-flat-pass scheme of phm_tiles.hip;
// the states of the merged segments go through a byte row per dwell row in global memory, transition counts go straight to the tile's
// counters (integer atomics: exact in any order).  The four waves of a workgroup walk four groups of branches of the SAME
// tile.  SMALL (n <= 32): the rows of B sit in LDS and the dwell sums of the workgroup are collected in one LDS table
// (64-bit fixed point, ds_add_u64) that is handed to the tile's accumulators once, as coalesced rows -- a scattered atomic
// leaves L2 as a 64-byte request of its own, and at one per merged segment they were a quarter of the kernel's HBM traffic
// on C5 (profiles/r02_pmc_C5_summary.json).  Larger n: B rows through L1/L2 (30 KB of LDS would halve the occupancy).
// BAND > 0 (with SMALL): the rows of B are banded with this half-bandwidth (SPARSEresamplebranchstates :218-261 walks the
// non-zeros of a row of the sparse matrix): a forward draw multiplies and adds the 2 BAND + 1 in-band terms only -- every
// other term of the probability vector is an exact +0, so total, running sums and the state drawn keep their bits.
template <bool KS, bool SMALL, bool B2L, int BAND>
__global__ __launch_bounds__(SMALL ? WT_BRANCH_BLOCK_SMALL : WT_BRANCH_BLOCK_BIG) void wt_branch_kernel(WtParams p, int it) {
  constexpr int BLOCK = SMALL ? WT_BRANCH_BLOCK_SMALL : WT_BRANCH_BLOCK_BIG;
  extern __shared__ __align__(16) unsigned char s_dyn[];            // SMALL: [n][ldt] rows of B, then [n][64] dwell sums (u64)
  __shared__ double s_scale[64];
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];        // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane8 = (uint32_t)lane * 8u;
  const int n = p.n_states, ldt = p.ldt;
  const int n_slots = SMALL ? p.n_slots : 0;
  // reduced output (statistics summed over replicas, the form 10^4 sites are asked in): what is summed anyway is summed here --
  // the workgroup's counts per transition and, for n > 32, its dwell sums per state collect in LDS and reach the tile's totals
  // as a few coalesced atomics at the end.  Per replica they are one scattered global atomic per segment (the counters of a tile,
  // n^2 x 64 x 4 B = 0.95 MB at 61 states, do not stay in L2): half of the C4 branch kernel (profiles/r03_probe_branch_ablation.log).
  const bool red = p.cnt_tile != nullptr;
  const bool red_dw = red && !SMALL, red_pc = red && n_slots == 0;
  const int ncnt = KS ? n * n : n * (n - 1);
  const BranchLds L = branch_lds(n, ldt, n_slots, SMALL, B2L, BAND, red, KS);
  double* s_B2 = reinterpret_cast<double*>(s_dyn + L.b2);
  unsigned long long* s_dw = reinterpret_cast<unsigned long long*>(s_dyn + L.dw);
  // SMALL with a sparse B (at most WT_MAX_SLOTS possible transitions, e.g. a banded rate matrix): the workgroup's transition
  // counts too are collected in LDS, one 32-bit counter per (possible pair, lane), and handed over as coalesced rows
  uint32_t* s_ct = reinterpret_cast<uint32_t*>(s_dyn + L.ct);                    // [n_slots][64]
  int16_t* s_slot = reinterpret_cast<int16_t*>(s_dyn + L.slot);                  // [n*n] pair -> slot, -1: none
  unsigned long long* s_dwt = reinterpret_cast<unsigned long long*>(s_dyn + L.dwt);   // [n][16]
  uint32_t* s_pc = reinterpret_cast<uint32_t*>(s_dyn + L.pc);                    // [ncnt]
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += BLOCK) s_ltab[i] = logtab_entry(i);
  if ((int)threadIdx.x < n) s_scale[threadIdx.x] = p.scale[threadIdx.x];
  // B2L: the rows of B in LDS (always for n <= 32; beyond, 30 KB at 61 states, when a workgroup walks enough branches to pay for
  // staging them -- a third of a draw's per-lane table reads: C4 branch kernel 20.8 -> 18.3-19.8 ms at 65 536 replicas)
  if (BAND > 0) { for (int i = threadIdx.x; i < n * (2 * BAND + 1); i += BLOCK) s_B2[i] = p.B2band[i]; }      // the band of B instead of its rows
  else if (B2L) for (int i = threadIdx.x; i < n * ldt; i += BLOCK) s_B2[i] = p.B2[i];
  if (SMALL) {
    for (int i = threadIdx.x; i < n * 64; i += BLOCK) s_dw[i] = 0ull;
    for (int i = threadIdx.x; i < n_slots * 64; i += BLOCK) s_ct[i] = 0u;
    if (n_slots > 0) for (int i = threadIdx.x; i < n * n; i += BLOCK) s_slot[i] = p.pair_slot[i];
  }
  if (red_dw) for (int i = threadIdx.x; i < n * 16; i += BLOCK) s_dwt[i] = 0ull;
  if (red_pc) for (int i = threadIdx.x; i < ncnt; i += BLOCK) s_pc[i] = 0u;
  __syncthreads();
  const int tile = blockIdx.x % p.n_tiles;
  const int grp = (blockIdx.x / p.n_tiles) * (BLOCK / 64) + wave;
  const bool active = grp < p.n_groups;              // wave-uniform; every wave reaches the barrier at the end
  const double* __restrict__ Brows = B2L ? s_B2 : p.B2;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  const bool valid = tile * 64 + lane < p.n_rep;
  uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  uint32_t* gc = p.cnt + ((size_t)tile * n * n) * 64 + lane;
  unsigned long long* gdw = p.dwfx + ((size_t)tile * n) * 64;
  uint32_t err = 0;
  uint32_t segs = 0;
  auto add_dwell = [&](int s, double len) {                                    // updatedwelltimes :752, per merged segment
    const unsigned long long v = (unsigned long long)__double2ll_rn(len * p.fx_scale);
    if (SMALL) atomicAdd(s_dw + s * 64 + lane, v);
    else if (red_dw) { if (valid) atomicAdd(s_dwt + s * 16 + (lane & 15), v); }
    else atomicAdd(gdw + s * 64 + lane, v);
  };
  auto count = [&](int a, int c) {                                             // shortener :65-66 / shortenerbf :1010-1014
    if (n_slots > 0) {
      const int sl = s_slot[a * n + c];
      if (sl >= 0) { atomicAdd(s_ct + sl * 64 + lane, 1u); return; }
    }
    const int col = KS ? a * n + c : a * (n - 1) + (c > a ? c - 1 : c);
    if (red_pc) { if (valid) atomicAdd(s_pc + col, 1u); }
    else atomicAdd(gc + col * 64, 1u);
  };
  if (active) {
  const int q1 = min((grp + 1) * p.group, p.n_edge);
  for (int q = grp * p.group; q < q1; ++q) {
  const int b = p.branch_order[q];
  const int m = mct[b * 64 + lane];
  const int es = p.estate[((size_t)tile * p.n_edge + b) * 64 + lane];
  const int ps = es & 255, cs = es >> 8;
  const int roff = p.slot[b];
  const int cap = p.slot[b + 1] - roff;
  double* __restrict__ in = p.dw[it & 1] + ((size_t)tile * p.rows + roff) * 64;
  double* __restrict__ out = p.dw[(it & 1) ^ 1] + ((size_t)tile * p.rows + roff) * 64;
  auto IN = [&](int k) -> double& { return at(in, (uint32_t)k * 512u + lane8); };
  // states of the merged segments, a byte per (row, lane) beside the dwell rows: pass A writes, pass B reads one step ahead.
  // (They sat in LDS, 4 KB per wave: with them in global memory -- 64-byte coalesced rows, L2-resident between the passes -- an
  // eight-wave workgroup needs 29 KB instead of 61 KB at 20 states and the SIMDs hold eight waves instead of four.)
  uint8_t* __restrict__ msrow = p.mstate + ((size_t)tile * p.rows + roff) * 64 + lane;
  auto MS = [&](int k) -> uint8_t& { return msrow[(uint32_t)k * 64u]; };

  Stream su, se;
  su.open(ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
  se.open(ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
  const int mmax = wave_max_count(m);
  int mnew = 0;

  // s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end          (resamplebranchstates :290, :301-304)
  auto draw_state_w = [&](int i, int sprev, uint32_t word) -> int {
    int kk = m - i - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    if (BAND > 0) {
      constexpr int W = 2 * BAND + 1;
      // The 2 BAND + 1 table entries are loaded UNCONDITIONALLY from clamped columns and issued together (one wait for all of
      // them: as per-term `if (in range) load` the compiler emitted a branch and a full wait per term -- three serial L2 round
      // trips per draw); a term outside the matrix has a zero band coefficient (the host's table), so its product is an exact
      // +0 whatever was loaded.  32-bit byte offsets from a scalar base (the tables are far below 4 GB: checked on the host).
      const uint32_t row = (uint32_t)((kk * n + cs) * ldt);
      const double* __restrict__ bb = s_B2 + sprev * W;
      double bt[W], bv[W];
#pragma unroll
      for (int d = 0; d < W; ++d) {
        const int cc = min(max(sprev + d - BAND, 0), n - 1);
        bt[d] = at(p.colL, (row + (uint32_t)cc) * 8u);
        bv[d] = bb[d];
      }
      double pr[W];
#pragma unroll
      for (int d = 0; d < W; ++d) pr[d] = bv[d] * bt[d];
      double total = pr[0];                                                      // 0 + .. + 0 + pr_0 = pr_0 exactly
#pragma unroll
      for (int d = 1; d < W; ++d) total += pr[d];
      if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
      const double thr = u01(word) * total;
      // states below the band: running sum +0, passed unless the threshold itself is 0; states above it: running sum = total >= thr
      int idx = (thr <= 0.0) ? 0 : max(sprev - BAND, 0);
      double cum = 0.0;
#pragma unroll
      for (int d = 0; d < W; ++d) {
        const int c = sprev + d - BAND;
        cum += pr[d];
        idx += (c >= 0 && c < n - 1 && !(thr <= cum)) ? 1 : 0;
      }
      return idx;
    }
    // The running sums of the probability vector p_c = B[s_prev][c] (B^kk e_end)[c] are a function of (kk, s_prev, end) alone.
    // The host has formed them with the sampler's own unfused left-to-right additions and keeps every eighth one (blkL: the sum
    // after states 7, 15, ..., and the total): the lane finds the block of eight states its threshold falls into from that one
    // cache line, takes the running sum at the block's start from it and walks the eight states of that block only -- the same
    // partial sums, hence the same state, as the full scan, for 12 instead of ~35 table reads at 61 states.
    const int nb = p.nblk;                                                     // ceil(n / 8)
    const double* __restrict__ blk = p.blkL + (((size_t)kk * n + sprev) * n + cs) * p.ldb;
    double e[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 v = {0.0, 0.0};
      if (2 * q < nb) v = reinterpret_cast<const double2*>(blk)[q];
      e[2 * q] = v.x; e[2 * q + 1] = v.y;
    }
    double total = e[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) total = (q == nb - 1) ? e[q] : total;          // the last kept sum is the total
    if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
    const double thr = u01(word) * total;
    int b0 = 0;
    double cum = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {                                              // running sums only grow: blocks wholly below thr
      const bool below = (q < nb - 1) && !(thr <= e[q]);
      b0 += below ? 1 : 0;
      cum = below ? e[q] : cum;
    }
    const int c0 = 8 * b0;
    const double2* __restrict__ beta = reinterpret_cast<const double2*>(p.colL + ((size_t)kk * n + cs) * ldt + c0);
    const double2* __restrict__ brow = reinterpret_cast<const double2*>(Brows + sprev * ldt + c0);
    double2 bt[4], br[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const bool ok = c0 + 2 * q4 < n - 1;
      bt[q4] = ok ? beta[q4] : double2{0.0, 0.0};
      br[q4] = ok ? brow[q4] : double2{0.0, 0.0};
    }
    int idx = c0;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {                                           // comparisons at states 0 .. n-2 decide (sample_cat)
      const int c = c0 + 2 * q4;
      cum += br[q4].x * bt[q4].x;                      // a padding term is +0: the sum keeps its bits
      idx += ((c < n - 1) && !(thr <= cum)) ? 1 : 0;
      cum += br[q4].y * bt[q4].y;
      idx += ((c + 1 < n - 1) && !(thr <= cum)) ? 1 : 0;
    }
    return idx;
  };

  if (mmax <= 64) {
    // Pass A: one old segment per step for every lane; merged segments written back in place over the consumed rows.
    int w = 0;
    int cur_s = (m == 1) ? cs : ps;            // updatenodestates :469-472 (m==1: child wins)
    const int s_first = cur_s;                 // state of the first merged segment
    // the first two and the last merged segment (length and state) stay in registers; only the ones between them pass through the
    // slot's rows and the byte rows beside them: with at most three merged segments on a branch pass B reads nothing back
    double first_len = 0.0, second_len = 0.0;
    int second_s = 0;
    double cur_len = IN(0);
    double dnext = (m > 1) ? IN(1) : 0.0;
    double dnext2 = (m > 2) ? IN(2) : 0.0;            // old segments two steps ahead: a step is two dependent table reads long, an HBM read longer
    for (int i0 = 1; i0 < mmax; i0 += 4) {
      uint32_t wd[4] = {0u, 0u, 0u, 0u};
      if (i0 < mmax - 1)                       // some lane still draws in this group (draws exist for i < m - 1)
        philox4x32((uint32_t)((i0 - 1) >> 2), ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi, wd);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int i = i0 + qq;
        if (i < m) {
          const int si = (i == m - 1) ? cs : draw_state_w(i, cur_s, wd[qq]);
          const double di = dnext;
          dnext = dnext2;
          if (i + 2 < m) dnext2 = IN(i + 2);
          if (KS) count(cur_s, si);
          if (si == cur_s) cur_len = cur_len + di;                           // shortener :54
          else {
            if (w == 0) first_len = cur_len;
            else if (w == 1) { second_len = cur_len; second_s = cur_s; }
            else { IN(w) = cur_len; MS(w) = (uint8_t)cur_s; }
            if (!KS) count(cur_s, si);
            ++w; cur_s = si; cur_len = di;
          }
        }
      }
    }
    const int nmerged = w + 1;
    const double len0 = (w == 0) ? cur_len : first_len;

    // Pass B: one new piece per step for every lane (virtual jumps :391-410).
    int j = 0;
    int s = s_first;
    double len = len0;
    double lnext = (nmerged > 1) ? ((w == 1) ? cur_len : second_len) : 0.0;
    int snext = (nmerged > 1) ? ((w == 1) ? cur_s : second_s) : 0;
    double tot = 0.0, scale = s_scale[s];
    bool stuck = false, done = false;
    for (uint32_t t0 = 0; __any(!done); t0 += 4) {
      uint32_t wd[4];
      philox4x32(t0 >> 2, ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi, wd);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        if (done) continue;
        double piece;
        bool adv;
        if (stuck || !(0.0 < len)) { stuck = true; piece = len; adv = true; }
        else {
          const double rl = scale * neglog_u32(wd[qq], s_ltab);                // :398
          if ((tot + rl) < len) { piece = rl; tot += rl; adv = false; }
          else { piece = len - tot; adv = true; }
        }
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = piece; else err |= DERR_CAPACITY;
        ++mnew;
        if (adv) {
          add_dwell(s, len);
          ++j;
          if (j >= nmerged) done = true;
          else {
            len = lnext; s = snext;
            if (j + 1 < w) { lnext = IN(j + 1); snext = MS(j + 1); } else { lnext = cur_len; snext = cur_s; }      // the last one never left its registers
            scale = s_scale[s]; tot = 0.0;
          }
        }
      }
    }
  } else {
    // General path (a lane with more than 64 segments on this branch): the reference's loop nest as written.
    uint32_t edraw = 0;
    bool stuck = false;
    auto finalize = [&](int s, double len) {
      add_dwell(s, len);
      if (stuck || !(0.0 < len)) {
        stuck = true;
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = len; else err |= DERR_CAPACITY;
        ++mnew;
        return;
      }
      const double scale = s_scale[s];
      double tot = 0.0;
      while (tot < len) {
        const double rl = scale * neglog_u32(se.draw_word(edraw++), s_ltab);
        double piece;
        if ((tot + rl) < len) { piece = rl; tot += rl; }
        else { piece = len - tot; tot = len; }
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = piece; else err |= DERR_CAPACITY;
        ++mnew;
      }
    };
    int cur_s = (m == 1) ? cs : ps;
    double cur_len = IN(0);
    for (int i = 1; i <= m; ++i) {              // i == m: sentinel that flushes the last merged segment
      int si = -1;
      double di = 0.0;
      if (i < m) {
        si = (i == m - 1) ? cs : draw_state_w(i, cur_s, su.draw_word((uint32_t)(i - 1)));
        di = IN(i);
      }
      if (KS && si >= 0) count(cur_s, si);
      if (si == cur_s) cur_len = cur_len + di;
      else {
        finalize(cur_s, cur_len);
        if (!KS && si >= 0) count(cur_s, si);
        cur_s = si; cur_len = di;
      }
    }
  }
  if (mnew > cap) mnew = cap;
  if (mnew > 65535) { err |= DERR_CAPACITY; mnew = 65535; }
  mct[b * 64 + lane] = (uint16_t)mnew;
  if (valid) segs += (uint32_t)(m + mnew);
  }      // next branch of the group
  }      // active

#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) segs += __shfl_xor(segs, off, 64);
  if (lane == 0 && active) atomicAdd(p.segacc + (size_t)tile * 64 + (grp & 63), (unsigned long long)segs);
  if (SMALL || red) __syncthreads();
  if (red_dw)
    for (int i = threadIdx.x; i < n * 16; i += BLOCK) {
      const unsigned long long v = s_dwt[i];
      if (v) atomicAdd(p.dwfx_tile + (size_t)tile * n * 16 + i, v);
    }
  if (red_pc)
    for (int i = threadIdx.x; i < ncnt; i += BLOCK) {
      const uint32_t v = s_pc[i];
      if (v) atomicAdd(p.cnt_tile + (size_t)tile * n * n + i, v);
    }
  if (SMALL) {                                       // the workgroup's dwell sums -> the tile's accumulators, row by row
    for (int i = threadIdx.x; i < n * 64; i += BLOCK) {
      const unsigned long long v = s_dw[i];
      if (v) atomicAdd(gdw + i, v);
    }
    for (int i = threadIdx.x; i < n_slots * 64; i += BLOCK) {
      const uint32_t v = s_ct[i];
      if (v) atomicAdd(p.cnt + ((size_t)tile * n * n + p.slot_col[i >> 6]) * 64 + (i & 63), v);
    }
  }
  if (err) atomicOr(p.err, err);
}

// The statistics row: a wave per (tile, chunk of 64 columns).  Columns: n dwell sums (fixed point -> double), the counters,
// (ks) the root state; the accumulators are cleared for the next sweep.
__global__ __launch_bounds__(WT_BLOCK) void wt_stats_kernel(WtParams p, int it, int n_chunks) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WT_BLOCK / 64) + (threadIdx.x >> 6);
  if (item >= n_chunks * p.n_tiles) return;
  const int tile = item / n_chunks, chunk = item % n_chunks;
  const int n = p.n_states;
  const int ncnt = p.ks ? n * n : n * (n - 1);
  const int dcols = n + ncnt + (p.ks ? 1 : 0);
  const int rep_local = tile * 64 + lane;
  const bool valid = rep_local < p.n_rep;
  unsigned long long* gdw = p.dwfx + ((size_t)tile * n) * 64 + lane;
  uint32_t* gc = p.cnt + ((size_t)tile * n * n) * 64 + lane;
  const int c1 = min(dcols, (chunk + 1) * 64);
  for (int c = chunk * 64; c < c1; ++c) {
    double v;
    // reduced output: the branch kernel has summed part of the statistics over the lanes already (dwfx_tile: sixteen partial
    // sums per state, each over four lanes' worth of a tile -- 64-bit fixed point below 2^62; cnt_tile: exact counts); the
    // per-lane accumulators it then leaves untouched are not read
    const bool tile_dw = p.cnt_tile != nullptr && n > 32, tile_pc = p.cnt_tile != nullptr && p.n_slots == 0;
    if (c < n) {
      if (tile_dw) {
        unsigned long long* td = p.dwfx_tile + ((size_t)tile * n + c) * 16;
        v = 0.0;
        if (lane < 16) { v = (double)(long long)td[lane] * p.fx_inv; td[lane] = 0ull; }
      } else { v = (double)(long long)gdw[c * 64] * p.fx_inv; gdw[c * 64] = 0ull; v = (valid || !p.reduce) ? v : 0.0; }
    } else if (c < n + ncnt) {
      uint32_t k = 0;
      if (!tile_pc) { k = gc[(c - n) * 64]; gc[(c - n) * 64] = 0u; }
      v = (valid || !p.reduce) ? (double)k : 0.0;
    } else v = (valid || !p.reduce) ? (double)p.nstate[((size_t)tile * p.n_node + p.root) * 64 + lane] : 0.0;   // :1350-1352
    if (p.reduce) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0 && tile_pc && c >= n && c < n + ncnt) { v += (double)p.cnt_tile[(size_t)tile * n * n + (c - n)]; p.cnt_tile[(size_t)tile * n * n + (c - n)] = 0u; }
      if (lane == 0) p.stats[((size_t)it * p.n_tiles + tile) * p.n_cols + c] = v;
    } else {
      p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + rep_local] = v;
    }
  }
  if (chunk == 0) {
    unsigned long long sg = p.segacc[(size_t)tile * 64 + lane];
    p.segacc[(size_t)tile * 64 + lane] = 0ull;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sg += __shfl_xor(sg, off, 64);
    if (lane == 0) atomicAdd(p.segcnt, sg);
  }
}

template <int MT>
hipError_t launch_up_levels(const WtParams& p, const std::vector<int32_t>& up_off, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)(WT_BLOCK / 64) * p.n_states * 64;
  {
    const hipError_t ae = allow_dynamic_lds(reinterpret_cast<const void*>(wt_up_kernel<MT>),
                                            (int)(sizeof(double) * (size_t)(WT_BLOCK / 64) * 16 * MT * 64));      // the largest n of this MT
    if (ae != hipSuccess) return ae;
  }
  // Measured crossovers (profiles/r02_probe_few_tiles.log): the row-split workgroups win below about 64 tiles at 20 states and
  // below about 512 tiles at 61 states; the sorted blocks of the per-tile kernel beyond.
  // Round 4: beyond 32 states the per-tile kernel runs two waves per SIMD (wt_up2_kernel) and takes over from 256 tiles at 61 states
  // (profiles/r04_probe_c4_up2.log: 128 tiles 4.96 split / 6.40 ms, 256 tiles 9.46 / 9.03, 512 tiles 17.7 / 15.3, 1 024 tiles 32.7 / 28.8).
  const int msplit_tiles = MT == 2 ? 64 : MT == 3 ? 192 : 256;
  const bool few_tiles = p.n_tiles < WT_FEW_TILES;      // n <= 16 (one row block): not enough (node, tile) items to fill the chip -> a wave per block
  for (size_t l = 0; l + 1 < up_off.size(); ++l) {
    const int cnt = up_off[l + 1] - up_off[l];
    if (cnt <= 0) continue;
    const bool split = p.up_form == 2 || (p.up_form == 0 && (MT >= 2 ? p.n_tiles < msplit_tiles : few_tiles));
    if (MT >= 2 && split) {
      const int64_t items = (int64_t)cnt * p.n_tiles * 4;    // a workgroup of MT waves per (node, tile, block)
      const unsigned grid = (unsigned)std::min<int64_t>(items, 4096);
      hipLaunchKernelGGL(wt_up_msplit_kernel<MT>, dim3(grid), dim3(64 * MT), 0, stream, p, up_off[l], up_off[l + 1]);
    } else if (split) {
      const int64_t items = (int64_t)cnt * p.n_tiles * 4;    // a wave per (node, tile, block)
      const unsigned grid = (unsigned)std::min<int64_t>((items + 3) / 4, 2048);
      hipLaunchKernelGGL(wt_up_blocks_kernel<MT>, dim3(grid), dim3(WT_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]);
    } else if (MT >= 2 && p.up_form != 1) {             // up_form 1 keeps the kernel with the matrix in registers (measurement)
      const int64_t items = (int64_t)cnt * p.n_tiles;        // a wave per (node, tile), two (17 .. 32 states: three) waves per SIMD
      constexpr int M2 = MT >= 2 ? MT : 2;
      const unsigned grid = (unsigned)std::min<int64_t>((items + 3) / 4, 256 * wt_up2_waves(M2));      // persistent workgroups
      switch ((p.n_states + 3) / 4 - 4 * (M2 - 1)) {         // k-steps in the last row block: 1 .. 4
        case 1: hipLaunchKernelGGL((wt_up2_kernel<M2, 4 * M2 - 3>), dim3(grid), dim3(WT_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]); break;
        case 2: hipLaunchKernelGGL((wt_up2_kernel<M2, 4 * M2 - 2>), dim3(grid), dim3(WT_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]); break;
        case 3: hipLaunchKernelGGL((wt_up2_kernel<M2, 4 * M2 - 1>), dim3(grid), dim3(WT_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]); break;
        default: hipLaunchKernelGGL((wt_up2_kernel<M2, 4 * M2>), dim3(grid), dim3(WT_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]); break;
      }
    } else {
      const int64_t items = (int64_t)cnt * p.n_tiles;        // a wave per (node, tile)
      const unsigned grid = (unsigned)std::min<int64_t>((items + 3) / 4, 2048);      // persistent waves: the matrix fragments load once
      hipLaunchKernelGGL(wt_up_kernel<MT>, dim3(grid), dim3(WT_BLOCK), lds, stream, p, up_off[l], up_off[l + 1]);
    }
  }
  return hipSuccess;
}

}  // namespace

template <int NP, int HB>
void launch_up_band(const WtParams& p, const WtBand& band, const std::vector<int32_t>& up_off, const std::vector<int32_t>& tier_off, hipStream_t stream) {
  if (p.cl_nodes && tier_off.size() > 1) {           // deep tree: one launch per tier of subtree clusters
    for (size_t t = 0; t + 1 < tier_off.size(); ++t) {
      const int n_cl = tier_off[t + 1] - tier_off[t];
      hipLaunchKernelGGL((wt_up_band_cluster_kernel<NP, HB>), dim3((unsigned)((int64_t)n_cl * p.n_tiles)), dim3(WT_CL_BLOCK), 0, stream, p, band, tier_off[t], n_cl);
    }
    return;
  }
  for (size_t l = 0; l + 1 < up_off.size(); ++l) {
    const int cnt = up_off[l + 1] - up_off[l];
    if (cnt <= 0) continue;
    const int64_t items = (int64_t)cnt * p.n_tiles;        // a wave per (node, tile)
    hipLaunchKernelGGL((wt_up_band_kernel<NP, HB>), dim3((unsigned)((items + 3) / 4)), dim3(WT_BLOCK), 0, stream, p, band, up_off[l], up_off[l + 1]);
  }
}

hipError_t launch_wtiles_up(const WtParams& p, const WtBand& band, const WtSparseUp& sparse, const std::vector<int32_t>& up_off,
                            const std::vector<int32_t>& tier_off, hipStream_t stream) {
  if (sparse.kernel) return launch_sparse_up(*sparse.kernel, sparse.params, up_off, stream);      // unstructured sparse chain matrix: the kernel generated for its pattern
  if (p.band_up > 0) {                                  // banded chain matrix: per-lane FMAs over the band
    const int np = (p.n_states + 3) / 4;               // vectors padded to a multiple of four states
#define PHM_BAND_CASE(NPQ)                                                                          \
    case NPQ: if (p.band_up == 1) launch_up_band<4 * NPQ, 1>(p, band, up_off, tier_off, stream);  \
              else launch_up_band<4 * NPQ, 2>(p, band, up_off, tier_off, stream);                  \
              break;
    switch (np) {
      PHM_BAND_CASE(2) PHM_BAND_CASE(3) PHM_BAND_CASE(4) PHM_BAND_CASE(5) PHM_BAND_CASE(6) PHM_BAND_CASE(7) PHM_BAND_CASE(8)
      default: return hipErrorInvalidValue;
    }
#undef PHM_BAND_CASE
    return hipGetLastError();
  }
  const int mt = (p.n_states + 15) / 16;
  hipError_t e;
  if (mt == 1) e = launch_up_levels<1>(p, up_off, stream);
  else if (mt == 2) e = launch_up_levels<2>(p, up_off, stream);
  else if (mt == 3) e = launch_up_levels<3>(p, up_off, stream);
  else e = launch_up_levels<4>(p, up_off, stream);
  return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_wtiles_sweep(const WtParams& p, const WtBand& band, const WtSparseUp& sparse, const std::vector<int32_t>& up_off,
                               const std::vector<int32_t>& down_off, const std::vector<int32_t>& tier_off, int it, hipStream_t stream,
                               hipEvent_t* phase_ev) {
  constexpr int WPB = WT_BLOCK / 64;
  auto blocks = [&](int64_t items) { return dim3((unsigned)((items + WPB - 1) / WPB)); };
  auto mark = [&](int i) { if (phase_ev) (void)hipEventRecord(phase_ev[i], stream); };
  mark(0);
  hipError_t e = launch_wtiles_up(p, band, sparse, up_off, tier_off, stream);
  if (e != hipSuccess) return e;
  mark(1);
  hipLaunchKernelGGL(wt_root_kernel, blocks(p.n_tiles), dim3(WT_BLOCK), 0, stream, p, it);
  const int mt = (p.n_states + 15) / 16;
  if (p.cl_nodes && tier_off.size() > 1 && mt <= 2) {      // deep tree, n <= 32: the node draws tier by tier from the top
    for (int t = (int)tier_off.size() - 2; t >= 0; --t) {
      const int n_cl = tier_off[t + 1] - tier_off[t];
      const dim3 g((unsigned)((int64_t)n_cl * p.n_tiles));
#define PHM_D1RC(NPQ) case NPQ: hipLaunchKernelGGL(wt_down1r_cluster_kernel<4 * NPQ>, g, dim3(WT_CL_BLOCK), 0, stream, p, it, tier_off[t], n_cl); break;
      switch ((p.n_states + 3) / 4) { PHM_D1RC(2) PHM_D1RC(3) PHM_D1RC(4) PHM_D1RC(5) PHM_D1RC(6) PHM_D1RC(7) PHM_D1RC(8) default: return hipErrorInvalidValue; }
#undef PHM_D1RC
    }
  } else
  for (size_t l = 0; l + 1 < down_off.size(); ++l) {
    const int n = down_off[l + 1] - down_off[l];
    if (n <= 0) continue;
    const dim3 g = blocks((int64_t)n * p.n_tiles);
    if (mt >= 3) {      // n > 32: one pass, a wave per workgroup
      const dim3 g1((unsigned)((int64_t)n * p.n_tiles));
      if (mt == 3) hipLaunchKernelGGL(wt_down1_kernel<3>, g1, dim3(64), 32 * 64 * sizeof(double), stream, p, it, down_off[l], down_off[l + 1]);
      else hipLaunchKernelGGL(wt_down1_kernel<4>, g1, dim3(64), 32 * 64 * sizeof(double), stream, p, it, down_off[l], down_off[l + 1]);
      continue;
    }
    {
#define PHM_D1R(NPQ) case NPQ: hipLaunchKernelGGL(wt_down1r_kernel<4 * NPQ>, g, dim3(WT_BLOCK), 0, stream, p, it, down_off[l], down_off[l + 1]); break;
      switch ((p.n_states + 3) / 4) { PHM_D1R(2) PHM_D1R(3) PHM_D1R(4) PHM_D1R(5) PHM_D1R(6) PHM_D1R(7) PHM_D1R(8) default: return hipErrorInvalidValue; }
#undef PHM_D1R
    }
  }
  mark(2);
  {
    const bool small = p.n_states <= 32;
    const bool red = p.cnt_tile != nullptr;
    // n <= 32: eight waves share the workgroup's LDS tables (B rows, dwell and count accumulators: 26 KB at 20 states)
    const int wpb = small ? WT_BRANCH_BLOCK_SMALL / 64 : WT_BRANCH_BLOCK_BIG / 64;
    const dim3 g((unsigned)(((int64_t)p.n_groups + wpb - 1) / wpb * p.n_tiles));
    const bool b2l = !small && p.group >= 4;           // n > 32: B rows in LDS once a wave walks four or more branches
    const BranchLds L = branch_lds(p.n_states, p.ldt, small ? p.n_slots : 0, small, b2l, small ? p.band_draw : 0, red, p.ks != 0);
    const size_t lds_now = L.total;
    const int band = small ? p.band_draw : 0;
    // beyond the default 64 KB of dynamic LDS (n = 32 with 96 countable pairs) the kernel variant about to be launched is told so -- once
    // per (variant, device), and only then: the per-iteration drivers launch a sweep per rate update
#define PHM_BRANCH(KSV, SM, BL, BD)                                                                                                    \
    do {                                                                                                                               \
      if (lds_now > 64 * 1024) {                                                                                                       \
        const hipError_t ae = allow_dynamic_lds((const void*)wt_branch_kernel<KSV, SM, BL, BD>, 96 * 1024);                            \
        if (ae != hipSuccess) return ae;                                                                                               \
      }                                                                                                                                \
      hipLaunchKernelGGL((wt_branch_kernel<KSV, SM, BL, BD>), g, dim3(SM ? WT_BRANCH_BLOCK_SMALL : WT_BRANCH_BLOCK_BIG), lds_now, stream, p, it); \
    } while (0)
    if (p.ks) {
      if (small && band == 1) PHM_BRANCH(true, true, true, 1);
      else if (small && band == 2) PHM_BRANCH(true, true, true, 2);
      else if (small) PHM_BRANCH(true, true, true, 0);
      else if (b2l) PHM_BRANCH(true, false, true, 0);
      else PHM_BRANCH(true, false, false, 0);
    } else {
      if (small && band == 1) PHM_BRANCH(false, true, true, 1);
      else if (small && band == 2) PHM_BRANCH(false, true, true, 2);
      else if (small) PHM_BRANCH(false, true, true, 0);
      else if (b2l) PHM_BRANCH(false, false, true, 0);
      else PHM_BRANCH(false, false, false, 0);
    }
#undef PHM_BRANCH
  }
  mark(3);
  const int ncnt = p.ks ? p.n_states * p.n_states : p.n_states * (p.n_states - 1);
  const int dcols = p.n_states + ncnt + (p.ks ? 1 : 0);
  const int n_chunks = (dcols + 63) / 64;
  hipLaunchKernelGGL(wt_stats_kernel, blocks((int64_t)n_chunks * p.n_tiles), dim3(WT_BLOCK), 0, stream, p, it, n_chunks);
  mark(4);
  return hipGetLastError();
}

}  // namespace phm
