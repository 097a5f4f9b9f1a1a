// phm_exp.h -- launchers of the matrix-exponentiation path (sumstatEXP): batched transition matrices
// (K1 eigen route, K1' Pade scaling-and-squaring), pruning with P(t_b), and the per-sample sweep with the
// end-point-conditioned uniformisation sampler (phm_exp.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

constexpr int UNIF_CAP = 300;     // newunifSample gives up beyond 300 jumps (src/phylomap.cpp:120)
constexpr int EXP_BLOCK = 256;

// P_b = | L diag(exp(d_k t_b)) R |, all row-major; out[b][i][j]
hipError_t launch_expm_eigen(int n, const double* L, const double* R, const double* dvals, const double* t, int n_t,
                             double* out, hipStream_t stream);

// same product on the matrix cores (v_mfma_f64_16x16x4), 16 < n <= 64; last-bit differences from the exact kernel
hipError_t launch_expm_eigen_mfma(int n, const double* L, const double* R, const double* dvals, const double* t, int n_t,
                                  double* out, hipStream_t stream);

// P_b = expmat(Q t_b): Pade(6) + s_b squarings; `s` holds the squaring counts, `work` 5*n*n doubles per matrix
hipError_t launch_expm_pade(int n, const double* Q, const double* t, const int32_t* s, int n_t, double* work,
                            double* out, uint32_t* err, hipStream_t stream);

// the same on the matrix cores (16 < n <= 64): Pade powers, block Gauss-Jordan solve(D,E) and squarings as MFMA f64 products.
// bad[b] (n_t ints, zeroed by the caller) is set for a matrix with a pivot below piv_min in a diagonal block (out[b] is then
// not written): the caller hands those to launch_expm_pade.
hipError_t launch_expm_pade_mfma(int n, const double* Q, const double* t, const int32_t* s, int n_t, double* out,
                                 int32_t* bad, double piv_min, hipStream_t stream);

// PL[parent] = (P_a PL[child_a]) (.) (P_b PL[child_b]); PL is (2T-1) x n row-major, tips pre-filled one-hot;
// rescale: every internal row divided by its sum (not in the reference; the node draws do not depend on a row's scale)
hipError_t launch_exp_pl(int n, int n_node, int n_tips, const UpStep* up, const double* P, double* PL, int rescale,
                         hipStream_t stream);
// the same values, one launch per height level of the tree, a thread per node (`order`: positions of `up` grouped by height)
hipError_t launch_exp_pl_levels(int n, int n_tips, const UpStep* up, const int32_t* order, const std::vector<int32_t>& level_off,
                                const double* P, double* PL, int rescale, hipStream_t stream);

// log p(y|Q): pruning with P(t_b), rows normalised, log scale factors summed in the order of `up` (DIC drivers).
// `order` (device): positions of `up` grouped by height level, `level_off` (host) the level boundaries; `logs`: n_node doubles.
hipError_t launch_exp_pl_loglik(int n, int n_node, int n_tips, const UpStep* up, const int32_t* order,
                                const std::vector<int32_t>& level_off, const double* P, double* PL, double* logs,
                                const double* pid, int root_node, double* out_ll, hipStream_t stream);

template <int NS>
struct ExpParams {
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t N;                                 // samples
  int32_t n_tiles;
  int32_t it0;                               // global index of this call's first sample (multi-device calls split the N samples)
  uint32_t seed_lo, seed_hi, replica;
  double poisson_rate;                       // -min diag(Q), src/phylomap.cpp:3008
  double pid[NS];
  const DownStep* down;
  const double* P;                           // [n_edge][NS][NS]
  const double* PL;                          // [2T-1][NS]
  const double* edge_length;                 // [n_edge]
  const double* colpow;                      // [UNIF_CAP+1][NS][NS]: (B^k e_j)[r], B = I + Q/poisson_rate
  const double* B2;                          // [NS][NS]
  const uint8_t* tips;                       // [n_tips] 0-based
  uint8_t* nstate;                           // [tile][n_node][64]
  double* times;                             // [tile][UNIF_CAP][64] jump-time scratch
  double* out;                               // N x cols column-major
  uint32_t* err;
};

template <int NS> hipError_t launch_exp_sample(const ExpParams<NS>& p, hipStream_t stream);

struct ExpWideParams {                         // 5 <= n <= 64: runtime state count, model vectors in global memory
  int32_t n_states;
  int32_t n_tips, n_node, n_edge, root;
  int32_t N, n_tiles;
  int32_t it0;                                 // global index of the first sample
  uint32_t seed_lo, seed_hi, replica;
  double poisson_rate;
  const double* pid;                           // [n]
  const DownStep* down;
  const double* P;                             // [n_edge][n][n]
  const double* PL;                            // [2T-1][n]
  const double* edge_length;
  const double* colpow;                        // [UNIF_CAP+1][n][n]
  const double* B2;                            // [n][n]
  const uint8_t* tips;
  uint8_t* nstate;                             // [tile][n_node][64]
  double* times;                               // [tile][UNIF_CAP][64]
  double* out;                                 // N x cols column-major, zeroed by the host, accumulated in place
  uint32_t* err;
};
hipError_t launch_exp_wide(const ExpWideParams& p, hipStream_t stream);

// sumstatEXP with one wavefront per (tile of 64 samples, branch): the mapping for the sample counts the R function is called
// with (N = 10^3 .. 10^4).  exp_sample_kernel / exp_wide_kernel give a tile of 64 samples ONE wave that walks the whole tree, so
// N = 1 000 samples keep 16 waves busy on a chip with 1 024 SIMDs; given the node states the branches of a sample are
// conditionally independent and every random number is addressed by (sample, node | branch), so here the node states are drawn
// level by level (a wave per (tile, edge)) and newunifSample runs in a wave per (tile, branch).  Same draws, same counts;
// dwell sums are collected in 64-bit fixed point (integer atomics: exact, order-independent) and agree to 1e-15 of the tree length.
struct ExpTilesParams {
  int32_t n_states;
  int32_t n_tips, n_node, n_edge, root;        // root: internal index
  int32_t N, n_tiles;
  int32_t it0;                                 // global index of the first sample
  uint32_t seed_lo, seed_hi, replica;
  double poisson_rate;
  double fx_scale, fx_inv;                     // fixed-point scale of the dwell accumulators (powers of two)
  const double* pid;                           // [n]
  const DownStep* down;                        // [n_edge] pre-order
  const int32_t* node_order;                   // positions into down[] of the edges with an internal child, grouped by depth
  const double* P;                             // [n_edge][n][n]
  const double* PL;                            // [2T-1][n]
  const double* edge_length;
  const double* colpow;                        // [UNIF_CAP+1][n][n]
  const double* B2;                            // [n][n]
  const uint8_t* tips;
  uint8_t* nstate;                             // [tile][n_node][64]
  double* times;                               // [resident wave][UNIF_CAP][64] jump-time scratch
  unsigned long long* dwfx;                    // [n][n_tiles*64] dwell sums, fixed point
  uint32_t* cnt;                               // [n(n-1)][n_tiles*64]
  double* out;                                 // N x cols column-major
  uint32_t* err;
};
// level_off: boundaries of the depth levels in node_order (host)
hipError_t launch_exp_tiles(const ExpTilesParams& p, const std::vector<int32_t>& level_off, int branch_blocks, hipStream_t stream);

}  // namespace phm
