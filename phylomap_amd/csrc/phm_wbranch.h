// phm_wbranch.h -- the MCMC sweep for 5 <= n <= 64 states with one wavefront per (replica, branch).
//
// phm_wide.hip gives a wave 64 replicas and lets them take turns at the n-vector work (one state per lane), one branch
// after the other: every step of every replica is a dependent chain through L2, and a chip needs ~10^5 replicas before its
// waves hide that latency (measured on C5, 20 states x 5 000 tips: 0.48 s per sweep for ONE replica per wave).  Here the
// lanes are still the states, but a wave owns ONE branch of ONE replica: control flow is wave-uniform (scalar registers),
// only the n-vectors live in the lanes, and S replicas expose S x E waves -- the branches of a sweep are conditionally
// independent given the node states and every random number is addressed by (replica, iteration, node | branch).
//   up     : one launch per HEIGHT level (runs of narrow levels near the root: one launch), TWO waves per (node, replica), one
//            per child chain; dense chain matrix: a row per lane in registers, the vector broadcast from LDS; banded: neighbours
//            through DPP row shifts; other sparse matrices: ELLPACK rows                  makePLrcpp* :503-529
//   root   : a wave per replica                                           :618-627
//   down   : a TRANSITION MAP per (edge, replica) -- lane = candidate parent state -- then one workgroup per replica walks the
//            depth levels by table look-up (round 3; one launch per depth level when the maps would not fit)   :640-657, :460-475
//   branch : a wave per (branch, replica), longest slots first; state-independent operands a step ahead, merged segments in LDS
//                                                                          :264-413, :44-73, :745-757
//   stats  : a workgroup per (replica, column): fixed-order reduction of the per-branch dwell sums; counters added with f64
//            atomics (integers: exact in any order)
// Per-replica layout as in phm_narrow.h (CSR dwell slots, two buffers + merge scratch); chain powers from the full-length
// tables.  Counts bit-identical to the oracle, dwell sums <= 1e-10 relative (added per branch, then reduced).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

constexpr int WB_ELL_MAX = 16;          // widest ELLPACK row the pruning kernel keeps in LDS
constexpr int WB_BLOCK = 256;          // four waves = four (item, replica) pairs share the LDS copy of B

struct WideBranchParams {
  int32_t n_states;
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t n_rep, n_rep_pad, replica_offset, n_tiles;
  int32_t normalise, tips_per_replica, sparse, ks, tip_masks, count_self, reduce, n_cols;
  int32_t klong;                             // rows of the chain tables
  int32_t ell_w;                             // > 0: the chain matrix has at most ell_w non-zeros per row (ELLPACK copy below)
  int32_t ell2_w;                            // the same for the dense-step matrix B2 (rows of the forward draws)
  int32_t band_hb;                           // ell_w > 0 and every non-zero of the chain matrix within |row - col| <= band_hb (1 or 2); else 0
  uint32_t seed_lo, seed_hi;
  int64_t total_cap;                         // doubles per replica in one dwell buffer
  const double* B2;                          // [n][n] dense B, row-major
  const double* Bc;                          // [n][n] chain matrix (B, or thresholded B for SPARSE)
  const double* scale;                       // [n] 1/(Omega+q_ss)
  const int32_t* ell_col;                    // [n][ell_w] column of the t-th non-zero of a row, ascending; padding: own row, value 0
  const double* ell_val;                     // [n][ell_w]
  const int32_t* ell2_col;                   // [n][ell2_w]
  const double* ell2_val;                    // [n][ell2_w]
  const double* pid;                         // [n]
  const UpStep* up;
  const DownStep* down;
  const int32_t* up_order;
  const int32_t* down_order;
  const int32_t* up_off;                     // level boundaries into up_order / down_order (device copies of the host arrays)
  const int32_t* down_off;
  const int32_t* branch_order;
  const int64_t* off;                        // [n_edge + 1] CSR offsets of the branch slots
  const double* colL;                        // [klong][n][n]  (Bc^k e_j)[r]
  const double* rowL;                        // [klong][n][n]  ((Bc^T)^k e_j)[c]
  const double* maskL;                       // [klong][2][n]
  const uint8_t* tips;                       // [n_tips] or [replica][n_tips], 0-based
  int32_t* mcount;                           // [replica][n_edge]
  double* dw[2];                             // [replica][total_cap]; sweep `it` reads dw[it & 1], writes the other
  double* mlen;                              // [replica][total_cap] merged lengths (scratch of one sweep)
  uint8_t* mstate;                           // [replica][total_cap] ... and their states
  uint8_t* estate;                           // [replica][n_edge][2]
  double* PL;                                // [replica][n_node][n]
  uint8_t* nstate;                           // [replica][n_node]
  uint8_t* dmap;                             // [replica][n_edge (depth-level order)][n] transition maps of the sampling sweep; null: level launches
  double* part;                              // [replica][n_edge][n + 1] per-branch dwell sums, segments touched
  double* cnt;                               // [replica][n_cols] transition counters of the sweep (f64 atomics on integers)
  double* rowbuf;                            // [replica][n_cols]
  double* stats;                             // engine layout: reduce ? [iter][tile][cols] : [iter][cols][n_rep_pad]
  uint32_t* err;
  unsigned long long* segcnt;
};

size_t wbranch_lds_bytes(int n, bool sparse);

hipError_t launch_wbranch_sweep(const WideBranchParams& p, const std::vector<int32_t>& up_off,
                                const std::vector<int32_t>& down_off, int it, hipStream_t stream);

}  // namespace phm
