// phm_wtiles.h -- the MCMC sweep for 5 <= n <= 64 states with one LANE per replica: a wavefront per (tile of 64 replicas, item).
//
// The state-per-lane kernels (phm_wbranch.hip: a wave per (replica, branch); phm_wide.hip: a wave per tile, replicas in turn)
// pay an O(n) chain of v_readlane broadcasts for every mat-vec row and every categorical draw and leave 64 - n lanes idle
// (round-1 profile, profiles/r02_base_wide_*: C4 1.2e8, C5 3.7e8 realisations/s).  Here the lanes are the replicas, as in
// the n <= 4 kernels of phm_tiles.hip: every per-replica array is [tile][entity][state | slot][64 lanes], every access of a
// wave is a contiguous 512-byte row, the 64 lanes run the same control flow on the same branch, and a categorical draw is
// a private running sum over the n states of the lane's own probability vector (two passes over L2-resident table rows).
// The one GEMM-shaped piece of the sweep -- pruning through an internal child, X <- B X repeated m - 1 times on the
// child's partial-likelihood vectors -- runs on the matrix cores: v_mfma_f64_16x16x4 with B as the A operand held in
// registers for the whole launch and a 16-replica block of vectors as the B operand; an accumulator tile IS the next
// step's B-operand slice (row = 4 * reg-block + lane group), so a chain never leaves the register file, and each replica
// keeps the product of its own step m - 1 (segment counts differ per replica).
//   up     : one launch per HEIGHT level, a wave per (node, tile): four 16-replica MFMA blocks,   makePLrcpp* :503-529   [MFMA]
//            replicas dealt to the blocks in the order of their chain lengths; with few tiles a
//            workgroup per (node, tile, block), a wave per 16-state row block of the products
//   root   : a wave per tile                                                               :618-627
//   down   : one launch per DEPTH level, a wave per (tile, edge)                           :640-657, :460-475
//   branch : a wave per (tile, group of branches)                                          :264-413, :44-73, :745-757
//   stats  : a wave per (tile, chunk of columns)
// Arithmetic = DESIGN.md section 2 for n > 4: chain products are fused multiply-adds with j ascending from +0 (what the
// MFMA accumulates), normalisation sums four interleaved partials; draws and counts are bit-identical to the oracle.
// Dwell sums are accumulated in 64-bit fixed point (integer atomics: exact, order-independent, identical from run to run)
// with a quantum of 2^-(61 - ceil(log2 tree length)) -- relative 1e-15 of a replica's tree length.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "phm_device.h"
#include "phm_rtc.h"
#include "phm_sched.h"

namespace phm {

constexpr int WT_BLOCK = 256;
#ifndef WT_BRANCH_BLOCK_SMALL
#define WT_BRANCH_BLOCK_SMALL 512
#endif
#ifndef WT_BRANCH_BLOCK_BIG             // n > 32: waves sharing the workgroup's LDS (rows of B, reduced counters: 55 KB at 61 states). Eight:
#define WT_BRANCH_BLOCK_BIG 512         // two workgroups = 16 waves per CU, what 110 VGPRs allow (four-wave workgroups: three per CU by LDS, 12.6 -> 8.6 ms on C4)
#endif
constexpr int WT_FEW_TILES = 112;       // n <= 16: below this many tiles the pruning pass runs a wave per 16-replica block (latency) instead of per tile (throughput)
constexpr int WT_BAND_MAX = 2;          // largest half-bandwidth served by the band kernels (tridiagonal: 1; make2sQ hidden rates: 2)
constexpr int WT_BAND_NMAX = 32;        // ... up to this many states (the vectors of a chain live in registers)
constexpr int WT_MAX_SLOTS = 96;        // possible transitions (non-zero entries of B) up to which the branch kernel counts in LDS

// the band of the chain matrix as kernel-argument constants (scalar loads): c[i * (2 hb + 1) + d] = Bc[i][i + d - hb]
struct WtBand {
  double c[WT_BAND_NMAX * (2 * WT_BAND_MAX + 1)];
};

struct WtParams {
  int32_t n_states, ldt;                     // ldt: row stride of the tables (n rounded up to even: 16-byte rows)
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t n_tiles, n_rep, n_rep_pad, replica_offset;
  int32_t normalise, tips_per_replica, ks, tip_masks, reduce, n_cols;
  int32_t klong;                             // rows of the chain tables
  int32_t group, n_groups;                   // branches walked by one wave of the branch kernel; ceil(n_edge / group)
  int32_t band_up;                           // > 0: the chain matrix is banded with this half-bandwidth (<= WT_BAND_MAX, n <= WT_BAND_NMAX): pruning
                                             //   chains as per-lane FMAs over the band (wt_up_band_kernel) instead of the matrix cores
  int32_t band_draw;                         // > 0: the rows of the dense B are banded likewise: a forward draw walks the band of its row only
  int32_t up_form;                           // pruning kernel: 0 chosen by tile count, 1 a wave per (node, tile), 2 split into 16-replica blocks
  uint32_t seed_lo, seed_hi;
  int64_t rows;                              // rows of one tile in one dwell buffer (sum of the slot sizes)
  double fx_scale, fx_inv;                   // fixed-point scale of the dwell accumulators and its inverse (powers of two)
  const double* B2;                          // [n][ldt] dense B, rows of the forward draws
  const double* B2band;                      // [n][2 band_draw + 1]: B2[s][s - band_draw .. s + band_draw] (0 outside the matrix)
  const double* Bc;                          // [n][n] chain matrix (B, or thresholded B for SPARSE), row-major
  const double* scale;                       // [n] 1/(Omega+q_ss)
  const double* pid;                         // [n]
  const UpStep* up;
  const DownStep* down;
  const int32_t* up_order;
  const int32_t* down_order;
  const ClusterNode* cl_nodes;               // deep trees (not null): the tree passes over subtree clusters (phm_sched.h ClusterPlan), tier by tier
  const int32_t* cl_lvl_ptr;                 // [n_clusters + 1] into cl_lvl_off
  const int32_t* cl_lvl_off;                 // per cluster: boundaries of its height levels (positions in cl_nodes)
  const int32_t* branch_order;               // edge rows, largest slot first
  const int32_t* slot;                       // [n_edge + 1] first row of every branch slot
  const double* colL;                        // [klong][n][ldt]  (Bc^k e_j)[r]
  const double* rowL;                        // [klong][n][ldt]  ((Bc^T)^k e_j)[c]
  const double* maskL;                       // [klong][2][ldt]
  const double* blkL;                        // [klong][n s_prev][n end][ldb]: every eighth running sum of the forward draw's probability
                                             //   vector p_c = B2[s][c] * colL[k][e][c] (left to right, unfused): entry q < nblk-1 = the sum
                                             //   after state 8q+7, entry nblk-1 = the total (:301)
  int32_t n_slots;                           // n <= 32 and at most WT_MAX_SLOTS countable pairs (a, c) with B2[a][c] != 0: count in LDS; else 0
  const int16_t* pair_slot;                  // [n*n] pair a*n+c -> slot, -1: none
  const int32_t* slot_col;                   // [n_slots] slot -> counter column (the index the global counters use)
  int32_t nblk, ldb;                         // ceil(n / 8); row stride of blkL (nblk rounded up to even: 16-byte reads)
  const uint8_t* tips;                       // [n_tips] or [tile][n_tips][64]
  uint16_t* mcount;                          // [tile][n_edge][64]
  double* dw[2];                             // [tile][rows][64]; sweep `it` reads dw[it & 1], writes the other
  uint8_t* mstate;                           // [tile][rows][64] states of the merged segments of a branch between the two passes of the branch kernel
  uint16_t* estate;                          // [tile][n_edge][64]: parent-side state | child-side state << 8
  double* PL;                                // [tile][n_node][n][64]
  uint8_t* nstate;                           // [tile][n_node][64]
  unsigned long long* dwfx;                  // [tile][n][64] dwell sums of the sweep, fixed point
  uint32_t* cnt;                             // [tile][n*n][64] transition counters of the sweep
  unsigned long long* dwfx_tile;             // reduced output: [tile][n][16] dwell sums over the lanes of a tile (n > 32; entry lane & 15: four lanes each), fixed point; else NULL
  uint32_t* cnt_tile;                        // reduced output: [tile][n*n] transition counts summed over the lanes of a tile; else NULL
  unsigned long long* segacc;                // [tile][64] segments read + written by the valid replicas (spread over 64 slots)
  double* stats;                             // engine layout: reduce ? [iter][tile][cols] : [iter][cols][n_rep_pad]
  uint32_t* err;
  unsigned long long* segcnt;
};

// the pruning kernel generated for the pattern of an unstructured sparse chain matrix (phm_rtc.h), or none
struct WtSparseUp {
  const SparseUpKernel* kernel = nullptr;
  RtcUpParams params;
};

// phase_ev: optional 5 events, as in launch_tiles_sweep
// tier_off: boundaries of the cluster tiers (empty, or p.cl_nodes null: one launch per tree level)
hipError_t launch_wtiles_sweep(const WtParams& p, const WtBand& band, const WtSparseUp& sparse, const std::vector<int32_t>& up_off,
                               const std::vector<int32_t>& down_off, const std::vector<int32_t>& tier_off, int it, hipStream_t stream,
                               hipEvent_t* phase_ev = nullptr);
// the pruning (up) sweep alone, for bench.py's roofline block
hipError_t launch_wtiles_up(const WtParams& p, const WtBand& band, const WtSparseUp& sparse, const std::vector<int32_t>& up_off,
                            const std::vector<int32_t>& tier_off, hipStream_t stream);

}  // namespace phm
