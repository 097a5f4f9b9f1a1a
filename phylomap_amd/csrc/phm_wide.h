// phm_wide.h -- parameter block and launchers of the 5..64-state MCMC sweep (phm_wide.hip)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

constexpr int WIDE_BLOCK = 384;     // 6 wavefronts share the LDS copy of B (two workgroups per CU at n = 61)
constexpr int WIDE_KTAB = 16;       // B^k e_j tables in global memory for k < KTAB
constexpr int WIDE_MAXSEG = 128;    // most segments one branch of one replica may hold (LDS state scratch) ...
constexpr int WIDE_MAXSEG_BIG_N = 128;
inline int wide_maxseg(int n) { return n > 40 ? WIDE_MAXSEG_BIG_N : WIDE_MAXSEG; }

struct WideParams {
  int32_t n_states;
  int32_t n_tips, n_node, n_edge, root;
  int32_t n_tiles, n_rep, n_rep_pad, replica_offset;
  int32_t normalise, tips_per_replica, reduce, n_cols, ktab;
  int32_t sparse;                            // 1: Bc != B2 (SPARSE variant), both staged in LDS
  int32_t ks;                                // 1: bf/ks layout (n x n counts incl. self pairs, root-state column)
  int32_t tip_masks;                         // 1 (ks): parity tip masks, tips re-sampled
  int32_t count_self;                        // 1: n x n transition counts incl. self pairs (shortenerbf :1010-1014)
  int32_t rep_stride;                        // replicas placed on one tile (1..64; 64 = dense), single tree only
  int32_t tiles_per_tree;                    // 0: one tree; else tile t walks tree t / tiles_per_tree (up, down hold one
  const int32_t* roots;                      //    schedule per tree back to back, roots[tree] the internal root index)
  uint32_t seed_lo, seed_hi;
  int64_t rows;
  const double* B2;                          // [n][n] dense B, row-major
  const double* Bc;                          // [n][n] chain matrix (B, or thresholded B for SPARSE)
  const double* scale;                       // [n] 1/(Omega+q_ss)
  const double* pid;                         // [n]
  const UpStep* up;
  const DownStep* down;
  const double* colpow;                      // [ktab][n][n]
  const double* rowpow;                      // [ktab][n][n]
  const double* maskpow;                     // [ktab][2][n] (ks)
  const uint8_t* tips;
  uint16_t* mcount;                          // [tile][n_edge][64]
  double* dwell0;                            // [tile][rows][64] ring (consumed stream + produced stream)
  int32_t* cursor;                           // [tile][2]
  double* PL;                                // [tile][n_node][64][n]  (a replica's vector is contiguous)
  uint8_t* nstate;                           // [tile][n_node][64]
  double* stats;                             // reduce: [iter][tile][cols] (atomics); else [iter][cols][n_rep_pad]; zeroed at create
  uint32_t* err;
  unsigned long long* segcnt;
};

size_t wide_lds_bytes(int n, bool sparse);
hipError_t launch_mcmc_wide(const WideParams& p, int iter0, int n_iters, hipStream_t stream);

}  // namespace phm
