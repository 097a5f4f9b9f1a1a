// phm_mcmc.h -- kernel parameter block and launchers of the fixed-Q MCMC sweep (phm_mcmc.hip)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

constexpr int MCMC_BLOCK = 256;   // 4 wavefronts share one copy of the LDS tables
constexpr int MCMC_KTAB = 32;     // rows of the column-chain table B^k e_j kept in LDS; longer chains read the full-length table

// Passed by value: lives in the kernarg segment, so B / pid are read through scalar loads.
template <int NS>
struct McmcParams {
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t n_tiles, n_rep, n_rep_pad, replica_offset;
  int32_t normalise, tips_per_replica, reduce, n_cols;
  int32_t ktab;                              // rows of the column-chain table staged in LDS (= MCMC_KTAB, checked at launch)
  int32_t klong;                             // rows of the tables in global memory (>= ktab; covers every possible segment count)
  int32_t ks;                                // 1: bf/ks layout: n x n counts incl. self pairs (shortenerbf), root-state column
  int32_t tip_masks;                         // 1 (ks): tips observed up to parity and re-sampled; 0 (bf): tips observed
  int32_t prune_only;                        // measurement aid: run only the pruning (up) sweep of each iteration
  int32_t tiles_per_tree;                    // 0: one tree; else tile t walks tree t / tiles_per_tree (up, down hold one
  const int32_t* roots;                      //    schedule per tree back to back, roots[tree] the internal root index)
  uint32_t seed_lo, seed_hi;
  int64_t rows;                              // capacity (64-lane rows) of one tile's dwell stream
  double B2[NS * NS];                        // dense B = I + Q/Omega, row-major
  double Bc[NS * NS];                        // chain matrix: B, or B with entries <= 1e-7 dropped (SPARSE)
  double scale[NS];                          // 1/(Omega + q_ss): Rcpp::rexp(n, rate) multiplies by 1/rate
  double pid[NS];
  const UpStep* up;
  const DownStep* down;
  const double* colpow;                      // [klong][NS][NS]: (Bc^k e_j)[r]; the first ktab rows are copied to LDS
  const double* rowpow;                      // [klong][NS][NS]: ((Bc^T)^k e_j)[c]; read from here (one row per node draw)
  const double* maskpow;                     // [klong][2][NS]: Bc^k applied to the even / odd state mask (ks only)
  const uint8_t* tips;                       // 0-based tip states: [n_tips] or [tile][n_tips][64]
  uint16_t* mcount;                          // [tile][n_edge][64] segments per branch
  double* dwell0;                            // [tile][rows][64] ring holding the consumed and the produced dwell stream
  double* dwell1;                            // NULL: ring mode; else the second buffer of the two-buffer mode
  int32_t* cursor;                           // [tile][2]: ring: first row of the current stream, first free row after it;
                                             //            two buffers: {sweep parity, unused}
  double* PL;                                // [tile][n_node][NS][64] internal nodes only
  uint8_t* nstate;                           // [tile][n_node][64] sampled internal-node states
  double* stats;                             // reduce: [iter][tile][cols]; else [iter][cols][n_rep_pad]
  uint32_t* err;
  unsigned long long* segcnt;
};

template <int NS> size_t mcmc_lds_bytes(int ktab, bool ks);
template <int NS> hipError_t launch_mcmc(const McmcParams<NS>& p, int iter0, int n_iters, hipStream_t stream);

hipError_t launch_mcmc_init(int n_edge, int n_tiles, int64_t rows, const DownStep* down, const int32_t* init_row,
                            const int32_t* map_off, const double* maps, double* dwell0, uint16_t* mcount,
                            hipStream_t stream);
// the statistics columns of the replicas that exist, packed (few replicas on padded tiles: a read of the padded rows moves 64x the data)
hipError_t launch_stats_gather(const double* stats, int64_t n_rows, int n_rep_pad, int n_pick, const int32_t* pick, double* out, hipStream_t stream);
hipError_t launch_stats_reduce(const double* partial, int n_iters, int n_tiles, int n_cols, double* out,
                               hipStream_t stream, const double* init = nullptr);

}  // namespace phm
