// phm_narrow.h -- the MCMC sweep for ONE chain (or a handful) on a LARGE tree (n <= 4): the branch mapping.
// The reference's own workflows run a single chain (every driver's default; e.g. the 3 951-tip squamate analysis,
// vignettes/Squamate_DIC_model_selection.Rnw:76-120, 10 000 sweeps of ~880 000 segments each); with one lane per replica
// such a call would walk the whole tree in ONE lane.  Given the node states the branches of a sweep are conditionally
// independent (sampleabranch src/phylomap.cpp:370-413 touches one branch), and the random numbers are addressed by
// (replica, iteration, node | branch), so the work of a sweep can be laid out for the LATENCY of its longest dependent line
// (round 3) with the same results:
//   pruning : subtree CLUSTERS in tiers (phm_sched.h ClusterPlan), one workgroup per cluster, vectors in LDS, eight lanes
//             per node (a quad per child, mat-vec rows across the quad by DPP)                 makePLrcpp* :503-529
//   states  : a TRANSITION MAP per edge (the child's draw for each possible parent state, all edges side by side), then one
//             workgroup per chain walks the depth levels by table look-up                      :618-657, :460-475, :1384-1397
//   branches: eight lanes per branch (a wave for each of the longest): transition maps of the interior change points and the
//             exponential variates wave-wide, then the state machine -- merge, count, virtual jumps -- walked from LDS    :264-413
//   stats   : one row per wavefront of the branch kernel, added in a fixed order by spare workgroups of the next sweep's first
//             launch (the last sweep of a call: a launch of its own)                           :745-757
// Dwell paths live in CSR form (one slot of `cap_b` doubles per branch, two buffers swapped every sweep); the chain powers
// B^k e_j are read from tables that cover every possible segment count (built on the host with the kernels' arithmetic),
// so each state draw costs O(1) instead of the O(m) continuation the LDS tables of phm_mcmc.hip need beyond k = 32.
// Transition counts are bit-identical to the oracle; dwell sums are added per branch and then reduced, so they agree to
// rounding (<= 1e-10 relative, the stated bar) rather than bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

#ifndef PHM_NARROW_CLUSTER_NODES
#define PHM_NARROW_CLUSTER_NODES 256
#endif
#ifndef PHM_NARROW_LONG
#define PHM_NARROW_LONG 128
#endif
#ifndef PHM_NARROW_WIDE_SEGMENTS
#define PHM_NARROW_WIDE_SEGMENTS 96
#endif
constexpr int NARROW_LONG = PHM_NARROW_LONG;                     // the branch kernel gives at least min(n_edge / 16, this many) branches a wave each,
constexpr int NARROW_WIDE_SEGMENTS = PHM_NARROW_WIDE_SEGMENTS;   // and every branch expected to hold this many segments (1 + Omega t_b, or what the caller's path holds)
constexpr int NARROW_CLUSTER_NODES = PHM_NARROW_CLUSTER_NODES;   // internal nodes per pruning cluster (their vectors: 8 KB of LDS at 4 states)
#ifndef PHM_NARROW_CLUSTER_BLOCK
#define PHM_NARROW_CLUSTER_BLOCK 512
#endif
constexpr int NARROW_CLUSTER_BLOCK = PHM_NARROW_CLUSTER_BLOCK;   // eight lanes per node, 64 nodes per pass
#ifndef PHM_NARROW_BRANCH_LANES
#define PHM_NARROW_BRANCH_LANES 8
#endif
constexpr int NARROW_BRANCH_LANES = PHM_NARROW_BRANCH_LANES;   // lanes that share one branch in narrow_branch_kernel
constexpr int NARROW_BLOCK = 64;      // one wavefront per workgroup: latency-bound work spread over as many CUs as possible

template <int NS>
struct NarrowParams {
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t n_rep, n_rep_pad, replica_offset, n_tiles;
  int32_t normalise, tips_per_replica, ks, tip_masks, reduce, n_cols;
  int32_t klong;                             // rows of the long chain tables (> every branch capacity)
  int32_t n_wide;                            // branch kernel: the first n_wide branches of branch_order get a wavefront each
  int32_t cluster_async;                     // pruning sweep: the dependency-driven form of the cluster kernel (long chains)
  uint32_t seed_lo, seed_hi;
  int64_t total_cap;                         // doubles per replica in one dwell buffer
  double B2[NS * NS], Bc[NS * NS], scale[NS], pid[NS];
  const ClusterNode* cl_nodes;               // pruning sweep: clusters of <= NARROW_CLUSTER_NODES nodes, tier by tier (phm_sched.h)
  const int32_t* cl_item_off;                // [n_clusters + 1] into cl_nodes
  const int32_t* cl_lvl_ptr;                 // [n_clusters + 1] into cl_lvl_off
  const int32_t* cl_lvl_off;                 // per cluster: boundaries of its height levels (positions in cl_nodes)
  const DownStep* down_lv;                   // [n_edge] sampling steps: edges to internal nodes grouped by depth level, then the tip edges
  const int32_t* walk_off;                   // depth-level boundaries of the first part
  const int32_t* edge_parent;                // [n_edge] internal index of the parent node of every edge row
  const int32_t* branch_order;               // edge rows, largest capacity first
  const int64_t* off;                        // [n_edge + 1] CSR offsets of the branch slots
  const double* colL;                        // [klong][NS][NS]  (Bc^k e_j)[r]
  const double* rowL;                        // [klong][NS][NS]  ((Bc^T)^k e_j)[c]
  const double* maskL;                       // [klong][2][NS]   Bc^k applied to the even / odd state mask (ks)
  const uint8_t* tips;                       // [n_tips] or [replica][n_tips], 0-based
  int32_t* mcount;                           // [replica][n_edge]
  double* dw[2];                             // [replica][total_cap] each; sweep `it` reads dw[it & 1], writes the other
  double* PL;                                // [replica][n_node][NS]
  uint8_t* nstate;                           // [replica][n_node]
  uint16_t* dmap;                            // [replica][n_edge] transition map of every edge (sampling sweep), down_lv order
  uint16_t* dmap_edge;                       // the same maps by edge row (branch kernel: end states of its edge)
  double* part;                              // [replica][n_edge][NS + NS*NS + 1] per-branch dwell sums, counts, segments touched
  double* rowbuf;                            // [replica][n_cols] statistics row of the sweep
  double* stats;                             // engine layout: reduce ? [iter][tile][cols] : [iter][cols][n_rep_pad]
  uint32_t* err;
  unsigned long long* segcnt;
  double* host_row;                          // page-locked host memory (or null): the statistics kernel of a ONE-chain engine leaves the
                                             // row, the error word and the segment counter there too -- n_cols + 2 doubles; a loop that
                                             // reads one row per sweep (the rate-updating drivers) then needs no device-to-host copy
};

// one full sweep (iteration index `it`) enqueued on `stream`; tier boundaries (clusters) and depth-level boundaries are host arrays
template <int NS>
hipError_t launch_narrow_sweep(const NarrowParams<NS>& p, const std::vector<int32_t>& tier_off,
                               const std::vector<int32_t>& walk_off, int it, hipStream_t stream, bool stats_pending);
// statistics of the last sweep of a call (the others are added up in the first launch of the sweep that follows)
template <int NS>
hipError_t launch_narrow_stats(const NarrowParams<NS>& p, int it, hipStream_t stream);

}  // namespace phm
