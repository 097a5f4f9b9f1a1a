// phm_tiles.h -- the MCMC sweep for a MODERATE number of replicas (n <= 4): one wavefront per (tile of 64 replicas, branch).
//
// The replica mapping of phm_mcmc.hip gives every tile of 64 replicas ONE wave that walks the whole tree, so it needs
// tens of thousands of replicas to fill the 1 024 SIMDs of an MI355X; an alignment of 10^3 - 10^4 sites leaves it mostly
// idle (measured on C2 at 4 096 replicas: 30 ms per sweep, 0.27 G realisations/s).  Here the lanes are still replicas --
// every access stays a contiguous 512-byte row, the 64 lanes of a wave still run the same control flow on the same branch
// -- but the branches of a tile are spread over different waves: given the node states they are conditionally independent
// and the random numbers are addressed by (replica, iteration, node | branch), so the results do not depend on the order.
//   up     : one launch per HEIGHT level, a wave per (tile, internal node)          makePLrcpp* :503-529
//   root   : a wave per tile                                                       :618-627
//   down   : one launch per DEPTH level, a wave per (tile, edge)                   :640-657, :460-475
//   branch : a wave per (tile, branch)                                             :264-413, :44-73, :745-757
//   stats  : fixed-order reduction of the per-branch dwell sums (two stages), integer counters
// Few tiles (10^2 .. 10^3 replicas: the sites of an alignment): a launch per tree level is 60+ launches of a handful of waves each
// on a 10 000-tip tree.  There the two tree passes run over CLUSTERS cut by height (build_band_plan, phm_sched.h): tier k = the nodes
// of heights [8k, 8k + 8), a cluster = a maximal subtree inside its tier; a workgroup per (cluster, tile) walks the cluster's levels
// with a workgroup barrier between them -- up: one launch per tier; down: the tiers in reverse, root draw folded into the first --
// 4 + 4 launches on C3 instead of 31 + 1 + 31, and the levels walked one after the other still number the height of the tree.
// Same draws, same bits.  C3 at 64 / 256 replicas: 0.86 / 1.03 -> 0.36 / 0.69 ms per sweep (with the counter copies below).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "phm_device.h"
#include "phm_sched.h"

namespace phm {

constexpr int TILES_BLOCK = 256;          // four waves = four (tile, item) pairs per workgroup
constexpr int TILES_CHUNK = 64;           // group partials per second-stage sum of the dwell reduction
constexpr int TILES_PERSISTENT_WGS = 2048;      // workgroups of a node-draw launch: 256 CUs x 8 (four waves each: eight waves per SIMD)
constexpr int TILES_KTAB = 24;            // chain-table rows staged in LDS by the branch kernel (longer chains: full table in L2)
constexpr int TILES_CL_BLOCK = 512;       // cluster kernels: eight waves walk the levels of one (cluster, tile) with workgroup barriers
constexpr int TILES_CL_NODES = 256;       // internal nodes per cluster when cut by subtree size (build_cluster_plan; measurement only)
#ifndef TILES_CL_BAND
#define TILES_CL_BAND 8                   // height levels per tier (build_band_plan): 5 / 8 / 12 measured, profiles/r04_probe_level_groups.log
#endif
// automatic choice: clusters while tiles x internal nodes stays below this (C3 up to 6 tiles, C2 up to 64), one launch per level beyond
constexpr int64_t TILES_CL_MAX_WORK = 65536;

template <int NS>
struct TileParams {
  int32_t n_tips, n_node, n_edge, root;      // root: internal index
  int32_t n_tiles, n_rep, n_rep_pad, replica_offset;
  int32_t normalise, tips_per_replica, ks, tip_masks, reduce, n_cols;
  int32_t klong;                             // rows of the long chain tables
  int32_t group, n_groups;                   // branches walked by one wave of the branch kernel; ceil(n_edge / group)
  int32_t n_chunks;                          // ceil(n_groups / TILES_CHUNK)
  uint32_t seed_lo, seed_hi;
  int64_t rows;                              // rows of one tile in one dwell buffer (sum of the slot sizes)
  double B2[NS * NS], Bc[NS * NS], scale[NS], pid[NS];
  const UpStep* up;
  const DownStep* down;
  const int32_t* up_order;                   // positions into up[], grouped by height level
  const int32_t* down_order;                 // positions into down[], grouped by depth level
  const ClusterNode* cl_nodes;               // subtree clusters (phm_sched.h ClusterPlan), or NULL: one launch per tree level
  const int32_t* cl_lvl_ptr;                 // [n_clusters + 1] into cl_lvl_off
  const int32_t* cl_lvl_off;                 // per cluster: boundaries of its height levels (positions in cl_nodes)
  const int32_t* branch_order;               // edge rows, largest slot first
  const int32_t* slot;                       // [n_edge + 1] first row of every branch slot
  const double* colL;                        // [klong][NS][NS]
  const double* rowL;                        // [klong][NS][NS]
  const double* maskL;                       // [klong][2][NS]
  const uint8_t* tips;                       // [n_tips] or [tile][n_tips][64]
  uint16_t* mcount;                          // [tile][n_edge][64]
  double* dw[2];                             // [tile][rows][64]; sweep `it` reads dw[it & 1], writes the other
  uint8_t* estate;                           // [tile][n_edge][64]: parent-side state | child-side state << 4
  uint8_t* mstate;                           // long paths (not null): [tile][rows][64] states of the merged segments of the branch at hand,
                                             //   a byte per (row, lane) beside the dwell rows (tiles_branch_kernel<NS, KS, true>)
  double* PL;                                // [tile][n_node][NS][64]
  uint8_t* nstate;                           // [tile][n_node][64]
  double* pdw;                               // [tile][n_edge][NS][64] dwell sums of every group of branches (n_groups rows used)
  double* pchunk;                            // [tile][n_chunks][NS][64] first-stage sums
  uint32_t* cnt;                             // [tile][cnt_copies][NS*NS][64] transition counters of the sweep (integer atomics)
  int32_t cnt_copies;                        // power of two: the waves of a tile spread their atomics over this many copies (few tiles:
                                             // 20 000 waves adding to ONE tile's 12 counter rows serialise in L2 -- 0.35 of 0.86 ms per sweep on C3 at 64 replicas)
  uint32_t* pseg;                            // [tile][n_chunks][64] segments held by each chunk of branches after the sweep
  uint32_t* segprev;                         // [tile] segments held by the tile's valid replicas before the sweep
  double* stats;                             // engine layout: reduce ? [iter][tile][cols] : [iter][cols][n_rep_pad]
  uint32_t* err;
  unsigned long long* segcnt;
};

// initial paths -> slot rows of buffer 0 and the segment counts, every lane of every tile
hipError_t launch_tiles_init(int n_edge, int n_tiles, int64_t rows, const int32_t* slot, const int32_t* map_off,
                             const double* maps, double* dw0, uint16_t* mcount, hipStream_t stream);

// phase_ev: optional 5 events recorded before the pruning levels and after the pruning levels, the node draws, the branch kernel
// and the reductions (measurement: bench.py's per-kernel roofline)
// tier_off: cluster tiers (ClusterPlan::tier_off) when p.cl_nodes is set
template <int NS>
hipError_t launch_tiles_sweep(const TileParams<NS>& p, const std::vector<int32_t>& up_off,
                              const std::vector<int32_t>& down_off, const std::vector<int32_t>& tier_off, int it, hipStream_t stream,
                              hipEvent_t* phase_ev = nullptr);

}  // namespace phm
