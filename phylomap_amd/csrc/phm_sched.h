// phm_sched.h -- host-side tree validation and sweep schedules.
//
// Replaces, in O(E), what the reference gets from R: the pruningwise edge order `nen`, the top-down
// `nodelist` and `root` (R/sumstatMCMC.R:1-18, interpreted O(E^2) loops), plus the per-node linear edge
// search inside the C++ node sweep (src/phylomap.cpp:643).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace phm {

struct UpStep {        // one internal node of the post-order (tips-to-root) sweep
  int32_t parent;      // internal index (node id - n_tips - 1)
  int32_t child[2];    // >= 0: internal index; < 0: ~tip (tip index 0-based)
  int32_t edge[2];     // edge rows (0-based) leading to child[0], child[1]
};

struct DownStep {      // one branch of the pre-order (root-to-tips) sweep
  int32_t edge;        // edge row (0-based)
  int32_t parent;      // internal index of the parent node
  int32_t child;       // >= 0 internal index; < 0: ~tip
  int32_t pad;
};

// Pruning sweep of the one-chain mapping (phm_narrow.hip): the internal nodes cut into CLUSTERS -- maximal subtrees of at most
// `max_nodes` not-yet-assigned nodes -- in TIERS: tier 0 holds the bottom subtrees, tier 1 the maximal subtrees of what is left,
// ... (two tiers for a 10 000-tip tree at 256 nodes per cluster).  Clusters of one tier are independent (one workgroup each, one
// launch per tier); inside a cluster the nodes are ordered by height and a child of the same cluster is addressed by its position
// in the cluster (its partial-likelihood vector stays in LDS).
struct ClusterNode {
  int32_t parent;      // internal index
  int32_t child[2];    // >= 0: internal index; < 0: ~tip
  int32_t edge[2];     // edge rows (0-based)
  int32_t slot[2];     // child in the same cluster: its position in the cluster; else -1
  int32_t pad;
};

struct ClusterPlan {
  std::vector<ClusterNode> nodes;       // all clusters, tier by tier; inside a cluster by height
  std::vector<int32_t> item_off;        // [n_clusters + 1] into nodes
  std::vector<int32_t> lvl_ptr;         // [n_clusters + 1] into lvl_off
  std::vector<int32_t> lvl_off;         // per cluster: boundaries of its height levels (absolute positions in nodes), levels + 1 entries
  std::vector<int32_t> tier_off;        // [n_tiers + 1] into the cluster list
};

struct Schedule;
void build_cluster_plan(const Schedule& s, int32_t max_nodes, ClusterPlan& plan);
// The same structure cut by HEIGHT: tier k holds the nodes of heights [k band, (k + 1) band), a cluster is a maximal subtree inside
// its tier (at most 2^band - 1 nodes).  The levels walked one after the other over all tiers then number exactly the height of the
// tree (subtree clusters by size walk up to the height of a cluster per tier); the last tier is one cluster and holds the root.
void build_band_plan(const Schedule& s, int32_t band, ClusterPlan& plan);

struct Schedule {
  int32_t n_tips = 0, n_node = 0, n_edge = 0;
  int32_t root = 0;                 // internal index of the root
  std::vector<UpStep> up;           // n_node entries, children before parents
  std::vector<DownStep> down;       // n_edge entries, parents before children
  std::vector<int32_t> edge_of_child;  // node id (0-based) -> edge row, -1 for the root
  bool down_is_row_order = false;   // true when x$edge already is a valid pre-order (e.g. ape cladewise)
};

// Validates a strictly bifurcating rooted tree (src/phylomap.cpp:508-510 assumes it) and builds the
// sweeps. `edge` is n_edge x 2 column-major, 1-based. Returns false and fills `err` on malformed input.
bool build_schedule(int32_t n_tips, int32_t n_node, int32_t n_edge, const int32_t* edge, Schedule& s,
                    std::string& err);

// Checks that the caller's nen / nodelist / root (R/sumstatMCMC.R:1-18) describe this tree:
// nen a permutation with sibling edges adjacent and children before parents, nodelist parents before
// children, root the node that is nobody's child.
bool check_reference_orders(const Schedule& s, const int32_t* edge, const int32_t* nen,
                            const int32_t* nodelist, int32_t root, std::string& err);

// O(E) native replacement of the R helper preamble (R/sumstatMCMC.R:1-18): pruningwiseedgeorder(), makenodelist(),
// myreorder().  ape's "pruningwise" order restated from its published scan algorithm: a node is collected in scan
// number height(node); within one scan, in the order its last child edge is met in the cladewise edge table.
// nen: n_edge 1-based rows; nodelist: n_node-1 node ids (root side first); root: node id.
bool pruningwise_orders(int32_t n_tips, int32_t n_edge, const int32_t* edge, int32_t* nen, int32_t* nodelist,
                        int32_t* root, std::string& err);

// Upper quantile of the segment count 1 + Poisson(lambda) of a branch in stationarity:
// 1 + smallest c with P(Poisson(lambda) >= c) < tail.
int32_t poisson_capacity(double lambda, double tail);

}  // namespace phm
