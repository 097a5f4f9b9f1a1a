// phm_sched.cpp -- see phm_sched.h
#include "phm_sched.h"

#include <algorithm>
#include <cmath>

namespace phm {

bool build_schedule(int32_t T, int32_t n_node, int32_t E, const int32_t* edge, Schedule& s, std::string& err) {
  if (T < 2 || n_node != T - 1 || E != 2 * T - 2) {
    err = "tree must be strictly bifurcating: Nnode = n_tips-1 and nrow(edge) = 2*n_tips-2";
    return false;
  }
  const int32_t nn = 2 * T - 1;
  const int32_t* e1 = edge;
  const int32_t* e2 = edge + E;
  s.n_tips = T; s.n_node = n_node; s.n_edge = E;
  s.edge_of_child.assign(nn, -1);
  std::vector<int32_t> kids(2 * (size_t)n_node, -1);
  for (int32_t r = 0; r < E; ++r) {
    int32_t p = e1[r], c = e2[r];
    if (p <= T || p > nn || c < 1 || c > nn || c == p) { err = "edge row " + std::to_string(r + 1) + ": node id out of range"; return false; }
    if (s.edge_of_child[c - 1] != -1) { err = "node " + std::to_string(c) + " has two parents"; return false; }
    s.edge_of_child[c - 1] = r;
    int32_t pi = p - T - 1;
    if (kids[2 * pi] == -1) kids[2 * pi] = r;
    else if (kids[2 * pi + 1] == -1) kids[2 * pi + 1] = r;
    else { err = "node " + std::to_string(p) + " has more than two children"; return false; }
  }
  int32_t root = -1;
  for (int32_t v = T; v < nn; ++v) {
    if (kids[2 * (v - T) + 1] == -1) { err = "internal node " + std::to_string(v + 1) + " does not have two children"; return false; }
    if (s.edge_of_child[v] == -1) {
      if (root != -1) { err = "tree has more than one root"; return false; }
      root = v - T;
    }
  }
  if (root == -1) { err = "tree has no root"; return false; }
  s.root = root;

  // Is the edge table already a pre-order (every parent seen as a child earlier, or the root)?
  std::vector<char> seen(nn, 0);
  seen[root + T] = 1;
  bool row_order = true;
  for (int32_t r = 0; r < E && row_order; ++r) {
    if (!seen[e1[r] - 1]) row_order = false;
    seen[e2[r] - 1] = 1;
  }
  std::vector<int32_t> order;
  order.reserve(E);
  if (row_order) {
    for (int32_t r = 0; r < E; ++r) order.push_back(r);
  } else {
    std::vector<int32_t> stack;
    stack.push_back(kids[2 * root + 1]);
    stack.push_back(kids[2 * root]);
    while (!stack.empty()) {
      int32_t r = stack.back(); stack.pop_back();
      order.push_back(r);
      int32_t c = e2[r];
      if (c > T) { stack.push_back(kids[2 * (c - T - 1) + 1]); stack.push_back(kids[2 * (c - T - 1)]); }
    }
  }
  if ((int32_t)order.size() != E) { err = "edge table is not a connected tree"; return false; }
  s.down_is_row_order = row_order;

  s.down.resize(E);
  std::vector<int32_t> internal_order;   // internal nodes in the order the down sweep reaches them
  internal_order.reserve(n_node);
  internal_order.push_back(root);
  std::vector<char> reached(nn, 0);
  reached[root + T] = 1;
  for (int32_t k = 0; k < E; ++k) {
    int32_t r = order[k];
    if (!reached[e1[r] - 1]) { err = "edge table is not a connected tree"; return false; }
    reached[e2[r] - 1] = 1;
    DownStep& d = s.down[k];
    d.edge = r;
    d.parent = e1[r] - T - 1;
    d.child = (e2[r] > T) ? (e2[r] - T - 1) : ~(e2[r] - 1);
    d.pad = 0;
    if (e2[r] > T) internal_order.push_back(e2[r] - T - 1);
  }
  s.up.resize(n_node);
  for (int32_t k = 0; k < n_node; ++k) {     // reverse pre-order: children always before parents
    int32_t v = internal_order[n_node - 1 - k];
    UpStep& u = s.up[k];
    u.parent = v;
    for (int j = 0; j < 2; ++j) {
      int32_t r = kids[2 * v + j];
      u.edge[j] = r;
      u.child[j] = (e2[r] > T) ? (e2[r] - T - 1) : ~(e2[r] - 1);
    }
  }
  return true;
}

bool check_reference_orders(const Schedule& s, const int32_t* edge, const int32_t* nen,
                            const int32_t* nodelist, int32_t root, std::string& err) {
  const int32_t T = s.n_tips, E = s.n_edge, nn = 2 * T - 1;
  const int32_t* e1 = edge;
  const int32_t* e2 = edge + E;
  if (root != s.root + T + 1) { err = "root does not match the edge table"; return false; }
  if (nen) {
    std::vector<char> used(E, 0), done(nn, 0);
    for (int32_t v = 0; v < T; ++v) done[v] = 1;
    for (int32_t i = 0; i < s.n_node; ++i) {
      int32_t a = nen[2 * i] - 1, b = nen[2 * i + 1] - 1;
      if (a < 0 || a >= E || b < 0 || b >= E || used[a] || used[b] || a == b) { err = "nen is not a permutation of the edge rows"; return false; }
      used[a] = used[b] = 1;
      if (e1[a] != e1[b]) { err = "nen: edges at positions " + std::to_string(2 * i + 1) + "," + std::to_string(2 * i + 2) + " are not siblings"; return false; }
      if (!done[e2[a] - 1] || !done[e2[b] - 1]) { err = "nen: a parent is pruned before its children"; return false; }
      done[e1[a] - 1] = 1;
    }
  }
  if (nodelist) {
    std::vector<char> have(nn, 0);
    have[root - 1] = 1;
    for (int32_t i = 0; i < s.n_node - 1; ++i) {
      int32_t v = nodelist[i];
      if (v <= T || v > nn || have[v - 1]) { err = "nodelist must list every non-root internal node once"; return false; }
      int32_t r = s.edge_of_child[v - 1];
      if (r < 0 || !have[e1[r] - 1]) { err = "nodelist: a node comes before its parent"; return false; }
      have[v - 1] = 1;
    }
  }
  return true;
}

bool pruningwise_orders(int32_t T, int32_t E, const int32_t* edge, int32_t* nen, int32_t* nodelist, int32_t* root_out,
                        std::string& err) {
  if (T < 2 || E < 2 || !edge) { err = "bad tree"; return false; }
  const int32_t* e1 = edge;
  const int32_t* e2 = edge + E;
  int32_t nn = 0;
  for (int32_t r = 0; r < E; ++r) { nn = std::max(nn, std::max(e1[r], e2[r])); if (e1[r] < 1 || e2[r] < 1) { err = "node ids must be >= 1"; return false; } }
  std::vector<std::vector<int32_t>> kids(nn + 1);
  std::vector<char> is_child(nn + 1, 0);
  for (int32_t r = 0; r < E; ++r) { kids[e1[r]].push_back(r); is_child[e2[r]] = 1; }
  int32_t root = -1;
  for (int32_t v = 1; v <= nn; ++v)
    if (!kids[v].empty() && !is_child[v]) { if (root != -1) { err = "tree has more than one root"; return false; } root = v; }
  if (root == -1) { err = "tree has no root"; return false; }
  // cladewise position of every edge: pre-order, children in row order
  std::vector<int32_t> pos(E, -1), by_pos;
  by_pos.reserve(E);
  std::vector<int32_t> stack(kids[root].rbegin(), kids[root].rend());
  while (!stack.empty()) {
    int32_t r = stack.back(); stack.pop_back();
    if (pos[r] != -1) { err = "edge table is not a tree"; return false; }
    pos[r] = (int32_t)by_pos.size();
    by_pos.push_back(r);
    const auto& k = kids[e2[r]];
    for (auto it = k.rbegin(); it != k.rend(); ++it) stack.push_back(*it);
  }
  if ((int32_t)by_pos.size() != E) { err = "edge table is not a connected rooted tree"; return false; }
  std::vector<int32_t> height(nn + 1, 0), lastpos(nn + 1, -1);
  for (int32_t i = E - 1; i >= 0; --i) {
    int32_t r = by_pos[i];
    height[e1[r]] = std::max(height[e1[r]], height[e2[r]] + 1);
    lastpos[e1[r]] = std::max(lastpos[e1[r]], pos[r]);
  }
  std::vector<int32_t> internal;
  for (int32_t v = 1; v <= nn; ++v) if (!kids[v].empty() && v != root) internal.push_back(v);
  std::sort(internal.begin(), internal.end(), [&](int32_t a, int32_t b) {
    return height[a] != height[b] ? height[a] < height[b] : lastpos[a] < lastpos[b];
  });
  internal.push_back(root);
  std::vector<int32_t> rows;
  rows.reserve(E);
  for (int32_t v : internal) {
    std::vector<int32_t> k = kids[v];
    std::sort(k.begin(), k.end(), [&](int32_t a, int32_t b) { return pos[a] < pos[b]; });
    for (int32_t r : k) rows.push_back(r);
  }
  for (int32_t i = 0; i < E; ++i) nen[i] = rows[i] + 1;
  const int32_t n_node = (int32_t)internal.size();
  for (int32_t i = 1; i < n_node; ++i) nodelist[i - 1] = e1[rows[E - 2 * i - 1]];     // R/sumstatMCMC.R:11-17
  *root_out = e1[rows[E - 1]];                                                          // :18
  return true;
}

int32_t poisson_capacity(double lambda, double tail) {
  if (!(lambda > 0.0)) return 1;
  if (!(tail > 0.0)) tail = 1e-16;
  const double ltail = std::log(tail);
  // P(X >= c) <= pmf(c) / (1 - lambda/(c+1)) for c+1 > lambda
  int64_t c = (int64_t)std::ceil(lambda) + 1;
  for (;; ++c) {
    double lp = -lambda + (double)c * std::log(lambda) - std::lgamma((double)c + 1.0);
    double ratio = lambda / ((double)c + 1.0);
    if (ratio < 1.0 && lp - std::log1p(-ratio) < ltail) break;
    if (c > (int64_t)(lambda * 4 + 4096)) break;
  }
  int64_t cap = c + 1 /* base segment */;
  if (cap > 60000) cap = 60000;
  return (int32_t)cap;
}

void build_cluster_plan(const Schedule& s, int32_t max_nodes, ClusterPlan& plan) {
  const int Nn = s.n_node;
  plan = ClusterPlan();
  // s.up lists children before parents: one pass gives heights and the parent of every internal node
  std::vector<int32_t> height(Nn, 0), parent(Nn, -1), step_of(Nn, -1);
  for (int k = 0; k < Nn; ++k) {
    const UpStep& u = s.up[k];
    step_of[u.parent] = k;
    int h = 0;
    for (int c = 0; c < 2; ++c)
      if (u.child[c] >= 0) { h = std::max(h, height[u.child[c]] + 1); parent[u.child[c]] = u.parent; }
    height[u.parent] = h;
  }
  std::vector<int32_t> cluster_of(Nn, -1), rsize(Nn, 0);
  std::vector<std::vector<int32_t>> members;             // per cluster: internal indices
  std::vector<int32_t> croot;                            // per cluster: its root node
  plan.tier_off.push_back(0);
  int assigned = 0;
  while (assigned < Nn) {
    // sizes of the not-yet-assigned part of every subtree (children before parents)
    for (int k = 0; k < Nn; ++k) {
      const UpStep& u = s.up[k];
      if (cluster_of[u.parent] >= 0) { rsize[u.parent] = 0; continue; }
      int sz = 1;
      for (int c = 0; c < 2; ++c) if (u.child[c] >= 0) sz += rsize[u.child[c]];
      rsize[u.parent] = sz;
    }
    const int first = (int)members.size();
    // parents before children: a node joins its parent's cluster of this tier, or starts one when its subtree fits
    for (int k = Nn - 1; k >= 0; --k) {
      const int v = s.up[k].parent;
      if (cluster_of[v] >= 0) continue;
      const int pv = parent[v];
      if (pv >= 0 && cluster_of[pv] >= first) { cluster_of[v] = cluster_of[pv]; members[cluster_of[v]].push_back(v); ++assigned; continue; }
      if (rsize[v] <= max_nodes) {
        cluster_of[v] = (int)members.size();
        members.emplace_back(1, v);
        croot.push_back(v);
        ++assigned;
      }
    }
    // big clusters first: they bound the duration of the tier's launch
    std::vector<int32_t> order((int)members.size() - first);
    for (size_t i = 0; i < order.size(); ++i) order[i] = first + (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return members[a].size() > members[b].size(); });
    std::vector<std::vector<int32_t>> sorted_m;
    for (int c : order) sorted_m.push_back(std::move(members[c]));
    for (size_t i = 0; i < order.size(); ++i) {
      members[first + i] = std::move(sorted_m[i]);
      for (int v : members[first + i]) cluster_of[v] = first + (int)i;
    }
    plan.tier_off.push_back((int)members.size());
  }
  std::vector<int32_t> pos(Nn, -1);
  plan.item_off.push_back(0);
  plan.lvl_ptr.push_back(0);
  for (size_t c = 0; c < members.size(); ++c) {
    std::vector<int32_t>& m = members[c];
    std::stable_sort(m.begin(), m.end(), [&](int a, int b) { return height[a] < height[b]; });
    const int base = (int)plan.nodes.size();
    for (size_t i = 0; i < m.size(); ++i) pos[m[i]] = (int)i;
    for (size_t i = 0; i < m.size(); ++i) {
      const UpStep& u = s.up[step_of[m[i]]];
      ClusterNode nd;
      nd.parent = u.parent; nd.pad = 0;
      for (int k = 0; k < 2; ++k) {
        nd.child[k] = u.child[k]; nd.edge[k] = u.edge[k];
        nd.slot[k] = (u.child[k] >= 0 && cluster_of[u.child[k]] == (int)c) ? pos[u.child[k]] : -1;
      }
      if (i == 0 || height[m[i]] != height[m[i - 1]]) plan.lvl_off.push_back(base + (int)i);
      plan.nodes.push_back(nd);
    }
    plan.lvl_off.push_back(base + (int)m.size());
    plan.item_off.push_back((int)plan.nodes.size());
    plan.lvl_ptr.push_back((int)plan.lvl_off.size());
  }
}

void build_band_plan(const Schedule& s, int32_t band, ClusterPlan& plan) {
  const int Nn = s.n_node;
  plan = ClusterPlan();
  if (band < 1) band = 1;
  std::vector<int32_t> height(Nn, 0), parent(Nn, -1), step_of(Nn, -1);
  int max_h = 0;
  for (int k = 0; k < Nn; ++k) {
    const UpStep& u = s.up[k];
    step_of[u.parent] = k;
    int h = 0;
    for (int c = 0; c < 2; ++c)
      if (u.child[c] >= 0) { h = std::max(h, height[u.child[c]] + 1); parent[u.child[c]] = u.parent; }
    height[u.parent] = h;
    max_h = std::max(max_h, h);
  }
  const int n_tiers = max_h / band + 1;
  // parents before children: a node joins its parent's cluster when both lie in the same band, else it starts a cluster
  std::vector<int32_t> cluster_of(Nn, -1);
  std::vector<std::vector<int32_t>> members;
  std::vector<int32_t> tier_of_cluster;
  for (int k = Nn - 1; k >= 0; --k) {
    const int v = s.up[k].parent, pv = parent[v];
    if (pv >= 0 && height[pv] / band == height[v] / band) cluster_of[v] = cluster_of[pv];
    else { cluster_of[v] = (int)members.size(); members.emplace_back(); tier_of_cluster.push_back(height[v] / band); }
    members[cluster_of[v]].push_back(v);
  }
  // clusters tier by tier, big ones first inside a tier
  std::vector<int32_t> order(members.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (tier_of_cluster[a] != tier_of_cluster[b]) return tier_of_cluster[a] < tier_of_cluster[b];
    return members[a].size() > members[b].size();
  });
  std::vector<int32_t> new_id(members.size());
  for (size_t i = 0; i < order.size(); ++i) new_id[order[i]] = (int)i;
  plan.tier_off.assign(n_tiers + 1, 0);
  for (size_t i = 0; i < order.size(); ++i) plan.tier_off[tier_of_cluster[order[i]] + 1]++;
  for (int t = 0; t < n_tiers; ++t) plan.tier_off[t + 1] += plan.tier_off[t];
  std::vector<int32_t> pos(Nn, -1);
  plan.item_off.push_back(0);
  plan.lvl_ptr.push_back(0);
  for (size_t ci = 0; ci < order.size(); ++ci) {
    std::vector<int32_t>& m = members[order[ci]];
    std::stable_sort(m.begin(), m.end(), [&](int a, int b) { return height[a] < height[b]; });
    const int base = (int)plan.nodes.size();
    for (size_t i = 0; i < m.size(); ++i) pos[m[i]] = (int)i;
    for (size_t i = 0; i < m.size(); ++i) {
      const UpStep& u = s.up[step_of[m[i]]];
      ClusterNode nd;
      nd.parent = u.parent; nd.pad = 0;
      for (int k = 0; k < 2; ++k) {
        nd.child[k] = u.child[k]; nd.edge[k] = u.edge[k];
        nd.slot[k] = (u.child[k] >= 0 && new_id[cluster_of[u.child[k]]] == (int)ci) ? pos[u.child[k]] : -1;
      }
      if (i == 0 || height[m[i]] != height[m[i - 1]]) plan.lvl_off.push_back(base + (int)i);
      plan.nodes.push_back(nd);
    }
    plan.lvl_off.push_back(base + (int)m.size());
    plan.item_off.push_back((int)plan.nodes.size());
    plan.lvl_ptr.push_back((int)plan.lvl_off.size());
  }
}

}  // namespace phm
