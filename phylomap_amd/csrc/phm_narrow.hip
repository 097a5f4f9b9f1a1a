// phm_narrow.hip -- branch-parallel MCMC sweep for few chains on a large tree (see phm_narrow.h).
// Same arithmetic as phm_mcmc.hip / the oracle: unfused left-to-right sums, the same categorical draw, the same Philox
// streams addressed by (replica, iteration, node | branch), the same deterministic log.
#include "phm_narrow.h"

#include <algorithm>

namespace phm {

namespace {

// level boundaries and other wave-uniform read-only words: through the scalar cache
typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;
__device__ __forceinline__ int32_t uniform_word(const int32_t* a, int i) { return ((const_i32_ptr)(uintptr_t)a)[i]; }

// LDS-only release / barrier / acquire: the waves wait for their LDS traffic, not for the global stores in flight
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NS>
__device__ __forceinline__ void narrow_stats_body(const NarrowParams<NS>& p, int it, int n_rows, int r, int c, double* red);

// element J of the quad's four lanes, to all four (DPP quad_perm: no LDS traffic)
template <int J>
__device__ __forceinline__ double quad_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, J * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, J * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// the value held four lanes away: lanes 0-3 and 4-7 of every group of eight swap (DPP row shifts by four, bank-masked)
__device__ __forceinline__ double swap_quads(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int tl = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xf, 0x5, false);       // row_shl:4 -> banks 0 and 2 read lane + 4
  int th = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
  tl = __builtin_amdgcn_update_dpp(tl, lo, 0x114, 0xf, 0xa, false);           // row_shr:4 -> banks 1 and 3 read lane - 4
  th = __builtin_amdgcn_update_dpp(th, hi, 0x114, 0xf, 0xa, false);
  return __hiloint2double(th, tl);
}

// Pruning sweep, PL[parent] = (B^(ma-1) PL[a]) (.) (B^(mb-1) PL[b])   (mmmmvFORpl src/phylomap.cpp:446-450, :508-510, :525).
// What bounds one chain on a big tree is the LATENCY of the longest root-to-tip line of dependent steps, so the sweep is laid
// out for latency:
//  * clusters (phm_sched.h ClusterPlan): one workgroup per subtree of <= NARROW_CLUSTER_NODES internal nodes walks ITS height
//    levels with the vectors of its nodes in LDS and an LDS-only barrier per level; all clusters of a tier in one launch, the
//    next tier (the subtrees of what is left above) in the next: 2 launches for 10 000 tips instead of one per height level;
//  * eight lanes per node: a quad per child, lane q of the quad holds component q of the child's vector and computes row q of
//    every chain step -- (((M_q0 x_0 + M_q1 x_1) + M_q2 x_2) + M_q3 x_3), the reference's order -- with the other three
//    components read across the quad by DPP: a chain step is 7 dependent f64 operations instead of 28 per lane;
//  * a tip child's chain is a table row (started from a one-hot row or, ks, a parity mask :1838-1845); the records, chain
//    lengths and tip data of the next level are requested before the barrier of this one.  Loads are unconditional (a load under
//    a lane condition is compiled to a branch with a full s_waitcnt behind it): a lane without an item reads a valid
//    neighbouring address and drops the value.
// The vectors also go to the global PL array (the transition maps of the sampling sweep and the clusters above read them);
// nobody in the kernel waits for those stores.
// the phases of the dependency-driven form below
enum : int { CL_NEED = 0, CL_WAIT = 1, CL_RUN = 2, CL_JOIN = 3, CL_EXIT = 4 };
#ifndef PHM_CLUSTER_STRIDE
#define PHM_CLUSTER_STRIDE 16
#endif
constexpr int NARROW_CLUSTER_STRIDE = PHM_CLUSTER_STRIDE;   // chain steps between two looks at the other phases

__device__ __forceinline__ int swap_quads_int(int x) {
  int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);
  return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);
}

// ASYNC: the same cluster without level barriers (long chains).  With level barriers a level costs its LONGEST chain, and the
// sweep the sum of those over the levels; on a tree whose branches hold hundreds of segments (the reference's squamate analysis:
// Omega = 10, ~111 segments per branch, 2 280 on the longest) that sum is 19 758 chain steps where the longest line of DEPENDENT
// steps is 2 280.  Here the nodes of the cluster (listed by height, a topological order) are handed out in that order to whichever
// group of eight lanes is free; a quad starts its child's chain as soon as that child's vector is there (a flag per node in LDS),
// and the two quads of a group join when both chains are done.  A wave holds eight groups in different phases, so nothing in the
// loop waits: every pass over the loop body gives each quad NARROW_CLUSTER_STRIDE chain steps or one look at a flag.  Progress:
// the first unfinished node of the order has been handed out before any later one and its children are finished.
// Every node is computed by the same instructions on the same operands as in the level form: same bits.
template <int NS, bool ASYNC>
__global__ __launch_bounds__(NARROW_CLUSTER_BLOCK) void narrow_cluster_kernel(NarrowParams<NS> p, int first_cluster, int n_clusters, int it, int stats_rows) {
  // Blocks beyond the tier's clusters (first tier of a sweep, stats_rows > 0): the statistics row of the PREVIOUS sweep, one
  // column each -- a launch of its own costs the sweep ~6 us of ramp-up for ~1 us of work, here it rides along for free.
  // (The previous sweep's per-wavefront rows and node states are untouched until this sweep's later kernels.)
  if ((int)blockIdx.x >= n_clusters) {
    __shared__ double red[NARROW_CLUSTER_BLOCK];
    narrow_stats_body<NS>(p, it - 1, stats_rows, blockIdx.y, (int)blockIdx.x - n_clusters, red);
    return;
  }
  constexpr int G = NARROW_CLUSTER_BLOCK / 8;        // nodes per pass
  constexpr int PASSES = NARROW_CLUSTER_NODES / G;
  __shared__ double s_pl[NARROW_CLUSTER_NODES * NS];           // vectors computed in this cluster
  __shared__ double s_v0[NARROW_CLUSTER_NODES * 2 * NS];       // start vectors that come from outside: table rows, clusters below
  __shared__ int32_t s_steps[NARROW_CLUSTER_NODES * 2];        // chain steps of (node, child): m - 1, 0 for a tip
  __shared__ int32_t s_slot[NARROW_CLUSTER_NODES * 2];         // child of the same cluster: its position; else -1
  __shared__ int32_t s_parent[NARROW_CLUSTER_NODES];
  __shared__ int32_t s_done[ASYNC ? NARROW_CLUSTER_NODES : 1];  // ASYNC: the node's vector is in s_pl
  __shared__ int32_t s_next;                                   // ASYNC: the next node to hand out
  const int cl = first_cluster + blockIdx.x;
  const int r = blockIdx.y;
  const int tid = threadIdx.x;
  const int g = tid >> 3, c = (tid >> 2) & 1, q = min(tid & 3, NS - 1);
  const bool lane_on = (tid & 3) < NS;
  uint32_t err = 0;
#ifdef PHM_DEBUG_LEVEL_CLOCK
  __shared__ unsigned long long s_clk[96];
  __shared__ int s_mx[96];
  int n_clk = 0;
#define PHM_CLK(X) do { if (tid == 0 && n_clk < 96) { s_mx[n_clk] = (X); s_clk[n_clk++] = wall_clock64(); } } while (0)
  PHM_CLK(0);
#else
#define PHM_CLK(X) do {} while (0)
#endif
  const int item0 = uniform_word(p.cl_item_off, cl);
  const int n_items = uniform_word(p.cl_item_off, cl + 1) - item0;
  const int lv0 = uniform_word(p.cl_lvl_ptr, cl), lv1 = uniform_word(p.cl_lvl_ptr, cl + 1) - 1;      // levels lv0 .. lv1 - 1
  const int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  double* __restrict__ PLr = p.PL + (size_t)r * p.n_node * NS;
  double mrow[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) mrow[j] = p.Bc[q * NS + j];

  // Everything the cluster reads from memory, once, all passes side by side (three dependent round trips in total): records,
  // then chain lengths and tip data, then the start vectors of the children that are not in the cluster.  The level loop below
  // touches LDS only.
  {
    int child[PASSES], slot[PASSES], edge[PASSES], parent[PASSES], k[PASSES], tip[PASSES];
    double v0[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const ClusterNode nd = p.cl_nodes[item0 + min(ps * G + g, n_items - 1)];
      parent[ps] = nd.parent;
      child[ps] = c ? nd.child[1] : nd.child[0];
      slot[ps] = c ? nd.slot[1] : nd.slot[0];
      edge[ps] = c ? nd.edge[1] : nd.edge[0];
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      k[ps] = mc[edge[ps]] - 1;
      tip[ps] = (int)tips[child[ps] < 0 ? ~child[ps] : 0];
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const bool tipc = child[ps] < 0;
      int kk = k[ps];
      if (tipc && kk >= p.klong) { if (ps * G + g < n_items) err |= DERR_CAPACITY; kk = p.klong - 1; }
      const double* src = tipc ? ((p.ks && p.tip_masks) ? p.maskL + ((size_t)kk * 2 + (tip[ps] & 1)) * NS + q
                                                        : p.colL + ((size_t)kk * NS + tip[ps]) * NS + q)
                               : PLr + (size_t)child[ps] * NS + q;
      v0[ps] = *src;                                 // a child of this cluster: a stale value, never used
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int local = ps * G + g;
      if (local < n_items) {
        if (lane_on) s_v0[(local * 2 + c) * NS + q] = v0[ps];
        if ((tid & 3) == 0) {
          s_steps[local * 2 + c] = child[ps] < 0 ? 0 : k[ps];
          s_slot[local * 2 + c] = slot[ps];
          if (c == 0) s_parent[local] = parent[ps];
        }
      }
    }
  }
  if (ASYNC) {
    for (int i = tid; i < NARROW_CLUSTER_NODES; i += NARROW_CLUSTER_BLOCK) s_done[i] = 0;
    if (tid == 0) s_next = 0;
  }
  lds_barrier();
  PHM_CLK(n_items);

  if (ASYNC) {
    int phase = CL_NEED, local = 0, slot = -1, steps = 0, i = 0;
    double v = 0.0;
    while (__any(phase != CL_EXIT)) {                // wave-uniform: the DPP exchanges below see every lane
      if (!__any(phase == CL_RUN)) __builtin_amdgcn_s_sleep(1);                // a wave that only waits leaves the SIMD to the one it shares it with
      const int sib = swap_quads_int(phase);
      if (phase == CL_NEED) {
        int nxt = 0;
        if ((tid & 7) == 0) nxt = atomicAdd(&s_next, 1);
        local = __shfl(nxt, (tid & 63) & ~7, 64);
        if (local >= n_items) phase = CL_EXIT;
        else { slot = s_slot[local * 2 + c]; steps = s_steps[local * 2 + c]; phase = CL_WAIT; }
      }
      if (phase == CL_WAIT) {
        const bool ready = slot < 0 || __hip_atomic_load(&s_done[max(slot, 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        if (ready) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          v = (slot >= 0) ? s_pl[slot * NS + q] : s_v0[(local * 2 + c) * NS + q];
          i = 0;
          phase = CL_RUN;
        }
      }
      if (phase == CL_RUN) {
        auto chain_step = [&]() {                    // uniform over the quad
          double acc = mrow[0] * quad_bcast<0>(v);
          if (NS > 1) acc += mrow[NS > 1 ? 1 : 0] * quad_bcast<1>(v);
          if (NS > 2) acc += mrow[NS > 2 ? 2 : 0] * quad_bcast<2>(v);
          if (NS > 3) acc += mrow[NS > 3 ? 3 : 0] * quad_bcast<3>(v);
          v = acc;
        };
        if (steps - i >= NARROW_CLUSTER_STRIDE) {    // the long chains: a stride without a look at the counter
#pragma unroll
          for (int u = 0; u < NARROW_CLUSTER_STRIDE; ++u) chain_step();
          i += NARROW_CLUSTER_STRIDE;
        } else {
          for (; i < steps; ++i) chain_step();
        }
        if (i >= steps) phase = CL_JOIN;
      } else if (phase == CL_JOIN && sib == CL_JOIN) {                         // both chains of the node were done a pass ago
        const double other = swap_quads(v);
        double x = c ? v * other : other * v;                                  // "first" (child[1]) times "second" (:510)
        if (p.normalise) {                                                     // :525
          double sum = quad_bcast<0>(x);
          if (NS > 1) sum += quad_bcast<1>(x);
          if (NS > 2) sum += quad_bcast<2>(x);
          if (NS > 3) sum += quad_bcast<3>(x);
          x = x / sum;
        }
        if (c == 0 && lane_on) {
          s_pl[local * NS + q] = x;
          PLr[(size_t)s_parent[local] * NS + q] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((tid & 7) == 0) __hip_atomic_store(&s_done[local], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        phase = CL_NEED;
      }
    }
    if (err) atomicOr(p.err, err);
    return;
  }

  int lo = uniform_word(p.cl_lvl_off, lv0) - item0;
  for (int lv = lv0; lv < lv1; ++lv) {
    const int hi = uniform_word(p.cl_lvl_off, lv + 1) - item0;
    for (int base = lo; base < hi; base += G) {
      const int local = min(base + g, hi - 1);
      const bool live = base + g < hi;
      const int slot = s_slot[local * 2 + c];
      const int steps = live ? s_steps[local * 2 + c] : 0;
      const double outside = s_v0[(local * 2 + c) * NS + q];
      const double inside = s_pl[max(slot, 0) * NS + q];
      double v = (slot >= 0) ? inside : outside;
      for (int i = 0; i < steps; ++i) {                                        // uniform over the quad
        double acc = mrow[0] * quad_bcast<0>(v);
        if (NS > 1) acc += mrow[NS > 1 ? 1 : 0] * quad_bcast<1>(v);
        if (NS > 2) acc += mrow[NS > 2 ? 2 : 0] * quad_bcast<2>(v);
        if (NS > 3) acc += mrow[NS > 3 ? 3 : 0] * quad_bcast<3>(v);
        v = acc;
      }
      const double other = swap_quads(v);                                      // the sibling quad's component q
      double x = c ? v * other : other * v;                                    // "first" (child[1]) times "second" (:510)
      if (p.normalise) {                                                       // :525
        double sum = quad_bcast<0>(x);
        if (NS > 1) sum += quad_bcast<1>(x);
        if (NS > 2) sum += quad_bcast<2>(x);
        if (NS > 3) sum += quad_bcast<3>(x);
        x = x / sum;
      }
      if (live && c == 0 && lane_on) {
        s_pl[local * NS + q] = x;
        PLr[(size_t)s_parent[local] * NS + q] = x;
      }
    }
    lds_barrier();
    PHM_CLK(hi - lo);
    lo = hi;
  }
  if (err) atomicOr(p.err, err);
#ifdef PHM_DEBUG_LEVEL_CLOCK
  if (tid == 0 && r == 0 && blockIdx.x == 0 && it == 30) {
    printf("clusterclock first %d items %d:", first_cluster, n_items);
    for (int i = 1; i < n_clk; ++i) printf(" %d(%d)", (int)(s_clk[i] - s_clk[i - 1]), s_mx[i]);
    printf("\n");
  }
#endif
#undef PHM_CLK
}

template <int NS>
__device__ __forceinline__ void root_node(const NarrowParams<NS>& p, int r, int it, uint32_t& err) {
  const double* PLr = p.PL + (size_t)r * p.n_node * NS;
  double pr[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) pr[c] = p.pid[c] * PLr[p.root * NS + c];        // :618
  const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it,
                            ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
  p.nstate[(size_t)r * p.n_node + p.root] = (uint8_t)sample_cat<NS>(pr, u, err);   // :627
}

// Sampling sweep: child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their parity mask
// (:1384-1397).  Everything in that draw except the parent's state ps is known once the pruning sweep is done -- PL[child], the
// chain length, the node's uniform -- and ps has NS possible values.  So the sweep is split:
//   narrow_downmap_kernel   every edge of every chain side by side, the whole device wide: the draw is carried out for EACH
//                           possible parent state; the NS outcomes (2 bits each + "probabilities were all zero") are one 16-bit
//                           word per edge, the edge's transition map;
//   narrow_downwalk_kernel  one workgroup per chain walks the levels root to tips: state[child] = map[edge][state[parent]],
//                           a table look-up per edge and an LDS barrier per level -- no arithmetic on the critical path; only
//                           the edges that lead to an internal node take part (nothing hangs below a tip);
//   narrow_branch_kernel    reads the end states of its edge (updatenodestates :460-475) off the edge's map and the parent's
//                           state, tip edges included.
// The outcome for the parent state that materialises is exactly the draw of the reference's sweep: same operands, same order.
template <int NS>
__global__ __launch_bounds__(NARROW_BLOCK) void narrow_downmap_kernel(NarrowParams<NS> p, int it) {
  const int idx = blockIdx.x * NARROW_BLOCK + threadIdx.x;
  const int r = blockIdx.y;
  if (idx >= p.n_edge) return;
  const DownStep ds = p.down_lv[idx];
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  const bool internal = ds.child >= 0;
  const int tip_state = (int)tips[internal ? 0 : ~ds.child];
  const bool draws = internal || (p.ks && p.tip_masks);
  int kk = p.mcount[(size_t)r * p.n_edge + ds.edge] - 1;
  uint32_t err = 0;
  if (kk >= p.klong) { if (draws) err |= DERR_CAPACITY; kk = p.klong - 1; }
  const double* __restrict__ PLc = p.PL + ((size_t)r * p.n_node + (internal ? ds.child : 0)) * NS;
  const double* __restrict__ rows = p.rowL + (size_t)(draws ? kk : 0) * NS * NS;
  const int par = tip_state & 1;
  double wgt[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    const double pl = PLc[c];
    wgt[c] = internal ? pl : (((c & 1) == par) ? 1.0 : 0.0);
  }
  const uint32_t node_id = internal ? (uint32_t)(ds.child + p.n_tips) : (uint32_t)~ds.child;
  const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it, ENT_NODE | node_id, 0);
  uint32_t code = 0;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    double w[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) w[c] = rows[q * NS + c] * wgt[c];
    uint32_t e2 = 0;
    const int drawn = sample_cat<NS>(w, u, e2);                                // :655
    const uint32_t out = draws ? ((uint32_t)drawn | (e2 ? 4u : 0u)) : (uint32_t)tip_state;   // :612
    code |= out << (4 * q);
  }
  p.dmap[(size_t)r * p.n_edge + idx] = (uint16_t)code;
  p.dmap_edge[(size_t)r * p.n_edge + ds.edge] = (uint16_t)code;
  if (err) atomicOr(p.err, err);
}

#ifndef PHM_NARROW_WALK_BLOCK
#define PHM_NARROW_WALK_BLOCK 1024
#endif
constexpr int NARROW_WALK_BLOCK = PHM_NARROW_WALK_BLOCK;
constexpr int NARROW_LDS_NODES = 60 * 1024;        // node states of one chain kept in LDS during the walk (1 byte each)

// Root draw, then the walk over the edges that lead to internal nodes.  The node states of the chain live in LDS (LDSN; trees
// of up to NARROW_LDS_NODES internal nodes) and go to the global array in one coalesced pass at the end; larger trees walk the
// global array behind workgroup-scope fences.  A level of the walk is far shorter than a memory round trip, so a lane requests
// its record of level l + 4 when it has used the one of level l (a ring of four) -- if its wave has one: near the root a level
// holds a handful of edges, and sixteen waves issuing loads nobody uses cost more than the level itself.
template <int NS, bool LDSN>
__global__ __launch_bounds__(NARROW_WALK_BLOCK) void narrow_downwalk_kernel(NarrowParams<NS> p, int it, int n_levels) {
  extern __shared__ uint8_t s_nst[];
  constexpr int RING = 4;
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  const int wave_base = tid & ~63;
  uint32_t err = 0;
  uint8_t* __restrict__ nst = p.nstate + (size_t)r * p.n_node;
  const uint16_t* __restrict__ dmap = p.dmap + (size_t)r * p.n_edge;
  auto level_lo = [&](int l) { return uniform_word(p.walk_off, min(l, n_levels)); };
  // level boundaries of the next eight levels in scalar registers, refilled four at a time one round ahead of their use
  int off[2 * RING + 1], off_next[RING];
#pragma unroll
  for (int k = 0; k <= 2 * RING; ++k) off[k] = level_lo(k);
  DownStep ring_ds[RING];
  uint32_t ring_code[RING];
  auto request = [&](int k, int lo_l, int hi_l) {
    if (wave_base < hi_l - lo_l) {
      const int at = min(lo_l + tid, p.n_edge - 1);
      ring_ds[k] = p.down_lv[at];
      ring_code[k] = dmap[at];
    }
  };
#pragma unroll
  for (int k = 0; k < RING; ++k) { ring_ds[k] = DownStep{0, 0, 0, 0}; ring_code[k] = 0u; request(k, off[k], off[k + 1]); }      // before the root draw
  if (tid == 0) {
    root_node<NS>(p, r, it, err);
    if (LDSN) s_nst[p.root] = nst[p.root];
  }
  if (LDSN) lds_barrier();
  else { __threadfence_block(); __syncthreads(); }
  for (int l0 = 0; l0 < n_levels; l0 += RING) {
#pragma unroll
    for (int k = 0; k < RING; ++k) off_next[k] = level_lo(l0 + 2 * RING + 1 + k);
#pragma unroll
    for (int k = 0; k < RING; ++k) {
      const int l = l0 + k;
      if (l < n_levels) {
        const int lo = off[k], hi = off[k + 1];
        DownStep ds = ring_ds[k];
        uint32_t code = ring_code[k];
        request(k, off[k + RING], off[k + RING + 1]);
        for (int idx = lo + tid; idx < hi; idx += NARROW_WALK_BLOCK) {
          if (idx != lo + tid) { ds = p.down_lv[idx]; code = dmap[idx]; }
          const int ps = LDSN ? s_nst[ds.parent] : nst[ds.parent];
          const uint32_t out = code >> (4 * ps);
          if (out & 4u) err |= DERR_ZERO_PROB;
          if (LDSN) s_nst[ds.child] = (uint8_t)(out & 3u);
          else nst[ds.child] = (uint8_t)(out & 3u);
        }
        if (LDSN) lds_barrier();
        else { __threadfence_block(); __syncthreads(); }
      }
    }
#pragma unroll
    for (int k = 0; k <= RING; ++k) off[k] = off[k + RING];
#pragma unroll
    for (int k = 0; k < RING; ++k) off[RING + 1 + k] = off_next[k];
  }
  if (LDSN) {                                        // the states of the sweep, for the branch kernel and the statistics
    for (int i = tid; i < p.n_node; i += NARROW_WALK_BLOCK) nst[i] = s_nst[i];
  }
  if (err) atomicOr(p.err, err);
}

// One branch of one chain: resamplebranchstates :264-308, shortener :44-73 (shortenerbf :997-1030), virtual jumps
// sampleabranch :391-410, dwell sums updatedwelltimes :745-757 -- EIGHT LANES PER BRANCH, eight branches per wave.
// The step is a short sequential state machine (previous state -> next state, running lengths, consumption of exponential gaps)
// fed by expensive values that do NOT depend on the state:
//   * a TRANSITION MAP per interior change point: the draw s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end (:290, :301-304) carried out
//     for each of the NS possible previous states (2 bits + "all-zero" flag each), beside the old segment's length;
//   * the standard exponential variates of the branch's stream (:398).
// Those are computed WAVE-WIDE first: the interior change points of the wave's eight branches are numbered consecutively and
// taken 64 at a time (two passes for a typical wave), likewise a first allotment of m + 8 variates per branch; then the eight
// lanes of a branch walk its state machine redundantly (same inputs, same results, one of them stores), reading maps, lengths
// and variates from LDS.  Merged segments (:54) are completed and cut by virtual jumps on the fly: no merged-segment scratch.
// What does not fit the wave's LDS windows (NARROW_MAP_WINDOW entries, NARROW_EXP_WINDOW variates: very long branches) and
// variates beyond the allotment are computed eight at a time by the branch's own lanes.
// branch_order lists the branches longest first; group g of wave k takes position g * n_waves + k: one of the longest branches
// and seven progressively shorter ones in every wave.  One chain on 20 000 branches has far more SIMDs than waves: the latency
// of the slowest wave is what the sweep waits for.
#ifndef PHM_NARROW_MAP_WINDOW
#define PHM_NARROW_MAP_WINDOW 192
#endif
#ifndef PHM_NARROW_EXP_WINDOW
#define PHM_NARROW_EXP_WINDOW 256
#endif
constexpr int NARROW_MAP_WINDOW = PHM_NARROW_MAP_WINDOW, NARROW_EXP_WINDOW = PHM_NARROW_EXP_WINDOW;

constexpr int NARROW_OUT_STAGE = 32;
template <int NS>
struct BranchLds {
  static constexpr int MAXG = 8;
  uint32_t cnt[NS * NS * MAXG];
  double wlen[NARROW_MAP_WINDOW];                    // wave-wide: lengths of the old segments 1 .. m - 1 of every branch,
  uint16_t wmap[NARROW_MAP_WINDOW];                  //            their transition maps,
  double we[NARROW_EXP_WINDOW];                      //            the first variates of every branch
  double xlen[NARROW_BLOCK];                         // per branch, L at a time: what lies beyond the windows
  uint32_t xmap[NARROW_BLOCK];
  double xe[NARROW_BLOCK];
  double out[MAXG * NARROW_OUT_STAGE];               // new segment lengths of a branch, written out 32 at a time by its lanes
  int32_t gm[MAXG], gcs[MAXG], gb[MAXG];
  long long goff[MAXG];
  alignas(16) double ltab[2 * PHM_LOGTAB_N];         // (1/c_j, log c_j) of the exponential variates (neglog_u32)
};

// L lanes per branch, 64 / L branches in the wave; the wave's group g takes sorted position first + g * stride
template <int NS, int L>
__device__ __forceinline__ void narrow_branch_body(const NarrowParams<NS>& p, int it, int first, int stride, BranchLds<NS>& sh) {
  constexpr int GROUPS = NARROW_BLOCK / L, OUTW = NARROW_OUT_STAGE;
  uint32_t* s_cnt = sh.cnt;
  double* s_wlen = sh.wlen; uint16_t* s_wmap = sh.wmap; double* s_we = sh.we;
  double* s_xlen = sh.xlen; uint32_t* s_xmap = sh.xmap; double* s_xe = sh.xe; double* s_out = sh.out;
  int32_t* s_gm = sh.gm; int32_t* s_gcs = sh.gcs; int32_t* s_gb = sh.gb; long long* s_goff = sh.goff;
  double* s_ltab = sh.ltab;
  const int lane = threadIdx.x;
  const int grp = lane / L, j = lane % L, gbase = grp * L;
  const int r = blockIdx.y;
#ifdef PHM_DEBUG_LEVEL_CLOCK
  unsigned long long tk[6];
  tk[0] = wall_clock64();
#define PHM_TK(K) tk[K] = wall_clock64()
#else
#define PHM_TK(K) do {} while (0)
#endif
  for (int i = lane; i < 2 * PHM_LOGTAB_N; i += NARROW_BLOCK) s_ltab[i] = logtab_entry(i);
  for (int i = lane; i < NS * NS * GROUPS; i += NARROW_BLOCK) s_cnt[i] = 0u;
  const int idx = first + grp * stride;
  const bool have = idx < p.n_edge;
  const bool KS = p.ks != 0;
  uint32_t err = 0;
  double acc[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;

  const int b = p.branch_order[have ? idx : 0];
  const uint32_t rep = (uint32_t)(p.replica_offset + r);
  int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  const int m = have ? mc[b] : 1;
  const int ps = p.nstate[(size_t)r * p.n_node + p.edge_parent[b]];            // updatenodestates :460-475: the edge's end states
  const uint32_t ends = (uint32_t)p.dmap_edge[(size_t)r * p.n_edge + b] >> (4 * ps);
  const int cs = (int)(ends & 3u);
  if (have && (ends & 4u)) err |= DERR_ZERO_PROB;
  const int64_t o = p.off[b];
  const int cap = (int)(p.off[b + 1] - o);
  const double* __restrict__ dw_in = p.dw[it & 1] + (size_t)r * p.total_cap;
  const double* __restrict__ in = dw_in + o;
  double* __restrict__ out = p.dw[(it & 1) ^ 1] + (size_t)r * p.total_cap + o;
  const uint32_t ent_s = ENT_BSTATE | (uint32_t)b, ent_e = ENT_BEXP | (uint32_t)b;
  if (j == 0) { s_gm[grp] = m; s_gcs[grp] = cs; s_gb[grp] = b; s_goff[grp] = (long long)o; }
  __syncthreads();
  PHM_TK(1);

  // windows: branch g owns map entries [wb[g], wb[g] + m_g - 1) and variates [eb[g], eb[g] + m_g + 8), cut at the window ends
  int wb[GROUPS + 1], eb[GROUPS + 1];
  wb[0] = 0; eb[0] = 0;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    const int mg = s_gm[g];
    wb[g + 1] = min(wb[g] + (mg - 1), NARROW_MAP_WINDOW);
    eb[g + 1] = min(eb[g] + (mg + 8), NARROW_EXP_WINDOW);
  }
  int my_wb = 0, my_eb = 0, pre = 0, epre = 0;       // this branch's window positions and sizes
#pragma unroll
  for (int g = 0; g < GROUPS; ++g)
    if (g == grp) { my_wb = wb[g]; pre = wb[g + 1] - wb[g]; my_eb = eb[g]; epre = eb[g + 1] - eb[g]; }
  for (int e0 = 0; e0 < wb[GROUPS]; e0 += NARROW_BLOCK) {      // transition maps, 64 change points at a time
    const int e = min(e0 + lane, wb[GROUPS] - 1);
    int g = 0;
#pragma unroll
    for (int k = 1; k < GROUPS; ++k) g += (e >= wb[k]) ? 1 : 0;
    int wbg = 0;
#pragma unroll
    for (int k = 1; k < GROUPS; ++k) wbg = (g == k) ? wb[k] : wbg;
    const int gm = s_gm[g], gcs = s_gcs[g];
    const int i = e - wbg + 1;                       // old segment 1 .. gm - 1 of branch g
    const double seglen = dw_in[s_goff[g] + i];
    int kk = gm - i - 1;
    if (kk >= p.klong) { if (i < gm - 1) err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double* beta = p.colL + ((size_t)kk * NS + gcs) * NS;
    double bv[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) bv[c] = beta[c];
    const double u = u01(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_BSTATE | (uint32_t)s_gb[g], (uint32_t)(i - 1)));
    uint32_t code = 0;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      double pr[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = p.B2[q * NS + c] * bv[c];
      uint32_t e2 = 0;
      const int sq = sample_cat<NS>(pr, u, e2);
      code |= ((uint32_t)sq | (e2 ? 4u : 0u)) << (4 * q);
    }
    if (i == gm - 1) code = (uint32_t)gcs * 0x1111u;                           // the last segment ends in the child's state
    s_wlen[e] = seglen; s_wmap[e] = (uint16_t)code;
  }
  PHM_TK(2);
  for (int e0 = 0; e0 < eb[GROUPS]; e0 += NARROW_BLOCK) {      // exponential variates, 64 at a time
    const int e = min(e0 + lane, eb[GROUPS] - 1);
    int g = 0;
#pragma unroll
    for (int k = 1; k < GROUPS; ++k) g += (e >= eb[k]) ? 1 : 0;
    int ebg = 0;
#pragma unroll
    for (int k = 1; k < GROUPS; ++k) ebg = (g == k) ? eb[k] : ebg;
    s_we[e] = neglog_u32(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_BEXP | (uint32_t)s_gb[g], (uint32_t)(e - ebg)), s_ltab);
  }
  __syncthreads();
  PHM_TK(3);

  int segs = 0;
  if (have) {
    int cur_s = (m == 1) ? cs : ps;                  // updatenodestates :469-472 (m == 1: the child end wins)
    double cur_len = in[0];
    int mnew = 0, flushed = 0;
    auto emit = [&](double piece) {                  // one more new segment (all eight lanes alike)
      if (mnew < cap) {
        s_out[grp * OUTW + (mnew - flushed)] = piece;
        if (mnew - flushed == OUTW - 1) {
#pragma unroll
          for (int t = 0; t < OUTW; t += L) if (L <= OUTW || t + j < OUTW) out[flushed + t + j] = s_out[grp * OUTW + t + j];
          flushed += OUTW;
        }
      } else {
        err |= DERR_CAPACITY;
      }
      ++mnew;
    };
    uint32_t edraw = 0, ehave = (uint32_t)epre;      // exponential variates consumed / available so far
    bool stuck = false;
    // the next window entries ride in registers: an iteration of the walk is a few dependent operations, an LDS read in the
    // middle of it would double it
    double pf_len = s_wlen[my_wb], pf_e = s_we[my_eb];
    uint32_t pf_map = s_wmap[my_wb];
    for (int i = 1; i <= m; ++i) {
      int si = -1;                                   // i == m: past the last old segment, the running one is complete
      double li = 0.0;
      if (i < m) {
        uint32_t map_i;
        if (i - 1 < pre) {
          li = pf_len; map_i = pf_map;
          const int nx = my_wb + min(i, pre - 1);
          pf_len = s_wlen[nx]; pf_map = s_wmap[nx];
        } else {
          const int past = i - 1 - pre;              // beyond the wave's window: eight old segments at a time, lane j takes i + j
          if ((past & (L - 1)) == 0) {
            const int ii = i + j;
            const int iic = min(ii, m - 1);
            const double len = in[iic];
            int kk = m - iic - 1;
            if (kk >= p.klong) { if (ii < m - 1) err |= DERR_CAPACITY; kk = p.klong - 1; }
            const double* beta = p.colL + ((size_t)kk * NS + cs) * NS;
            double bv[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) bv[c] = beta[c];
            const double u = u01(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ent_s, (uint32_t)max(iic - 1, 0)));
            uint32_t code = 0;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
              double pr[NS];
#pragma unroll
              for (int c = 0; c < NS; ++c) pr[c] = p.B2[q * NS + c] * bv[c];
              uint32_t e2 = 0;
              const int sq = sample_cat<NS>(pr, u, e2);
              code |= ((uint32_t)sq | (e2 ? 4u : 0u)) << (4 * q);
            }
            if (iic == m - 1) code = (uint32_t)cs * 0x1111u;
            s_xlen[lane] = len; s_xmap[lane] = code;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
          }
          li = s_xlen[gbase + (past & (L - 1))]; map_i = s_xmap[gbase + (past & (L - 1))];
        }
        const uint32_t out_i = map_i >> (4 * cur_s);
        si = (int)(out_i & 3u);
        if (out_i & 4u) err |= DERR_ZERO_PROB;
        if (KS && j == 0) atomicAdd(&s_cnt[(cur_s * NS + si) * GROUPS + grp], 1u);      // shortenerbf :1010-1014
        if (si == cur_s) { cur_len = cur_len + li; continue; }                  // :54
        if (!KS && j == 0) atomicAdd(&s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * GROUPS + grp], 1u);   // shortener :65-66
      }
      // the merged segment (cur_s, cur_len) is complete: virtual jumps, gaps ~ Exp(Omega + q_ss) until it is used up (:391-410);
      // a segment that is not positive leaves itself and everything after it untouched (the reference's iterators stop
      // advancing, :397, :405-406)
      {
        const int sg = cur_s;
        const double len = cur_len;
        double add = 0.0;
        if (stuck || !(0.0 < len)) {
          stuck = true;
          emit(len);
          add = len;
        } else {
          double scale = p.scale[0];
#pragma unroll
          for (int c = 1; c < NS; ++c) scale = (sg == c) ? p.scale[c] : scale;
          double tot = 0.0;
          while (tot < len) {
            double ev;
            if (edraw < (uint32_t)epre) {
              ev = pf_e;
              pf_e = s_we[my_eb + min((int)edraw + 1, epre - 1)];
            } else {
              if (edraw == ehave) {                  // beyond the allotment: the next eight variates of the stream, one per lane
                s_xe[lane] = neglog_u32(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ent_e, ehave + (uint32_t)j), s_ltab);
                ehave += L;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
              }
              ev = s_xe[gbase + (int)((edraw - (uint32_t)epre) & (L - 1))];
            }
            const double rl = scale * ev;                                      // :398
            ++edraw;
            double piece;
            if ((tot + rl) < len) { piece = rl; tot += rl; }
            else { piece = len - tot; tot = len; }
            emit(piece);
            add += piece;
          }
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) acc[c] += (sg == c) ? add : 0.0;          // updatedwelltimes :752
      }
      cur_s = si; cur_len = li;
    }
    if (mnew > cap) mnew = cap;
    for (int t = j; t < mnew - flushed; t += L) out[flushed + t] = s_out[grp * OUTW + t];
    if (j == 0) mc[b] = mnew;
    segs = m + mnew;
  }
  PHM_TK(4);
#ifdef PHM_DEBUG_LEVEL_CLOCK
  if (it == 30 && r == 0 && (blockIdx.x == 0 || blockIdx.x == 7 || blockIdx.x == 1200 || blockIdx.x == gridDim.x - 1) && j == 0)
    printf("branchclock block %d grp %d m %d: loads %d maps %d exps %d walk %d ticks (windows %d %d) start %llu\n", (int)blockIdx.x, grp, m,
           (int)(tk[1] - tk[0]), (int)(tk[2] - tk[1]), (int)(tk[3] - tk[2]), (int)(tk[4] - tk[3]), wb[GROUPS], eb[GROUPS], tk[0]);
#endif
#undef PHM_TK
  if (err) atomicOr(p.err, err);

  // one row per wavefront: the eight branches' sums, added in a fixed tree (every lane of a group holds the group's values)
  const int ncnt = KS ? NS * NS : NS * (NS - 1);
  constexpr int PC = NS + NS * NS + 1;
  double* part = p.part + ((size_t)r * gridDim.x + blockIdx.x) * PC;
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    double v = acc[c];
#pragma unroll
    for (int d = L; d < NARROW_BLOCK; d <<= 1) v = v + __shfl_xor(v, d);
    if (lane == 0) part[c] = v;
  }
#pragma unroll
  for (int d = L; d < NARROW_BLOCK; d <<= 1) segs += __shfl_xor(segs, d);
  if (lane == 0) part[PC - 1] = (double)segs;        // segments read + written (one global counter would serialise every lane)
  __syncthreads();
  if (lane < ncnt) {
    uint32_t v = 0;
#pragma unroll
    for (int g2 = 0; g2 < GROUPS; ++g2) v += s_cnt[lane * GROUPS + g2];
    part[NS + lane] = (double)v;
  }
}

// ONE branch, all 64 lanes of the wave (the branches that get a wave of their own): the walk of narrow_branch_body restated wave-wide.
// A branch of thousands of segments -- the reference's squamate analysis runs Omega = 10 on a tree whose longest branch holds 2 280
// of them -- spent a millisecond in the eight-lane state machine above, ~100 instructions per step of a lone wave.  Here, 64 old
// segments at a time:
//   * lane t computes the transition map of segment i0 + t (as above) and reads its length (one coalesced row);
//   * the states come from an inclusive SCAN of map compositions over the lanes (maps are functions on NS states: integers, exact),
//     entered with the state the previous block left;
//   * transitions are counted by ballots;
//   * what stays sequential is what the arithmetic specification makes sequential -- the left-to-right sums of a merged segment's
//     lengths and of the exponential gaps inside it (:54, :391-410) -- as loops over lane indices on wave-uniform values
//     (v_readlane + one addition per step); the variates are produced 64 at a time, one per lane, and the new pieces ARE those
//     lanes' values (all but the last piece of a merged segment): they leave as one store per merged segment and variate block.
// Same operations on the same operands in the same order as the eight-lane walk: same bits.
__device__ __forceinline__ double readlane_f64(double v, int t) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, t), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), t);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int NS>
__device__ __forceinline__ uint32_t compose_maps(uint32_t first, uint32_t then) {      // apply `first`, then `then`; bit 2 of an entry: "all-zero probabilities" met on the way
  uint32_t out = 0;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const uint32_t a = (first >> (4 * q)) & 7u;
    const uint32_t b = (then >> (4 * (a & 3u))) & 7u;
    out |= (b | (a & 4u)) << (4 * q);
  }
  return out;
}

template <int NS>
__device__ __forceinline__ void narrow_branch_wide(const NarrowParams<NS>& p, int it, int idx, BranchLds<NS>& sh) {
  double* s_ltab = sh.ltab;
  const int lane = threadIdx.x;
  const int r = blockIdx.y;
  for (int i = lane; i < 2 * PHM_LOGTAB_N; i += NARROW_BLOCK) s_ltab[i] = logtab_entry(i);
  const bool KS = p.ks != 0;
  uint32_t err = 0;
  const int b = p.branch_order[idx];
  const uint32_t rep = (uint32_t)(p.replica_offset + r);
  int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  const int m = mc[b];
  const int ps = p.nstate[(size_t)r * p.n_node + p.edge_parent[b]];            // updatenodestates :460-475: the edge's end states
  const uint32_t ends = (uint32_t)p.dmap_edge[(size_t)r * p.n_edge + b] >> (4 * ps);
  const int cs = (int)(ends & 3u);
  if (ends & 4u) err |= DERR_ZERO_PROB;
  const int64_t o = p.off[b];
  const int cap = (int)(p.off[b + 1] - o);
  const double* __restrict__ in = p.dw[it & 1] + (size_t)r * p.total_cap + o;
  double* __restrict__ out = p.dw[(it & 1) ^ 1] + (size_t)r * p.total_cap + o;
  const uint32_t ent_s = ENT_BSTATE | (uint32_t)b, ent_e = ENT_BEXP | (uint32_t)b;
  __syncthreads();

  double acc[NS];
  uint32_t cnt[NS * NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;
#pragma unroll
  for (int c = 0; c < NS * NS; ++c) cnt[c] = 0u;
  int cur_s = (m == 1) ? cs : ps;                    // updatenodestates :469-472 (m == 1: the child end wins)
  double cur_len = in[0];
  int mnew = 0;
  bool stuck = false;
  uint32_t edraw = 0, eblock = 0xFFFFFFFFu;          // variates consumed; the block of 64 the wave holds (lane k: variate 64 eblock + k)
  double evar = 0.0;

  // the merged segment (sg, len) is complete: virtual jumps, gaps ~ Exp(Omega + q_ss) until it is used up (:391-410); a segment
  // that is not positive leaves itself and everything after it untouched (the reference's iterators stop advancing, :397, :405-406).
  // The running sum of the gaps (tot, :399-404) is the only thing taken one gap at a time; the pieces are the gaps themselves,
  // each in its lane, except the last one (what is left of the segment), and leave as one store.  The dwell sum of the pieces
  // (:752, added one by one from +0) goes through the same additions as tot up to the last piece: it is tot + (len - tot).
  auto complete = [&](int sg, double len) {
    double add;
    if (stuck || !(0.0 < len)) {
      stuck = true;
      if (mnew < cap) { if (lane == 0) out[mnew] = len; } else err |= DERR_CAPACITY;
      ++mnew;
      add = len;
    } else {
      double scale = p.scale[0];
#pragma unroll
      for (int c = 1; c < NS; ++c) scale = (sg == c) ? p.scale[c] : scale;
      double tot = 0.0;
      for (;;) {
        if ((edraw >> 6) != eblock) {
          eblock = edraw >> 6;
          evar = neglog_u32(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ent_e, (eblock << 6) + (uint32_t)lane), s_ltab);
        }
        const double rlv = scale * evar;             // :398, lane k: gap k of the block at this state's rate
        const int k0 = (int)(edraw & 63u);
        int k = k0;
        bool stop = false;
        while (k < NARROW_BLOCK) {
          const double nt = tot + readlane_f64(rlv, k);
          ++k;
          if (!(nt < len)) { stop = true; break; }
          tot = nt;
        }
        const int nuse = k - k0, j = lane - k0;      // gaps k0 .. k - 1 were used: pieces mnew .. mnew + nuse - 1
        if (j >= 0 && j < nuse) {
          const double piece = (stop && j == nuse - 1) ? len - tot : rlv;
          if (mnew + j < cap) out[mnew + j] = piece; else err |= DERR_CAPACITY;
        }
        mnew += nuse;
        edraw += (uint32_t)nuse;
        if (stop) break;
      }
      add = tot + (len - tot);
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) acc[c] += (sg == c) ? add : 0.0;              // updatedwelltimes :752
  };

  for (int i0 = 1; i0 < m; i0 += NARROW_BLOCK) {     // old segments i0 .. i0 + 63, one per lane
    const int ii = i0 + lane;
    const bool valid = ii < m;
    const int iic = min(ii, m - 1);
    const double len = in[iic];
    int kk = m - iic - 1;
    if (kk >= p.klong) { if (ii < m - 1) err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double* beta = p.colL + ((size_t)kk * NS + cs) * NS;
    double bv[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) bv[c] = beta[c];
    const double u = u01(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ent_s, (uint32_t)max(iic - 1, 0)));
    uint32_t code = 0;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      double pr[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = p.B2[q * NS + c] * bv[c];
      uint32_t e2 = 0;
      const int sq = sample_cat<NS>(pr, u, e2);
      code |= ((uint32_t)sq | (e2 ? 4u : 0u)) << (4 * q);
    }
    if (iic == m - 1) code = (uint32_t)cs * 0x1111u;                            // the last segment ends in the child's state
    if (!valid) code = 0x3210u;                                               // past the branch: the identity (never looked at)
    uint32_t F = code;                               // inclusive scan: F_t = map_t o ... o map_0 (map_0 applied first)
#pragma unroll
    for (int off = 1; off < NARROW_BLOCK; off <<= 1) {
      const uint32_t G = (uint32_t)__shfl_up((int)F, off, NARROW_BLOCK);
      if (lane >= off) F = compose_maps<NS>(G, F);
    }
    const uint32_t mine = (F >> (4 * cur_s)) & 7u;   // the state after this lane's segment (entered with the wave's running state), and the flag
    const int my_s = (int)(mine & 3u);
    int prev_s = __shfl_up(my_s, 1, NARROW_BLOCK);
    if (lane == 0) prev_s = cur_s;
    if (valid && (mine & 4u)) err |= DERR_ZERO_PROB;
    // transitions: shortenerbf :1010-1014 counts every consecutive pair, shortener :65-66 the changes
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        if (!KS && a == c) continue;
        const unsigned long long hit = __ballot(valid && prev_s == a && my_s == c);
        const int col = KS ? a * NS + c : a * (NS - 1) + (c > a ? c - 1 : c);
        cnt[col] += (uint32_t)__popcll(hit);
      }
    // merged segments: lengths added left to right (:54), completed where the state changes
    const int nv = min(NARROW_BLOCK, m - i0);
    const unsigned long long chg = __ballot(valid && my_s != prev_s);           // bit t: segment i0 + t opens a new merged segment
    int t = 0;
    while (t < nv) {
      const unsigned long long rest = chg >> t;
      if (rest & 1ull) {
        complete(cur_s, cur_len);
        cur_s = __builtin_amdgcn_readlane(my_s, t); cur_len = readlane_f64(len, t);
        ++t;
        continue;
      }
      const int tn = rest ? min(nv, t + (int)__builtin_ctzll(rest)) : nv;
      for (; t < tn; ++t) cur_len = cur_len + readlane_f64(len, t);
    }
  }
  complete(cur_s, cur_len);                          // i == m: the running merged segment is complete
  if (mnew > cap) mnew = cap;
  if (lane == 0) mc[b] = mnew;
  if (err) atomicOr(p.err, err);

  // one row per wavefront, as narrow_branch_body writes it
  const int ncnt = KS ? NS * NS : NS * (NS - 1);
  constexpr int PC = NS + NS * NS + 1;
  double* part = p.part + ((size_t)r * gridDim.x + blockIdx.x) * PC;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NS; ++c) part[c] = acc[c];
#pragma unroll
    for (int c = 0; c < NS * NS; ++c) if (c < ncnt) part[NS + c] = (double)cnt[c];
    part[PC - 1] = (double)(m + mnew);               // segments read + written
  }
}

// The longest branches set the duration of the kernel: the first n_long (sorted order: NarrowParams::n_wide -- at least the 128
// longest, and every branch expected to hold 96 segments or more) get a wave each and are walked wave-wide (narrow_branch_wide),
// the others go eight to a wave: group g of wave k takes position n_long + g * n_waves8 + k, one of
// the longer branches and seven progressively shorter ones in every wave.
template <int NS>
__global__ __launch_bounds__(NARROW_BLOCK) void narrow_branch_kernel(NarrowParams<NS> p, int it, int n_long) {
  static_assert(NARROW_BLOCK == 64, "one wavefront per workgroup");
  __shared__ BranchLds<NS> sh;
  const int k = (int)blockIdx.x;
  if (k < n_long) narrow_branch_wide<NS>(p, it, k, sh);
  else narrow_branch_body<NS, 8>(p, it, n_long + (k - n_long), (int)gridDim.x - n_long, sh);
}

// Statistics row of one chain: every column is the sum of the per-wavefront rows of the branch kernel, added in a fixed order
// (thread t takes rows t, t+256, ... then a fixed tree over the 256 partial sums) -> identical from run to run.
// One workgroup per (chain, column): the columns of a row are reduced side by side (blockIdx.y), not one after the other.
// Without the reduction over replicas the value goes straight into the engine's statistics layout ([iter][cols][n_rep_pad]).
template <int NS>
__device__ __forceinline__ void narrow_stats_body(const NarrowParams<NS>& p, int it, int n_rows, int r, int c, double* red) {
  const int nt = (int)blockDim.x;                    // 256 or 512 threads
  const int ncnt = p.ks ? NS * NS : NS * (NS - 1);   // c = 0 .. NS + ncnt; the last one adds the segment counts (column pc - 1)
  const int pc = NS + NS * NS + 1;
  const double* part = p.part + (size_t)r * n_rows * pc;
  const int src_c = (c == NS + ncnt) ? pc - 1 : c;
  double s = 0.0;
  for (int e = threadIdx.x; e < n_rows; e += nt) s += part[(size_t)e * pc + src_c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int half = nt / 2; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + half];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* host = (p.host_row && r == 0 && !p.reduce) ? p.host_row : nullptr;
    if (c < NS + ncnt) {
      if (p.reduce) p.rowbuf[(size_t)r * p.n_cols + c] = red[0];
      else p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + r] = red[0];
      if (host) host[c] = red[0];
    } else {
      const unsigned long long before = atomicAdd(p.segcnt, (unsigned long long)red[0]);
      if (host) {                                    // the status words of the sweep (no kernel of the sweep is still running)
        host[p.n_cols] = (double)(before + (unsigned long long)red[0]);
        host[p.n_cols + 1] = (double)*p.err;
      }
    }
    if (p.ks && c == 0) {                                                      // root state, 0-based (:1350-1352)
      const double rs = (double)p.nstate[(size_t)r * p.n_node + p.root];
      if (p.reduce) p.rowbuf[(size_t)r * p.n_cols + NS + ncnt] = rs;
      else p.stats[((size_t)it * p.n_cols + NS + ncnt) * p.n_rep_pad + r] = rs;
      if (host) host[NS + ncnt] = rs;
    }
  }
}

template <int NS>
__global__ __launch_bounds__(256) void narrow_stats_kernel(NarrowParams<NS> p, int it, int n_rows) {
  __shared__ double red[256];
  narrow_stats_body<NS>(p, it, n_rows, blockIdx.x, blockIdx.y, red);
}

// rows of the chains of one 64-replica tile summed in replica order -> the engine's reduced statistics layout
template <int NS>
__global__ void narrow_emit_kernel(NarrowParams<NS> p, int it) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= p.n_tiles * p.n_cols) return;
  const int tile = gid / p.n_cols, c = gid % p.n_cols;
  double s = 0.0;
  for (int r = tile * 64; r < tile * 64 + 64 && r < p.n_rep; ++r) s += p.rowbuf[(size_t)r * p.n_cols + c];
  p.stats[((size_t)it * p.n_tiles + tile) * p.n_cols + c] = s;
}

}  // namespace

template <int NS>
static unsigned narrow_branch_waves(const NarrowParams<NS>& p, int& n_long) {
  n_long = std::max(0, std::min(p.n_wide, p.n_edge));
  return (unsigned)(n_long + (p.n_edge - n_long + 7) / 8);
}

template <int NS>
hipError_t launch_narrow_sweep(const NarrowParams<NS>& p, const std::vector<int32_t>& tier_off,
                               const std::vector<int32_t>& walk_off, int it, hipStream_t stream, bool stats_pending) {
  const unsigned S = (unsigned)p.n_rep;
  int n_long = 0;
  const unsigned n_waves = narrow_branch_waves<NS>(p, n_long);
  const unsigned stat_cols = (unsigned)(NS + (p.ks ? NS * NS : NS * (NS - 1)) + 1);
  // pruning sweep: one launch per tier of clusters (the first one also carries the previous sweep's statistics, if they are
  // still to be added up); then the sampling sweep: transition maps of all edges, root draw + walk; branch paths
  const int DL = (int)walk_off.size() - 1;
  for (size_t t = 0; t + 1 < tier_off.size(); ++t) {
    const unsigned ncl = (unsigned)(tier_off[t + 1] - tier_off[t]);
    const bool carry = t == 0 && stats_pending && !p.reduce;   // reduce: the sweep added its own row below (a second pass would count its segments twice)
    if (p.cluster_async)
      hipLaunchKernelGGL((narrow_cluster_kernel<NS, true>), dim3(ncl + (carry ? stat_cols : 0u), S), dim3(NARROW_CLUSTER_BLOCK), 0, stream, p,
                         tier_off[t], (int)ncl, it, carry ? (int)n_waves : 0);
    else
      hipLaunchKernelGGL((narrow_cluster_kernel<NS, false>), dim3(ncl + (carry ? stat_cols : 0u), S), dim3(NARROW_CLUSTER_BLOCK), 0, stream, p,
                         tier_off[t], (int)ncl, it, carry ? (int)n_waves : 0);
  }
  hipLaunchKernelGGL(narrow_downmap_kernel<NS>, dim3((p.n_edge + NARROW_BLOCK - 1) / NARROW_BLOCK, S), dim3(NARROW_BLOCK), 0, stream,
                     p, it);
  if (p.n_node <= NARROW_LDS_NODES)
    hipLaunchKernelGGL((narrow_downwalk_kernel<NS, true>), dim3(S), dim3(NARROW_WALK_BLOCK), (size_t)((p.n_node + 15) & ~15), stream, p,
                       it, DL);
  else
    hipLaunchKernelGGL((narrow_downwalk_kernel<NS, false>), dim3(S), dim3(NARROW_WALK_BLOCK), 0, stream, p, it, DL);
  hipLaunchKernelGGL(narrow_branch_kernel<NS>, dim3(n_waves, S), dim3(NARROW_BLOCK), 0, stream, p, it, n_long);
  if (p.reduce) {                                    // summed over replicas: the row is needed by the tile sums right away
    hipLaunchKernelGGL(narrow_stats_kernel<NS>, dim3(S, stat_cols), dim3(256), 0, stream, p, it, (int)n_waves);
    const int items = p.n_tiles * p.n_cols;
    hipLaunchKernelGGL(narrow_emit_kernel<NS>, dim3((items + 255) / 256), dim3(256), 0, stream, p, it);
  }
  return hipGetLastError();
}

// statistics of sweep `it` when no further sweep follows in this call (launch_narrow_sweep leaves them to the next sweep's first launch)
template <int NS>
hipError_t launch_narrow_stats(const NarrowParams<NS>& p, int it, hipStream_t stream) {
  if (p.reduce) return hipSuccess;
  int n_long = 0;
  const unsigned n_waves = narrow_branch_waves<NS>(p, n_long);
  hipLaunchKernelGGL(narrow_stats_kernel<NS>, dim3((unsigned)p.n_rep, (unsigned)(NS + (p.ks ? NS * NS : NS * (NS - 1)) + 1)), dim3(256), 0,
                     stream, p, it, (int)n_waves);
  return hipGetLastError();
}

template hipError_t launch_narrow_sweep<2>(const NarrowParams<2>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, bool);
template hipError_t launch_narrow_stats<2>(const NarrowParams<2>&, int, hipStream_t);
template hipError_t launch_narrow_sweep<3>(const NarrowParams<3>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, bool);
template hipError_t launch_narrow_stats<3>(const NarrowParams<3>&, int, hipStream_t);
template hipError_t launch_narrow_sweep<4>(const NarrowParams<4>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, bool);
template hipError_t launch_narrow_stats<4>(const NarrowParams<4>&, int, hipStream_t);

}  // namespace phm
