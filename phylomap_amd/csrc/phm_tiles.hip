// phm_tiles.hip -- one wavefront per (tile of 64 replicas, branch): the sweep for 10^2 .. 10^5 replicas (see phm_tiles.h).
// The per-branch code is the two-flat-pass scheme of phm_mcmc.hip (same arithmetic, same draws), with the branch's rows
// taken from its own slot instead of the tile's sequential stream and the chain powers read from the long tables.
#include "phm_tiles.h"

namespace phm {

namespace {

// B^k applied to a child's partial-likelihood vector (mmmmvFORpl, src/phylomap.cpp:446-450)
template <int NS>
__device__ __forceinline__ void child_vec(const TileParams<NS>& p, const double* __restrict__ PLt,
                                          const uint8_t* __restrict__ tips_t, int child, int k, int lane, double (&v)[NS],
                                          uint32_t& err) {
  if (child < 0) {
    const int tip = ~child;
    const int st = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];
    if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
    const double* src = (p.ks && p.tip_masks) ? p.maskL + ((size_t)k * 2 + (st & 1)) * NS : p.colL + ((size_t)k * NS + st) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) v[c] = src[c];
  } else {
#pragma unroll
    for (int c = 0; c < NS; ++c) v[c] = PLt[(child * NS + c) * 64 + lane];
    for (int i = 0; i < k; ++i) matvec_u<NS>(p.Bc, v);
  }
}

// PL[parent] = (B^(m1-1) PL[c1]) (.) (B^(m0-1) PL[c0]) (/ sum)  for the 64 replicas of a tile   (:503-529)
template <int NS>
__device__ __forceinline__ void up_node(const TileParams<NS>& p, int tile, int parent, int child0, int child1, int edge0, int edge1, int lane,
                                        uint32_t& err) {
  double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * NS * 64;
  const uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  double x[NS], y[NS];
  child_vec<NS>(p, PLt, tips_t, child1, (int)mct[edge1 * 64 + lane] - 1, lane, x, err);   // "first"  (:508)
  child_vec<NS>(p, PLt, tips_t, child0, (int)mct[edge0 * 64 + lane] - 1, lane, y, err);   // "second" (:509)
#pragma unroll
  for (int c = 0; c < NS; ++c) x[c] = x[c] * y[c];                             // :510
  if (p.normalise) {                                                           // :525
    double s = x[0];
#pragma unroll
    for (int c = 1; c < NS; ++c) s += x[c];
#pragma unroll
    for (int c = 0; c < NS; ++c) x[c] = x[c] / s;
  }
#pragma unroll
  for (int c = 0; c < NS; ++c) PLt[(parent * NS + c) * 64 + lane] = x[c];
}

template <int NS>
__global__ __launch_bounds__(TILES_BLOCK) void tiles_up_kernel(TileParams<NS> p, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (TILES_BLOCK / 64) + (threadIdx.x >> 6);
  const int n_lvl = end - begin;
  if (item >= n_lvl * p.n_tiles) return;
  const int tile = item / n_lvl;
  const UpStep st = p.up[p.up_order[begin + item % n_lvl]];
  uint32_t err = 0;
  up_node<NS>(p, tile, st.parent, st.child[0], st.child[1], st.edge[0], st.edge[1], lane, err);
  if (err) atomicOr(p.err, err);
}

// The same pass over one TIER of subtree clusters: a workgroup per (cluster, tile) walks the cluster's height levels, its waves
// sharing the nodes of a level, a workgroup barrier between levels (the waves of a workgroup share their CU's L1: what one wrote
// before the barrier the others read after it).  Big clusters first.
// ASYNC (long paths: TileParams::mstate is set): no barriers -- a level barrier charges the level its longest chain, and on paths of
// hundreds of segments the sum of those is many times the longest line of dependent steps (phm_narrow.hip, narrow_cluster_kernel).
// The waves take the cluster's nodes (listed by height: a topological order) off a counter as they get free and wait for the
// children that belong to the cluster on a flag per node; the first unfinished node was handed out before any later one and its
// children are finished, so some wave always runs.  A cluster of a band of eight levels holds at most 255 nodes.
constexpr int TILES_CL_ASYNC_NODES = 256;
template <int NS, bool ASYNC>
__global__ __launch_bounds__(TILES_CL_BLOCK) void tiles_up_cluster_kernel(TileParams<NS> p, int cl_begin, int n_cl) {
  constexpr int W = TILES_CL_BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = cl_begin + (int)(blockIdx.x % (unsigned)n_cl);
  const int tile = (int)(blockIdx.x / (unsigned)n_cl);
  const int l0 = p.cl_lvl_ptr[cl], l1 = p.cl_lvl_ptr[cl + 1] - 1;
  uint32_t err = 0;
  if (ASYNC) {
    __shared__ int32_t s_done[TILES_CL_ASYNC_NODES];
    __shared__ int32_t s_next;
    const int first = p.cl_lvl_off[l0], n_items = p.cl_lvl_off[l1] - first;
    if (n_items <= TILES_CL_ASYNC_NODES) {             // (a bigger cluster -- not from the band plan -- walks its levels below)
      for (int i = threadIdx.x; i < TILES_CL_ASYNC_NODES; i += TILES_CL_BLOCK) s_done[i] = 0;
      if (threadIdx.x == 0) s_next = 0;
      __syncthreads();
      for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(&s_next, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= n_items) break;
        const ClusterNode nd = p.cl_nodes[first + idx];
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (nd.slot[k] >= 0)
            while (__hip_atomic_load(&s_done[nd.slot[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        up_node<NS>(p, tile, nd.parent, nd.child[0], nd.child[1], nd.edge[0], nd.edge[1], lane, err);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_done[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (err) atomicOr(p.err, err);
      return;
    }
  }
  for (int l = l0; l < l1; ++l) {
    const int i1 = p.cl_lvl_off[l + 1];
    for (int i = p.cl_lvl_off[l] + wave; i < i1; i += W) {
      const ClusterNode nd = p.cl_nodes[i];
      up_node<NS>(p, tile, nd.parent, nd.child[0], nd.child[1], nd.edge[0], nd.edge[1], lane, err);
    }
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// root ~ pid (.) PL[root]   (:618-627)
template <int NS>
__device__ __forceinline__ void draw_root(const TileParams<NS>& p, int tile, int it, int lane, uint32_t& err) {
  const double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * NS * 64;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  double pr[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) pr[c] = p.pid[c] * PLt[(p.root * NS + c) * 64 + lane];   // :618
  const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
  p.nstate[((size_t)tile * p.n_node + p.root) * 64 + lane] = (uint8_t)sample_cat<NS>(pr, u, err);   // :627
}

template <int NS>
__global__ __launch_bounds__(TILES_BLOCK) void tiles_root_kernel(TileParams<NS> p, int it) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (TILES_BLOCK / 64) + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;
  uint32_t err = 0;
  draw_root<NS>(p, tile, it, lane, err);
  if (err) atomicOr(p.err, err);
}

// child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their parity mask (:1384-1397);
// end states of the edge (updatenodestates :460-475)
template <int NS>
__device__ __forceinline__ void down_edge(const TileParams<NS>& p, int tile, int it, int b, int parent, int child, int lane, uint32_t& err) {
  const double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * NS * 64;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  const int m = p.mcount[((size_t)tile * p.n_edge + b) * 64 + lane];
  const int ps = nst[parent * 64 + lane];
  int cs;
  if (child >= 0 || (p.ks && p.tip_masks)) {
    int kk = m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double* src = p.rowL + ((size_t)kk * NS + ps) * NS;
    double w[NS];
    uint32_t node_id;
    if (child >= 0) {
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * PLt[(child * NS + c) * 64 + lane];
      node_id = (uint32_t)(child + p.n_tips);
    } else {
      const int tip = ~child;
      const int par = (p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip]) & 1;
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * (((c & 1) == par) ? 1.0 : 0.0);
      node_id = (uint32_t)tip;
    }
    const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | node_id, 0);
    cs = sample_cat<NS>(w, u, err);                                            // :655
    if (child >= 0) nst[child * 64 + lane] = (uint8_t)cs;
  } else {
    const int tip = ~child;
    cs = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];          // :612
  }
  p.estate[((size_t)tile * p.n_edge + b) * 64 + lane] = (uint8_t)(ps | (cs << 4));   // updatenodestates :460-475
}

// what a node draw reads from memory that does not depend on anything it computes: requested one item ahead by tiles_down_kernel
template <int NS>
struct DownLoads {
  int32_t tile, b, parent, child;
  int32_t m, ps, tipst;
  double pl[NS];
};

template <int NS>
__device__ __forceinline__ DownLoads<NS> down_request(const TileParams<NS>& p, int item, int n_lvl, int begin, int lane) {
  DownLoads<NS> d;
  const DownStep ds = p.down[p.down_order[begin + item % n_lvl]];
  d.tile = item / n_lvl; d.b = ds.edge; d.parent = ds.parent; d.child = ds.child;
  d.m = p.mcount[((size_t)d.tile * p.n_edge + d.b) * 64 + lane];
  d.ps = p.nstate[((size_t)d.tile * p.n_node + d.parent) * 64 + lane];
  d.tipst = 0;
  if (d.child >= 0) {
    const double* __restrict__ PLc = p.PL + ((size_t)d.tile * p.n_node + d.child) * NS * 64 + lane;
#pragma unroll
    for (int c = 0; c < NS; ++c) d.pl[c] = PLc[c * 64];
  } else {
    const int tip = ~d.child;
#pragma unroll
    for (int c = 0; c < NS; ++c) d.pl[c] = 0.0;
    d.tipst = p.tips_per_replica ? p.tips[((size_t)d.tile * p.n_tips + tip) * 64 + lane] : p.tips[tip];
  }
  return d;
}

// the draw itself on what down_request brought: the same expression as down_edge
template <int NS>
__device__ __forceinline__ void down_finish(const TileParams<NS>& p, const DownLoads<NS>& d, int it, int lane, uint32_t& err) {
  const uint32_t rep = (uint32_t)(p.replica_offset + d.tile * 64 + lane);
  int cs;
  if (d.child >= 0 || (p.ks && p.tip_masks)) {
    int kk = d.m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double* src = p.rowL + ((size_t)kk * NS + d.ps) * NS;
    double w[NS];
    uint32_t node_id;
    if (d.child >= 0) {
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * d.pl[c];
      node_id = (uint32_t)(d.child + p.n_tips);
    } else {
      const int par = d.tipst & 1;
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * (((c & 1) == par) ? 1.0 : 0.0);
      node_id = (uint32_t)(~d.child);
    }
    const double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | node_id, 0);
    cs = sample_cat<NS>(w, u, err);                                            // :655
    if (d.child >= 0) p.nstate[((size_t)d.tile * p.n_node + d.child) * 64 + lane] = (uint8_t)cs;
  } else {
    cs = d.tipst;                                                             // :612
  }
  p.estate[((size_t)d.tile * p.n_edge + d.b) * 64 + lane] = (uint8_t)(d.ps | (cs << 4));   // updatenodestates :460-475
}

template <int NS>
__global__ __launch_bounds__(TILES_BLOCK) void tiles_down_kernel(TileParams<NS> p, int it, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int n_lvl = end - begin;
  const int n_items = n_lvl * p.n_tiles;
  const int stride = gridDim.x * (TILES_BLOCK / 64);
  uint32_t err = 0;
  // Persistent waves (kernel arguments and wave set-up once per wave: 2.28 -> 2.01 ms per sweep on C3; the pruning kernel above loses by
  // the same change: 3.16 -> 3.7), and -- round 4 -- the NEXT item's segment count, parent state and partial likelihoods are requested
  // before the current item is drawn: an item is three dependent memory round trips and a dozen operations, the kernel waited 0.84 of
  // its time (profiles/r04_pmc_C3_summary.json).  A level's items only read states drawn by the previous launch, so running ahead is safe.
  int item = blockIdx.x * (TILES_BLOCK / 64) + (threadIdx.x >> 6);
  if (item < n_items) {
    DownLoads<NS> cur = down_request<NS>(p, item, n_lvl, begin, lane);
    for (;;) {
      const int next = item + stride;
      const bool more = next < n_items;                // wave-uniform
      DownLoads<NS> nxt = cur;
      if (more) nxt = down_request<NS>(p, next, n_lvl, begin, lane);
      down_finish<NS>(p, cur, it, lane, err);
      if (!more) break;
      cur = nxt; item = next;
    }
  }
  if (err) atomicOr(p.err, err);
}

// The node draws over one tier of subtree clusters, top level first: a wave per (node of the level, child side) draws the child's
// state and writes the edge's end states; the tier that holds the root draws it first (with_root).
template <int NS>
__global__ __launch_bounds__(TILES_CL_BLOCK) void tiles_down_cluster_kernel(TileParams<NS> p, int it, int cl_begin, int n_cl, int with_root) {
  constexpr int W = TILES_CL_BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = cl_begin + (int)(blockIdx.x % (unsigned)n_cl);
  const int tile = (int)(blockIdx.x / (unsigned)n_cl);
  const int l0 = p.cl_lvl_ptr[cl], l1 = p.cl_lvl_ptr[cl + 1] - 1;
  uint32_t err = 0;
  if (with_root) {                                   // the last tier is one cluster and its top node is the root
    if (wave == 0) draw_root<NS>(p, tile, it, lane, err);
    __syncthreads();
  }
  for (int l = l1 - 1; l >= l0; --l) {
    const int i0 = p.cl_lvl_off[l], cnt = 2 * (p.cl_lvl_off[l + 1] - i0);
    for (int j = wave; j < cnt; j += W) {
      const ClusterNode nd = p.cl_nodes[i0 + (j >> 1)];
      down_edge<NS>(p, tile, it, nd.edge[j & 1], nd.parent, nd.child[j & 1], lane, err);
    }
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// One branch for the 64 replicas of a tile: resamplebranchstates :264-308, shortener :44-73 (shortenerbf :997-1030),
// virtual jumps sampleabranch :391-410, dwell sums updatedwelltimes :745-757.
// LONG (paths of more than 64 segments expected on some branch -- the reference's squamate run holds 2 280 on one): the two passes
// below without their limit of 64 merged segments per branch and lane.  The states of the merged segments, two bits apiece in
// two registers otherwise, go to a byte per (row, lane) beside the dwell rows (TileParams::mstate: 64-byte coalesced rows, written
// by pass A, read by pass B a step ahead), so the passes run to any length at the cost of a byte store and load per merged
// segment; the lane-sequential general loop (~20 us per segment of ONE wave: 46-51 ms per sweep on that tree whatever the
// chain count) is not compiled into this form.  On long paths the lanes of a wave also agree better: the longest of 64
// Poisson(1 000) counts is 1.1x their mean where the longest of 64 Poisson(4) counts is 2.4x.
template <int NS, bool KS, bool LONG>
__global__ __launch_bounds__(TILES_BLOCK, LONG ? 5 : (KS ? 7 : 8)) void tiles_branch_kernel(TileParams<NS> p, int it) {      // KS: NS*NS counters per lane -> 7 waves per SIMD; LONG: the prefetch ring takes registers
  constexpr int NCNT = KS ? NS * NS : NS * (NS - 1);
  __shared__ double s_dw_all[(TILES_BLOCK / 64) * NS * 64];
  __shared__ uint16_t s_cnt_all[(TILES_BLOCK / 64) * NCNT * 64];      // counts of ONE branch (<= 65 535 segments): 16 bits keep the block under 20 KB of LDS, 8 waves per SIMD
  __shared__ double s_B2[NS * NS], s_scale[NS];      // indexed by a per-lane state: LDS, not the kernarg segment
  __shared__ double s_col[TILES_KTAB * NS * NS];     // B^k e_j for the short chains (most draws); longer ones go to L2
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];        // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: slot bases live in scalar registers
  const int item = blockIdx.x * (TILES_BLOCK / 64) + wave;
  const uint32_t lane8 = (uint32_t)lane * 8u;
  const int ktab = min(TILES_KTAB, p.klong);
  for (int i = threadIdx.x; i < ktab * NS * NS; i += TILES_BLOCK) s_col[i] = p.colL[i];
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += TILES_BLOCK) s_ltab[i] = logtab_entry(i);
  if (threadIdx.x < NS * NS) s_B2[threadIdx.x] = p.B2[threadIdx.x];
  if (threadIdx.x < NS) s_scale[threadIdx.x] = p.scale[threadIdx.x];
  __syncthreads();
  if (item >= p.n_groups * p.n_tiles) return;        // whole waves only; no barrier below this line
  // neighbouring waves take the same branches of different tiles: similar run times inside a workgroup, longest branches
  // first.  A wave walks `group` consecutive entries of the order (1 when replicas are few, up to 16 when there are plenty of
  // waves anyway) and writes ONE partial dwell sum for all of them.
  const int tile = item % p.n_tiles;
  const int grp = item / p.n_tiles;
  double* s_dw = s_dw_all + wave * NS * 64;
  uint16_t* s_cnt = s_cnt_all + wave * NCNT * 64;
  const uint32_t rep = (uint32_t)(p.replica_offset + tile * 64 + lane);
  uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  uint32_t* gc = p.cnt + (((size_t)tile * p.cnt_copies + (grp & (p.cnt_copies - 1))) * NS * NS) * 64 + lane;
  uint32_t err = 0;
#pragma unroll
  for (int c = 0; c < NS; ++c) s_dw[c * 64 + lane] = 0.0;
#pragma unroll
  for (int c = 0; c < NCNT; ++c) s_cnt[c * 64 + lane] = (uint16_t)0;
  // The 16-bit LDS counters are handed to the tile's global counters (integer atomics, exact in any order) once per group of
  // branches -- every atomic instruction leaves L2 as uncached 64-byte requests, which at one flush per branch was a fifth of
  // the kernel's HBM traffic (profiles/r02_pmc_C3_summary.json) -- or earlier, before a lane could exceed 65 535 transitions.
  auto flush_counts = [&]() {
#pragma unroll
    for (int c = 0; c < NCNT; ++c) {
      const uint32_t v = s_cnt[c * 64 + lane];
      if (v) { atomicAdd(gc + c * 64, v); s_cnt[c * 64 + lane] = (uint16_t)0; }
    }
  };
  uint32_t pending = 0;                              // segments whose transitions sit in the 16-bit counters
  const int q1 = min((grp + 1) * p.group, p.n_edge);
  for (int q = grp * p.group; q < q1; ++q) {
  const int b = p.branch_order[q];
  const int m = mct[b * 64 + lane];
  if (__any(pending + (uint32_t)m > 65535u)) { flush_counts(); pending = 0; }
  pending += (uint32_t)m;
  const int es = p.estate[((size_t)tile * p.n_edge + b) * 64 + lane];
  const int ps = es & 15, cs = es >> 4;
  const int roff = p.slot[b];
  const int cap = p.slot[b + 1] - roff;
  // the slot of this (tile, branch): scalar base + 32-bit byte offset (a slot holds at most 65 535 rows of 512 B)
  double* __restrict__ in = p.dw[it & 1] + ((size_t)tile * p.rows + roff) * 64;
  double* __restrict__ out = p.dw[(it & 1) ^ 1] + ((size_t)tile * p.rows + roff) * 64;
  auto IN = [&](int k) -> double& { return at(in, (uint32_t)k * 512u + lane8); };
  uint8_t* __restrict__ msrow = LONG ? p.mstate + ((size_t)tile * p.rows + roff) * 64 + lane : nullptr;
  auto MS = [&](int k) -> uint8_t& { return msrow[(uint32_t)k * 64u]; };

  Stream su, se;
  su.open(ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
  se.open(ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
  const int mmax = wave_max_count(m);
  int mnew = 0;

  // s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end          (resamplebranchstates :290, :301-304)
  auto draw_state_w = [&](int i, int sprev, uint32_t word) -> int {
    int kk = m - i - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    double pr[NS];
    {                                          // always an LDS read (ds_read); the rare long chain overwrites it from global
      const double* beta = s_col + ((kk < ktab ? kk : ktab - 1) * NS + cs) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = beta[c];
    }
    if (kk >= ktab) {
      const double* __restrict__ beta = p.colL + ((size_t)kk * NS + cs) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = beta[c];
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) pr[c] = s_B2[sprev * NS + c] * pr[c];
    return sample_cat<NS>(pr, u01(word), err);
  };
  auto draw_state = [&](int i, int sprev) -> int { return draw_state_w(i, sprev, su.draw_word((uint32_t)(i - 1))); };

  if (LONG || mmax <= 64) {
    // Pass A: one old segment per step for every lane; merged segments written back in place over the consumed rows of the
    // slot, their states packed 2 bits apiece into two registers.
    uint64_t pk0 = 0, pk1 = 0;
    int w = 0;
    int cur_s = (m == 1) ? cs : ps;            // updatenodestates :469-472 (m==1: child wins)
    // The first two and the last merged segment stay in registers (first_len, second_len, cur_len); only the ones between them
    // pass through the slot's rows -- with at most three merged segments on a branch (the common case) pass B reads nothing back
    // (they all went through the rows: 17.2 -> 15.9 ms with two of them in registers, profiles/r03_probe_branch_ablation.log).
    double first_len = 0.0, second_len = 0.0;
    double cur_len = IN(0);
    double dnext = (!LONG && m > 1) ? IN(1) : 0.0;
    // LONG: what a step reads from memory -- the old segment's length and, beyond the rows of the chain table kept in LDS, the row
    // B^(m-i-1) e_end -- does not depend on the draws before it and is requested FOUR steps ahead (a branch of thousands of segments is
    // walked by one wave: with the loads a step ahead a step costs a memory round trip)
    double dq[4] = {0.0, 0.0, 0.0, 0.0}, bq[4][NS];
    auto beta_row = [&](int i, double (&bv)[NS]) {
      const int kk = min(max(m - i - 1, 0), p.klong - 1);
      const double* bl = s_col + ((kk < ktab ? kk : ktab - 1) * NS + cs) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) bv[c] = bl[c];
      if (kk >= ktab) {
        const double* __restrict__ bg = p.colL + ((size_t)kk * NS + cs) * NS;
#pragma unroll
        for (int c = 0; c < NS; ++c) bv[c] = bg[c];
      }
    };
    if (LONG) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < NS; ++c) bq[q][c] = 0.0;
        if (1 + q < m) { dq[q] = IN(1 + q); beta_row(1 + q, bq[q]); }
      }
    }
    // Four steps per Philox block of the state stream (the step index is wave-uniform: draw i - 1 is a fixed word of it).
    for (int i0 = 1; i0 < mmax; i0 += 4) {
      uint32_t wd[4] = {0u, 0u, 0u, 0u};
      if (i0 < mmax - 1)                       // some lane still draws in this group (draws exist for i < m - 1)
        philox4x32((uint32_t)((i0 - 1) >> 2), ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi, wd);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
      const int i = i0 + q;
      if (i < m) {
        int si;
        double di;
        if (LONG) {
          if (m - i - 1 >= p.klong) err |= DERR_CAPACITY;
          double pr[NS];
#pragma unroll
          for (int c = 0; c < NS; ++c) pr[c] = s_B2[cur_s * NS + c] * bq[q][c];
          si = (i == m - 1) ? cs : sample_cat<NS>(pr, u01(wd[q]), err);
          di = dq[q];
          if (i + 4 < m) { dq[q] = IN(i + 4); beta_row(i + 4, bq[q]); }
        } else {
          si = (i == m - 1) ? cs : draw_state_w(i, cur_s, wd[q]);
          di = dnext;
          if (i + 1 < m) dnext = IN(i + 1);
        }
        if (KS) s_cnt[(cur_s * NS + si) * 64 + lane] = (uint16_t)(s_cnt[(cur_s * NS + si) * 64 + lane] + 1u);               // shortenerbf :1010-1014
        if (si == cur_s) cur_len = cur_len + di;                           // shortener :54
        else {
          if (w == 0) first_len = cur_len; else if (w == 1) second_len = cur_len; else IN(w) = cur_len;
          if (LONG) MS(w) = (uint8_t)cur_s;
          else if (w < 32) pk0 |= (uint64_t)cur_s << (2 * w); else pk1 |= (uint64_t)cur_s << (2 * (w - 32));
          if (!KS) s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * 64 + lane] = (uint16_t)(s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * 64 + lane] + 1u);   // shortener :65-66
          ++w; cur_s = si; cur_len = di;
        }
      }
      }
    }
    if (LONG) MS(w) = (uint8_t)cur_s;
    else if (w < 32) pk0 |= (uint64_t)cur_s << (2 * w); else pk1 |= (uint64_t)cur_s << (2 * (w - 32));
    const int first_s = (w == 0) ? cur_s : (LONG ? (int)MS(0) : (int)(pk0 & 3u));
    const int nmerged = w + 1;
    const double len0 = (w == 0) ? cur_len : first_len;

    // Pass B: one new piece per step for every lane (virtual jumps :391-410, dwell sums :745-757).
    int j = 0;
    int s = first_s;
    int snext = (LONG && nmerged > 1) ? (int)MS(1) : 0;      // LONG: the next merged segment's state rides a step ahead, like its length
    double len = len0;
    double lnext = (nmerged > 1) ? ((w == 1) ? cur_len : second_len) : 0.0;
    double tot = 0.0, scale = s_scale[s], acc = s_dw[s * 64 + lane];
    bool stuck = false, done = false;
    // Four steps per Philox block of the exponential stream: a lane draws one variate per piece until it meets a zero-length
    // segment and none after (`stuck`), so whenever it draws its draw counter equals the step index.
    for (uint32_t t0 = 0; __any(!done); t0 += 4) {
      uint32_t wd[4];
      philox4x32(t0 >> 2, ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi, wd);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (done) continue;
        double piece;
        bool adv;
        if (stuck || !(0.0 < len)) { stuck = true; piece = len; adv = true; }
        else {
          double rl = scale * neglog_u32(wd[q], s_ltab);                       // :398
          if ((tot + rl) < len) { piece = rl; tot += rl; adv = false; }
          else { piece = len - tot; adv = true; }
        }
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = piece; else err |= DERR_CAPACITY;
        acc += piece;                                                        // updatedwelltimes :752
        ++mnew;
        if (adv) {
          s_dw[s * 64 + lane] = acc;
          ++j;
          if (j >= nmerged) done = true;
          else {
            len = lnext;
            if (j + 1 < w) lnext = IN(j + 1); else lnext = cur_len;          // the last merged segment never left its register
            if (LONG) { s = snext; if (j + 1 < nmerged) snext = (int)MS(j + 1); }
            else s = (int)(((j < 32) ? (pk0 >> (2 * j)) : (pk1 >> (2 * (j - 32)))) & 3u);
            scale = s_scale[s]; tot = 0.0; acc = s_dw[s * 64 + lane];
          }
        }
      }
    }
  } else if (!LONG) {
    // General path (a lane with more than 64 segments on this branch): the reference's loop nest as written.
    uint32_t edraw = 0;
    bool stuck = false;
    auto finalize = [&](int s, double len) {
      if (stuck || !(0.0 < len)) {
        stuck = true;
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = len; else err |= DERR_CAPACITY;
        s_dw[s * 64 + lane] += len;
        ++mnew;
        return;
      }
      const double scale = s_scale[s];
      double tot = 0.0;
      double acc = s_dw[s * 64 + lane];
      while (tot < len) {
        double rl = scale * neglog_u32(se.draw_word(edraw++), s_ltab);
        double piece;
        if ((tot + rl) < len) { piece = rl; tot += rl; }
        else { piece = len - tot; tot = len; }
        if (mnew < cap) at(out, (uint32_t)mnew * 512u + lane8) = piece; else err |= DERR_CAPACITY;
        acc += piece;
        ++mnew;
      }
      s_dw[s * 64 + lane] = acc;
    };
    int cur_s = (m == 1) ? cs : ps;
    double cur_len = IN(0);
    for (int i = 1; i <= m; ++i) {              // i == m: sentinel that flushes the last merged segment
      int si = -1;
      double di = 0.0;
      if (i < m) {
        si = (i == m - 1) ? cs : draw_state(i, cur_s);
        di = IN(i);
      }
      if (KS && si >= 0) s_cnt[(cur_s * NS + si) * 64 + lane] = (uint16_t)(s_cnt[(cur_s * NS + si) * 64 + lane] + 1u);
      if (si == cur_s) cur_len = cur_len + di;
      else {
        finalize(cur_s, cur_len);
        if (!KS && si >= 0) s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * 64 + lane] = (uint16_t)(s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * 64 + lane] + 1u);
        cur_s = si; cur_len = di;
      }
    }
  }
  if (mnew > cap) mnew = cap;
  if (mnew > 65535) { err |= DERR_CAPACITY; mnew = 65535; }
  mct[b * 64 + lane] = (uint16_t)mnew;
  }      // next branch of the group
  flush_counts();

  double* pd = p.pdw + (((size_t)tile * p.n_edge + grp) * NS) * 64 + lane;      // [tile][group][NS][64] (n_edge rows reserved)
#pragma unroll
  for (int c = 0; c < NS; ++c) pd[c * 64] = s_dw[c * 64 + lane];
  if (err) atomicOr(p.err, err);
}

// Dwell sums, second stage: a wave per (tile, chunk of TILES_CHUNK group partials), added in group order; the segments now
// held by an equal share of the edges (for the read + written counter; one global atomic per wave would serialise).
template <int NS>
__global__ __launch_bounds__(TILES_BLOCK) void tiles_chunk_kernel(TileParams<NS> p) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (TILES_BLOCK / 64) + (threadIdx.x >> 6);
  if (item >= p.n_chunks * p.n_tiles) return;
  const int tile = item / p.n_chunks, chunk = item % p.n_chunks;
  const int b0 = chunk * TILES_CHUNK, b1 = min(b0 + TILES_CHUNK, p.n_groups);
  double s[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) s[c] = 0.0;
  const double* src = p.pdw + ((size_t)tile * p.n_edge * NS) * 64 + lane;
  for (int b = b0; b < b1; ++b)
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] += src[((size_t)b * NS + c) * 64];
  double* dst = p.pchunk + (((size_t)tile * p.n_chunks + chunk) * NS) * 64 + lane;
#pragma unroll
  for (int c = 0; c < NS; ++c) dst[c * 64] = s[c];
  uint32_t segs = 0;
  const uint16_t* mc = p.mcount + ((size_t)tile * p.n_edge) * 64 + lane;
  const int per = (p.n_edge + p.n_chunks - 1) / p.n_chunks;
  for (int b = chunk * per; b < min((chunk + 1) * per, p.n_edge); ++b) segs += mc[(size_t)b * 64];
  p.pseg[((size_t)tile * p.n_chunks + chunk) * 64 + lane] = segs;
}

// Third stage and the statistics row: a workgroup of TILES_STATS_WAVES waves per tile.  Wave w adds its contiguous share of the
// chunk sums in chunk order (and its share of the counter copies); wave 0 then adds the waves' partial sums in wave order -- a fixed
// shape that depends on the number of chunks only, so a run reproduces itself bit for bit -- and writes the row.  (One wave per
// tile walked all 313 chunk sums of C3 alone: with a handful of tiles the reductions were a quarter of the sweep.)
// Columns: n dwell sums, the counters, (ks) the root state.
constexpr int TILES_STATS_WAVES = 8;

template <int NS, bool KS>
__global__ __launch_bounds__(64 * TILES_STATS_WAVES) void tiles_stats_kernel(TileParams<NS> p, int it) {
  constexpr int NCNT = KS ? NS * NS : NS * (NS - 1);
  constexpr int DCOLS = NS + NCNT + (KS ? 1 : 0);
  constexpr int SW = TILES_STATS_WAVES;
  __shared__ double s_dw[SW][NS][64];
  __shared__ uint32_t s_ct[SW][NCNT][64];
  __shared__ uint32_t s_sg[SW][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int rep_local = tile * 64 + lane;
  const bool valid = rep_local < p.n_rep;
  {
    const int per = (p.n_chunks + SW - 1) / SW;
    const int c0 = min(wave * per, p.n_chunks), c1 = min(c0 + per, p.n_chunks);
    double part[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) part[c] = 0.0;
    const double* src = p.pchunk + ((size_t)tile * p.n_chunks * NS) * 64 + lane;
    uint32_t segs = 0;
    for (int ch = c0; ch < c1; ++ch) {
#pragma unroll
      for (int c = 0; c < NS; ++c) part[c] += src[((size_t)ch * NS + c) * 64];
      segs += p.pseg[((size_t)tile * p.n_chunks + ch) * 64 + lane];
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) s_dw[wave][c][lane] = part[c];
    s_sg[wave][lane] = segs;
    uint32_t tot[NCNT];
#pragma unroll
    for (int c = 0; c < NCNT; ++c) tot[c] = 0u;
    for (int k = wave; k < p.cnt_copies; k += SW) {         // integer sums: exact in any order
      uint32_t* gc = p.cnt + (((size_t)tile * p.cnt_copies + k) * NS * NS) * 64 + lane;
#pragma unroll
      for (int c = 0; c < NCNT; ++c) { tot[c] += gc[c * 64]; gc[c * 64] = 0u; }
    }
#pragma unroll
    for (int c = 0; c < NCNT; ++c) s_ct[wave][c][lane] = tot[c];
  }
  __syncthreads();
  if (wave != 0) return;
  double col[DCOLS];
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    double v = s_dw[0][c][lane];
#pragma unroll
    for (int w = 1; w < SW; ++w) v += s_dw[w][c][lane];
    col[c] = v;
  }
#pragma unroll
  for (int c = 0; c < NCNT; ++c) {
    uint32_t v = 0u;
#pragma unroll
    for (int w = 0; w < SW; ++w) v += s_ct[w][c][lane];
    col[NS + c] = (double)v;
  }
  {   // segments read (the previous sweep's total) + written (this sweep's), valid replicas only
    uint32_t segs = 0;
#pragma unroll
    for (int w = 0; w < SW; ++w) segs += s_sg[w][lane];
    if (!valid) segs = 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) segs += __shfl_xor(segs, off, 64);
    if (lane == 0) { atomicAdd(p.segcnt, (unsigned long long)p.segprev[tile] + segs); p.segprev[tile] = segs; }
  }
  if (KS) col[NS + NCNT] = (double)p.nstate[((size_t)tile * p.n_node + p.root) * 64 + lane];   // :1350-1352
  if (p.reduce) {
    double* dst = p.stats + ((size_t)it * p.n_tiles + tile) * p.n_cols;
#pragma unroll
    for (int c = 0; c < DCOLS; ++c) {
      double v = valid ? col[c] : 0.0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == c) dst[c] = v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < DCOLS; ++c) p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + rep_local] = col[c];
  }
}

// Load the caller's initial paths (x$maps, makeabranch src/phylomap.cpp:24-34) into every replica of every tile.
__global__ void tiles_init_kernel(int n_edge, int n_tiles, int64_t rows, const int32_t* __restrict__ slot,
                                  const int32_t* __restrict__ map_off, const double* __restrict__ maps,
                                  double* __restrict__ dw0, uint16_t* __restrict__ mcount) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (item >= n_edge * n_tiles) return;
  const int tile = item / n_edge, b = item % n_edge;
  const int o = map_off[b], m = map_off[b + 1] - o;
  double* dst = dw0 + ((size_t)tile * rows + slot[b]) * 64 + lane;
  for (int i = 0; i < m; ++i) dst[(size_t)i * 64] = maps[o + i];
  mcount[((size_t)tile * n_edge + b) * 64 + lane] = (uint16_t)m;
}

}  // namespace

hipError_t launch_tiles_init(int n_edge, int n_tiles, int64_t rows, const int32_t* slot, const int32_t* map_off,
                             const double* maps, double* dw0, uint16_t* mcount, hipStream_t stream) {
  const int64_t items = (int64_t)n_edge * n_tiles;
  hipLaunchKernelGGL(tiles_init_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, n_edge, n_tiles, rows, slot, map_off,
                     maps, dw0, mcount);
  return hipGetLastError();
}

template <int NS>
hipError_t launch_tiles_sweep(const TileParams<NS>& p, const std::vector<int32_t>& up_off,
                              const std::vector<int32_t>& down_off, const std::vector<int32_t>& tier_off, int it, hipStream_t stream,
                              hipEvent_t* phase_ev) {
  constexpr int WPB = TILES_BLOCK / 64;
  auto blocks = [&](int64_t items) { return dim3((unsigned)((items + WPB - 1) / WPB)); };
  auto pblocks = [&](int64_t items) { return dim3((unsigned)std::min<int64_t>((items + WPB - 1) / WPB, TILES_PERSISTENT_WGS)); };      // node draws: persistent waves, 8 per SIMD
  auto mark = [&](int i) { if (phase_ev) (void)hipEventRecord(phase_ev[i], stream); };
  mark(0);
  if (p.cl_nodes) {                                  // few tiles: subtree clusters, one launch per tier and pass
    const int n_tiers = (int)tier_off.size() - 1;
    for (int t = 0; t < n_tiers; ++t) {
      const int n_cl = tier_off[t + 1] - tier_off[t];
      if (p.mstate) hipLaunchKernelGGL((tiles_up_cluster_kernel<NS, true>), dim3((unsigned)((int64_t)n_cl * p.n_tiles)), dim3(TILES_CL_BLOCK), 0, stream, p, tier_off[t], n_cl);
      else hipLaunchKernelGGL((tiles_up_cluster_kernel<NS, false>), dim3((unsigned)((int64_t)n_cl * p.n_tiles)), dim3(TILES_CL_BLOCK), 0, stream, p, tier_off[t], n_cl);
    }
    mark(1);
    for (int t = n_tiers - 1; t >= 0; --t) {
      const int n_cl = tier_off[t + 1] - tier_off[t];
      hipLaunchKernelGGL(tiles_down_cluster_kernel<NS>, dim3((unsigned)((int64_t)n_cl * p.n_tiles)), dim3(TILES_CL_BLOCK), 0, stream, p, it, tier_off[t], n_cl,
                         t == n_tiers - 1 ? 1 : 0);
    }
  } else {
    for (size_t l = 0; l + 1 < up_off.size(); ++l) {
      const int n = up_off[l + 1] - up_off[l];
      if (n > 0) hipLaunchKernelGGL(tiles_up_kernel<NS>, blocks((int64_t)n * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, up_off[l], up_off[l + 1]);
    }
    mark(1);
    hipLaunchKernelGGL(tiles_root_kernel<NS>, blocks(p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it);
    for (size_t l = 0; l + 1 < down_off.size(); ++l) {
      const int n = down_off[l + 1] - down_off[l];
      if (n > 0) hipLaunchKernelGGL(tiles_down_kernel<NS>, pblocks((int64_t)n * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it, down_off[l], down_off[l + 1]);
    }
  }
  mark(2);
  if (p.mstate) {
    if (p.ks) hipLaunchKernelGGL((tiles_branch_kernel<NS, true, true>), blocks((int64_t)p.n_groups * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it);
    else hipLaunchKernelGGL((tiles_branch_kernel<NS, false, true>), blocks((int64_t)p.n_groups * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it);
  } else {
    if (p.ks) hipLaunchKernelGGL((tiles_branch_kernel<NS, true, false>), blocks((int64_t)p.n_groups * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it);
    else hipLaunchKernelGGL((tiles_branch_kernel<NS, false, false>), blocks((int64_t)p.n_groups * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p, it);
  }
  mark(3);
  hipLaunchKernelGGL(tiles_chunk_kernel<NS>, blocks((int64_t)p.n_chunks * p.n_tiles), dim3(TILES_BLOCK), 0, stream, p);
  if (p.ks) hipLaunchKernelGGL((tiles_stats_kernel<NS, true>), dim3(p.n_tiles), dim3(64 * TILES_STATS_WAVES), 0, stream, p, it);
  else hipLaunchKernelGGL((tiles_stats_kernel<NS, false>), dim3(p.n_tiles), dim3(64 * TILES_STATS_WAVES), 0, stream, p, it);
  mark(4);
  return hipGetLastError();
}

template hipError_t launch_tiles_sweep<2>(const TileParams<2>&, const std::vector<int32_t>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, hipEvent_t*);
template hipError_t launch_tiles_sweep<3>(const TileParams<3>&, const std::vector<int32_t>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, hipEvent_t*);
template hipError_t launch_tiles_sweep<4>(const TileParams<4>&, const std::vector<int32_t>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t, hipEvent_t*);

}  // namespace phm
