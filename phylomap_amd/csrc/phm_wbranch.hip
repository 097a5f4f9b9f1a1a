// phm_wbranch.hip -- 5..64 states, one wavefront per (replica, branch); see phm_wbranch.h.
// Lanes are states; everything else (segment counts, end states, merged paths, draws) is the same in all 64 lanes of the
// wave, so the compiler keeps it in scalar registers and the loops never diverge.  Uniform values that go through memory
// (merged segments, end states) are written by ALL lanes (identical stores) so that each lane's own program order
// guarantees what it reads back.  Arithmetic, draws and stream addressing are those of phm_wide.hip / the oracle.
#include "phm_wbranch.h"

#include "phm_coop.h"

namespace phm {

namespace {

__host__ __device__ inline size_t wb_model_lds_bytes(int n, bool sparse) {
  return sizeof(double) * ((size_t)(sparse ? 2 : 1) * n * (n | 1) + n);
}

struct Lds {
  double* Bc;      // [n][ldn] chain matrix
  double* B2;      // [n][ldn] dense rows for the forward step (= Bc unless SPARSE)
  double* scale;   // [n]
  int ldn;
};

// stage the model matrices (odd row stride: conflict-free row and column access); ends with a barrier
__device__ __forceinline__ Lds stage_model(const WideBranchParams& p, unsigned char* smem, int bs = WB_BLOCK) {
  const int n = p.n_states;
  Lds l;
  l.ldn = n | 1;
  l.Bc = reinterpret_cast<double*>(smem);
  l.B2 = p.sparse ? l.Bc + n * l.ldn : l.Bc;
  l.scale = l.Bc + (p.sparse ? 2 : 1) * n * l.ldn;
  for (int i = threadIdx.x; i < n * n; i += bs) {
    const int r = i / n, cc = i - r * n;
    l.Bc[r * l.ldn + cc] = p.Bc[i];
    if (p.sparse) l.B2[r * l.ldn + cc] = p.B2[i];
  }
  if ((int)threadIdx.x < n) l.scale[threadIdx.x] = p.scale[threadIdx.x];
  __syncthreads();
  return l;
}

// One height level of the pruning sweep for the waves of a workgroup of BS threads: TWO waves per node, one per child -- the two
// chains B^(m-1) PL[child] are the longest dependent line of a level (up to ~15 steps of n fused multiply-adds each) and do not
// depend on one another; the "second" child's wave hands its vector over through LDS and the "first" child's wave finishes the
// node.  `first` = position of the workgroup's first node in the level order.  Contains one barrier (all waves reach it).
struct UpLds {
  const int32_t* ecol; const double* eval;   // ELLPACK rows of a sparse chain matrix (ell_w > 0), else the dense model in `l`
  Lds l;
  double (*second)[64];                      // [BS / 128][64]
  double (*vec)[64];                         // [BS / 64][64] the chain vector of each wave (coop_matvec_lds)
};

template <int BS>
__device__ __forceinline__ void wb_up_level(const WideBranchParams& p, const UpLds& sh, const double (&brow)[64], int first, int end, int r,
                                            uint32_t& err) {
  const int n = p.n_states, lane = threadIdx.x & 63, ell_w = p.ell_w;
  const int wave = threadIdx.x >> 6, slot = wave >> 1, which = 1 - (wave & 1);      // even wave: child[1] ("first"), odd wave: child[0]
  const int idx = first + slot;
  const bool live = idx < end;
  const int c = lane < n ? lane : n - 1;
  const UpStep st = p.up[p.up_order[live ? idx : first]];
  const int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  double* PLr = p.PL + (size_t)r * p.n_node * n;
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  // B^k applied to a child's partial-likelihood vector (mmmmvFORpl :446-450); tips: a row of the chain table
  auto child_vec = [&](int child, int k) -> double {
    if (child < 0) {
      const int ts = tips[~child];
      if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
      return p.tip_masks ? p.maskL[((size_t)k * 2 + (ts & 1)) * n + c] : p.colL[((size_t)k * n + ts) * n + c];
    }
    double v = PLr[(size_t)child * n + c];
    if (p.band_hb == 1) { const double (&cf)[5] = reinterpret_cast<const double (&)[5]>(brow); for (int i = 0; i < k; ++i) v = coop_matvec_band<1>(cf, v, lane); }
    else if (p.band_hb == 2) { const double (&cf)[5] = reinterpret_cast<const double (&)[5]>(brow); for (int i = 0; i < k; ++i) v = coop_matvec_band<2>(cf, v, lane); }
    else if (ell_w > 0) for (int i = 0; i < k; ++i) v = coop_matvec_ell(sh.ecol, sh.eval, v, ell_w, c);
    else for (int i = 0; i < k; ++i) v = coop_matvec_regs(brow, sh.vec[wave], v, n, lane);
    return v;
  };
  double x = 0.0;
  if (live) x = child_vec(st.child[which], mc[st.edge[which]] - 1);       // which = 1: "first" (:508), 0: "second" (:509)
  if (which == 0) sh.second[slot][lane] = x;
  __syncthreads();
  if (live && which == 1) {
    x = x * sh.second[slot][lane];                              // :510
    if (p.normalise) x = x / coop_sum(x, n);                    // :525
    if (lane < n) PLr[(size_t)st.parent * n + lane] = x;
  }
}

template <int BS>
__device__ __forceinline__ UpLds wb_up_stage(const WideBranchParams& p, unsigned char* smem, double (*second)[64], double (*vec)[64]) {
  UpLds sh;
  const int ell_w = p.ell_w;                 // sparse chain matrix: only its ELLPACK rows are staged; else the dense matrix
  double* s_eval = reinterpret_cast<double*>(smem);
  int32_t* s_ecol = reinterpret_cast<int32_t*>(s_eval + p.n_states * ell_w);
  sh.l = {nullptr, nullptr, nullptr, 0};
  if (ell_w > 0) {
    for (int i = threadIdx.x; i < p.n_states * ell_w; i += BS) { s_ecol[i] = p.ell_col[i]; s_eval[i] = p.ell_val[i]; }
    __syncthreads();
  } else {
    sh.l = stage_model(p, smem, BS);       // ends with a barrier
  }
  sh.ecol = s_ecol; sh.eval = s_eval; sh.second = second; sh.vec = vec;
  return sh;
}

// the lane's row of the dense chain matrix, in registers for the whole kernel (coop_matvec_regs); zeros beyond n / for ELLPACK
__device__ __forceinline__ void wb_load_row(const WideBranchParams& p, const UpLds& sh, double (&brow)[64]) {
  const int n = p.n_states, lane = threadIdx.x & 63;
  const int c = lane < n ? lane : n - 1;
#pragma unroll
  for (int j = 0; j < 64; ++j) brow[j] = (p.ell_w == 0 && j < n) ? sh.l.Bc[c * sh.l.ldn + j] : 0.0;
  if (p.band_hb > 0) {                             // banded: brow[d] = M[c][c + d - hb] gathered from the lane's ELLPACK row
    for (int t = 0; t < p.ell_w; ++t) {
      const int d = sh.ecol[c * p.ell_w + t] - c + p.band_hb;
      const double val = sh.eval[c * p.ell_w + t];
#pragma unroll
      for (int q = 0; q < 5; ++q) brow[q] = (q == d) ? brow[q] + val : brow[q];      // padding entries carry value 0
    }
  }
}

__global__ __launch_bounds__(WB_BLOCK) void wb_up_kernel(WideBranchParams p, int begin, int end) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_second[WB_BLOCK / 128][64];
  __shared__ __align__(16) double s_vec[WB_BLOCK / 64][64];
  const UpLds sh = wb_up_stage<WB_BLOCK>(p, smem, s_second, s_vec);
  double brow[64];
  wb_load_row(p, sh, brow);
  uint32_t err = 0;
  wb_up_level<WB_BLOCK>(p, sh, brow, begin + blockIdx.x * (WB_BLOCK / 128), end, blockIdx.y, err);
  if (err) atomicOr(p.err, err);
}

// A RUN of consecutive narrow height levels (at most four or eight nodes each) in one launch: one workgroup per chain stages the
// model once and walks the levels with a workgroup-scope fence and a barrier in between -- near the root a level is a handful of
// nodes, and a launch of its own (ramp-up + staging 30 KB of B at 61 states) costs more than its chains.
// The dense chain matrix keeps a row per lane in registers (128 VGPRs: 512-lane workgroups, four nodes per level); an ELLPACK
// matrix does not (1 024 lanes, eight nodes per level).
template <int BS, bool DENSE>
__global__ __launch_bounds__(BS) void wb_up_run_kernel(WideBranchParams p, int l0, int l1) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_second[BS / 128][64];
  __shared__ __align__(16) double s_vec[BS / 64][64];
  const UpLds sh = wb_up_stage<BS>(p, smem, s_second, s_vec);
  double brow[64];
  if (DENSE || p.band_hb > 0) wb_load_row(p, sh, brow);
  else {
#pragma unroll
    for (int j = 0; j < 64; ++j) brow[j] = 0.0;
  }
  uint32_t err = 0;
  for (int l = l0; l < l1; ++l) {
    wb_up_level<BS>(p, sh, brow, p.up_off[l], p.up_off[l + 1], blockIdx.x, err);
    __threadfence_block();
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

__global__ __launch_bounds__(WB_BLOCK) void wb_root_kernel(WideBranchParams p, int it) {
  const int n = p.n_states, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (WB_BLOCK / 64) + (threadIdx.x >> 6);
  if (r >= p.n_rep) return;
  const int c = lane < n ? lane : n - 1;
  const double* PLr = p.PL + (size_t)r * p.n_node * n;
  uint32_t err = 0;
  const double pr = (lane < n) ? p.pid[c] * PLr[(size_t)p.root * n + c] : 0.0;     // :618
  const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it,
                            ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
  p.nstate[(size_t)r * p.n_node + p.root] = (uint8_t)coop_sample(pr, u, n, lane, err);   // :627 (every lane stores the same value)
  if (err) atomicOr(p.err, err);
}

// child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their parity mask (:1384-1397)
__global__ __launch_bounds__(WB_BLOCK) void wb_down_kernel(WideBranchParams p, int it, int begin, int end) {
  const int n = p.n_states, lane = threadIdx.x & 63;
  const int idx = begin + blockIdx.x * (WB_BLOCK / 64) + (threadIdx.x >> 6);
  const int r = blockIdx.y;
  if (idx >= end) return;
  const int c = lane < n ? lane : n - 1;
  const DownStep ds = p.down[p.down_order[idx]];
  const int b = ds.edge;
  const int m = p.mcount[(size_t)r * p.n_edge + b];
  uint8_t* nst = p.nstate + (size_t)r * p.n_node;
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  const int ps = nst[ds.parent];
  uint32_t err = 0;
  int cs;
  if (ds.child >= 0 || p.tip_masks) {
    int kk = m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    double w = p.rowL[((size_t)kk * n + ps) * n + c];
    uint32_t node_id;
    if (ds.child >= 0) {
      w = (lane < n) ? w * p.PL[((size_t)r * p.n_node + ds.child) * n + c] : 0.0;
      node_id = (uint32_t)(ds.child + p.n_tips);
    } else {
      const int par = tips[~ds.child] & 1;
      w = (lane < n) ? w * (((c & 1) == par) ? 1.0 : 0.0) : 0.0;
      node_id = (uint32_t)(~ds.child);
    }
    const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it, ENT_NODE | node_id, 0);
    cs = coop_sample(w, u, n, lane, err);                                      // :655
    if (ds.child >= 0) nst[ds.child] = (uint8_t)cs;
  } else {
    cs = tips[~ds.child];                                                      // :612
  }
  uint8_t* es = p.estate + ((size_t)r * p.n_edge + b) * 2;
  es[0] = (uint8_t)ps; es[1] = (uint8_t)cs;                                    // updatenodestates :460-475
  if (err) atomicOr(p.err, err);
}

// The sampling sweep in two steps, as in phm_narrow.hip: the draw child ~ e_ps^T B^(m-1) (.) PL[child] depends on the sweep only
// through the parent's state ps, so wb_downmap_kernel (a wave per (replica, edge), all edges side by side) carries it out for
// EACH of the n possible ps -- one byte per candidate: the drawn state, bit 7 = "probabilities all zero" -- and wb_walk_kernel (one
// workgroup per chain) walks the depth levels with state[child] = map[edge][state[parent]], node states in LDS, one byte
// gathered per edge and level.  Same operands, same order as wb_down_kernel for the ps that materialises.
__global__ __launch_bounds__(WB_BLOCK) void wb_downmap_kernel(WideBranchParams p, int it) {
  __shared__ double s_w[WB_BLOCK / 64][64];
  const int n = p.n_states, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x * (WB_BLOCK / 64) + (threadIdx.x >> 6);          // position in depth-level order
  const int r = blockIdx.y;
  if (idx >= p.n_edge) return;
  const int c = lane < n ? lane : n - 1;
  const DownStep ds = p.down[p.down_order[idx]];
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  uint8_t* __restrict__ map = p.dmap + ((size_t)r * p.n_edge + idx) * n;
  uint32_t err = 0;
  if (ds.child >= 0 || p.tip_masks) {
    int kk = p.mcount[(size_t)r * p.n_edge + ds.edge] - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    double wgt;
    uint32_t node_id;
    if (ds.child >= 0) {
      wgt = p.PL[((size_t)r * p.n_node + ds.child) * n + c];
      node_id = (uint32_t)(ds.child + p.n_tips);
    } else {
      const int par = tips[~ds.child] & 1;
      wgt = ((c & 1) == par) ? 1.0 : 0.0;
      node_id = (uint32_t)(~ds.child);
    }
    const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it, ENT_NODE | node_id, 0);
    // lane = CANDIDATE parent state q: the lane walks its own row of (Bc^T)^kk against the weights (broadcast from LDS), once for
    // the total and once for the count of partial sums below u * total -- the sums coop_sample forms (index order, unfused),
    // 2 n short steps per wave instead of n draws of n readlane steps each
    const double* __restrict__ rq = p.rowL + ((size_t)kk * n + c) * n;
    s_w[wave][lane] = (lane < n) ? wgt : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    double run = rq[0] * s_w[wave][0];
    for (int j = 1; j < n; ++j) run += rq[j] * s_w[wave][j];
    const bool bad = !(run > 0.0) || isinf(run);
    const double thr = u * run;
    double cum = rq[0] * s_w[wave][0];
    int cnt = (thr <= cum) ? 0 : 1;
    for (int j = 1; j < n; ++j) { cum += rq[j] * s_w[wave][j]; cnt += (thr <= cum) ? 0 : 1; }
    const int cs = cnt < n ? cnt : n - 1;
    if (lane < n) map[lane] = (uint8_t)(cs | (bad ? 0x80 : 0));
  } else {
    if (lane < n) map[lane] = tips[~ds.child];                                 // :612
  }
  if (err) atomicOr(p.err, err);
}

constexpr int WB_WALK_BLOCK = 1024;
__global__ __launch_bounds__(WB_WALK_BLOCK) void wb_walk_kernel(WideBranchParams p, int it, int n_levels, int use_lds) {
  extern __shared__ uint8_t s_nst[];
  const int n = p.n_states, tid = threadIdx.x, r = blockIdx.x;
  uint8_t* __restrict__ nst = p.nstate + (size_t)r * p.n_node;
  uint8_t* __restrict__ est = p.estate + (size_t)r * p.n_edge * 2;
  const uint8_t* __restrict__ map = p.dmap + (size_t)r * p.n_edge * n;
  uint32_t err = 0;
  if (tid < 64) {                                    // the root draw (:618-627), by the first wave
    const int c = tid < n ? tid : n - 1;
    const double pr = (tid < n) ? p.pid[c] * p.PL[((size_t)r * p.n_node + p.root) * n + c] : 0.0;
    const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it, ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
    const int rs = coop_sample(pr, u, n, tid, err);
    nst[p.root] = (uint8_t)rs;                       // every lane stores the same value
    if (use_lds) s_nst[p.root] = (uint8_t)rs;
  }
  if (!use_lds) __threadfence_block();
  __syncthreads();
  for (int l = 0; l < n_levels; ++l) {
    const int lo = p.down_off[l], hi = p.down_off[l + 1];
    for (int idx = lo + tid; idx < hi; idx += WB_WALK_BLOCK) {
      const DownStep ds = p.down[p.down_order[idx]];
      const int ps = use_lds ? s_nst[ds.parent] : nst[ds.parent];
      const int out = map[(size_t)idx * n + ps];
      const int cs = out & 0x7f;
      if (out & 0x80) err |= DERR_ZERO_PROB;
      if (ds.child >= 0) {
        if (use_lds) s_nst[ds.child] = (uint8_t)cs;
        nst[ds.child] = (uint8_t)cs;
      }
      est[ds.edge * 2] = (uint8_t)ps; est[ds.edge * 2 + 1] = (uint8_t)cs;     // updatenodestates :460-475
    }
    if (!use_lds) __threadfence_block();
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// One branch of one replica: resamplebranchstates :264-308, shortener :44-73 (shortenerbf :997-1030), virtual jumps
// sampleabranch :391-410, dwell sums updatedwelltimes :745-757.
__global__ __launch_bounds__(WB_BLOCK) void wb_branch_kernel(WideBranchParams p, int it, int b2_in_lds) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef PHM_DEBUG_LEVEL_CLOCK
  const unsigned long long tk0 = wall_clock64();
  unsigned long long tk1 = 0, tk2 = 0;
#endif
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];        // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += WB_BLOCK) s_ltab[i] = logtab_entry(i);
  // This kernel touches one row of the forward-step matrix per draw: it reads that row from global memory (L2) instead of
  // staging the whole matrix per workgroup -- n^2 loads for four waves that use a handful of rows -- which also leaves the
  // LDS to the log table and the ELLPACK rows (when used), i.e. full occupancy.
  const int w2 = p.ell2_w;
  double* s_e2val = reinterpret_cast<double*>(smem);
  int32_t* s_e2col = reinterpret_cast<int32_t*>(s_e2val + p.n_states * w2);
  for (int i = threadIdx.x; i < p.n_states * w2; i += WB_BLOCK) { s_e2col[i] = p.ell2_col[i]; s_e2val[i] = p.ell2_val[i]; }
  // A step of this kernel is ~1 us of wave-cooperative arithmetic; a global round trip in it doubles or triples it, and with a
  // handful of chains the kernel lasts as long as its longest branch.  So: the dense forward-step matrix is staged in LDS when the
  // chains are few (b2_in_lds; its row index is the state just drawn), the operands that do not depend on the state -- the next
  // old segment's length, the next backward row -- are requested one step ahead, and the merged segments wait for pass B in LDS
  // (the first WB_MERGED_LDS of a branch; the rest in the global scratch).
  constexpr int WB_MERGED_LDS = 192;
  __shared__ double s_ml[WB_BLOCK / 64][WB_MERGED_LDS];
  __shared__ uint8_t s_ms[WB_BLOCK / 64][WB_MERGED_LDS];
  __shared__ __align__(16) double s_pv[WB_BLOCK / 64][64];       // the probability vector of a draw (coop_sample_lds)
  __shared__ uint16_t s_tr[WB_BLOCK / 64][64];       // counter columns of the transitions met since the last flush (n^2 <= 4 096)
  const int ldb = p.n_states | 1;
  const double* s_B2 = reinterpret_cast<const double*>(smem);
  if (b2_in_lds && w2 == 0) {
    double* dst = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < p.n_states * p.n_states; i += WB_BLOCK) dst[(i / p.n_states) * ldb + i % p.n_states] = p.B2[i];
  }
  __syncthreads();
  const int n = p.n_states, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * (WB_BLOCK / 64) + (threadIdx.x >> 6);
  const int r = blockIdx.y;
  if (idx >= p.n_edge) return;                    // whole waves only; no barrier below this line
  const int c = lane < n ? lane : n - 1;
  const int b = p.branch_order[idx];
  const uint32_t rep = (uint32_t)(p.replica_offset + r);
  int32_t* mc = p.mcount + (size_t)r * p.n_edge;
  const int m = mc[b];
  const uint8_t* es = p.estate + ((size_t)r * p.n_edge + b) * 2;
  const int ps = es[0], cs = es[1];
  const int64_t o = p.off[b];
  const int cap = (int)(p.off[b + 1] - o);
  const double* in = p.dw[it & 1] + (size_t)r * p.total_cap + o;
  double* out = p.dw[(it & 1) ^ 1] + (size_t)r * p.total_cap + o;
  double* ml = p.mlen + (size_t)r * p.total_cap + o;
  uint8_t* ms = p.mstate + (size_t)r * p.total_cap + o;
  double* cnt = p.cnt + (size_t)r * p.n_cols + n;           // transition counters start after the n dwell columns
  uint32_t err = 0;

  Stream su;
  su.open(ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);

  // pass A: interior states s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end (:290, :301-304), neighbours merged (:54)
  const int wv = threadIdx.x >> 6;
  auto put_merged = [&](int k, double len, int st) {
    if (k < WB_MERGED_LDS) { s_ml[wv][k] = len; s_ms[wv][k] = (uint8_t)st; }
    else { ml[k] = len; ms[k] = (uint8_t)st; }
  };
  // Transitions are NOTED in LDS and counted (f64 atomics on integers: exact in any order) 64 at a time: an atomic per step sits
  // in the same in-order memory queue as the loads the next step waits for, i.e. an L2 round trip per step.
  int n_tr = 0;
  auto flush_transitions = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if (lane < n_tr) atomicAdd(cnt + s_tr[wv][lane], 1.0);
    __builtin_amdgcn_wave_barrier();
    n_tr = 0;
  };
  auto note_transition = [&](int col) {
    s_tr[wv][n_tr] = (uint16_t)col;                  // every lane writes the same value
    if (++n_tr == 64) flush_transitions();
  };
  int w = 0;
  int cur_s = (m == 1) ? cs : ps;                    // updatenodestates :469-472 (m == 1: the child end wins)
  double cur_len = in[0];
  // operands of step 1, then always those of the next step while the current one is drawn (clamped addresses, dropped when unused)
  auto beta_row = [&](int i) {                       // lane c: (Bc^(m-i-1) e_cs)[c]
    int kk = m - i - 1;
    if (kk < 0) kk = 0;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    return p.colL[((size_t)kk * n + cs) * n + c];
  };
  double d_next = in[min(1, m - 1)];
  double beta_next = beta_row(1);
#ifdef PHM_DEBUG_LEVEL_CLOCK
  if (cur_len >= 0.0) tk1 = wall_clock64();
#endif
  for (int i = 1; i < m; ++i) {
    const double di = d_next, beta = beta_next;
    d_next = in[min(i + 1, m - 1)];
    beta_next = beta_row(min(i + 1, m - 1));
    int si;
    if (i == m - 1) si = cs;
    else {
      if (w2 > 0) {
        // Sparse forward row: lanes are the row's non-zero slots (columns ascending).  The zero entries of the dense vector
        // add +0 to every partial sum, so the draw over the slots picks the same column as the draw over all n states.
        const int slot = lane < w2 ? cur_s * w2 + lane : cur_s * w2;
        const int mycol = s_e2col[slot];
        const double beta_sel = __shfl(beta, mycol, 64);       // by every lane: a shuffle under a lane condition reads inactive lanes
        const double pr = (lane < w2) ? s_e2val[slot] * beta_sel : 0.0;
        const int t = coop_sample(pr, su.draw((uint32_t)(i - 1)), w2, lane, err);
        si = __builtin_amdgcn_readlane(mycol, t);
      } else {
        const double b2 = b2_in_lds ? s_B2[cur_s * ldb + c] : p.B2[cur_s * n + c];
        const double pr = (lane < n) ? b2 * beta : 0.0;
        si = coop_sample_lds(pr, su.draw((uint32_t)(i - 1)), n, lane, s_pv[wv], err);
      }
    }
    if (p.count_self) note_transition(cur_s * n + si);                                      // shortenerbf :1010-1014
    if (si == cur_s) cur_len = cur_len + di;                                                // shortener :54
    else {
      put_merged(w, cur_len, cur_s);
      if (!p.count_self) note_transition(cur_s * (n - 1) + (si > cur_s ? si - 1 : si));     // :65-66
      ++w; cur_s = si; cur_len = di;
    }
  }
  flush_transitions();
  put_merged(w, cur_len, cur_s);
  const int nmerged = w + 1;
#ifdef PHM_DEBUG_LEVEL_CLOCK
  tk2 = wall_clock64();
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

  // pass B: virtual jumps, gaps ~ Exp(Omega + q_ss) until each merged segment is used up (:391-410); a segment that is not
  // positive leaves itself and everything after it untouched (:397, :405-406).  Lane s carries the dwell sum of state s.
  // The control flow of a branch is wave-uniform, so the exponential variates are produced 64 at a time -- lane t computes
  // -log(U) of draw gen_base + t from its own Philox block -- and consumed one by one through v_readlane; `tot` is still
  // accumulated draw by draw, in order, so every comparison sees the bits the sequential loop would.
  double mine = 0.0;
  int mnew = 0;
  uint32_t edraw = 0, gen_base = 0;
  double nl = 0.0;
  bool have_gen = false;
  bool stuck = false;
  for (int j = 0; j < nmerged; ++j) {
    const int s = (j < WB_MERGED_LDS) ? (int)s_ms[wv][j] : (int)ms[j];
    const double len = (j < WB_MERGED_LDS) ? s_ml[wv][j] : ml[j];
    if (stuck || !(0.0 < len)) {
      stuck = true;
      if (mnew < cap) out[mnew] = len; else err |= DERR_CAPACITY;
      if (lane == s) mine += len;
      ++mnew;
    } else {
      const double scale = p.scale[s];
      double tot = 0.0;
      while (tot < len) {
        if (!have_gen || edraw - gen_base >= 64u) {
          gen_base = edraw; have_gen = true;
          nl = neglog_u32(stream_word(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_BEXP | (uint32_t)b, edraw + (uint32_t)lane), s_ltab);
        }
        const double rl = scale * readlane_f64(nl, (int)(edraw - gen_base));   // :398
        ++edraw;
        double piece;
        if ((tot + rl) < len) { piece = rl; tot += rl; }
        else { piece = len - tot; tot = len; }
        if (mnew < cap) out[mnew] = piece; else err |= DERR_CAPACITY;
        if (lane == s) mine += piece;                                          // updatedwelltimes :752
        ++mnew;
      }
    }
  }
  if (mnew > cap) mnew = cap;
  mc[b] = mnew;
#ifdef PHM_DEBUG_LEVEL_CLOCK
  if (it == 30 && r == 0 && lane == 0 && (idx < 4 || idx == 200))
    printf("wbbranch idx %d m %d mnew %d draws %d: loads %d passA %d passB %d ticks\n", idx, m, mnew, (int)edraw, (int)(tk1 - tk0), (int)(tk2 - tk1),
           (int)(wall_clock64() - tk2));
#endif
  double* part = p.part + ((size_t)r * p.n_edge + b) * (n + 1);
  if (lane < n) part[lane] = mine;
  if (lane == 0) part[n] = (double)(m + mnew);
  if (err) atomicOr(p.err, err);
}

// Statistics row of one replica.  Dwell columns = sums of the per-branch values, one workgroup per (replica, column): thread t
// adds branches t, t + 256, ... then a fixed tree over the 256 partial sums -- a fixed order, identical from run to run, and the
// n + 1 columns of a row reduced side by side (one workgroup streaming the whole [edge][column] block took 0.36 ms per sweep for
// one chain on C5: 40 % of the sweep).  Counters are copied out of the atomic buffer (and cleared) by the same workgroups, a
// slice each; (ks) the root state appended.
__global__ __launch_bounds__(256) void wb_stats_kernel(WideBranchParams p, int it) {
  __shared__ double red[256];
  const int r = blockIdx.x, c = blockIdx.y, n = p.n_states, tid = threadIdx.x;
  const int ncnt = p.count_self ? n * n : n * (n - 1);
  const int pc = n + 1;                             // column n: segments read + written
  const double* __restrict__ part = p.part + (size_t)r * p.n_edge * pc;
  double acc = 0.0;
  for (int e = tid; e < p.n_edge; e += 256) acc += part[(size_t)e * pc + c];
  red[tid] = acc;
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if (tid < half) red[tid] = red[tid] + red[tid + half];
    __syncthreads();
  }
  // without the reduction over replicas the values go straight into the engine's statistics layout ([iter][cols][n_rep_pad])
  auto put = [&](int col, double v) {
    if (p.reduce) p.rowbuf[(size_t)r * p.n_cols + col] = v;
    else p.stats[((size_t)it * p.n_cols + col) * p.n_rep_pad + r] = v;
  };
  if (tid == 0) {
    if (c < n) put(c, red[0]);
    else atomicAdd(p.segcnt, (unsigned long long)red[0]);
  }
  double* cnt = p.cnt + (size_t)r * p.n_cols + n;
  for (int k = c * 256 + tid; k < ncnt; k += 256 * pc) { put(n + k, cnt[k]); cnt[k] = 0.0; }
  if (p.ks && tid == 0 && c == 0)                                              // root state, 0-based (:1350-1352)
    put(n + ncnt, (double)p.nstate[(size_t)r * p.n_node + p.root]);
}

__global__ void wb_emit_kernel(WideBranchParams p, int it) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!p.reduce) {
    if (gid >= (int64_t)p.n_rep * p.n_cols) return;
    const int r = (int)(gid / p.n_cols), c = (int)(gid % p.n_cols);
    p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + r] = p.rowbuf[(size_t)r * p.n_cols + c];
  } else {
    if (gid >= (int64_t)p.n_tiles * p.n_cols) return;
    const int tile = (int)(gid / p.n_cols), c = (int)(gid % p.n_cols);
    double s = 0.0;
    for (int r = tile * 64; r < tile * 64 + 64 && r < p.n_rep; ++r) s += p.rowbuf[(size_t)r * p.n_cols + c];
    p.stats[((size_t)it * p.n_tiles + tile) * p.n_cols + c] = s;
  }
}

}  // namespace

size_t wbranch_lds_bytes(int n, bool sparse) { return wb_model_lds_bytes(n, sparse); }

hipError_t launch_wbranch_sweep(const WideBranchParams& p, const std::vector<int32_t>& up_off,
                                const std::vector<int32_t>& down_off, int it, hipStream_t stream) {
  constexpr int WPB = WB_BLOCK / 64;
  const unsigned S = (unsigned)p.n_rep;
  const size_t lds = wbranch_lds_bytes(p.n_states, p.sparse != 0);
  if (lds > 48 * 1024) {      // SPARSE with ~60 states: two copies of B
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(wb_up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess) return e1;
  }
  const size_t up_lds = p.ell_w > 0 ? (size_t)p.n_states * p.ell_w * 12 : lds;
  const bool dense = p.ell_w == 0;
  const int run_block = dense ? 512 : 1024;
  if (lds > 48 * 1024) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(wb_up_run_kernel<512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess) return e1;
  }
  const int UL = (int)up_off.size() - 1;
  auto narrow = [&](int l) { return up_off[l + 1] - up_off[l] <= run_block / 128; };
  for (int l = 0; l < UL;) {
    if (narrow(l) && l + 1 < UL && narrow(l + 1)) {              // at least two narrow levels in a row: one launch for the run
      int l1 = l;
      while (l1 < UL && narrow(l1)) ++l1;
      if (dense) hipLaunchKernelGGL((wb_up_run_kernel<512, true>), dim3(S), dim3(512), up_lds, stream, p, l, l1);
      else hipLaunchKernelGGL((wb_up_run_kernel<1024, false>), dim3(S), dim3(1024), up_lds, stream, p, l, l1);
      l = l1;
      continue;
    }
    const int cnt = up_off[l + 1] - up_off[l];
    if (cnt > 0) hipLaunchKernelGGL(wb_up_kernel, dim3((cnt + WPB / 2 - 1) / (WPB / 2), S), dim3(WB_BLOCK), up_lds, stream, p, up_off[l], up_off[l + 1]);
    ++l;
  }
  if (p.dmap) {      // transition maps of all edges, then one workgroup per chain draws the root and walks the levels
    hipLaunchKernelGGL(wb_downmap_kernel, dim3((p.n_edge + WPB - 1) / WPB, S), dim3(WB_BLOCK), 0, stream, p, it);
    const int use_lds = p.n_node <= 60 * 1024;
    hipLaunchKernelGGL(wb_walk_kernel, dim3(S), dim3(WB_WALK_BLOCK), use_lds ? (size_t)((p.n_node + 15) & ~15) : 0, stream, p, it,
                       (int)down_off.size() - 1, use_lds);
  } else {
    hipLaunchKernelGGL(wb_root_kernel, dim3((S + WPB - 1) / WPB), dim3(WB_BLOCK), 0, stream, p, it);
    for (size_t l = 0; l + 1 < down_off.size(); ++l) {
      const int cnt = down_off[l + 1] - down_off[l];
      if (cnt > 0) hipLaunchKernelGGL(wb_down_kernel, dim3((cnt + WPB - 1) / WPB, S), dim3(WB_BLOCK), 0, stream, p, it, down_off[l], down_off[l + 1]);
    }
  }
  const int b2_in_lds = (p.ell2_w == 0 && (int64_t)S * p.n_edge <= 32768) ? 1 : 0;      // a handful of chains: latency of the longest branch
  const size_t br_lds = b2_in_lds ? sizeof(double) * (size_t)p.n_states * (p.n_states | 1) : (size_t)p.n_states * p.ell2_w * 12;
  hipLaunchKernelGGL(wb_branch_kernel, dim3((p.n_edge + WPB - 1) / WPB, S), dim3(WB_BLOCK), br_lds, stream, p, it, b2_in_lds);
  hipLaunchKernelGGL(wb_stats_kernel, dim3(S, (unsigned)(p.n_states + 1)), dim3(256), 0, stream, p, it);
  if (p.reduce) {
    const int64_t items = (int64_t)p.n_tiles * p.n_cols;
    hipLaunchKernelGGL(wb_emit_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, p, it);
  }
  return hipGetLastError();
}

}  // namespace phm
