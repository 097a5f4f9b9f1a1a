// phm_narrow.hip -- branch-parallel MCMC sweep for few chains on a large tree (see phm_narrow.h).
// Same arithmetic as phm_mcmc.hip / the oracle: unfused left-to-right sums, the same categorical draw, the same Philox
// streams addressed by (replica, iteration, node | branch), the same deterministic log.
#include "phm_narrow.h"

namespace phm {

namespace {

// B^k applied to a child's partial-likelihood vector (mmmmvFORpl, src/phylomap.cpp:446-450).  Tips: the chain started from a
// one-hot row (or, ks, from a parity mask :1838-1845) is a table row; internal children: the chain itself.
template <int NS>
__device__ __forceinline__ void child_vec(const NarrowParams<NS>& p, const double* __restrict__ PLr,
                                          const uint8_t* __restrict__ tips, int child, int k, double (&v)[NS], uint32_t& err) {
  if (child < 0) {
    const int st = tips[~child];
    if (k >= p.klong) { err |= DERR_CAPACITY; k = p.klong - 1; }
    const double* src = (p.ks && p.tip_masks) ? p.maskL + ((size_t)k * 2 + (st & 1)) * NS : p.colL + ((size_t)k * NS + st) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) v[c] = src[c];
  } else {
#pragma unroll
    for (int c = 0; c < NS; ++c) v[c] = PLr[child * NS + c];
    for (int i = 0; i < k; ++i) matvec_u<NS>(p.Bc, v);
  }
}

// one internal node of one chain: PL[parent] = (B^(ma-1) PL[a]) (.) (B^(mb-1) PL[b])
template <int NS>
__device__ __forceinline__ void up_node(const NarrowParams<NS>& p, int r, int idx, uint32_t& err) {
  const UpStep st = p.up[p.up_order[idx]];
  const int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  double* PLr = p.PL + (size_t)r * p.n_node * NS;
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  double x[NS], y[NS];
  child_vec<NS>(p, PLr, tips, st.child[1], mc[st.edge[1]] - 1, x, err);        // "first"  (:508)
  child_vec<NS>(p, PLr, tips, st.child[0], mc[st.edge[0]] - 1, y, err);        // "second" (:509)
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
  for (int c = 0; c < NS; ++c) PLr[st.parent * NS + c] = x[c];
}

template <int NS>
__global__ __launch_bounds__(NARROW_BLOCK) void narrow_up_kernel(NarrowParams<NS> p, int begin, int end) {
  const int idx = begin + blockIdx.x * NARROW_BLOCK + threadIdx.x;
  if (idx >= end) return;
  uint32_t err = 0;
  up_node<NS>(p, blockIdx.y, idx, err);
  if (err) atomicOr(p.err, err);
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

// child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their parity mask (:1384-1397)
template <int NS>
__device__ __forceinline__ void down_edge(const NarrowParams<NS>& p, int r, int it, int idx, uint32_t& err) {
  const DownStep ds = p.down[p.down_order[idx]];
  const int b = ds.edge;
  const int m = p.mcount[(size_t)r * p.n_edge + b];
  uint8_t* __restrict__ nst = p.nstate + (size_t)r * p.n_node;
  const uint8_t* __restrict__ tips = p.tips_per_replica ? p.tips + (size_t)r * p.n_tips : p.tips;
  const int ps = nst[ds.parent];
  int cs;
  if (ds.child >= 0 || (p.ks && p.tip_masks)) {
    int kk = m - 1;
    if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
    const double* src = p.rowL + ((size_t)kk * NS + ps) * NS;
    double w[NS];
    uint32_t node_id;
    if (ds.child >= 0) {
      const double* __restrict__ PLc = p.PL + ((size_t)r * p.n_node + ds.child) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * PLc[c];
      node_id = (uint32_t)(ds.child + p.n_tips);
    } else {
      const int tip = ~ds.child;
      const int par = tips[tip] & 1;
#pragma unroll
      for (int c = 0; c < NS; ++c) w[c] = src[c] * (((c & 1) == par) ? 1.0 : 0.0);
      node_id = (uint32_t)tip;
    }
    const double u = stream_u(p.seed_lo, p.seed_hi, (uint32_t)(p.replica_offset + r), (uint32_t)it, ENT_NODE | node_id, 0);
    cs = sample_cat<NS>(w, u, err);                                            // :655
    if (ds.child >= 0) nst[ds.child] = (uint8_t)cs;
  } else {
    cs = tips[~ds.child];                                                      // :612
  }
  uint8_t* es = p.estate + ((size_t)r * p.n_edge + b) * 2;
  es[0] = (uint8_t)ps; es[1] = (uint8_t)cs;                                    // updatenodestates :460-475
}

template <int NS>
__global__ __launch_bounds__(NARROW_BLOCK) void narrow_down_kernel(NarrowParams<NS> p, int it, int begin, int end) {
  const int idx = begin + blockIdx.x * NARROW_BLOCK + threadIdx.x;
  if (idx >= end) return;
  uint32_t err = 0;
  down_edge<NS>(p, blockIdx.y, it, idx, err);
  if (err) atomicOr(p.err, err);
}

// The levels near the root hold a handful of nodes each; one workgroup per chain walks them in a single launch -- the top
// of the pruning sweep, the root draw, the first levels of the sampling sweep -- with a WORKGROUP-scope fence and a barrier
// between levels: producer and consumer are waves of one workgroup on one CU, and a device-scope fence would write back and
// invalidate the XCD's L2 at every level (the kernel boundary publishes the results to the other kernels).
constexpr int NARROW_MID_BLOCK = 256;
template <int NS>
__global__ __launch_bounds__(NARROW_MID_BLOCK) void narrow_mid_kernel(NarrowParams<NS> p, int it, int up_first, int up_levels,
                                                                      int down_levels) {
  const int r = blockIdx.x;
  uint32_t err = 0;
  for (int l = up_first; l < up_levels; ++l) {
    for (int idx = p.up_off[l] + threadIdx.x; idx < p.up_off[l + 1]; idx += NARROW_MID_BLOCK) up_node<NS>(p, r, idx, err);
    __threadfence_block();
    __syncthreads();
  }
  if (threadIdx.x == 0) root_node<NS>(p, r, it, err);
  __threadfence_block();
  __syncthreads();
  for (int l = 0; l < down_levels; ++l) {
    for (int idx = p.down_off[l] + threadIdx.x; idx < p.down_off[l + 1]; idx += NARROW_MID_BLOCK) down_edge<NS>(p, r, it, idx, err);
    __threadfence_block();
    __syncthreads();
  }
  if (err) atomicOr(p.err, err);
}

// One branch of one chain: resamplebranchstates :264-308, shortener :44-73 (shortenerbf :997-1030), virtual jumps
// sampleabranch :391-410, dwell sums updatedwelltimes :745-757.
template <int NS>
__global__ __launch_bounds__(NARROW_BLOCK) void narrow_branch_kernel(NarrowParams<NS> p, int it) {
  __shared__ double s_dw[NS * NARROW_BLOCK];
  __shared__ uint32_t s_cnt[NS * NS * NARROW_BLOCK];
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];        // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  const int lane = threadIdx.x;
  const int idx = blockIdx.x * NARROW_BLOCK + lane;
  const int r = blockIdx.y;
  for (int i = lane; i < 2 * PHM_LOGTAB_N; i += NARROW_BLOCK) s_ltab[i] = logtab_entry(i);
  __syncthreads();
  if (idx >= p.n_edge) return;
  const int b = p.branch_order[idx];
  const uint32_t rep = (uint32_t)(p.replica_offset + r);
  const bool KS = p.ks != 0;
  const int ncnt = KS ? NS * NS : NS * (NS - 1);
  int32_t* __restrict__ mc = p.mcount + (size_t)r * p.n_edge;
  const int m = mc[b];
  const uint8_t* es = p.estate + ((size_t)r * p.n_edge + b) * 2;
  const int ps = es[0], cs = es[1];
  const int64_t o = p.off[b];
  const int cap = (int)(p.off[b + 1] - o);
  const double* __restrict__ in = p.dw[it & 1] + (size_t)r * p.total_cap + o;
  double* __restrict__ out = p.dw[(it & 1) ^ 1] + (size_t)r * p.total_cap + o;
  double* __restrict__ ml = p.mlen + (size_t)r * p.total_cap + o;
  uint8_t* __restrict__ ms = p.mstate + (size_t)r * p.total_cap + o;
  uint32_t err = 0;
#pragma unroll
  for (int c = 0; c < NS; ++c) s_dw[c * NARROW_BLOCK + lane] = 0.0;
  for (int c = 0; c < NS * NS; ++c) s_cnt[c * NARROW_BLOCK + lane] = 0u;

  Stream su, se;
  su.open(ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
  se.open(ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);

  // pass A: states of the interior change points, s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end (:290, :301-304); equal
  // neighbours merged (:54), merged lengths and states written to the scratch slots.  The input lengths and the table
  // rows do not depend on the states drawn so far, so they are fetched eight steps at a time ahead of the dependent chain.
  constexpr int CH = 8;
  int w = 0;
  int cur_s = (m == 1) ? cs : ps;                    // updatenodestates :469-472 (m == 1: the child end wins)
  double cur_len = in[0];
  for (int i0 = 1; i0 < m; i0 += CH) {
    double dbuf[CH], bbuf[CH][NS];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int i = i0 + q;
      dbuf[q] = (i < m) ? in[i] : 0.0;
      int kk = m - i - 1;
      if (kk >= p.klong) { err |= DERR_CAPACITY; kk = p.klong - 1; }
      const double* beta = p.colL + ((size_t)(kk > 0 ? kk : 0) * NS + cs) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) bbuf[q][c] = (i < m - 1) ? beta[c] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int i = i0 + q;
      if (i < m) {
        int si;
        if (i == m - 1) si = cs;
        else {
          double pr[NS];
#pragma unroll
          for (int c = 0; c < NS; ++c) pr[c] = p.B2[cur_s * NS + c] * bbuf[q][c];
          si = sample_cat<NS>(pr, su.draw((uint32_t)(i - 1)), err);
        }
        const double di = dbuf[q];
        if (KS) s_cnt[(cur_s * NS + si) * NARROW_BLOCK + lane] += 1u;            // shortenerbf :1010-1014
        if (si == cur_s) cur_len = cur_len + di;
        else {
          ml[w] = cur_len; ms[w] = (uint8_t)cur_s;
          if (!KS) s_cnt[(cur_s * (NS - 1) + (si > cur_s ? si - 1 : si)) * NARROW_BLOCK + lane] += 1u;   // shortener :65-66
          ++w; cur_s = si; cur_len = di;
        }
      }
    }
  }
  ml[w] = cur_len; ms[w] = (uint8_t)cur_s;
  const int nmerged = w + 1;

  // pass B: virtual jumps, gaps ~ Exp(Omega + q_ss) until each merged segment is used up (:391-410); a segment that is not
  // positive leaves itself and everything after it untouched (the reference's iterators stop advancing, :397, :405-406)
  int mnew = 0;
  uint32_t edraw = 0;
  bool stuck = false;
  for (int j = 0; j < nmerged; ++j) {
    const int s = ms[j];
    const double len = ml[j];
    double acc = s_dw[s * NARROW_BLOCK + lane];
    if (stuck || !(0.0 < len)) {
      stuck = true;
      if (mnew < cap) out[mnew] = len; else err |= DERR_CAPACITY;
      acc += len;
      ++mnew;
    } else {
      const double scale = p.scale[s];
      double tot = 0.0;
      while (tot < len) {
        const double rl = scale * neglog_u32(se.draw_word(edraw++), s_ltab);      // :398
        double piece;
        if ((tot + rl) < len) { piece = rl; tot += rl; }
        else { piece = len - tot; tot = len; }
        if (mnew < cap) out[mnew] = piece; else err |= DERR_CAPACITY;
        acc += piece;                                                          // updatedwelltimes :752
        ++mnew;
      }
    }
    s_dw[s * NARROW_BLOCK + lane] = acc;
  }
  if (mnew > cap) mnew = cap;
  mc[b] = mnew;

  double* part = p.part + ((size_t)r * p.n_edge + b) * (NS + NS * NS + 1);
#pragma unroll
  for (int c = 0; c < NS; ++c) part[c] = s_dw[c * NARROW_BLOCK + lane];
  for (int c = 0; c < ncnt; ++c) part[NS + c] = (double)s_cnt[c * NARROW_BLOCK + lane];
  part[NS + NS * NS] = (double)(m + mnew);           // segments read + written (one global counter would serialise every lane)
  if (err) atomicOr(p.err, err);
}

// Statistics row of one chain: every column is the sum of the per-branch values, added in a fixed order (thread t takes
// branches t, t+256, ... in edge order, then a fixed tree over the 256 partial sums) -> identical from run to run.
// One workgroup per (chain, column): the columns of a row are reduced side by side (blockIdx.y), not one after the other.
template <int NS>
__global__ __launch_bounds__(256) void narrow_stats_kernel(NarrowParams<NS> p) {
  __shared__ double red[256];
  const int r = blockIdx.x;
  const int c = blockIdx.y;                          // 0 .. NS + ncnt; the last one adds the segment counts (column pc - 1)
  const int ncnt = p.ks ? NS * NS : NS * (NS - 1);
  const int pc = NS + NS * NS + 1;
  const double* part = p.part + (size_t)r * p.n_edge * pc;
  const int src_c = (c == NS + ncnt) ? pc - 1 : c;
  double s = 0.0;
  for (int e = threadIdx.x; e < p.n_edge; e += 256) s += part[(size_t)e * pc + src_c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + half];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (c < NS + ncnt) p.rowbuf[(size_t)r * p.n_cols + c] = red[0];
    else atomicAdd(p.segcnt, (unsigned long long)red[0]);
    if (p.ks && c == 0)                                                        // root state, 0-based (:1350-1352)
      p.rowbuf[(size_t)r * p.n_cols + NS + ncnt] = (double)p.nstate[(size_t)r * p.n_node + p.root];
  }
}

// rows -> the engine's statistics layout (per replica, or summed over each 64-replica tile in replica order)
template <int NS>
__global__ void narrow_emit_kernel(NarrowParams<NS> p, int it) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (!p.reduce) {
    if (gid >= p.n_rep * p.n_cols) return;
    const int r = gid / p.n_cols, c = gid % p.n_cols;
    p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + r] = p.rowbuf[(size_t)r * p.n_cols + c];
  } else {
    if (gid >= p.n_tiles * p.n_cols) return;
    const int tile = gid / p.n_cols, c = gid % p.n_cols;
    double s = 0.0;
    for (int r = tile * 64; r < tile * 64 + 64 && r < p.n_rep; ++r) s += p.rowbuf[(size_t)r * p.n_cols + c];
    p.stats[((size_t)it * p.n_tiles + tile) * p.n_cols + c] = s;
  }
}

}  // namespace

template <int NS>
hipError_t launch_narrow_sweep(const NarrowParams<NS>& p, const std::vector<int32_t>& up_off,
                               const std::vector<int32_t>& down_off, int it, hipStream_t stream) {
  const unsigned S = (unsigned)p.n_rep;
  // big levels: one launch each; the small levels around the root: one launch for all of them
  const int UL = (int)up_off.size() - 1, DL = (int)down_off.size() - 1;
  int up_first = UL, down_levels = 0;
  while (up_first > 0 && up_off[up_first] - up_off[up_first - 1] <= NARROW_MID_BLOCK) --up_first;
  while (down_levels < DL && down_off[down_levels + 1] - down_off[down_levels] <= NARROW_MID_BLOCK) ++down_levels;
  for (int l = 0; l < up_first; ++l) {
    const int n = up_off[l + 1] - up_off[l];
    if (n <= 0) continue;
    hipLaunchKernelGGL(narrow_up_kernel<NS>, dim3((n + NARROW_BLOCK - 1) / NARROW_BLOCK, S), dim3(NARROW_BLOCK), 0, stream, p,
                       up_off[l], up_off[l + 1]);
  }
  hipLaunchKernelGGL(narrow_mid_kernel<NS>, dim3(S), dim3(NARROW_MID_BLOCK), 0, stream, p, it, up_first, UL, down_levels);
  for (int l = down_levels; l < DL; ++l) {
    const int n = down_off[l + 1] - down_off[l];
    if (n <= 0) continue;
    hipLaunchKernelGGL(narrow_down_kernel<NS>, dim3((n + NARROW_BLOCK - 1) / NARROW_BLOCK, S), dim3(NARROW_BLOCK), 0, stream, p,
                       it, down_off[l], down_off[l + 1]);
  }
  hipLaunchKernelGGL(narrow_branch_kernel<NS>, dim3((p.n_edge + NARROW_BLOCK - 1) / NARROW_BLOCK, S), dim3(NARROW_BLOCK), 0,
                     stream, p, it);
  hipLaunchKernelGGL(narrow_stats_kernel<NS>, dim3(S, NS + (p.ks ? NS * NS : NS * (NS - 1)) + 1), dim3(256), 0, stream, p);
  const int items = (p.reduce ? p.n_tiles : p.n_rep) * p.n_cols;
  hipLaunchKernelGGL(narrow_emit_kernel<NS>, dim3((items + 255) / 256), dim3(256), 0, stream, p, it);
  return hipGetLastError();
}

template hipError_t launch_narrow_sweep<2>(const NarrowParams<2>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t);
template hipError_t launch_narrow_sweep<3>(const NarrowParams<3>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t);
template hipError_t launch_narrow_sweep<4>(const NarrowParams<4>&, const std::vector<int32_t>&, const std::vector<int32_t>&, int, hipStream_t);

}  // namespace phm
