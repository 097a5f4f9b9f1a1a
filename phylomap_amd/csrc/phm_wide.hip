// phm_wide.hip -- the fixed-Q MCMC sweep for 5 <= n <= 64 states (C4: dense 61-state Q, C5: sparse 20-state Q).
//
// Same per-replica algorithm, HBM streams and RNG keys as phm_mcmc.hip; what changes is who does the n-vector
// arithmetic.  A wavefront still owns 64 replicas and the scalar work (segment bookkeeping, merging, virtual
// jumps, Philox, log) still runs one replica per LANE.  The n-vector work of a replica -- B^k chains,
// PL products, row normalisation, categorical draws -- is done by the whole wave with one STATE per lane,
// the 64 replicas taking turns (their scalars are broadcast with v_readlane, results handed back to the owning
// lane).  Summation orders are the spec's left-to-right ones, so results equal the oracle bit for bit:
//   y_c = ((B_c0 x_0 + B_c1 x_1) + ...)   lane c accumulates while x_j is broadcast, j ascending;
//   prefix sums of a probability vector are formed in index order by the same broadcast loop.
// B (row-major and transposed) and the dense forward rows are staged in LDS; the B^k e_j tables
// (k < WIDE_KTAB) stay in global memory / L2, read coalesced ([k][j][lane]).
#include "phm_wide.h"

#include "phm_coop.h"

namespace phm {

// wave-uniform uniform draw d of stream (rep, it, ent)
__device__ __forceinline__ double uni_draw(const WideParams& p, uint32_t rep, uint32_t it, uint32_t ent, uint32_t d) {
  return stream_u(p.seed_lo, p.seed_hi, rep, it, ent, d);
}

__global__ __launch_bounds__(WIDE_BLOCK) void mcmc_wide_kernel(WideParams p, int iter0, int n_iters) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = p.n_states;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * (WIDE_BLOCK / 64) + wave;
  const int c = lane < n ? lane : n - 1;       // clamped state index of this lane

  const int ldn = n | 1;                              // odd row stride: conflict-free column AND row access
  double* s_Bc = reinterpret_cast<double*>(smem);     // [r][ldn] chain matrix (v <- Bc v and w <- Bc^T w)
  double* s_B2 = p.sparse ? s_Bc + n * ldn : s_Bc;    // [r][ldn] dense B rows for the forward step (= Bc unless SPARSE)
  double* s_scale = s_Bc + (p.sparse ? 2 : 1) * n * ldn;   // [n]
  const int maxseg = n > 40 ? WIDE_MAXSEG_BIG_N : WIDE_MAXSEG;
  uint8_t* s_st = reinterpret_cast<uint8_t*>(s_scale + n) + (size_t)wave * maxseg * 64;   // [slot][replica]
  for (int i = threadIdx.x; i < n * n; i += WIDE_BLOCK) {
    int r = i / n, cc = i - r * n;
    s_Bc[r * ldn + cc] = p.Bc[i];
    if (p.sparse) s_B2[r * ldn + cc] = p.B2[i];
  }
  if ((int)threadIdx.x < n) s_scale[threadIdx.x] = p.scale[threadIdx.x];
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];        // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += WIDE_BLOCK) s_ltab[i] = logtab_entry(i);
  __syncthreads();
  if (tile >= p.n_tiles) return;

  const int rep_local = tile * 64 + lane;                 // position in the padded statistics layout
  // list of trees (maketreelistMCMCmt :2267): consecutive groups of tiles walk different topologies with one model
  const int tree = p.tiles_per_tree ? tile / p.tiles_per_tree : 0;
  // Replicas of this tile that exist.  The n-vector work below takes the replicas of a tile in turn and skips the padding
  // lanes; with few replicas the host therefore spreads them thinly (rep_stride < 64 replicas per tile) so that many waves
  // share the work.  Padding lanes carry a one-segment dummy path through the scalar part.
  const int stride = p.tiles_per_tree ? 64 : p.rep_stride;
  const int nr = min(stride, p.n_rep - (tile - tree * p.tiles_per_tree) * stride);
  const bool valid = lane < nr;
  const uint32_t rep0 = (uint32_t)(p.replica_offset + tile * stride);       // Philox replica word = logical replica id
  const uint32_t rep = rep0 + (uint32_t)lane;
  const UpStep* __restrict__ up = p.up + (size_t)tree * p.n_node;
  const DownStep* __restrict__ down = p.down + (size_t)tree * p.n_edge;
  const int root = p.tiles_per_tree ? p.roots[tree] : p.root;
  uint32_t err = 0;
  const double pid_c = p.pid[c];

  double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * 64 * n;      // [node][replica][n]
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;

  // B^k applied to the PL vector of `child` for replica r (lanes = states): tips via the chain table
  auto child_vec = [&](int child, int k, int r, int tipstate_r) -> double {
    double v;
    int done;
    if (child < 0) {
      int kt = k < p.ktab ? k : p.ktab - 1;
      v = p.tip_masks ? p.maskpow[((size_t)kt * 2 + (tipstate_r & 1)) * n + c] : p.colpow[((size_t)kt * n + tipstate_r) * n + c];
      done = kt;
    } else {
      v = PLt[((size_t)child * 64 + r) * n + c];
      done = 0;
    }
    for (int i = done; i < k; ++i) v = coop_matvec(s_Bc, v, n, ldn, c);
    return v;
  };

  double* ring = p.dwell0 + (size_t)tile * p.rows * 64;
  const int C = (int)p.rows;
  int cur_r = p.cursor[tile * 2], cur_w = p.cursor[tile * 2 + 1];
  for (int it = iter0; it < iter0 + n_iters; ++it) {
    const int rbase = cur_r, wbase = cur_w;              // one ring per tile, as in phm_mcmc.hip
    const int r_in = (wbase >= rbase) ? wbase - rbase : wbase - rbase + C;
    auto IN = [&](int k) -> double& { int idx = rbase + k; idx = idx >= C ? idx - C : idx; return ring[idx * 64 + lane]; };
    auto OUT = [&](int k) -> double& { int idx = wbase + k; idx = idx >= C ? idx - C : idx; return ring[idx * 64 + lane]; };
    // statistics: reduce = 0 -> [iter][col][replica], plain read-modify-write by the owning lane;
    //             reduce = 1 -> [iter][tile][col] shared by the 64 lanes of the wave, f64 atomics (counts are exact
    //             in any order; dwell sums agree to rounding) -- n(n-1) columns per replica would not fit for n = 61
    const bool red = p.reduce != 0;
    double* srow = red ? p.stats + ((size_t)it * p.n_tiles + tile) * p.n_cols
                       : p.stats + (size_t)it * p.n_cols * p.n_rep_pad + rep_local;
    const size_t cstride = red ? 1 : (size_t)p.n_rep_pad;
    auto stat_add = [&](int col, double v) {
      if (red) { if (valid) atomicAdd(srow + col, v); }
      else srow[(size_t)col * cstride] += v;
    };
    uint32_t seg_rw = 0;
    int in_row = 0, out_row = 0;

    // ------------------------------ up sweep ------------------------------
    for (int k = 0; k < p.n_node; ++k) {
      const UpStep st = up[k];
      const int ma = mct[st.edge[0] * 64 + lane];
      const int mb = mct[st.edge[1] * 64 + lane];
      int ta = 0, tb = 0;
      if (st.child[0] < 0) ta = p.tips_per_replica ? tips_t[(~st.child[0]) * 64 + lane] : p.tips[~st.child[0]];
      if (st.child[1] < 0) tb = p.tips_per_replica ? tips_t[(~st.child[1]) * 64 + lane] : p.tips[~st.child[1]];
      for (int r = 0; r < nr; ++r) {
        const int mar = __builtin_amdgcn_readlane(ma, r), mbr = __builtin_amdgcn_readlane(mb, r);
        const int tar = __builtin_amdgcn_readlane(ta, r), tbr = __builtin_amdgcn_readlane(tb, r);
        double x = child_vec(st.child[1], mbr - 1, r, tbr);        // "first"  (:508)
        double y = child_vec(st.child[0], mar - 1, r, tar);        // "second" (:509)
        x = x * y;                                                  // :510
        if (p.normalise) x = x / coop_sum(x, n);                    // :525
        if (lane < n) PLt[((size_t)st.parent * 64 + r) * n + lane] = x;
      }
    }

    // ------------------------------ root ------------------------------
    {
      int mine = 0;
      for (int r = 0; r < nr; ++r) {
        double pr = (lane < n) ? pid_c * PLt[((size_t)root * 64 + r) * n + c] : 0.0;     // :618
        double u = uni_draw(p, rep0 + r, (uint32_t)it, ENT_NODE | (uint32_t)(root + p.n_tips), 0);
        int rs = coop_sample(pr, u, n, lane, err);                                         // :627
        if (lane == r) mine = rs;
      }
      nst[root * 64 + lane] = (uint8_t)mine;
      if (p.ks) stat_add(n + n * n, (double)mine);                                           // :1350-1352
    }

    // ------------------------------ down sweep ------------------------------
    for (int k = 0; k < p.n_edge; ++k) {
      const DownStep ds = down[k];
      const int b = ds.edge;
      int m = valid ? (int)mct[b * 64 + lane] : 1;
      const int ps = nst[ds.parent * 64 + lane];
      int mmax = wave_max_w(m);
      int cs = 0;
      if (ds.child < 0) cs = p.tips_per_replica ? tips_t[(~ds.child) * 64 + lane] : p.tips[~ds.child];
      const int tobs = cs;                         // observed tip state (ks keeps only its parity)
      if (mmax > maxseg) {           // LDS state scratch is `maxseg` slots per replica: report, stay memory-safe
        err |= DERR_CAPACITY;
        m = m < maxseg ? m : maxseg;
        mmax = maxseg;
      }

      // ---- n-vector work, replicas in turn: child state, then the interior states of the branch ----
      for (int r = 0; r < nr; ++r) {
        const int mr = __builtin_amdgcn_readlane(m, r);
        const int psr = __builtin_amdgcn_readlane(ps, r);
        int csr;
        if (ds.child >= 0 || p.tip_masks) {
          // child ~ e_ps^T B^(m-1) (.) PL[child]     (Tvmmp :431-436, :651-655)
          int kk = mr - 1;
          int kt = kk < p.ktab ? kk : p.ktab - 1;
          double w = p.rowpow[((size_t)kt * n + psr) * n + c];
          for (int i = kt; i < kk; ++i) {
            double acc = s_Bc[c] * readlane_f64(w, 0);
            for (int q = 1; q < n; ++q) acc += s_Bc[q * ldn + c] * readlane_f64(w, q);
            w = acc;
          }
          uint32_t node_id;
          if (ds.child >= 0) {
            w = (lane < n) ? w * PLt[((size_t)ds.child * 64 + r) * n + c] : 0.0;
            node_id = (uint32_t)(ds.child + p.n_tips);
          } else {                                  // ks: hidden tip state against the parity mask (:1384-1397)
            const int par = __builtin_amdgcn_readlane(tobs, r) & 1;
            w = (lane < n) ? w * (((c & 1) == par) ? 1.0 : 0.0) : 0.0;
            node_id = (uint32_t)(~ds.child);
          }
          double u = uni_draw(p, rep0 + r, (uint32_t)it, ENT_NODE | node_id, 0);
          csr = coop_sample(w, u, n, lane, err);
          if (lane == r) cs = csr;
        } else {
          csr = __builtin_amdgcn_readlane(cs, r);
        }
        // interior states s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end     (resamplebranchstates :290, :301-304)
        int prev = psr;
        for (int i = 1; i < mr - 1; ++i) {
          int kk = mr - i - 1;
          int kt = kk < p.ktab ? kk : p.ktab - 1;
          double beta = p.colpow[((size_t)kt * n + csr) * n + c];
          for (int q = kt; q < kk; ++q) beta = coop_matvec(s_Bc, beta, n, ldn, c);
          double pr = (lane < n) ? s_B2[prev * ldn + c] * beta : 0.0;
          double u = uni_draw(p, rep0 + r, (uint32_t)it, ENT_BSTATE | (uint32_t)b, (uint32_t)(i - 1));
          int si = coop_sample(pr, u, n, lane, err);
          if (lane == 0) s_st[(i - 1) * 64 + r] = (uint8_t)si;
          prev = si;
        }
      }
      if (ds.child >= 0) nst[ds.child * 64 + lane] = (uint8_t)cs;

      // ---- scalar work, one replica per lane: merge, count, virtual jumps, dwell sums ----
      Stream se;
      se.open(ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
      const int roff = in_row, woff = out_row;
      const int cap = C - (r_in - in_row) - out_row;
      int mnew = 0;
      {
        // pass A: merged segments written back in place (lengths into the consumed input rows, states into s_st)
        int w = 0;
        int cur_s = (m == 1) ? cs : ps;                              // updatenodestates :469-472
        double cur_len = IN(roff);
        double dnext = (m > 1) ? IN(roff + 1) : 0.0;
        for (int i = 1; i < mmax; ++i) {
          if (i < m) {
            int si = (i == m - 1) ? cs : (int)s_st[(i - 1) * 64 + lane];
            double di = dnext;
            if (i + 1 < m) dnext = IN(roff + i + 1);
            if (p.count_self) stat_add(n + cur_s * n + si, 1.0);                            // shortenerbf :1010-1014
            if (si == cur_s) cur_len = cur_len + di;                 // shortener :54
            else {
              IN(roff + w) = cur_len;
              s_st[w * 64 + lane] = (uint8_t)cur_s;                  // w <= i-1: slot already consumed
              if (!p.count_self) {
                int col = cur_s * (n - 1) + (si > cur_s ? si - 1 : si);                        // shortener :65-66
                stat_add(n + col, 1.0);
              }
              ++w; cur_s = si; cur_len = di;
            }
          }
        }
        const int nmerged = w + 1;
        const double len0 = (w == 0) ? cur_len : IN(roff);
        const int s0 = (w == 0) ? cur_s : (int)s_st[lane];
        if (w > 0) { IN(roff + w) = cur_len; s_st[w * 64 + lane] = (uint8_t)cur_s; }

        // pass B: one new piece per step per lane (virtual jumps :391-410, updatedwelltimes :745-757)
        int j = 0, s = s0;
        double len = len0;
        double lnext = (nmerged > 1) ? ((w == 1) ? cur_len : IN(roff + 1)) : 0.0;
        double tot = 0.0, scale = s_scale[s];
        // running dwell sum of state s: continued piece by piece from the stored total (the reference's order,
        // :752) when statistics are per replica; a fresh partial handed to one atomic when they are summed
        double acc = red ? 0.0 : srow[(size_t)s * cstride];
        uint32_t edraw = 0;
        bool stuck = !valid, done = false;             // padding lanes: one piece, no draws
        while (!done) {
          double piece;
          bool adv;
          if (stuck || !(0.0 < len)) { stuck = true; piece = len; adv = true; }
          else {
            double rl = scale * neglog_u32(se.draw_word(edraw++), s_ltab);   // :398
            if ((tot + rl) < len) { piece = rl; tot += rl; adv = false; }
            else { piece = len - tot; adv = true; }
          }
          if (mnew < cap) OUT(woff + mnew) = piece; else err |= DERR_CAPACITY;
          acc += piece;
          ++mnew;
          if (adv) {
            if (red) { if (valid) atomicAdd(srow + s, acc); } else srow[(size_t)s * cstride] = acc;
            ++j;
            if (j >= nmerged) done = true;
            else {
              len = lnext;
              if (j + 1 < nmerged) lnext = IN(roff + j + 1);
              s = (int)s_st[j * 64 + lane];
              scale = s_scale[s]; tot = 0.0;
              acc = red ? 0.0 : srow[(size_t)s * cstride];
            }
          }
        }
      }
      if (mnew > 65535) { err |= DERR_CAPACITY; mnew = 65535; }
      mct[b * 64 + lane] = (uint16_t)mnew;
      seg_rw += (uint32_t)(m + mnew);
      in_row += mmax;
      out_row += wave_max_w(mnew);
      if (out_row > C) out_row = C;
    }
    cur_r = wbase;
    cur_w = wbase + out_row; if (cur_w >= C) cur_w -= C;
    {
      uint32_t v = valid ? seg_rw : 0u;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) atomicAdd(p.segcnt, (unsigned long long)v);
    }
  }
  if (lane == 0) { p.cursor[tile * 2] = cur_r; p.cursor[tile * 2 + 1] = cur_w; }
  if (err) atomicOr(p.err, err);
}

size_t wide_lds_bytes(int n, bool sparse) {
  return sizeof(double) * ((size_t)(sparse ? 2 : 1) * n * (n | 1) + n) + (size_t)(WIDE_BLOCK / 64) * wide_maxseg(n) * 64;
}

hipError_t launch_mcmc_wide(const WideParams& p, int iter0, int n_iters, hipStream_t stream) {
  const int wpb = WIDE_BLOCK / 64;
  size_t lds = wide_lds_bytes(p.n_states, p.sparse != 0);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mcmc_wide_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(mcmc_wide_kernel, dim3((p.n_tiles + wpb - 1) / wpb), dim3(WIDE_BLOCK), lds, stream, p, iter0, n_iters);
  return hipGetLastError();
}

}  // namespace phm
