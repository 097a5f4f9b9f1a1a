// phm_mcmc.hip -- the fixed-Q MCMC sweep (sumstatMCMC / sumstatMCMC_bigtree / SPARSEsumstatMCMC)
// as one CDNA4 kernel.
//
// Mapping (DESIGN.md "Kernel K-MCMC"): one LANE owns one replica (an independent chain / site); the 64
// lanes of a wavefront walk the SAME tree in lock step, so
//   * the topology (schedules) is wave-uniform -> scalar loads, no divergence from the tree shape;
//   * every per-replica array is laid out [entity][state|slot][64 lanes] -> each access is one fully
//     coalesced 512-B (f64) row of HBM;
//   * no inter-lane or inter-workgroup communication exists: a replica never leaves its lane, so the
//     whole N-loop runs on the device without grid synchronisation.
// Q-derived tables (B, B^k e_j chains, rexp scales) are staged in LDS once per workgroup.
//
// What one iteration does for a replica (reference call stack: treesample, src/phylomap.cpp:775-785):
//   up   : Felsenstein pruning with B^(m_b-1)           makePLrcpp :503-514 (_bigtree :516-529)
//   root : root ~ pid (.) PL[root]                       sampleinternalnodesMCMC :618-627
//   down : per branch in pre-order: child ~ row_ps(B^(m-1)) (.) PL[child]   :640-657
//          write endpoints                               updatenodestates :460-475
//          resample interior states                      resamplebranchstates :264-308
//          drop self transitions, count transitions      shortener :44-73
//          re-insert virtual jumps ~ Exp(Omega+q_ss)     sampleabranch :391-410
//          accumulate dwell per state                    updatedwelltimes :745-757
#include "phm_mcmc.h"

namespace phm {

// B^k applied to a child's partial-likelihood vector (mmmmvFORpl, src/phylomap.cpp:446-450)
template <int NS, bool KS>
__device__ __forceinline__ void child_vector(const McmcParams<NS>& p, const double* __restrict__ s_col,
                                             const double* __restrict__ s_mask, const double* __restrict__ PLt,
                                             const uint8_t* __restrict__ tips_t, int child, int k, int lane,
                                             double (&v)[NS]) {
  if (child < 0) {
    // tip: the chain started from the tip's PL row is a table entry (bit-identical to running it):
    // one-hot row -> column `state` of B^k; ks: parity mask row (:1838-1845) -> B^k applied to that mask
    int tip = ~child;
    int st = p.tips_per_replica ? at(tips_t, (uint32_t)tip * 64u + (uint32_t)lane) : p.tips[tip];
    // chains of up to ktab - 1 steps from the LDS copy; longer ones from the full-length table in global memory (L2), whose
    // rows are the same chain run on the host; only beyond that table (never, by construction of klong) is the chain continued
    // The host sizes the full-length table past every count a branch can hold (klong > capacity) and the sweep never stores
    // a count above the capacity, so k < klong always.
    const int kt = k;
    const int kl = kt < MCMC_KTAB ? kt : MCMC_KTAB - 1;
    {                                        // always an LDS read (ds_read); the rare long chain overwrites it from global
      const double* src = (KS && p.tip_masks) ? s_mask + (kl * 2 + (st & 1)) * NS : s_col + (kl * NS + st) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) v[c] = src[c];
    }
    if (kt >= MCMC_KTAB) {
      const double* __restrict__ src = (KS && p.tip_masks) ? p.maskpow + ((size_t)kt * 2 + (st & 1)) * NS : p.colpow + ((size_t)kt * NS + st) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) v[c] = src[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NS; ++c) v[c] = at(PLt, (uint32_t)(child * NS + c) * 512u + (uint32_t)lane * 8u);
    for (int i = 0; i < k; ++i) matvec_u<NS>(p.Bc, v);
  }
}

// KS = the tree sweep of sumstatMCMCks (treesampleks, src/phylomap.cpp:1422-1432) with Q held fixed: parity tip masks,
// hidden tip states re-sampled every sweep (:1384-1397), every consecutive state pair counted, self pairs included,
// into n x n counters (shortenerbf :1010-1014), root state recorded (:1350-1352).
// RING = one ring of C rows per tile holds both dwell streams (half the HBM, two extra VALU ops per access);
// !RING = two buffers of C rows, swapped every sweep (chosen by the host when HBM is plentiful).
// MT = a list of trees (phm_engine_create_multi): the tile picks its topology; kept out of the single-tree kernels, whose
// scalar registers are fully used.
template <int NS, bool KS, bool RING, bool MT = false>
__device__ __forceinline__ void sweep_body(const McmcParams<NS>& p, int iter0, int n_iters) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: per-tile bases live in scalar registers
  const int tile = blockIdx.x * (MCMC_BLOCK / 64) + wave;
  const uint32_t lane8 = (uint32_t)lane * 8u;
  constexpr int NCNT = KS ? NS * NS : NS * (NS - 1);

  // ---- LDS carve-up: tables shared by the workgroup, accumulators private to each lane ----
  double* s_col = reinterpret_cast<double*>(smem);            // [ktab][NS][NS]  B^k e_j      (column chains)
  double* s_B2 = s_col + MCMC_KTAB * NS * NS;                    // [NS][NS] dense B, rows for the forward step
  double* s_scale = s_B2 + NS * NS;                           // [NS] 1/(Omega+q_ss)
  double* s_dw = s_scale + NS + (size_t)wave * NS * 64;       // [NS][64] dwell accumulators of this wave
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_scale + NS + (size_t)(MCMC_BLOCK / 64) * NS * 64) +
                    (size_t)wave * NCNT * 64;                 // [NCNT][64] transition counters of this wave
  double* s_mask = reinterpret_cast<double*>(reinterpret_cast<uint32_t*>(s_scale + NS + (size_t)(MCMC_BLOCK / 64) * NS * 64) +
                                             (size_t)(MCMC_BLOCK / 64) * NS * NS * 64);   // [ktab][2][NS] (ks only)
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];                 // (1/c_j, log c_j) of the exponential variates (neglog_u32)
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += MCMC_BLOCK) s_ltab[i] = logtab_entry(i);
  for (int i = threadIdx.x; i < MCMC_KTAB * NS * NS; i += MCMC_BLOCK) s_col[i] = p.colpow[i];      // the host passes ktab = MCMC_KTAB
  if (KS && p.tip_masks) for (int i = threadIdx.x; i < MCMC_KTAB * 2 * NS; i += MCMC_BLOCK) s_mask[i] = p.maskpow[i];
  if (threadIdx.x < NS * NS) s_B2[threadIdx.x] = p.B2[threadIdx.x];
  if (threadIdx.x < NS) s_scale[threadIdx.x] = p.scale[threadIdx.x];
  __syncthreads();
  if (tile >= p.n_tiles) return;     // whole waves only; no barrier below this line

  const int rep_local = tile * 64 + lane;
  const uint32_t rep = (uint32_t)(p.replica_offset + rep_local);
  // list of trees (maketreelistMCMCmt :2267): consecutive groups of tiles walk different topologies with one model
  const int tree = MT ? tile / p.tiles_per_tree : 0;
  const bool valid = MT ? (rep_local - tree * p.tiles_per_tree * 64) < p.n_rep : rep_local < p.n_rep;
  const UpStep* __restrict__ up = MT ? p.up + (size_t)tree * p.n_node : p.up;
  const DownStep* __restrict__ down = MT ? p.down + (size_t)tree * p.n_edge : p.down;
  const int root = MT ? p.roots[tree] : p.root;
  uint32_t err = 0;

  double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * NS * 64;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  uint16_t* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  const uint8_t* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;

  double* ring = p.dwell0 + (size_t)tile * p.rows * 64;
  double* ring2 = RING ? ring : p.dwell1 + (size_t)tile * p.rows * 64;
  const int C = (int)p.rows;
  int cur_r = p.cursor[tile * 2], cur_w = p.cursor[tile * 2 + 1];      // wave-uniform: start of the current stream, end of it
  for (int it = iter0; it < iter0 + n_iters; ++it) {
    // One ring of C rows per tile holds the stream being consumed and, behind it, the stream being produced: the
    // producer starts where the consumer's data end and may wrap into rows the consumer has already freed.
    const int rbase = cur_r, wbase = cur_w;
    const int r_in = (wbase >= rbase) ? wbase - rbase : wbase - rbase + C;      // rows of the input stream
    double* buf_in = RING ? ring : (cur_r ? ring2 : ring);                      // !RING: cur_r is the sweep parity
    double* buf_out = RING ? ring : (cur_r ? ring : ring2);
    auto IN = [&](int k) -> double& {
      int idx = k;
      if (RING) { idx = rbase + k; idx = idx >= C ? idx - C : idx; }
      return at(buf_in, (uint32_t)idx * 512u + lane8);
    };
    auto OUT = [&](int k) -> double& {
      int idx = k;
      if (RING) { idx = wbase + k; idx = idx >= C ? idx - C : idx; }
      return at(buf_out, (uint32_t)idx * 512u + lane8);
    };
#pragma unroll
    for (int c = 0; c < NS; ++c) s_dw[c * 64 + lane] = 0.0;
#pragma unroll
    for (int c = 0; c < NCNT; ++c) s_cnt[c * 64 + lane] = 0u;
    uint32_t seg_rw = 0;
    int in_row = 0, out_row = 0;     // wave-uniform cursors into the sequential dwell streams

    // ------------------------------ up sweep: partial likelihoods ------------------------------
    for (int k = 0; k < p.n_node; ++k) {
      const UpStep st = up[k];
      double x[NS], y[NS];
      int ma = at(mct, (uint32_t)st.edge[0] * 128u + (uint32_t)lane * 2u);
      int mb = at(mct, (uint32_t)st.edge[1] * 128u + (uint32_t)lane * 2u);
      child_vector<NS, KS>(p, s_col, s_mask, PLt, tips_t, st.child[1], mb - 1, lane, x);   // "first"  (:508)
      child_vector<NS, KS>(p, s_col, s_mask, PLt, tips_t, st.child[0], ma - 1, lane, y);   // "second" (:509)
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = x[c] * y[c];                         // :510
      if (p.normalise) {                                                       // :525
        double s = x[0];
#pragma unroll
        for (int c = 1; c < NS; ++c) s += x[c];
#pragma unroll
        for (int c = 0; c < NS; ++c) x[c] = x[c] / s;
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) at(PLt, (uint32_t)(st.parent * NS + c) * 512u + (uint32_t)lane * 8u) = x[c];
    }

    if (p.prune_only) continue;      // wave-uniform: the pruning sweep alone (bench.py "pruning" roofline)

    // ------------------------------ root state ------------------------------
    int my_root = 0;
    {
      double pr[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = p.pid[c] * at(PLt, (uint32_t)(root * NS + c) * 512u + (uint32_t)lane * 8u);   // :618
      double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | (uint32_t)(root + p.n_tips), 0);
      my_root = sample_cat<NS>(pr, u, err);                                                 // :627
      at(nst, (uint32_t)root * 64u + (uint32_t)lane) = (uint8_t)my_root;
    }

    // ------------------------------ down sweep: node states + branch paths ------------------------------
    for (int k = 0; k < p.n_edge; ++k) {
      const DownStep ds = down[k];
      const int b = ds.edge;
      const int m = at(mct, (uint32_t)b * 128u + (uint32_t)lane * 2u);
      const int ps = at(nst, (uint32_t)ds.parent * 64u + (uint32_t)lane);
      int cs;
      if (ds.child >= 0 || (KS && p.tip_masks)) {
        // child ~ e_ps^T B^(m-1) (.) PL[child]   (Tvmmp :431-436, :651); ks: tips too, against their mask (:1384-1397)
        double w[NS];
        const int kk = m - 1;
        const int kt = kk;                                   // kk < klong by construction (see child_vector)
        {      // one row per child draw: read from the full-length table in global memory (L2); LDS holds the column table only
          const double* __restrict__ src = p.rowpow + ((size_t)kt * NS + ps) * NS;
#pragma unroll
          for (int c = 0; c < NS; ++c) w[c] = src[c];
        }
        uint32_t node_id;
        if (ds.child >= 0) {
#pragma unroll
          for (int c = 0; c < NS; ++c) w[c] = w[c] * at(PLt, (uint32_t)(ds.child * NS + c) * 512u + (uint32_t)lane * 8u);
          node_id = (uint32_t)(ds.child + p.n_tips);
        } else {
          int tip = ~ds.child;
          int par = (p.tips_per_replica ? at(tips_t, (uint32_t)tip * 64u + (uint32_t)lane) : p.tips[tip]) & 1;
#pragma unroll
          for (int c = 0; c < NS; ++c) w[c] = w[c] * (((c & 1) == par) ? 1.0 : 0.0);
          node_id = (uint32_t)tip;
        }
        double u = stream_u(p.seed_lo, p.seed_hi, rep, (uint32_t)it, ENT_NODE | node_id, 0);
        cs = sample_cat<NS>(w, u, err);                                        // :655
        if (ds.child >= 0) at(nst, (uint32_t)ds.child * 64u + (uint32_t)lane) = (uint8_t)cs;
      } else {
        int tip = ~ds.child;
        cs = p.tips_per_replica ? at(tips_t, (uint32_t)tip * 64u + (uint32_t)lane) : p.tips[tip];        // :612
      }

      // ---- branch path: resample states, merge, count, re-insert virtual jumps ----
      // The dwell lists of a tile form one sequential stream of 64-lane rows in sweep order: branch k
      // occupies wave_max_count(m) rows of the input stream and wave_max_count(m') rows of the output stream.
      Stream su, se;
      su.open(ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
      se.open(ENT_BEXP | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi);
      const int roff = in_row;
      const int woff = out_row;
      const int cap = RING ? C - (r_in - in_row) - out_row    // free rows: the ring minus unread input minus output so far
                           : C - out_row;
      const int mmax = wave_max_count(m);
      int mnew = 0;                              // pieces emitted = new segment count

      // s_i ~ B[s_{i-1},:] (.) B^(m-i-1) e_end          (resamplebranchstates :290, :301-304)
      auto draw_state_w = [&](int i, int sprev, uint32_t word) -> int {
        const int kk = m - i - 1;
        double pr[NS];
        const int kt = kk;
        const int kl = kt < MCMC_KTAB ? kt : MCMC_KTAB - 1;
        {
          const double* beta = s_col + (kl * NS + cs) * NS;
#pragma unroll
          for (int c = 0; c < NS; ++c) pr[c] = beta[c];
        }
        if (kt >= MCMC_KTAB) {
          const double* __restrict__ beta = p.colpow + ((size_t)kt * NS + cs) * NS;
#pragma unroll
          for (int c = 0; c < NS; ++c) pr[c] = beta[c];
        }
        const double* row = s_B2 + sprev * NS;
#pragma unroll
        for (int c = 0; c < NS; ++c) pr[c] = row[c] * pr[c];
        return sample_cat<NS>(pr, u01(word), err);
      };
      auto draw_state = [&](int i, int sprev) -> int { return draw_state_w(i, sprev, su.draw_word((uint32_t)(i - 1))); };

      if (mmax <= 64 && NS <= 4) {
        // Two flat passes, so that a wave pays max-over-lanes ONCE per pass instead of once per nesting level.
        // Pass A: one old segment per step for every lane; merged segments are written back in place over the
        // consumed part of the input stream, their states packed 2 bits apiece into two registers.
        uint64_t pk0 = 0, pk1 = 0;
        int w = 0;
        int cur_s = (m == 1) ? cs : ps;            // updatenodestates :469-472 (m==1: child wins)
        // the first two and the last merged segment stay in registers; only the ones between them pass through the stream's rows
        double first_len = 0.0, second_len = 0.0;
        double cur_len = IN(roff);
        double dnext = (m > 1) ? IN(roff + 1) : 0.0;
        // Four steps per Philox block of the state stream: the step index is wave-uniform here, so the block is computed
        // once per group by every lane and draw i - 1 is a fixed word of it (no per-step block test or word select).
        for (int i0 = 1; i0 < mmax; i0 += 4) {
          uint32_t wd[4] = {0u, 0u, 0u, 0u};
          if (i0 < mmax - 1)                       // some lane still draws in this group (draws exist for i < m - 1)
            philox4x32((uint32_t)((i0 - 1) >> 2), ENT_BSTATE | (uint32_t)b, (uint32_t)it, rep, p.seed_lo, p.seed_hi, wd);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
          const int i = i0 + q;
          if (i < m) {
            int si = (i == m - 1) ? cs : draw_state_w(i, cur_s, wd[q]);
            double di = dnext;
            if (i + 1 < m) dnext = IN(roff + i + 1);
            if (KS) s_cnt[(cur_s * NS + si) * 64 + lane] += 1u;               // shortenerbf :1010-1014
            if (si == cur_s) cur_len = cur_len + di;                           // shortener :54
            else {
              if (w == 0) first_len = cur_len; else if (w == 1) second_len = cur_len; else IN(roff + w) = cur_len;
              if (w < 32) pk0 |= (uint64_t)cur_s << (2 * w); else pk1 |= (uint64_t)cur_s << (2 * (w - 32));
              if (!KS) {
                int col = cur_s * (NS - 1) + (si > cur_s ? si - 1 : si);       // shortener :65-66
                s_cnt[col * 64 + lane] += 1u;
              }
              ++w; cur_s = si; cur_len = di;
            }
          }
          }
        }
        if (w < 32) pk0 |= (uint64_t)cur_s << (2 * w); else pk1 |= (uint64_t)cur_s << (2 * (w - 32));
        const int nmerged = w + 1;
        const double len0 = (w == 0) ? cur_len : first_len;

        // Pass B: one new piece per step for every lane (virtual jumps :391-410, dwell sums :745-757).
        int j = 0;
        int s = (int)(pk0 & 3u);
        double len = len0;
        double lnext = (nmerged > 1) ? ((w == 1) ? cur_len : second_len) : 0.0;
        double tot = 0.0, scale = s_scale[s], acc = s_dw[s * 64 + lane];
        bool stuck = false, done = false;
        // Four steps per Philox block of the exponential stream.  A lane emits one piece per step and draws one variate for
        // it until it meets a zero-length segment, after which it draws no more on this branch (`stuck`): its draw counter
        // equals the step index whenever it draws, so draw t0 + q is word q of the group's block.
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
            if (mnew < cap) OUT(woff + mnew) = piece; else err |= DERR_CAPACITY;
            acc += piece;                                                        // updatedwelltimes :752
            ++mnew;
            if (adv) {
              s_dw[s * 64 + lane] = acc;
              ++j;
              if (j >= nmerged) done = true;
              else {
                len = lnext;
                if (j + 1 < w) lnext = IN(roff + j + 1); else lnext = cur_len;      // the last merged segment never left its register
                s = (int)(((j < 32) ? (pk0 >> (2 * j)) : (pk1 >> (2 * (j - 32)))) & 3u);
                scale = s_scale[s]; tot = 0.0; acc = s_dw[s * 64 + lane];
              }
            }
          }
        }
      } else {
        // General path (a lane with more than 64 segments on this branch): the reference's loop nest as written.
        uint32_t edraw = 0;
        bool stuck = false;
        auto finalize = [&](int s, double len) {
          if (stuck || !(0.0 < len)) {
            stuck = true;
            if (mnew < cap) OUT(woff + mnew) = len; else err |= DERR_CAPACITY;
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
            if (mnew < cap) OUT(woff + mnew) = piece; else err |= DERR_CAPACITY;
            acc += piece;
            ++mnew;
          }
          s_dw[s * 64 + lane] = acc;
        };
        int cur_s = (m == 1) ? cs : ps;
        double cur_len = IN(roff);
        for (int i = 1; i <= m; ++i) {              // i == m: sentinel that flushes the last merged segment
          int si = -1;
          double di = 0.0;
          if (i < m) {
            si = (i == m - 1) ? cs : draw_state(i, cur_s);
            di = IN(roff + i);
          }
          if (KS && si >= 0) s_cnt[(cur_s * NS + si) * 64 + lane] += 1u;
          if (si == cur_s) cur_len = cur_len + di;
          else {
            finalize(cur_s, cur_len);
            if (!KS && si >= 0) {
              int col = cur_s * (NS - 1) + (si > cur_s ? si - 1 : si);
              s_cnt[col * 64 + lane] += 1u;
            }
            cur_s = si; cur_len = di;
          }
        }
      }
      if (mnew > 65535) { err |= DERR_CAPACITY; mnew = 65535; }
      if (mnew > cap) mnew = cap > 1 ? cap : 1;      // after a capacity error: keep the stored count inside the stream (and the chain tables)
      at(mct, (uint32_t)b * 128u + (uint32_t)lane * 2u) = (uint16_t)mnew;
      seg_rw += (uint32_t)(m + mnew);
      in_row += mmax;
      out_row += wave_max_count(mnew);
      if (out_row > C) out_row = C;
    }

    // ------------------------------ statistics row of this iteration ------------------------------
    if (RING) {
      cur_r = wbase;                                        // the stream just written is the next sweep's input
      cur_w = wbase + out_row; if (cur_w >= C) cur_w -= C;
    } else cur_r ^= 1;

    // columns: n dwell sums, NCNT transition counters, then (ks) the root state, 0-based (:1350-1352)
    constexpr int DCOLS = NS + NCNT + (KS ? 1 : 0);
    if (p.reduce) {
      double* dst = p.stats + ((size_t)it * p.n_tiles + tile) * p.n_cols;
#pragma unroll
      for (int c = 0; c < DCOLS; ++c) {
        double v = 0.0;
        if (valid) v = (c < NS) ? s_dw[c * 64 + lane] : (c < NS + NCNT) ? (double)s_cnt[(c - NS) * 64 + lane] : (double)my_root;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == c) dst[c] = v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < DCOLS; ++c) {
        double v = (c < NS) ? s_dw[c * 64 + lane] : (c < NS + NCNT) ? (double)s_cnt[(c - NS) * 64 + lane] : (double)my_root;
        p.stats[((size_t)it * p.n_cols + c) * p.n_rep_pad + rep_local] = v;
      }
    }
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

template <int NS, bool KS, bool RING>
__global__ __launch_bounds__(MCMC_BLOCK, 6) void mcmc_sweep_kernel(McmcParams<NS> p, int iter0, int n_iters) {
  sweep_body<NS, KS, RING>(p, iter0, n_iters);
}

// the same sweep over a list of trees (ring storage only)
template <int NS, bool KS>
__global__ __launch_bounds__(MCMC_BLOCK, 5) void mcmc_sweep_trees_kernel(McmcParams<NS> p, int iter0, int n_iters) {
  sweep_body<NS, KS, true, true>(p, iter0, n_iters);
}

// The pruning (up) sweep alone, under its own name so that profiles separate it from the full sweep (p.prune_only = 1).
template <int NS, bool KS, bool RING>
__global__ __launch_bounds__(MCMC_BLOCK, 6) void mcmc_pruning_kernel(McmcParams<NS> p, int iter0, int n_iters) {
  sweep_body<NS, KS, RING>(p, iter0, n_iters);
}

// Load the caller's initial paths (x$maps, makeabranch src/phylomap.cpp:24-34) into every replica:
// one sequential stream per tile in sweep order, init_row[k] = first row of branch down[k].
__global__ void mcmc_init_kernel(int n_edge, int n_tiles, int64_t rows, const DownStep* __restrict__ down,
                                 const int32_t* __restrict__ init_row, const int32_t* __restrict__ map_off,
                                 const double* __restrict__ maps, double* __restrict__ dwell0,
                                 uint16_t* __restrict__ mcount) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.y;
  for (int k = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); k < n_edge; k += gridDim.x * (blockDim.x / 64)) {
    const DownStep ds = down[k];
    const int o = map_off[ds.edge];
    const int m = map_off[ds.edge + 1] - o;
    double* dst = dwell0 + ((size_t)tile * rows + init_row[k]) * 64;
    for (int i = 0; i < m; ++i) dst[i * 64 + lane] = maps[o + i];
    mcount[((size_t)tile * n_edge + ds.edge) * 64 + lane] = (uint16_t)m;
  }
}

// reduce mode: out[it][col] = sum over tiles of the per-wave partials, fixed order
// `init` (may be NULL): the fold continues from the totals of the devices before this one (multi-device one-shot calls)
__global__ void stats_reduce_kernel(const double* __restrict__ partial, int n_iters, int n_tiles, int n_cols,
                                    double* __restrict__ out, const double* __restrict__ init) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_iters * n_cols) return;
  int it = idx / n_cols, c = idx - it * n_cols;
  double acc = init ? init[idx] : 0.0;
  for (int t = 0; t < n_tiles; ++t) acc += partial[((size_t)it * n_tiles + t) * n_cols + c];
  out[idx] = acc;
}

// per-replica statistics rows [row][n_rep_pad] -> the columns of the replicas that exist, packed: [row][n_pick] (pick[r] = padded index)
__global__ void stats_gather_kernel(const double* __restrict__ stats, int64_t n_rows, int n_rep_pad, int n_pick, const int32_t* __restrict__ pick,
                                    double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * n_pick) return;
  const int64_t row = idx / n_pick;
  const int r = (int)(idx - row * n_pick);
  out[idx] = stats[row * n_rep_pad + pick[r]];
}

template <int NS>
size_t mcmc_lds_bytes(int ktab, bool ks) {
  const size_t ncnt = ks ? NS * NS : NS * (NS - 1);
  // the mask-chain table sits behind a full n x n counter block (kernel carve-up), so ks reserves that much
  return sizeof(double) * ((size_t)ktab * NS * NS + NS * NS + NS + (size_t)(MCMC_BLOCK / 64) * NS * 64) +
         sizeof(uint32_t) * (size_t)(MCMC_BLOCK / 64) * (ks ? NS * NS : ncnt) * 64 +
         (ks ? sizeof(double) * (size_t)ktab * 2 * NS : 0);
}

template <int NS>
hipError_t launch_mcmc(const McmcParams<NS>& p, int iter0, int n_iters, hipStream_t stream) {
  const int waves_per_block = MCMC_BLOCK / 64;
  dim3 grid((p.n_tiles + waves_per_block - 1) / waves_per_block);
  if (p.ktab != MCMC_KTAB || p.klong < MCMC_KTAB) return hipErrorInvalidValue;      // the kernels carve LDS for MCMC_KTAB rows
  size_t lds = mcmc_lds_bytes<NS>(p.ktab, p.ks != 0);
  const bool ring = p.dwell1 == nullptr;
  if (p.tiles_per_tree) {
    if (!ring || p.prune_only) return hipErrorInvalidValue;
    if (p.ks) hipLaunchKernelGGL((mcmc_sweep_trees_kernel<NS, true>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    else hipLaunchKernelGGL((mcmc_sweep_trees_kernel<NS, false>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    return hipGetLastError();
  }
  if (p.prune_only) {
    if (p.ks && ring) hipLaunchKernelGGL((mcmc_pruning_kernel<NS, true, true>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    else if (p.ks) hipLaunchKernelGGL((mcmc_pruning_kernel<NS, true, false>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    else if (ring) hipLaunchKernelGGL((mcmc_pruning_kernel<NS, false, true>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    else hipLaunchKernelGGL((mcmc_pruning_kernel<NS, false, false>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
    return hipGetLastError();
  }
  if (p.ks && ring) hipLaunchKernelGGL((mcmc_sweep_kernel<NS, true, true>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
  else if (p.ks) hipLaunchKernelGGL((mcmc_sweep_kernel<NS, true, false>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
  else if (ring) hipLaunchKernelGGL((mcmc_sweep_kernel<NS, false, true>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
  else hipLaunchKernelGGL((mcmc_sweep_kernel<NS, false, false>), grid, dim3(MCMC_BLOCK), lds, stream, p, iter0, n_iters);
  return hipGetLastError();
}

template hipError_t launch_mcmc<2>(const McmcParams<2>&, int, int, hipStream_t);
template hipError_t launch_mcmc<3>(const McmcParams<3>&, int, int, hipStream_t);
template hipError_t launch_mcmc<4>(const McmcParams<4>&, int, int, hipStream_t);
template size_t mcmc_lds_bytes<2>(int, bool);
template size_t mcmc_lds_bytes<3>(int, bool);
template size_t mcmc_lds_bytes<4>(int, bool);

hipError_t launch_mcmc_init(int n_edge, int n_tiles, int64_t rows, const DownStep* down, const int32_t* init_row,
                            const int32_t* map_off, const double* maps, double* dwell0, uint16_t* mcount,
                            hipStream_t stream) {
  dim3 grid(64, n_tiles);
  hipLaunchKernelGGL(mcmc_init_kernel, grid, dim3(256), 0, stream, n_edge, n_tiles, rows, down, init_row, map_off, maps,
                     dwell0, mcount);
  return hipGetLastError();
}

hipError_t launch_stats_reduce(const double* partial, int n_iters, int n_tiles, int n_cols, double* out,
                               hipStream_t stream, const double* init) {
  int total = n_iters * n_cols;
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, partial, n_iters, n_tiles,
                     n_cols, out, init);
  return hipGetLastError();
}

hipError_t launch_stats_gather(const double* stats, int64_t n_rows, int n_rep_pad, int n_pick, const int32_t* pick, double* out, hipStream_t stream) {
  const int64_t total = n_rows * n_pick;
  hipLaunchKernelGGL(stats_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, stats, n_rows, n_rep_pad, n_pick, pick, out);
  return hipGetLastError();
}

}  // namespace phm
