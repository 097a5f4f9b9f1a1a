// phm_engine.cpp -- the resident engine behind the C-ABI (include/phylomap_hip.h): input validation, HBM layout of the
// three mappings of the sweep (phm_mcmc.hip / phm_tiles.hip / phm_narrow.hip, phm_wide.hip for n > 4), launches, statistics.
//
// Host-side counterpart of what the exported drivers do before and after their N-loop (src/phylomap.cpp:891-986, 822-870):
// unpack `x`, set up B, allocate the statistics matrix.  Built with -ffp-contract=off: the B^k chain tables computed here
// must carry exactly the bits the kernels (and the oracle) would produce by running the chains themselves.
#include "phm_internal.h"

namespace {

// Automatic choice of the mapping (n <= 4).  Round 1 on C2 (profiles/r01_probe_mapping.log; ms per sweep):
//   chains                  1     32     64    256   1024   4096   16384   65536   131072   196608   327680   393216
//   lane = branch         0.17   0.22   0.25   0.62   2.4    9.3
//   wave = tile x branch         0.32   0.32   0.36   0.48   1.0     3.0    10.0     19.3    (28.5 at 182 GiB; 262 144: 37.5)
//   lane = replica        19.0                 21.4   25.8   26.7    26.7    28.2     34.4     39.0     48.6     53.2
// One lane group per branch for a handful of chains; one wave per (tile, branch) as long as its slots fit in HBM; the replica
// mapping (a single wave per tile, compact sequential streams) for the largest replica counts.
// Round 3 (latency-shaped one-chain kernels, profiles/r03_probe_crossover.log): the branch mapping costs
// 0.04 + 0.002 x levels + 5.2e-7 x S x E ms per sweep, the (tile, branch) mapping 0.008 x levels + 2.6e-5 x E while its tiles
// do not fill the device (C1 / C2 / C3: 0.16 / 0.33 / 0.84 ms up to ~512 chains) -> the branch mapping up to
//   S* = (0.008 levels + 2.6e-5 E - 0.02) / (5.2e-7 E)      (C3: 72 chains, measured crossover 64..96; C2: 208, measured 256..384;
//                                                             C1: 1 020, measured > 1 024)
inline int narrow_auto_max_replicas(const phm::Schedule& s) {
  std::vector<int32_t> depth(s.n_node, 0);
  int levels = 1;
  for (const phm::DownStep& d : s.down)              // parents before children
    if (d.child >= 0) { depth[d.child] = depth[d.parent] + 1; levels = std::max(levels, depth[d.child] + 1); }
  const double E = (double)s.n_edge;
  // Round 4 (profiles/r04_probe_crossover.log): with its tree passes over level clusters, counter copies and a workgroup per tile for the
  // statistics, the (tile, branch) mapping with ONE tile costs 0.03 + 0.0068 levels + 6e-6 E ms per sweep (C1 / C2 / C3: 0.11 / 0.21 / 0.33);
  // the branch mapping 0.04 + 0.002 levels + 6.5e-7 S E  ->  S* = 454 / 87 / 20 chains (measured crossovers ~500 / ~110 / ~20)
  const double cap = (0.0048 * levels + 6e-6 * E - 0.01) / (6.5e-7 * E);
  return (int)std::max(8.0, std::min(4096.0, cap));
}
constexpr int TILES_AUTO_MAX_REPLICAS = 262144;
// 5..64 states: a wave per (replica, branch) (phm_wbranch.hip) exposes S x E waves whatever S is; the lane-per-replica mapping
// (phm_wtiles.hip) needs whole tiles of 64 replicas and pays a fixed serial cost per tree level (one tile: 1.04 ms per sweep on
// C4, 0.81 ms on C5).  Round 3 (two waves per node, LDS-fed chains, transition maps, per-column statistics) moved the crossover
// (profiles/r03_probe_small_S_C{4,5}.log): C4 (1 000 branches) 0.76 ms at one chain, + 9 us per further chain -> ~30 chains;
// C5 (10 000 branches) 0.38 ms, + 60 us per chain -> ~7 chains.  The slope follows the branch count: 50 000 / E chains, at most 32.
// Late round 4 (tools/probes/probe_auto_choice.py, profiles/r04_probe_auto_choice.log): the cap of 32 chains handed small trees
// (118 branches: crossover ~200 chains; 64 chains 0.15-0.42 ms against 0.26-0.89) and deep ones (a launch per tree level and pass is
// the tile mapping's floor there: 1 200-tip ladder, 64 chains at 8 states 3.1 against 7.2 ms, 200 chains at 33 states 6.4 against
// 17.9) to the tile mapping too early: 50 000 / E chains, at most 256; 256 on a deep tree.
inline int wbranch_auto_max_replicas(const phm::Schedule& s) {
  std::vector<int32_t> depth(s.n_node, 0);
  int levels = 1;
  for (const phm::DownStep& d : s.down)              // parents before children
    if (d.child >= 0) { depth[d.child] = depth[d.parent] + 1; levels = std::max(levels, depth[d.child] + 1); }
  int lg2 = 0;
  while ((1 << lg2) < s.n_node + 1) ++lg2;
  const bool deep = levels > 4 * lg2 + 32;
  return deep ? 256 : std::max(1, std::min(256, 50000 / std::max(1, s.n_edge)));
}

// Tail at which the fixed slots of the branch-parallel mappings are provisioned, per (replica, branch, sweep).  A slot that
// overflows costs a rebuild with doubled slots and a replay (recover_capacity), so the tail is set from the number of draws
// the engine is created for: at most 0.05 expected recoveries over S x E x max_iters draws, and never looser than 1e-9 (the
// quantile of 1 + Poisson(Omega t_b) moves by one segment per factor ~6 at Omega t_b = 4, so short runs get ~20 % smaller
// slots than the 1e-14 of round 2 and the full-length C3 run keeps about what it had).
inline double default_slot_tail(int64_t S, int64_t E, int64_t max_iters) {
  const double draws = (double)std::max<int64_t>(S, 1) * (double)std::max<int64_t>(E, 1) * (double)std::max<int64_t>(max_iters, 1);
  return std::max(1e-16, std::min(1e-9, 0.05 / draws));
}

bool ks_layout(int v) { return v == PHM_MCMC_KS || v == PHM_MCMC_BF || v == PHM_MCMC_MT || v == PHM_MCMC_KSMT; }   // n x n counts, root column
bool hidden_rates(int v) { return v == PHM_MCMC_KS || v == PHM_MCMC_KSMT; }                                         // parity tip masks
bool normalised_variant(int v) { return v == PHM_MCMC_BIGTREE || v == PHM_MCMC_KS || v == PHM_MCMC_BF; }           // makePLnormalized :1085

template <int NS>
void fill_params(phm_engine* e, phm::McmcParams<NS>& p, const double* B2, const double* Bc, const double* scale,
                 const double* pid, const phm_options& o) {
  p.n_tips = e->sched.n_tips; p.n_node = e->sched.n_node; p.n_edge = e->sched.n_edge; p.root = e->sched.root;
  p.n_tiles = e->tiles; p.n_rep = e->n_trees > 1 ? e->S_tree : e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset;
  p.tiles_per_tree = e->n_trees > 1 ? e->tpt : 0; p.roots = e->d_roots.as<int32_t>();
  p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
  p.reduce = e->reduce; p.n_cols = e->dcols; p.ktab = phm::MCMC_KTAB; p.klong = std::max(e->nw_klong, phm::MCMC_KTAB); p.prune_only = 0;
  p.ks = ks_layout(e->variant); p.tip_masks = hidden_rates(e->variant);
  p.maskpow = e->d_mask.as<double>();
  p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
  p.rows = e->rows;
  for (int i = 0; i < NS * NS; ++i) { p.B2[i] = B2[i]; p.Bc[i] = Bc[i]; }
  for (int i = 0; i < NS; ++i) { p.scale[i] = scale[i]; p.pid[i] = pid[i]; }
  p.up = e->d_up.as<phm::UpStep>(); p.down = e->d_down.as<phm::DownStep>();
  p.colpow = e->d_col.as<double>(); p.rowpow = e->d_row.as<double>();
  p.tips = e->d_tips.as<uint8_t>(); p.mcount = e->d_mcount.as<uint16_t>();
  p.dwell0 = e->d_dw0.as<double>(); p.dwell1 = e->ring ? nullptr : e->d_dw1.as<double>(); p.cursor = e->d_cursor.as<int32_t>();
  p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>(); p.stats = e->d_stats.as<double>();
  p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
}

// Model matrices from R's column-major Q / B: dense B2, chain matrix Bc (thresholded for SPARSE), rexp scales, and the
// parameter columns recordQ / recordQks write (src/phylomap.cpp:1181-1185, :1789-1798).
int32_t compute_model(int variant, int n, const double* Q, const double* B, double Omega, std::vector<double>& B2,
                      std::vector<double>& Bc, std::vector<double>& scale, std::vector<double>& qparams) {
  B2.assign((size_t)n * n, 0.0); Bc.assign((size_t)n * n, 0.0); scale.assign(n, 0.0);
  for (int i = 0; i < n; ++i) {
    double r = Omega + Q[i + (size_t)i * n];
    if (!(r >= 0.0)) return fail(PHM_ERR_BAD_INPUT, "Omega must be at least |q_ii| for every state (man/sumstatMCMC.Rd:14)");
    scale[i] = 1.0 / r;
    for (int j = 0; j < n; ++j) {
      double b = B ? B[i + (size_t)j * n] : ((i == j ? 1.0 : 0.0) + Q[i + (size_t)j * n] / Omega);   // R/sumstatMCMC.R:25
      if (!(b >= 0.0) || !std::isfinite(b)) return fail(PHM_ERR_BAD_INPUT, "B = I + Q/Omega must be non-negative");
      B2[(size_t)i * n + j] = b;
      Bc[(size_t)i * n + j] = (variant == PHM_MCMC_SPARSE) ? (b > 1e-7 ? b : 0.0) : b;              // matTospmat :811
    }
  }
  qparams.clear();
  if (ks_layout(variant)) {
    const int k = hidden_rates(variant) ? n / 2 - 1 : 0;
    auto Qe = [&](int i, int j) { return Q[i + (size_t)j * n]; };
    qparams.push_back(Qe(0, 1));
    qparams.push_back(Qe(1, 0));
    for (int i = 0; i < k; ++i) qparams.push_back(Qe(2 * i, 2 * i + 2));
    for (int i = 0; i < k; ++i) qparams.push_back(Qe(2 * i + 2, 2 * i));
    for (int i = 0; i < k; ++i) qparams.push_back(Qe(2 * (i + 1), 2 * (i + 1) + 1) / Qe(0, 1));
  }
  return PHM_OK;
}

// chain tables for the current model -> device; refresh the by-value kernel parameter blocks
int32_t upload_model(phm_engine* e) {
  const int n = e->n;
  // rows of the chain tables: every mapping keeps full-length tables in global memory (nw_klong > every possible segment
  // count); the replica kernels additionally stage the first MCMC_KTAB rows in LDS
  const int ktab = (e->narrow || e->tiled) ? e->nw_klong : std::max(e->nw_klong, e->wide ? phm::WIDE_KTAB : phm::MCMC_KTAB);      // narrow covers both branch mappings (n <= 4 and 5..64)
  const double* Bc = e->hBc.data();
  std::vector<double> col, row;
  build_chain_tables(Bc, n, ktab, col, row, n > 4);
  std::vector<double> maskpow((size_t)ktab * 2 * n, 0.0);      // ks: Bc^k applied to the even / odd state masks (:1838-1845)
  for (int par = 0; par < 2; ++par) {
    for (int c = 0; c < n; ++c) maskpow[(size_t)par * n + c] = ((c & 1) == par) ? 1.0 : 0.0;
    for (int k = 1; k < ktab; ++k)
      host_chain_matvec(Bc, n, &maskpow[((size_t)(k - 1) * 2 + par) * n], &maskpow[((size_t)k * 2 + par) * n], n > 4);
  }
  if (e->tiled && e->wide) {        // phm_wtiles.hip: table rows padded to an even length (16-byte rows), model in global memory
    const int ldt = e->pwt.ldt;
    auto padded = [&](const std::vector<double>& src, size_t rows) {
      std::vector<double> dst(rows * ldt, 0.0);
      for (size_t r = 0; r < rows; ++r) std::memcpy(&dst[r * ldt], &src[r * n], sizeof(double) * n);
      return dst;
    };
    const std::vector<double> colp = padded(col, (size_t)ktab * n), rowp = padded(row, (size_t)ktab * n),
                              maskp = padded(maskpow, (size_t)ktab * 2), b2p = padded(e->hB2, (size_t)n);
    HIPCHK(hipMemcpy(e->d_nw_colL.p, colp.data(), sizeof(double) * colp.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_rowL.p, rowp.data(), sizeof(double) * rowp.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_maskL.p, maskp.data(), sizeof(double) * maskp.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_wt_B2.p, b2p.data(), sizeof(double) * b2p.size(), hipMemcpyHostToDevice));
    {   // running sums of the forward draws' probability vectors p_c = B2[s][c] * (Bc^k e_end)[c] (resamplebranchstates :301),
        // added exactly as the sampler adds them (left to right, unfused; this file is built with -ffp-contract=off); every
        // eighth one is kept, and the total
      const int nb = e->pwt.nblk, ldb = e->pwt.ldb;
      std::vector<double> blk((size_t)ktab * n * n * ldb, 0.0);
      auto rows_of = [&](int k0, int k1) {             // the rows of different k are independent: dealt to a few host threads when there are many
        for (int k = k0; k < k1; ++k)
          for (int sp = 0; sp < n; ++sp)
            for (int en = 0; en < n; ++en) {
              const double* beta = &col[((size_t)k * n + en) * n];
              const double* brow = &e->hB2[(size_t)sp * n];
              double* out = &blk[(((size_t)k * n + sp) * n + en) * ldb];
              double t = brow[0] * beta[0];
              for (int c = 1; c < n; ++c) {
                if ((c & 7) == 0) out[(c >> 3) - 1] = t;      // the sum after state c - 1 = 8q + 7
                t += brow[c] * beta[c];
              }
              out[nb - 1] = t;
            }
      };
      const int nt = (double)ktab * n * n * n < 4e6 ? 1 : std::min<int>({8, ktab, (int)std::max(1u, std::thread::hardware_concurrency())});
      if (nt <= 1) rows_of(0, ktab);
      else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(rows_of, (int)((int64_t)ktab * t / nt), (int)((int64_t)ktab * (t + 1) / nt));
        for (std::thread& t : th) t.join();
      }
      HIPCHK(hipMemcpy(e->d_wt_totL.p, blk.data(), sizeof(double) * blk.size(), hipMemcpyHostToDevice));
    }
    {   // transitions that can occur at all: a -> c with B2[a][c] != 0 (c != a unless self pairs are counted).  Few of them
        // (a banded rate matrix) and n <= 32: the branch kernel counts them in LDS slots
      const bool ks = ks_layout(e->variant);
      std::vector<int16_t> pair_slot((size_t)n * n, (int16_t)-1);
      std::vector<int32_t> slot_col;
      for (int a = 0; a < n; ++a)
        for (int c = 0; c < n; ++c)
          if ((ks || a != c) && e->hB2[(size_t)a * n + c] != 0.0) {
            pair_slot[(size_t)a * n + c] = (int16_t)slot_col.size();
            slot_col.push_back(ks ? a * n + c : a * (n - 1) + (c > a ? c - 1 : c));
          }
      int n_slots = (int)slot_col.size();
      if (n > 32 || n_slots > phm::WT_MAX_SLOTS) { n_slots = 0; std::fill(pair_slot.begin(), pair_slot.end(), (int16_t)-1); }
      slot_col.resize(phm::WT_MAX_SLOTS, 0);
      e->pwt.n_slots = n_slots;
      HIPCHK(hipMemcpy(e->d_wt_pair_slot.p, pair_slot.data(), sizeof(int16_t) * pair_slot.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_wt_slot_col.p, slot_col.data(), sizeof(int32_t) * slot_col.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(e->d_Bc.p, e->hBc.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_scale.p, e->hscale.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    {   // Banded matrices (SPARSEmakePLrcpp :490-501 and SPARSEresamplebranchstates :218-261 exploit sp_mat; here: the band).
        // Decided per model: a rate update of the Q-updating drivers may widen or narrow the band between sweeps.
      auto half_bandwidth = [&](const std::vector<double>& M) {
        int hb = 0;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) if (M[(size_t)i * n + j] != 0.0) hb = std::max(hb, std::abs(i - j));
        return hb;
      };
      auto usable = [&](int hb) { return e->sparse_req != 2 && n <= phm::WT_BAND_NMAX && hb >= 1 && hb <= phm::WT_BAND_MAX && 2 * hb + 1 < n; };
      const int hbc = half_bandwidth(e->hBc), hb2 = half_bandwidth(e->hB2);
      e->pwt.band_up = usable(hbc) ? hbc : 0;
      e->pwt.band_draw = usable(hb2) ? hb2 : 0;
      // Any other sparse pattern (an amino-acid neighbour structure is not banded): a pruning kernel generated for the pattern of the
      // chain matrix and compiled now (phm_rtc.h); cached per pattern, so a rate update that keeps the zeros only refreshes the values.
      e->wt_sparse.kernel = nullptr;
      if (!e->pwt.band_up && e->sparse_req != 2 && n <= phm::RTC_SPARSE_NMAX) {
        std::vector<int32_t> rp(1, 0), cj;
        std::vector<double> cv;
        for (int i = 0; i < n; ++i) {
          for (int j = 0; j < n; ++j) if (e->hBc[(size_t)i * n + j] != 0.0) { cj.push_back(j); cv.push_back(e->hBc[(size_t)i * n + j]); }
          rp.push_back((int32_t)cj.size());
        }
        if ((double)cj.size() <= phm::RTC_SPARSE_MAX_FILL * n * n) {
          std::string rerr;
          const phm::SparseUpKernel* k = phm::rtc_sparse_up_kernel(n, rp, cj, rerr);
          if (!k && e->sparse_req == 1) return fail(PHM_ERR_UNSUPPORTED, "sparse_chains = 1: " + rerr);
          if (k) {
            if (!e->d_wt_coef.p) HIPCHK(e->d_wt_coef.alloc(sizeof(double) * (size_t)n * n));
            HIPCHK(hipMemcpy(e->d_wt_coef.p, cv.data(), sizeof(double) * cv.size(), hipMemcpyHostToDevice));
            phm::RtcUpParams& r = e->wt_sparse.params;
            const phm::WtParams& w = e->pwt;
            r.n_states = n; r.ldt = w.ldt; r.n_tips = w.n_tips; r.n_node = w.n_node; r.n_edge = w.n_edge; r.n_tiles = w.n_tiles;
            r.normalise = w.normalise; r.tips_per_replica = w.tips_per_replica; r.tip_masks = w.tip_masks; r.klong = w.klong;
            r.up = reinterpret_cast<const phm::RtcUpStep*>(w.up); r.up_order = w.up_order; r.colL = w.colL; r.maskL = w.maskL; r.tips = w.tips;
            r.mcount = w.mcount; r.PL = w.PL; r.err = w.err; r.coef = e->d_wt_coef.as<double>();
            e->wt_sparse.kernel = k;
          }
        }
      }
      if (e->sparse_req == 1 && !e->pwt.band_up && !e->wt_sparse.kernel)
        return fail(PHM_ERR_UNSUPPORTED, "sparse_chains = 1: the sparse pruning kernels take n <= 32 states and a chain matrix that is banded (half-bandwidth <= 2) or at most half full");
      std::memset(&e->wt_band, 0, sizeof e->wt_band);
      if (e->pwt.band_up) {
        const int hb = e->pwt.band_up, w = 2 * hb + 1;
        for (int i = 0; i < n; ++i)
          for (int d = 0; d < w; ++d) { const int j = i + d - hb; if (j >= 0 && j < n) e->wt_band.c[i * w + d] = e->hBc[(size_t)i * n + j]; }
      }
      std::vector<double> b2band((size_t)n * (2 * phm::WT_BAND_MAX + 1), 0.0);
      if (e->pwt.band_draw) {
        const int hb = e->pwt.band_draw, w = 2 * hb + 1;
        for (int i = 0; i < n; ++i)
          for (int d = 0; d < w; ++d) { const int j = i + d - hb; if (j >= 0 && j < n) b2band[(size_t)i * w + d] = e->hB2[(size_t)i * n + j]; }
      }
      HIPCHK(hipMemcpy(e->d_wt_B2band.p, b2band.data(), sizeof(double) * b2band.size(), hipMemcpyHostToDevice));
    }
    return PHM_OK;
  }
  if (e->narrow || e->tiled) {      // tables long enough for every possible segment count, read from global memory / L2
    // through page-locked staging (the rate-updating drivers come here after every sweep: three pageable hipMemcpy calls were
    // ~50 us of a 190 us iteration)
    const size_t nc = col.size(), nr = row.size(), nm = maskpow.size();
    HIPCHK(e->pin_up.reserve(sizeof(double) * (nc + nr + nm)));
    double* stage = e->pin_up.as<double>();
    std::memcpy(stage, col.data(), sizeof(double) * nc);
    std::memcpy(stage + nc, row.data(), sizeof(double) * nr);
    std::memcpy(stage + nc + nr, maskpow.data(), sizeof(double) * nm);
    if (e->narrow && !e->wide) {                       // phm_narrow.hip: one block, one copy
      HIPCHK(hipMemcpyAsync(e->d_nw_colL.p, stage, sizeof(double) * (nc + nr + nm), hipMemcpyHostToDevice, e->last_stream));
    } else {
      HIPCHK(hipMemcpyAsync(e->d_nw_colL.p, stage, sizeof(double) * nc, hipMemcpyHostToDevice, e->last_stream));
      HIPCHK(hipMemcpyAsync(e->d_nw_rowL.p, stage + nc, sizeof(double) * nr, hipMemcpyHostToDevice, e->last_stream));
      HIPCHK(hipMemcpyAsync(e->d_nw_maskL.p, stage + nc + nr, sizeof(double) * nm, hipMemcpyHostToDevice, e->last_stream));
    }
    HIPCHK(wait_stream(e->last_stream));               // one wait for the three: the next sweep may be enqueued on another stream
    if (e->wide) {      // 5..64 states: the model matrices live in global memory
      HIPCHK(hipMemcpy(e->d_B2.p, e->hB2.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_Bc.p, e->hBc.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_scale.p, e->hscale.data(), sizeof(double) * n, hipMemcpyHostToDevice));
      // A chain matrix with few non-zeros per row (tridiagonal amino-acid-style Q, the SPARSE threshold) also goes up in
      // ELLPACK form: skipping exact zeros leaves every left-to-right row sum bit-identical (all terms are >= +0).
      auto ellpack = [&](const std::vector<double>& M, DevBuf& dcol, DevBuf& dval, int32_t& w_out) -> int32_t {
        int w = 0;
        for (int i = 0; i < n; ++i) {
          int cnt = 0;
          for (int j = 0; j < n; ++j) cnt += M[(size_t)i * n + j] != 0.0;
          w = std::max(w, cnt);
        }
        w_out = (w >= 1 && w <= phm::WB_ELL_MAX && 3 * w <= n) ? w : 0;
        if (!w_out) return PHM_OK;
        std::vector<int32_t> ec((size_t)n * w);
        std::vector<double> ev((size_t)n * w, 0.0);
        for (int i = 0; i < n; ++i) {
          int t = 0;
          for (int j = 0; j < n; ++j) if (M[(size_t)i * n + j] != 0.0) { ec[(size_t)i * w + t] = j; ev[(size_t)i * w + t] = M[(size_t)i * n + j]; ++t; }
          for (; t < w; ++t) ec[(size_t)i * w + t] = i;
        }
        HIPCHK(hipMemcpy(dcol.p, ec.data(), sizeof(int32_t) * ec.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dval.p, ev.data(), sizeof(double) * ev.size(), hipMemcpyHostToDevice));
        return PHM_OK;
      };
      { int32_t st = ellpack(e->hBc, e->d_ell_col, e->d_ell_val, e->pwb.ell_w); if (st) return st; }
      {   // a banded chain matrix (half-bandwidth 1 or 2): neighbours through wave shifts (coop_matvec_band)
        int hb = 0;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) if (e->hBc[(size_t)i * n + j] != 0.0) hb = std::max(hb, std::abs(i - j));
        e->pwb.band_hb = (e->pwb.ell_w > 0 && hb >= 1 && hb <= 2 && e->sparse_req != 2) ? hb : 0;
      }
      { int32_t st = ellpack(e->hB2, e->d_ell2_col, e->d_ell2_val, e->pwb.ell2_w); if (st) return st; }
      return PHM_OK;
    }
    auto refresh_n = [&](auto& p) {
      for (int i = 0; i < n * n; ++i) { p.B2[i] = e->hB2[i]; p.Bc[i] = e->hBc[i]; }
      for (int i = 0; i < n; ++i) p.scale[i] = e->hscale[i];
    };
    if (e->narrow) { if (n == 2) refresh_n(e->n2); if (n == 3) refresh_n(e->n3); if (n == 4) refresh_n(e->n4); }
    else { if (n == 2) refresh_n(e->t2); if (n == 3) refresh_n(e->t3); if (n == 4) refresh_n(e->t4); }
    return PHM_OK;
  }
  HIPCHK(hipMemcpy(e->d_col.p, col.data(), sizeof(double) * col.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_row.p, row.data(), sizeof(double) * row.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_mask.p, maskpow.data(), sizeof(double) * maskpow.size(), hipMemcpyHostToDevice));
  if (e->wide) {
    HIPCHK(hipMemcpy(e->d_B2.p, e->hB2.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_Bc.p, e->hBc.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_scale.p, e->hscale.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  }
  auto refresh = [&](auto& p) {
    for (int i = 0; i < n * n; ++i) { p.B2[i] = e->hB2[i]; p.Bc[i] = e->hBc[i]; }
    for (int i = 0; i < n; ++i) p.scale[i] = e->hscale[i];
  };
  if (n == 2) refresh(e->p2);
  if (n == 3) refresh(e->p3);
  if (n == 4) refresh(e->p4);
  return PHM_OK;
}

// Branch-parallel engine state (phm_narrow.hip): level schedules, CSR branch slots, long chain tables, per-replica buffers.
template <int NS>
void fill_narrow_params(phm_engine* e, phm::NarrowParams<NS>& p, const phm_options& o) {
  const phm::Schedule& s = e->sched;
  p.n_tips = s.n_tips; p.n_node = s.n_node; p.n_edge = s.n_edge; p.root = s.root;
  p.n_rep = e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset; p.n_tiles = e->tiles;
  p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
  p.ks = ks_layout(e->variant); p.tip_masks = hidden_rates(e->variant); p.reduce = e->reduce; p.n_cols = e->dcols;
  p.klong = e->nw_klong; p.n_wide = e->nw_n_wide; p.cluster_async = e->nw_cluster_async;
  p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
  p.total_cap = e->nw_total_cap;
  for (int i = 0; i < NS * NS; ++i) { p.B2[i] = e->hB2[i]; p.Bc[i] = e->hBc[i]; }
  for (int i = 0; i < NS; ++i) { p.scale[i] = e->hscale[i]; p.pid[i] = e->hpid[i]; }
  p.cl_nodes = e->d_nw_cl_nodes.as<phm::ClusterNode>(); p.cl_item_off = e->d_nw_cl_item_off.as<int32_t>();
  p.cl_lvl_ptr = e->d_nw_cl_lvl_ptr.as<int32_t>(); p.cl_lvl_off = e->d_nw_cl_lvl_off.as<int32_t>();
  p.down_lv = e->d_nw_down_lv.as<phm::DownStep>(); p.walk_off = e->d_nw_walk_off.as<int32_t>();
  p.edge_parent = e->d_nw_edge_parent.as<int32_t>(); p.dmap_edge = e->d_nw_dmap_edge.as<uint16_t>();
  p.branch_order = e->d_nw_border.as<int32_t>(); p.off = e->d_nw_off.as<int64_t>();
  {
    const size_t tab = (size_t)e->nw_klong * NS * NS;      // one block: colL | rowL | maskL (narrow_setup)
    p.colL = e->d_nw_colL.as<double>(); p.rowL = p.colL + tab; p.maskL = p.colL + 2 * tab;
  }
  p.tips = e->d_tips.as<uint8_t>();
  p.mcount = e->d_nw_mcount.as<int32_t>(); p.dw[0] = e->d_nw_dwA.as<double>(); p.dw[1] = e->d_nw_dwB.as<double>();
  p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>(); p.part = e->d_nw_part.as<double>();
  p.rowbuf = e->d_nw_rowbuf.as<double>(); p.stats = e->d_stats.as<double>();
  p.dmap = e->d_nw_dmap.as<uint16_t>();
  p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
  p.host_row = nullptr;
  if (e->S == 1 && !e->reduce && e->pin_row.reserve(sizeof(double) * (e->dcols + 2)) == hipSuccess) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, e->pin_row.p, 0) == hipSuccess) p.host_row = reinterpret_cast<double*>(dp);
  }
}

// Level schedules shared by the branch-parallel mappings: positions of up[] grouped by HEIGHT (children strictly below their
// parent) and of down[] grouped by DEPTH; uploads the two order arrays, leaves the level boundaries in the engine.
int32_t build_level_orders(phm_engine* e) {
  const phm::Schedule& s = e->sched;
  const int E = s.n_edge, Nn = s.n_node;
  std::vector<int32_t> height(Nn, 0), depth(Nn, 0);
  int max_h = 0, max_d = 0;
  for (int k = 0; k < Nn; ++k) {
    const phm::UpStep& u = s.up[k];
    int h = 0;
    for (int c = 0; c < 2; ++c) if (u.child[c] >= 0) h = std::max(h, height[u.child[c]] + 1);
    height[u.parent] = h; max_h = std::max(max_h, h);
  }
  std::vector<int32_t> edepth(E, 0);
  for (int k = 0; k < E; ++k) {
    const phm::DownStep& d = s.down[k];
    edepth[k] = depth[d.parent];
    if (d.child >= 0) depth[d.child] = depth[d.parent] + 1;
    max_d = std::max(max_d, edepth[k]);
  }
  std::vector<int32_t> up_order(Nn), down_order(E);
  e->nw_up_off.assign(max_h + 2, 0); e->nw_down_off.assign(max_d + 2, 0);
  for (int k = 0; k < Nn; ++k) e->nw_up_off[height[s.up[k].parent] + 1]++;
  for (int k = 0; k < E; ++k) e->nw_down_off[edepth[k] + 1]++;
  for (size_t l = 1; l < e->nw_up_off.size(); ++l) e->nw_up_off[l] += e->nw_up_off[l - 1];
  for (size_t l = 1; l < e->nw_down_off.size(); ++l) e->nw_down_off[l] += e->nw_down_off[l - 1];
  {
    std::vector<int32_t> pu(e->nw_up_off.begin(), e->nw_up_off.end() - 1), pd(e->nw_down_off.begin(), e->nw_down_off.end() - 1);
    for (int k = 0; k < Nn; ++k) up_order[pu[height[s.up[k].parent]]++] = k;
    for (int k = 0; k < E; ++k) down_order[pd[edepth[k]]++] = k;
  }
  HIPCHK(e->d_nw_up_order.alloc(sizeof(int32_t) * Nn)); HIPCHK(e->d_nw_down_order.alloc(sizeof(int32_t) * E));
  HIPCHK(hipMemcpy(e->d_nw_up_order.p, up_order.data(), e->d_nw_up_order.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_down_order.p, down_order.data(), e->d_nw_down_order.bytes, hipMemcpyHostToDevice));
  if (!e->wide) {
    // phm_narrow.hip: the sampling steps themselves (no indirection).  Edges that lead to an INTERNAL node first, grouped by depth
    // level -- the walk propagates states only along those --, then the tip edges; boundaries of the first part in nw_walk_off
    std::vector<phm::DownStep> walk_lv;
    walk_lv.reserve(E);
    e->nw_walk_off.assign(1, 0);
    for (int l = 0; l + 1 < (int)e->nw_down_off.size(); ++l) {
      for (int i = e->nw_down_off[l]; i < e->nw_down_off[l + 1]; ++i)
        if (s.down[down_order[i]].child >= 0) walk_lv.push_back(s.down[down_order[i]]);
      if ((int)walk_lv.size() > e->nw_walk_off.back()) e->nw_walk_off.push_back((int)walk_lv.size());
    }
    for (int i = 0; i < E; ++i) if (s.down[down_order[i]].child < 0) walk_lv.push_back(s.down[down_order[i]]);
    std::vector<int32_t> edge_parent(E);
    for (int i = 0; i < E; ++i) edge_parent[s.down[i].edge] = s.down[i].parent;
    HIPCHK(e->d_nw_down_lv.alloc(sizeof(phm::DownStep) * E));
    HIPCHK(hipMemcpy(e->d_nw_down_lv.p, walk_lv.data(), e->d_nw_down_lv.bytes, hipMemcpyHostToDevice));
    HIPCHK(e->d_nw_walk_off.alloc(sizeof(int32_t) * e->nw_walk_off.size()));
    HIPCHK(hipMemcpy(e->d_nw_walk_off.p, e->nw_walk_off.data(), e->d_nw_walk_off.bytes, hipMemcpyHostToDevice));
    HIPCHK(e->d_nw_edge_parent.alloc(sizeof(int32_t) * E));
    HIPCHK(hipMemcpy(e->d_nw_edge_parent.p, edge_parent.data(), e->d_nw_edge_parent.bytes, hipMemcpyHostToDevice));
  }
  HIPCHK(e->d_nw_up_off.alloc(sizeof(int32_t) * e->nw_up_off.size())); HIPCHK(e->d_nw_down_off.alloc(sizeof(int32_t) * e->nw_down_off.size()));
  HIPCHK(hipMemcpy(e->d_nw_up_off.p, e->nw_up_off.data(), e->d_nw_up_off.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_down_off.p, e->nw_down_off.data(), e->d_nw_down_off.bytes, hipMemcpyHostToDevice));
  return PHM_OK;
}

int32_t narrow_setup(phm_engine* e, const phm_tree* x, const phm_model* model, const phm_options& o, int32_t max_iters) {
  const phm::Schedule& s = e->sched;
  const int E = s.n_edge, T = s.n_tips, Nn = s.n_node, n = e->n, S = e->S;
  // One slot per branch: 1 + Poisson(Omega t_b) segments in stationarity, provisioned far into the tail because a slot
  // has no neighbour to borrow from (default_slot_tail; an overflow is recovered by rebuilding with doubled slots); longer caller-supplied paths get m0 on top.
  const double tail = o.cap_tail > 0.0 ? o.cap_tail : default_slot_tail(S, E, max_iters);
  e->nw_off.assign(E + 1, 0);
  std::vector<int32_t> cap(E);
  int max_cap = 0, n_wide = 0;
  for (int b = 0; b < E; ++b) {
    double tb = 0.0;
    for (int i = x->map_off[b]; i < x->map_off[b + 1]; ++i) tb += x->maps[i];
    const int m0 = x->map_off[b + 1] - x->map_off[b];
    const int q = phm::poisson_capacity(model->Omega * tb, tail);
    cap[b] = (std::max(q, m0 + q - 1) + 2) * e->cap_boost;
    max_cap = std::max(max_cap, cap[b]);
    e->nw_off[b + 1] = e->nw_off[b] + cap[b];
    if (std::max(1.0 + model->Omega * tb, (double)m0) >= phm::NARROW_WIDE_SEGMENTS) ++n_wide;      // a long path: the wave-wide walk (phm_narrow.hip)
  }
  e->nw_n_wide = std::max(n_wide, std::min(E / 16, phm::NARROW_LONG));
  // long chains: the pruning clusters without level barriers (phm_narrow.hip); debug pruning_form 1 / 2 = never / always
  e->nw_cluster_async = e->dbg.pruning_form == 2 || (e->dbg.pruning_form == 0 && n_wide > 0);
  e->nw_total_cap = e->nw_off[E];
  e->nw_klong = max_cap + 1;
  e->rows = e->nw_total_cap;
  std::vector<int32_t> border(E);
  for (int b = 0; b < E; ++b) border[b] = b;
  std::stable_sort(border.begin(), border.end(), [&](int a, int b) { return cap[a] > cap[b]; });

  // tips: [T] shared, or [replica][T]
  if (e->tips_per_replica) {
    e->tips_host.resize((size_t)S * T);
    for (int r = 0; r < S; ++r) for (int t = 0; t < T; ++t) e->tips_host[(size_t)r * T + t] = (uint8_t)(x->states[(size_t)r * T + t] - 1);
  } else {
    e->tips_host.resize(T);
    for (int t = 0; t < T; ++t) e->tips_host[t] = (uint8_t)(x->states[t] - 1);
  }

  const size_t stats_bytes = e->reduce ? sizeof(double) * (size_t)max_iters * e->tiles * e->dcols
                                       : sizeof(double) * (size_t)max_iters * e->dcols * e->S_pad;
  const size_t dw_bytes = sizeof(double) * (size_t)S * e->nw_total_cap;
  const size_t tab = (size_t)e->nw_klong * n * n;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const size_t part_cols = e->wide ? (size_t)n + 1 : (size_t)n + n * n + 1;      // n > 4: counters go through atomics, not per branch
  const size_t n_dw = e->wide ? 3 : 2;             // n <= 4: the branch kernel needs no merged-segment scratch (phm_narrow.hip)
  // transition maps of the sampling sweep: n <= 4 two 16-bit maps per (chain, edge); 5..64 states n bytes per (chain, edge) while that stays below 256 MiB
  const size_t dmap_bytes = e->wide ? (((size_t)S * E * n <= (256u << 20)) ? (size_t)S * E * n : 0) : 2 * sizeof(uint16_t) * (size_t)S * E;
  const size_t small_bytes = (size_t)S * (E * (sizeof(int32_t) + 2) + (size_t)Nn * (sizeof(double) * n + 1)) + sizeof(phm::ClusterNode) * (size_t)Nn + 16 * (size_t)E;   // segment counts, end states, PL, node states, schedules
  const size_t need = n_dw * dw_bytes + (e->wide ? (size_t)S * e->nw_total_cap : 0) + stats_bytes + dmap_bytes + small_bytes +
                      sizeof(double) * (3 * tab + (size_t)S * E * part_cols + (e->wide ? 2 * (size_t)S * e->dcols : 0));
  if (need + (64u << 20) > free_b) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "engine needs %.2f GiB of HBM, %.2f GiB free (reduce n_replicas or max_iters)", need / 1073741824.0, free_b / 1073741824.0);
    return fail(PHM_ERR_OOM, buf);
  }
  HIPCHK(e->d_up.alloc(sizeof(phm::UpStep) * Nn)); HIPCHK(e->d_down.alloc(sizeof(phm::DownStep) * E));
  { int32_t lst = build_level_orders(e); if (lst) return lst; }
  HIPCHK(e->d_nw_border.alloc(sizeof(int32_t) * E)); HIPCHK(e->d_nw_off.alloc(sizeof(int64_t) * (E + 1)));
  if (e->wide) {
    HIPCHK(e->d_nw_colL.alloc(sizeof(double) * tab)); HIPCHK(e->d_nw_rowL.alloc(sizeof(double) * tab));
    HIPCHK(e->d_nw_maskL.alloc(sizeof(double) * (size_t)e->nw_klong * 2 * n));
  } else {      // n <= 4: the three tables in one block (colL | rowL | maskL): a model update is ONE host-to-device copy
    HIPCHK(e->d_nw_colL.alloc(sizeof(double) * (2 * tab + (size_t)e->nw_klong * 2 * n)));
  }
  HIPCHK(e->d_tips.alloc(e->tips_host.size()));
  HIPCHK(e->d_nw_mcount.alloc(sizeof(int32_t) * (size_t)S * E));
  HIPCHK(e->d_nw_dwA.alloc(dw_bytes)); HIPCHK(e->d_nw_dwB.alloc(dw_bytes));
  if (e->wide) {
    HIPCHK(e->d_nw_mstate.alloc((size_t)S * e->nw_total_cap));
    HIPCHK(e->d_nw_mlen.alloc(dw_bytes));
    HIPCHK(e->d_nw_estate.alloc((size_t)S * E * 2));
  }
  HIPCHK(e->d_PL.alloc(sizeof(double) * (size_t)S * Nn * n));
  HIPCHK(e->d_nstate.alloc((size_t)S * Nn));
  HIPCHK(e->d_nw_part.alloc(sizeof(double) * (size_t)S * E * part_cols));
  HIPCHK(e->d_nw_rowbuf.alloc(sizeof(double) * (size_t)S * e->dcols));
  if (!e->wide) {
    HIPCHK(e->d_nw_dmap.alloc(sizeof(uint16_t) * (size_t)S * E));
    HIPCHK(e->d_nw_dmap_edge.alloc(sizeof(uint16_t) * (size_t)S * E));
    phm::ClusterPlan plan;                           // pruning sweep of phm_narrow.hip: subtrees in tiers
    phm::build_cluster_plan(s, phm::NARROW_CLUSTER_NODES, plan);
    e->nw_tier_off = plan.tier_off;
    HIPCHK(e->d_nw_cl_nodes.alloc(sizeof(phm::ClusterNode) * plan.nodes.size()));
    HIPCHK(e->d_nw_cl_item_off.alloc(sizeof(int32_t) * plan.item_off.size()));
    HIPCHK(e->d_nw_cl_lvl_ptr.alloc(sizeof(int32_t) * plan.lvl_ptr.size()));
    HIPCHK(e->d_nw_cl_lvl_off.alloc(sizeof(int32_t) * plan.lvl_off.size()));
    HIPCHK(hipMemcpy(e->d_nw_cl_nodes.p, plan.nodes.data(), e->d_nw_cl_nodes.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_cl_item_off.p, plan.item_off.data(), e->d_nw_cl_item_off.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_cl_lvl_ptr.p, plan.lvl_ptr.data(), e->d_nw_cl_lvl_ptr.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_cl_lvl_off.p, plan.lvl_off.data(), e->d_nw_cl_lvl_off.bytes, hipMemcpyHostToDevice));
  }
  HIPCHK(e->d_stats.alloc(stats_bytes));
  HIPCHK(e->d_err.alloc(sizeof(uint32_t))); HIPCHK(e->d_seg.alloc(sizeof(unsigned long long)));
  if (e->reduce) HIPCHK(e->d_red.alloc(sizeof(double) * (size_t)max_iters * e->dcols));
  e->bytes = (int64_t)(n_dw * dw_bytes + e->d_nw_mstate.bytes + e->d_nw_part.bytes + e->d_PL.bytes + e->d_stats.bytes + e->d_red.bytes +
                       sizeof(double) * 3 * tab + e->d_nw_mcount.bytes + e->d_nw_dmap.bytes + e->d_nw_dmap_edge.bytes + e->d_nw_cl_nodes.bytes +
                       (e->wide ? dmap_bytes : 0));
  HIPCHK(hipMemcpy(e->d_up.p, s.up.data(), e->d_up.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_down.p, s.down.data(), e->d_down.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_border.p, border.data(), e->d_nw_border.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_off.p, e->nw_off.data(), e->d_nw_off.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_tips.p, e->tips_host.data(), e->tips_host.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(e->d_err.p, 0, sizeof(uint32_t)));
  HIPCHK(hipMemset(e->d_seg.p, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemset(e->d_stats.p, 0, stats_bytes));
  HIPCHK(hipMemset(e->d_nstate.p, 0, e->d_nstate.bytes));
  {   // initial paths (makeabranch, src/phylomap.cpp:24-34, :901) into every chain's first buffer
    std::vector<double> init((size_t)e->nw_total_cap, 0.0);
    std::vector<int32_t> m0(E);
    for (int b = 0; b < E; ++b) {
      m0[b] = x->map_off[b + 1] - x->map_off[b];
      std::memcpy(&init[(size_t)e->nw_off[b]], x->maps + x->map_off[b], sizeof(double) * m0[b]);
    }
    for (int r = 0; r < S; ++r) {
      HIPCHK(hipMemcpy(e->d_nw_dwA.as<double>() + (size_t)r * e->nw_total_cap, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_nw_mcount.as<int32_t>() + (size_t)r * E, m0.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
    }
  }
  if (n == 2) fill_narrow_params<2>(e, e->n2, o);
  if (n == 3) fill_narrow_params<3>(e, e->n3, o);
  if (n == 4) fill_narrow_params<4>(e, e->n4, o);
  if (e->wide) {      // 5..64 states: one wave per (replica, branch), lanes = states (phm_wbranch.hip)
    HIPCHK(e->d_wb_cnt.alloc(sizeof(double) * (size_t)S * e->dcols));
    // transition maps of the sampling sweep (n bytes per edge and chain) while they stay small beside the paths; beyond: one launch
    // per depth level
    if ((size_t)S * s.n_edge * n <= (256u << 20)) HIPCHK(e->d_nw_dmap.alloc((size_t)S * s.n_edge * n));
    HIPCHK(hipMemset(e->d_wb_cnt.p, 0, e->d_wb_cnt.bytes));
    HIPCHK(e->d_B2.alloc(sizeof(double) * n * n)); HIPCHK(e->d_Bc.alloc(sizeof(double) * n * n));
    HIPCHK(e->d_ell_col.alloc(sizeof(int32_t) * n * phm::WB_ELL_MAX)); HIPCHK(e->d_ell_val.alloc(sizeof(double) * n * phm::WB_ELL_MAX));
    HIPCHK(e->d_ell2_col.alloc(sizeof(int32_t) * n * phm::WB_ELL_MAX)); HIPCHK(e->d_ell2_val.alloc(sizeof(double) * n * phm::WB_ELL_MAX));
    HIPCHK(e->d_scale.alloc(sizeof(double) * n)); HIPCHK(e->d_pid.alloc(sizeof(double) * n));
    HIPCHK(hipMemcpy(e->d_pid.p, e->hpid.data(), e->d_pid.bytes, hipMemcpyHostToDevice));
    phm::WideBranchParams& p = e->pwb;
    p.n_states = n; p.n_tips = s.n_tips; p.n_node = s.n_node; p.n_edge = s.n_edge; p.root = s.root;
    p.n_rep = e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset; p.n_tiles = e->tiles;
    p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
    p.sparse = (e->variant == PHM_MCMC_SPARSE); p.ks = ks_layout(e->variant); p.tip_masks = hidden_rates(e->variant);
    p.count_self = p.ks; p.reduce = e->reduce; p.n_cols = e->dcols; p.klong = e->nw_klong;
    p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
    p.total_cap = e->nw_total_cap;
    p.B2 = e->d_B2.as<double>(); p.Bc = e->d_Bc.as<double>(); p.scale = e->d_scale.as<double>(); p.pid = e->d_pid.as<double>();
    p.ell_w = 0; p.band_hb = 0; p.ell_col = e->d_ell_col.as<int32_t>(); p.ell_val = e->d_ell_val.as<double>();
    p.ell2_w = 0; p.ell2_col = e->d_ell2_col.as<int32_t>(); p.ell2_val = e->d_ell2_val.as<double>();
    p.up = e->d_up.as<phm::UpStep>(); p.down = e->d_down.as<phm::DownStep>();
    p.up_order = e->d_nw_up_order.as<int32_t>(); p.down_order = e->d_nw_down_order.as<int32_t>();
    p.up_off = e->d_nw_up_off.as<int32_t>(); p.down_off = e->d_nw_down_off.as<int32_t>();
    p.branch_order = e->d_nw_border.as<int32_t>(); p.off = e->d_nw_off.as<int64_t>();
    p.colL = e->d_nw_colL.as<double>(); p.rowL = e->d_nw_rowL.as<double>(); p.maskL = e->d_nw_maskL.as<double>();
    p.tips = e->d_tips.as<uint8_t>(); p.mcount = e->d_nw_mcount.as<int32_t>();
    p.dw[0] = e->d_nw_dwA.as<double>(); p.dw[1] = e->d_nw_dwB.as<double>(); p.mlen = e->d_nw_mlen.as<double>();
    p.mstate = e->d_nw_mstate.as<uint8_t>(); p.estate = e->d_nw_estate.as<uint8_t>();
    p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>(); p.part = e->d_nw_part.as<double>();
    p.cnt = e->d_wb_cnt.as<double>(); p.rowbuf = e->d_nw_rowbuf.as<double>(); p.stats = e->d_stats.as<double>();
    p.dmap = e->d_nw_dmap.as<uint8_t>();
    p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
  }
  return PHM_OK;
}

// copies of a tile's transition counters (phm_tiles.h): enough that 512 counter sets exist however few the tiles
inline int tiles_cnt_copies(int tiles) {
  int c = 1;
  while (c < 64 && c * tiles < 512) c *= 2;
  return c;
}

// Engine state of the wave-per-(tile, branch) mapping (phm_tiles.hip).
template <int NS>
void fill_tile_params(phm_engine* e, phm::TileParams<NS>& p, const phm_options& o) {
  const phm::Schedule& s = e->sched;
  p.n_tips = s.n_tips; p.n_node = s.n_node; p.n_edge = s.n_edge; p.root = s.root;
  p.n_tiles = e->tiles; p.n_rep = e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset;
  p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
  p.ks = ks_layout(e->variant); p.tip_masks = hidden_rates(e->variant); p.reduce = e->reduce; p.n_cols = e->dcols;
  p.klong = e->nw_klong;
  // branches per wave of the branch kernel: one while waves are scarce, up to 16 once there are 65 536 of them anyway
  p.group = (int32_t)std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)e->tiles * s.n_edge / 65536));
  p.n_groups = (s.n_edge + p.group - 1) / p.group;
  p.n_chunks = (p.n_groups + phm::TILES_CHUNK - 1) / phm::TILES_CHUNK;
  p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
  p.rows = e->nw_total_cap;
  for (int i = 0; i < NS * NS; ++i) { p.B2[i] = e->hB2[i]; p.Bc[i] = e->hBc[i]; }
  for (int i = 0; i < NS; ++i) { p.scale[i] = e->hscale[i]; p.pid[i] = e->hpid[i]; }
  p.up = e->d_up.as<phm::UpStep>(); p.down = e->d_down.as<phm::DownStep>();
  p.up_order = e->d_nw_up_order.as<int32_t>(); p.down_order = e->d_nw_down_order.as<int32_t>();
  p.branch_order = e->d_nw_border.as<int32_t>(); p.slot = e->d_tl_slot.as<int32_t>();
  p.cl_nodes = e->d_nw_cl_nodes.as<phm::ClusterNode>();      // NULL unless tiles_setup chose the subtree clusters
  p.cl_lvl_ptr = e->d_nw_cl_lvl_ptr.as<int32_t>(); p.cl_lvl_off = e->d_nw_cl_lvl_off.as<int32_t>();
  p.colL = e->d_nw_colL.as<double>(); p.rowL = e->d_nw_rowL.as<double>(); p.maskL = e->d_nw_maskL.as<double>();
  p.tips = e->d_tips.as<uint8_t>(); p.mcount = e->d_mcount.as<uint16_t>();
  p.dw[0] = e->d_dw0.as<double>(); p.dw[1] = e->d_dw1.as<double>();
  p.estate = e->d_tl_estate.as<uint8_t>(); p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>();
  p.mstate = e->d_wt_mstate.p ? e->d_wt_mstate.as<uint8_t>() : nullptr;      // long paths (tiles_setup)
  p.pdw = e->d_tl_pdw.as<double>(); p.pchunk = e->d_tl_pchunk.as<double>(); p.cnt = e->d_tl_cnt.as<uint32_t>();
  p.cnt_copies = tiles_cnt_copies(e->tiles);
  p.pseg = e->d_tl_pseg.as<uint32_t>(); p.segprev = e->d_tl_segprev.as<uint32_t>();
  p.stats = e->d_stats.as<double>(); p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
}

int32_t tiles_setup(phm_engine* e, const phm_tree* x, const phm_model* model, const phm_options& o, int32_t max_iters) {
  const phm::Schedule& s = e->sched;
  const int E = s.n_edge, T = s.n_tips, Nn = s.n_node, n = e->n, tiles = e->tiles;
  // One slot of rows per branch; a row holds the 64 replicas of the tile, so the slot must take the LARGEST of 64 segment
  // counts: provisioned at default_slot_tail per replica, branch and sweep (1 + Poisson(Omega t_b), plus the caller's initial length).
  const double tail = o.cap_tail > 0.0 ? o.cap_tail : default_slot_tail(e->S, E, max_iters);
  e->tl_slot.assign(E + 1, 0);
  std::vector<int32_t> cap(E);
  int max_cap = 0;
  double max_seg = 0.0;                             // most segments a branch is expected to hold (or holds in the caller's path)
  int64_t rows = 0;
  for (int b = 0; b < E; ++b) {
    double tb = 0.0;
    for (int i = x->map_off[b]; i < x->map_off[b + 1]; ++i) tb += x->maps[i];
    const int m0 = x->map_off[b + 1] - x->map_off[b];
    const int q = phm::poisson_capacity(model->Omega * tb, tail);
    cap[b] = (std::max(q, m0 + q - 1) + 2) * e->cap_boost;
    max_seg = std::max(max_seg, std::max(1.0 + model->Omega * tb, (double)m0));
    if (cap[b] >= (1 << 23)) return fail(PHM_ERR_UNSUPPORTED, "branch too long: a slot of the (tile, branch) mapping exceeds 4 GB");   // 32-bit offsets, phm_tiles.hip
    max_cap = std::max(max_cap, cap[b]);
    rows += cap[b];
    if (rows > 0x7fffff00ll / 64) return fail(PHM_ERR_UNSUPPORTED, "tree too large: dwell rows per replica tile exceed 32-bit indexing");
    e->tl_slot[b + 1] = (int32_t)rows;
  }
  e->nw_total_cap = rows;
  e->nw_klong = max_cap + 1;
  e->rows = rows;
  std::vector<int32_t> border(E);
  for (int b = 0; b < E; ++b) border[b] = b;
  std::stable_sort(border.begin(), border.end(), [&](int a, int b) { return cap[a] > cap[b]; });

  if (e->tips_per_replica) {
    e->tips_host.assign((size_t)tiles * T * 64, 0);
    for (int r = 0; r < e->S_pad; ++r) {
      const int src = r < e->S ? r : e->S - 1;
      for (int t = 0; t < T; ++t) e->tips_host[((size_t)(r / 64) * T + t) * 64 + (r % 64)] = (uint8_t)(x->states[(size_t)src * T + t] - 1);
    }
  } else {
    e->tips_host.resize(T);
    for (int t = 0; t < T; ++t) e->tips_host[t] = (uint8_t)(x->states[t] - 1);
  }

  const int n_chunks = (E + phm::TILES_CHUNK - 1) / phm::TILES_CHUNK;
  const size_t stats_bytes = e->reduce ? sizeof(double) * (size_t)max_iters * tiles * e->dcols
                                       : sizeof(double) * (size_t)max_iters * e->dcols * e->S_pad;
  const size_t dw_bytes = sizeof(double) * (size_t)tiles * rows * 64;
  const size_t tab = (size_t)e->nw_klong * n * n;
  const size_t pdw_bytes = sizeof(double) * (size_t)tiles * E * n * 64;
  const size_t pl_bytes = sizeof(double) * (size_t)tiles * Nn * n * 64;
  // Paths of more than 64 segments on some branch (expected 48: a Poisson count of that mean passes 64 once in 70 draws): the branch
  // kernel without its limit of 64 merged segments per branch and lane -- their states in a byte per (row, lane), an eighth of a dwell buffer
#ifndef PHM_TILES_LONG_SEGMENTS
#define PHM_TILES_LONG_SEGMENTS 48.0
#endif
  const bool long_paths = max_seg > PHM_TILES_LONG_SEGMENTS;
  const size_t ms_bytes = long_paths ? (size_t)tiles * rows * 64 : 0;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const size_t need = 2 * dw_bytes + ms_bytes + pdw_bytes + pl_bytes + stats_bytes + sizeof(double) * 3 * tab + (size_t)tiles * (4 * (size_t)E + Nn) * 64;
  if (need + (64u << 20) > free_b) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "engine needs %.2f GiB of HBM, %.2f GiB free (reduce n_replicas or max_iters)", need / 1073741824.0, free_b / 1073741824.0);
    return fail(PHM_ERR_OOM, buf);
  }
  HIPCHK(e->d_up.alloc(sizeof(phm::UpStep) * Nn)); HIPCHK(e->d_down.alloc(sizeof(phm::DownStep) * E));
  { int32_t lst = build_level_orders(e); if (lst) return lst; }
  HIPCHK(e->d_nw_border.alloc(sizeof(int32_t) * E)); HIPCHK(e->d_tl_slot.alloc(sizeof(int32_t) * (E + 1)));
  HIPCHK(e->d_nw_colL.alloc(sizeof(double) * tab)); HIPCHK(e->d_nw_rowL.alloc(sizeof(double) * tab));
  HIPCHK(e->d_nw_maskL.alloc(sizeof(double) * (size_t)e->nw_klong * 2 * n));
  HIPCHK(e->d_tips.alloc(e->tips_host.size()));
  HIPCHK(e->d_mcount.alloc(sizeof(uint16_t) * (size_t)tiles * E * 64));
  HIPCHK(e->d_dw0.alloc(dw_bytes)); HIPCHK(e->d_dw1.alloc(dw_bytes));
  if (long_paths) HIPCHK(e->d_wt_mstate.alloc(ms_bytes));
  HIPCHK(e->d_tl_estate.alloc((size_t)tiles * E * 64));
  HIPCHK(e->d_PL.alloc(pl_bytes));
  HIPCHK(e->d_nstate.alloc((size_t)tiles * Nn * 64));
  HIPCHK(e->d_tl_pdw.alloc(pdw_bytes));
  HIPCHK(e->d_tl_pchunk.alloc(sizeof(double) * (size_t)tiles * n_chunks * n * 64));
  HIPCHK(e->d_tl_cnt.alloc(sizeof(uint32_t) * (size_t)tiles * tiles_cnt_copies(tiles) * n * n * 64));
  HIPCHK(e->d_tl_pseg.alloc(sizeof(uint32_t) * (size_t)tiles * n_chunks * 64)); HIPCHK(e->d_tl_segprev.alloc(sizeof(uint32_t) * tiles));
  HIPCHK(e->d_stats.alloc(stats_bytes));
  HIPCHK(e->d_err.alloc(sizeof(uint32_t))); HIPCHK(e->d_seg.alloc(sizeof(unsigned long long)));
  if (e->reduce) HIPCHK(e->d_red.alloc(sizeof(double) * (size_t)max_iters * e->dcols));
  e->bytes = (int64_t)(2 * dw_bytes + ms_bytes + pdw_bytes + pl_bytes + e->d_stats.bytes + e->d_red.bytes + e->d_mcount.bytes + e->d_tl_pchunk.bytes +
                       sizeof(double) * 3 * tab);
  HIPCHK(hipMemcpy(e->d_up.p, s.up.data(), e->d_up.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_down.p, s.down.data(), e->d_down.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_border.p, border.data(), e->d_nw_border.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_tl_slot.p, e->tl_slot.data(), e->d_tl_slot.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_tips.p, e->tips_host.data(), e->tips_host.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(e->d_err.p, 0, sizeof(uint32_t)));
  HIPCHK(hipMemset(e->d_seg.p, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemset(e->d_stats.p, 0, stats_bytes));
  HIPCHK(hipMemset(e->d_nstate.p, 0, e->d_nstate.bytes));
  HIPCHK(hipMemset(e->d_tl_cnt.p, 0, e->d_tl_cnt.bytes));
  {   // initial paths -> every replica (makeabranch, src/phylomap.cpp:24-34, :901)
    DevBuf d_off, d_maps;
    HIPCHK(d_off.alloc(sizeof(int32_t) * (E + 1)));
    HIPCHK(d_maps.alloc(sizeof(double) * (size_t)x->map_off[E]));
    HIPCHK(hipMemcpy(d_off.p, x->map_off, d_off.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_maps.p, x->maps, d_maps.bytes, hipMemcpyHostToDevice));
    HIPCHK(phm::launch_tiles_init(E, tiles, rows, e->d_tl_slot.as<int32_t>(), d_off.as<int32_t>(), d_maps.as<double>(),
                                  e->d_dw0.as<double>(), e->d_mcount.as<uint16_t>(), nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint32_t> segprev(tiles);
    for (int t = 0; t < tiles; ++t) segprev[t] = (uint32_t)((int64_t)x->map_off[E] * std::min(64, e->S - t * 64));
    HIPCHK(hipMemcpy(e->d_tl_segprev.p, segprev.data(), e->d_tl_segprev.bytes, hipMemcpyHostToDevice));
  }
  // few tiles (the sites of an alignment): the two tree passes over clusters cut by height, one launch per tier, instead of one per
  // level (phm_tiles.h); phm_debug_options.level_groups: 1 = never, 2 = always, 3 = always, clusters cut by subtree size
  // A DEEP tree (a ladder-like phylogeny: far more height levels than a balanced tree of its size would have) at any tile count:
  // clusters cut by subtree size, a handful of tiers instead of a launch per level and pass (2 000-tip ladder at 8 192 replicas:
  // 4 002 launches, 19.2 ms per sweep -> 19 launches, 9.0 ms; bands of eight levels 10.8).
  e->nw_tier_off.clear();
  int lg2 = 0;
  while ((1 << lg2) < Nn + 1) ++lg2;
  const bool deep = (int)e->nw_up_off.size() - 1 > 4 * lg2 + 32;
  if (e->dbg.level_groups >= 2 || (e->dbg.level_groups == 0 && (deep || (int64_t)tiles * Nn <= phm::TILES_CL_MAX_WORK))) {
    phm::ClusterPlan plan;
    if (e->dbg.level_groups == 3 || (e->dbg.level_groups == 0 && deep)) phm::build_cluster_plan(s, phm::TILES_CL_NODES, plan);      // subtrees by size
    else phm::build_band_plan(s, TILES_CL_BAND, plan);
    e->nw_tier_off = plan.tier_off;
    HIPCHK(e->d_nw_cl_nodes.alloc(sizeof(phm::ClusterNode) * plan.nodes.size()));
    HIPCHK(e->d_nw_cl_lvl_ptr.alloc(sizeof(int32_t) * plan.lvl_ptr.size()));
    HIPCHK(e->d_nw_cl_lvl_off.alloc(sizeof(int32_t) * plan.lvl_off.size()));
    HIPCHK(hipMemcpy(e->d_nw_cl_nodes.p, plan.nodes.data(), e->d_nw_cl_nodes.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_cl_lvl_ptr.p, plan.lvl_ptr.data(), e->d_nw_cl_lvl_ptr.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nw_cl_lvl_off.p, plan.lvl_off.data(), e->d_nw_cl_lvl_off.bytes, hipMemcpyHostToDevice));
  }
  if (n == 2) fill_tile_params<2>(e, e->t2, o);
  if (n == 3) fill_tile_params<3>(e, e->t3, o);
  if (n == 4) fill_tile_params<4>(e, e->t4, o);
  return PHM_OK;
}

// Engine state of the lane-per-replica mapping for 5..64 states (phm_wtiles.hip): the slot layout of tiles_setup, the
// statistics in integer accumulators per tile.
int32_t wtiles_setup(phm_engine* e, const phm_tree* x, const phm_model* model, const phm_options& o, int32_t max_iters) {
  const phm::Schedule& s = e->sched;
  const int E = s.n_edge, T = s.n_tips, Nn = s.n_node, n = e->n, tiles = e->tiles;
  const double tail = o.cap_tail > 0.0 ? o.cap_tail : default_slot_tail(e->S, E, max_iters);
  e->tl_slot.assign(E + 1, 0);
  std::vector<int32_t> cap(E);
  int max_cap = 0;
  int64_t rows = 0;
  double tree_len = 0.0;
  for (int b = 0; b < E; ++b) {
    double tb = 0.0;
    for (int i = x->map_off[b]; i < x->map_off[b + 1]; ++i) tb += x->maps[i];
    tree_len += tb;
    const int m0 = x->map_off[b + 1] - x->map_off[b];
    const int q = phm::poisson_capacity(model->Omega * tb, tail);
    cap[b] = (std::max(q, m0 + q - 1) + 2) * e->cap_boost;
    if (cap[b] >= (1 << 23)) return fail(PHM_ERR_UNSUPPORTED, "branch too long: a slot of the (tile, branch) mapping exceeds 4 GB");
    max_cap = std::max(max_cap, cap[b]);
    rows += cap[b];
    if (rows > 0x7fffff00ll / 64) return fail(PHM_ERR_UNSUPPORTED, "tree too large: dwell rows per replica tile exceed 32-bit indexing");
    e->tl_slot[b + 1] = (int32_t)rows;
  }
  e->nw_total_cap = rows;
  e->nw_klong = max_cap + 1;
  e->rows = rows;
  std::vector<int32_t> border(E);
  for (int b = 0; b < E; ++b) border[b] = b;
  std::stable_sort(border.begin(), border.end(), [&](int a, int b) { return cap[a] > cap[b]; });

  if (e->tips_per_replica) {
    e->tips_host.assign((size_t)tiles * T * 64, 0);
    for (int r = 0; r < e->S_pad; ++r) {
      const int src = r < e->S ? r : e->S - 1;
      for (int t = 0; t < T; ++t) e->tips_host[((size_t)(r / 64) * T + t) * 64 + (r % 64)] = (uint8_t)(x->states[(size_t)src * T + t] - 1);
    }
  } else {
    e->tips_host.resize(T);
    for (int t = 0; t < T; ++t) e->tips_host[t] = (uint8_t)(x->states[t] - 1);
  }

  const int ldt = (n + 1) & ~1;
  if ((uint64_t)e->nw_klong * n * ldt * sizeof(double) >= (1ull << 32))      // the kernels address a chain table with 32-bit byte offsets
    return fail(PHM_ERR_UNSUPPORTED, "branch too long for the lane-per-replica mapping: a chain table would exceed 4 GB");
  const size_t stats_bytes = e->reduce ? sizeof(double) * (size_t)max_iters * tiles * e->dcols
                                       : sizeof(double) * (size_t)max_iters * e->dcols * e->S_pad;
  const size_t dw_bytes = sizeof(double) * (size_t)tiles * rows * 64;
  const size_t tab = (size_t)e->nw_klong * n * ldt;
  const size_t pl_bytes = sizeof(double) * (size_t)tiles * Nn * n * 64;
  const size_t cnt_bytes = sizeof(uint32_t) * (size_t)tiles * n * n * 64;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const int nblk_need = (n + 7) / 8, ldb_need = (nblk_need + 1) & ~1;
  const size_t tot_bytes = sizeof(double) * (size_t)e->nw_klong * n * n * ldb_need;                     // blkL: running sums of the forward draws
  const size_t acc_bytes = sizeof(unsigned long long) * (size_t)tiles * (n + 1) * 64;                   // dwfx + segacc
  const size_t red_bytes = e->reduce ? sizeof(double) * (size_t)max_iters * e->dcols : 0;
  const size_t need = 2 * dw_bytes + dw_bytes / 8 + pl_bytes + cnt_bytes + stats_bytes + tot_bytes + acc_bytes + red_bytes + sizeof(double) * 3 * tab +
                      (size_t)tiles * (5 * (size_t)E + Nn + 8 * (size_t)n) * 64;
  if (need + (64u << 20) > free_b) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "engine needs %.2f GiB of HBM, %.2f GiB free (reduce n_replicas or max_iters)", need / 1073741824.0, free_b / 1073741824.0);
    return fail(PHM_ERR_OOM, buf);
  }
  HIPCHK(e->d_up.alloc(sizeof(phm::UpStep) * Nn)); HIPCHK(e->d_down.alloc(sizeof(phm::DownStep) * E));
  { int32_t lst = build_level_orders(e); if (lst) return lst; }
  // A DEEP tree (tiles_setup has the rule): the band pruning kernel and the node draws for n <= 32 run over subtree clusters, a launch per
  // tier instead of one per level and pass (phm_wtiles.hip); phm_debug_options.level_groups: 1 = never, 2 / 3 = always
  e->nw_tier_off.clear();
  {
    int lg2 = 0;
    while ((1 << lg2) < Nn + 1) ++lg2;
    const bool deep = (int)e->nw_up_off.size() - 1 > 4 * lg2 + 32;
    if (n <= 32 && (e->dbg.level_groups >= 2 || (e->dbg.level_groups == 0 && deep))) {
      phm::ClusterPlan plan;
      phm::build_cluster_plan(s, phm::TILES_CL_NODES, plan);
      e->nw_tier_off = plan.tier_off;
      HIPCHK(e->d_nw_cl_nodes.alloc(sizeof(phm::ClusterNode) * plan.nodes.size()));
      HIPCHK(e->d_nw_cl_lvl_ptr.alloc(sizeof(int32_t) * plan.lvl_ptr.size()));
      HIPCHK(e->d_nw_cl_lvl_off.alloc(sizeof(int32_t) * plan.lvl_off.size()));
      HIPCHK(hipMemcpy(e->d_nw_cl_nodes.p, plan.nodes.data(), e->d_nw_cl_nodes.bytes, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_nw_cl_lvl_ptr.p, plan.lvl_ptr.data(), e->d_nw_cl_lvl_ptr.bytes, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_nw_cl_lvl_off.p, plan.lvl_off.data(), e->d_nw_cl_lvl_off.bytes, hipMemcpyHostToDevice));
    }
  }
  HIPCHK(e->d_nw_border.alloc(sizeof(int32_t) * E)); HIPCHK(e->d_tl_slot.alloc(sizeof(int32_t) * (E + 1)));
  HIPCHK(e->d_nw_colL.alloc(sizeof(double) * tab)); HIPCHK(e->d_nw_rowL.alloc(sizeof(double) * tab));
  HIPCHK(e->d_nw_maskL.alloc(sizeof(double) * (size_t)e->nw_klong * 2 * ldt));
  HIPCHK(e->d_wt_B2.alloc(sizeof(double) * (size_t)n * ldt)); HIPCHK(e->d_Bc.alloc(sizeof(double) * n * n));
  HIPCHK(e->d_wt_B2band.alloc(sizeof(double) * (size_t)n * (2 * phm::WT_BAND_MAX + 1)));
  const int nblk = (n + 7) / 8, ldb = (nblk + 1) & ~1;
  HIPCHK(e->d_wt_totL.alloc(sizeof(double) * (size_t)e->nw_klong * n * n * ldb));
  HIPCHK(e->d_wt_pair_slot.alloc(sizeof(int16_t) * (size_t)n * n)); HIPCHK(e->d_wt_slot_col.alloc(sizeof(int32_t) * phm::WT_MAX_SLOTS));
  HIPCHK(e->d_scale.alloc(sizeof(double) * n)); HIPCHK(e->d_pid.alloc(sizeof(double) * n));
  HIPCHK(hipMemcpy(e->d_pid.p, e->hpid.data(), e->d_pid.bytes, hipMemcpyHostToDevice));
  HIPCHK(e->d_tips.alloc(e->tips_host.size()));
  HIPCHK(e->d_mcount.alloc(sizeof(uint16_t) * (size_t)tiles * E * 64));
  HIPCHK(e->d_dw0.alloc(dw_bytes)); HIPCHK(e->d_dw1.alloc(dw_bytes));
  HIPCHK(e->d_tl_estate.alloc(sizeof(uint16_t) * (size_t)tiles * E * 64));
  HIPCHK(e->d_wt_mstate.alloc((size_t)tiles * rows * 64));
  HIPCHK(e->d_PL.alloc(pl_bytes));
  HIPCHK(e->d_nstate.alloc((size_t)tiles * Nn * 64));
  HIPCHK(e->d_tl_cnt.alloc(cnt_bytes));
  HIPCHK(e->d_wt_dwfx.alloc(sizeof(unsigned long long) * (size_t)tiles * n * 64));
  HIPCHK(e->d_wt_segacc.alloc(sizeof(unsigned long long) * (size_t)tiles * 64));
  if (e->reduce) {      // statistics summed over replicas: per-tile totals the branch kernel adds to directly (phm_wtiles.hip)
    HIPCHK(e->d_wt_dwfx_tile.alloc(sizeof(unsigned long long) * (size_t)tiles * n * 16));
    HIPCHK(e->d_wt_cnt_tile.alloc(sizeof(uint32_t) * (size_t)tiles * n * n));
    HIPCHK(hipMemset(e->d_wt_dwfx_tile.p, 0, e->d_wt_dwfx_tile.bytes));
    HIPCHK(hipMemset(e->d_wt_cnt_tile.p, 0, e->d_wt_cnt_tile.bytes));
  }
  HIPCHK(e->d_stats.alloc(stats_bytes));
  HIPCHK(e->d_err.alloc(sizeof(uint32_t))); HIPCHK(e->d_seg.alloc(sizeof(unsigned long long)));
  if (e->reduce) HIPCHK(e->d_red.alloc(sizeof(double) * (size_t)max_iters * e->dcols));
  e->bytes = (int64_t)(2 * dw_bytes + e->d_wt_mstate.bytes + pl_bytes + cnt_bytes + e->d_stats.bytes + e->d_red.bytes + e->d_mcount.bytes + e->d_tl_estate.bytes +
                       e->d_nstate.bytes + e->d_wt_dwfx.bytes + e->d_wt_segacc.bytes + e->d_wt_totL.bytes + sizeof(double) * 3 * tab);
  HIPCHK(hipMemcpy(e->d_up.p, s.up.data(), e->d_up.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_down.p, s.down.data(), e->d_down.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_nw_border.p, border.data(), e->d_nw_border.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_tl_slot.p, e->tl_slot.data(), e->d_tl_slot.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_tips.p, e->tips_host.data(), e->tips_host.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(e->d_err.p, 0, sizeof(uint32_t)));
  HIPCHK(hipMemset(e->d_seg.p, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemset(e->d_stats.p, 0, stats_bytes));
  HIPCHK(hipMemset(e->d_nstate.p, 0, e->d_nstate.bytes));
  HIPCHK(hipMemset(e->d_tl_cnt.p, 0, e->d_tl_cnt.bytes));
  HIPCHK(hipMemset(e->d_wt_dwfx.p, 0, e->d_wt_dwfx.bytes));
  HIPCHK(hipMemset(e->d_wt_segacc.p, 0, e->d_wt_segacc.bytes));
  {   // initial paths -> every replica (makeabranch, src/phylomap.cpp:24-34, :901)
    DevBuf d_off, d_maps;
    HIPCHK(d_off.alloc(sizeof(int32_t) * (E + 1)));
    HIPCHK(d_maps.alloc(sizeof(double) * (size_t)x->map_off[E]));
    HIPCHK(hipMemcpy(d_off.p, x->map_off, d_off.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_maps.p, x->maps, d_maps.bytes, hipMemcpyHostToDevice));
    HIPCHK(phm::launch_tiles_init(E, tiles, rows, e->d_tl_slot.as<int32_t>(), d_off.as<int32_t>(), d_maps.as<double>(),
                                  e->d_dw0.as<double>(), e->d_mcount.as<uint16_t>(), nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  phm::WtParams& p = e->pwt;
  p.n_states = n; p.ldt = ldt;
  p.n_tips = s.n_tips; p.n_node = s.n_node; p.n_edge = s.n_edge; p.root = s.root;
  p.n_tiles = tiles; p.n_rep = e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset;
  p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
  p.ks = ks_layout(e->variant); p.tip_masks = hidden_rates(e->variant); p.reduce = e->reduce; p.n_cols = e->dcols;
  p.klong = e->nw_klong;
  // branches per wave of the branch kernel: one while waves are scarce, up to 8 once there are 65 536 of them anyway
  p.group = (int32_t)std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)tiles * E / 65536));
  if (e->dbg.branch_group > 0) p.group = std::min(64, (int)e->dbg.branch_group);      // measurement aid (phm_debug_options)
  p.n_groups = (E + p.group - 1) / p.group;
  p.up_form = e->dbg.pruning_form & 3; p.band_up = 0; p.band_draw = 0; p.B2band = e->d_wt_B2band.as<double>();
  p.cl_nodes = e->nw_tier_off.empty() ? nullptr : e->d_nw_cl_nodes.as<phm::ClusterNode>();
  p.cl_lvl_ptr = e->d_nw_cl_lvl_ptr.as<int32_t>(); p.cl_lvl_off = e->d_nw_cl_lvl_off.as<int32_t>();
  e->sparse_req = o.sparse_chains;
  p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
  p.rows = rows;
  {   // dwell accumulators: 64-bit fixed point, a replica's column never exceeds the tree length
    int ex = 0;
    (void)std::frexp(std::max(tree_len, 1.0), &ex);       // tree_len < 2^ex
    p.fx_scale = std::ldexp(1.0, 60 - ex); p.fx_inv = std::ldexp(1.0, ex - 60);      // four lanes' sums share an accumulator in the reduced output: < 2^62
  }
  p.B2 = e->d_wt_B2.as<double>(); p.Bc = e->d_Bc.as<double>(); p.scale = e->d_scale.as<double>(); p.pid = e->d_pid.as<double>();
  p.up = e->d_up.as<phm::UpStep>(); p.down = e->d_down.as<phm::DownStep>();
  p.up_order = e->d_nw_up_order.as<int32_t>(); p.down_order = e->d_nw_down_order.as<int32_t>();
  p.branch_order = e->d_nw_border.as<int32_t>(); p.slot = e->d_tl_slot.as<int32_t>();
  p.colL = e->d_nw_colL.as<double>(); p.rowL = e->d_nw_rowL.as<double>(); p.maskL = e->d_nw_maskL.as<double>();
  p.blkL = e->d_wt_totL.as<double>(); p.nblk = nblk; p.ldb = ldb;
  p.n_slots = 0; p.pair_slot = e->d_wt_pair_slot.as<int16_t>(); p.slot_col = e->d_wt_slot_col.as<int32_t>();
  p.tips = e->d_tips.as<uint8_t>(); p.mcount = e->d_mcount.as<uint16_t>();
  p.dw[0] = e->d_dw0.as<double>(); p.dw[1] = e->d_dw1.as<double>();
  p.estate = e->d_tl_estate.as<uint16_t>(); p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>();
  p.mstate = e->d_wt_mstate.as<uint8_t>();
  p.dwfx = e->d_wt_dwfx.as<unsigned long long>(); p.cnt = e->d_tl_cnt.as<uint32_t>();
  p.segacc = e->d_wt_segacc.as<unsigned long long>();
  p.dwfx_tile = e->reduce ? e->d_wt_dwfx_tile.as<unsigned long long>() : nullptr;
  p.cnt_tile = e->reduce ? e->d_wt_cnt_tile.as<uint32_t>() : nullptr;
  p.stats = e->d_stats.as<double>(); p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
  return PHM_OK;
}

// the engine an entry point works on: the rebuilt one after a capacity recovery
phm_engine* live(phm_engine* e) {
  while (e && e->fwd) e = e->fwd;
  return e;
}
// a handle whose capacity recovery failed has no device state left (recover_capacity)
inline int32_t dead_engine() { return fail(PHM_ERR_CAPACITY, "the engine outgrew its dwell capacity and could not be rebuilt; destroy it and create a new one (larger cap_tail margin, fewer replicas)"); }

std::shared_ptr<SavedInput> save_input(const phm_tree* trees, int32_t n_trees, const phm_model* model, const phm_options& o, int32_t max_iters) {
  auto sv = std::make_shared<SavedInput>();
  const int n = model->n_states;
  sv->trees.resize(n_trees);
  sv->flat.resize(n_trees);
  for (int j = 0; j < n_trees; ++j) {
    const phm_tree& x = trees[j];
    SavedInput::TreeCopy& c = sv->trees[j];
    const size_t E = (size_t)x.n_edge, nt = (size_t)x.n_tips * (o.tips_per_replica ? std::max(1, o.n_replicas) : 1);
    c.edge.assign(x.edge, x.edge + 2 * E);
    c.states.assign(x.states, x.states + nt);
    c.map_off.assign(x.map_off, x.map_off + E + 1);
    c.maps.assign(x.maps, x.maps + x.map_off[E]);
    c.mapnames.assign(x.mapnames, x.mapnames + x.map_off[E]);
    if (x.edge_length) c.edge_length.assign(x.edge_length, x.edge_length + E);
    c.t = x;
    c.t.edge = c.edge.data(); c.t.states = c.states.data(); c.t.map_off = c.map_off.data(); c.t.maps = c.maps.data();
    c.t.mapnames = c.mapnames.data(); c.t.edge_length = x.edge_length ? c.edge_length.data() : nullptr;
    sv->flat[j] = c.t;
  }
  sv->Q.assign(model->Q, model->Q + (size_t)n * n);
  sv->pid.assign(model->pid, model->pid + n);
  if (model->B) sv->B.assign(model->B, model->B + (size_t)n * n);
  sv->model = *model;
  sv->model.Q = sv->Q.data(); sv->model.pid = sv->pid.data(); sv->model.B = model->B ? sv->B.data() : nullptr;
  sv->opt = o;
  sv->max_iters = max_iters;
  return sv;
}

}  // namespace

extern "C" {

int32_t phm_version(void) { return PHM_VERSION; }

int32_t phm_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(phm_options);
    case 1: return (int32_t)sizeof(phm_info);
    case 2: return (int32_t)sizeof(phm_tree);
    case 3: return (int32_t)sizeof(phm_model);
    case 4: return (int32_t)sizeof(phm_debug_options);
    default: return -1;
  }
}

int32_t phm_set_debug_options(const phm_debug_options* dbg) {
  if (dbg) g_phm_debug = *dbg; else g_phm_debug = phm_debug_options{};
  return PHM_OK;
}

int32_t phm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* phm_last_error(void) { return g_phm_err.c_str(); }

const char* phm_status_string(int32_t s) {
  switch (s) {
    case PHM_OK: return "ok";
    case PHM_ERR_BAD_INPUT: return "bad input";
    case PHM_ERR_UNSUPPORTED: return "unsupported";
    case PHM_ERR_NO_DEVICE: return "no HIP device / HIP failure";
    case PHM_ERR_OOM: return "out of device memory";
    case PHM_ERR_ZERO_PROB: return "zero probability vector";
    case PHM_ERR_CAPACITY: return "branch capacity exceeded";
    case PHM_ERR_UNIF_CAP: return "uniformisation jump cap exceeded";
    case PHM_ERR_STATE: return "invalid engine state";
    default: return "unknown";
  }
}

int32_t phm_engine_create(const phm_tree* x, const phm_model* model, const phm_options* opt_in, int32_t max_iters,
                          phm_engine** out) {
  return phm_engine_create_multi(x, 1, model, opt_in, max_iters, out);
}

int32_t phm_engine_create_multi(const phm_tree* trees, int32_t n_trees, const phm_model* model, const phm_options* opt_in,
                                int32_t max_iters, phm_engine** out) {
  return phm_engine_create_impl(trees, n_trees, model, opt_in, g_phm_debug, 0, max_iters, out);
}

}  // extern "C"

// The one constructor behind phm_engine_create / _multi, the multi-device one-shot calls (worker threads pass the CALLER's
// debug options) and the capacity recovery (boost_log2: log2 of the multiplier applied to the provisioned capacities).
int32_t phm_engine_create_impl(const phm_tree* trees, int32_t n_trees, const phm_model* model, const phm_options* opt_in,
                               const phm_debug_options& dbg, int boost_log2, int32_t max_iters, phm_engine** out) {
  if (!out) return fail(PHM_ERR_BAD_INPUT, "out is NULL");
  *out = nullptr;
  if (!trees || !model) return fail(PHM_ERR_BAD_INPUT, "tree/model is NULL");
  if (n_trees < 1) return fail(PHM_ERR_BAD_INPUT, "n_trees must be >= 1");
  const phm_tree* x = trees;
  phm_options o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opt_in) o = *opt_in;
  if (o.n_replicas <= 0) o.n_replicas = 1;
  const int n = model->n_states;
  if (n < 2) return fail(PHM_ERR_BAD_INPUT, "n_states must be >= 2");
  if (n > 64) return fail(PHM_ERR_UNSUPPORTED, "this build has MCMC kernels for n_states <= 64 only");
  if (!model->Q || !model->pid) return fail(PHM_ERR_BAD_INPUT, "model: Q/pid missing");
  if (model->variant < PHM_MCMC || model->variant > PHM_MCMC_KSMT) return fail(PHM_ERR_BAD_INPUT, "unknown variant");
  if (hidden_rates(model->variant) && (n & 1)) return fail(PHM_ERR_BAD_INPUT, "sumstatMCMCks needs a hidden-rates Q of even size n = 2k+2 (src/phylomap.cpp:1820)");
  // The bf SWEEP (treesamplebf :1169-1179: sampleinternalnodesMCMCbf :1091-1163, sampleabranchbf :1031-1074, shortenerbf :997-1028)
  // is written for any n; what hard-wires two states is around it: the 9-column matrix and root column 8 of the driver
  // (:1129, :1293) and the rate updates (:1181-1253) -- those restrictions live in phm_maketreelistMCMCbf / 2sDICt.  Here PHM_MCMC_BF
  // with n states gives n dwell sums, n x n counts incl. self pairs, Q[0,1], Q[1,0] and the root state in column n + n*n + 2
  // (= 8 at n = 2).  The multi-tree sweep keeps n = 2 (recordQmtNS :2169-2173 writes columns 6 and 7).
  if (model->variant == PHM_MCMC_MT && n != 2) return fail(PHM_ERR_BAD_INPUT, "sumstatMCMCmt is a two-state model (hard-wired columns, src/phylomap.cpp:2169-2173)");
  if (n_trees > 1 && (o.reduce || o.tips_per_replica)) return fail(PHM_ERR_UNSUPPORTED, "a list of trees takes neither reduce nor tips_per_replica");
  if (max_iters < 1) return fail(PHM_ERR_BAD_INPUT, "max_iters must be >= 1");
  if (!(model->Omega > 0.0) || !std::isfinite(model->Omega)) return fail(PHM_ERR_BAD_INPUT, "Omega must be positive");
  int32_t st = PHM_OK;
  for (int j = 0; j < n_trees; ++j) {
    if (trees[j].n_tips != x->n_tips || trees[j].n_edge != x->n_edge || trees[j].n_node != x->n_node)
      return fail(PHM_ERR_BAD_INPUT, "every tree of the list must have the same number of tips and edges (src/phylomap.cpp:2275-2287)");
    st = validate_tree_paths(&trees[j], n, o.tips_per_replica ? o.n_replicas : 1);
    if (st) return st;
  }

  // model matrices, row-major copies (inputs are R's column-major)
  std::vector<double> B2v, Bcv, scalev, pidv(n), qp;
  st = compute_model(model->variant, n, model->Q, model->B, model->Omega, B2v, Bcv, scalev, qp);
  if (st) return st;
  for (int i = 0; i < n; ++i) {
    pidv[i] = model->pid[i];
    if (!(pidv[i] >= 0.0) || !std::isfinite(pidv[i])) return fail(PHM_ERR_BAD_INPUT, "pid must be non-negative");
  }
  double *B2 = B2v.data(), *Bc = Bcv.data(), *scale = scalev.data(), *pid = pidv.data();

  phm_engine* e = new phm_engine();
  std::unique_ptr<phm_engine> guard(e);
  e->n = n; e->cols = n + n * (n - 1); e->variant = model->variant;
  e->dcols = e->cols;
  e->wide = n > 4;
  if (ks_layout(e->variant)) {
    const int k = hidden_rates(e->variant) ? n / 2 - 1 : 0;
    e->cols = n + n * n + 2 + 3 * k + 1;        // man/sumstatMCMCks.Rd:19; bf: src/phylomap.cpp:1293
    e->dcols = n + n * n + 1;
  }
  e->qparams = qp;
  e->Omega = model->Omega;
  e->hB2 = B2v; e->hBc = Bcv; e->hscale = scalev; e->hpid = pidv;
  e->n_trees = n_trees; e->S_tree = o.n_replicas; e->tpt = (o.n_replicas + 63) / 64;
  e->S = n_trees * o.n_replicas; e->tiles = n_trees * e->tpt; e->S_pad = e->tiles * 64;
  e->max_iters = max_iters; e->reduce = o.reduce ? 1 : 0;
  e->ipl = o.iters_per_launch > 0 ? o.iters_per_launch : 8;
  e->tips_per_replica = o.tips_per_replica != 0 || n_trees > 1;      // a list of trees: tip data per tile
  e->dbg = dbg;
  e->phase_timing = dbg.phase_timing != 0;
  // rescale_pruning with a fixed-Q MCMC variant: the pruning pass of sumstatMCMC / SPARSEsumstatMCMC rescaled like _bigtree's (:525)
  e->normalise = normalised_variant(e->variant) || (o.rescale_pruning != 0 && (e->variant == PHM_MCMC || e->variant == PHM_MCMC_SPARSE));
  e->cap_boost = 1 << std::max(0, std::min(10, boost_log2));      // set by the capacity recovery
  e->recover = o.no_recovery == 0;
  if (e->recover) { e->saved = save_input(trees, n_trees, model, o, max_iters); e->saved->dbg = dbg; }      // the replay needs the caller's inputs; nothing is kept otherwise

  std::string serr;
  e->scheds.resize(n_trees);
  for (int j = 0; j < n_trees; ++j)
    if (!phm::build_schedule(x->n_tips, x->n_node, x->n_edge, trees[j].edge, e->scheds[j], serr)) return fail(PHM_ERR_BAD_INPUT, "tree: " + serr);
  e->sched = e->scheds[0];
  phm::Schedule& s = e->sched;
  st = select_device(o.device);          // every input check above runs without a device
  if (st) return st;
  HIPCHK(hipGetDevice(&e->device));
  const int E = s.n_edge, T = s.n_tips;

  // Mapping of the sweep onto lanes (phm_options.mapping): with few
  // chains the replica mapping would leave all but a handful of lanes idle and walk the tree sequentially.
  const bool small_n = !e->wide && n_trees == 1;
  const int map_req = o.mapping;
  const bool auto_map = map_req == PHM_MAP_AUTO && o.storage == 0;      // a ring / two-buffer request names the replica layout
  if ((map_req == 2 || map_req == 3) && n_trees != 1) return fail(PHM_ERR_UNSUPPORTED, "the branch-parallel mappings take a single tree");
  if (e->wide) {      // 5..64 states: lane = replica, wave per (tile, item) (phm_wtiles.hip); a handful of chains: wave per (replica, branch)
    e->tiled = n_trees == 1 && (map_req == 3 || (auto_map && e->S > wbranch_auto_max_replicas(s)));
    e->narrow = n_trees == 1 && !e->tiled && (map_req == 2 || auto_map);
  } else {
    int narrow_cap = narrow_auto_max_replicas(s);
    bool long_paths = false;             // a branch expected to hold more than 64 segments: the replica mapping's lane-sequential loop is no place for it
    if (small_n && auto_map) {
      // Long paths: the (tile, branch) mapping cannot end a sweep before ONE wave has walked the longest branch twice (its two passes,
      // ~1.35 us per segment; the reference's squamate tree at Omega = 10, 2 280 segments on one branch: 3.1 ms per sweep up to 64
      // chains, 3.7 at 256), while the branch mapping costs 4.4e-8 ms per segment and chain (0.68 ms at 8 chains, 2.8 at 64, 10.2 at
      // 256; profiles/r04_probe_squamate_crossover.log).  S* = floor / slope.
      double max_seg = 0.0, tot_seg = 0.0;
      for (int b = 0; b < E; ++b) {
        double tb = 0.0;
        for (int i = x->map_off[b]; i < x->map_off[b + 1]; ++i) tb += x->maps[i];
        const double seg = std::max(1.0 + model->Omega * tb, (double)(x->map_off[b + 1] - x->map_off[b]));
        max_seg = std::max(max_seg, seg); tot_seg += seg;
      }
      if (max_seg > 64.0) narrow_cap = std::max(narrow_cap, (int)std::min(65535.0, 1.35e-3 * max_seg / (4.4e-8 * tot_seg)));
      long_paths = max_seg > 64.0;
    }
    e->narrow = small_n && (map_req == 2 || (auto_map && e->S <= narrow_cap));
    e->tiled = small_n && !e->narrow && (map_req == 3 || (auto_map && (long_paths || e->S <= TILES_AUTO_MAX_REPLICAS)));      // (no room: the replica layout, below)
  }
  if (e->narrow && e->S > 65535) {      // the replica index is the grid's y dimension in these kernels
    if (!auto_map) return fail(PHM_ERR_UNSUPPORTED, "the one-lane-per-branch / wave-per-(replica, branch) mappings take at most 65 535 replicas");
    e->narrow = false; e->tiled = e->wide || e->S <= TILES_AUTO_MAX_REPLICAS;
  }
  if (e->narrow) {
    st = narrow_setup(e, x, model, o, max_iters);
    if (st == PHM_ERR_OOM && auto_map) { e->narrow = false; e->tiled = true; st = PHM_OK; }
  }
  if (e->tiled) {
    st = e->wide ? wtiles_setup(e, x, model, o, max_iters) : tiles_setup(e, x, model, o, max_iters);
    if (st == PHM_ERR_OOM && auto_map) { e->tiled = false; st = PHM_OK; }     // automatic choice: fall back to the replica layout
  }
  if (e->narrow || e->tiled) {
    if (st) return st;
    st = upload_model(e);
    if (st) return st;
    HIPCHK(hipEventCreate(&e->ev0));
    HIPCHK(hipEventCreate(&e->ev1));
    *out = guard.release();
    return PHM_OK;
  }

  // Capacity of a tile's dwell stream.  Branch b holds 1 + Poisson(Omega t_b) segments in stationarity
  // (t_b = sum(x$maps[[b]])) and occupies max-over-64-lanes rows; provision the per-branch quantile at
  // `cap_tail` (default 1e-3, i.e. beyond the expected maximum of 64 draws) and check at run time.
  int64_t rows = 0;
  std::vector<std::vector<int32_t>> init_row(n_trees, std::vector<int32_t>(E));
  std::vector<int64_t> init_rows(n_trees, 0);
  for (int j = 0; j < n_trees; ++j) {
    const phm_tree* xt = &trees[j];
    int64_t rows_j = 0, max_q = 0;
    double sum_lambda = 0.0;
    for (int k = 0; k < E; ++k) {
      const phm::DownStep& d = e->scheds[j].down[k];
      double tb = 0.0;
      for (int i = xt->map_off[d.edge]; i < xt->map_off[d.edge + 1]; ++i) tb += xt->maps[i];
      int m0 = xt->map_off[d.edge + 1] - xt->map_off[d.edge];
      if (e->wide && m0 > phm::wide_maxseg(n))
        return fail(PHM_ERR_UNSUPPORTED, "the replica mapping for 5..64 states holds at most " + std::to_string(phm::wide_maxseg(n)) +
                                         " segments per branch (a path of " + std::to_string(m0) + " was given); use PHM_MAP_TILES or PHM_MAP_BRANCHES (the automatic choice for one tree)");
      init_row[j][k] = (int32_t)init_rows[j];
      init_rows[j] += m0;
      // a sweep keeps at most the m merged segments it was given and adds Poisson(<= Omega t_b) virtual jumps, so a
      // caller-supplied path longer than the stationary quantile (e.g. 100 equal pieces) needs m0 + that quantile
      int q = phm::poisson_capacity(model->Omega * tb, o.cap_tail > 0.0 ? o.cap_tail : 1e-3);
      rows_j += (int64_t)std::max(q, m0 + q - 1) * e->cap_boost;
      max_q = std::max<int64_t>(max_q, (int64_t)std::max(q, m0 + q - 1) * e->cap_boost);
      sum_lambda += model->Omega * tb;
    }
    // Ring capacity: the stream being read, plus head-room for the stream being written behind it.  While branch k is
    // processed its input rows are still occupied and its output rows are already being written (one full branch of
    // slack), and rows written so far minus rows freed so far performs a random walk whose standard deviation is about
    // 0.7 sqrt(sum lambda) (wave-maximum of 64 Poisson counts per branch); 6 sigma of that on top.
    rows_j = std::max(rows_j, init_rows[j]) + max_q + (int64_t)(6.0 * 0.7 * std::sqrt(sum_lambda)) + 64;
    rows = std::max(rows, rows_j);
  }
  if (rows * 64 > 0x7fffff00ll) return fail(PHM_ERR_UNSUPPORTED, "tree too large: dwell rows per replica tile exceed 32-bit indexing");
  e->rows = rows;

  if (e->wide && n_trees == 1) {
    // n > 4: a wave takes the replicas of its tile in turn for the n-vector work (phm_wide.hip), so a tile that holds
    // 64 replicas is 64 sequential passes in ONE wave.  With few replicas, place fewer of them on a tile (the unused lanes
    // are skipped) until about 8 192 waves exist or the padded layout would take more than a quarter of the free HBM.
    size_t free_now = 0, total_now = 0;
    HIPCHK(hipMemGetInfo(&free_now, &total_now));
    int rpt = 64;
    while (rpt > 1 && (e->S + rpt / 2 - 1) / (rpt / 2) <= 8192) rpt /= 2;
    auto tile_bytes = [&](int r) {
      const size_t tl = (size_t)(e->S + r - 1) / r;
      return tl * (sizeof(double) * ((size_t)rows * 64 + (size_t)s.n_node * n * 64) + 3 * (size_t)E * 64) +
             sizeof(double) * (size_t)max_iters * e->dcols * (e->reduce ? tl : tl * 64);
    };
    while (rpt < 64 && tile_bytes(rpt) > free_now / 4) rpt *= 2;
    e->rpt = rpt;
    e->tiles = (e->S + rpt - 1) / rpt; e->S_pad = e->tiles * 64; e->tpt = e->tiles;
  }

  // tips (0-based u8)
  if (n_trees > 1) {
    e->tips_host.assign((size_t)e->tiles * T * 64, 0);
    for (int tl = 0; tl < e->tiles; ++tl)
      for (int t = 0; t < T; ++t)
        std::memset(&e->tips_host[((size_t)tl * T + t) * 64], trees[tl / e->tpt].states[t] - 1, 64);
  } else if (e->tips_per_replica) {
    e->tips_host.assign((size_t)e->tiles * T * 64, (uint8_t)(x->states[0] - 1));      // padding lanes: any valid state
    for (int r = 0; r < e->S; ++r) {
      const int pr = e->pad_index(r);
      for (int t = 0; t < T; ++t) e->tips_host[((size_t)(pr / 64) * T + t) * 64 + (pr % 64)] = (uint8_t)(x->states[(size_t)r * T + t] - 1);
    }
    if (e->rpt == 64)      // dense tiles of the n <= 4 kernels: padding lanes are run like replicas, give them the last site
      for (int r = e->S; r < e->S_pad; ++r)
        for (int t = 0; t < T; ++t) e->tips_host[((size_t)(r / 64) * T + t) * 64 + (r % 64)] = (uint8_t)(x->states[(size_t)(e->S - 1) * T + t] - 1);
  } else {
    e->tips_host.resize(T);
    for (int t = 0; t < T; ++t) e->tips_host[t] = (uint8_t)(x->states[t] - 1);
  }

  // The n <= 4 replica kernel addresses a tile's arrays with 32-bit byte offsets from a scalar base (phm_mcmc.hip `at`).
  if (!e->wide && ((uint64_t)rows * 512u >= (1ull << 32) || (uint64_t)s.n_node * n * 512u >= (1ull << 32) ||
                   (uint64_t)s.n_edge * 128u >= (1ull << 32)))
    return fail(PHM_ERR_UNSUPPORTED, "tree too large for the replica layout (a tile's dwell stream or PL rows exceed 4 GB); use mapping=tiles");

  std::vector<double> col, row;
  // Full-length chain tables in global memory: the LDS copies of the replica kernels hold MCMC_KTAB rows, a longer chain
  // reads row k of these instead of being continued step by step (which made a draw on an m-segment branch cost O(m)).
  e->nw_klong = e->wide ? phm::wide_maxseg(n) + 1 : (int)std::min<int64_t>(65536, rows + 1);
  const int ktab = std::max(e->nw_klong, e->wide ? phm::WIDE_KTAB : phm::MCMC_KTAB);
  col.assign((size_t)ktab * n * n, 0.0); row.assign((size_t)ktab * n * n, 0.0);     // sizes only; filled by upload_model
  std::vector<double> maskpow((size_t)ktab * 2 * n, 0.0);

  const size_t stats_bytes = e->reduce ? sizeof(double) * (size_t)max_iters * e->tiles * e->dcols
                                                     : sizeof(double) * (size_t)max_iters * e->dcols * e->S_pad;
  const size_t dw_bytes = sizeof(double) * (size_t)e->tiles * rows * 64;
  // Two buffers save two VALU operations per dwell access (about 5 % of the n <= 4 sweep) and cost twice the HBM:
  // used when they take less than a third of the free memory, unless the caller asks (phm_options.storage: 1 ring, 2 two buffers).
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  e->ring = e->wide || n_trees > 1 || o.storage == 1 || (o.storage != 2 && 2 * dw_bytes > free_b / 3);
  size_t need = (e->ring ? 1 : 2) * dw_bytes + stats_bytes + sizeof(double) * (size_t)e->tiles * s.n_node * n * 64 +
                (size_t)e->tiles * (s.n_node + 2 * (size_t)E) * 64 + e->tips_host.size();
  if (need + (64u << 20) > free_b) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "engine needs %.2f GiB of HBM, %.2f GiB free (reduce n_replicas or max_iters)", need / 1073741824.0, free_b / 1073741824.0);
    return fail(PHM_ERR_OOM, buf);
  }
  HIPCHK(e->d_up.alloc(sizeof(phm::UpStep) * s.up.size() * n_trees));
  HIPCHK(e->d_down.alloc(sizeof(phm::DownStep) * s.down.size() * n_trees));
  HIPCHK(e->d_roots.alloc(sizeof(int32_t) * n_trees));
  HIPCHK(e->d_col.alloc(sizeof(double) * col.size()));
  HIPCHK(e->d_row.alloc(sizeof(double) * row.size()));
  HIPCHK(e->d_mask.alloc(sizeof(double) * maskpow.size()));

  HIPCHK(e->d_tips.alloc(e->tips_host.size()));
  HIPCHK(e->d_mcount.alloc(sizeof(uint16_t) * (size_t)e->tiles * E * 64));
  HIPCHK(e->d_dw0.alloc(dw_bytes));
  HIPCHK(e->d_cursor.alloc(sizeof(int32_t) * 2 * e->tiles));
  if (!e->ring) HIPCHK(e->d_dw1.alloc(dw_bytes));
  HIPCHK(e->d_PL.alloc(sizeof(double) * (size_t)e->tiles * s.n_node * n * 64));
  HIPCHK(e->d_nstate.alloc((size_t)e->tiles * s.n_node * 64));
  HIPCHK(e->d_stats.alloc(stats_bytes));
  HIPCHK(e->d_err.alloc(sizeof(uint32_t)));
  HIPCHK(e->d_seg.alloc(sizeof(unsigned long long)));
  if (e->reduce) HIPCHK(e->d_red.alloc(sizeof(double) * (size_t)max_iters * e->dcols));
  e->bytes = (int64_t)(e->d_up.bytes + e->d_down.bytes + e->d_col.bytes + e->d_row.bytes + e->d_tips.bytes + e->d_mcount.bytes +
                       e->d_dw0.bytes + e->d_dw1.bytes + e->d_cursor.bytes + e->d_PL.bytes + e->d_nstate.bytes + e->d_stats.bytes + e->d_red.bytes);

  for (int j = 0; j < n_trees; ++j) {
    const phm::Schedule& sj = e->scheds[j];
    HIPCHK(hipMemcpy(e->d_up.as<phm::UpStep>() + (size_t)j * sj.up.size(), sj.up.data(), sizeof(phm::UpStep) * sj.up.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_down.as<phm::DownStep>() + (size_t)j * sj.down.size(), sj.down.data(), sizeof(phm::DownStep) * sj.down.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_roots.as<int32_t>() + j, &sj.root, sizeof(int32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpy(e->d_tips.p, e->tips_host.data(), e->tips_host.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(e->d_err.p, 0, sizeof(uint32_t)));
  HIPCHK(hipMemset(e->d_seg.p, 0, sizeof(unsigned long long)));
  HIPCHK(hipMemset(e->d_stats.p, 0, stats_bytes));
  HIPCHK(hipMemset(e->d_nstate.p, 0, e->d_nstate.bytes));

  {   // initial paths -> every replica (makeabranch, src/phylomap.cpp:24-34, :901), tree by tree
    std::vector<int32_t> cur(2 * (size_t)e->tiles);
    for (int j = 0; j < n_trees; ++j) {
      const phm_tree* xt = &trees[j];
      DevBuf d_off, d_maps, d_irow;
      HIPCHK(d_irow.alloc(sizeof(int32_t) * E));
      HIPCHK(hipMemcpy(d_irow.p, init_row[j].data(), d_irow.bytes, hipMemcpyHostToDevice));
      HIPCHK(d_off.alloc(sizeof(int32_t) * (E + 1)));
      HIPCHK(d_maps.alloc(sizeof(double) * (size_t)xt->map_off[E]));
      HIPCHK(hipMemcpy(d_off.p, xt->map_off, d_off.bytes, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_maps.p, xt->maps, d_maps.bytes, hipMemcpyHostToDevice));
      const size_t tile0 = (size_t)j * e->tpt;
      HIPCHK(phm::launch_mcmc_init(E, e->tpt, rows, e->d_down.as<phm::DownStep>() + (size_t)j * E, d_irow.as<int32_t>(), d_off.as<int32_t>(),
                                   d_maps.as<double>(), e->d_dw0.as<double>() + tile0 * rows * 64, e->d_mcount.as<uint16_t>() + tile0 * E * 64, nullptr));
      HIPCHK(hipDeviceSynchronize());
      for (int t = 0; t < e->tpt; ++t) { cur[2 * (tile0 + t)] = 0; cur[2 * (tile0 + t) + 1] = e->ring ? (int32_t)(init_rows[j] % rows) : 0; }
    }
    HIPCHK(hipMemcpy(e->d_cursor.p, cur.data(), e->d_cursor.bytes, hipMemcpyHostToDevice));
  }

  if (e->wide) {
    HIPCHK(e->d_B2.alloc(sizeof(double) * n * n)); HIPCHK(e->d_Bc.alloc(sizeof(double) * n * n));
    HIPCHK(e->d_scale.alloc(sizeof(double) * n)); HIPCHK(e->d_pid.alloc(sizeof(double) * n));
    HIPCHK(hipMemcpy(e->d_B2.p, B2, e->d_B2.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_Bc.p, Bc, e->d_Bc.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_scale.p, scale, e->d_scale.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_pid.p, pid, e->d_pid.bytes, hipMemcpyHostToDevice));
    phm::WideParams& p = e->pw;
    p.n_states = n; p.n_tips = s.n_tips; p.n_node = s.n_node; p.n_edge = s.n_edge; p.root = s.root;
    p.n_tiles = e->tiles; p.n_rep = n_trees > 1 ? e->S_tree : e->S; p.n_rep_pad = e->S_pad; p.replica_offset = o.replica_offset;
    p.tiles_per_tree = n_trees > 1 ? e->tpt : 0; p.roots = e->d_roots.as<int32_t>(); p.rep_stride = e->rpt;
    p.normalise = e->normalise; p.tips_per_replica = e->tips_per_replica ? 1 : 0;
    p.reduce = e->reduce; p.n_cols = e->dcols; p.ktab = std::max(e->nw_klong, phm::WIDE_KTAB); p.sparse = (e->variant == PHM_MCMC_SPARSE); p.ks = ks_layout(e->variant); p.count_self = p.ks; p.tip_masks = hidden_rates(e->variant);
    p.maskpow = e->d_mask.as<double>();
    p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32);
    p.rows = e->rows;
    p.B2 = e->d_B2.as<double>(); p.Bc = e->d_Bc.as<double>(); p.scale = e->d_scale.as<double>(); p.pid = e->d_pid.as<double>();
    p.up = e->d_up.as<phm::UpStep>(); p.down = e->d_down.as<phm::DownStep>();
    p.colpow = e->d_col.as<double>(); p.rowpow = e->d_row.as<double>();
    p.tips = e->d_tips.as<uint8_t>(); p.mcount = e->d_mcount.as<uint16_t>();
    p.dwell0 = e->d_dw0.as<double>(); p.cursor = e->d_cursor.as<int32_t>();
    p.PL = e->d_PL.as<double>(); p.nstate = e->d_nstate.as<uint8_t>(); p.stats = e->d_stats.as<double>();
    p.err = e->d_err.as<uint32_t>(); p.segcnt = e->d_seg.as<unsigned long long>();
  }
  if (n == 2) fill_params<2>(e, e->p2, B2, Bc, scale, pid, o);
  if (n == 3) fill_params<3>(e, e->p3, B2, Bc, scale, pid, o);
  if (n == 4) fill_params<4>(e, e->p4, B2, Bc, scale, pid, o);
  st = upload_model(e);
  if (st) return st;
  HIPCHK(hipEventCreate(&e->ev0));
  HIPCHK(hipEventCreate(&e->ev1));
  *out = guard.release();
  return PHM_OK;
}

extern "C" {

int32_t phm_engine_run(phm_engine* e, int32_t n_iters, void* hip_stream) {
  e = live(e);
  if (!e) return fail(PHM_ERR_STATE, "engine is NULL");
  if (e->dead) return dead_engine();
  if (n_iters < 0 || e->iters_done + n_iters > e->max_iters) return fail(PHM_ERR_STATE, "iteration range exceeds max_iters");
  HIPCHK(hipSetDevice(e->device));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIPCHK(hipEventRecord(e->ev0, stream));
  int launches = 0;
  if (e->narrow) {
    hipError_t le = hipSuccess;
    for (int i = 0; i < n_iters && le == hipSuccess; ++i) {
      const int it = e->iters_done + i;
      if (e->n == 2) le = phm::launch_narrow_sweep<2>(e->n2, e->nw_tier_off, e->nw_walk_off, it, stream, i > 0);
      if (e->n == 3) le = phm::launch_narrow_sweep<3>(e->n3, e->nw_tier_off, e->nw_walk_off, it, stream, i > 0);
      if (e->n == 4) le = phm::launch_narrow_sweep<4>(e->n4, e->nw_tier_off, e->nw_walk_off, it, stream, i > 0);
      if (e->wide) le = phm::launch_wbranch_sweep(e->pwb, e->nw_up_off, e->nw_down_off, it, stream);
      launches += (int)(e->nw_up_off.size() + e->nw_down_off.size()) + 2;
    }
    if (n_iters > 0 && le == hipSuccess && !e->wide) {       // the last sweep's statistics (phm_narrow.hip defers them by one launch)
      const int it = e->iters_done + n_iters - 1;
      if (e->n == 2) le = phm::launch_narrow_stats<2>(e->n2, it, stream);
      if (e->n == 3) le = phm::launch_narrow_stats<3>(e->n3, it, stream);
      if (e->n == 4) le = phm::launch_narrow_stats<4>(e->n4, it, stream);
    }
    HIPCHK(le);
  }
  if (e->tiled) {
    hipError_t le = hipSuccess;
    e->phase_iters = 0;
    if (e->phase_timing) {
      while ((int)e->phase_ev.size() < 5 * n_iters) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); e->phase_ev.push_back(ev); }
      e->phase_iters = n_iters;
    }
    for (int i = 0; i < n_iters && le == hipSuccess; ++i) {
      const int it = e->iters_done + i;
      hipEvent_t* pev = e->phase_timing ? &e->phase_ev[5 * (size_t)i] : nullptr;
      if (e->n == 2) le = phm::launch_tiles_sweep<2>(e->t2, e->nw_up_off, e->nw_down_off, e->nw_tier_off, it, stream, pev);
      if (e->n == 3) le = phm::launch_tiles_sweep<3>(e->t3, e->nw_up_off, e->nw_down_off, e->nw_tier_off, it, stream, pev);
      if (e->n == 4) le = phm::launch_tiles_sweep<4>(e->t4, e->nw_up_off, e->nw_down_off, e->nw_tier_off, it, stream, pev);
      if (e->wide) le = phm::launch_wtiles_sweep(e->pwt, e->wt_band, e->wt_sparse, e->nw_up_off, e->nw_down_off, e->nw_tier_off, it, stream, pev);
      const bool clusters = !e->nw_tier_off.empty();
      const int tiers = (int)e->nw_tier_off.size() - 1;
      if (clusters && e->wide)             // 5 .. 32 states on a deep tree: the node draws by tier, the pruning pass too when it runs on the band kernel
        launches += (e->pwt.band_up > 0 && !e->wt_sparse.kernel ? tiers : (int)e->nw_up_off.size() - 1) + tiers + 4;
      else
        launches += clusters ? 2 * tiers + 3      // a launch per tier and pass, branch kernel, two reductions
                             : (int)(e->nw_up_off.size() + e->nw_down_off.size()) + 2;
    }
    HIPCHK(le);
  }
  for (int done = 0; done < n_iters && !e->narrow && !e->tiled;) {
    int chunk = std::min(e->ipl, n_iters - done);
    hipError_t le = hipSuccess;
    if (e->n == 2) le = phm::launch_mcmc<2>(e->p2, e->iters_done + done, chunk, stream);
    if (e->n == 3) le = phm::launch_mcmc<3>(e->p3, e->iters_done + done, chunk, stream);
    if (e->n == 4) le = phm::launch_mcmc<4>(e->p4, e->iters_done + done, chunk, stream);
    if (e->wide && !e->narrow) le = phm::launch_mcmc_wide(e->pw, e->iters_done + done, chunk, stream);
    HIPCHK(le);
    done += chunk;
    ++launches;
  }
  HIPCHK(hipEventRecord(e->ev1, stream));
  for (int i = 0; i < n_iters; ++i) e->qhist.push_back(e->qparams);      // recordQ / recordQks at the start of each sweep
  e->iters_done += n_iters;
  e->epi_iter = (e->narrow && !e->wide && n_iters > 0 && ((e->n == 2 && e->n2.host_row) || (e->n == 3 && e->n3.host_row) || (e->n == 4 && e->n4.host_row)))
                    ? e->iters_done - 1 : -1;
  e->last_stream = stream;
  e->timing_pending = true;
  e->last_launches = launches;
  return PHM_OK;
}

// A sweep outgrew its slots / streams: free this engine's device memory, build a replacement with doubled capacities from the
// saved inputs and replay the iterations run so far (bit-identical: streams are addressed, not consumed), model changes of the
// Q-updating drivers at their iterations included; every later call on the handle continues on the replacement.
static int32_t recover_capacity(phm_engine* e) {
  const int T = e->iters_done;
  std::shared_ptr<SavedInput> sv = e->saved;
  hipStream_t stream = e->last_stream;
  int boost_log2 = 0;
  while ((1 << boost_log2) < e->cap_boost) ++boost_log2;
  if (boost_log2 >= 8) return fail(PHM_ERR_CAPACITY, "a branch outgrew its dwell capacity after 8 doublings of the slots");
  // The chain state of this engine is past an overflow and of no further use; its buffers go first because the doubled slots
  // of the replacement rarely fit beside them.  From here on the handle is either forwarded to a working replacement or DEAD:
  // every entry point refuses a dead engine (PHM_ERR_CAPACITY) instead of launching on freed memory.
  e->release_device();
  e->dead = true;
  phm_options o = sv->opt;
  phm_engine* r = nullptr;
  const std::vector<std::pair<int32_t, std::vector<double>>> hist = sv->model_hist;      // the replay appends its own copy
  int32_t st = phm_engine_create_impl(sv->flat.data(), (int32_t)sv->flat.size(), &sv->model, &o, sv->dbg, boost_log2 + 1, sv->max_iters, &r);
  if (!st && sv->dbg.fail_recovery) {      // test aid: a replacement that "does not fit" (tests/test_gpu_parity.py)
    delete r; r = nullptr;
    st = fail(PHM_ERR_OOM, "phm_debug_options.fail_recovery is set");
  }
  if (st) return (st == PHM_ERR_OOM) ? fail(PHM_ERR_CAPACITY, "a branch outgrew its dwell capacity and larger slots do not fit in HBM: " + g_phm_err) : st;
  r->recoveries = e->recoveries + 1;
  size_t h = 0;
  for (int it = 0; it < T && !st;) {
    while (!st && h < hist.size() && hist[h].first <= it) { st = phm_engine_set_model(r, hist[h].second.data()); ++h; }
    if (st) break;
    const int next = (h < hist.size()) ? std::min(T, hist[h].first) : T;
    st = phm_engine_run(r, next - it, stream);
    it = next;
  }
  while (!st && h < hist.size()) { st = phm_engine_set_model(r, hist[h].second.data()); ++h; }
  if (!st) st = phm_engine_sync(r);   // may recover again (the replacement forwards in turn; it ends up dead itself if that fails)
  if (st) { const std::string msg = g_phm_err; delete r; return fail(st, "capacity recovery failed, the engine is no longer usable: " + msg); }
  e->fwd = r;
  e->dead = false;
  return PHM_OK;
}

int32_t phm_engine_sync(phm_engine* e) {
  e = live(e);
  if (!e) return fail(PHM_ERR_STATE, "engine is NULL");
  if (e->dead) return dead_engine();
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(wait_stream(e->last_stream));
  if (e->timing_pending) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->last_ms = ms;
    e->timing_pending = false;
    for (double& v : e->phase_ms) v = 0.0;
    for (int i = 0; i < e->phase_iters; ++i)
      for (int ph = 0; ph < 4; ++ph) {
        float pm = 0.f;
        HIPCHK(hipEventElapsedTime(&pm, e->phase_ev[5 * (size_t)i + ph], e->phase_ev[5 * (size_t)i + ph + 1]));
        e->phase_ms[ph] += pm;
      }
  }
  if (e->epi_iter == e->iters_done - 1 && e->epi_iter >= 0) {      // one chain: the statistics kernel left both words in host memory
    const double* hr = e->pin_row.as<double>();
    const uint32_t derr = (uint32_t)hr[e->dcols + 1];
    e->seg_total = (unsigned long long)hr[e->dcols];
    if ((derr & phm::DERR_CAPACITY) && !(derr & ~phm::DERR_CAPACITY) && e->recover && e->saved) return recover_capacity(e);
    return device_status(derr);
  }
  // error word and segment counter: two asynchronous copies into page-locked memory and ONE (polling) wait -- two pageable
  // hipMemcpy calls were ~40 us of every iteration of the rate-updating drivers
  HIPCHK(e->pin_status.reserve(16));
  HIPCHK(hipMemcpyAsync(e->pin_status.p, e->d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->last_stream));
  HIPCHK(hipMemcpyAsync(e->pin_status.as<unsigned char>() + 8, e->d_seg.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->last_stream));
  HIPCHK(wait_stream(e->last_stream));
  uint32_t derr = 0;
  std::memcpy(&derr, e->pin_status.p, sizeof derr);
  std::memcpy(&e->seg_total, e->pin_status.as<unsigned char>() + 8, sizeof(unsigned long long));
  const bool wide_replicas = e->wide && !e->narrow && !e->tiled;      // phm_wide.hip: a fixed 128-segment state scratch per replica in LDS, not recoverable
  if ((derr & phm::DERR_CAPACITY) && !(derr & ~phm::DERR_CAPACITY) && e->recover && e->saved && !wide_replicas) return recover_capacity(e);
  if ((derr & phm::DERR_CAPACITY) && wide_replicas)
    return fail(PHM_ERR_CAPACITY, "a branch outgrew the replica mapping for 5..64 states (at most " + std::to_string(phm::wide_maxseg(e->n)) +
                                  " segments per branch and replica, or the dwell ring); use PHM_MAP_TILES or PHM_MAP_BRANCHES (the automatic choice for one tree)");
  return device_status(derr);
}

int32_t phm_engine_read_stats(phm_engine* e, int32_t iter0, int32_t n, double* out) {
  e = live(e);
  if (!e || !out) return fail(PHM_ERR_STATE, "engine/out is NULL");
  if (e->dead) return dead_engine();
  if (iter0 < 0 || n < 0 || iter0 + n > e->iters_done) return fail(PHM_ERR_STATE, "statistics requested for iterations that have not run");
  if (n == 0) return PHM_OK;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(wait_stream(e->last_stream));
  const int cols = e->cols, dcols = e->dcols;
  // device column -> result column: identical except for ks, whose parameter columns (recordQks) sit between the
  // counters and the root state and are constants of the fixed Q
  auto out_col = [&](int dc) { return (dcols != cols && dc == dcols - 1) ? cols - 1 : dc; };
  auto fill_params = [&](double* mat) {      // mat: n x cols column-major
    if (dcols == cols) return;
    for (int i = 0; i < n; ++i) {
      const std::vector<double>& qp = e->qhist[iter0 + i];
      for (size_t q = 0; q < qp.size(); ++q) mat[(size_t)(dcols - 1 + q) * n + i] = qp[q];
    }
  };
  if (e->reduce) {
    HIPCHK(phm::launch_stats_reduce(e->d_stats.as<double>() + (size_t)iter0 * e->tiles * dcols, n, e->tiles, dcols,
                                      e->d_red.as<double>(), e->last_stream));
    std::vector<double> h((size_t)n * dcols);
    HIPCHK(hipMemcpyAsync(h.data(), e->d_red.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, e->last_stream));
    HIPCHK(hipStreamSynchronize(e->last_stream));
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < dcols; ++c) out[(size_t)out_col(c) * n + i] = h[(size_t)i * dcols + c];
    fill_params(out);
  } else {
    const size_t hn = (size_t)n * dcols * e->S_pad;
    std::vector<double> hv;
    const double* h;
    if (n == 1 && iter0 == e->epi_iter && e->S == 1) {   // the row the statistics kernel wrote to host memory
      hv.assign(hn, 0.0);
      for (int c = 0; c < dcols; ++c) hv[(size_t)c * e->S_pad + e->pad_index(0)] = e->pin_row.as<double>()[c];
      h = hv.data();
    } else if (sizeof(double) * hn <= (1u << 20)) {    // a row or a few: page-locked staging, one asynchronous copy
      HIPCHK(e->pin_down.reserve(sizeof(double) * hn));
      HIPCHK(hipMemcpyAsync(e->pin_down.p, e->d_stats.as<double>() + (size_t)iter0 * dcols * e->S_pad, sizeof(double) * hn, hipMemcpyDeviceToHost, e->last_stream));
      HIPCHK(wait_stream(e->last_stream));
      h = e->pin_down.as<double>();
    } else if (2 * (int64_t)e->S <= e->S_pad) {        // few replicas on padded tiles (one chain at 61 states, 100 sweeps: 194 MB of padding for 3 MB
      // of statistics, 53 of the call's 96 ms): their columns are packed on the device first
      std::vector<int32_t> pick(e->S);
      for (int r = 0; r < e->S; ++r) pick[r] = e->pad_index(r);
      DevBuf d_pick, d_pack;
      const size_t n_rows = (size_t)n * dcols;
      HIPCHK(d_pick.alloc(sizeof(int32_t) * pick.size())); HIPCHK(d_pack.alloc(sizeof(double) * n_rows * e->S));
      HIPCHK(hipMemcpyAsync(d_pick.p, pick.data(), d_pick.bytes, hipMemcpyHostToDevice, e->last_stream));
      HIPCHK(phm::launch_stats_gather(e->d_stats.as<double>() + (size_t)iter0 * dcols * e->S_pad, (int64_t)n_rows, e->S_pad, e->S, d_pick.as<int32_t>(),
                                      d_pack.as<double>(), e->last_stream));
      hv.resize(n_rows * e->S);
      HIPCHK(hipMemcpyAsync(hv.data(), d_pack.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, e->last_stream));
      HIPCHK(hipStreamSynchronize(e->last_stream));
      for (int r = 0; r < e->S; ++r) {
        for (int c = 0; c < dcols; ++c)
          for (int i = 0; i < n; ++i) out[((size_t)r * cols + out_col(c)) * n + i] = hv[((size_t)i * dcols + c) * e->S + r];
        fill_params(out + (size_t)r * cols * n);
      }
      return PHM_OK;
    } else {
      hv.resize(hn);
      HIPCHK(hipMemcpy(hv.data(), e->d_stats.as<double>() + (size_t)iter0 * dcols * e->S_pad, sizeof(double) * hn, hipMemcpyDeviceToHost));
      h = hv.data();
    }
    for (int r = 0; r < e->S; ++r) {
      for (int c = 0; c < dcols; ++c)
        for (int i = 0; i < n; ++i) out[((size_t)r * cols + out_col(c)) * n + i] = h[((size_t)i * dcols + c) * e->S_pad + e->pad_index(r)];
      fill_params(out + (size_t)r * cols * n);
    }
  }
  return PHM_OK;
}

}  // extern "C"

// Multi-device one-shot calls, reduce = 1: the statistics of iterations [iter0, iter0 + n) summed over this engine's tiles in tile order,
// CONTINUING from `acc` (n x dcols row-major: the totals of the devices before this one; empty = start from zero) -- the fold of
// stats_reduce_kernel carried across devices, so whole-tile shards reproduce the one-device sum term by term.  `acc` is replaced.
int32_t phm_engine_fold_reduced(phm_engine* e, int32_t iter0, int32_t n, std::vector<double>& acc) {
  e = live(e);
  if (!e) return fail(PHM_ERR_STATE, "engine is NULL");
  if (e->dead) return dead_engine();
  if (!e->reduce) return fail(PHM_ERR_STATE, "engine was not created with reduce = 1");
  if (iter0 < 0 || n < 1 || iter0 + n > e->iters_done) return fail(PHM_ERR_STATE, "statistics requested for iterations that have not run");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(wait_stream(e->last_stream));
  const size_t cnt = (size_t)n * e->dcols;
  if (!acc.empty() && acc.size() != cnt) return fail(PHM_ERR_STATE, "fold: accumulator size mismatch");
  const double* init = nullptr;
  DevBuf dinit;
  if (!acc.empty()) {
    HIPCHK(dinit.alloc(sizeof(double) * cnt));
    HIPCHK(hipMemcpyAsync(dinit.p, acc.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, e->last_stream));
    init = dinit.as<double>();
  }
  HIPCHK(phm::launch_stats_reduce(e->d_stats.as<double>() + (size_t)iter0 * e->tiles * e->dcols, n, e->tiles, e->dcols,
                                    e->d_red.as<double>() + (size_t)iter0 * e->dcols, e->last_stream, init));
  acc.resize(cnt);
  HIPCHK(hipMemcpyAsync(acc.data(), e->d_red.as<double>() + (size_t)iter0 * e->dcols, sizeof(double) * cnt, hipMemcpyDeviceToHost, e->last_stream));
  HIPCHK(hipStreamSynchronize(e->last_stream));
  return PHM_OK;
}

// ... and the result matrix from the finished fold: device columns -> result columns (column-major n x cols), parameter columns of
// the fixed Q filled in as phm_engine_read_stats does
int32_t phm_engine_finish_reduced(phm_engine* e, int32_t iter0, int32_t n, const std::vector<double>& acc, double* out) {
  e = live(e);
  if (!e || !out) return fail(PHM_ERR_STATE, "engine/out is NULL");
  const int cols = e->cols, dcols = e->dcols;
  if (acc.size() != (size_t)n * dcols) return fail(PHM_ERR_STATE, "fold: accumulator size mismatch");
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < dcols; ++c) out[(size_t)((dcols != cols && c == dcols - 1) ? cols - 1 : c) * n + i] = acc[(size_t)i * dcols + c];
  if (dcols != cols)
    for (int i = 0; i < n; ++i) {
      const std::vector<double>& qp = e->qhist[iter0 + i];
      for (size_t q = 0; q < qp.size(); ++q) out[(size_t)(dcols - 1 + q) * n + i] = qp[q];
    }
  return PHM_OK;
}

extern "C" {

int32_t phm_engine_dump(phm_engine* e, int32_t replica, int32_t* seg_count, double* seg_dwell, int32_t seg_cap,
                        int32_t* node_states, double* PL) {
  e = live(e);
  if (!e) return fail(PHM_ERR_STATE, "engine is NULL");
  if (e->dead) return dead_engine();
  if (replica < 0 || replica >= e->S) return fail(PHM_ERR_BAD_INPUT, "replica out of range");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->last_stream));
  if (e->narrow) {      // branch-parallel layout: CSR slots of the buffer the next sweep will read
    const phm::Schedule& s = e->sched;
    const int E = s.n_edge, T = s.n_tips, n = e->n;
    std::vector<int32_t> mc(E);
    HIPCHK(hipMemcpy(mc.data(), e->d_nw_mcount.as<int32_t>() + (size_t)replica * E, sizeof(int32_t) * E, hipMemcpyDeviceToHost));
    if (seg_count) for (int b = 0; b < E; ++b) seg_count[b] = mc[b];
    if (seg_dwell) {
      std::vector<double> dw((size_t)e->nw_total_cap);
      const double* src = ((e->iters_done & 1) ? e->d_nw_dwB.as<double>() : e->d_nw_dwA.as<double>()) + (size_t)replica * e->nw_total_cap;
      HIPCHK(hipMemcpy(dw.data(), src, sizeof(double) * dw.size(), hipMemcpyDeviceToHost));
      for (int b = 0; b < E; ++b)
        for (int i = 0; i < std::min<int>(mc[b], seg_cap); ++i) seg_dwell[(size_t)b * seg_cap + i] = dw[(size_t)e->nw_off[b] + i];
    }
    auto tip_state = [&](int t) -> int { return e->tips_per_replica ? e->tips_host[(size_t)replica * T + t] : e->tips_host[t]; };
    if (node_states) {
      std::vector<uint8_t> ns(s.n_node);
      HIPCHK(hipMemcpy(ns.data(), e->d_nstate.as<uint8_t>() + (size_t)replica * s.n_node, ns.size(), hipMemcpyDeviceToHost));
      for (int t = 0; t < T; ++t) node_states[t] = tip_state(t) + 1;
      for (int v = 0; v < s.n_node; ++v) node_states[T + v] = ns[v] + 1;
    }
    if (PL) {
      std::vector<double> pl((size_t)s.n_node * n);
      HIPCHK(hipMemcpy(pl.data(), e->d_PL.as<double>() + (size_t)replica * s.n_node * n, sizeof(double) * pl.size(), hipMemcpyDeviceToHost));
      for (int t = 0; t < T; ++t)
        for (int c = 0; c < n; ++c) PL[(size_t)t * n + c] = (c == tip_state(t)) ? 1.0 : 0.0;
      for (int v = 0; v < s.n_node; ++v)
        for (int c = 0; c < n; ++c) PL[(size_t)(T + v) * n + c] = pl[(size_t)v * n + c];
    }
    return PHM_OK;
  }
  const int padded = e->pad_index(replica), tile = padded / 64, lane = padded % 64;
  const phm::Schedule& s = e->scheds[e->n_trees > 1 ? tile / e->tpt : 0];
  const int E = s.n_edge, T = s.n_tips, n = e->n;
  std::vector<uint16_t> mc((size_t)E * 64);
  HIPCHK(hipMemcpy(mc.data(), e->d_mcount.as<uint16_t>() + (size_t)tile * E * 64, sizeof(uint16_t) * mc.size(), hipMemcpyDeviceToHost));
  if (seg_count) for (int b = 0; b < E; ++b) seg_count[b] = mc[(size_t)b * 64 + lane];
  if (seg_dwell && e->tiled) {      // slots of the buffer the next sweep will read
    std::vector<double> dw((size_t)e->rows * 64);
    const double* src = ((e->iters_done & 1) ? e->d_dw1.as<double>() : e->d_dw0.as<double>()) + (size_t)tile * e->rows * 64;
    HIPCHK(hipMemcpy(dw.data(), src, sizeof(double) * dw.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < E; ++b)
      for (int i = 0; i < std::min<int>(mc[(size_t)b * 64 + lane], seg_cap); ++i)
        seg_dwell[(size_t)b * seg_cap + i] = dw[((size_t)e->tl_slot[b] + i) * 64 + lane];
  } else if (seg_dwell) {
    std::vector<double> dw((size_t)e->rows * 64);
    int32_t cur[2];
    HIPCHK(hipMemcpy(cur, e->d_cursor.as<int32_t>() + 2 * tile, sizeof cur, hipMemcpyDeviceToHost));
    const double* src = ((!e->ring && cur[0]) ? e->d_dw1.as<double>() : e->d_dw0.as<double>()) + (size_t)tile * e->rows * 64;
    HIPCHK(hipMemcpy(dw.data(), src, sizeof(double) * dw.size(), hipMemcpyDeviceToHost));
    size_t row = e->ring ? (size_t)cur[0] : 0;     // replay the stream layout from the ring cursor: branch down[k] occupies max-over-lanes(m) rows
    for (int k = 0; k < E; ++k) {
      const phm::DownStep& d = s.down[k];
      int m = std::min<int>(mc[(size_t)d.edge * 64 + lane], seg_cap);
      for (int i = 0; i < m; ++i) seg_dwell[(size_t)d.edge * seg_cap + i] = dw[((row + i) % (size_t)e->rows) * 64 + lane];
      int mx = 0;
      for (int l = 0; l < 64; ++l) mx = std::max<int>(mx, mc[(size_t)d.edge * 64 + l]);
      row += mx;
    }
  }
  auto tip_state = [&](int t) -> int {
    return e->tips_per_replica ? e->tips_host[((size_t)tile * T + t) * 64 + lane] : e->tips_host[t];
  };
  if (node_states) {
    std::vector<uint8_t> ns((size_t)s.n_node * 64);
    HIPCHK(hipMemcpy(ns.data(), e->d_nstate.as<uint8_t>() + (size_t)tile * s.n_node * 64, ns.size(), hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t) node_states[t] = tip_state(t) + 1;
    for (int v = 0; v < s.n_node; ++v) node_states[T + v] = ns[(size_t)v * 64 + lane] + 1;
  }
  if (PL) {
    std::vector<double> pl((size_t)s.n_node * n * 64);
    HIPCHK(hipMemcpy(pl.data(), e->d_PL.as<double>() + (size_t)tile * s.n_node * n * 64, sizeof(double) * pl.size(), hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < n; ++c) PL[(size_t)t * n + c] = (c == tip_state(t)) ? 1.0 : 0.0;
    for (int v = 0; v < s.n_node; ++v)
      for (int c = 0; c < n; ++c)
        PL[(size_t)(T + v) * n + c] = (e->wide && !e->tiled) ? pl[((size_t)v * 64 + lane) * n + c] : pl[((size_t)v * n + c) * 64 + lane];
  }
  return PHM_OK;
}

int32_t phm_engine_info(phm_engine* e, phm_info* info) {
  e = live(e);
  if (!e || !info) return fail(PHM_ERR_STATE, "engine/info is NULL");
  if (e->dead) return dead_engine();
  std::memset(info, 0, sizeof(*info));
  info->n_states = e->n; info->n_edge = e->sched.n_edge; info->n_replicas = e->S; info->n_replicas_padded = e->S_pad;
  info->n_cols = e->cols; info->max_iters = e->max_iters; info->device_bytes = e->bytes;
  info->rows_per_replica = e->rows;
  info->seg_read = (int64_t)e->seg_total; info->seg_written = 0;
  info->last_run_ms = e->last_ms; info->last_run_launches = e->last_launches; info->iters_done = e->iters_done;
  info->recoveries = e->recoveries;
  info->mapping = e->narrow ? PHM_MAP_BRANCHES : e->tiled ? PHM_MAP_TILES : PHM_MAP_REPLICAS;
  info->sparse_chains = (e->tiled && e->wide) ? (e->pwt.band_up > 0 || e->wt_sparse.kernel != nullptr) + 2 * (e->pwt.band_draw > 0) + 4 * (e->wt_sparse.kernel != nullptr) : 0;
  return PHM_OK;
}

void phm_engine_destroy(phm_engine* e) { delete e; }      // deletes the chain of rebuilt engines too

int32_t phm_engine_phase_ms(phm_engine* e, double* out4) {
  e = live(e);
  if (!e || !out4) return fail(PHM_ERR_STATE, "engine/out is NULL");
  if (e->dead) return dead_engine();
  if (!e->phase_timing || !e->tiled) return fail(PHM_ERR_STATE, "phase timing needs phm_options.phase_timing = 1 and a (tile, item) mapping");
  for (int i = 0; i < 4; ++i) out4[i] = e->phase_ms[i];
  return PHM_OK;
}

}  // extern "C"


// Device-resident reduced statistics (reduce mode): runs the fixed-order tile reduction on `hip_stream` and
// returns a DEVICE pointer to n x cols doubles, row-major [iteration][column].  For multi-GPU callers that
// hand the buffer to RCCL without a host round trip.  Valid until the next call on this engine.
extern "C" int32_t phm_engine_reduced_stats_device(phm_engine* e, int32_t iter0, int32_t n, void* hip_stream,
                                                   void** out_dev) {
  e = live(e);
  if (!e || !out_dev) return fail(PHM_ERR_STATE, "engine/out is NULL");
  if (e->dead) return dead_engine();
  if (!e->reduce) return fail(PHM_ERR_STATE, "engine was not created with reduce = 1");
  if (iter0 < 0 || n < 1 || iter0 + n > e->iters_done) return fail(PHM_ERR_STATE, "statistics requested for iterations that have not run");
  HIPCHK(hipSetDevice(e->device));
  // its own buffer: phm_engine_read_stats reduces into d_red, which must not overwrite a matrix the caller has already
  // handed to (and had all-reduced in place by) RCCL
  if (!e->d_red_out.p) HIPCHK(e->d_red_out.alloc(sizeof(double) * (size_t)e->max_iters * e->dcols));
  HIPCHK(phm::launch_stats_reduce(e->d_stats.as<double>() + (size_t)iter0 * e->tiles * e->dcols, n, e->tiles, e->dcols,
                                    e->d_red_out.as<double>(), reinterpret_cast<hipStream_t>(hip_stream)));
  *out_dev = e->d_red_out.p;
  return PHM_OK;
}

// Measurement aid: time `n_iters` repetitions of the pruning (up) sweep alone -- makePLrcpp*, src/phylomap.cpp:503-529 --
// on the engine's current chain state (segment counts are left untouched, so every repetition does the same work).
// Returns the HIP-event time in milliseconds.  n <= 4 kernels only.
extern "C" int32_t phm_engine_time_pruning(phm_engine* e, int32_t n_iters, void* hip_stream, double* ms_out) {
  e = live(e);
  if (!e || !ms_out) return fail(PHM_ERR_STATE, "engine/ms_out is NULL");
  if (e->dead) return dead_engine();
  const bool wt = e->wide && e->tiled;
  if (!wt && (e->wide || e->narrow || e->tiled || e->n_trees > 1)) return fail(PHM_ERR_UNSUPPORTED, "pruning-only timing is implemented for the replica mapping with n_states <= 4 and the lane-per-replica mapping of 5..64 states");
  if (n_iters < 1) return fail(PHM_ERR_BAD_INPUT, "n_iters must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIPCHK(hipEventRecord(e->ev0, stream));
  hipError_t le = hipSuccess;
  if (wt) for (int i = 0; i < n_iters && le == hipSuccess; ++i) le = phm::launch_wtiles_up(e->pwt, e->wt_band, e->wt_sparse, e->nw_up_off, e->nw_tier_off, stream);
  // iteration index = iters_done keeps the dwell ping-pong parity; nothing but PL is written
  if (!wt && e->n == 2) { auto p = e->p2; p.prune_only = 1; for (int i = 0; i < n_iters && le == hipSuccess; ++i) le = phm::launch_mcmc<2>(p, e->iters_done, 1, stream); }
  if (e->n == 3) { auto p = e->p3; p.prune_only = 1; for (int i = 0; i < n_iters && le == hipSuccess; ++i) le = phm::launch_mcmc<3>(p, e->iters_done, 1, stream); }
  if (e->n == 4) { auto p = e->p4; p.prune_only = 1; for (int i = 0; i < n_iters && le == hipSuccess; ++i) le = phm::launch_mcmc<4>(p, e->iters_done, 1, stream); }
  HIPCHK(le);
  HIPCHK(hipEventRecord(e->ev1, stream));
  HIPCHK(hipStreamSynchronize(stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *ms_out = ms;
  e->last_stream = stream;
  e->timing_pending = false;
  return PHM_OK;
}

// Replace the rate matrix between sweeps (the Q-updating variants edit Q and B after every iteration,
// src/phylomap.cpp:1212-1217, :1862-1866).  Q column-major; B = I + Q/Omega is recomputed.  The chain state is kept.
extern "C" int32_t phm_engine_set_model(phm_engine* e, const double* Q) {
  e = live(e);
  if (!e || !Q) return fail(PHM_ERR_STATE, "engine/Q is NULL");
  if (e->dead) return dead_engine();
  if (e->saved) {
    // the replay of a capacity recovery needs every model the chain has run under; kept while that history stays small
    // (256 MB of host memory: 10^4 iterations at 58 states), dropped -- and the recovery with it -- beyond
    auto& hist = e->saved->model_hist;
    if (!hist.empty() && hist.back().first == e->iters_done) hist.pop_back();      // two updates before the same sweep: the later one counts
    const size_t entry = sizeof(double) * (size_t)e->n * e->n;
    if ((hist.size() + 1) * entry > (256u << 20)) { e->saved.reset(); e->recover = false; }
    else hist.emplace_back(e->iters_done, std::vector<double>(Q, Q + (size_t)e->n * e->n));
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(wait_stream(e->last_stream));
  std::vector<double> B2, Bc, scale, qp;
  int32_t st = compute_model(e->variant, e->n, Q, nullptr, e->Omega, B2, Bc, scale, qp);
  if (st) return st;
  e->hB2 = B2; e->hBc = Bc; e->hscale = scale; e->qparams = qp;
  return upload_model(e);
}
