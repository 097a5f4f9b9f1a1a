// phm_internal.h -- shared by the translation units behind the C-ABI (phm_engine.cpp, phm_drivers.cpp, phm_expm_api.cpp):
// error reporting, device buffers, the engine object.  Not installed; include/phylomap_hip.h is the public interface.
#pragma once

#include "../../include/phylomap_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "phm_exp.h"
#include "phm_mcmc.h"
#include "phm_narrow.h"
#include "phm_qupdate.h"
#include "phm_sched.h"
#include "phm_tiles.h"
#include "phm_wbranch.h"
#include "phm_wide.h"
#include "phm_wtiles.h"

inline thread_local std::string g_phm_err;      // phm_last_error()
inline thread_local double g_phm_last_kernel_ms = 0.0;      // phm_last_kernel_ms()
inline thread_local phm_debug_options g_phm_debug = {};     // phm_set_debug_options()

inline int32_t fail(int32_t st, const std::string& msg) { g_phm_err = msg; return st; }

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      int32_t _st = (_e == hipErrorOutOfMemory) ? PHM_ERR_OOM : PHM_ERR_NO_DEVICE;                \
      return fail(_st, std::string(#call) + ": " + hipGetErrorString(_e));                        \
    }                                                                                             \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void reset() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  // re-allocating frees the previous block first (the automatic mapping fallback of phm_engine_create_multi sets the same
  // members up again after an out-of-memory attempt)
  hipError_t alloc(size_t n) {
    reset();
    hipError_t e = hipMalloc(&p, n ? n : 16);
    if (e == hipSuccess) bytes = n; else p = nullptr;
    return e;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// page-locked host staging (async copies of a few KB between sweeps: a pageable hipMemcpy costs a synchronous ~15-30 us each)
struct PinnedBuf {
  void* p = nullptr;
  size_t bytes = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; bytes = 0;
    hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
    if (e == hipSuccess) bytes = n; else p = nullptr;
    return e;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Wait for a stream the way a loop of short sweeps needs it: hipStreamSynchronize sleeps on an interrupt and wakes 20-50 us after
// the work is done (more than half a one-chain sweep); polling the stream for the first half millisecond returns within
// microseconds, and anything longer falls through to the sleeping wait.
inline hipError_t wait_stream(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) return q;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) return hipStreamSynchronize(s);
  }
}

inline int32_t select_device(int32_t device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PHM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device >= 0) {
    if (device >= n) return fail(PHM_ERR_NO_DEVICE, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
  }
  return PHM_OK;
}

inline int32_t device_status(uint32_t derr) {
  if (derr & phm::DERR_ZERO_PROB) return fail(PHM_ERR_ZERO_PROB, "all-zero or non-finite probability vector while sampling a state (RcppArmadillo::sample would throw)");
  if (derr & phm::DERR_CAPACITY) return fail(PHM_ERR_CAPACITY, "a branch outgrew its dwell capacity; lower phm_options.cap_tail");
  if (derr & phm::DERR_UNIF_CAP) return fail(PHM_ERR_UNIF_CAP, "newunifSample needed more than 300 jumps on a branch (src/phylomap.cpp:120)");
  if (derr & phm::DERR_SAMPLEONCE) return fail(PHM_ERR_ZERO_PROB, "sampleOnce ran past the last state (src/phylomap.cpp:85-89)");
  return PHM_OK;
}

// R's column-major matrix -> row-major
inline void cm_to_rm(const double* cm, int n, std::vector<double>& rm) {
  rm.resize((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) rm[(size_t)i * n + j] = cm[i + (size_t)j * n];
}

// squaring count of arma::expmat: s = max(0, exponent(frexp(log2 ||A||_inf)) + 1)
inline int pade_squarings(const double* Q_rm, int n, double t) {
  double norm = 0.0;
  for (int i = 0; i < n; ++i) {
    double r = 0.0;
    for (int j = 0; j < n; ++j) r += std::fabs(Q_rm[(size_t)i * n + j] * t);
    if (r > norm) norm = r;
  }
  double l2 = (norm > 0.0) ? std::log2(norm) : 0.0;
  int ex = 0;
  (void)std::frexp(l2, &ex);
  return std::max(0, ex + 1);
}

// Mat-vec of the MCMC chains on the host.  n <= 4: unfused left-to-right sums, what the n <= 4 kernels'
// matvec_u does.  n > 4: one fused multiply-add per term, j ascending, from +0 -- what v_mfma_f64_16x16x4 accumulates in
// the pruning kernel of phm_wtiles.hip and what phm_coop.h does lane-wise.  y = M x (row-major M).
inline void host_chain_matvec(const double* M, int n, const double* x, double* y, bool fused) {
  for (int i = 0; i < n; ++i) {
    if (fused) {
      double acc = 0.0;
      for (int c = 0; c < n; ++c) acc = std::fma(M[(size_t)i * n + c], x[c], acc);
      y[i] = acc;
    } else {
      double acc = M[i * n] * x[0];
      for (int c = 1; c < n; ++c) acc += M[i * n + c] * x[c];
      y[i] = acc;
    }
  }
}
// y = M^T x
inline void host_chain_matTvec(const double* M, int n, const double* x, double* y, bool fused) {
  for (int c = 0; c < n; ++c) {
    if (fused) {
      double acc = 0.0;
      for (int r = 0; r < n; ++r) acc = std::fma(M[(size_t)r * n + c], x[r], acc);
      y[c] = acc;
    } else {
      double acc = M[c] * x[0];
      for (int r = 1; r < n; ++r) acc += M[r * n + c] * x[r];
      y[c] = acc;
    }
  }
}

// chain tables: col[k][j][:] = Bc^k e_j  (v <- Bc v), row[k][j][:] = (Bc^T)^k e_j (w <- Bc^T w);
// the same sums as the kernels' own chains, so entries are bit-identical to running the chain.  `fused`: the MCMC sweep with
// n > 4; the sumstatEXP path (newunifSample :127) keeps the unfused left-to-right sums for every n.
// col[k][j] = Bc^k e_j, row[k][j] = (Bc^T)^k e_j (want_row).  The chain of a start vector j depends on nothing but itself, so the
// start vectors are dealt to a few host threads when the tables are big (61 states, 301 rows for sumstatEXP: 2 x 68 M
// multiply-adds, 57 ms of a call whose kernels take 1.5 ms) -- the same operations on the same operands per entry.
inline void build_chain_tables(const double* Bc, int n, int ktab, std::vector<double>& col, std::vector<double>& row, bool fused,
                               bool want_row = true) {
  col.assign((size_t)ktab * n * n, 0.0);
  row.assign(want_row ? (size_t)ktab * n * n : 0, 0.0);
  for (int j = 0; j < n; ++j) { col[(size_t)j * n + j] = 1.0; if (want_row) row[(size_t)j * n + j] = 1.0; }
  auto chains = [&](int j0, int j1) {
    for (int k = 1; k < ktab; ++k)
      for (int j = j0; j < j1; ++j) {
        host_chain_matvec(Bc, n, &col[((size_t)(k - 1) * n + j) * n], &col[((size_t)k * n + j) * n], fused);
        if (want_row) host_chain_matTvec(Bc, n, &row[((size_t)(k - 1) * n + j) * n], &row[((size_t)k * n + j) * n], fused);
      }
  };
  const double work = (double)ktab * n * n * n * (want_row ? 2 : 1);
  const int nt = work < 4e6 ? 1 : std::min<int>({8, n, (int)std::max(1u, std::thread::hardware_concurrency())});
  if (nt <= 1) { chains(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back(chains, (int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
  for (std::thread& t : th) t.join();
}

inline int32_t validate_tree_paths(const phm_tree* x, int n, int n_tip_vectors) {
  if (!x || !x->edge || !x->states || !x->map_off || !x->maps || !x->mapnames) return fail(PHM_ERR_BAD_INPUT, "tree: missing field");
  for (int64_t i = 0; i < (int64_t)n_tip_vectors * x->n_tips; ++i)
    if (x->states[i] < 1 || x->states[i] > n) return fail(PHM_ERR_BAD_INPUT, "x$states must be in 1..n");
  if (x->map_off[0] != 0) return fail(PHM_ERR_BAD_INPUT, "map_off[0] must be 0");
  for (int b = 0; b < x->n_edge; ++b) {
    int m = x->map_off[b + 1] - x->map_off[b];
    if (m < 1) return fail(PHM_ERR_BAD_INPUT, "every branch needs at least one segment in x$maps");
    if (m > 50000) return fail(PHM_ERR_BAD_INPUT, "more than 50000 segments on one branch");
    for (int i = x->map_off[b]; i < x->map_off[b + 1]; ++i) {
      if (x->mapnames[i] < 1 || x->mapnames[i] > n) return fail(PHM_ERR_BAD_INPUT, "x$mapnames must be in 1..n");
      if (!std::isfinite(x->maps[i]) || x->maps[i] < 0.0) return fail(PHM_ERR_BAD_INPUT, "x$maps must be finite and non-negative");
    }
  }
  return PHM_OK;
}

// The caller's inputs, copied at creation: the reference's std::list paths are unbounded (src/phylomap.cpp:18-21), the fixed HBM
// layouts here are not, so when a sweep outgrows its slots the engine is rebuilt with larger ones and the iterations run so far
// are replayed (every random number is addressed by (replica, iteration, entity): the replay is bit-identical).
struct SavedInput {
  struct TreeCopy {
    phm_tree t;
    std::vector<int32_t> edge, states, map_off, mapnames;
    std::vector<double> edge_length, maps;
  };
  std::vector<TreeCopy> trees;
  std::vector<phm_tree> flat;                      // the phm_tree array handed back to phm_engine_create_multi
  phm_model model;
  std::vector<double> Q, pid, B;
  phm_options opt;
  phm_debug_options dbg = {};
  int32_t max_iters = 0;
  std::vector<std::pair<int32_t, std::vector<double>>> model_hist;      // (first iteration it applies to, Q column-major): phm_engine_set_model calls
};

// -------------------------------------------------------------------------------------------------
struct phm_engine {
  std::shared_ptr<SavedInput> saved;
  phm_engine* fwd = nullptr;                       // set after a capacity recovery: every entry point continues on the rebuilt engine
  int cap_boost = 1;                               // multiplier of the provisioned slot / stream capacities (doubles per recovery)
  bool recover = true;                             // phm_options.no_recovery = 1 switches the recovery off (overflow -> PHM_ERR_CAPACITY)
  int recoveries = 0;
  bool dead = false;                               // a capacity recovery failed: no device state left, every entry point refuses
  int n = 0, cols = 0, dcols = 0, variant = 0;   // cols: result columns; dcols: columns kept on the device
  std::vector<double> qparams;                     // bf/ks: l01, l10, rkappas, lkappas, gammas of the CURRENT Q (recordQks :1789-1798)
  std::vector<std::vector<double>> qhist;          // ... as recorded at the start of every iteration that has run
  double Omega = 0.0;
  std::vector<double> hB2, hBc, hscale, hpid;      // current model, row-major
  int S = 0, S_pad = 0, tiles = 0, max_iters = 0, iters_done = 0, ipl = 0;
  int reduce = 0, device = 0;
  bool normalise = false;               // rows of the pruning pass divided by their sum (the variant's own rule, or phm_options.rescale_pruning)
  phm::Schedule sched;                 // tree 0 (every tree of a list has the same tip / edge counts)
  std::vector<phm::Schedule> scheds;   // one per tree
  int n_trees = 1, S_tree = 0, tpt = 0;   // list of trees: S_tree chains per tree on tpt tiles each; S = n_trees * S_tree
  DevBuf d_roots;
  // logical replica r (tree-major) -> lane index in the padded device layout
  int rpt = 64;                        // replicas placed on one tile (n > 4 spreads few replicas thinly, see phm_wide.hip)
  int pad_index(int r) const { return n_trees > 1 ? (r / S_tree) * tpt * 64 + r % S_tree : (r / rpt) * 64 + r % rpt; }
  std::vector<uint8_t> tips_host;      // 0-based, [n_tips] or [tile][n_tips][64]
  bool tips_per_replica = false;
  int64_t rows = 0;
  DevBuf d_mask;
  DevBuf d_up, d_down, d_col, d_row, d_tips, d_mcount, d_dw0, d_dw1, d_cursor, d_PL, d_nstate, d_stats, d_err, d_seg, d_red, d_red_out;
  phm::McmcParams<2> p2;
  phm::McmcParams<3> p3;
  phm::McmcParams<4> p4;
  bool wide = false;                   // 5..64 states: phm_wide.hip
  bool ring = true;                    // one ring per tile for both dwell streams (else two buffers)
  phm::WideParams pw;
  DevBuf d_B2, d_Bc, d_scale, d_pid;
  // branch-parallel mapping for few chains on a large tree (phm_narrow.hip)
  bool narrow = false;
  std::vector<int32_t> nw_walk_off;                // depth-level boundaries of the internal-child edges (walk of phm_narrow.hip)
  std::vector<int32_t> nw_tier_off;                // cluster tiers of the one-chain pruning sweep (phm_sched.h ClusterPlan)
  std::vector<int32_t> nw_up_off, nw_down_off;     // level boundaries into up_order / down_order
  std::vector<int64_t> nw_off;                     // CSR offsets of the branch slots
  int nw_klong = 0;
  bool nw_cluster_async = false;               // pruning clusters of the branch mapping in their dependency-driven form
  int nw_n_wide = 0;                           // branches that get a wavefront each in narrow_branch_kernel (narrow_setup)
  int64_t nw_total_cap = 0;
  DevBuf d_nw_up_off, d_nw_down_off, d_nw_up_order, d_nw_down_order, d_nw_border, d_nw_off, d_nw_colL, d_nw_rowL, d_nw_maskL, d_nw_mcount, d_nw_dwA, d_nw_dwB,
      d_nw_mstate, d_nw_mlen, d_nw_estate, d_nw_part, d_nw_rowbuf, d_nw_down_lv, d_nw_dmap, d_nw_dmap_edge, d_nw_walk_off, d_nw_edge_parent, d_nw_cl_nodes, d_nw_cl_item_off, d_nw_cl_lvl_ptr, d_nw_cl_lvl_off, d_ell_col, d_ell_val, d_ell2_col, d_ell2_val;
  DevBuf d_wb_cnt;
  phm::WideBranchParams pwb;                  // n > 4 with `narrow` set: one wave per (replica, branch) (phm_wbranch.hip)
  phm::NarrowParams<2> n2;
  phm::NarrowParams<3> n3;
  phm::NarrowParams<4> n4;
  // wave per (tile, branch) mapping for 10^2 .. 10^5 replicas (phm_tiles.hip); shares the level schedules and long tables
  bool tiled = false;
  std::vector<int32_t> tl_slot;                    // first row of every branch slot
  DevBuf d_tl_slot, d_tl_pdw, d_tl_pchunk, d_tl_cnt, d_tl_estate, d_tl_pseg, d_tl_segprev;
  phm::TileParams<2> t2;
  phm::TileParams<3> t3;
  phm::TileParams<4> t4;
  // 5..64 states with `tiled` set: one lane per replica, wave per (tile, item), pruning on the matrix cores (phm_wtiles.hip)
  phm::WtParams pwt;
  DevBuf d_wt_dwfx, d_wt_segacc, d_wt_B2, d_wt_totL, d_wt_pair_slot, d_wt_slot_col, d_wt_B2band, d_wt_mstate, d_wt_dwfx_tile, d_wt_cnt_tile;
  phm::WtBand wt_band;                            // band of the chain matrix (kernel-argument constants of wt_up_band_kernel)
  phm::WtSparseUp wt_sparse;                      // pruning kernel generated for the pattern of an unstructured sparse chain matrix (phm_rtc.h)
  DevBuf d_wt_coef;                               // ... and its coefficients (the non-zeros row by row)
  int sparse_req = 0;                              // phm_options.sparse_chains
  phm_debug_options dbg = {};                      // the creating thread's phm_set_debug_options at creation
  bool phase_timing = false;                       // phm_debug_options.phase_timing: HIP events between the phases of a (tile, item) sweep
  std::vector<hipEvent_t> phase_ev;                // 5 per iteration of the last run
  int phase_iters = 0;
  double phase_ms[4] = {0.0, 0.0, 0.0, 0.0};       // pruning levels, node draws, branch kernel, reductions (sums over the last run)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t last_stream = nullptr;
  PinnedBuf pin_up, pin_down, pin_status, pin_row;
  int epi_iter = -1;                               // iteration whose statistics row / status words sit in pin_row (one-chain engines)
                      // staging of the model tables (host -> device) and of small statistics reads
  bool timing_pending = false;
  double last_ms = 0.0;
  int last_launches = 0;
  int64_t bytes = 0;
  unsigned long long seg_total = 0;
  // frees every device buffer (the host-side description of the problem stays: sched, cols, ...)
  void release_device() {
    DevBuf* all[] = {&d_roots, &d_mask, &d_up, &d_down, &d_col, &d_row, &d_tips, &d_mcount, &d_dw0, &d_dw1, &d_cursor, &d_PL, &d_nstate,
                     &d_stats, &d_err, &d_seg, &d_red, &d_red_out, &d_B2, &d_Bc, &d_scale, &d_pid, &d_nw_up_off, &d_nw_down_off,
                     &d_nw_up_order, &d_nw_down_order, &d_nw_border, &d_nw_off, &d_nw_colL, &d_nw_rowL, &d_nw_maskL, &d_nw_mcount,
                     &d_nw_dwA, &d_nw_dwB, &d_nw_mstate, &d_nw_mlen, &d_nw_estate, &d_nw_part, &d_nw_rowbuf, &d_nw_down_lv, &d_nw_dmap, &d_nw_dmap_edge, &d_nw_walk_off, &d_nw_edge_parent, &d_nw_cl_nodes, &d_nw_cl_item_off, &d_nw_cl_lvl_ptr, &d_nw_cl_lvl_off, &d_ell_col, &d_ell_val,
                     &d_ell2_col, &d_ell2_val, &d_wb_cnt, &d_tl_slot, &d_tl_pdw, &d_tl_pchunk, &d_tl_cnt, &d_tl_estate, &d_tl_pseg,
                     &d_tl_segprev, &d_wt_dwfx, &d_wt_segacc, &d_wt_B2, &d_wt_totL, &d_wt_pair_slot, &d_wt_slot_col, &d_wt_B2band, &d_wt_mstate, &d_wt_dwfx_tile, &d_wt_cnt_tile, &d_wt_coef};
    for (DevBuf* b : all) b->reset();
  }
  ~phm_engine() {
    delete fwd;
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (hipEvent_t ev : phase_ev) (void)hipEventDestroy(ev);
  }
};

// the constructor behind phm_engine_create / phm_engine_create_multi (phm_engine.cpp)
int32_t phm_engine_create_impl(const phm_tree* trees, int32_t n_trees, const phm_model* model, const phm_options* opt_in,
                               const phm_debug_options& dbg, int boost_log2, int32_t max_iters, phm_engine** out);
int32_t phm_engine_fold_reduced(phm_engine* e, int32_t iter0, int32_t n, std::vector<double>& acc);
int32_t phm_engine_finish_reduced(phm_engine* e, int32_t iter0, int32_t n, const std::vector<double>& acc, double* out);

// multi-device one-shot calls (phm_drivers.cpp): contiguous ranges of `units` (replicas / sites / EXP samples) per device
struct phm_shard { int32_t device; int64_t first, count; };
int32_t phm_plan_shards(const phm_options& o, int64_t units, std::vector<phm_shard>& shards);
