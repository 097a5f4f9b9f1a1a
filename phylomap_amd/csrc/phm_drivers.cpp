// phm_drivers.cpp -- the reference-shaped entry points (one per exported driver of src/phylomap.cpp) on top of the engine:
// fixed-Q drivers, the Q-updating drivers (bf / ks / DIC) and the multi-tree drivers (mt / ksmt); host-side glue only.
#include "phm_internal.h"

#include <chrono>
#include <cstdio>
#include <thread>

// ---- replica sharding over the GPUs of a node (phm_options.n_devices) ------------------------------------------------------
// The reference's caller is ONE R function -> .Call -> ONE C++ driver (R/sumstatMCMC_bigtree.R:21-29 -> src/phylomap.cpp:942-986);
// to reach the other GPUs of the node the sharding has to live below the C-ABI.  Chains are independent given (seed, global
// replica id), so device d gets a contiguous range of replica ids, runs it on an engine of its own (one host thread per device,
// nothing crosses between devices while sampling) and the only exchange is the N x cols statistics at the end.
int32_t phm_plan_shards(const phm_options& o, int64_t units, std::vector<phm_shard>& shards) {
  shards.clear();
  const int D = o.n_devices;
  if (D < 0 || D > PHM_MAX_DEVICES) return fail(PHM_ERR_BAD_INPUT, "n_devices must be in 0..PHM_MAX_DEVICES");
  if (D <= 1) { shards.push_back({D == 1 ? o.devices[0] : o.device, 0, units}); return PHM_OK; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PHM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  for (int d = 0; d < D; ++d)
    if (o.devices[d] < 0 || o.devices[d] >= ndev) return fail(PHM_ERR_NO_DEVICE, "phm_options.devices: ordinal out of range");
  // whole 64-lane tiles per device when there are enough of them (a tile then holds the same replicas as on one device: the
  // per-tile sums are the one-device ones), single replicas otherwise (a handful of chains: one or a few per GPU)
  const int64_t grain = units >= (int64_t)64 * D ? 64 : 1;
  const int64_t blocks = (units + grain - 1) / grain;
  for (int d = 0; d < D; ++d) {
    const int64_t b0 = blocks * d / D, b1 = blocks * (d + 1) / D;
    const int64_t r0 = std::min(units, b0 * grain), r1 = std::min(units, b1 * grain);
    if (r1 > r0) shards.push_back({o.devices[d], r0, r1 - r0});
  }
  return PHM_OK;
}

namespace {

struct ShardRun {
  phm_shard sh;
  phm_engine* e = nullptr;
  hipStream_t stream = nullptr;
  int32_t st = PHM_OK;
  std::string err;
};

struct ShardSet {      // engines and streams of a sharded call; destroyed on every exit path
  std::vector<ShardRun> runs;
  ~ShardSet() {
    for (ShardRun& r : runs) {
      if (r.e) phm_engine_destroy(r.e);
      if (r.stream) { (void)hipSetDevice(r.sh.device); (void)hipStreamDestroy(r.stream); }
    }
  }
  int32_t first_error() const {
    for (const ShardRun& r : runs) if (r.st) return fail(r.st, "device " + std::to_string(r.sh.device) + ": " + r.err);
    return PHM_OK;
  }
};

// one host thread per shard: engine on the shard's device for its replica range (global ids: replica_offset + first), then
// `body(run)` -- typically run + sync (+ read).  Worker threads report through ShardRun (phm_last_error is thread-local).
template <typename Body>
int32_t run_shards(ShardSet& set, const phm_tree* x, const phm_model& model, const phm_options& base, int32_t max_iters, Body body) {
  const phm_debug_options dbg = g_phm_debug;
  auto work = [&](ShardRun& r) {
    phm_options o = base;
    o.n_devices = 0; o.device = r.sh.device;
    o.n_replicas = (int32_t)r.sh.count; o.replica_offset = base.replica_offset + (int32_t)r.sh.first;
    phm_tree xt = *x;
    if (base.tips_per_replica) xt.states = x->states + (size_t)r.sh.first * x->n_tips;
    r.st = phm_engine_create_impl(&xt, 1, &model, &o, dbg, 0, max_iters, &r.e);
    if (!r.st && hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess) r.st = fail(PHM_ERR_NO_DEVICE, "hipStreamCreate failed");
    if (!r.st) r.st = body(r);
    if (r.st) r.err = g_phm_err;
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < set.runs.size(); ++i) th.emplace_back(work, std::ref(set.runs[i]));
  work(set.runs[0]);
  for (std::thread& t : th) t.join();
  return set.first_error();
}

}  // namespace

extern "C" {

// ---- reference-shaped one-shot drivers -------------------------------------------------------------
static int32_t run_mcmc_sharded(int variant, const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                const phm_options& o, double* out) {
  phm_model model;
  model.n_states = n; model.Q = Q; model.pid = pid; model.B = B; model.Omega = Omega; model.variant = variant;
  if (!x) return fail(PHM_ERR_BAD_INPUT, "tree is NULL");
  const int S = std::max(1, (int)o.n_replicas);
  ShardSet set;
  {
    std::vector<phm_shard> shards;
    int32_t st = phm_plan_shards(o, S, shards);
    if (st) return st;
    for (const phm_shard& sh : shards) { set.runs.emplace_back(); set.runs.back().sh = sh; }
  }
  int cols = 0;
  bool orders_ok = true;
  std::string serr;
  int32_t st = run_shards(set, x, model, o, N, [&](ShardRun& r) -> int32_t {
    if (&r == &set.runs[0]) {      // the caller's nen / nodelist / root against the tree (host-only check, once)
      cols = r.e->cols;
      orders_ok = phm::check_reference_orders(r.e->sched, x->edge, nen, nodelist, root, serr);
      if (!orders_ok) return fail(PHM_ERR_BAD_INPUT, serr);
    }
    int32_t s2 = phm_engine_run(r.e, N, r.stream);
    if (!s2) s2 = phm_engine_sync(r.e);
    if (!s2 && !o.reduce) s2 = phm_engine_read_stats(r.e, 0, N, out + (size_t)r.sh.first * N * r.e->cols);
    return s2;
  });
  if (st) return st;
  if (!o.reduce) return PHM_OK;
  std::vector<double> acc;                              // the fold of the per-tile sums, device after device
  for (ShardRun& r : set.runs) { st = phm_engine_fold_reduced(r.e, 0, N, acc); if (st) return st; }
  return phm_engine_finish_reduced(set.runs[0].e, 0, N, acc, out);
}

static int32_t run_mcmc_oneshot(int variant, const phm_tree* x, int32_t n, const double* Q, const double* pid,
                                const double* B, double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root,
                                int32_t N, const phm_options* opt, double* out) {
  if (!out) return fail(PHM_ERR_BAD_INPUT, "out is NULL");
  if (N < 1) return fail(PHM_ERR_BAD_INPUT, "N must be >= 1");
  if (opt && opt->n_devices > 1) return run_mcmc_sharded(variant, x, n, Q, pid, B, Omega, nen, nodelist, root, N, *opt, out);
  if (opt && opt->n_devices < 0) return fail(PHM_ERR_BAD_INPUT, "n_devices must be in 0..PHM_MAX_DEVICES");
  phm_options one;
  if (opt && opt->n_devices == 1) { one = *opt; one.device = one.devices[0]; one.n_devices = 0; opt = &one; }
  phm_model model;
  model.n_states = n; model.Q = Q; model.pid = pid; model.B = B; model.Omega = Omega; model.variant = variant;
  phm_engine* e = nullptr;
  int32_t st = phm_engine_create(x, &model, opt, N, &e);
  if (st) return st;
  std::string serr;
  if (!phm::check_reference_orders(e->sched, x->edge, nen, nodelist, root, serr)) { phm_engine_destroy(e); return fail(PHM_ERR_BAD_INPUT, serr); }
  st = phm_engine_run(e, N, nullptr);
  if (!st) st = phm_engine_sync(e);
  if (!st) st = phm_engine_read_stats(e, 0, N, out);
  phm_engine_destroy(e);
  return st;
}

int32_t phm_maketreelistMCMC(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B, double Omega,
                             const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const phm_options* opt,
                             double* out) {
  return run_mcmc_oneshot(PHM_MCMC, x, n, Q, pid, B, Omega, nen, nodelist, root, N, opt, out);
}
int32_t phm_maketreelistMCMC_bigtree(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                     double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                     const phm_options* opt, double* out) {
  return run_mcmc_oneshot(PHM_MCMC_BIGTREE, x, n, Q, pid, B, Omega, nen, nodelist, root, N, opt, out);
}
int32_t phm_maketreelistMCMCks_sweep(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                     double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                     const phm_options* opt, double* out) {
  return run_mcmc_oneshot(PHM_MCMC_KS, x, n, Q, pid, B, Omega, nen, nodelist, root, N, opt, out);
}
int32_t phm_maketreelistMCMCbf_sweep(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                     double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                     const phm_options* opt, double* out) {
  return run_mcmc_oneshot(PHM_MCMC_BF, x, n, Q, pid, B, Omega, nen, nodelist, root, N, opt, out);
}
int32_t phm_SPARSEmaketreelistMCMC(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                   double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                   const phm_options* opt, double* out) {
  return run_mcmc_oneshot(PHM_MCMC_SPARSE, x, n, Q, pid, B, Omega, nen, nodelist, root, N, opt, out);
}

}  // extern "C"

// the HIP source of the pruning kernel generated for the non-zero pattern of M (phm_rtc.h); inspection / build checks
extern "C" int32_t phm_sparse_kernel_source(int32_t n, const double* M, char* buf, int32_t cap) {
  if (!M || n < 2 || n > phm::RTC_SPARSE_NMAX) return fail(PHM_ERR_BAD_INPUT, "phm_sparse_kernel_source: 2 <= n_states <= 32, M not NULL") ? -1 : -1;
  std::vector<int32_t> rp(1, 0), cj;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) if (M[i + (size_t)j * n] != 0.0) cj.push_back(j);
    rp.push_back((int32_t)cj.size());
  }
  const std::string src = phm::rtc_sparse_up_source(n, rp, cj);
  if (buf && cap > 0) {
    const size_t k = std::min((size_t)cap - 1, src.size());
    std::memcpy(buf, src.data(), k);
    buf[k] = '\0';
  }
  return (int32_t)src.size() + 1;
}

// Native O(E) replacement of pruningwiseedgeorder / makenodelist / myreorder (R/sumstatMCMC.R:1-18); pure host code.
extern "C" int32_t phm_tree_orders(int32_t n_tips, int32_t n_edge, const int32_t* edge, int32_t* nen, int32_t* nodelist,
                                   int32_t* root) {
  if (!edge || !nen || !nodelist || !root) return fail(PHM_ERR_BAD_INPUT, "phm_tree_orders: NULL argument");
  std::string serr;
  if (!phm::pruningwise_orders(n_tips, n_edge, edge, nen, nodelist, root, serr)) return fail(PHM_ERR_BAD_INPUT, serr);
  return PHM_OK;
}

// ---- Q-updating drivers: sweep on the device, rate-matrix update on the host, every iteration ---------------------------
// maketreelistMCMCbf src/phylomap.cpp:1258-1305 (R/sumstatMCMCbf.R) and maketreelistMCMCks :1802-1872 (R/sumstatMCMCks.R).
// With opt->n_replicas = S > 1 the replicas are sites sharing one Q: the update sees the statistics summed over sites and
// `out` holds those sums (S = 1 is the reference's semantics exactly).
// The rate-updating drivers with S > 1 sites sharing Q on several GPUs: every iteration each device sweeps its sites with the
// current Q, the host adds the devices' (site-summed) rows in device order, draws the new rates from the total and hands the
// new model to every device.  One row of n + n^2 + 1 doubles per device and iteration is all that moves.
static int32_t run_qupdate_sharded(int variant, const phm_tree* x, int32_t n, const double* Q, const double* pid, double Omega,
                                   const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const double* prior,
                                   const phm_model& model, const phm_options& o, double* out) {
  (void)pid;
  ShardSet set;
  {
    std::vector<phm_shard> shards;
    int32_t st = phm_plan_shards(o, o.n_replicas, shards);
    if (st) return st;
    for (const phm_shard& sh : shards) { set.runs.emplace_back(); set.runs.back().sh = sh; }
  }
  std::string serr;
  int32_t st = run_shards(set, x, model, o, N, [&](ShardRun& r) -> int32_t {
    if (&r == &set.runs[0] && !phm::check_reference_orders(r.e->sched, x->edge, nen, nodelist, root, serr)) return fail(PHM_ERR_BAD_INPUT, serr);
    return PHM_OK;
  });
  if (st) return st;
  const int ecols = set.runs[0].e->cols;
  const size_t nn = (size_t)n * n;
  std::vector<double> Qw(Q, Q + nn), row(ecols), part(ecols);
  for (int i = 0; i < N; ++i) {
    for (ShardRun& r : set.runs) { st = phm_engine_run(r.e, 1, r.stream); if (st) return st; }
    for (size_t d = 0; d < set.runs.size(); ++d) {
      ShardRun& r = set.runs[d];
      st = phm_engine_sync(r.e);
      if (!st) st = phm_engine_read_stats(r.e, i, 1, d == 0 ? row.data() : part.data());
      if (st) return st;
      if (d > 0) {      // dwell sums, counts and the (site-summed) root-state column; the parameter columns are the same on every device
        for (int c = 0; c < n + (int)nn; ++c) row[c] += part[c];
        row[ecols - 1] += part[ecols - 1];
      }
    }
    for (int c = 0; c < ecols; ++c) out[(size_t)c * N + i] = row[c];
    if (variant == PHM_MCMC_BF) phm::bf_updates(Qw.data(), Omega, prior, row.data(), o.seed, (uint32_t)i);
    else phm::ks_updates(Qw.data(), n, Omega, prior, row.data(), o.seed, (uint32_t)i);
    if (i + 1 < N) for (ShardRun& r : set.runs) { st = phm_engine_set_model(r.e, Qw.data()); if (st) return st; }
  }
  return PHM_OK;
}

static int32_t run_qupdate(int variant, bool dic, const phm_tree* x, int32_t n, const double* Q, const double* pid,
                           const double* B, double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                           const double* prior, int32_t n_prior, const phm_options* opt_in, double* out) {
  if (!out || !prior || !Q) return fail(PHM_ERR_BAD_INPUT, "out/prior/Q is NULL");
  if (N < 1) return fail(PHM_ERR_BAD_INPUT, "N must be >= 1");
  const int need = (variant == PHM_MCMC_BF) ? 4 : 6;
  if (n_prior < need) return fail(PHM_ERR_BAD_INPUT, variant == PHM_MCMC_BF ? "the two-state drivers need prior = c(a01, b01, a10, b10)" : "the hidden-rates drivers need prior = c(a_l, b_l, a_k, b_k, a_g, b_g)");
  if (variant == PHM_MCMC_BF && n != 2) return fail(PHM_ERR_BAD_INPUT, "sumstatMCMCbf / sumstatMCMC2sDICt are two-state models (9 hard-wired columns and the two-rate updates, src/phylomap.cpp:1129, :1181-1253, :1293); the sweep alone takes any n: phm_maketreelistMCMCbf_sweep");
  if (variant == PHM_MCMC_KS && (n < 4 || (n & 1))) return fail(PHM_ERR_BAD_INPUT, "sumstatMCMCks needs n = 2k+2 states with k >= 1 (src/phylomap.cpp:1820; updateksl01 reads rkappas(0))");
  if (dic && (!x || !x->edge_length || !nen)) return fail(PHM_ERR_BAD_INPUT, "the DIC drivers need x$edge.length and nen (src/phylomap.cpp:3223, :3158)");
  phm_options o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opt_in) o = *opt_in;
  if (o.n_replicas <= 0) o.n_replicas = 1;
  if (dic && o.n_replicas != 1) return fail(PHM_ERR_UNSUPPORTED, "the DIC drivers run one chain (log p(y|Q) is per data set)");
  o.reduce = o.n_replicas > 1;     // one chain: its own statistics, accumulated in the reference's order (bit-exact vs the oracle)
  o.iters_per_launch = 1;
  (void)B;     // the reference aliases the caller's B and then overwrites it entry by entry; B = I + Q/Omega throughout
  phm_model model;
  model.n_states = n; model.Q = Q; model.pid = pid; model.B = nullptr; model.Omega = Omega; model.variant = variant;
  if (o.n_devices < 0 || o.n_devices > PHM_MAX_DEVICES) return fail(PHM_ERR_BAD_INPUT, "n_devices must be in 0..PHM_MAX_DEVICES");
  if (o.n_devices > 1 && o.n_replicas > 1 && !dic)
    return run_qupdate_sharded(variant, x, n, Q, pid, Omega, nen, nodelist, root, N, prior, model, o, out);
  if (o.n_devices >= 1) { o.device = o.devices[0]; o.n_devices = 0; }      // one chain lives on one device
  phm_engine* e = nullptr;
  int32_t st = phm_engine_create(x, &model, &o, N, &e);
  if (st) return st;
  std::unique_ptr<phm_engine, void (*)(phm_engine*)> guard(e, phm_engine_destroy);
  std::string serr;
  if (!phm::check_reference_orders(e->sched, x->edge, nen, nodelist, root, serr)) return fail(PHM_ERR_BAD_INPUT, serr);
  const int ecols = e->cols, E = e->sched.n_edge, T = e->sched.n_tips, Nn = e->sched.n_node;
  const size_t nn = (size_t)n * n;

  // DIC: device state of the per-iteration log-likelihood (expmat(Q t_b) for every branch, then pruning in nen order)
  DevBuf dQ, dt, ds, dwork, dP, dPL0, dPL, dpid, dup, dll, derr, dorder, dlogs;
  std::vector<double> loglik;
  PinnedBuf pin_dic, pin_ll;                        // staging of (Q, squarings); log p(y|Q) of every iteration, written by the device
  double* ll_dev = nullptr;
  std::vector<int32_t> ll_level_off;
  std::vector<int32_t> sq(E);
  if (dic) {
    std::vector<phm::UpStep> upn(Nn);
    const int32_t* e1 = x->edge; const int32_t* e2 = x->edge + E;
    auto code = [&](int32_t node) { return node > T ? node - T - 1 : ~(node - 1); };
    std::vector<int32_t> height(Nn, 0);
    int max_h = 0;
    for (int i = 0; i < Nn; ++i) {
      const int ea = nen[2 * i] - 1, eb = nen[2 * i + 1] - 1;
      upn[i].parent = e1[ea] - T - 1;
      upn[i].child[0] = code(e2[ea]); upn[i].child[1] = code(e2[eb]);
      upn[i].edge[0] = ea; upn[i].edge[1] = eb;
      int h = 0;                                        // nen lists children before parents (checked above)
      for (int c = 0; c < 2; ++c) if (upn[i].child[c] >= 0) h = std::max(h, height[upn[i].child[c]] + 1);
      height[upn[i].parent] = h; max_h = std::max(max_h, h);
    }
    ll_level_off.assign(max_h + 2, 0);
    for (int i = 0; i < Nn; ++i) ll_level_off[height[upn[i].parent] + 1]++;
    for (size_t l = 1; l < ll_level_off.size(); ++l) ll_level_off[l] += ll_level_off[l - 1];
    std::vector<int32_t> ll_order(Nn), pos(ll_level_off.begin(), ll_level_off.end() - 1);
    for (int i = 0; i < Nn; ++i) ll_order[pos[height[upn[i].parent]]++] = i;
    HIPCHK(dorder.alloc(sizeof(int32_t) * Nn)); HIPCHK(dlogs.alloc(sizeof(double) * Nn));
    HIPCHK(hipMemcpy(dorder.p, ll_order.data(), dorder.bytes, hipMemcpyHostToDevice));
    std::vector<double> PLh((size_t)(2 * T - 1) * n, 0.0);
    for (int t = 0; t < T; ++t) {
      if (variant == PHM_MCMC_BF) PLh[(size_t)t * n + (x->states[t] - 1)] = 1.0;                         // :3165
      else for (int j = (x->states[t] % 2 == 0) ? 1 : 0; j < n; j += 2) PLh[(size_t)t * n + j] = 1.0;   // :3275-3282
    }
    HIPCHK(dQ.alloc(sizeof(double) * nn)); HIPCHK(dt.alloc(sizeof(double) * E)); HIPCHK(ds.alloc(sizeof(int32_t) * E));
    HIPCHK(dwork.alloc(sizeof(double) * nn * 5 * E)); HIPCHK(dP.alloc(sizeof(double) * nn * E));
    HIPCHK(dPL0.alloc(sizeof(double) * PLh.size())); HIPCHK(dPL.alloc(sizeof(double) * PLh.size()));
    HIPCHK(dpid.alloc(sizeof(double) * n)); HIPCHK(dup.alloc(sizeof(phm::UpStep) * Nn)); HIPCHK(dll.alloc(sizeof(double)));
    HIPCHK(derr.alloc(sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dt.p, x->edge_length, dt.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dPL0.p, PLh.data(), dPL0.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dpid.p, pid, dpid.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dup.p, upn.data(), dup.bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(derr.p, 0, sizeof(uint32_t)));
    loglik.resize(N);
    HIPCHK(pin_dic.reserve(sizeof(double) * nn + sizeof(int32_t) * E));
    HIPCHK(pin_ll.reserve(sizeof(double) * N));
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, pin_ll.p, 0));
    ll_dev = reinterpret_cast<double*>(dp);
  }

  std::vector<double> Qw(Q, Q + nn), Qr, row(ecols);
  // phm_debug_options.q_timing (measurement aid): mean host time of the phases of an iteration, printed once at the end
  const bool qtiming = g_phm_debug.q_timing != 0;
  double t_run = 0, t_sync = 0, t_read = 0, t_upd = 0, t_set = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int i = 0; i < N && !st; ++i) {
    const double t0 = qtiming ? now() : 0;
    st = phm_engine_run(e, 1, nullptr);
    const double t1 = qtiming ? now() : 0;
    if (!st) st = phm_engine_sync(e);
    const double t2 = qtiming ? now() : 0;
    if (!st) st = phm_engine_read_stats(e, i, 1, row.data());
    const double t3 = qtiming ? now() : 0;
    t_run += t1 - t0; t_sync += t2 - t1; t_read += t3 - t2;
    if (st) break;
    if (dic) {                                        // :3239-3251 / :3379-3391, with the Q that drove this sweep
      // nothing here waits: Q and the squarings go up from page-locked staging, log p(y|Q) of iteration i lands in slot i of a
      // page-locked array; the wait of the next model update (same stream) covers the kernels before the staging is reused
      cm_to_rm(Qw.data(), n, Qr);
      double* stQ = pin_dic.as<double>();
      int32_t* stS = reinterpret_cast<int32_t*>(stQ + nn);
      std::memcpy(stQ, Qr.data(), sizeof(double) * nn);
      for (int b = 0; b < E; ++b) stS[b] = pade_squarings(Qr.data(), n, x->edge_length[b]);
      HIPCHK(hipMemcpyAsync(dQ.p, stQ, dQ.bytes, hipMemcpyHostToDevice, nullptr));
      HIPCHK(hipMemcpyAsync(ds.p, stS, ds.bytes, hipMemcpyHostToDevice, nullptr));
      HIPCHK(hipMemcpyAsync(dPL.p, dPL0.p, dPL.bytes, hipMemcpyDeviceToDevice, nullptr));
      HIPCHK(phm::launch_expm_pade(n, dQ.as<double>(), dt.as<double>(), ds.as<int32_t>(), E, dwork.as<double>(), dP.as<double>(), derr.as<uint32_t>(), nullptr));
      HIPCHK(phm::launch_exp_pl_loglik(n, Nn, T, dup.as<phm::UpStep>(), dorder.as<int32_t>(), ll_level_off, dP.as<double>(), dPL.as<double>(),
                                       dlogs.as<double>(), dpid.as<double>(), root - 1, ll_dev + i, nullptr));
    }
    const double t4 = qtiming ? now() : 0;
    if (variant == PHM_MCMC_BF) phm::bf_updates(Qw.data(), Omega, prior, row.data(), o.seed, (uint32_t)i);
    else phm::ks_updates(Qw.data(), n, Omega, prior, row.data(), o.seed, (uint32_t)i);
    const double t5 = qtiming ? now() : 0;
    if (i + 1 < N) st = phm_engine_set_model(e, Qw.data());
    t_upd += t5 - t4; t_set += (qtiming ? now() : 0) - t5;
  }
  if (qtiming)
    std::fprintf(stderr, "phm qtiming (us per iteration): launch %.1f  wait for the sweep %.1f  read row %.1f  rate updates %.1f  new model %.1f\n",
                 t_run / N, t_sync / N, t_read / N, t_upd / N, t_set / N);
  if (st) return st;
  if (!dic) return phm_engine_read_stats(e, 0, N, out);
  HIPCHK(hipStreamSynchronize(nullptr));
  std::memcpy(loglik.data(), pin_ll.p, sizeof(double) * N);
  std::vector<double> tmp((size_t)N * ecols);
  st = phm_engine_read_stats(e, 0, N, tmp.data());
  if (st) return st;
  std::memcpy(out, tmp.data(), sizeof(double) * tmp.size());           // column-major: the first ecols columns are unchanged
  for (int i = 0; i < N; ++i) out[(size_t)ecols * N + i] = loglik[i];     // log p(y|Q) after the root-state column
  return PHM_OK;
}

extern "C" int32_t phm_maketreelistMCMCbf(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                          double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                          const double* prior, int32_t n_prior, const phm_options* opt, double* out) {
  return run_qupdate(PHM_MCMC_BF, false, x, n, Q, pid, B, Omega, nen, nodelist, root, N, prior, n_prior, opt, out);
}

extern "C" int32_t phm_maketreelistMCMCks(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                          double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                          const double* prior, int32_t n_prior, const phm_options* opt, double* out) {
  return run_qupdate(PHM_MCMC_KS, false, x, n, Q, pid, B, Omega, nen, nodelist, root, N, prior, n_prior, opt, out);
}

// Host-only: apply one iteration's rate-matrix updates to Q (column-major, edited in place) given a statistics row
// (n dwell sums, n*n counts).  What phm_maketreelistMCMCbf / ks run between sweeps; exported for CPU-side tests.
extern "C" int32_t phm_qupdate_apply(int32_t variant, int32_t n, double* Q, double Omega, const double* prior, int32_t n_prior,
                                     const double* row, uint64_t seed, uint32_t iter) {
  if (!Q || !prior || !row) return fail(PHM_ERR_BAD_INPUT, "phm_qupdate_apply: NULL argument");
  if (variant == PHM_MCMC_BF || variant == PHM_MCMC_MT) {
    if (n != 2 || n_prior < 4) return fail(PHM_ERR_BAD_INPUT, "bf / mt: n = 2, prior[4]");
    if (variant == PHM_MCMC_MT) phm::mt_updates(Q, Omega, prior, row, seed, iter);
    else phm::bf_updates(Q, Omega, prior, row, seed, iter);
  } else if (variant == PHM_MCMC_KS || variant == PHM_MCMC_KSMT) {
    const bool mt = variant == PHM_MCMC_KSMT;
    if (n < 4 || (n & 1) || n > 64 || n_prior < (mt ? 8 : 6)) return fail(PHM_ERR_BAD_INPUT, "ks: n = 2k+2 in 4..64, prior[6] (ksmt: prior[8])");
    phm::ks_updates(Q, n, Omega, prior, row, seed, iter, mt);
  } else return fail(PHM_ERR_BAD_INPUT, "variant must be PHM_MCMC_BF, PHM_MCMC_KS, PHM_MCMC_MT or PHM_MCMC_KSMT");
  return PHM_OK;
}

// maketreelistMCMC2sDICt src/phylomap.cpp:3183-3264 and maketreelistMCMCksDICt :3300-3403: the bf / ks drivers plus, every
// iteration, log p(y|Q) by matrix exponentiation (expmat(Q t_b) for every branch, pruning with scale factors) in one more column.
extern "C" int32_t phm_maketreelistMCMC2sDICt(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                              double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                              const double* prior, int32_t n_prior, const phm_options* opt, double* out) {
  return run_qupdate(PHM_MCMC_BF, true, x, n, Q, pid, B, Omega, nen, nodelist, root, N, prior, n_prior, opt, out);
}

extern "C" int32_t phm_maketreelistMCMCksDICt(const phm_tree* x, int32_t n, const double* Q, const double* pid, const double* B,
                                              double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                              const double* prior, int32_t n_prior, const phm_options* opt, double* out) {
  return run_qupdate(PHM_MCMC_KS, true, x, n, Q, pid, B, Omega, nen, nodelist, root, N, prior, n_prior, opt, out);
}

// ---- multi-tree drivers -----------------------------------------------------------------------------------------------
// The list as one engine PER TREE (big trees; run_qupdate_mt decides): every iteration enqueues one sweep of every engine on a
// handful of streams, waits for all of them and reads a row from each; the rate update and the new model go to all engines.
static int32_t run_qupdate_mt_per_tree(int variant, const phm_tree* trees, int32_t n_trees, int32_t n, const phm_model& model,
                                       const phm_options& o_in, const int32_t* nen_m, const int32_t* nodelist_m, const int32_t* roots,
                                       int32_t N, const double* prior, double* out) {
  const bool ksmt = variant == PHM_MCMC_KSMT;
  struct Engines {
    std::vector<phm_engine*> e;
    std::vector<hipStream_t> streams;
    ~Engines() {
      for (phm_engine* x : e) if (x) phm_engine_destroy(x);
      for (hipStream_t s : streams) if (s) (void)hipStreamDestroy(s);
    }
  } es;
  es.e.assign(n_trees, nullptr);
  std::string serr;
  for (int j = 0; j < n_trees; ++j) {
    phm_options o = o_in;
    o.replica_offset = o_in.replica_offset + 64 * j;      // the list's engine keeps tree j's chain on replica tile j
    o.mapping = PHM_MAP_BRANCHES;
    const int32_t st = phm_engine_create(&trees[j], &model, &o, N, &es.e[j]);
    if (st) return fail(st, "tree " + std::to_string(j) + ": " + g_phm_err);
    const phm_engine* e = es.e[j];
    const int Nn = e->sched.n_node;
    if (nen_m || nodelist_m || roots) {      // R's matrices are column-major: row j = elements j, j + n_trees, ...
      std::vector<int32_t> nen(2 * (size_t)Nn), nodelist(Nn > 1 ? Nn - 1 : 0);
      if (nen_m) for (int i = 0; i < 2 * Nn; ++i) nen[i] = nen_m[j + (size_t)i * n_trees];
      if (nodelist_m) for (int i = 0; i < Nn - 1; ++i) nodelist[i] = nodelist_m[j + (size_t)i * n_trees];
      if (!phm::check_reference_orders(e->sched, trees[j].edge, nen_m ? nen.data() : nullptr, nodelist_m ? nodelist.data() : nullptr,
                                       roots ? roots[j] : e->sched.root + e->sched.n_tips + 1, serr))
        return fail(PHM_ERR_BAD_INPUT, "tree " + std::to_string(j) + ": " + serr);
    }
  }
  const int n_streams = std::min(n_trees, 16);
  es.streams.assign(n_streams, nullptr);
  for (int k = 0; k < n_streams; ++k)
    if (hipStreamCreateWithFlags(&es.streams[k], hipStreamNonBlocking) != hipSuccess) return fail(PHM_ERR_NO_DEVICE, "hipStreamCreate failed");
  const int ecols = es.e[0]->cols;
  const size_t nn = (size_t)n * n;
  std::vector<double> Qw(model.Q, model.Q + nn), row(ecols);
  for (int i = 0; i < N; ++i) {
    int32_t st = PHM_OK;
    for (int j = 0; j < n_trees && !st; ++j) st = phm_engine_run(es.e[j], 1, es.streams[j % n_streams]);
    for (int j = 0; j < n_trees && !st; ++j) st = phm_engine_sync(es.e[j]);
    if (st) return st;
    const uint32_t pick = phm::pick_tree(n_trees, o_in.seed, (uint32_t)i);
    if (pick >= (uint32_t)n_trees) return fail(PHM_ERR_ZERO_PROB, "sampleOnce ran past the last tree (src/phylomap.cpp:85-89)");
    st = phm_engine_read_stats(es.e[pick], i, 1, row.data());
    if (st) return st;
    for (int c = 0; c + 1 < ecols; ++c) out[(size_t)c * N + i] = row[c];
    out[(size_t)(ecols - 1) * N + i] = (double)pick;                   // :2350, 0-based as the reference stores it
    if (ksmt) phm::ks_updates(Qw.data(), n, model.Omega, prior, row.data(), o_in.seed, (uint32_t)i, true);
    else phm::mt_updates(Qw.data(), model.Omega, prior, row.data(), o_in.seed, (uint32_t)i);
    if (i + 1 < N)
      for (int j = 0; j < n_trees; ++j) { st = phm_engine_set_model(es.e[j], Qw.data()); if (st) return st; }
  }
  return PHM_OK;
}

// maketreelistMCMCmt src/phylomap.cpp:2267-2365 (R/sumstatMCMCmt.R) and maketreelistMCMCksmt :2722-2844 (R/sumstatMCMCksmt.R).
// One engine over the whole list: tree j's chain lives on replica tile j, so one launch per iteration sweeps every tree with
// the current Q (:2341-2345); the host then draws the tree whose row is kept (:2347-2350), updates Q from that row and
// uploads the new model once for all trees.
static int32_t run_qupdate_mt(int variant, const phm_tree* trees, int32_t n_trees, int32_t n, const double* Q, const double* pid,
                              double Omega, const int32_t* nen_m, const int32_t* nodelist_m, const int32_t* roots, int32_t N,
                              const double* prior, int32_t n_prior, const phm_options* opt_in, double* out) {
  if (!out || !prior || !Q || !trees) return fail(PHM_ERR_BAD_INPUT, "out/prior/Q/trees is NULL");
  if (N < 1 || n_trees < 1) return fail(PHM_ERR_BAD_INPUT, "N and n_trees must be >= 1");
  const bool ksmt = variant == PHM_MCMC_KSMT;
  if (n_prior < (ksmt ? 8 : 4)) return fail(PHM_ERR_BAD_INPUT, ksmt ? "sumstatMCMCksmt needs prior = c(a_l01, b_l01, a_l10, b_l10, a_k, b_k, a_g, b_g) (src/phylomap.cpp:2391-2663)" : "sumstatMCMCmt needs prior = c(a01, b01, a10, b10)");
  if (ksmt && (n < 4 || (n & 1))) return fail(PHM_ERR_BAD_INPUT, "sumstatMCMCksmt needs n = 2k+2 states with k >= 1 (src/phylomap.cpp:2729)");
  phm_options o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opt_in) o = *opt_in;
  if (o.n_replicas > 1) return fail(PHM_ERR_UNSUPPORTED, "the multi-tree drivers run one chain per tree");
  o.n_replicas = 1; o.reduce = 0; o.tips_per_replica = 0; o.iters_per_launch = 1;
  phm_model model;
  model.n_states = n; model.Q = Q; model.pid = pid; model.B = nullptr; model.Omega = Omega; model.variant = variant;
  // An engine per tree, or one engine over the list.  The one engine walks a whole tree in ONE lane and rebuilds its chain tables
  // with every new model: 18 ms per iteration for 64 trees of 100 tips, 71 ms at 600 tips, seconds on a tree of thousands of tips,
  // whatever the number of trees.  A tree's own engine in the branch mapping costs ~0.05 ms per tree and iteration, launches, wait
  // and model upload included (64 trees: 3.1 / 3.0 ms at 100 / 600 tips; profiles/r04_probe_multi_tree.log).  Same streams: tree j's
  // chain is replica 64 j of the list either way.  The list engine is kept for long lists of small trees and for
  // PHM_MAP_REPLICAS (its dwell sums are added in the reference's order: bit-identical to the oracle, not just to 1e-10).
  double segs = 0.0;
  for (int b = 0; b < trees[0].n_edge; ++b) {
    double tb = 0.0;
    for (int i = trees[0].map_off[b]; i < trees[0].map_off[b + 1]; ++i) tb += trees[0].maps[i];
    segs += std::max(1.0 + Omega * tb, (double)(trees[0].map_off[b + 1] - trees[0].map_off[b]));
  }
  const bool per_tree = o.mapping == PHM_MAP_BRANCHES ||
                        (o.mapping == PHM_MAP_AUTO && o.storage == 0 && (n_trees <= 256 || 1.5e-3 * segs > 0.05 * n_trees));
  if (per_tree) return run_qupdate_mt_per_tree(variant, trees, n_trees, n, model, o, nen_m, nodelist_m, roots, N, prior, out);
  phm_engine* e = nullptr;
  int32_t st = phm_engine_create_multi(trees, n_trees, &model, &o, N, &e);
  if (st) return st;
  std::unique_ptr<phm_engine, void (*)(phm_engine*)> guard(e, phm_engine_destroy);
  const int Nn = e->sched.n_node;
  if (nen_m || nodelist_m || roots) {      // R's matrices are column-major: row j = elements j, j + n_trees, ...
    std::vector<int32_t> nen(2 * (size_t)Nn), nodelist(Nn > 1 ? Nn - 1 : 0);
    std::string serr;
    for (int j = 0; j < n_trees; ++j) {
      if (nen_m) for (int i = 0; i < 2 * Nn; ++i) nen[i] = nen_m[j + (size_t)i * n_trees];
      if (nodelist_m) for (int i = 0; i < Nn - 1; ++i) nodelist[i] = nodelist_m[j + (size_t)i * n_trees];
      if (!phm::check_reference_orders(e->scheds[j], trees[j].edge, nen_m ? nen.data() : nullptr, nodelist_m ? nodelist.data() : nullptr,
                                       roots ? roots[j] : e->scheds[j].root + e->sched.n_tips + 1, serr))
        return fail(PHM_ERR_BAD_INPUT, "tree " + std::to_string(j) + ": " + serr);
    }
  }
  const int ecols = e->cols;                 // n + n*n + 2 + 3k + 1: the engine's root-state column becomes tree_number
  const size_t nn = (size_t)n * n;
  std::vector<double> Qw(Q, Q + nn), rows((size_t)n_trees * ecols);
  for (int i = 0; i < N; ++i) {
    st = phm_engine_run(e, 1, nullptr);
    if (!st) st = phm_engine_sync(e);
    if (!st) st = phm_engine_read_stats(e, i, 1, rows.data());       // one 1 x ecols row per tree
    if (st) return st;
    const uint32_t pick = phm::pick_tree(n_trees, o.seed, (uint32_t)i);
    if (pick >= (uint32_t)n_trees) return fail(PHM_ERR_ZERO_PROB, "sampleOnce ran past the last tree (src/phylomap.cpp:85-89)");
    const double* row = rows.data() + (size_t)pick * ecols;
    for (int c = 0; c + 1 < ecols; ++c) out[(size_t)c * N + i] = row[c];
    out[(size_t)(ecols - 1) * N + i] = (double)pick;                   // :2350, 0-based as the reference stores it
    if (ksmt) phm::ks_updates(Qw.data(), n, Omega, prior, row, o.seed, (uint32_t)i, true);
    else phm::mt_updates(Qw.data(), Omega, prior, row, o.seed, (uint32_t)i);
    if (i + 1 < N) { st = phm_engine_set_model(e, Qw.data()); if (st) return st; }
  }
  return PHM_OK;
}

extern "C" int32_t phm_maketreelistMCMCmt(const phm_tree* trees, int32_t n_trees, int32_t n, const double* Q, const double* pid,
                                          const double* B, double Omega, const int32_t* nen_m, const int32_t* nodelist_m,
                                          const int32_t* roots, int32_t N, const double* prior, int32_t n_prior,
                                          const phm_options* opt, double* out) {
  (void)B;
  return run_qupdate_mt(PHM_MCMC_MT, trees, n_trees, n, Q, pid, Omega, nen_m, nodelist_m, roots, N, prior, n_prior, opt, out);
}

extern "C" int32_t phm_maketreelistMCMCksmt(const phm_tree* trees, int32_t n_trees, int32_t n, const double* Q, const double* pid,
                                            const double* B, double Omega, const int32_t* nen_m, const int32_t* nodelist_m,
                                            const int32_t* roots, int32_t N, const double* prior, int32_t n_prior,
                                            const phm_options* opt, double* out) {
  (void)B;
  return run_qupdate_mt(PHM_MCMC_KSMT, trees, n_trees, n, Q, pid, Omega, nen_m, nodelist_m, roots, N, prior, n_prior, opt, out);
}
