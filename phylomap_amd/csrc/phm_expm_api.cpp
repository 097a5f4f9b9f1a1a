// phm_expm_api.cpp -- C-ABI of the matrix-exponentiation path: batched transition matrices (eigen route, Pade route, their
// MFMA f64 variants) and the sumstatEXP driver (maketreelistEXP, src/phylomap.cpp:3001-3051).
#include "phm_internal.h"

#include <thread>

namespace {

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  ~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

}  // namespace

constexpr int EXP_TILES_AUTO_MAX_TILES = 1 << 30;      // see the mapping note in phm_maketreelistEXP

extern "C" {

static int32_t expm_eigen_impl(bool mfma, int32_t n, const double* lefts, const double* rights, const double* d, const double* t,
                               int32_t n_t, int32_t device, double* out, double* kernel_ms) {
  if (n < 1 || n > 256 || !lefts || !rights || !d || !t || !out || n_t < 0) return fail(PHM_ERR_BAD_INPUT, "phm_expm_eigen: bad arguments");
  int32_t st = select_device(device);
  if (st) return st;
  if (n_t == 0) return PHM_OK;
  std::vector<double> L, R, dv(n);
  cm_to_rm(lefts, n, L); cm_to_rm(rights, n, R);
  for (int i = 0; i < n; ++i) dv[i] = d[i + (size_t)i * n];
  const size_t nn = (size_t)n * n;
  DevBuf dL, dR, dd, dt, dout;
  HIPCHK(dL.alloc(sizeof(double) * nn)); HIPCHK(dR.alloc(sizeof(double) * nn)); HIPCHK(dd.alloc(sizeof(double) * n));
  HIPCHK(dt.alloc(sizeof(double) * n_t)); HIPCHK(dout.alloc(sizeof(double) * nn * n_t));
  HIPCHK(hipMemcpy(dL.p, L.data(), dL.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dR.p, R.data(), dR.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dd.p, dv.data(), dd.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, t, dt.bytes, hipMemcpyHostToDevice));
  Timer tm;
  HIPCHK(hipEventCreate(&tm.a)); HIPCHK(hipEventCreate(&tm.b));
  HIPCHK(hipEventRecord(tm.a, nullptr));
  if (mfma) HIPCHK(phm::launch_expm_eigen_mfma(n, dL.as<double>(), dR.as<double>(), dd.as<double>(), dt.as<double>(), n_t, dout.as<double>(), nullptr));
  else HIPCHK(phm::launch_expm_eigen(n, dL.as<double>(), dR.as<double>(), dd.as<double>(), dt.as<double>(), n_t, dout.as<double>(), nullptr));
  HIPCHK(hipEventRecord(tm.b, nullptr));
  HIPCHK(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
  if (kernel_ms) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, tm.a, tm.b)); *kernel_ms = ms; }
  return PHM_OK;
}

int32_t phm_expm_eigen(int32_t n, const double* lefts, const double* rights, const double* d, const double* t,
                       int32_t n_t, int32_t device, double* out, double* kernel_ms) {
  return expm_eigen_impl(false, n, lefts, rights, d, t, n_t, device, out, kernel_ms);
}

int32_t phm_expm_eigen_mfma(int32_t n, const double* lefts, const double* rights, const double* d, const double* t,
                            int32_t n_t, int32_t device, double* out, double* kernel_ms) {
  if (n <= 16 || n > 64) return fail(PHM_ERR_UNSUPPORTED, "phm_expm_eigen_mfma: 16 < n_states <= 64 (smaller matrices do not fill an MFMA tile)");
  return expm_eigen_impl(true, n, lefts, rights, d, t, n_t, device, out, kernel_ms);
}

static int32_t expm_pade_impl(bool mfma, int32_t n, const double* Q, const double* t, int32_t n_t, int32_t device, double* out,
                              double* kernel_ms) {
  if (n < 1 || n > 128 || !Q || !t || !out || n_t < 0) return fail(PHM_ERR_BAD_INPUT, "phm_expm_pade: bad arguments (n <= 128)");
  int32_t st = select_device(device);
  if (st) return st;
  if (n_t == 0) return PHM_OK;
  std::vector<double> Qr;
  cm_to_rm(Q, n, Qr);
  std::vector<int32_t> sq(n_t);
  for (int b = 0; b < n_t; ++b) sq[b] = pade_squarings(Qr.data(), n, t[b]);
  const size_t nn = (size_t)n * n;
  DevBuf dQ, dt, ds, dwork, dout, derr, dbad;
  HIPCHK(dQ.alloc(sizeof(double) * nn)); HIPCHK(dt.alloc(sizeof(double) * n_t)); HIPCHK(ds.alloc(sizeof(int32_t) * n_t));
  HIPCHK(dwork.alloc(mfma ? 16 : sizeof(double) * nn * 5 * n_t)); HIPCHK(dout.alloc(sizeof(double) * nn * n_t)); HIPCHK(derr.alloc(sizeof(uint32_t)));
  HIPCHK(hipMemcpy(dQ.p, Qr.data(), dQ.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, t, dt.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, sq.data(), ds.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(derr.p, 0, sizeof(uint32_t)));
  if (mfma) { HIPCHK(dbad.alloc(sizeof(int32_t) * n_t)); HIPCHK(hipMemset(dbad.p, 0, dbad.bytes)); }
  // smallest pivot the unpivoted block elimination accepts (test aid: phm_debug_options.pade_pivot_min = 1e300 sends every matrix to the pivoted kernel)
  const double piv_min = g_phm_debug.pade_pivot_min > 0.0 ? g_phm_debug.pade_pivot_min : 1e-3;
  Timer tm;
  HIPCHK(hipEventCreate(&tm.a)); HIPCHK(hipEventCreate(&tm.b));
  HIPCHK(hipEventRecord(tm.a, nullptr));
  if (mfma) HIPCHK(phm::launch_expm_pade_mfma(n, dQ.as<double>(), dt.as<double>(), ds.as<int32_t>(), n_t, dout.as<double>(), dbad.as<int32_t>(), piv_min, nullptr));
  else HIPCHK(phm::launch_expm_pade(n, dQ.as<double>(), dt.as<double>(), ds.as<int32_t>(), n_t, dwork.as<double>(), dout.as<double>(), derr.as<uint32_t>(), nullptr));
  HIPCHK(hipEventRecord(tm.b, nullptr));
  if (mfma) {      // matrices the matrix-core kernel gave up on (a small pivot in a diagonal block of D): the pivoted kernel
    std::vector<int32_t> badh(n_t);
    HIPCHK(hipMemcpy(badh.data(), dbad.p, dbad.bytes, hipMemcpyDeviceToHost));
    std::vector<int32_t> idx;
    for (int b = 0; b < n_t; ++b) if (badh[b]) idx.push_back(b);
    if (!idx.empty()) {
      const int nb = (int)idx.size();
      std::vector<double> tb(nb); std::vector<int32_t> sb(nb);
      for (int i = 0; i < nb; ++i) { tb[i] = t[idx[i]]; sb[i] = sq[idx[i]]; }
      DevBuf dt2, ds2, dwork2, dout2;
      HIPCHK(dt2.alloc(sizeof(double) * nb)); HIPCHK(ds2.alloc(sizeof(int32_t) * nb));
      HIPCHK(dwork2.alloc(sizeof(double) * nn * 5 * nb)); HIPCHK(dout2.alloc(sizeof(double) * nn * nb));
      HIPCHK(hipMemcpy(dt2.p, tb.data(), dt2.bytes, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(ds2.p, sb.data(), ds2.bytes, hipMemcpyHostToDevice));
      HIPCHK(phm::launch_expm_pade(n, dQ.as<double>(), dt2.as<double>(), ds2.as<int32_t>(), nb, dwork2.as<double>(), dout2.as<double>(), derr.as<uint32_t>(), nullptr));
      for (int i = 0; i < nb; ++i)
        HIPCHK(hipMemcpy(dout.as<double>() + nn * idx[i], dout2.as<double>() + nn * i, sizeof(double) * nn, hipMemcpyDeviceToDevice));
      HIPCHK(hipEventRecord(tm.b, nullptr));
    }
  }
  HIPCHK(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
  uint32_t derrh = 0;
  HIPCHK(hipMemcpy(&derrh, derr.p, sizeof derrh, hipMemcpyDeviceToHost));
  if (kernel_ms) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, tm.a, tm.b)); *kernel_ms = ms; }
  if (derrh) return fail(PHM_ERR_BAD_INPUT, "phm_expm_pade: singular Pade denominator");
  return PHM_OK;
}

int32_t phm_expm_pade(int32_t n, const double* Q, const double* t, int32_t n_t, int32_t device, double* out, double* kernel_ms) {
  return expm_pade_impl(false, n, Q, t, n_t, device, out, kernel_ms);
}

int32_t phm_expm_pade_mfma(int32_t n, const double* Q, const double* t, int32_t n_t, int32_t device, double* out,
                           double* kernel_ms) {
  if (n <= 16 || n > 64) return fail(PHM_ERR_UNSUPPORTED, "phm_expm_pade_mfma: 16 < n_states <= 64 (smaller matrices do not fill an MFMA tile)");
  return expm_pade_impl(true, n, Q, t, n_t, device, out, kernel_ms);
}

// maketreelistEXP, src/phylomap.cpp:3001-3051.  P(t_b) and the pruning pass are computed ONCE (the reference
// recomputes both every iteration although Q never changes, :2980-2981).
// samples [it0, it0 + N) of the call on ONE device (o.device); out: N x cols column-major
static int32_t exp_oneshot(const phm_tree* x, int32_t n, const double* Q, const double* pid, const int32_t* nen,
                           const int32_t* nodelist, int32_t root, int32_t N, int32_t it0, const double* lefts, const double* rights,
                           const double* d, const phm_options* opt_in, double* out) {
  if (!x || !Q || !pid || !lefts || !rights || !d || !out) return fail(PHM_ERR_BAD_INPUT, "phm_maketreelistEXP: NULL argument");
  if (N < 1) return fail(PHM_ERR_BAD_INPUT, "N must be >= 1");
  if (n < 2) return fail(PHM_ERR_BAD_INPUT, "n_states must be >= 2");
  if (n > 64) return fail(PHM_ERR_UNSUPPORTED, "this build has EXP kernels for n_states <= 64 only");
  if (!x->edge_length) return fail(PHM_ERR_BAD_INPUT, "x$edge.length is required (src/phylomap.cpp:3034)");
  phm_options o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opt_in) o = *opt_in;
  int32_t st = validate_tree_paths(x, n, 1);
  if (st) return st;
  phm::Schedule s;
  std::string serr;
  if (!phm::build_schedule(x->n_tips, x->n_node, x->n_edge, x->edge, s, serr)) return fail(PHM_ERR_BAD_INPUT, "tree: " + serr);
  if (!phm::check_reference_orders(s, x->edge, nen, nodelist, root, serr)) return fail(PHM_ERR_BAD_INPUT, serr);
  const int E = s.n_edge, T = s.n_tips;
  for (int b = 0; b < E; ++b)
    if (!std::isfinite(x->edge_length[b]) || x->edge_length[b] < 0.0) return fail(PHM_ERR_BAD_INPUT, "edge.length must be finite and non-negative");

  std::vector<double> L, R, dv(n), B2((size_t)n * n);
  cm_to_rm(lefts, n, L); cm_to_rm(rights, n, R);
  double minq = Q[0];
  for (int i = 0; i < n; ++i) { dv[i] = d[i + (size_t)i * n]; minq = std::min(minq, Q[i + (size_t)i * n]); }
  const double rate = -1.0 * minq;                                             // :3008
  if (!(rate > 0.0)) return fail(PHM_ERR_BAD_INPUT, "Q must have a negative diagonal entry");
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double b = ((i == j) ? 1.0 : 0.0) + Q[i + (size_t)j * n] / rate;          // :3011
      if (!(b >= 0.0)) return fail(PHM_ERR_BAD_INPUT, "I + Q/poissonRate must be non-negative");
      B2[(size_t)i * n + j] = b;
    }
  std::vector<double> col, rowtab;
  build_chain_tables(B2.data(), n, phm::UNIF_CAP + 1, col, rowtab, false, false);      // newunifSample :127: unfused sums for every n; no row table

  st = select_device(o.device);
  if (st) return st;
  const size_t nn = (size_t)n * n;
  const int tiles = (N + 63) / 64;
  const int cols = n + n * (n - 1);
  DevBuf dL, dR, dd, dt, dP, dPL, dup, ddown, dcol, dB2, dtips, dnst, dtimes, dout, derr;
  HIPCHK(dL.alloc(sizeof(double) * nn)); HIPCHK(dR.alloc(sizeof(double) * nn)); HIPCHK(dd.alloc(sizeof(double) * n));
  HIPCHK(dt.alloc(sizeof(double) * E)); HIPCHK(dP.alloc(sizeof(double) * nn * E));
  HIPCHK(dPL.alloc(sizeof(double) * (size_t)(2 * T - 1) * n));
  HIPCHK(dup.alloc(sizeof(phm::UpStep) * s.up.size())); HIPCHK(ddown.alloc(sizeof(phm::DownStep) * s.down.size()));
  HIPCHK(dcol.alloc(sizeof(double) * col.size())); HIPCHK(dB2.alloc(sizeof(double) * nn)); HIPCHK(dtips.alloc(T));
  HIPCHK(dnst.alloc((size_t)tiles * s.n_node * 64)); HIPCHK(dtimes.alloc(sizeof(double) * (size_t)tiles * phm::UNIF_CAP * 64));
  HIPCHK(dout.alloc(sizeof(double) * (size_t)N * cols)); HIPCHK(derr.alloc(sizeof(uint32_t)));
  std::vector<double> PLh((size_t)(2 * T - 1) * n, 0.0);
  std::vector<uint8_t> tips(T);
  for (int t = 0; t < T; ++t) { tips[t] = (uint8_t)(x->states[t] - 1); PLh[(size_t)t * n + tips[t]] = 1.0; }   // :2883
  HIPCHK(hipMemcpy(dL.p, L.data(), dL.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dR.p, R.data(), dR.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dd.p, dv.data(), dd.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, x->edge_length, dt.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dPL.p, PLh.data(), dPL.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dup.p, s.up.data(), dup.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ddown.p, s.down.data(), ddown.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dcol.p, col.data(), dcol.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dB2.p, B2.data(), dB2.bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dtips.p, tips.data(), T, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(derr.p, 0, sizeof(uint32_t)));
  HIPCHK(hipMemset(dnst.p, 0, dnst.bytes));

  HIPCHK(phm::launch_expm_eigen(n, dL.as<double>(), dR.as<double>(), dd.as<double>(), dt.as<double>(), E, dP.as<double>(), nullptr));   // :3042
  {   // :3043 -- the pruning pass, level by level (heights: children strictly below their parent)
    std::vector<int32_t> height(s.n_node, 0), uorder, ulevel;
    std::vector<std::vector<int32_t>> by_h;
    for (int k = 0; k < s.n_node; ++k) {
      const phm::UpStep& u = s.up[k];
      int h = 0;
      for (int c = 0; c < 2; ++c) if (u.child[c] >= 0) h = std::max(h, height[u.child[c]] + 1);
      height[u.parent] = h;
      if ((int)by_h.size() <= h) by_h.resize(h + 1);
      by_h[h].push_back(k);
    }
    ulevel.push_back(0);
    for (auto& v : by_h) { uorder.insert(uorder.end(), v.begin(), v.end()); ulevel.push_back((int32_t)uorder.size()); }
    DevBuf duord;
    HIPCHK(duord.alloc(sizeof(int32_t) * uorder.size()));
    HIPCHK(hipMemcpy(duord.p, uorder.data(), duord.bytes, hipMemcpyHostToDevice));
    HIPCHK(phm::launch_exp_pl_levels(n, T, dup.as<phm::UpStep>(), duord.as<int32_t>(), ulevel, dP.as<double>(), dPL.as<double>(),
                                     o.rescale_pruning != 0, nullptr));
    HIPCHK(hipDeviceSynchronize());      // duord goes out of scope
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;      // time of the sampling kernel alone (phm_last_kernel_ms)
  HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
  HIPCHK(hipEventRecord(ev0, nullptr));

  // Mapping (phm_options.mapping): 1 = one wave per tile of 64 samples walks the tree (exp_sample_kernel / exp_wide_kernel);
  // 3 = one wave per (tile, branch) (exp_tiles_*); 0 = automatic: the (tile, branch) mapping unless there are so many samples
  // that the tiles alone fill the chip.
  const bool use_tiles = o.mapping == PHM_MAP_TILES || (o.mapping == PHM_MAP_AUTO && tiles < EXP_TILES_AUTO_MAX_TILES);
  if (use_tiles) {
    // edges with an internal child, grouped by depth (parents' states are drawn a level earlier)
    std::vector<int32_t> depth(s.n_node, 0), order, level_off;
    {
      std::vector<std::vector<int32_t>> by_depth;
      for (int k = 0; k < E; ++k) {
        const phm::DownStep& d = s.down[k];
        if (d.child < 0) continue;
        const int dl = depth[d.parent];
        depth[d.child] = dl + 1;
        if ((int)by_depth.size() <= dl) by_depth.resize(dl + 1);
        by_depth[dl].push_back(k);
      }
      level_off.push_back(0);
      for (auto& v : by_depth) { order.insert(order.end(), v.begin(), v.end()); level_off.push_back((int32_t)order.size()); }
    }
    double tree_len = 0.0;
    for (int b = 0; b < E; ++b) tree_len += x->edge_length[b];
    int ex = 0;
    (void)std::frexp(std::max(tree_len, 1.0), &ex);
    const int64_t items = (int64_t)E * tiles;
    const int branch_blocks = (int)std::min<int64_t>((items + 3) / 4, 2048);
    const size_t npad = (size_t)tiles * 64;
    DevBuf dorder, dpid, ddw, dcnt, dtm;
    HIPCHK(dorder.alloc(sizeof(int32_t) * std::max<size_t>(order.size(), 1))); HIPCHK(dpid.alloc(sizeof(double) * n));
    HIPCHK(ddw.alloc(sizeof(unsigned long long) * n * npad)); HIPCHK(dcnt.alloc(sizeof(uint32_t) * (size_t)n * (n - 1) * npad));
    HIPCHK(dtm.alloc(sizeof(double) * (size_t)branch_blocks * 4 * phm::UNIF_CAP * 64));
    if (!order.empty()) HIPCHK(hipMemcpy(dorder.p, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dpid.p, pid, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ddw.p, 0, ddw.bytes)); HIPCHK(hipMemset(dcnt.p, 0, dcnt.bytes));
    phm::ExpTilesParams p;
    p.n_states = n; p.n_tips = T; p.n_node = s.n_node; p.n_edge = E; p.root = s.root; p.N = N; p.n_tiles = tiles; p.it0 = it0;
    p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32); p.replica = (uint32_t)o.replica_offset;
    p.poisson_rate = rate; p.fx_scale = std::ldexp(1.0, 61 - ex); p.fx_inv = std::ldexp(1.0, ex - 61);
    p.pid = dpid.as<double>(); p.down = ddown.as<phm::DownStep>(); p.node_order = dorder.as<int32_t>();
    p.P = dP.as<double>(); p.PL = dPL.as<double>(); p.edge_length = dt.as<double>(); p.colpow = dcol.as<double>(); p.B2 = dB2.as<double>();
    p.tips = dtips.as<uint8_t>(); p.nstate = dnst.as<uint8_t>(); p.times = dtm.as<double>();
    p.dwfx = ddw.as<unsigned long long>(); p.cnt = dcnt.as<uint32_t>(); p.out = dout.as<double>(); p.err = derr.as<uint32_t>();
    HIPCHK(hipEventRecord(ev0, nullptr));      // (re-recorded: the set-up above is not part of the sampler's time)
    HIPCHK(phm::launch_exp_tiles(p, level_off, branch_blocks, nullptr));
    HIPCHK(hipEventRecord(ev1, nullptr));
    HIPCHK(hipEventSynchronize(ev1));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) g_phm_last_kernel_ms = ms; }
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    HIPCHK(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
    uint32_t derrh = 0;
    HIPCHK(hipMemcpy(&derrh, derr.p, sizeof derrh, hipMemcpyDeviceToHost));
    return device_status(derrh);
  }
  auto fill = [&](auto& p) {
    p.n_tips = T; p.n_node = s.n_node; p.n_edge = E; p.root = s.root; p.N = N; p.n_tiles = tiles; p.it0 = it0;
    p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32); p.replica = (uint32_t)o.replica_offset;
    p.poisson_rate = rate;
    for (int i = 0; i < n; ++i) p.pid[i] = pid[i];
    p.down = ddown.as<phm::DownStep>(); p.P = dP.as<double>(); p.PL = dPL.as<double>(); p.edge_length = dt.as<double>();
    p.colpow = dcol.as<double>(); p.B2 = dB2.as<double>(); p.tips = dtips.as<uint8_t>(); p.nstate = dnst.as<uint8_t>();
    p.times = dtimes.as<double>(); p.out = dout.as<double>(); p.err = derr.as<uint32_t>();
  };
  hipError_t le = hipSuccess;
  if (n == 2) { phm::ExpParams<2> p; fill(p); le = phm::launch_exp_sample<2>(p, nullptr); }
  if (n == 3) { phm::ExpParams<3> p; fill(p); le = phm::launch_exp_sample<3>(p, nullptr); }
  if (n == 4) { phm::ExpParams<4> p; fill(p); le = phm::launch_exp_sample<4>(p, nullptr); }
  if (n > 4) {
    DevBuf dpid;
    HIPCHK(dpid.alloc(sizeof(double) * n));
    HIPCHK(hipMemcpy(dpid.p, pid, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout.p, 0, dout.bytes));
    phm::ExpWideParams p;
    p.n_states = n; p.n_tips = T; p.n_node = s.n_node; p.n_edge = E; p.root = s.root; p.N = N; p.n_tiles = tiles; p.it0 = it0;
    p.seed_lo = (uint32_t)(o.seed & 0xFFFFFFFFull); p.seed_hi = (uint32_t)(o.seed >> 32); p.replica = (uint32_t)o.replica_offset;
    p.poisson_rate = rate; p.pid = dpid.as<double>();
    p.down = ddown.as<phm::DownStep>(); p.P = dP.as<double>(); p.PL = dPL.as<double>(); p.edge_length = dt.as<double>();
    p.colpow = dcol.as<double>(); p.B2 = dB2.as<double>(); p.tips = dtips.as<uint8_t>(); p.nstate = dnst.as<uint8_t>();
    p.times = dtimes.as<double>(); p.out = dout.as<double>(); p.err = derr.as<uint32_t>();
    le = phm::launch_exp_wide(p, nullptr);
    HIPCHK(le);
    HIPCHK(hipDeviceSynchronize());
  }
  HIPCHK(le);
  HIPCHK(hipEventRecord(ev1, nullptr));
  HIPCHK(hipEventSynchronize(ev1));
  { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) g_phm_last_kernel_ms = ms; }
  (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  HIPCHK(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
  uint32_t derrh = 0;
  HIPCHK(hipMemcpy(&derrh, derr.p, sizeof derrh, hipMemcpyDeviceToHost));
  return device_status(derrh);
}

// The samples are i.i.d. and addressed by their index (src/phylomap.cpp:3045-3048): with phm_options.n_devices > 1 device d draws
// a contiguous range of the N samples (one host thread per device) -- the matrix is the one-device matrix row for row.
int32_t phm_maketreelistEXP(const phm_tree* x, int32_t n, const double* Q, const double* pid, const int32_t* nen,
                            const int32_t* nodelist, int32_t root, int32_t N, const double* lefts, const double* rights,
                            const double* d, const phm_options* opt_in, double* out) {
  if (!out) return fail(PHM_ERR_BAD_INPUT, "phm_maketreelistEXP: NULL argument");
  if (N < 1) return fail(PHM_ERR_BAD_INPUT, "N must be >= 1");
  phm_options o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opt_in) o = *opt_in;
  std::vector<phm_shard> shards;
  int32_t st = phm_plan_shards(o, N, shards);
  if (st) return st;
  if (shards.size() == 1) {
    o.device = shards[0].device; o.n_devices = 0;
    return exp_oneshot(x, n, Q, pid, nen, nodelist, root, N, 0, lefts, rights, d, &o, out);
  }
  const int cols = n + n * (n - 1);
  struct Run { int32_t st = PHM_OK; std::string err; double ms = 0.0; std::vector<double> buf; };
  std::vector<Run> runs(shards.size());
  auto work = [&](size_t i) {
    phm_options oi = o;
    oi.device = shards[i].device; oi.n_devices = 0;
    const int32_t Ni = (int32_t)shards[i].count;
    runs[i].buf.assign((size_t)Ni * cols, 0.0);
    runs[i].st = exp_oneshot(x, n, Q, pid, nen, nodelist, root, Ni, (int32_t)shards[i].first, lefts, rights, d, &oi, runs[i].buf.data());
    if (runs[i].st) runs[i].err = g_phm_err;
    runs[i].ms = g_phm_last_kernel_ms;
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < shards.size(); ++i) th.emplace_back(work, i);
  work(0);
  for (std::thread& t : th) t.join();
  double ms = 0.0;
  for (size_t i = 0; i < shards.size(); ++i) {
    if (runs[i].st) return fail(runs[i].st, "device " + std::to_string(shards[i].device) + ": " + runs[i].err);
    ms = std::max(ms, runs[i].ms);
    const size_t Ni = (size_t)shards[i].count, r0 = (size_t)shards[i].first;
    for (int c = 0; c < cols; ++c) std::memcpy(out + (size_t)c * N + r0, runs[i].buf.data() + (size_t)c * Ni, sizeof(double) * Ni);
  }
  g_phm_last_kernel_ms = ms;      // the devices sample side by side: the longest
  return PHM_OK;
}

double phm_last_kernel_ms(void) { return g_phm_last_kernel_ms; }

}  // extern "C"
