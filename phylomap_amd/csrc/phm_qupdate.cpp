// phm_qupdate.cpp -- host glue of the Q-updating drivers: the Gibbs / Metropolis-Hastings updates the reference runs
// on the rate matrix after every tree sweep (maketreelistMCMCbf src/phylomap.cpp:1299-1300 with updatel01/updatel10
// :1189-1253; maketreelistMCMCks :1862-1866 with updateksl01/l10 :1435-1578, updaterkappas :1582-1644,
// updatelkappas :1648-1710, updategammas :1714-1785; the multi-tree twins maketreelistMCMCmt :2351-2352 with updatel01mtNS /
// updatel10mtNS :2192-2262 and maketreelistMCMCksmt :2828-2832 with update*mt :2371-2705).  O(k) scalar work per iteration on statistics the device has
// already reduced; the sweeps themselves stay on the GPU (phm_drivers.cpp drives both).
//
// Random numbers: the reference draws Rf_rgamma and runif from R's global stream.  Here every update owns a Philox
// stream (replica word 0xFFFFFFFF, entity 0xFFFFFF00 | update id, iteration word = sweep index); the gamma variate is
// Marsaglia & Tsang's (2000) squeeze with a Box-Muller normal (R's Ahrens-Dieter code is third-party and not
// restated).  Same arithmetic as the oracle's twin, so the two agree bit for bit on a given host.
#include "phm_qupdate.h"

#include <cmath>

namespace phm {

namespace {

void philox_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
  for (int r = 0; r < 7; ++r) {      // Philox4x32-7, as every stream of the engine (phm_device.h)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

struct UpdateStream {
  uint64_t seed;
  uint32_t iter, id, next = 0;
  double uniform() {
    uint32_t o[4];
    const uint32_t d = next++;
    philox_host(d >> 2, 0xFFFFFF00u | id, iter, 0xFFFFFFFFu, (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32), o);
    return ((double)o[d & 3u] + 0.5) * 2.3283064365386962890625e-10;      // draw d = word d & 3 of block d >> 2, (x + 0.5) 2^-32
  }
  double gamma(double shape, double scale) {
    double boost = 1.0;
    if (shape < 1.0) { const double u = uniform(); boost = std::pow(u, 1.0 / shape); shape += 1.0; }
    const double d = shape - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (;;) {
      const double u1 = uniform(), u2 = uniform();
      const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
      double v = 1.0 + c * z;
      if (v <= 0.0) continue;
      v = v * v * v;
      const double u = uniform();
      if (std::log(u) < 0.5 * z * z + d - d * v + d * std::log(v)) return d * v * boost * scale;
    }
  }
};

// View of one statistics row (device column order: n dwell sums, n*n counts row-major) and of the column-major Q
struct Ctx {
  double* Q; int n; double Omega; const double* prior; const double* row;
  double& q(int i, int j) const { return Q[i + (size_t)j * n]; }
  double time(int s) const { return row[s]; }
  double count(int from, int to) const { return row[n + from * n + to]; }
};

// hidden-rates parameters as the reference re-derives them at the top of every update (:1441-1450)
struct HiddenRates {
  int k;
  double lam[2], rk[32], lk[32], gm[33];
  explicit HiddenRates(const Ctx& c) : k(c.n / 2 - 1) {
    lam[0] = c.q(0, 1); lam[1] = c.q(1, 0);
    for (int i = 0; i < k; ++i) rk[i] = c.q(2 * i, 2 * i + 2);
    for (int i = 0; i < k; ++i) lk[i] = c.q(2 * i + 2, 2 * i);
    gm[0] = 1;
    for (int i = 1; i <= k; ++i) gm[i] = c.q(2 * i, 2 * i + 1) / lam[0];
  }
};

}  // namespace

void bf_updates(double* Q, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter) {
  const Ctx c{Q, 2, Omega, prior, row};
  {   // updatel01 :1189-1219: Gibbs draw; the acceptance ratio is computed and never tested in the reference
    UpdateStream rs{seed, iter, 0};
    const int n01 = (int)c.count(0, 1);
    const double fresh = rs.gamma(prior[0] + n01, 1 / (prior[1] + c.time(0)));
    if (!(fresh > Omega)) { (void)rs.uniform(); c.q(0, 0) = -fresh; c.q(0, 1) = fresh; }
  }
  {   // updatel10 :1221-1253
    UpdateStream rs{seed, iter, 1};
    const int n10 = (int)c.count(1, 0);
    const double fresh = rs.gamma(prior[2] + n10, 1 / (prior[3] + c.time(1)));
    if (!(fresh > Omega)) { (void)rs.uniform(); c.q(1, 0) = fresh; c.q(1, 1) = -fresh; }
  }
}

void mt_updates(double* Q, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter) {
  const Ctx c{Q, 2, Omega, prior, row};
  // updatel01mtNS :2192-2226 (side 0), updatel10mtNS :2228-2262 (side 1): unlike the single-tree twins the acceptance
  // ratio IS tested; the runif is drawn only for proposals not above Omega; `acceptcompare` is dead code there
  for (int side = 0; side < 2; ++side) {
    UpdateStream rs{seed, iter, (uint32_t)side};
    const int other = 1 - side;
    const int stay = (int)c.count(side, side), leave = (int)c.count(side, other);
    const double cur = c.q(side, other);
    const double fresh = rs.gamma(prior[2 * side] + leave, 1 / (prior[2 * side + 1] + c.time(side)));
    if (fresh > Omega) continue;
    double accept = std::pow((Omega - fresh) / (Omega - cur), stay) * std::exp(c.time(side) * (fresh - cur));
    if (accept > 1) accept = 1;
    const double cmp = rs.uniform();
    if (accept < cmp) continue;
    c.q(side, side) = -fresh; c.q(side, other) = fresh;
  }
}

uint32_t pick_tree(int n_trees, uint64_t seed, uint32_t iter) {
  UpdateStream rs{seed, iter, 0xFDu};
  const double u = rs.uniform();
  double total = 1.0;                                   // sampleOnce :81-90 over unit weights (:2332-2333, :2348)
  for (int j = 1; j < n_trees; ++j) total += 1.0;
  double cum = 0.0;
  for (int i = 0; i < n_trees; ++i) { cum += 1.0 / total; if (u < cum) return (uint32_t)i; }
  return (uint32_t)n_trees;                             // ran off the end: the reference would index out of range
}

void ks_updates(double* Q, int n, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter, bool mt) {
  const Ctx c{Q, n, Omega, prior, row};
  // the mt twins read their hyper-parameters two places further on (l10: prior(2), prior(3); kappas: 4, 5; gammas: 6, 7),
  // keep `betaprime - prior(1)` for side 1 (:2471) and drop the `< 1e-300` guards
  const double* prior_k = prior + (mt ? 2 : 0);
  const double tiny = mt ? -1.0 : 1e-300;
  const int k = n / 2 - 1;

  // ---- the two base rates: updateksl01 :1435-1505 (side 0), updateksl10 :1509-1578 (side 1) ----
  for (int side = 0; side < 2; ++side) {
    UpdateStream rs{seed, iter, (uint32_t)side};
    const HiddenRates h(c);
    const int other = 1 - side;
    double shape = prior[(mt && side) ? 2 : 0];
    for (int i = 0; i <= k; ++i) shape = shape + c.count(2 * i + side, 2 * i + other);
    double rate = prior[(mt && side) ? 3 : 1];
    for (int i = 0; i <= k; ++i) rate = rate + h.gm[i] * c.time(2 * i + side);
    const double fresh = rs.gamma(shape, 1 / rate);
    const double cur = h.lam[side];
    const double exposure = rate - prior[1];
    double logacc = (fresh - cur) * exposure;
    logacc = logacc + c.count(side, side) * std::log((Omega - h.rk[0] - h.gm[0] * fresh) / (Omega - h.rk[0] - h.gm[0] * cur));
    for (int i = 1; i < k; ++i)
      logacc = logacc + c.count(2 * i + side, 2 * i + side) *
               std::log((Omega - h.rk[i] - h.lk[i - 1] - h.gm[i] * fresh) / (Omega - h.rk[i] - h.lk[i - 1] - h.gm[i] * cur));
    logacc = logacc + c.count(2 * k + side, 2 * k + side) *
             std::log((Omega - h.lk[k - 1] - h.gm[k] * fresh) / (Omega - h.lk[k - 1] - h.gm[k] * cur));
    const double cmp = rs.uniform();
    bool ok = !(fresh + h.rk[0] > Omega);
    for (int i = 1; i < k && ok; ++i) ok = !(h.gm[i] * fresh + h.rk[i] + h.lk[i - 1] > Omega);
    ok = ok && !(h.gm[k] * fresh + h.lk[k - 1] > Omega) && !(fresh < tiny) && !(logacc < std::log(cmp));
    if (!ok) continue;
    c.q(side, side) = -h.rk[0] - h.gm[0] * fresh;
    c.q(side, other) = h.gm[0] * fresh;
    for (int i = 1; i < k; ++i) {
      c.q(2 * i + side, 2 * i + side) = -h.lk[i - 1] - h.rk[i] - h.gm[i] * fresh;
      c.q(2 * i + side, 2 * i + other) = h.gm[i] * fresh;
    }
    c.q(2 * k + side, 2 * k + side) = -h.lk[k - 1] - h.gm[k] * fresh;
    c.q(2 * k + side, 2 * k + other) = h.gm[k] * fresh;
  }

  // ---- rates into the next regime: updaterkappas :1582-1644 ----
  for (int j = 0; j < k; ++j) {
    UpdateStream rs{seed, iter, (uint32_t)(2 + j)};
    const HiddenRates h(c);
    const double shape = prior_k[2] + c.count(2 * j, 2 * j + 2) + c.count(2 * j + 1, 2 * j + 3);
    const double rate = prior_k[3] + c.time(2 * j) + c.time(2 * j + 1);
    const double fresh = rs.gamma(shape, 1 / rate);
    double logacc = (fresh - h.rk[j]) * (c.time(2 * j) + c.time(2 * j + 1));
    bool ok = true;
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      if (j == 0) logacc = logacc + c.count(side, side) * std::log((Omega - fresh - gl) / (Omega - h.rk[j] - gl));
      else logacc = logacc + c.count(2 * j + side, 2 * j + side) * std::log((Omega - h.lk[j - 1] - fresh - gl) / (Omega - h.lk[j - 1] - h.rk[j] - gl));
    }
    const double cmp = rs.uniform();
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      if (j == 0) ok = ok && !(fresh + gl > Omega);
      else ok = ok && !(fresh + gl + h.lk[j - 1] > Omega);
    }
    ok = ok && !(fresh < tiny) && !(logacc < std::log(cmp));
    if (!ok) continue;
    c.q(2 * j, 2 * j + 2) = fresh;
    c.q(2 * j + 1, 2 * j + 3) = fresh;
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      c.q(2 * j + side, 2 * j + side) = (j == 0) ? -fresh - gl : -fresh - h.lk[j - 1] - gl;
    }
  }

  // ---- rates back to the previous regime: updatelkappas :1648-1710 ----
  for (int j = 1; j <= k; ++j) {
    UpdateStream rs{seed, iter, (uint32_t)(2 + k + j)};
    const HiddenRates h(c);
    const double shape = prior_k[2] + c.count(2 * j, 2 * j - 2) + c.count(2 * j + 1, 2 * j - 1);
    const double rate = prior_k[3] + c.time(2 * j) + c.time(2 * j + 1);
    const double fresh = rs.gamma(shape, 1 / rate);
    double logacc = (fresh - h.lk[j - 1]) * (c.time(2 * j) + c.time(2 * j + 1));
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      if (j == k) logacc = logacc + c.count(2 * j + side, 2 * j + side) * std::log((Omega - fresh - gl) / (Omega - h.lk[j - 1] - gl));
      else logacc = logacc + c.count(2 * j + side, 2 * j + side) * std::log((Omega - h.rk[j] - fresh - gl) / (Omega - h.rk[j] - h.lk[j - 1] - gl));
    }
    const double cmp = rs.uniform();
    bool ok = true;
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      if (j == k) ok = ok && !(fresh + gl > Omega);
      else ok = ok && !(fresh + gl + h.rk[j] > Omega);
    }
    ok = ok && !(fresh < tiny) && !(logacc < std::log(cmp));
    if (!ok) continue;
    c.q(2 * j, 2 * j - 2) = fresh;
    c.q(2 * j + 1, 2 * j - 1) = fresh;
    for (int side = 0; side < 2; ++side) {
      const double gl = h.gm[j] * h.lam[side];
      c.q(2 * j + side, 2 * j + side) = (j == k) ? -fresh - gl : -fresh - h.rk[j] - gl;
    }
  }

  // ---- regime multipliers: updategammas :1714-1785 ----
  for (int j = 1; j <= k; ++j) {
    UpdateStream rs{seed, iter, (uint32_t)(2 + 2 * k + j)};
    const HiddenRates h(c);
    const double shape = prior_k[4] + c.count(2 * j, 2 * j + 1) + c.count(2 * j + 1, 2 * j);
    const double exposure = c.time(2 * j) * h.lam[0] + c.time(2 * j + 1) * h.lam[1];
    const double rate = prior_k[5] + c.time(2 * j) * h.lam[0] + c.time(2 * j + 1) * h.lam[1];
    const double fresh = rs.gamma(shape, 1 / rate);
    double logacc = (fresh - h.gm[j]) * exposure;
    for (int side = 0; side < 2; ++side) {
      if (j == k) logacc = logacc + c.count(2 * j + side, 2 * j + side) *
                                    std::log((Omega - h.lk[j - 1] - fresh * h.lam[side]) / (Omega - h.lk[j - 1] - h.gm[j] * h.lam[side]));
      else logacc = logacc + c.count(2 * j + side, 2 * j + side) *
                             std::log((Omega - h.lk[j - 1] - h.rk[j] - fresh * h.lam[side]) / (Omega - h.rk[j] - h.lk[j - 1] - h.gm[j] * h.lam[side]));
    }
    const double cmp = rs.uniform();
    bool ok = true;
    for (int side = 0; side < 2; ++side) {
      if (j == k) ok = ok && !(h.lk[j - 1] + fresh * h.lam[side] > Omega);
      else ok = ok && !(h.lk[j - 1] + fresh * h.lam[side] + h.rk[j] > Omega);
    }
    ok = ok && !(fresh < tiny) && !(logacc < std::log(cmp));
    if (!ok) continue;
    c.q(2 * j, 2 * j + 1) = fresh * h.lam[0];
    c.q(2 * j + 1, 2 * j) = fresh * h.lam[1];
    for (int side = 0; side < 2; ++side)
      c.q(2 * j + side, 2 * j + side) = (j == k) ? -h.lk[j - 1] - fresh * h.lam[side] : -h.lk[j - 1] - h.rk[j] - fresh * h.lam[side];
  }
}

}  // namespace phm
