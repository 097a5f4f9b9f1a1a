/*
 * phm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See phm_oracle.h.
 *
 * Plain-C restatement of the hot path of vnminin/phylomap, src/phylomap.cpp.
 * "parity unpinned": the reference has no tests/golden vectors and cannot be built here.
 *
 * Arithmetic contract (DESIGN.md "Arithmetic spec"; the HIP kernels restate the same):
 *   - IEEE binary64, round-to-nearest, NO fused multiply-add (build with -ffp-contract=off);
 *   - mat-vec  y_i = ((M_i0*x_0 + M_i1*x_1) + M_i2*x_2) + ...   left to right
 *     (what Armadillo's gemv_emul_tinysq does for n<=4 on a default R build);
 *   - categorical draw: first j with u*sum(p) <= p_0+..+p_j (index order, no sort; see
 *     sample_cat below for why RcppArmadillo's descending sort is not restated);
 *   - Exp(rate r) gap = (1/r) * (-phm_log(u))   (Rcpp::rexp(n,rate) multiplies by scale=1/rate);
 *   - uniforms from Philox4x32-7 keyed (seed) with counter (block, entity, iteration, replica); draw d of a stream is
 *     word d%4 of block d/4, mapped to (0,1) as (x + 0.5) 2^-32.
 */
#include "phm_oracle.h"
#include "../phylomap_amd/csrc/phm_logtab.h"     /* generated data shared with the kernels: tools/gen_log_table.py */

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32 (Salmon et al. 2011, Random123), `rounds` rounds.  Pinned by Random123's kat_vectors for   */
/* 7 and 10 rounds.  The streams of the sampler use SEVEN rounds (the fewest that pass BigCrush).          */
/* ------------------------------------------------------------------------------------------ */
#define ORC_STREAM_ROUNDS 7
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(ctr, key, 10, out); }
static void philox_stream(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(ctr, key, ORC_STREAM_ROUNDS, out); }

/* 32 random bits -> double in the OPEN interval (0,1): (x + 0.5) * 2^-32, exact (the resolution of R's unif_rand). */
double orc_u01(uint32_t x) {
  return ((double)x + 0.5) * 2.3283064365386962890625e-10;   /* 2^-32 */
}

/* draw number `draw` of stream (replica, iter, entity): Philox block draw/4, word draw%4 */
double orc_stream_u(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t iter,
                    uint32_t entity, uint32_t draw) {
  uint32_t ctr[4] = { draw >> 2, entity, iter, replica };
  uint32_t key[2] = { seed_lo, seed_hi };
  uint32_t o[4];
  philox_stream(ctr, key, o);
  return orc_u01(o[draw & 3u]);
}

/* the 32 random bits behind draw number `draw` of a stream */
uint32_t orc_stream_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t iter,
                         uint32_t entity, uint32_t draw) {
  uint32_t ctr[4] = { draw >> 2, entity, iter, replica };
  uint32_t key[2] = { seed_lo, seed_hi };
  uint32_t o[4];
  philox_stream(ctr, key, o);
  return o[draw & 3u];
}

/* Standard exponential variate from 32 random bits: -log(U), U = (k + 0.5) 2^-32 -- what Rcpp::rexp's exp_rand delivers
 * in distribution (src/phylomap.cpp:398).  y = 2k + 1 = 2^e f, f in [0.5, 1); table point c_j = 0.5 + (j + 0.5)/256 below f;
 * -log U = -((e - 33) ln 2 + log c_j + log1p(r)), r = (f - c_j)/c_j with f - c_j exact, log1p by its series to r^7; within
 * 2^-8 of U = 1 the series runs on r = f - 1 directly.  Same expression, same table (phm_logtab.h) as the kernels. */
double orc_neglog_u32(uint32_t k) {
  const double y = (double)k * 2.0 + 1.0;
  int e;
  const double f = frexp(y, &e);
  const int top = (e == 33) && (f >= 0.99609375);
  const int j = (int)((f - 0.5) * 256.0);
  const double c = 0.501953125 + (double)j * 0.00390625;
  double inv, lc;
  memcpy(&inv, &PHM_LOGTAB_INV_BITS[j], sizeof inv);
  memcpy(&lc, &PHM_LOGTAB_LOG_BITS[j], sizeof lc);
  const double r = top ? f - 1.0 : (f - c) * inv;
  const double c0 = top ? 0.0 : lc;
  const double ee = top ? 0.0 : (double)(e - 33);
  double p = 1.0 / 7.0;
  p = p * r - 1.0 / 6.0;
  p = p * r + 0.2;
  p = p * r - 0.25;
  p = p * r + 1.0 / 3.0;
  p = p * r - 0.5;
  p = p * r * r + r;
  return -(ee * 6.93147180369123816490e-01 + (c0 + (p + ee * 1.90821492927058770002e-10)));
}

/* entity tags (top two bits of the entity word) */
#define ENT_NODE   0u            /* node-state draw: entity = node id (0-based), draw 0 */
#define ENT_BSTATE (1u << 30)    /* branch interior-state uniforms: draw i-1 for position i */
#define ENT_BEXP   (2u << 30)    /* branch virtual-jump exponentials: sequential */
#define ENT_BUNIF  (3u << 30)    /* EXP path newunifSample draws: sequential */

/* ------------------------------------------------------------------------------------------ */
/* phm_log / phm_exp: deterministic elementary functions (fdlibm-style argument reduction and   */
/* minimax polynomials, one code path, basic IEEE operations only) so CPU and GPU agree bitwise. */
/* Pinned against libm in tests (<= 2 ulp).                                                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double   u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

double orc_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
    Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
    Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
    Lg7 = 1.479819860511658591e-01;
  if (!(x > 0.0)) return (x == 0.0) ? -INFINITY : NAN;
  if (isinf(x)) return x;
  int k = 0;
  if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; k -= 54; }   /* subnormal: *2^54 */
  uint64_t ux = d2u(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;          /* mantissa >= sqrt(2): use x/2 */
  ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffu);
  k += (int)(i >> 20);
  double f = u2d(ux) - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

double orc_exp(double x) {
  static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
    invln2 = 1.44269504088896338700e+00,
    P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
    P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 7.09782712893383973096e+02) return INFINITY;
  if (x < -7.45133219101941108420e+02) return 0.0;
  double hi = x, lo = 0.0;
  int k = 0;
  if (fabs(x) > 0.34657359027997264) {               /* 0.5*ln2 */
    k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    double t = (double)k;
    hi = x - t * ln2HI;
    lo = t * ln2LO;
  }
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return ldexp(y, k);
}


/* ------------------------------------------------------------------------------------------ */
/* "R-stream" mode (rng.mode = 2): R's default generator consumed sequentially in the reference's draw order, so that a
 * machine WITH R + phylomap can be compared sample for sample (tools/r_parity/).  Restated from the published R sources
 * (RNG.c: Mersenne-Twister with set.seed's LCG scrambling; sexp.c: Ahrens-Dieter exp_rand) -- third-party code that is
 * not under /root/reference and could NOT be checked here (no R in the build image): UNVERIFIED.  Covers what the
 * fixed-Q MCMC variants draw: unif_rand (via runif / RcppArmadillo::sample) and exp_rand (via Rcpp::rexp).            */
/* ------------------------------------------------------------------------------------------ */
static uint32_t r_mt[624];
static int r_mti = 625;
static void r_set_seed(uint32_t seed) {
  for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;             /* RNG_Init: initial scrambling */
  uint32_t dummy0 = 0;
  for (int j = 0; j < 625; ++j) {
    seed = 69069u * seed + 1u;
    if (j == 0) dummy0 = seed; else r_mt[j - 1] = seed;
  }
  (void)dummy0;
  r_mti = 624;                                                        /* FixupSeeds: dummy[0] = 624 */
}
static double r_unif_rand(void) {
  static const uint32_t mag01[2] = { 0x0u, 0x9908b0dfu };
  if (r_mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (r_mt[kk] & 0x80000000u) | (r_mt[kk + 1] & 0x7fffffffu); r_mt[kk] = r_mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u]; }
    for (; kk < 623; kk++) { uint32_t y = (r_mt[kk] & 0x80000000u) | (r_mt[kk + 1] & 0x7fffffffu); r_mt[kk] = r_mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u]; }
    uint32_t y = (r_mt[623] & 0x80000000u) | (r_mt[0] & 0x7fffffffu);
    r_mt[623] = r_mt[396] ^ (y >> 1) ^ mag01[y & 1u];
    r_mti = 0;
  }
  uint32_t y = r_mt[r_mti++];
  y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
  double x = (double)y * 2.3283064365386963e-10;                      /* [0,1) */
  if (x <= 0.0) return 0.5 * 2.328306437080797e-10;                   /* fixup(): never 0 or 1 */
  if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * 2.328306437080797e-10;
  return x;
}
static double r_exp_rand(void) {                                      /* sexp.c, Ahrens & Dieter (1972) */
  static const double q[] = { 0.6931471805599453, 0.9333736875190459, 0.9888777961838675, 0.9984589039328339,
    0.9998292811061389, 0.9999833164100727, 0.9999985691438767, 0.9999998906925558, 0.9999999924734159,
    0.9999999995283275, 0.9999999999728814, 0.9999999999985598, 0.9999999999999289, 0.9999999999999968,
    0.9999999999999999, 1.0000000000000000 };
  double a = 0.;
  double u = r_unif_rand();
  while (u <= 0. || u >= 1.) u = r_unif_rand();
  for (;;) { u += u; if (u > 1.) break; a += q[0]; }
  u -= 1.;
  if (u <= q[0]) return a + u;
  int i = 0;
  double ustar = r_unif_rand(), umin = ustar;
  do { ustar = r_unif_rand(); if (umin > ustar) umin = ustar; i++; } while (u > q[i]);
  return a + umin * q[0];
}
/* RcppArmadillo::sample(sts,1,TRUE,p): FixProb (sequential sum of the positive entries, p/sum), sort DESCENDING
 * (std::sort on <= 16 elements is an insertion sort, hence stable for ties; larger n is implementation-defined),
 * cumulative sum, first jj < n-1 with u <= cum[jj], else the last; returns the original index. */
static int sample_cat_R(const double* p, int n, double u, int* err) {
  double sum = 0.0; int npos = 0;
  for (int i = 0; i < n; ++i) { if (!isfinite(p[i]) || p[i] < 0.0) { *err |= ORC_ERR_ZERO_PROB; return 0; } if (p[i] > 0.0) { sum += p[i]; npos++; } }
  if (npos == 0) { *err |= ORC_ERR_ZERO_PROB; return 0; }
  double pr[64]; int perm[64];
  if (n > 64) { *err |= ORC_ERR_BAD_INPUT; return 0; }
  for (int i = 0; i < n; ++i) { pr[i] = p[i] / sum; perm[i] = i; }
  for (int i = 1; i < n; ++i) {                                       /* stable insertion sort, descending */
    double v = pr[i]; int id = perm[i]; int j = i - 1;
    while (j >= 0 && pr[j] < v) { pr[j + 1] = pr[j]; perm[j + 1] = perm[j]; --j; }
    pr[j + 1] = v; perm[j + 1] = id;
  }
  double cum = 0.0; int jj;
  for (jj = 0; jj < n - 1; ++jj) { cum += pr[jj]; if (u <= cum) break; }
  return perm[jj];
}
/* Rf_dpois(k, lambda, 0) as newunifSample calls it (src/phylomap.cpp:107,128): R's saddle-point dpois_raw (nmath/dpois.c,
 * Loader 2000) with stirlerr (nmath/stirlerr.c) and bd0 (nmath/bd0.c), restated from the published sources as they stood up to
 * R 4.0.x.  R >= 4.1 evaluates the deviance part through ebd0() (a 128-entry table of logarithms, not restated): the two differ
 * in the last ulps, which moves a jump count only when the uniform falls within ~1e-16 of a partial sum.  UNVERIFIED here
 * (no R); checked against exp(-lambda) lambda^k / k! in 50-digit arithmetic to <= 4 ulp (tests/test_oracle_cpu.py). */
static double r_stirlerr(double n) {
  static const double sferr_halves[31] = {
    0.0, 0.1534264097200273452913848, 0.0810614667953272582196702, 0.0548141210519176538961390, 0.0413406959554092940938221,
    0.03316287351993628748511048, 0.02767792568499833914878929, 0.02374616365629749597132920, 0.02079067210376509311152277,
    0.01848845053267318523077934, 0.01664469118982119216319487, 0.01513497322191737887351255, 0.01387612882307074799874573,
    0.01281046524292022692424986, 0.01189670994589177009505572, 0.01110455975820691732662991, 0.010411265261972096497478567,
    0.009799416126158803298389475, 0.009255462182712732917728637, 0.008768700134139385462952823, 0.008330563433362871256469318,
    0.007934114564314020547248100, 0.007573675487951840794972024, 0.007244554301320383179543912, 0.006942840107209529865664152,
    0.006665247032707682442354394, 0.006408994188004207068439631, 0.006171712263039457647532867, 0.005951370112758847735624416,
    0.005746216513010115682023589, 0.005554733551962801371038690 };
  const double S0 = 0.083333333333333333333, S1 = 0.00277777777777777777778, S2 = 0.00079365079365079365079365,
               S3 = 0.000595238095238095238095238, S4 = 0.0008417508417508417508417508;
  if (n <= 15.0) {
    double nn = n + n;
    if (nn == (int)nn) return sferr_halves[(int)nn];
    return lgamma(n + 1.) - (n + 0.5) * log(n) + n - 0.918938533204672741780329736406;      /* M_LN_SQRT_2PI */
  }
  double nn = n * n;
  if (n > 500) return (S0 - S1 / nn) / n;
  if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
  if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
  return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}
static double r_bd0(double x, double np) {
  if (fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np);
    double s = (x - np) * v;
    if (fabs(s) < 2.2250738585072014e-308) return s;
    double ej = 2 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      double s1 = s + ej / ((j << 1) + 1);
      if (s1 == s) return s1;
      s = s1;
    }
  }
  return x * log(x / np) + np - x;
}
static double r_dpois(double x, double lambda) {
  if (lambda == 0) return (x == 0) ? 1. : 0.;
  if (x < 0) return 0.;
  if (x <= lambda * 2.2250738585072014e-308) return exp(-lambda);
  if (lambda < x * 2.2250738585072014e-308) return exp(-lambda + x * log(lambda) - lgamma(x + 1));
  return exp(-r_stirlerr(x) - r_bd0(x, lambda)) / sqrt(6.283185307179586476925286766559 * x);   /* R_D_fexp(M_2PI x, .) */
}
double orc_r_dpois(double x, double lambda) { return r_dpois(x, lambda); }

/* Rf_rgamma(a, scale) as the rate updates call it (src/phylomap.cpp:1202,1235,1463,1538,1612,1681,1748): R's nmath/rgamma.c --
 * Ahrens & Dieter GS (1974) for a < 1, GD (1982) for a >= 1 -- with norm_rand() of the default kind INVERSION (two unif_rand()
 * through qnorm5 = Wichura's AS 241, nmath/qnorm.c) and exp_rand().  Third-party code outside /root/reference, restated from the
 * published sources and UNVERIFIED against R here (no R): the quantile function is checked against scipy to 1e-15 and the
 * variates distributionally (tests/test_oracle_cpu.py); tools/r_parity pins them wherever R exists.  R keeps the a-dependent
 * set-up of GD in static variables keyed by a; recomputing it on every call gives the same values. */
static double r_qnorm_std(double p) {                                  /* qnorm5(p, 0, 1, lower = TRUE, log = FALSE), 0 < p < 1 */
  const double q = p - 0.5;
  double r, val;
  if (fabs(q) <= 0.425) {                                              /* 0.075 <= p <= 0.925 */
    r = .180625 - q * q;
    return q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r + 45921.953931549871457) * r +
                   13731.693765509461125) * r + 1971.5909503065514427) * r + 133.14166789178437745) * r + 3.387132872796366608) /
           (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r + 21213.794301586595867) * r +
               5394.1960214247511077) * r + 687.1870074920579083) * r + 42.313330701600911252) * r + 1.);
  }
  r = (q < 0) ? p : 1.0 - p;                                           /* min(p, 1 - p) */
  r = sqrt(-log(r));
  if (r <= 5.) {
    r += -1.6;
    val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r + 1.27045825245236838258) * r +
              3.64784832476320460504) * r + 5.7694972214606914055) * r + 4.6303378461565452959) * r + 1.42343711074968357734) /
          (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r + .14810397642748007459) * r +
              .68976733498510000455) * r + 1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.);
  } else {
    r += -5.;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r + .026532189526576123093) * r +
              .29656057182850489123) * r + 1.7848265399172913358) * r + 5.4637849111641143699) * r + 6.6579046435011037772) /
          (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
              .0148753612908506148525) * r + .13692988092273580531) * r + .59983220655588793769) * r + 1.);
  }
  return (q < 0.0) ? -val : val;
}
static double r_norm_rand(void) {                                      /* snorm.c, INVERSION: unif_rand() alone has too few bits */
  const double BIG = 134217728;                                        /* 2^27 */
  double u = r_unif_rand();
  u = (int)(BIG * u) + r_unif_rand();
  return r_qnorm_std(u / BIG);
}
static double r_rgamma(double a, double scale) {
  const double sqrt32 = 5.656854, exp_m1 = 0.36787944117144233;
  const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191, q4 = 0.00144121, q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
  const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062, a4 = -0.1662921, a5 = 0.1423657, a6 = -0.1367177, a7 = 0.1233795;
  double e, p, q, r, t, u, v, w, x, ret_val;
  if (isnan(a) || isnan(scale)) return NAN;
  if (a <= 0.0 || scale <= 0.0) return (scale == 0. || a == 0.) ? 0. : NAN;
  if (!isfinite(a) || !isfinite(scale)) return INFINITY;
  if (a < 1) {                                                         /* GS */
    e = 1.0 + exp_m1 * a;
    for (;;) {
      p = e * r_unif_rand();
      if (p >= 1.0) {
        x = -log((e - p) / a);
        if (r_exp_rand() >= (1.0 - a) * log(x)) break;
      } else {
        x = exp(log(p) / a);
        if (r_exp_rand() >= x) break;
      }
    }
    return scale * x;
  }
  /* GD.  Step 1 */
  const double s2 = a - 0.5, s = sqrt(s2), d = sqrt32 - s * 12;
  /* Step 2: t = standard normal deviate, x = (s, 1/2)-normal deviate; immediate acceptance */
  t = r_norm_rand();
  x = s + 0.5 * t;
  ret_val = x * x;
  if (t >= 0) return scale * ret_val;
  /* Step 3: squeeze acceptance */
  u = r_unif_rand();
  if (d * u <= t * t * t) return scale * ret_val;
  /* Step 4 */
  r = 1 / a;
  const double q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
  double b, si, c;
  if (a <= 3.686) { b = 0.463 + s + 0.178 * s2; si = 1.235; c = 0.195 / s - 0.079 + 0.16 * s; }
  else if (a <= 13.022) { b = 1.654 + 0.0076 * s2; si = 1.68 / s + 0.275; c = 0.062 / s + 0.024; }
  else { b = 1.77; si = 0.75; c = 0.1515 / s; }
  /* Step 5-7: quotient test */
  if (x > 0.0) {
    v = t / (s + s);
    if (fabs(v) <= 0.25) q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
    else q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
    if (log(1.0 - u) <= q) return scale * ret_val;
  }
  for (;;) {
    /* Step 8: e = standard exponential, u = uniform, t = (b, si)-double exponential */
    e = r_exp_rand();
    u = r_unif_rand();
    u = u + u - 1.0;
    t = (u < 0.0) ? b - si * e : b + si * e;
    /* Step 9: rejection if t < tau(1) */
    if (t >= -0.71874483771719) {
      v = t / (s + s);
      if (fabs(v) <= 0.25) q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
      else q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
      /* Step 11: hat acceptance */
      if (q > 0.0) {
        w = expm1(q);
        if (c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
      }
    }
  }
  x = s + 0.5 * t;
  return scale * x * x;
}
double orc_r_qnorm(double p) { return r_qnorm_std(p); }
/* set.seed(seed); rnorm(n_norm); rgamma(n_gamma, shape, scale = scale) */
int orc_rstream_gamma_selftest(uint32_t seed, int n_norm, int n_gamma, double shape, double scale, double* norm_out, double* gamma_out) {
  r_set_seed(seed);
  for (int i = 0; i < n_norm; ++i) norm_out[i] = r_norm_rand();
  for (int i = 0; i < n_gamma; ++i) gamma_out[i] = r_rgamma(shape, scale);
  return 0;
}

int orc_rstream_selftest(uint32_t seed, int n_unif, int n_exp, double* unif_out, double* exp_out) {   /* set.seed; runif(n_unif); rexp(n_exp) */
  r_set_seed(seed);
  for (int i = 0; i < n_unif; ++i) unif_out[i] = r_unif_rand();
  for (int i = 0; i < n_exp; ++i) exp_out[i] = r_exp_rand();
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RNG front end                                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { orc_rng* r; int err; uint32_t iter_base; } rngctx;   /* iter_base: added to the Philox iteration word */

static double draw_u(rngctx* c, uint32_t iter, uint32_t entity, uint32_t draw) {
  orc_rng* r = c->r;
  if (r->mode == 1) {
    if (r->pos_u >= r->n_u) { c->err |= ORC_ERR_TAPE; return 0.5; }
    return r->tape_u[r->pos_u++];
  }
  if (r->mode == 2) return r_unif_rand();
  return orc_stream_u(r->seed_lo, r->seed_hi, r->replica, iter + c->iter_base, entity, draw);
}
/* standard exponential: orc_neglog_u32 of the draw's 32 bits (R's exp_rand is Ahrens-Dieter; restated in R-stream mode only) */
static double draw_e(rngctx* c, uint32_t iter, uint32_t entity, uint32_t draw) {
  orc_rng* r = c->r;
  if (r->mode == 1) {
    if (r->pos_e >= r->n_e) { c->err |= ORC_ERR_TAPE; return 1.0; }
    return r->tape_e[r->pos_e++];
  }
  if (r->mode == 2) return r_exp_rand();
  return orc_neglog_u32(orc_stream_word(r->seed_lo, r->seed_hi, r->replica, iter + c->iter_base, entity, draw));
}

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers; all matrices row-major here                                             */
/* ------------------------------------------------------------------------------------------ */
/* y = M x  -- arma `(*B2)*(*vec)` at src/phylomap.cpp:448 (mmmmvFORpl), :290 */
static void matvec(const double* M, const double* x, double* y, int n) {
  for (int i = 0; i < n; ++i) {
    double acc = M[i * n] * x[0];
    for (int j = 1; j < n; ++j) acc += M[i * n + j] * x[j];
    y[i] = acc;
  }
}
/* y = M^T x -- `(*B4)*vec` with B4 = trans(B2), src/phylomap.cpp:434 (Tvmmp), :918 */
static void matTvec(const double* M, const double* x, double* y, int n) {
  for (int c = 0; c < n; ++c) {
    double acc = M[c] * x[0];
    for (int r = 1; r < n; ++r) acc += M[r * n + c] * x[r];
    y[c] = acc;
  }
}
/*
 * Mat-vec of the MCMC sweep (chains of B).  n <= 4: Armadillo's gemv_emul_tinysq, i.e. the unfused left-to-right sums above.
 * n > 4: the reference calls BLAS dgemv, whose summation order is unknowable, so the spec is the order
 * the MI355X matrix cores produce: one FUSED multiply-add per term, j ascending, starting from +0
 * (v_mfma_f64_16x16x4 accumulates its k-slices exactly like this; orc_matexp_fma is the precedent).
 */
static void mcmc_matvec(const double* M, const double* x, double* y, int n) {
  if (n <= 4) { matvec(M, x, y, n); return; }
  int i = 0;
  for (; i + 4 <= n; i += 4) {                     /* four rows side by side: independent chains, each in the spec's order */
    const double *r0 = M + (size_t)i * n, *r1 = r0 + n, *r2 = r1 + n, *r3 = r2 + n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = 0; j < n; ++j) {
      const double xj = x[j];
      a0 = fma(r0[j], xj, a0); a1 = fma(r1[j], xj, a1); a2 = fma(r2[j], xj, a2); a3 = fma(r3[j], xj, a3);
    }
    y[i] = a0; y[i + 1] = a1; y[i + 2] = a2; y[i + 3] = a3;
  }
  for (; i < n; ++i) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(M[i * n + j], x[j], acc);
    y[i] = acc;
  }
}
static void mcmc_matTvec(const double* M, const double* x, double* y, int n) {
  if (n <= 4) { matTvec(M, x, y, n); return; }
  for (int c = 0; c < n; ++c) y[c] = 0.0;           /* row r of M contributes to every y[c]: r ascending per entry = the spec's order */
  for (int r = 0; r < n; ++r) {
    const double xr = x[r];
    const double* row = M + (size_t)r * n;
    for (int c = 0; c < n; ++c) y[c] = fma(row[c], xr, y[c]);
  }
}
/* Sum of a partial-likelihood row for the normalisation of :525.  n <= 4: left to right.  n > 4: four interleaved partial
 * sums t_g = p_g + p_{g+4} + p_{g+8} + ... (ascending), combined as (t_0 + t_1) + (t_2 + t_3) -- the order in which the
 * accumulator tiles of the matrix cores hold a column (rows g, g+4, ... in lane group g). */
static double mcmc_rowsum(const double* p, int n) {
  if (n <= 4) {
    double s = p[0];
    for (int c = 1; c < n; ++c) s += p[c];
    return s;
  }
  double t[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c = 0; c < n; ++c) t[c & 3] += p[c];
  return (t[0] + t[1]) + (t[2] + t[3]);
}
/* v <- M^k v : mmmmvFORpl / spmmmmvFORpl, src/phylomap.cpp:446-457 */
static void chain(const double* M, double* v, int k, int n, double* tmp) {
  for (int i = 0; i < k; ++i) { mcmc_matvec(M, v, tmp, n); memcpy(v, tmp, sizeof(double) * n); }
}
/* v <- (M^T)^k v : Tvmmp / spvmmmm, src/phylomap.cpp:431-444 */
static void chainT(const double* M, double* v, int k, int n, double* tmp) {
  for (int i = 0; i < k; ++i) { mcmc_matTvec(M, v, tmp, n); memcpy(v, tmp, sizeof(double) * n); }
}

/*
 * Categorical draw.  Replaces RcppArmadillo::sample(sts,1,TRUE,p) (src/phylomap.cpp:304,627,655...).
 * RcppArmadillo normalises p, sorts it DESCENDING (tie order implementation-defined, so it cannot be
 * pinned), then inverse-CDF with `u <= cum`.  The sort changes which index a given u maps to but not the
 * distribution; with a counter-based stream there is no R stream to stay aligned with, so the oracle
 * uses index order, keeps the `<=`, and compares u*sum(p) with the running sum instead of dividing
 * every p_j.  All-zero / non-finite p (RcppArmadillo throws) raises ORC_ERR_ZERO_PROB.
 */
static int g_rstream = 0;     /* set by the drivers when rng.mode == 2 */
static int sample_cat(const double* p, int n, double u, int* err) {
  if (g_rstream) return sample_cat_R(p, n, u, err);
  double total = p[0];
  for (int j = 1; j < n; ++j) total += p[j];
  if (!(total > 0.0) || isinf(total)) { *err |= ORC_ERR_ZERO_PROB; return 0; }
  double thr = u * total;
  double cum = p[0];
  if (thr <= cum) return 0;
  for (int j = 1; j < n; ++j) { cum += p[j]; if (thr <= cum) return j; }
  int last = n - 1;
  while (last > 0 && !(p[last] > 0.0)) --last;
  return last;
}

/* sampleOnce, src/phylomap.cpp:81-90: no sort, per-element division, strict `<`; may run off the end */
static int sampleOnce(const double* w, int n, double u, int* err) {
  double total = w[0];
  for (int j = 1; j < n; ++j) total += w[j];
  double cum = 0.0;
  int i;
  for (i = 0; i < n; ++i) { cum += w[i] / total; if (u < cum) break; }
  if (i >= n) { *err |= ORC_ERR_SAMPLEONCE; i = n - 1; }   /* reference would index out of range */
  return i;
}

/* ------------------------------------------------------------------------------------------ */
/* Branch container: struct Branch + makeabranch, src/phylomap.cpp:18-34                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double* d; int32_t* s; int m, cap; } Branch;

static void br_reserve(Branch* b, int need) {
  if (need <= b->cap) return;
  int nc = b->cap ? b->cap : 8;
  while (nc < need) nc *= 2;
  b->d = (double*)realloc(b->d, sizeof(double) * nc);
  b->s = (int32_t*)realloc(b->s, sizeof(int32_t) * nc);
  b->cap = nc;
}
static void makeabranch(Branch* b, const double* maps, const int32_t* names, int m) {
  b->d = NULL; b->s = NULL; b->cap = 0; b->m = 0;
  br_reserve(b, m > 0 ? m : 1);
  for (int i = 0; i < m; ++i) { b->d[i] = maps[i]; b->s[i] = names[i] - 1; }   /* :29 1-based -> 0-based */
  b->m = m;
}

/* shortener, src/phylomap.cpp:44-73: merge equal neighbours, then count a->b (a != b) transitions */
static int shortener_arr(double* d, int32_t* s, int m, int n, double* stats, int64_t stride, int iter) {
  if (n < 0) {                 /* shortenerbf, src/phylomap.cpp:997-1028: count EVERY consecutive pair, self pairs */
    n = -n;                    /* included, into n + a*n + b BEFORE merging (:1010-1014); then merge as usual  */
    for (int i = 1; i < m; ++i) stats[(int64_t)(n + s[i - 1] * n + s[i]) * stride + iter] += 1.0;
    int w2 = 0;
    for (int i = 1; i < m; ++i) {
      if (s[i] != s[w2]) { ++w2; d[w2] = d[i]; s[w2] = s[i]; }
      else d[w2] = d[w2] + d[i];
    }
    return (m > 0) ? w2 + 1 : 0;
  }
  int w = 0;
  for (int i = 1; i < m; ++i) {
    if (s[i] != s[w]) { ++w; d[w] = d[i]; s[w] = s[i]; }
    else d[w] = d[w] + d[i];                                   /* :54 */
  }
  int mm = (m > 0) ? w + 1 : 0;
  for (int i = 1; i < mm; ++i) {
    int a = s[i - 1], b = s[i];
    if (a < b) stats[(int64_t)(n + a * (n - 1) + b - 1) * stride + iter] += 1.0;   /* :65 */
    if (a > b) stats[(int64_t)(n + a * (n - 1) + b) * stride + iter] += 1.0;       /* :66 */
  }
  return mm;
}
int orc_shortener(double* d, int32_t* s, int m, int n, double* stats_row) {
  return shortener_arr(d, s, m, n, stats_row, 1, 0);
}

/* matTospmat, src/phylomap.cpp:801-816: keep entries > 1e-7 only */
void orc_matTospmat(const double* B, int n, double* out) {
  for (int i = 0; i < n * n; ++i) out[i] = (B[i] > 1e-7) ? B[i] : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* resamplebranchstates, src/phylomap.cpp:264-308 (SPARSE twin :218-261)                        */
/*   Bchain: matrix of the backward recursion (B2, or the thresholded B3 in the SPARSE variant)  */
/*   Brow:   matrix whose rows give the forward step (always the dense B2, :254/:301)            */
/* ------------------------------------------------------------------------------------------ */
static void resamplebranchstates(Branch* br, const double* Bchain, const double* Brow, int n,
                                 rngctx* rc, uint32_t iter, uint32_t branch_id, double* scratch) {
  int ss = br->m;
  if (ss <= 2) return;                                         /* :269-270 */
  double* bpws = scratch;                                      /* n x ss, column j at bpws + j*n */
  double* p = scratch + (size_t)n * ss;
  for (int i = 0; i < n; ++i) bpws[i] = 0.0;
  bpws[br->s[ss - 1]] = 1.0;                                   /* :280 */
  for (int j = 1; j < ss - 1; ++j) mcmc_matvec(Bchain, bpws + (size_t)(j - 1) * n, bpws + (size_t)j * n, n);   /* :290 */
  for (int i = 1; i < ss - 1; ++i) {
    const double* row = Brow + (size_t)br->s[i - 1] * n;
    const double* beta = bpws + (size_t)(ss - i - 1) * n;
    for (int c = 0; c < n; ++c) p[c] = row[c] * beta[c];       /* :301 */
    double u = draw_u(rc, iter, ENT_BSTATE | branch_id, (uint32_t)(i - 1));
    br->s[i] = sample_cat(p, n, u, &rc->err);                  /* :304 */
  }
}

/* sampleabranch, src/phylomap.cpp:370-413 (SPARSE twin :318-364) */
static void sampleabranch(Branch* br, const double* Bchain, const double* Brow, double Omega,
                          const double* Qdiag, int n_signed, double* stats, int64_t stride, int iter,
                          rngctx* rc, uint32_t branch_id, double** scratch, size_t* scratch_len,
                          Branch* tmp) {
  const int n = n_signed < 0 ? -n_signed : n_signed;       /* negative: sampleabranchbf :1031-1074 (bf counting) */
  size_t need = (size_t)n * (br->m + 2);
  if (need > *scratch_len) { *scratch = (double*)realloc(*scratch, sizeof(double) * need); *scratch_len = need; }
  resamplebranchstates(br, Bchain, Brow, n, rc, (uint32_t)iter, branch_id, *scratch);   /* :376 */
  br->m = shortener_arr(br->d, br->s, br->m, n_signed, stats, stride, iter);            /* :377 / :1038 */
  /* re-insert virtual jumps: gaps ~ Exp(rate Omega + Q[s,s]) until the segment is used up (:391-410) */
  tmp->m = 0;
  uint32_t edraw = 0;
  int stuck = 0;
  for (int i = 0; i < br->m; ++i) {
    double segmentlength = br->d[i];
    double totallengthinserted = 0.0;
    int s = br->s[i];
    /* A merged segment whose length is not > 0 skips the while loop WITHOUT advancing the list
       iterators (:397, :405-406), so the reference's remaining for-iterations all look at that same
       element: it and every later segment are left untouched. */
    if (stuck || !(0.0 < segmentlength)) {
      stuck = 1;
      br_reserve(tmp, tmp->m + 1);
      tmp->d[tmp->m] = segmentlength; tmp->s[tmp->m] = s; tmp->m++;
      continue;
    }
    double r = Omega + Qdiag[s];                               /* :395 */
    double scale = 1.0 / r;                                    /* Rcpp::rexp(n, rate): scale = 1/rate */
    while (totallengthinserted < segmentlength) {
      double rl = scale * draw_e(rc, (uint32_t)iter, ENT_BEXP | branch_id, edraw++);   /* :398 */
      br_reserve(tmp, tmp->m + 1);
      if ((totallengthinserted + rl) < segmentlength) {
        tmp->d[tmp->m] = rl; tmp->s[tmp->m] = s; tmp->m++;
        totallengthinserted += rl;
      } else {
        tmp->d[tmp->m] = segmentlength - totallengthinserted; tmp->s[tmp->m] = s; tmp->m++;
        totallengthinserted = segmentlength;
      }
    }
  }
  br_reserve(br, tmp->m);
  memcpy(br->d, tmp->d, sizeof(double) * tmp->m);
  memcpy(br->s, tmp->s, sizeof(int32_t) * tmp->m);
  br->m = tmp->m;
}

/* updatedwelltimes, src/phylomap.cpp:745-757 */
static void updatedwelltimes(int iter, const Branch* br, double* stats, int64_t stride) {
  for (int i = 0; i < br->m; ++i) stats[(int64_t)br->s[i] * stride + iter] += br->d[i];
}

/* updatenodestates, src/phylomap.cpp:460-475: rm is 1-based states per node */
static void updatenodestates(Branch* brs, const int32_t* edge1, const int32_t* edge2, int E, const int32_t* rm) {
  for (int i = 0; i < E; ++i) {
    brs[i].s[0] = rm[edge1[i] - 1] - 1;                        /* :469 */
    brs[i].s[brs[i].m - 1] = rm[edge2[i] - 1] - 1;             /* :472 (m==1: child wins) */
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Pruning sweeps                                                                              */
/* ------------------------------------------------------------------------------------------ */
static int check_tree(const orc_tree* x) {
  if (!x || x->n_tips < 2 || x->n_node != x->n_tips - 1 || x->n_edge != 2 * x->n_tips - 2) return ORC_ERR_BAD_INPUT;
  return 0;
}

/* makePLrcpp :503-514 / makePLrcpp_bigtree :516-529 / SPARSEmakePLrcpp :490-501 */
static void makePL(const int32_t* edge1, const int32_t* edge2, int Nnode, double* PL, const int32_t* ne,
                   const double* Bchain, const int32_t* branchlengths, int normalise, int n, double* w) {
  double* first = w; double* second = w + n; double* tmp = w + 2 * n;
  for (int i = 0; i < Nnode; ++i) {
    int ea = ne[2 * i] - 1, eb = ne[2 * i + 1] - 1;
    memcpy(first, PL + (size_t)(edge2[eb] - 1) * n, sizeof(double) * n);     /* :508 */
    memcpy(second, PL + (size_t)(edge2[ea] - 1) * n, sizeof(double) * n);    /* :509 */
    chain(Bchain, first, branchlengths[eb] - 1, n, tmp);
    chain(Bchain, second, branchlengths[ea] - 1, n, tmp);
    double* row = PL + (size_t)(edge1[ea] - 1) * n;
    for (int c = 0; c < n; ++c) row[c] = first[c] * second[c];               /* :510 */
    if (normalise) {                                                         /* :525 */
      double s = mcmc_rowsum(row, n);
      for (int c = 0; c < n; ++c) row[c] = row[c] / s;
    }
  }
}

int orc_makePL(const orc_tree* x, int n, const double* Bchain, const int32_t* nen,
               const int32_t* seg_count, int normalise, double* PL) {
  int e = check_tree(x); if (e) return e;
  int E = x->n_edge, T = x->n_tips;
  const int32_t* edge1 = x->edge; const int32_t* edge2 = x->edge + E;
  memset(PL, 0, sizeof(double) * (size_t)(2 * x->n_node + 1) * n);
  for (int i = 0; i < T; ++i) PL[(size_t)i * n + (x->states[i] - 1)] = 1.0;  /* :912-914 */
  double* w = (double*)malloc(sizeof(double) * 3 * n);
  makePL(edge1, edge2, x->n_node, PL, nen, Bchain, seg_count, normalise, n, w);
  free(w);
  return 0;
}

/* makePLold :2877-2895 / makePLexp :2899-2906: P_b = TransProb.slice(b), row-major n x n per edge */
/* rescale != 0: every internal row is divided by its sum (left to right) -- NOT in the reference (makePLexp has no rescaling,
 * so sumstatEXP underflows on multi-hundred-tip trees); the node draws are invariant to positive scaling of a row, so this
 * is the same sampler in exact arithmetic (what makePLrcpp_bigtree :525 does for the MCMC side). */
static void makePLexp(const int32_t* edge1, const int32_t* edge2, int Nnode, double* PL, const int32_t* ne,
                      const double* P, int n, double* w, int rescale) {
  double* a = w; double* b = w + n;
  for (int i = 0; i < Nnode; ++i) {
    int ea = ne[2 * i] - 1, eb = ne[2 * i + 1] - 1;
    matvec(P + (size_t)ea * n * n, PL + (size_t)(edge2[ea] - 1) * n, a, n);
    matvec(P + (size_t)eb * n * n, PL + (size_t)(edge2[eb] - 1) * n, b, n);
    double* row = PL + (size_t)(edge1[ea] - 1) * n;
    for (int c = 0; c < n; ++c) row[c] = a[c] * b[c];                        /* :2903 */
    if (rescale) {
      double s = row[0];
      for (int c = 1; c < n; ++c) s += row[c];
      for (int c = 0; c < n; ++c) row[c] = row[c] / s;
    }
  }
}
int orc_makePLexp(const orc_tree* x, int n, const double* P, const int32_t* nen, double* PL) {
  int e = check_tree(x); if (e) return e;
  int E = x->n_edge, T = x->n_tips;
  memset(PL, 0, sizeof(double) * (size_t)(2 * x->n_node + 1) * n);
  for (int i = 0; i < T; ++i) PL[(size_t)i * n + (x->states[i] - 1)] = 1.0;
  double* w = (double*)malloc(sizeof(double) * 2 * n);
  makePLexp(x->edge, x->edge + E, x->n_node, PL, nen, P, n, w, 0);
  free(w);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sampleinternalnodesMCMC :591-663 (_bigtree :666-738, SPARSE :535-587)                        */
/* ------------------------------------------------------------------------------------------ */
static void sampleinternalnodesMCMC(Branch* brs, int E, double* PL, const double* pid, const double* Bchain,
                                    int root, const int32_t* nodelist, int nll, const int32_t* ne,
                                    const int32_t* edge1, const int32_t* edge2, int Nnode,
                                    const int32_t* states, int T, int normalise, int n,
                                    int faithful_search, const int32_t* edge_of_child,
                                    rngctx* rc, uint32_t iter, int32_t* rm, int32_t* branchlengths, double* w,
                                    int ks, double* root_out) {
  for (int i = 0; i < E; ++i) branchlengths[i] = brs[i].m;                   /* :598-599 */
  for (int i = 0; i < 2 * T - 1; ++i) rm[i] = 0;
  for (int i = 0; i < T; ++i) rm[i] = states[i] - 1;                         /* :612 */
  makePL(edge1, edge2, Nnode, PL, ne, Bchain, branchlengths, normalise, n, w);   /* :616 */
  double* p = w; double* vecc = w + n; double* tmp = w + 2 * n;
  for (int c = 0; c < n; ++c) p[c] = pid[c] * PL[(size_t)(root - 1) * n + c];    /* :618 */
  rm[root - 1] = sample_cat(p, n, draw_u(rc, iter, ENT_NODE | (uint32_t)(root - 1), 0), &rc->err);   /* :627 */
  (void)root_out;
  for (int i = 0; i < nll; ++i) {
    int cn = nodelist[i] - 1;                                                /* :641 */
    int j = 0;
    if (faithful_search) { while (edge2[j] != nodelist[i]) j++; }            /* :643, O(E) per node */
    else j = edge_of_child[cn];
    int pn = edge1[j] - 1;
    int ps = rm[pn];
    for (int c = 0; c < n; ++c) vecc[c] = 0.0;
    vecc[ps] = 1.0;                                                          /* :650 */
    chainT(Bchain, vecc, branchlengths[j] - 1, n, tmp);                      /* :651 Tvmmp / :576 spvmmmm */
    for (int c = 0; c < n; ++c) p[c] = vecc[c] * PL[(size_t)cn * n + c];
    rm[cn] = sample_cat(p, n, draw_u(rc, iter, ENT_NODE | (uint32_t)cn, 0), &rc->err);   /* :655 */
  }
  if (ks) {
    *root_out = (double)rm[root - 1];                                        /* :1350-1352 / :1123, 0-based */
    for (int i = 0; i < E && ks == 1; ++i) {                                 /* ks: tips "don't remain the same" :1384-1397 */
      if (edge2[i] <= T) {
        int cn = edge2[i] - 1, ps = rm[edge1[i] - 1];
        for (int c = 0; c < n; ++c) vecc[c] = 0.0;
        vecc[ps] = 1.0;
        chainT(Bchain, vecc, branchlengths[i] - 1, n, tmp);
        for (int c = 0; c < n; ++c) p[c] = vecc[c] * PL[(size_t)cn * n + c];
        rm[cn] = sample_cat(p, n, draw_u(rc, iter, ENT_NODE | (uint32_t)cn, 0), &rc->err);
      }
    }
  }
  for (int i = 0; i < 2 * T - 1; ++i) rm[i] = rm[i] + 1;                     /* :659 */
}

/* ------------------------------------------------------------------------------------------ */
/* maketreelistMCMC :891-935, _bigtree :942-986, SPARSE :822-870; per-iteration sweep =         */
/* treesample :775-785 / treesample_bigtree :787-797 / SPARSEtreesample :761-770                */
/* ------------------------------------------------------------------------------------------ */
static void fill_dump(orc_dump* dump, const Branch* brs, int E, const int32_t* rm, int nn, const double* PL, size_t pl_len) {
  if (!dump) return;
  if (dump->node_states && rm) memcpy(dump->node_states, rm, sizeof(int32_t) * nn);
  if (dump->seg_count) for (int i = 0; i < E; ++i) dump->seg_count[i] = brs[i].m;
  if (dump->seg_dwell && dump->seg_state) {
    for (int i = 0; i < E; ++i) {
      int m = brs[i].m < dump->seg_cap ? brs[i].m : dump->seg_cap;
      for (int j = 0; j < m; ++j) {
        dump->seg_dwell[(size_t)i * dump->seg_cap + j] = brs[i].d[j];
        dump->seg_state[(size_t)i * dump->seg_cap + j] = brs[i].s[j];
      }
    }
  }
  if (dump->PL && PL) memcpy(dump->PL, PL, sizeof(double) * pl_len);
}


/* ------------------------------------------------------------------------------------------ */
/* Rate-matrix updates of the Q-updating variants (host-side scalar work, one call per sweep)   */
/* ------------------------------------------------------------------------------------------ */
/* Host draws use the Philox stream (replica word 0xFFFFFFFF, entity 0xFFFFFF00 | update id); Rf_rgamma is R's
 * Ahrens-Dieter generator, which is third-party code outside /root/reference, so the gamma variate is
 * Marsaglia & Tsang (2000) with a Box-Muller normal -- same distribution, different stream. */
typedef struct { rngctx* rc; uint32_t iter; uint32_t ent; uint32_t d; } hstream;
static double hs_u(hstream* h) {
  orc_rng* r = h->rc->r;
  if (r->mode == 2) return r_unif_rand();                                 /* R-stream mode: runif(1) of R's own generator, in call order */
  if (r->mode == 1) return draw_u(h->rc, h->iter, h->ent, h->d++);
  return orc_stream_u(r->seed_lo, r->seed_hi, 0xFFFFFFFFu, h->iter, h->ent, h->d++);
}
static double hs_rgamma(hstream* h, double a, double scale) {
  if (h->rc->r->mode == 2) return r_rgamma(a, scale);                     /* R-stream mode: Rf_rgamma itself (restated above) */
  double boost = 1.0;
  if (a < 1.0) { double u = hs_u(h); boost = pow(u, 1.0 / a); a += 1.0; }
  const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (;;) {
    double u1 = hs_u(h), u2 = hs_u(h);
    double xn = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    double v = 1.0 + c * xn;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = hs_u(h);
    if (log(u) < 0.5 * xn * xn + d - d * v + d * log(v)) return d * v * boost * scale;
  }
}

#define QQ(i, j) Q[(size_t)(i) * n + (j)]
#define TC(idx) stats[(int64_t)(n + (idx)) * stride + it]      /* transitioncounts(idx) = dwelltimes(iteration, n+idx) */
#define SJ(i) stats[(int64_t)(i) * stride + it]                /* sojourntimes(i) */

/* updatel01 / updatel10, src/phylomap.cpp:1189-1253 (two states; `accept` is computed and never tested there) */
static void updatel01(double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h) {
  int n01 = (int)stats[(int64_t)3 * stride + it];
  double t0 = stats[(int64_t)0 * stride + it];
  double newl01 = hs_rgamma(h, prior[0] + n01, 1 / (prior[1] + t0));       /* :1202 */
  if (newl01 > Omega) return;                                             /* :1205 */
  (void)hs_u(h);                                                          /* compare, :1210 (unused) */
  QQ(0, 0) = -newl01; QQ(0, 1) = newl01;                                  /* :1212-1213 */
}
static void updatel10(double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h) {
  int n10 = (int)stats[(int64_t)4 * stride + it];
  double t1 = stats[(int64_t)1 * stride + it];
  double newl10 = hs_rgamma(h, prior[2] + n10, 1 / (prior[3] + t1));       /* :1235 */
  if (newl10 > Omega) return;
  (void)hs_u(h);
  QQ(1, 0) = newl10; QQ(1, 1) = -newl10;                                  /* :1244-1245 */
}

/* updatel01mtNS :2192-2226 / updatel10mtNS :2228-2262 (side = 0 / 1): unlike the single-tree twins these DO test the
 * acceptance ratio -- accept = ((Omega-new)/(Omega-old))^n_ss * exp(t_s (new-old)), capped at 1, against one runif that is
 * drawn only when the proposal is not above Omega; `acceptcompare` (metropolis*hastings, :2176-2190) is never used. */
static void updatelmtNS(int side, double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h) {
  int nss = (int)stats[(int64_t)(side == 0 ? 2 : 5) * stride + it];         /* n00 / n11 */
  int nsd = (int)stats[(int64_t)(side == 0 ? 3 : 4) * stride + it];         /* n01 / n10 */
  double ts = stats[(int64_t)side * stride + it];
  double old = side == 0 ? QQ(0, 1) : QQ(1, 0);
  double nw = hs_rgamma(h, prior[2 * side] + nsd, 1 / (prior[2 * side + 1] + ts));
  if (nw > Omega) return;
  double accept = pow((Omega - nw) / (Omega - old), nss) * exp(ts * (nw - old));
  if (accept > 1) accept = 1;
  double compare = hs_u(h);
  if (accept < compare) return;
  if (side == 0) { QQ(0, 0) = -nw; QQ(0, 1) = nw; } else { QQ(1, 0) = nw; QQ(1, 1) = -nw; }
}

/* parameters as every ks update re-reads them from Q (e.g. :1441-1450) */
static void ks_params(const double* Q, int n, int k, double* lambdas, double* rk, double* lk, double* gm) {
  lambdas[0] = QQ(0, 1); lambdas[1] = QQ(1, 0);
  for (int i = 0; i < k; ++i) rk[i] = QQ(2 * i, 2 * i + 2);
  for (int i = 0; i < k; ++i) lk[i] = QQ(2 * i + 2, 2 * i);
  gm[0] = 1;
  for (int i = 1; i <= k; ++i) gm[i] = QQ(2 * i, 2 * i + 1) / lambdas[0];
}

/* updateksl01 :1435-1505 (side = 0) and updateksl10 :1509-1578 (side = 1) */
/* mt != 0: the multi-tree twins updateksl01mt :2371-2439 / updateksl10mt :2443-2510 -- side 1 reads prior(2), prior(3) yet
 * still subtracts prior(1) for gammatimes (:2471), and the `< 1e-300` guard is absent */
static void updateksl(int side, double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h, int mt) {
  int k = n / 2 - 1, i;
  double lambdas[2], rk[32], lk[32], gm[33];
  ks_params(Q, n, k, lambdas, rk, lk, gm);
  double alphaprime = prior[(mt && side) ? 2 : 0];
  for (i = 0; i <= k; ++i) alphaprime = alphaprime + (side == 0 ? TC(2 * i * n + 2 * i + 1) : TC((2 * i + 1) * n + 2 * i));
  double betaprime = prior[(mt && side) ? 3 : 1];
  for (i = 0; i <= k; ++i) betaprime = betaprime + gm[i] * SJ(2 * i + side);
  double newl = hs_rgamma(h, alphaprime, 1 / betaprime);
  double old = lambdas[side];
  double gammatimes = betaprime - prior[1];
  double logaccept = (newl - old) * gammatimes;
  logaccept = logaccept + (side == 0 ? TC(0) : TC(n + 1)) * log((Omega - rk[0] - gm[0] * newl) / (Omega - rk[0] - gm[0] * old));
  for (i = 1; i < k; ++i)
    logaccept = logaccept + (side == 0 ? TC(2 * i * n + 2 * i) : TC((2 * i + 1) * n + 2 * i + 1)) *
                log((Omega - rk[i] - lk[i - 1] - gm[i] * newl) / (Omega - rk[i] - lk[i - 1] - gm[i] * old));
  logaccept = logaccept + (side == 0 ? TC(2 * k * n + 2 * k) : TC((2 * k + 1) * n + 2 * k + 1)) *
              log((Omega - lk[k - 1] - gm[k] * newl) / (Omega - lk[k - 1] - gm[k] * old));
  double compare = hs_u(h);
  if (newl + rk[0] > Omega) return;
  for (i = 1; i < k; ++i) if (gm[i] * newl + rk[i] + lk[i - 1] > Omega) return;
  if (gm[k] * newl + lk[k - 1] > Omega) return;
  if (!mt && newl < 1e-300) return;
  if (logaccept < log(compare)) return;
  int a = side, b = 1 - side;                                              /* row a, column b of each 2x2 block */
  QQ(a, a) = -rk[0] - gm[0] * newl; QQ(a, b) = gm[0] * newl;
  for (i = 1; i < k; ++i) { QQ(2 * i + a, 2 * i + a) = -lk[i - 1] - rk[i] - gm[i] * newl; QQ(2 * i + a, 2 * i + b) = gm[i] * newl; }
  QQ(2 * k + a, 2 * k + a) = -lk[k - 1] - gm[k] * newl; QQ(2 * k + a, 2 * k + b) = gm[k] * newl;
}

/* updaterkappas :1582-1644 */
/* mt != 0: updaterkappasmt :2514-2576 (prior indices two further on, no `< 1e-300` guard) */
static void updaterkappas(double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h, int j, int mt) {
  if (mt) prior += 2;
  int k = n / 2 - 1;
  double lambdas[2], rk[32], lk[32], gm[33];
  ks_params(Q, n, k, lambdas, rk, lk, gm);
  double alphaprime = prior[2] + TC((2 * j) * n + 2 * j + 2) + TC((2 * j + 1) * n + 2 * j + 3);
  double betaprime = prior[3] + SJ(2 * j) + SJ(2 * j + 1);
  double nw = hs_rgamma(h, alphaprime, 1 / betaprime);
  double logaccept = (nw - rk[j]) * (SJ(2 * j) + SJ(2 * j + 1));
  double lkp = (j > 0) ? lk[j - 1] : 0.0;
  if (j == 0) {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - nw - gm[j] * lambdas[0]) / (Omega - rk[j] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - nw - gm[j] * lambdas[1]) / (Omega - rk[j] - gm[j] * lambdas[1]));
  } else {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - lk[j - 1] - nw - gm[j] * lambdas[0]) / (Omega - lk[j - 1] - rk[j] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - lk[j - 1] - nw - gm[j] * lambdas[1]) / (Omega - lk[j - 1] - rk[j] - gm[j] * lambdas[1]));
  }
  double compare = hs_u(h);
  if (j == 0) { if (nw + gm[j] * lambdas[0] > Omega) return; if (nw + gm[j] * lambdas[1] > Omega) return; }
  else { if (nw + gm[j] * lambdas[0] + lk[j - 1] > Omega) return; if (nw + gm[j] * lambdas[1] + lk[j - 1] > Omega) return; }
  if (!mt && nw < 1e-300) return;
  if (logaccept < log(compare)) return;
  QQ(2 * j, 2 * j + 2) = nw; QQ(2 * j + 1, 2 * j + 3) = nw;
  if (j == 0) { QQ(0, 0) = -nw - gm[j] * lambdas[0]; QQ(1, 1) = -nw - gm[j] * lambdas[1]; }
  else { QQ(2 * j, 2 * j) = -nw - lkp - gm[j] * lambdas[0]; QQ(2 * j + 1, 2 * j + 1) = -nw - lkp - gm[j] * lambdas[1]; }
}

/* updatelkappas :1648-1710 */
/* mt != 0: updatelkappasmt :2580-2639 (prior indices two further on, no `< 1e-300` guard) */
static void updatelkappas(double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h, int j, int mt) {
  if (mt) prior += 2;
  int k = n / 2 - 1;
  double lambdas[2], rk[32], lk[32], gm[33];
  ks_params(Q, n, k, lambdas, rk, lk, gm);
  double alphaprime = prior[2] + TC((2 * j) * n + 2 * j - 2) + TC((2 * j + 1) * n + 2 * j - 1);
  double betaprime = prior[3] + SJ(2 * j) + SJ(2 * j + 1);
  double nw = hs_rgamma(h, alphaprime, 1 / betaprime);
  double logaccept = (nw - lk[j - 1]) * (SJ(2 * j) + SJ(2 * j + 1));
  if (j == k) {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - nw - gm[j] * lambdas[0]) / (Omega - lk[j - 1] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - nw - gm[j] * lambdas[1]) / (Omega - lk[j - 1] - gm[j] * lambdas[1]));
  } else {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - rk[j] - nw - gm[j] * lambdas[0]) / (Omega - rk[j] - lk[j - 1] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - rk[j] - nw - gm[j] * lambdas[1]) / (Omega - rk[j] - lk[j - 1] - gm[j] * lambdas[1]));
  }
  double compare = hs_u(h);
  if (j == k) { if (nw + gm[j] * lambdas[0] > Omega) return; if (nw + gm[j] * lambdas[1] > Omega) return; }
  else { if (nw + gm[j] * lambdas[0] + rk[j] > Omega) return; if (nw + gm[j] * lambdas[1] + rk[j] > Omega) return; }
  if (!mt && nw < 1e-300) return;
  if (logaccept < log(compare)) return;
  QQ(2 * j, 2 * j - 2) = nw; QQ(2 * j + 1, 2 * j - 1) = nw;
  if (j == k) { QQ(2 * j, 2 * j) = -nw - gm[j] * lambdas[0]; QQ(2 * j + 1, 2 * j + 1) = -nw - gm[j] * lambdas[1]; }
  else { QQ(2 * j, 2 * j) = -nw - rk[j] - gm[j] * lambdas[0]; QQ(2 * j + 1, 2 * j + 1) = -nw - rk[j] - gm[j] * lambdas[1]; }
}

/* updategammas :1714-1785 */
/* mt != 0: updategammasmt :2643-2705 (prior indices two further on, no `< 1e-300` guard) */
static void updategammas(double* Q, int n, double Omega, const double* prior, const double* stats, int64_t stride, int it, hstream* h, int j, int mt) {
  if (mt) prior += 2;
  int k = n / 2 - 1;
  double lambdas[2], rk[32], lk[32], gm[33];
  ks_params(Q, n, k, lambdas, rk, lk, gm);
  double alphaprime = prior[4] + TC((2 * j) * n + 2 * j + 1) + TC((2 * j + 1) * n + 2 * j);
  double betaprime = prior[5] + SJ(2 * j) * lambdas[0] + SJ(2 * j + 1) * lambdas[1];
  double nw = hs_rgamma(h, alphaprime, 1 / betaprime);
  double logaccept = (nw - gm[j]) * (SJ(2 * j) * lambdas[0] + SJ(2 * j + 1) * lambdas[1]);
  if (j == k) {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - lk[j - 1] - nw * lambdas[0]) / (Omega - lk[j - 1] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - lk[j - 1] - nw * lambdas[1]) / (Omega - lk[j - 1] - gm[j] * lambdas[1]));
  } else {
    logaccept = logaccept + TC((2 * j) * n + 2 * j) * log((Omega - lk[j - 1] - rk[j] - nw * lambdas[0]) / (Omega - rk[j] - lk[j - 1] - gm[j] * lambdas[0]));
    logaccept = logaccept + TC((2 * j + 1) * n + 2 * j + 1) * log((Omega - lk[j - 1] - rk[j] - nw * lambdas[1]) / (Omega - rk[j] - lk[j - 1] - gm[j] * lambdas[1]));
  }
  double compare = hs_u(h);
  if (j == k) { if (lk[j - 1] + nw * lambdas[0] > Omega) return; if (lk[j - 1] + nw * lambdas[1] > Omega) return; }
  else { if (lk[j - 1] + nw * lambdas[0] + rk[j] > Omega) return; if (lk[j - 1] + nw * lambdas[1] + rk[j] > Omega) return; }
  if (!mt && nw < 1e-300) return;
  if (logaccept < log(compare)) return;
  QQ(2 * j, 2 * j + 1) = nw * lambdas[0]; QQ(2 * j + 1, 2 * j) = nw * lambdas[1];
  if (j == k) { QQ(2 * j, 2 * j) = -lk[j - 1] - nw * lambdas[0]; QQ(2 * j + 1, 2 * j + 1) = -lk[j - 1] - nw * lambdas[1]; }
  else { QQ(2 * j, 2 * j) = -lk[j - 1] - rk[j] - nw * lambdas[0]; QQ(2 * j + 1, 2 * j + 1) = -lk[j - 1] - rk[j] - nw * lambdas[1]; }
}
#undef TC
#undef SJ

/* One tree with its chain state: what maketreelistMCMC* unpacks from the R list x (:895-915) and carries across sweeps */
typedef struct {
  const orc_tree* x; int E, T, Nnode; const int32_t *edge1, *edge2;
  Branch* brs; double* PL; size_t pl_len; int32_t *rm, *bl, *eoc; double* w;
  double* scratch; size_t scratch_len; Branch tmp;
} treechain;

static void tips_into_PL(const orc_tree* x, int n, int masks, double* PL) {
  if (!masks) for (int i = 0; i < x->n_tips; ++i) PL[(size_t)i * n + (x->states[i] - 1)] = 1.0;   /* :914 */
  else for (int i = 0; i < x->n_tips; ++i)                    /* only the binary trait is observed :1838-1845 */
    for (int j = (x->states[i] % 2 == 0) ? 1 : 0; j < n; j += 2) PL[(size_t)i * n + j] = 1.0;
}

static int chain_init(treechain* c, const orc_tree* x, int n, int masks) {
  memset(c, 0, sizeof *c);
  int e = check_tree(x); if (e) return e;
  c->x = x; c->E = x->n_edge; c->T = x->n_tips; c->Nnode = x->n_node;
  c->edge1 = x->edge; c->edge2 = x->edge + c->E;
  c->brs = (Branch*)calloc(c->E, sizeof(Branch));
  for (int i = 0; i < c->E; ++i) {
    int o = x->map_off[i], m = x->map_off[i + 1] - o;
    if (m < 1) { e = ORC_ERR_BAD_INPUT; m = 0; }
    makeabranch(&c->brs[i], x->maps + o, x->mapnames + o, m);    /* :901 */
  }
  c->pl_len = (size_t)(2 * c->Nnode + 1) * n;
  c->PL = (double*)calloc(c->pl_len, sizeof(double));
  tips_into_PL(x, n, masks, c->PL);
  c->rm = (int32_t*)calloc(2 * c->T - 1, sizeof(int32_t));
  c->bl = (int32_t*)calloc(c->E, sizeof(int32_t));
  c->eoc = (int32_t*)calloc(2 * c->T - 1, sizeof(int32_t));
  for (int i = 0; i < c->E; ++i) c->eoc[c->edge2[i] - 1] = i;
  c->w = (double*)malloc(sizeof(double) * 3 * n);
  return e;
}

static void chain_free(treechain* c) {
  if (c->brs) for (int i = 0; i < c->E; ++i) { free(c->brs[i].d); free(c->brs[i].s); }
  free(c->tmp.d); free(c->tmp.s); free(c->scratch); free(c->w); free(c->eoc); free(c->bl); free(c->rm); free(c->PL); free(c->brs);
  memset(c, 0, sizeof *c);
}

/* treesample :766-785 / treesampleks :1422-1432 / treesamplemtNS :2157-2165: nodes, then every branch, then dwell times.
 * stats(it, .) is column-major with `stride` rows; returns the root state in *rootst (ks / bf sweeps record it). */
static void chain_sweep(treechain* c, int n, const double* pid, const double* Bc, const double* B2, double Omega, const double* Qd,
                        const int32_t* nen, const int32_t* nodelist, int32_t root, int normalise, int ks, int faithful_search,
                        rngctx* rc, double* stats, int64_t stride, int it, double* rootst) {
  sampleinternalnodesMCMC(c->brs, c->E, c->PL, pid, Bc, root, nodelist, c->Nnode - 1, nen, c->edge1, c->edge2, c->Nnode,
                          c->x->states, c->T, normalise, n, faithful_search, c->eoc, rc, (uint32_t)it, c->rm, c->bl, c->w, ks, rootst);
  updatenodestates(c->brs, c->edge1, c->edge2, c->E, c->rm);                                /* :779 / :1426 */
  for (int i = 0; i < c->E; ++i)                                                            /* :781 / :1428 */
    sampleabranch(&c->brs[i], Bc, B2, Omega, Qd, ks ? -n : n, stats, stride, it, rc, (uint32_t)i, &c->scratch, &c->scratch_len, &c->tmp);
  for (int i = 0; i < c->E; ++i) updatedwelltimes(it, &c->brs[i], stats, stride);           /* :782 */
}

/* recordQ :1181-1185 / recordQks :1789-1798 / recordQmtNS :2169-2173 / recordQksmt :2709-2718 */
static void record_Q(const double* Q, int n, int kk, double* stats, int64_t stride, int it) {
  int base = n + n * n;
  stats[(int64_t)base * stride + it] = QQ(0, 1);
  stats[(int64_t)(base + 1) * stride + it] = QQ(1, 0);
  for (int i = 0; i < kk; ++i) {
    stats[(int64_t)(base + 2 + i) * stride + it] = QQ(2 * i, 2 * i + 2);
    stats[(int64_t)(base + 2 + kk + i) * stride + it] = QQ(2 * i + 2, 2 * i);
    stats[(int64_t)(base + 2 + 2 * kk + i) * stride + it] = QQ(2 * (i + 1), 2 * (i + 1) + 1) / QQ(0, 1);
  }
}

/* B2 = I + Q/Omega entry by entry, as the updates write it (e.g. :1214-1217) */
static void model_from_Q(const double* Q, int n, double Omega, double* B2, double* Bc, double* Qd) {
  for (int i = 0; i < n; ++i) {
    Qd[i] = QQ(i, i);
    for (int j = 0; j < n; ++j) B2[i * n + j] = (i == j) ? 1 + QQ(i, j) / Omega : QQ(i, j) / Omega;
  }
  memcpy(Bc, B2, sizeof(double) * n * n);
}

static int mcmc_driver(const orc_tree* x, int n, const double* Q_cm, const double* pid,
                         const double* B_cm, double Omega, const int32_t* nen,
                         const int32_t* nodelist, int32_t root, int32_t N, int variant,
                         int faithful_search, orc_rng* rng, double* out, orc_dump* dump, const double* prior, int dic) {
  int e = check_tree(x); if (e) return e;
  if (dic && (!prior || !x->edge_length)) return ORC_ERR_BAD_INPUT;
  if (n < 2 || N < 0) return ORC_ERR_BAD_INPUT;
  /* variant | ORC_FORCE_NORMALISE: divide every internal partial-likelihood row by its sum (what makePLrcpp_bigtree :525 does)
   * in a driver that does not -- NOT in the reference: sumstatMCMC / SPARSEsumstatMCMC underflow on trees of thousands of tips
   * (man/sumstatMCMC_bigtree.Rd:17); node draws do not depend on a row's scale, so it is the same sampler in exact arithmetic */
  const int force_norm = (variant & ORC_FORCE_NORMALISE) != 0;
  variant &= ~ORC_FORCE_NORMALISE;
  int E = x->n_edge, T = x->n_tips, Nnode = x->n_node;
  const int32_t* edge2 = x->edge + E; const int32_t* edge1 = x->edge;
  const int ks = (variant == ORC_MCMC_KS) ? 1 : (variant == ORC_MCMC_BF) ? 2 : 0;   /* 1: ks sweep, 2: bf sweep */
  if (ks == 1 && (n & 1)) return ORC_ERR_BAD_INPUT;           /* hidden-rates structure: n = 2k+2 (:1820) */
  /* The bf SWEEP (treesamplebf :1169-1179, sampleinternalnodesMCMCbf :1091-1163, sampleabranchbf :1031-1074, shortenerbf
   * :997-1028) is written for any n; two states are hard-wired around it only: the N x 9 matrix (:1293), root column 8 (:1129)
   * and updatel01 / updatel10 (:1187-1253).  With Q held fixed (prior == NULL) any n is taken: the root state goes to column
   * n + n*n + 2, which is the reference's 8 at n = 2. */
  if (ks == 2 && n != 2 && prior) return ORC_ERR_BAD_INPUT;
  if (prior && (!ks || (ks == 1 && n < 4) || n > 64)) return ORC_ERR_BAD_INPUT;
  const int kk = (ks == 1) ? n / 2 - 1 : 0;
  int cols = ks ? n + n * n + 2 + 3 * kk + 1 + (dic ? 1 : 0) : n + n * (n - 1);
  double* Q = (double*)malloc(sizeof(double) * n * n);        /* row-major working copy; the updates edit it */
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) QQ(i, j) = Q_cm[i + (size_t)j * n];
  rngctx rc = { rng, 0, 0 };
  g_rstream = (rng->mode == 2);
  if (g_rstream) r_set_seed(rng->seed_lo);

  double* B2 = (double*)malloc(sizeof(double) * n * n);       /* row-major copies */
  double* Bc = (double*)malloc(sizeof(double) * n * n);
  double* Qd = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) { Qd[i] = Q_cm[i + (size_t)i * n]; for (int j = 0; j < n; ++j) B2[i * n + j] = B_cm[i + (size_t)j * n]; }
  if (variant == ORC_MCMC_SPARSE) orc_matTospmat(B2, n, Bc);  /* :848 */
  else memcpy(Bc, B2, sizeof(double) * n * n);
  int normalise = (variant == ORC_MCMC_BIGTREE) || ks || force_norm;        /* makePLnormalized :1085 */

  treechain ch;
  e = chain_init(&ch, x, n, ks == 1);
  const size_t pl_len = ch.pl_len;
  double* w = ch.w;
  memset(out, 0, sizeof(double) * (size_t)N * cols);          /* :926 */

  if (!e) for (int it = 0; it < N; ++it) {
    double rootst = 0.0;
    if (ks) record_Q(Q, n, kk, out, N, it);
    chain_sweep(&ch, n, pid, Bc, B2, Omega, Qd, nen, nodelist, root, normalise, ks, faithful_search, &rc, out, N, it, &rootst);
    if (ks) out[(int64_t)(n + n * n + 2 + 3 * kk) * N + it] = rootst;      /* root column; the DIC drivers add log p(y|Q) after it */
    if (dic) {
      /* log p(y|Q) by matrix exponentiation, maketreelistMCMC2sDICt :3239-3251 / ksDICt :3379-3391:
         P_b = expmat(Q t_b); PPmakePLD / PPmakePLksD (:3158-3178, :3268-3297) prune with row normalisation and sum the
         log scale factors in nen order; log(sum_j PL[root,j] pid_j) + S goes into the last column */
      size_t nn2 = (size_t)n * n;
      double* Pm = (double*)malloc(sizeof(double) * nn2 * (E + 1));
      double* A = Pm + nn2 * E;
      for (int b = 0; b < E; ++b) {
        for (size_t q = 0; q < nn2; ++q) A[q] = Q[q] * x->edge_length[b];
        if (orc_expmat_pade(A, n, Pm + nn2 * b)) rc.err |= ORC_ERR_BAD_INPUT;
      }
      double* PL2 = (double*)calloc(pl_len, sizeof(double));
      tips_into_PL(x, n, ks == 1, PL2);
      double S = 0;
      double* va = w; double* vb = w + n;
      for (int i = 0; i < Nnode; ++i) {
        int ea = nen[2 * i] - 1, eb = nen[2 * i + 1] - 1;
        matvec(Pm + nn2 * ea, PL2 + (size_t)(edge2[ea] - 1) * n, va, n);
        matvec(Pm + nn2 * eb, PL2 + (size_t)(edge2[eb] - 1) * n, vb, n);
        double* row = PL2 + (size_t)(edge1[ea] - 1) * n;
        for (int c = 0; c < n; ++c) row[c] = va[c] * vb[c];
        double sm = row[0];
        for (int c = 1; c < n; ++c) sm += row[c];
        S = S + orc_log(sm);
        for (int c = 0; c < n; ++c) row[c] = row[c] / sm;
      }
      double X = 0;
      for (int j = 0; j < n; ++j) X = X + PL2[(size_t)(root - 1) * n + j] * pid[j];
      out[(int64_t)(cols - 1) * N + it] = orc_log(X) + S;
      free(PL2); free(Pm);
    }
    if (prior) {                                              /* maketreelistMCMCbf :1299-1300 / maketreelistMCMCks :1862-1866 */
      hstream h = { &rc, (uint32_t)it, 0, 0 };
#define HS(id) (h.ent = 0xFFFFFF00u | (uint32_t)(id), h.d = 0, &h)
      if (ks == 2) {
        updatel01(Q, n, Omega, prior, out, N, it, HS(0));
        updatel10(Q, n, Omega, prior, out, N, it, HS(1));
      } else {
        updateksl(0, Q, n, Omega, prior, out, N, it, HS(0), 0);
        updateksl(1, Q, n, Omega, prior, out, N, it, HS(1), 0);
        for (int j = 0; j < kk; ++j) updaterkappas(Q, n, Omega, prior, out, N, it, HS(2 + j), j, 0);
        for (int j = 1; j <= kk; ++j) updatelkappas(Q, n, Omega, prior, out, N, it, HS(2 + kk + j), j, 0);
        for (int j = 1; j <= kk; ++j) updategammas(Q, n, Omega, prior, out, N, it, HS(2 + 2 * kk + j), j, 0);
      }
#undef HS
      model_from_Q(Q, n, Omega, B2, Bc, Qd);
    }
  }
  free(Q);
  if (!e) fill_dump(dump, ch.brs, E, ch.rm, 2 * T - 1, ch.PL, pl_len);
  chain_free(&ch);
  free(Qd); free(Bc); free(B2);
  return e | rc.err;
}

/* maketreelistMCMCmt :2267-2365 (variant ORC_MCMC_MT, two states, prior = 4 numbers) and maketreelistMCMCksmt :2722-2844
 * (variant ORC_MCMC_KSMT, n = 2k+2 states, prior = 8 numbers).  Every iteration sweeps EVERY tree of the list with the current
 * Q (treesamplemtNS :2157-2165: the ks-style sweep -- B4 chain for the nodes, tips re-drawn under their PL rows, virtual
 * jumps counted -- but with the UN-normalised pruning makePLrcppmt :1938-1950 and no root column), then draws one tree
 * uniformly (sampleOnce over unit weights, :2347-2348), copies that tree's row and its 0-based index into the output
 * (:2349-2350) and updates Q from that row with the mt twins of the updates.
 * nen_m: treecount x 2*Nnode, nodelist_m: treecount x (Nnode-1), row-major (one row per tree).
 * Random numbers: tree j sweeps on Philox replica word rng->replica + 64 j (one 64-lane tile per tree in the device layout);
 * the tree choice is host stream 0xFD of the iteration; the updates use the same stream ids as the single-tree drivers. */
int orc_maketreelistMCMCmt(const orc_tree* const* xs, int treecount, int n, const double* Q_cm, const double* pid,
                           const double* B_cm, double Omega, const int32_t* nen_m, const int32_t* nodelist_m,
                           const int32_t* roots, int32_t N, int variant, const double* prior, int faithful_search,
                           orc_rng* rng, double* out) {
  if (treecount < 1 || !xs || !prior || n < 2 || N < 0 || rng->mode != 0) return ORC_ERR_BAD_INPUT;
  if (variant != ORC_MCMC_MT && variant != ORC_MCMC_KSMT) return ORC_ERR_BAD_INPUT;
  const int mtks = (variant == ORC_MCMC_KSMT);
  if (mtks ? ((n & 1) || n < 4 || n > 64) : (n != 2)) return ORC_ERR_BAD_INPUT;   /* recordQmtNS hard-wires columns 6, 7 */
  const int kk = mtks ? n / 2 - 1 : 0;
  const int rowlen = n + n * n + 2 + 3 * kk, cols = rowlen + 1;
  const int Nnode = xs[0]->n_node;
  for (int j = 0; j < treecount; ++j)
    if (xs[j]->n_node != Nnode || xs[j]->n_edge != xs[0]->n_edge || xs[j]->n_tips != xs[0]->n_tips) return ORC_ERR_BAD_INPUT;
  g_rstream = 0;
  int e = 0;
  double* Q = (double*)malloc(sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) QQ(i, j) = Q_cm[i + (size_t)j * n];
  double* B2 = (double*)malloc(sizeof(double) * n * n);
  double* Bc = (double*)malloc(sizeof(double) * n * n);
  double* Qd = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) { Qd[i] = Q_cm[i + (size_t)i * n]; for (int j = 0; j < n; ++j) B2[i * n + j] = B_cm[i + (size_t)j * n]; }
  memcpy(Bc, B2, sizeof(double) * n * n);
  treechain* ch = (treechain*)calloc(treecount, sizeof(treechain));
  orc_rng* rngs = (orc_rng*)calloc(treecount, sizeof(orc_rng));
  rngctx* rcs = (rngctx*)calloc(treecount, sizeof(rngctx));
  double* rows = (double*)calloc((size_t)treecount * rowlen + 1, sizeof(double));   /* DwellTimes[j] :2335-2336 (+1: unused root slot) */
  double* weights = (double*)malloc(sizeof(double) * treecount);
  for (int j = 0; j < treecount; ++j) {
    e |= chain_init(&ch[j], xs[j], n, mtks);
    rngs[j] = *rng; rngs[j].replica = rng->replica + 64u * (uint32_t)j;
    rcs[j].r = &rngs[j]; rcs[j].err = 0; rcs[j].iter_base = 0;
    weights[j] = 1.0;
  }
  rngctx rc0 = { rng, 0, 0 };
  memset(out, 0, sizeof(double) * (size_t)N * cols);
  if (!e) for (int it = 0; it < N; ++it) {
    for (int j = 0; j < treecount; ++j) {
      double* row = rows + (size_t)j * rowlen;
      double rootst = 0.0;
      memset(row, 0, sizeof(double) * rowlen);                                               /* :2342 */
      record_Q(Q, n, kk, row, 1, 0);                                                           /* :2343 */
      rcs[j].iter_base = (uint32_t)it;                     /* the row is "iteration 0" of a one-row matrix; Philox sees `it` */
      chain_sweep(&ch[j], n, pid, Bc, B2, Omega, Qd, nen_m + (size_t)j * 2 * Nnode, nodelist_m + (size_t)j * (Nnode - 1),
                  roots[j], 0, mtks ? 1 : 2, faithful_search, &rcs[j], row, 1, 0, &rootst);
    }
    hstream h = { &rc0, (uint32_t)it, 0, 0 };
#define HS(id) (h.ent = 0xFFFFFF00u | (uint32_t)(id), h.d = 0, &h)
    double wt = hs_u(HS(0xFD));                                                                /* :2347 */
    int pick = sampleOnce(weights, treecount, wt, &rc0.err);                                   /* :2348 */
    if (pick >= treecount) { e |= ORC_ERR_SAMPLEONCE; break; }
    const double* row = rows + (size_t)pick * rowlen;
    for (int c = 0; c < rowlen; ++c) out[(int64_t)c * N + it] = row[c];                        /* :2349 */
    out[(int64_t)rowlen * N + it] = pick;                                                      /* :2350 */
    if (!mtks) {
      updatelmtNS(0, Q, n, Omega, prior, row, 1, 0, HS(0));                                    /* :2351-2352 */
      updatelmtNS(1, Q, n, Omega, prior, row, 1, 0, HS(1));
    } else {                                                                                   /* :2828-2832 */
      updateksl(0, Q, n, Omega, prior, row, 1, 0, HS(0), 1);
      updateksl(1, Q, n, Omega, prior, row, 1, 0, HS(1), 1);
      for (int j = 0; j < kk; ++j) updaterkappas(Q, n, Omega, prior, row, 1, 0, HS(2 + j), j, 1);
      for (int j = 1; j <= kk; ++j) updatelkappas(Q, n, Omega, prior, row, 1, 0, HS(2 + kk + j), j, 1);
      for (int j = 1; j <= kk; ++j) updategammas(Q, n, Omega, prior, row, 1, 0, HS(2 + 2 * kk + j), j, 1);
    }
#undef HS
    model_from_Q(Q, n, Omega, B2, Bc, Qd);
  }
  for (int j = 0; j < treecount; ++j) { e |= rcs[j].err; chain_free(&ch[j]); }
  e |= rc0.err;
  free(weights); free(rows); free(rcs); free(rngs); free(ch); free(Qd); free(Bc); free(B2); free(Q);
  return e;
}

#undef QQ

int orc_maketreelistMCMC(const orc_tree* x, int n, const double* Q_cm, const double* pid, const double* B_cm, double Omega,
                         const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, int variant,
                         int faithful_search, orc_rng* rng, double* out, orc_dump* dump) {
  return mcmc_driver(x, n, Q_cm, pid, B_cm, Omega, nen, nodelist, root, N, variant, faithful_search, rng, out, dump, NULL, 0);
}

/* maketreelistMCMCbf :1258-1305 (variant ORC_MCMC_BF, prior = 4 numbers) and maketreelistMCMCks :1802-1872
 * (variant ORC_MCMC_KS, prior = 6 numbers): the sweep followed by the rate-matrix updates, every iteration. */
int orc_maketreelistMCMC_qupdate(const orc_tree* x, int n, const double* Q_cm, const double* pid, const double* B_cm,
                                 double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                 int variant, const double* prior, int faithful_search, orc_rng* rng, double* out,
                                 orc_dump* dump) {
  if (!prior) return ORC_ERR_BAD_INPUT;
  /* variant | 16: the DIC drivers maketreelistMCMC2sDICt :3183-3264 / maketreelistMCMCksDICt :3300-3403 (one more column) */
  return mcmc_driver(x, n, Q_cm, pid, B_cm, Omega, nen, nodelist, root, N, variant & 15, faithful_search, rng, out, dump, prior,
                     (variant & 16) != 0);
}

/* one iteration's updates applied to a row-major Q given a statistics row (test entry point) */
int orc_qupdate_apply(int variant, int n, double* Q_rm, double Omega, const double* prior, const double* row,
                      uint32_t seed_lo, uint32_t seed_hi, uint32_t iter) {
  orc_rng r; memset(&r, 0, sizeof r); r.mode = 0; r.seed_lo = seed_lo; r.seed_hi = seed_hi;
  rngctx rc = { &r, 0, 0 };
  hstream h = { &rc, iter, 0, 0 };
  const int kk = n / 2 - 1;
#define HS(id) (h.ent = 0xFFFFFF00u | (uint32_t)(id), h.d = 0, &h)
  if (variant == ORC_MCMC_BF || variant == ORC_MCMC_MT) {
    if (n != 2) return ORC_ERR_BAD_INPUT;
    if (variant == ORC_MCMC_MT) {
      updatelmtNS(0, Q_rm, n, Omega, prior, row, 1, 0, HS(0));
      updatelmtNS(1, Q_rm, n, Omega, prior, row, 1, 0, HS(1));
    } else {
      updatel01(Q_rm, n, Omega, prior, row, 1, 0, HS(0));
      updatel10(Q_rm, n, Omega, prior, row, 1, 0, HS(1));
    }
  } else if (variant == ORC_MCMC_KS || variant == ORC_MCMC_KSMT) {
    const int mt = (variant == ORC_MCMC_KSMT);
    if (n < 4 || (n & 1) || n > 64) return ORC_ERR_BAD_INPUT;
    updateksl(0, Q_rm, n, Omega, prior, row, 1, 0, HS(0), mt);
    updateksl(1, Q_rm, n, Omega, prior, row, 1, 0, HS(1), mt);
    for (int j = 0; j < kk; ++j) updaterkappas(Q_rm, n, Omega, prior, row, 1, 0, HS(2 + j), j, mt);
    for (int j = 1; j <= kk; ++j) updatelkappas(Q_rm, n, Omega, prior, row, 1, 0, HS(2 + kk + j), j, mt);
    for (int j = 1; j <= kk; ++j) updategammas(Q_rm, n, Omega, prior, row, 1, 0, HS(2 + 2 * kk + j), j, mt);
  } else return ORC_ERR_BAD_INPUT;
#undef HS
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* EXP path                                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* matexp :2964-2968 followed by abs() (:2980, :3042).  D is diagonal, so (left*D)[i][k] = L[i][k]*exp(d_k t)
 * (the other products are exact zeros), then the second GEMM sums left to right. */
void orc_matexp(const double* L, const double* R, const double* dvals, int n, double t, double* P) {
  double ex[256];
  double* e = (n <= 256) ? ex : (double*)malloc(sizeof(double) * n);
  for (int k = 0; k < n; ++k) e[k] = orc_exp(dvals[k] * t);                  /* :2966 */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double acc = (L[i * n] * e[0]) * R[j];
      for (int k = 1; k < n; ++k) acc += (L[i * n + k] * e[k]) * R[k * n + j];
      P[i * n + j] = fabs(acc);
    }
  if (e != ex) free(e);
}

/* The same product with a k-ordered fused multiply-add chain per entry: what an MFMA f64 16x16x4 pipeline computes if,
 * like the f32 forms, it is bit-for-bit an fma chain.  Pins the matrix-core variant of K1 (phm_expm_eigen_mfma). */
void orc_matexp_fma(const double* L, const double* R, const double* dvals, int n, double t, double* P) {
  double* e = (double*)malloc(sizeof(double) * n);
  for (int k = 0; k < n; ++k) e[k] = orc_exp(dvals[k] * t);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc = fma(L[i * n + k] * e[k], R[k * n + j], acc);
      P[i * n + j] = fabs(acc);
    }
  free(e);
}

/* arma::expmat (call sites src/phylomap.cpp:3226,3243,3359,3383): Pade approximant of degree 6 with
 * scaling 2^s and s squarings.  Armadillo is not under /root/reference; this restates its published
 * algorithm: s from frexp(log2(||A||_inf)), E = sum c_i A^i, D = sum (-1)^i c_i A^i, solve(D,E)
 * by LU with partial pivoting, then s squarings. */
static void gemm_rm(const double* A, const double* B, double* C, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double acc = A[i * n] * B[j];
      for (int k = 1; k < n; ++k) acc += A[i * n + k] * B[k * n + j];
      C[i * n + j] = acc;
    }
}
int orc_expmat_pade(const double* A_in, int n, double* out) {
  size_t nn = (size_t)n * n;
  double* A = (double*)malloc(sizeof(double) * nn * 5);
  double *Em = A + nn, *Dm = A + 2 * nn, *X = A + 3 * nn, *T2 = A + 4 * nn;
  double norm = 0.0;
  for (int i = 0; i < n; ++i) { double r = 0.0; for (int j = 0; j < n; ++j) r += fabs(A_in[i * n + j]); if (r > norm) norm = r; }
  double log2v = (norm > 0.0) ? log2(norm) : 0.0;
  int ex = 0; (void)frexp(log2v, &ex);
  int s = ex + 1; if (s < 0) s = 0;
  double sc = ldexp(1.0, s);
  for (size_t i = 0; i < nn; ++i) A[i] = A_in[i] / sc;
  double c = 0.5;
  for (size_t i = 0; i < nn; ++i) { Em[i] = c * A[i]; Dm[i] = -(c * A[i]); X[i] = A[i]; }
  for (int i = 0; i < n; ++i) { Em[i * n + i] += 1.0; Dm[i * n + i] += 1.0; }
  int positive = 1;
  const int NP = 6;
  for (int i = 2; i <= NP; ++i) {
    c = c * (double)(NP - i + 1) / (double)(i * (2 * NP - i + 1));
    gemm_rm(A, X, T2, n); memcpy(X, T2, sizeof(double) * nn);
    for (size_t k = 0; k < nn; ++k) { Em[k] += c * X[k]; if (positive) Dm[k] += c * X[k]; else Dm[k] -= c * X[k]; }
    positive = !positive;
  }
  /* solve D * out = E : Gaussian elimination, partial pivoting */
  int err = 0;
  for (int col = 0; col < n; ++col) {
    int piv = col; double best = fabs(Dm[col * n + col]);
    for (int r = col + 1; r < n; ++r) { double v = fabs(Dm[r * n + col]); if (v > best) { best = v; piv = r; } }
    if (!(best > 0.0)) { err = ORC_ERR_BAD_INPUT; break; }
    if (piv != col) for (int k = 0; k < n; ++k) {
      double t = Dm[col * n + k]; Dm[col * n + k] = Dm[piv * n + k]; Dm[piv * n + k] = t;
      t = Em[col * n + k]; Em[col * n + k] = Em[piv * n + k]; Em[piv * n + k] = t;
    }
    for (int r = col + 1; r < n; ++r) {
      double f = Dm[r * n + col] / Dm[col * n + col];
      for (int k = col; k < n; ++k) Dm[r * n + k] -= f * Dm[col * n + k];
      for (int k = 0; k < n; ++k) Em[r * n + k] -= f * Em[col * n + k];
    }
  }
  if (!err) {
    for (int r = n - 1; r >= 0; --r)
      for (int k = 0; k < n; ++k) {
        double acc = Em[r * n + k];
        for (int j = r + 1; j < n; ++j) acc -= Dm[r * n + j] * X[j * n + k];   /* X reused as the solution */
        X[r * n + k] = acc / Dm[r * n + r];
      }
    for (int i = 0; i < s; ++i) { gemm_rm(X, X, T2, n); memcpy(X, T2, sizeof(double) * nn); }
    memcpy(out, X, sizeof(double) * nn);
  }
  free(A);
  return err;
}

/* sampleinternalnodesEXP :2910-2961 */
static void sampleinternalnodesEXP(const int32_t* edge1, const int32_t* edge2, const int32_t* states, int T,
                                   const double* PL, const double* pid, int root, const int32_t* nodelist, int nll,
                                   const double* P, int n, int faithful_search, const int32_t* edge_of_child,
                                   rngctx* rc, uint32_t iter, int32_t* rm, double* w) {
  for (int i = 0; i < 2 * T - 1; ++i) rm[i] = 0;
  for (int i = 0; i < T; ++i) rm[i] = states[i] - 1;                         /* :2923 */
  double* p = w;
  for (int c = 0; c < n; ++c) p[c] = pid[c] * PL[(size_t)(root - 1) * n + c];    /* :2926 */
  rm[root - 1] = sample_cat(p, n, draw_u(rc, iter, ENT_NODE | (uint32_t)(root - 1), 0), &rc->err);
  for (int i = 0; i < nll; ++i) {
    int cn = nodelist[i] - 1;
    int j = 0;
    if (faithful_search) { while (edge2[j] != nodelist[i]) j++; }            /* :2947 */
    else j = edge_of_child[cn];
    int ps = rm[edge1[j] - 1];
    const double* row = P + (size_t)j * n * n + (size_t)ps * n;
    for (int c = 0; c < n; ++c) p[c] = row[c] * PL[(size_t)cn * n + c];      /* :2953 */
    rm[cn] = sample_cat(p, n, draw_u(rc, iter, ENT_NODE | (uint32_t)cn, 0), &rc->err);
  }
  for (int i = 0; i < 2 * T - 1; ++i) rm[i] = rm[i] + 1;
}

/* dpois(k; lam) by the recurrence p_0 = exp(-lam), p_k = p_{k-1}*lam/k (Philox mode; the HIP kernels do the same).  The
 * reference calls R's Rf_dpois (saddle-point dpois_raw, src/phylomap.cpp:107,128), which is not under /root/reference: R-stream
 * mode uses its restatement r_dpois; the two agree to a few ulp.  */

/* newunifSample, src/phylomap.cpp:93-208.  Returns 1 when the 300-jump cap was hit (:120-125). */
static int newunifSample(int startState, int endState, double elapsedTime, double transProb, Branch* out,
                         double* stats, int64_t stride, int iteration, int n, double poissonRate,
                         const double* B2, rngctx* rc, uint32_t branch_id, double* bpws, double* times, int32_t* dom) {
  uint32_t dr = 0;
  uint32_t ent = ENT_BUNIF | branch_id;
  for (int i = 0; i < n; ++i) bpws[i] = 0.0;
  bpws[endState] = 1.0;                                                      /* :100 */
  double rU = draw_u(rc, (uint32_t)iteration, ent, dr++);                    /* :103 */
  double lam = poissonRate * elapsedTime;
  double pk = g_rstream ? r_dpois(0.0, lam) : orc_exp(-lam);                 /* dpois(0) */
  double cum = 0.0;
  if (startState == endState) cum = pk / transProb;                          /* :107 */
  int notExceed = !(cum > rU);
  int numJumps = 0;
  while (notExceed) {
    numJumps++;
    if (numJumps > 300) return 1;                                            /* :120 */
    matvec(B2, bpws + (size_t)(numJumps - 1) * n, bpws + (size_t)numJumps * n, n);   /* :127 */
    pk = g_rstream ? r_dpois((double)numJumps, lam) : pk * lam / (double)numJumps;
    double nextProb = pk * bpws[(size_t)numJumps * n + startState] / transProb;      /* :128 */
    cum += nextProb;
    if (cum > rU) notExceed = 0;
  }
  out->m = 0;
  if (numJumps == 0 || (numJumps == 1 && startState == endState)) {          /* :138 */
    br_reserve(out, 1);
    out->d[0] = elapsedTime - 0.0; out->s[0] = startState; out->m = 1;
  } else if (numJumps == 1) {                                                /* :144 */
    double tj = elapsedTime * draw_u(rc, (uint32_t)iteration, ent, dr++);    /* :147 */
    br_reserve(out, 2);
    out->d[0] = tj - 0.0; out->s[0] = startState;
    out->d[1] = elapsedTime - tj; out->s[1] = endState; out->m = 2;
  } else {
    for (int i = 0; i < numJumps; ++i) times[i] = elapsedTime * draw_u(rc, (uint32_t)iteration, ent, dr++);   /* :151 */
    for (int i = 1; i < numJumps; ++i) {                                     /* :152 ascending sort */
      double v = times[i]; int j = i - 1;
      while (j >= 0 && times[j] > v) { times[j + 1] = times[j]; --j; }
      times[j + 1] = v;
    }
    dom[0] = startState; dom[numJumps] = endState;
    double* p = bpws + (size_t)(numJumps + 1) * n;
    for (int i = 1; i < numJumps; ++i) {                                     /* :158-160 */
      const double* row = B2 + (size_t)dom[i - 1] * n;
      const double* beta = bpws + (size_t)(numJumps - i) * n;
      for (int c = 0; c < n; ++c) p[c] = row[c] * beta[c];
      dom[i] = sampleOnce(p, n, draw_u(rc, (uint32_t)iteration, ent, dr++), &rc->err);
    }
    /* remove virtual substitutions :163-176, then durations = diff(times) :186-190 */
    br_reserve(out, numJumps + 1);
    double tprev = 0.0; int sprev = startState;
    for (int i = 1; i <= numJumps; ++i) {
      if (dom[i - 1] != dom[i]) {
        out->d[out->m] = times[i - 1] - tprev; out->s[out->m] = sprev; out->m++;
        tprev = times[i - 1]; sprev = dom[i];
      }
    }
    out->d[out->m] = elapsedTime - tprev; out->s[out->m] = sprev; out->m++;
  }
  for (int i = 1; i < out->m; ++i) {                                         /* :194-204 */
    int a = out->s[i - 1], b = out->s[i];
    if (a < b) stats[(int64_t)(n + a * (n - 1) + b - 1) * stride + iteration] += 1.0;
    if (a > b) stats[(int64_t)(n + a * (n - 1) + b) * stride + iteration] += 1.0;
  }
  return 0;
}

/* maketreelistEXP :3001-3051 with treesampleEXP :2977-2996 */
int orc_maketreelistEXP(const orc_tree* x, int n, const double* Q_cm, const double* pid,
                        const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                        const double* lefts_cm, const double* rights_cm, const double* d_cm,
                        int faithful_search, int recompute_expm_each_iter,
                        orc_rng* rng, double* out, orc_dump* dump) {
  int e = check_tree(x); if (e) return e;
  if (n < 2 || N < 0 || !x->edge_length) return ORC_ERR_BAD_INPUT;
  int E = x->n_edge, T = x->n_tips, Nnode = x->n_node;
  const int32_t* edge1 = x->edge; const int32_t* edge2 = x->edge + E;
  int cols = n + n * (n - 1);
  size_t nn = (size_t)n * n;
  rngctx rc = { rng, 0, 0 };
  /* R-stream mode: set.seed(seed_lo), then unif_rand consumed in the reference's order -- per iteration the root and the nodes of
   * nodelist through RcppArmadillo::sample (:2934, :2956), then per branch in edge order rU (:103), the jump times runif(k) (:147,
   * :151) and one runif per interior state (:159, sampleOnce: no sort); Rf_dpois by r_dpois */
  g_rstream = (rng->mode == 2);
  if (g_rstream) r_set_seed(rng->seed_lo);

  double* L = (double*)malloc(sizeof(double) * nn * 3);
  double *R = L + nn, *B2 = L + 2 * nn;
  double* dv = (double*)malloc(sizeof(double) * n);
  double minq = Q_cm[0];
  for (int i = 0; i < n; ++i) { double q = Q_cm[i + (size_t)i * n]; if (q < minq) minq = q; dv[i] = d_cm[i + (size_t)i * n]; }
  double poissonRate = -1.0 * minq;                                          /* :3008 */
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    L[i * n + j] = lefts_cm[i + (size_t)j * n]; R[i * n + j] = rights_cm[i + (size_t)j * n];
    B2[i * n + j] = ((i == j) ? 1.0 : 0.0) + Q_cm[i + (size_t)j * n] / poissonRate;   /* :3011 */
  }
  Branch* brs = (Branch*)calloc(E, sizeof(Branch));
  for (int i = 0; i < E; ++i) {
    int o = x->map_off[i], m = x->map_off[i + 1] - o;
    if (m < 1) { e = ORC_ERR_BAD_INPUT; m = 0; }
    makeabranch(&brs[i], x->maps + o, x->mapnames + o, m);
  }
  double* P = (double*)malloc(sizeof(double) * nn * E);
  for (int i = 0; i < E; ++i) orc_matexp(L, R, dv, n, x->edge_length[i], P + nn * i);   /* :3042 */
  size_t pl_len = (size_t)(2 * Nnode + 1) * n;
  double* PL = (double*)calloc(pl_len, sizeof(double));
  for (int i = 0; i < T; ++i) PL[(size_t)i * n + (x->states[i] - 1)] = 1.0;
  double* w = (double*)malloc(sizeof(double) * 2 * n);
  const int rescale = (recompute_expm_each_iter & 2) != 0;                   /* bit 1: rescaled pruning (not in the reference) */
  makePLexp(edge1, edge2, Nnode, PL, nen, P, n, w, rescale);                 /* :3043 */
  int32_t* rm = (int32_t*)calloc(2 * T - 1, sizeof(int32_t));
  int32_t* eoc = (int32_t*)calloc(2 * T - 1, sizeof(int32_t));
  for (int i = 0; i < E; ++i) eoc[edge2[i] - 1] = i;
  double* bpws = (double*)malloc(sizeof(double) * n * 303);
  double* times = (double*)malloc(sizeof(double) * 301);
  int32_t* dom = (int32_t*)malloc(sizeof(int32_t) * 302);
  Branch tmp = { NULL, NULL, 0, 0 };
  memset(out, 0, sizeof(double) * (size_t)N * cols);

  if (!e) for (int it = 0; it < N; ++it) {
    if (recompute_expm_each_iter & 1) {                                      /* :2980-2981 (Q never changes) */
      for (int i = 0; i < E; ++i) orc_matexp(L, R, dv, n, x->edge_length[i], P + nn * i);
      makePLexp(edge1, edge2, Nnode, PL, nen, P, n, w, rescale);
    }
    sampleinternalnodesEXP(edge1, edge2, x->states, T, PL, pid, root, nodelist, Nnode - 1, P, n,
                           faithful_search, eoc, &rc, (uint32_t)it, rm, w);
    updatenodestates(brs, edge1, edge2, E, rm);                              /* :2983 */
    for (int i = 0; i < E; ++i) {                                            /* :2988-2993 */
      int a = rm[edge1[i] - 1] - 1, b = rm[edge2[i] - 1] - 1;
      int capped = newunifSample(a, b, x->edge_length[i], P[nn * i + (size_t)a * n + b], &tmp, out, N, it, n,
                                 poissonRate, B2, &rc, (uint32_t)i, bpws, times, dom);
      if (capped) rc.err |= ORC_ERR_UNIF_CAP;    /* reference keeps the stale branch and carries on */
      else {
        br_reserve(&brs[i], tmp.m);
        memcpy(brs[i].d, tmp.d, sizeof(double) * tmp.m); memcpy(brs[i].s, tmp.s, sizeof(int32_t) * tmp.m);
        brs[i].m = tmp.m;
      }
      updatedwelltimes(it, &brs[i], out, N);
    }
  }
  g_rstream = 0;
  fill_dump(dump, brs, E, rm, 2 * T - 1, PL, pl_len);
  for (int i = 0; i < E; ++i) { free(brs[i].d); free(brs[i].s); }
  free(tmp.d); free(tmp.s); free(dom); free(times); free(bpws); free(eoc); free(rm); free(w); free(PL); free(P);
  free(brs); free(dv); free(L);
  return e | rc.err;
}
