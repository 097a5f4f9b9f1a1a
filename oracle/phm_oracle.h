/*
 * phm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the stochastic-mapping hot path of vnminin/phylomap
 * (src/phylomap.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (phylomap_amd/) never does.
 *
 * PARITY STATUS: bit-level parity with the R package is "parity unpinned" -- the reference ships no tests / golden vectors
 * and cannot be built here (needs R, Rcpp, RcppArmadillo).  What the reference does hold are two known
 * answers, and the oracle reproduces both (tools/squamate_dic/):
 *   - exactly: R/simulate_2_state_tree.R:11 notes "n01 is 21" for set.seed(101) on the shipped squamate tree; the R random
 *     stream of this file (set.seed scrambling, Mersenne-Twister, unif_rand, Ahrens-Dieter exp_rand) driving the restated
 *     tip simulation gives n01 = 21;
 *   - statistically: the published DIC 2538.272 (2-state) / 2536.056 (4-state hidden rates) of
 *     vignettes/Squamate_DIC_model_selection.Rnw:120 -- four seeds of the bf / ks DIC drivers give 2538.30 .. 2538.34
 *     (Monte-Carlo s.e. 0.05) and 2528.9 .. 2537.3.
 * Further pins: (i) Random123 known-answer vectors for Philox4x32-10, (ii) libm for phm_log/phm_exp and 50-digit arithmetic
 * for the exponential variates, (iii) scipy.linalg.expm for the two expm routes, (iv) hand-derived tiny cases and an
 * independent pure-Python restatement (tests/pyref.py), (v) the reference's own validation idea: EXP == MCMC == SPARSE
 * in distribution (vignettes/phylomap_tutorial.Rnw:113-134).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef PHM_ORACLE_H
#define PHM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error bits (returned OR-ed; 0 = ok) */
#define ORC_ERR_ZERO_PROB   1   /* all-zero / non-finite probability vector (RcppArmadillo::sample would throw) */
#define ORC_ERR_UNIF_CAP    2   /* newunifSample exceeded 300 jumps (src/phylomap.cpp:120) */
#define ORC_ERR_BAD_INPUT   4
#define ORC_ERR_TAPE        8   /* scripted RNG tape exhausted */
#define ORC_ERR_SAMPLEONCE 16   /* sampleOnce ran off the end (src/phylomap.cpp:85-89 returns n) */

/* variants of the fixed-Q MCMC driver */
#define ORC_MCMC_PLAIN    0   /* maketreelistMCMC          src/phylomap.cpp:891-935 */
#define ORC_MCMC_BIGTREE  1   /* maketreelistMCMC_bigtree  src/phylomap.cpp:942-986 */
#define ORC_MCMC_SPARSE   2   /* SPARSEmaketreelistMCMC    src/phylomap.cpp:822-870 */
#define ORC_MCMC_KS       3   /* tree sweep of maketreelistMCMCks src/phylomap.cpp:1802-1872 with Q held fixed:
                                 parity tip masks :1838-1845, makePLnormalized :1077-1088, tip re-sampling :1384-1397,
                                 shortenerbf counts :1010-1014, recordQks :1789-1798, root state :1350-1352;
                                 the Gibbs/MH updates of Q (:1862-1866) are NOT run.  out: N x (n+n*n+2+3k+1) */

#define ORC_MCMC_BF       4   /* tree sweep of maketreelistMCMCbf src/phylomap.cpp:1258-1305 (tips observed, shortenerbf counts,
                                 columns l01 l10 root); out: N x (n + n*n + 3) -- N x 9 at the reference's two states; any n
                                 with Q held fixed (the sweep :1169-1179 is n-generic), n = 2 with the rate updates */
#define ORC_MCMC_MT       5   /* maketreelistMCMCmt   src/phylomap.cpp:2267-2365: list of trees, two states, Q updated */
#define ORC_MCMC_KSMT     6   /* maketreelistMCMCksmt src/phylomap.cpp:2722-2844: list of trees, hidden rates */
#define ORC_FORCE_NORMALISE 32 /* OR-ed into PLAIN / SPARSE: rows of the pruning pass divided by their sum (:525), not in the reference */

/* RNG: mode 0 = counter-based Philox4x32-7 streams (the mode the GPU matches bit for bit);
 *      mode 1 = scripted tapes consumed in the reference's draw order (for hand KATs);
 *      mode 2 = "R stream": set.seed(seed_lo) + unif_rand / exp_rand / sorted RcppArmadillo::sample consumed sequentially
 *               in the reference's order (the fixed-Q MCMC variants and sumstatEXP, whose Rf_dpois is restated as dpois_raw with
 *               stirlerr / bd0).  UNVERIFIED: no R here; see tools/r_parity/. */
typedef struct orc_rng {
  int32_t  mode;
  uint32_t seed_lo, seed_hi;   /* Philox key */
  uint32_t replica;            /* Philox counter word 3 */
  const double* tape_u; int64_t n_u; int64_t pos_u;   /* uniforms in (0,1) */
  const double* tape_e; int64_t n_e; int64_t pos_e;   /* standard exponentials */
} orc_rng;

/* flat phylomap tree object `x` (fields read at src/phylomap.cpp:896-910, 3034) */
typedef struct orc_tree {
  int32_t n_tips;              /* length of x$states */
  int32_t n_node;              /* x$Nnode */
  int32_t n_edge;              /* nrow(x$edge) */
  const int32_t* edge;         /* n_edge x 2, COLUMN-major (R layout), 1-based node ids */
  const double*  edge_length;  /* n_edge (EXP only; may be NULL for MCMC) */
  const int32_t* states;       /* n_tips, 1-based */
  const int32_t* map_off;      /* n_edge+1 offsets into maps/mapnames (x$maps, x$mapnames) */
  const double*  maps;         /* dwell times */
  const int32_t* mapnames;     /* 1-based states */
} orc_tree;

/* optional dump of the chain state after the last iteration (any pointer may be NULL) */
typedef struct orc_dump {
  int32_t* node_states;        /* 2*n_tips-1, 1-based (rm at src/phylomap.cpp:659) */
  int32_t* seg_count;          /* n_edge: segments per branch */
  double*  seg_dwell;          /* n_edge * seg_cap */
  int32_t* seg_state;          /* n_edge * seg_cap, 0-based */
  int32_t  seg_cap;
  double*  PL;                 /* (2*n_node+1) x n row-major */
} orc_dump;

/* ---- deterministic scalar maths (spec shared, by restatement, with the HIP kernels) ---- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);   /* the sampler's streams: 7 rounds */
double orc_u01(uint32_t x);   /* (x + 0.5) * 2^-32 */
double orc_log(double x);
double orc_exp(double x);
uint32_t orc_stream_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t iter, uint32_t entity, uint32_t draw);
double orc_neglog_u32(uint32_t k);   /* -log((k + 0.5) 2^-32), table-based, ~1 ulp */
double orc_stream_u(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t iter,
                    uint32_t entity, uint32_t draw);

double orc_r_dpois(double x, double lambda);   /* R's dpois_raw (stirlerr + bd0, R <= 4.0.x), what R-stream mode puts in place of the recurrence */
int orc_rstream_selftest(uint32_t seed, int n_unif, int n_exp, double* unif_out, double* exp_out);   /* set.seed(seed); runif(n_unif); rexp(n_exp) */
double orc_r_qnorm(double p);   /* R's qnorm5(p, 0, 1, TRUE, FALSE) (Wichura AS 241), behind norm_rand() of kind INVERSION */
/* set.seed(seed); rnorm(n_norm); rgamma(n_gamma, shape, scale = scale) -- Rf_rgamma (Ahrens-Dieter GD / GS) as restated for R-stream mode */
int orc_rstream_gamma_selftest(uint32_t seed, int n_norm, int n_gamma, double shape, double scale, double* norm_out, double* gamma_out);

/* ---- per-function entry points for known-answer tests ---- */
/* shortener  src/phylomap.cpp:44-73; returns new segment count; stats row gets += counts */
int  orc_shortener(double* d, int32_t* s, int m, int n, double* stats_row);
/* matTospmat src/phylomap.cpp:801-816 (dense copy with entries <= 1e-7 zeroed) */
void orc_matTospmat(const double* B_rm, int n, double* out_rm);
/* matexp src/phylomap.cpp:2964-2968 + abs() at :2980/:3042; all matrices row-major */
void orc_matexp(const double* L_rm, const double* R_rm, const double* dvals, int n, double t, double* P_rm);
/* same product as orc_matexp with a k-ordered fma chain per entry (model of the MFMA f64 pipeline) */
void orc_matexp_fma(const double* L_rm, const double* R_rm, const double* dvals, int n, double t, double* P_rm);
/* arma::expmat call sites src/phylomap.cpp:3226,3243,3359,3383: Pade(6) scaling-and-squaring */
int  orc_expmat_pade(const double* A_rm, int n, double* out_rm);
/* Felsenstein pruning with B^(m-1): makePLrcpp :503-514, _bigtree :516-529, SPARSE :490-501 */
int  orc_makePL(const orc_tree* x, int n, const double* Bchain_rm, const int32_t* nen,
                const int32_t* seg_count, int normalise, double* PL_rm);
/* Felsenstein pruning with P(t_b): makePLold :2877-2895 / makePLexp :2899-2906 */
int  orc_makePLexp(const orc_tree* x, int n, const double* P_rm_cube, const int32_t* nen, double* PL_rm);

/* ---- drivers ---- */
/* maketreelistMCMC / _bigtree / SPARSE.  Q, B column-major (R layout). out: N x (n+n(n-1)) column-major. */
int orc_maketreelistMCMC(const orc_tree* x, int n, const double* Q_cm, const double* pid,
                         const double* B_cm, double Omega, const int32_t* nen,
                         const int32_t* nodelist, int32_t root, int32_t N, int variant,
                         int faithful_search, orc_rng* rng, double* out_cm, orc_dump* dump);

/* the Q-updating drivers: sweep + Gibbs/MH updates of the rate matrix each iteration (bf: prior[4], ks: prior[6]) */
int orc_maketreelistMCMC_qupdate(const orc_tree* x, int n, const double* Q_cm, const double* pid, const double* B_cm,
                                 double Omega, const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                                 int variant, const double* prior, int faithful_search, orc_rng* rng, double* out_cm,
                                 orc_dump* dump);

int orc_qupdate_apply(int variant, int n, double* Q_rm, double Omega, const double* prior, const double* row,
                      uint32_t seed_lo, uint32_t seed_hi, uint32_t iter);

/* Multi-tree drivers (variant ORC_MCMC_MT / ORC_MCMC_KSMT): xs = `treecount` trees with equal tip / edge counts,
 * nen_m treecount x 2*Nnode, nodelist_m treecount x (Nnode-1) row-major, roots[treecount].  out: N x (n+n*n+2+3k+1)
 * column-major, the last column the 0-based index of the tree drawn that iteration (R/sumstatMCMCmt.R:41). */
int orc_maketreelistMCMCmt(const orc_tree* const* xs, int treecount, int n, const double* Q_cm, const double* pid,
                           const double* B_cm, double Omega, const int32_t* nen_m, const int32_t* nodelist_m,
                           const int32_t* roots, int32_t N, int variant, const double* prior, int faithful_search,
                           orc_rng* rng, double* out);

/* maketreelistEXP src/phylomap.cpp:3001-3051. lefts/rights column-major, d = n x n col-major (diag used).
 * recompute_expm_each_iter: bit 0 = recompute P(t_b) and the pruning pass every iteration as the reference does (:2980-2981);
 * bit 1 = divide every internal partial-likelihood row by its sum (not in the reference: same sampler, no underflow). */
int orc_maketreelistEXP(const orc_tree* x, int n, const double* Q_cm, const double* pid,
                        const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
                        const double* lefts_cm, const double* rights_cm, const double* d_cm,
                        int faithful_search, int recompute_expm_each_iter,
                        orc_rng* rng, double* out_cm, orc_dump* dump);

#ifdef __cplusplus
}
#endif
#endif
