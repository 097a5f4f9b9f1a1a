/*
 * phylomap_hip.h -- C-ABI of the MI355X-native stochastic-mapping engine.
 *
 * This is the drop-in boundary for the hot path of vnminin/phylomap: the `.Call` layer
 * (src/RcppExports.cpp:9-259, R/RcppExports.R:4-50) binds `phylomap_maketreelist*`; an Rcpp shim
 *  unpacks the SEXPs into the plain structs below and calls the `phm_maketreelist*`
 * entry points, which replace the C++ drivers of src/phylomap.cpp one for one.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/R/torch types;
 *  - all inputs are HOST buffers owned by the caller, read-only (the reference aliases and, in the
 *    Q-updating variants, mutates R's memory -- src/phylomap.cpp:917,1212-1217; this library copies);
 *  - matrices follow R's layout (COLUMN-major): Q, B, lefts, rights, d, edge, and the result;
 *  - every call returns a phm_status; phm_last_error() gives a thread-local message
 *    (the reference throws through BEGIN_RCPP/END_RCPP, src/RcppExports.cpp:35,53);
 *  - randomness: Philox4x32-7 keyed by phm_options.seed, counter (block, entity, iteration,
 *    replica), draw d = word d & 3 of block d >> 2 mapped to (x + 0.5) 2^-32; the Rcpp shim draws the seed from R's stream inside its RNGScope
 *    (src/RcppExports.cpp:38) so set.seed() still controls results;
 *  - there is NO CPU fallback: without a usable HIP device every compute call returns
 *    PHM_ERR_NO_DEVICE.
 */
#ifndef PHYLOMAP_HIP_H
#define PHYLOMAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHM_VERSION 300     /* 300: phm_options.n_devices / devices[] (replica sharding over the GPUs of a node inside the one-shot
                                   calls); the measurement / test aids moved out of phm_options into the struct
                                   phm_debug_options, set by phm_set_debug_options.  200: named option fields instead of reserved[6]; phm_info.recoveries */
#define PHM_MAX_DEVICES 8   /* GPUs of one node (MI355X: 8 per node over xGMI) */

typedef enum phm_status {
  PHM_OK = 0,
  PHM_ERR_BAD_INPUT = 1,      /* malformed tree / model / sizes (reference: no validation at all) */
  PHM_ERR_UNSUPPORTED = 2,    /* e.g. a state count this build has no kernel for */
  PHM_ERR_NO_DEVICE = 3,      /* no HIP device, or a HIP call failed */
  PHM_ERR_OOM = 4,            /* device memory */
  PHM_ERR_ZERO_PROB = 5,      /* all-zero / non-finite probability vector (RcppArmadillo::sample throws) */
  PHM_ERR_CAPACITY = 6,       /* a sweep outgrew its dwell capacity and could not be recovered (recovery switched off, larger slots
                                 do not fit in HBM, or the 128-segment scratch of the state-per-lane tile kernel, PHM_MAP_REPLICAS with
                                 n > 4); std::list in the reference is unbounded.  After a FAILED recovery the handle has no device
                                 state left: every later call on it returns this status until it is destroyed */
  PHM_ERR_UNIF_CAP = 7,       /* newunifSample needed > 300 jumps (src/phylomap.cpp:120-125) */
  PHM_ERR_STATE = 8           /* API misuse (engine not created, iteration range, ...) */
} phm_status;

/* which exported driver of src/phylomap.cpp the engine stands in for */
typedef enum phm_variant {
  PHM_MCMC = 0,               /* maketreelistMCMC          src/phylomap.cpp:891-935  */
  PHM_MCMC_BIGTREE = 1,       /* maketreelistMCMC_bigtree  src/phylomap.cpp:942-986  (row-normalised PL, :525) */
  PHM_MCMC_SPARSE = 2,        /* SPARSEmaketreelistMCMC    src/phylomap.cpp:822-870  (B entries <= 1e-7 dropped, :811) */
  PHM_MCMC_KS = 3             /* the TREE SWEEP of maketreelistMCMCks (treesampleks :1422-1432) with Q held fixed: hidden-rates
                                 Q of even size n = 2k+2, parity tip masks (:1838-1845), tips re-sampled (:1384-1397), all
                                 consecutive state pairs counted into n x n counters (shortenerbf :1010-1014).  Result layout
                                 man/sumstatMCMCks.Rd:19: N x (n + n*n + 2 + 3k + 1): dwell, counts (row-major from,to),
                                 l01, l10, rkappas, lkappas, gammas (recordQks :1789-1798), root state (0-based).
                                 The per-iteration Gibbs/MH updates of Q (:1862-1866) run in phm_maketreelistMCMCks. */
  ,
  PHM_MCMC_BF = 4             /* tree sweep of maketreelistMCMCbf (treesamplebf :1169-1179): tips observed, n x n counts incl. self
                                 pairs (shortenerbf :1010-1014), row-normalised pruning (:1085); n-generic in the reference.  Result:
                                 N x (n + n*n + 3): dwell, counts (row-major from,to), Q[0,1], Q[1,0], root state -- at n = 2 the
                                 reference's time0,time1,n00,n01,n10,n11,l01,l10,root_state (R/sumstatMCMCbf.R:33; its column 8 is
                                 hard-wired at :1129, which is n + n*n + 2 for two states).  The rate updates (two states only)
                                 run in phm_maketreelistMCMCbf */
  ,
  PHM_MCMC_MT = 5             /* tree sweep of maketreelistMCMCmt (treesamplemtNS :2157-2165): the bf sweep with the UN-normalised
                                 pruning makePLrcppmt :1938-1950; two states */
  ,
  PHM_MCMC_KSMT = 6           /* tree sweep of maketreelistMCMCksmt (:2722-2844): the ks sweep with the un-normalised pruning */
} phm_variant;

/* The phylomap tree object `x` (fields read at src/phylomap.cpp:896-910 and :3034). */
typedef struct phm_tree {
  int32_t n_tips;              /* length(x$states) */
  int32_t n_node;              /* x$Nnode (= n_tips-1: strictly bifurcating, :508-510) */
  int32_t n_edge;              /* nrow(x$edge) */
  const int32_t* edge;         /* n_edge x 2 column-major, 1-based (parent, child); tips are 1..n_tips (:904) */
  const double*  edge_length;  /* x$edge.length (:3034); EXP only, may be NULL for MCMC */
  const int32_t* states;       /* x$states, 1-based (:910); n_tips values, or n_replicas*n_tips when
                                  phm_options.tips_per_replica != 0 (replica-major) */
  const int32_t* map_off;      /* n_edge+1 offsets into maps/mapnames */
  const double*  maps;         /* x$maps flattened: dwell times per segment (:896) */
  const int32_t* mapnames;     /* x$mapnames flattened: 1-based states (:897, :29) */
} phm_tree;

typedef struct phm_model {
  int32_t n_states;            /* n = nrow(Q) */
  const double* Q;             /* n x n rate matrix, column-major */
  const double* pid;           /* n root prior */
  const double* B;             /* n x n, I + Q/Omega (R/sumstatMCMC.R:25), column-major; NULL -> computed */
  double Omega;                /* dominating rate; must exceed every |q_ii| (man/sumstatMCMC.Rd:14) */
  int32_t variant;             /* phm_variant */
} phm_model;

/* how a sweep is laid over the lanes (phm_options.mapping).  n <= 4: REPLICAS = one lane per replica, one wave per 64-replica
 * tile walks the tree (largest replica counts); BRANCHES = the sweep of one chain spread over the device for latency (one chain or a handful, large trees);
 * TILES = one wave per (64-replica tile, branch) (10^2 .. 10^5 replicas).  5..64 states: REPLICAS = one wave per 64-replica
 * tile, replicas in turn, lanes = states (phm_wide.hip; lists of trees); BRANCHES = one wave per (replica, branch), lanes =
 * states (a handful of chains); TILES = one lane per replica, one wave per (tile, item), pruning on the matrix cores or over
 * the non-zeros of a sparse B (the default beyond 50 000 / n_edge chains, clamped to 1..32: the measured crossover).
 * phm_maketreelistEXP: AUTO / TILES = one wave per (tile of 64 samples, branch); REPLICAS = one wave per tile of 64 samples
 * walks the tree (dwell sums then add in the reference's order). */
typedef enum phm_mapping { PHM_MAP_AUTO = 0, PHM_MAP_REPLICAS = 1, PHM_MAP_BRANCHES = 2, PHM_MAP_TILES = 3 } phm_mapping;

typedef struct phm_options {
  uint64_t seed;               /* Philox key */
  int32_t n_replicas;          /* S: independent chains / sites run side by side (>=1; 0 -> 1) */
  int32_t replica_offset;      /* global id of this device's first replica (multi-GPU sharding) */
  int32_t reduce;              /* 0: statistics per replica; 1: summed over replicas per iteration */
  int32_t tips_per_replica;    /* 0: all replicas share x$states; 1: one tip vector per replica (sites) */
  int32_t device;              /* HIP device ordinal; -1 = current device */
  int32_t iters_per_launch;    /* MCMC iterations fused into one kernel launch; 0 -> default */
  double  cap_tail;            /* dwell capacity: the 1+Poisson(Omega*t_b) quantile at this tail, per branch.  0 -> automatic: 1e-3 for
                                  the sequential streams of PHM_MAP_REPLICAS (an overflow there is rare per TILE and recovered); for the
                                  fixed slots of PHM_MAP_BRANCHES / PHM_MAP_TILES min(1e-9, 0.05 / (S * E * max_iters)), floored at
                                  1e-16 -- at most 0.05 expected recoveries over the draws the engine is created for, so the HBM
                                  held per replica grows (slowly: a quantile) with max_iters */
  int32_t mapping;             /* phm_mapping: how a sweep is laid over the lanes (one tree); PHM_MAP_AUTO = by replica count.
                                  Same draws and counts in every mapping; dwell sums differ in the last bits between
                                  PHM_MAP_REPLICAS and the other two (summation order).  Path lengths: 65 535 segments per branch
                                  in PHM_MAP_TILES, slot sizes in PHM_MAP_BRANCHES; PHM_MAP_REPLICAS with 5..64 states (never the
                                  automatic choice for one tree) holds at most 128 segments per branch and replica and answers
                                  PHM_ERR_UNSUPPORTED / PHM_ERR_CAPACITY beyond. */
  int32_t storage;             /* dwell-stream storage of PHM_MAP_REPLICAS: 0 = automatic, 1 = one ring per tile (half the HBM),
                                  2 = two buffers (5 % faster sweep for n <= 4) */
  int32_t rescale_pruning;     /* phm_maketreelistEXP: 1 = divide every internal partial-likelihood row by its sum in the pruning
                                  pass.  The reference's makePLexp (src/phylomap.cpp:2899-2906) does not rescale, so sumstatEXP
                                  underflows (PHM_ERR_ZERO_PROB) beyond a few hundred tips; node draws do not depend on a row's
                                  scale, so this is the same sampler in exact arithmetic.
                                  phm_maketreelistMCMC / phm_SPARSEmaketreelistMCMC: 1 = the same for their pruning pass (what
                                  makePLrcpp_bigtree :525 does; the plain and the SPARSE driver underflow on trees of thousands
                                  of tips, man/sumstatMCMC_bigtree.Rd:17) */
  int32_t no_recovery;         /* 1 = a sweep that outgrows its slots fails with PHM_ERR_CAPACITY.  Default (0): the engine is
                                  rebuilt with doubled slots and the iterations run so far are replayed -- bit-identical, every
                                  random number being addressed by (replica, iteration, entity) -- so a run cannot abort where
                                  the reference's std::list (src/phylomap.cpp:18-21) would grow */
  int32_t sparse_chains;       /* 5..64 states with PHM_MAP_TILES: pruning chains and forward draws over the NON-ZEROS of B only
                                  (what SPARSEmakePLrcpp :490-501 / SPARSEresamplebranchstates :218-261 get from sp_mat).  0 =
                                  automatic: used when n <= 32 and B has a half-bandwidth of 1 (tridiagonal) or 2 (make2sQ hidden
                                  rates), or is sparse enough for the pattern-specialised kernels; 1 = required
                                  (PHM_ERR_UNSUPPORTED otherwise); 2 = never (chains on the matrix cores).
                                  Same bits either way: a skipped term is an exact zero */
  int32_t n_devices;           /* one-shot calls (phm_maketreelist*): 0 / 1 = one GPU (`device`); 2..PHM_MAX_DEVICES = the
                                  n_replicas chains / sites (phm_maketreelistEXP: the N samples) are sharded by GLOBAL replica id
                                  over devices[0 .. n_devices-1], one host thread and one engine per device, no traffic between
                                  the devices while sampling; with reduce = 1 the per-tile sums are folded in device order (the
                                  only exchange: N x cols doubles per device).  Every random number is addressed by the global
                                  replica id, so counts are those of ONE device exactly and dwell sums agree to rounding
                                  (<= 1e-12; bit-identical on the mappings whose per-tile sums do not depend on the tile count).
                                  The resident engine (phm_engine_*) is per device: shard with replica_offset there.
                                  The caller of the reference -- R/sumstatMCMC_bigtree.R:21-29 -> .Call -> one function
                                  (src/phylomap.cpp:942-986) -- reaches all GPUs of the node through this field */
  int32_t devices[PHM_MAX_DEVICES]; /* HIP ordinals; an ordinal may repeat (several engines on one GPU: rehearsal on a one-GPU box) */
  int32_t reserved[3];         /* must be 0 */
} phm_options;

/* Measurement and test aids, kept out of phm_options: set per THREAD by phm_set_debug_options and read by the engines and
 * one-shot calls that thread creates afterwards (NULL resets to all-zero). */
typedef struct phm_debug_options {
  int32_t pruning_form;        /* 5..64 states with PHM_MAP_TILES: form of the pruning kernel, 0 = by tile count, 1 = one wave per
                                  (node, tile) with the chain matrix in registers, 2 = one workgroup / wave per 16-replica block,
                                  3 = one wave per (node, tile) with the chain matrix in LDS and two waves per SIMD (33..64 states;
                                  what 0 chooses there from 192 / 256 tiles on) (same bits).
                                  2..4 states with PHM_MAP_BRANCHES: the subtree clusters of the pruning sweep, 0 = by path length
                                  (some branch expected to hold >= 96 segments: dependency-driven), 1 = a barrier per tree level,
                                  2 = dependency-driven (same bits) */
  int32_t phase_timing;        /* 1 = record HIP events between the phases of a sweep (phm_engine_phase_ms) */
  int32_t fail_recovery;       /* 1 = every capacity recovery "does not fit" (exercises the dead-handle path) */
  int32_t branch_group;        /* > 0: branches per wave of the 5..64-state branch kernel (clamped to 1..64); 0 = automatic */
  int32_t level_groups;        /* (tile, branch) mapping, n <= 4 (5..32 states: 2 / 3 = the node draws and the band pruning kernel over
                                  subtree clusters; automatic on a deep tree): tree passes over clusters of tree levels (one launch per tier of eight
                                  levels) instead of one launch per level: 0 = automatic (tiles x internal nodes <= 65 536; a deep, ladder-like
                                  tree at any tile count, with clusters cut by subtree size), 1 = never, 2 = always, 3 = always with
                                  clusters cut by subtree size.  Same bits */
  int32_t q_timing;            /* 1 = the rate-updating drivers print the mean host time of the phases of an iteration to stderr */
  double  pade_pivot_min;      /* > 0: smallest pivot phm_expm_pade_mfma's unpivoted block elimination accepts (default 1e-3;
                                  1e300 sends every matrix to the pivoted kernel) */
  int32_t reserved[4];
} phm_debug_options;

typedef struct phm_info {
  int32_t n_states, n_edge, n_replicas, n_replicas_padded, n_cols, max_iters;
  int64_t device_bytes;        /* HBM held by the engine */
  int64_t rows_per_replica;    /* capacity (64-lane rows) of one tile's dwell storage */
  int64_t seg_read;            /* sum over branches x valid replicas x iterations run so far of (m_b + m'_b): segments
                                  read plus segments written; feeds the algorithmic-bytes figure of bench.py */
  int64_t seg_written;         /* reserved (0) */
  double  last_run_ms;         /* HIP-event time of the last phm_engine_run (all its launches) */
  int32_t last_run_launches;
  int32_t iters_done;
  int32_t recoveries;          /* capacity recoveries (rebuild + replay) this handle has gone through; a timed region asserts 0 */
  int32_t mapping;             /* the phm_mapping in force (the automatic choice resolved) */
  int32_t sparse_chains;       /* bit 0: pruning chains run over the non-zeros of the chain matrix; bit 1: forward draws over the band of B;
                                  bit 2: the pruning kernel was generated for the matrix's pattern (unstructured sparsity, hipRTC) */
  int32_t reserved;
} phm_info;

typedef struct phm_engine phm_engine;

/* ---- library ---- */
int32_t     phm_version(void);
/* sizeof of the plain structs of this header as the library was built (which: 0 phm_options, 1 phm_info, 2 phm_tree,
 * 3 phm_model, 4 phm_debug_options; anything else: -1) -- lets a foreign-function binding (ctypes, cgo, .C) check its mirror of the layout */
int32_t     phm_struct_size(int32_t which);
int32_t     phm_device_count(void);
const char* phm_last_error(void);
const char* phm_status_string(int32_t status);
/* measurement / test aids of this thread -- see the phm_debug_options struct; NULL = defaults */
int32_t     phm_set_debug_options(const phm_debug_options* dbg);
/* measurement aid: HIP-event milliseconds of the sampling kernel of this thread's last phm_maketreelistEXP call */
double      phm_last_kernel_ms(void);

/* ---- reference-shaped one-shot entry points (what the Rcpp shim binds) ----
 * Each mirrors the argument list of the exported C++ driver it replaces; `out` is the caller-allocated
 * N x (n + n(n-1)) column-major result (allocMatrix(REALSXP, N, cols) in the shim).  With
 * opt->n_replicas = S > 1 and reduce = 0, `out` holds S such matrices back to back.
 * nen / nodelist / root (R/sumstatMCMC.R:1-18) are checked for consistency with `x->edge`
 * and otherwise unused: the engine derives its own O(E) sweep schedules. */
int32_t phm_maketreelistMCMC(         /* src/phylomap.cpp:891, src/RcppExports.cpp:34 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const phm_options* opt, double* out);
int32_t phm_maketreelistMCMC_bigtree( /* src/phylomap.cpp:942, src/RcppExports.cpp:57 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const phm_options* opt, double* out);
int32_t phm_SPARSEmaketreelistMCMC(   /* src/phylomap.cpp:822, src/RcppExports.cpp:11 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const phm_options* opt, double* out);
/* Tree sweep of sumstatMCMCks with Q held fixed (see PHM_MCMC_KS); out: N x (n + n*n + 2 + 3k + 1) column-major. */
int32_t phm_maketreelistMCMCks_sweep( /* src/phylomap.cpp:1802 minus the Q updates of :1862-1866 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const phm_options* opt, double* out);
/* Tree sweep of sumstatMCMCbf with Q held fixed (see PHM_MCMC_BF), ANY n: tips observed, all consecutive pairs counted
 * (the "n + n^2" layout of shortenerbf); out: N x (n + n*n + 3) column-major: dwell, counts (row-major from,to), Q[0,1], Q[1,0],
 * root state (0-based). */
int32_t phm_maketreelistMCMCbf_sweep( /* treesamplebf src/phylomap.cpp:1169-1179 inside the N-loop of :1295-1301, minus the updates */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const phm_options* opt, double* out);
/* The Q-updating drivers: tree sweep on the device + Gibbs/MH update of the rate matrix on the host, every iteration.
 * bf: two states, prior = c(a01,b01,a10,b10), out N x 9 (R/sumstatMCMCbf.R:33).  ks: n = 2k+2 >= 4,
 * prior = c(a_l,b_l,a_k,b_k,a_g,b_g), out N x (n+n*n+2+3k+1) (man/sumstatMCMCks.Rd:19).  Inputs are never written
 * (the reference edits the caller's Q and B in place, src/phylomap.cpp:1212-1217).  Gamma variates: Marsaglia-Tsang on the
 * Philox stream (R's Rf_rgamma is third-party code).  With S > 1 replicas (sites sharing Q) the updates see, and `out`
 * holds, the statistics summed over sites. */
int32_t phm_maketreelistMCMCbf(       /* src/phylomap.cpp:1258, src/RcppExports.cpp:106 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const double* prior, int32_t n_prior,
    const phm_options* opt, double* out);
int32_t phm_maketreelistMCMCks(       /* src/phylomap.cpp:1802, src/RcppExports.cpp:132 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const double* prior, int32_t n_prior,
    const phm_options* opt, double* out);
/* The DIC drivers: the bf / ks drivers plus log p(y|Q) by matrix exponentiation every iteration (expmat(Q t_b) for every
 * branch -- Pade scaling-and-squaring on the device -- then pruning with scale factors in nen order), appended as the last
 * column: out is N x 10 (2sDICt) or N x (n+n*n+2+3k+2) (ksDICt).  One chain (n_replicas = 1); needs x->edge_length. */
int32_t phm_maketreelistMCMC2sDICt(   /* src/phylomap.cpp:3183, src/RcppExports.cpp:211 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const double* prior, int32_t n_prior,
    const phm_options* opt, double* out);
int32_t phm_maketreelistMCMCksDICt(   /* src/phylomap.cpp:3300, src/RcppExports.cpp:237 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid, const double* B, double Omega,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N, const double* prior, int32_t n_prior,
    const phm_options* opt, double* out);
/* The multi-tree drivers: `trees` is the R list x of `n_trees` phylomap trees with equal tip and edge counts
 * (R/sumstatMCMCmt.R:33-35).  Every iteration sweeps every tree with the current Q -- one launch, tree j on replica tile j --
 * draws one tree uniformly, writes that tree's row plus its 0-based index (the last column; R/sumstatMCMCmt.R:41
 * "tree_number") and updates Q from it with the mt twins of the updates (updatel01mtNS :2192, updateksl01mt :2371, ...).
 * nen_m: n_trees x 2*Nnode, nodelist_m: n_trees x (Nnode-1), both COLUMN-major as R passes its matrices; either may be
 * NULL (orders are derived from edge; when given they are checked against it).  out: N x (n+n*n+2+3k+1) column-major.
 * prior: 4 numbers (mt), 8 (ksmt: l01, l10, kappas, gammas shape/rate pairs as the mt updates index them). */
int32_t phm_maketreelistMCMCmt(       /* src/phylomap.cpp:2267, src/RcppExports.cpp:158 */
    const phm_tree* trees, int32_t n_trees, int32_t n_states, const double* Q, const double* pid, const double* B,
    double Omega, const int32_t* nen_m, const int32_t* nodelist_m, const int32_t* roots, int32_t N, const double* prior,
    int32_t n_prior, const phm_options* opt, double* out);
int32_t phm_maketreelistMCMCksmt(     /* src/phylomap.cpp:2722, src/RcppExports.cpp:184 */
    const phm_tree* trees, int32_t n_trees, int32_t n_states, const double* Q, const double* pid, const double* B,
    double Omega, const int32_t* nen_m, const int32_t* nodelist_m, const int32_t* roots, int32_t N, const double* prior,
    int32_t n_prior, const phm_options* opt, double* out);
int32_t phm_maketreelistEXP(          /* src/phylomap.cpp:3001, src/RcppExports.cpp:80 */
    const phm_tree* x, int32_t n_states, const double* Q, const double* pid,
    const int32_t* nen, const int32_t* nodelist, int32_t root, int32_t N,
    const double* lefts, const double* rights, const double* d,
    const phm_options* opt, double* out);

/* ---- host-side rate-matrix update of the Q-updating variants (no device needed) ----
 * One iteration of updatel01/l10 (bf) or updateksl01/l10, updaterkappas, updatelkappas, updategammas (ks) applied to Q
 * (column-major, edited in place) given a statistics row: n dwell sums then n*n counts, row-major (from,to). */
int32_t phm_qupdate_apply(int32_t variant, int32_t n_states, double* Q, double Omega, const double* prior, int32_t n_prior,
                          const double* row, uint64_t seed, uint32_t iter);

/* ---- inspection: the pruning kernel generated for an unstructured sparse chain matrix (no device needed) ----
 * 5..32 states, PHM_MAP_TILES: when the chain matrix (B, or B thresholded at 1e-7 for SPARSEmaketreelistMCMC, src/phylomap.cpp:801-816)
 * is at most half full and not banded, its pruning chains run in a kernel generated for the matrix's PATTERN and compiled at model
 * upload through hipRTC (what SPARSEmakePLrcpp :490-501 gets from sp_mat for any pattern).  This returns that kernel's HIP source for the
 * non-zero pattern of the n x n column-major matrix M: the number of bytes needed (including the terminator); up to `cap` bytes are
 * written to `buf` (may be NULL). */
int32_t phm_sparse_kernel_source(int32_t n_states, const double* M, char* buf, int32_t cap);

/* ---- host-side traversal orders (no device needed) ----
 * O(E) native replacement of the R helper preamble pruningwiseedgeorder / makenodelist / myreorder
 * (R/sumstatMCMC.R:1-18, interpreted O(E^2) loops around ape::reorder(x,"pruningwise")).
 * edge: n_edge x 2 column-major, 1-based.  nen: n_edge rows (1-based); nodelist: Nnode-1 node ids; root: node id. */
int32_t phm_tree_orders(int32_t n_tips, int32_t n_edge, const int32_t* edge, int32_t* nen, int32_t* nodelist,
                        int32_t* root);

/* ---- batched transition matrices (K1 / K1') ----
 * phm_expm_eigen: P_b = |L diag(exp(d_i t_b)) R|  (matexp, src/phylomap.cpp:2964-2968 + abs at :2980,:3042)
 * phm_expm_pade : P_b = expmat(Q t_b), Pade(6) scaling-and-squaring (arma::expmat call sites :3226,:3243,:3359,:3383)
 * out: n_t matrices, each n x n ROW-major (out[b*n*n + i*n + j]). */
int32_t phm_expm_eigen(int32_t n_states, const double* lefts, const double* rights, const double* d,
                       const double* t, int32_t n_t, int32_t device, double* out, double* kernel_ms);
/* phm_expm_eigen on the matrix cores (v_mfma_f64_16x16x4_f64), 16 < n_states <= 64: same product, fused k-slices, so the
 * last bits differ from phm_expm_eigen; the sumstatEXP sampler keeps the exact kernel. */
int32_t phm_expm_eigen_mfma(int32_t n_states, const double* lefts, const double* rights, const double* d,
                            const double* t, int32_t n_t, int32_t device, double* out, double* kernel_ms);
int32_t phm_expm_pade(int32_t n_states, const double* Q, const double* t, int32_t n_t, int32_t device,
                      double* out, double* kernel_ms);

/* phm_expm_pade with every matrix product on the matrix cores (v_mfma_f64_16x16x4_f64), 16 < n_states <= 64: solve(D, E) by
 * block Gauss-Jordan elimination without row exchanges between the 16 x 16 blocks; a matrix that meets a pivot below 1e-3 there
 * is recomputed by phm_expm_pade's pivoted kernel inside the same call.  Agrees with phm_expm_pade to rounding (<= 2e-13).
 * Test aid: phm_debug_options.pade_pivot_min overrides the 1e-3. */
int32_t phm_expm_pade_mfma(int32_t n_states, const double* Q, const double* t, int32_t n_t, int32_t device,
                           double* out, double* kernel_ms);

/* ---- resident engine (inputs stay in HBM between calls; what bench.py times) ---- */
int32_t phm_engine_create(const phm_tree* x, const phm_model* model, const phm_options* opt,
                          int32_t max_iters, phm_engine** out);
/* the same over a list of `n_trees` trees (equal tip and edge counts) sharing one model: opt->n_replicas chains PER TREE,
 * tree j's chains on their own 64-lane tiles (replica index j * n_replicas + c in every per-replica call; Philox replica
 * word replica_offset + 64 * tiles_per_tree * j + c).  reduce and tips_per_replica are not available here. */
int32_t phm_engine_create_multi(const phm_tree* trees, int32_t n_trees, const phm_model* model, const phm_options* opt,
                                int32_t max_iters, phm_engine** out);
/* enqueue iterations [iters_done, iters_done + n_iters) on `hip_stream` (a hipStream_t, NULL = default
 * stream); asynchronous */
int32_t phm_engine_run(phm_engine* e, int32_t n_iters, void* hip_stream);
/* wait for the stream, collect the device error word, fill timing */
int32_t phm_engine_sync(phm_engine* e);
/* copy statistics of iterations [iter0, iter0+n) to host.
 * reduce = 0: out[r][ (col)*n + (i-iter0) ] for replica r (n x cols column-major per replica)
 * reduce = 1: one n x cols column-major matrix (sum over replicas) */
int32_t phm_engine_read_stats(phm_engine* e, int32_t iter0, int32_t n, double* out);
/* reduce = 1 only: run the tile reduction on `hip_stream` and return a DEVICE pointer to n x cols doubles,
 * row-major [iteration][column] (for handing to RCCL without a host round trip).  The buffer belongs to this entry point
 * alone: it stays as the caller (or an in-place all-reduce) left it until the next phm_engine_reduced_stats_device call;
 * phm_engine_read_stats does not touch it */
int32_t phm_engine_reduced_stats_device(phm_engine* e, int32_t iter0, int32_t n, void* hip_stream, void** out_dev);
/* measurement aid: HIP-event time (ms) of n_iters repetitions of the pruning sweep alone (makePLrcpp*,
 * src/phylomap.cpp:503-529) on the current chain state; segment counts and paths are not modified */
int32_t phm_engine_time_pruning(phm_engine* e, int32_t n_iters, void* hip_stream, double* ms_out);
/* replace the rate matrix between sweeps (Q-updating variants, src/phylomap.cpp:1212-1217, :1862-1866); Q column-major,
 * B = I + Q/Omega recomputed, chain state kept; the bf/ks parameter columns record the Q in force at each sweep */
int32_t phm_engine_set_model(phm_engine* e, const double* Q);
/* chain state of one replica after the last iteration (tests): any pointer may be NULL.
 * seg_dwell: n_edge * seg_cap; node_states: 2*n_tips-1, 1-based; PL: (2*n_tips-1) x n row-major */
int32_t phm_engine_dump(phm_engine* e, int32_t replica, int32_t* seg_count, double* seg_dwell, int32_t seg_cap,
                        int32_t* node_states, double* PL);
int32_t phm_engine_info(phm_engine* e, phm_info* info);
/* measurement aid (phm_debug_options.phase_timing = 1, (tile, item) mappings): HIP-event milliseconds of the last phm_engine_run, summed over
 * its sweeps, for the four phases of a sweep: pruning levels (makePLrcpp*), root + node draws (sampleinternalnodes*), the branch
 * kernel (sampleabranch + updatedwelltimes), the statistics reductions.  Valid after phm_engine_sync. */
int32_t phm_engine_phase_ms(phm_engine* e, double* out4);
void    phm_engine_destroy(phm_engine* e);

#ifdef __cplusplus
}
#endif
#endif
