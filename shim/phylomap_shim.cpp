// phylomap_shim.cpp -- the ONLY file that includes R headers.  Replaces src/RcppExports.cpp for the hot-path
// symbols: same `.Call` names, same argument order (src/RcppExports.cpp:11,34,57,80,106,132; R/RcppExports.R:4-30), so
// R/sumstat*.R and user code run unchanged; plus `phylomap_tree_orders`, the native form of their helper preamble.  Each export unpacks the SEXPs into the plain structs of
// include/phylomap_hip.h, calls the C-ABI, and returns a fresh N x cols numeric matrix.
//
// NOT compiled in this repository's CI: R / Rcpp are not installed in the build image.  Build inside the
// phylomap package with  PKG_CPPFLAGS=-I<repo>/include  PKG_LIBS=-L<repo>/phylomap_amd -lphylomap_hip.
#include <Rcpp.h>

#include <algorithm>
#include <vector>

#include "phylomap_hip.h"

using namespace Rcpp;

namespace {

// The R list `x` flattened into owned arrays; view() builds the phm_tree from the CURRENT addresses of those arrays, so a
// FlatTree can live in any container (no pointers into itself are kept).
struct FlatTree {
  std::vector<int32_t> edge, states, map_off, mapnames;
  std::vector<double> edge_length, maps;
  int32_t n_node = 0, n_edge = 0;
  explicit FlatTree(List x) {
    IntegerMatrix e = as<IntegerMatrix>(x["edge"]);                 // src/phylomap.cpp:904
    IntegerVector st = as<IntegerVector>(x["states"]);              // :910 (REALSXP in the shipped RDS: coerced)
    List m = x["maps"], mn = x["mapnames"];                         // :896-897
    edge.assign(e.begin(), e.end());                                // column-major, 1-based
    states.assign(st.begin(), st.end());
    map_off.push_back(0);
    for (int b = 0; b < m.size(); ++b) {
      NumericVector d = as<NumericVector>(m[b]);
      IntegerVector s = as<IntegerVector>(mn[b]);
      if (d.size() != s.size()) stop("maps[[%d]] and mapnames[[%d]] differ in length", b + 1, b + 1);
      maps.insert(maps.end(), d.begin(), d.end());
      mapnames.insert(mapnames.end(), s.begin(), s.end());
      map_off.push_back((int32_t)maps.size());
    }
    if (x.containsElementNamed("edge.length")) {                    // :3034
      NumericVector el = as<NumericVector>(x["edge.length"]);
      edge_length.assign(el.begin(), el.end());
    }
    n_node = as<int>(x["Nnode"]);                                   // :907
    n_edge = e.nrow();
  }
  phm_tree view() const {
    phm_tree t;
    t.n_tips = (int32_t)states.size();
    t.n_node = n_node;
    t.n_edge = n_edge;
    t.edge = edge.data();
    t.edge_length = edge_length.empty() ? nullptr : edge_length.data();
    t.states = states.data();
    t.map_off = map_off.data();
    t.maps = maps.data();
    t.mapnames = mapnames.data();
    return t;
  }
};

// What the R session asks of the device beyond the reference's arguments.  All optional:
//   options(phylomap.hip.replicas = S)   S independent chains on the same data (default 1 = the reference's semantics exactly)
//   x$sites = S x length(x$states) matrix of 1-based tip states: S sites of an alignment on the same tree, one chain each
//   options(phylomap.hip.reduce = FALSE) return list of S matrices instead of their sum
//   options(phylomap.hip.device = d)     HIP device ordinal
//   options(phylomap.hip.devices = D)    D GPUs of the node (0..D-1), or an integer vector of ordinals: the chains / sites
//                                        (sumstatEXP: the N samples) are sharded over them inside the call (phm_options.n_devices)
//   options(phylomap.hip.rescale = TRUE) row-rescaled pruning pass for sumstatMCMC / SPARSEsumstatMCMC / sumstatEXP, which
//                                        underflow in the reference beyond a few hundred / thousand tips (phm_options.rescale_pruning)
//   options(phylomap.hip.mapping = "auto" | "replicas" | "branches" | "tiles")   how a sweep is laid over the lanes
//   options(phylomap.hip.cap_tail = p)   tail probability the per-branch path capacity is provisioned for (0 = automatic)
// With S > 1 the default result is the N x cols matrix of statistics SUMMED over the chains / sites (what a likelihood over
// sites needs, and the form in which 10^4 chains cost one row each); a single chain on a big tree uses one lane per branch, from a
// few hundred chains on the lanes are the chains -- the throughput mapping is reached from R this way.
struct HipRequest {
  phm_options o;
  int S = 1;
  bool summed = true;
  std::vector<int32_t> site_states;            // S * n_tips, replica-major (phm_tree.states with tips_per_replica)
};

// Called inside the RNGScope (src/RcppExports.cpp:38): two draws from R's stream seed Philox, so set.seed() keeps
// controlling the result.
HipRequest request_from_R(List x, int n_tips) {
  HipRequest rq;
  phm_options& o = rq.o;
  o = phm_options();
  uint64_t hi = (uint64_t)(unif_rand() * 4294967296.0), lo = (uint64_t)(unif_rand() * 4294967296.0);
  o.seed = (hi << 32) | lo;
  Environment base("package:base");
  Function getOption = base["getOption"];
  o.device = as<int>(getOption("phylomap.hip.device", -1));
  {
    IntegerVector devs = as<IntegerVector>(getOption("phylomap.hip.devices", IntegerVector(0)));
    if (devs.size() == 1 && devs[0] >= 1) {                          // a count: GPUs 0 .. D-1
      const int D = devs[0];
      if (D > PHM_MAX_DEVICES) stop("phylomap.hip.devices: at most %d GPUs", (int)PHM_MAX_DEVICES);
      o.n_devices = D;
      for (int d = 0; d < D; ++d) o.devices[d] = d;
    } else if (devs.size() > 1) {                                    // explicit ordinals
      if (devs.size() > PHM_MAX_DEVICES) stop("phylomap.hip.devices: at most %d GPUs", (int)PHM_MAX_DEVICES);
      o.n_devices = (int32_t)devs.size();
      for (int d = 0; d < devs.size(); ++d) o.devices[d] = devs[d];
    }
  }
  o.rescale_pruning = as<bool>(getOption("phylomap.hip.rescale", false)) ? 1 : 0;
  o.cap_tail = as<double>(getOption("phylomap.hip.cap_tail", 0.0));
  {
    std::string mp = as<std::string>(getOption("phylomap.hip.mapping", "auto"));
    if (mp == "auto") o.mapping = PHM_MAP_AUTO;
    else if (mp == "replicas") o.mapping = PHM_MAP_REPLICAS;
    else if (mp == "branches") o.mapping = PHM_MAP_BRANCHES;
    else if (mp == "tiles") o.mapping = PHM_MAP_TILES;
    else stop("phylomap.hip.mapping must be \"auto\", \"replicas\", \"branches\" or \"tiles\"");
  }
  rq.S = as<int>(getOption("phylomap.hip.replicas", 1));
  rq.summed = as<bool>(getOption("phylomap.hip.reduce", true));
  if (x.containsElementNamed("sites")) {
    IntegerMatrix sites = as<IntegerMatrix>(x["sites"]);             // REALSXP matrices are coerced, as x$states is
    if (sites.ncol() != n_tips) stop("x$sites must have one column per tip (%d), it has %d", n_tips, sites.ncol());
    rq.S = sites.nrow();
    rq.site_states.resize((size_t)rq.S * n_tips);
    for (int s = 0; s < rq.S; ++s)
      for (int t = 0; t < n_tips; ++t) rq.site_states[(size_t)s * n_tips + t] = sites.begin()[s + (size_t)rq.S * t];   // column-major
    o.tips_per_replica = 1;
  }
  if (rq.S < 1) stop("phylomap.hip.replicas must be >= 1");
  o.n_replicas = rq.S;
  o.reduce = (rq.S > 1 && rq.summed) ? 1 : 0;
  return rq;
}

void check(int32_t st) {
  if (st != PHM_OK) stop("phylomap_hip: %s: %s", phm_status_string(st), phm_last_error());   // END_RCPP turns it into an R error
}

// N x cols matrix (one chain, or the sum over chains), or a list of S such matrices
SEXP wrap_result(const std::vector<double>& buf, int N, int cols, const HipRequest& rq) {
  const bool single = rq.S == 1 || rq.summed;
  if (single) {
    NumericMatrix out(N, cols);
    std::copy(buf.begin(), buf.begin() + (size_t)N * cols, out.begin());
    return out;
  }
  List res(rq.S);
  for (int s = 0; s < rq.S; ++s) {
    NumericMatrix m(N, cols);
    std::copy(buf.begin() + (size_t)s * N * cols, buf.begin() + (size_t)(s + 1) * N * cols, m.begin());
    res[s] = m;
  }
  return res;
}

typedef int32_t (*mcmc_fn)(const phm_tree*, int32_t, const double*, const double*, const double*, double,
                           const int32_t*, const int32_t*, int32_t, int32_t, const phm_options*, double*);

SEXP run_mcmc(mcmc_fn fn, SEXP xSEXP, SEXP QSEXP, SEXP pidSEXP, SEXP BSEXP, SEXP OmegaSEXP, SEXP nenSEXP,
              SEXP nodelistSEXP, SEXP rootSEXP, SEXP NSEXP) {
  RNGScope scope;
  List x = as<List>(xSEXP);
  FlatTree ft(x);
  NumericMatrix Q(QSEXP), B(BSEXP);
  NumericVector pid(pidSEXP);
  IntegerVector nen(nenSEXP), nodelist(nodelistSEXP);
  const int n = Q.nrow(), N = as<int>(NSEXP), cols = n + n * (n - 1);  // :926
  HipRequest rq = request_from_R(x, (int)ft.states.size());
  phm_tree t = ft.view();
  if (!rq.site_states.empty()) t.states = rq.site_states.data();
  std::vector<double> buf((size_t)N * cols * ((rq.S == 1 || rq.summed) ? 1 : rq.S));
  check(fn(&t, n, Q.begin(), pid.begin(), B.begin(), as<double>(OmegaSEXP), nen.begin(), nodelist.begin(),
           as<int>(rootSEXP), N, &rq.o, buf.data()));
  return wrap_result(buf, N, cols, rq);
}

}  // namespace

RcppExport SEXP phylomap_maketreelistMCMC(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen, SEXP nodelist,
                                          SEXP root, SEXP N) {
  BEGIN_RCPP
  return run_mcmc(phm_maketreelistMCMC, x, Q, pid, B, Omega, nen, nodelist, root, N);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistMCMC_bigtree(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen,
                                                  SEXP nodelist, SEXP root, SEXP N) {
  BEGIN_RCPP
  return run_mcmc(phm_maketreelistMCMC_bigtree, x, Q, pid, B, Omega, nen, nodelist, root, N);
  END_RCPP
}

RcppExport SEXP phylomap_SPARSEmaketreelistMCMC(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen,
                                                SEXP nodelist, SEXP root, SEXP N) {
  BEGIN_RCPP
  return run_mcmc(phm_SPARSEmaketreelistMCMC, x, Q, pid, B, Omega, nen, nodelist, root, N);
  END_RCPP
}

typedef int32_t (*qupd_fn)(const phm_tree*, int32_t, const double*, const double*, const double*, double, const int32_t*,
                           const int32_t*, int32_t, int32_t, const double*, int32_t, const phm_options*, double*);

// maketreelistMCMCbf / maketreelistMCMCks (src/RcppExports.cpp:106,132): the reference edits the caller's Q and B in place
// (src/phylomap.cpp:1212-1217) -- vignettes re-create Q before every call for that reason; this binding leaves them alone.
static SEXP run_qupdate(qupd_fn fn, int cols_extra_k, int dic, SEXP xSEXP, SEXP QSEXP, SEXP pidSEXP, SEXP BSEXP, SEXP OmegaSEXP,
                        SEXP nenSEXP, SEXP nodelistSEXP, SEXP rootSEXP, SEXP NSEXP, SEXP priorSEXP) {
  RNGScope scope;
  List x = as<List>(xSEXP);
  FlatTree ft(x);
  NumericMatrix Q(QSEXP), B(BSEXP);
  NumericVector pid(pidSEXP), prior(priorSEXP);
  IntegerVector nen(nenSEXP), nodelist(nodelistSEXP);
  const int n = Q.nrow(), N = as<int>(NSEXP);
  const int k = cols_extra_k ? n / 2 - 1 : 0;
  NumericMatrix out(N, n + n * n + 2 + 3 * k + 1 + dic);            // :1293 (bf), :1857 (ks), :3230 / :3372 (DIC)
  // S > 1 (options / x$sites): the chains are sites sharing one Q; the updates see, and `out` holds, the statistics summed over them
  HipRequest rq = request_from_R(x, (int)ft.states.size());
  phm_tree t = ft.view();
  if (!rq.site_states.empty()) t.states = rq.site_states.data();
  check(fn(&t, n, Q.begin(), pid.begin(), B.begin(), as<double>(OmegaSEXP), nen.begin(), nodelist.begin(),
           as<int>(rootSEXP), N, prior.begin(), (int32_t)prior.size(), &rq.o, out.begin()));
  return out;
}

RcppExport SEXP phylomap_maketreelistMCMCbf(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen, SEXP nodelist, SEXP root,
                                            SEXP N, SEXP prior) {
  BEGIN_RCPP
  return run_qupdate(phm_maketreelistMCMCbf, 0, 0, x, Q, pid, B, Omega, nen, nodelist, root, N, prior);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistMCMCks(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen, SEXP nodelist, SEXP root,
                                            SEXP N, SEXP prior) {
  BEGIN_RCPP
  return run_qupdate(phm_maketreelistMCMCks, 1, 0, x, Q, pid, B, Omega, nen, nodelist, root, N, prior);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistMCMC2sDICt(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen, SEXP nodelist,
                                                SEXP root, SEXP N, SEXP prior) {      // src/RcppExports.cpp:211
  BEGIN_RCPP
  return run_qupdate(phm_maketreelistMCMC2sDICt, 0, 1, x, Q, pid, B, Omega, nen, nodelist, root, N, prior);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistMCMCksDICt(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen, SEXP nodelist,
                                                SEXP root, SEXP N, SEXP prior) {      // src/RcppExports.cpp:237
  BEGIN_RCPP
  return run_qupdate(phm_maketreelistMCMCksDICt, 1, 1, x, Q, pid, B, Omega, nen, nodelist, root, N, prior);
  END_RCPP
}

typedef int32_t (*mt_fn)(const phm_tree*, int32_t, int32_t, const double*, const double*, const double*, double, const int32_t*,
                         const int32_t*, const int32_t*, int32_t, const double*, int32_t, const phm_options*, double*);

// maketreelistMCMCmt / maketreelistMCMCksmt (src/RcppExports.cpp:158,184): x is the R list of trees; nen_m / nodelist_m are
// R integer matrices with one ROW per tree (R/sumstatMCMCmt.R:37-43), handed over column-major as R stores them.
static SEXP run_qupdate_mt(mt_fn fn, int hidden, SEXP xSEXP, SEXP QSEXP, SEXP pidSEXP, SEXP BSEXP, SEXP OmegaSEXP, SEXP nenSEXP,
                           SEXP nodelistSEXP, SEXP rootsSEXP, SEXP NSEXP, SEXP priorSEXP) {
  RNGScope scope;
  List xs(xSEXP);
  std::vector<FlatTree> fts;
  for (int j = 0; j < xs.size(); ++j) fts.emplace_back(as<List>(xs[j]));
  std::vector<phm_tree> trees;                                      // views taken once the container has stopped growing
  for (const FlatTree& f : fts) trees.push_back(f.view());
  NumericMatrix Q(QSEXP), B(BSEXP);
  NumericVector pid(pidSEXP), prior(priorSEXP);
  IntegerMatrix nen_m(nenSEXP), nodelist_m(nodelistSEXP);
  IntegerVector roots(rootsSEXP);
  const int n = Q.nrow(), N = as<int>(NSEXP);
  const int k = hidden ? n / 2 - 1 : 0;
  NumericMatrix out(N, n + n * n + 2 + 3 * k + 1);                  // :2338 (mt), :2815 (ksmt); last column tree_number
  HipRequest rq = request_from_R(List(), 0);                        // lists of trees: one chain per tree (R/sumstatMCMCmt.R)
  phm_options o = rq.o;
  o.n_replicas = 1; o.reduce = 0; o.tips_per_replica = 0;
  check(fn(trees.data(), (int32_t)trees.size(), n, Q.begin(), pid.begin(), B.begin(), as<double>(OmegaSEXP), nen_m.begin(),
           nodelist_m.begin(), roots.begin(), N, prior.begin(), (int32_t)prior.size(), &o, out.begin()));
  return out;
}

RcppExport SEXP phylomap_maketreelistMCMCmt(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen_m, SEXP nodelist_m, SEXP roots,
                                            SEXP N, SEXP prior) {          // src/RcppExports.cpp:159
  BEGIN_RCPP
  return run_qupdate_mt(phm_maketreelistMCMCmt, 0, x, Q, pid, B, Omega, nen_m, nodelist_m, roots, N, prior);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistMCMCksmt(SEXP x, SEXP Q, SEXP pid, SEXP B, SEXP Omega, SEXP nen_m, SEXP nodelist_m,
                                              SEXP roots, SEXP N, SEXP prior) {      // src/RcppExports.cpp:185
  BEGIN_RCPP
  return run_qupdate_mt(phm_maketreelistMCMCksmt, 1, x, Q, pid, B, Omega, nen_m, nodelist_m, roots, N, prior);
  END_RCPP
}

RcppExport SEXP phylomap_maketreelistEXP(SEXP xSEXP, SEXP QSEXP, SEXP pidSEXP, SEXP nenSEXP, SEXP nodelistSEXP,
                                         SEXP rootSEXP, SEXP NSEXP, SEXP leftsSEXP, SEXP rightsSEXP, SEXP dSEXP) {
  BEGIN_RCPP
  RNGScope scope;
  FlatTree ft(as<List>(xSEXP));
  NumericMatrix Q(QSEXP), lefts(leftsSEXP), rights(rightsSEXP), d(dSEXP);
  NumericVector pid(pidSEXP);
  IntegerVector nen(nenSEXP), nodelist(nodelistSEXP);
  const int n = Q.nrow(), N = as<int>(NSEXP);
  NumericMatrix out(N, n + n * (n - 1));                            // :3031
  HipRequest rq = request_from_R(List(), 0);                        // samples are i.i.d.: N is the replica axis already
  phm_options o = rq.o;
  o.n_replicas = 1; o.reduce = 0; o.tips_per_replica = 0;
  const phm_tree t = ft.view();
  check(phm_maketreelistEXP(&t, n, Q.begin(), pid.begin(), nen.begin(), nodelist.begin(), as<int>(rootSEXP), N,
                            lefts.begin(), rights.begin(), d.begin(), &o, out.begin()));
  return out;
  END_RCPP
}

// O(E) replacement of the helper preamble every R/sumstat*.R file carries (pruningwiseedgeorder / makenodelist / myreorder,
// R/sumstatMCMC.R:1-18: interpreted O(E^2) loops around ape::reorder(x, "pruningwise")): one call returns all three.
// R side: shim/R/phylomap_tree_orders.R.  edge: x$edge (E x 2 integer matrix, 1-based), n_tips: length(x$states).
RcppExport SEXP phylomap_tree_orders(SEXP edgeSEXP, SEXP ntipsSEXP) {
  BEGIN_RCPP
  IntegerMatrix edge(edgeSEXP);
  const int E = edge.nrow(), T = as<int>(ntipsSEXP);
  IntegerVector nen(E), nodelist(T > 2 ? T - 2 : 0);
  std::vector<int32_t> nl(T > 1 ? T - 1 : 1);
  int32_t root = 0;
  check(phm_tree_orders(T, E, edge.begin(), nen.begin(), nl.data(), &root));
  for (int i = 0; i < nodelist.size(); ++i) nodelist[i] = nl[i];
  return List::create(Named("nen") = nen, Named("nodelist") = nodelist, Named("root") = (int)root);
  END_RCPP
}
