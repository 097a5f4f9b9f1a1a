# O(E) native replacements of the helper preamble that every R/sumstat*.R file of phylomap carries
# (R/sumstatMCMC.R:1-18: pruningwiseedgeorder, makenodelist, myreorder -- interpreted O(E^2) / O(E Nnode) loops around
# ape::reorder(x, "pruningwise"); about 4e8 interpreted comparisons before the sampler starts on a 10 000-tip tree).
# Drop this file into the package's R/ directory AFTER the sumstat*.R files (NAMESPACE:2 exports by pattern, the
# last-sourced definition of a name wins -- that is how the ten identical copies of the preamble coexist today), or delete
# lines 1-18 from those files.  The values are the ones the reference computes: `nen` = for each position of ape's
# pruningwise edge order the row of x$edge; `nodelist` = parents of the pruningwise edge table at rows E - 2i,
# i = 1 .. Nnode - 1; `root` = the parent in its last row.  The C side (phm_tree_orders, include/phylomap_hip.h) checks the
# tree and fails with an R error for a tree that is not strictly bifurcating (src/phylomap.cpp:508-510 assumes it is).
.phylomap_orders <- function(x) .Call('phylomap_tree_orders', PACKAGE = 'phylomap', x$edge, length(x$states))
pruningwiseedgeorder <- function(x) .phylomap_orders(x)$nen
makenodelist <- function(x) .phylomap_orders(x)$nodelist
myreorder <- function(x) .phylomap_orders(x)$root
