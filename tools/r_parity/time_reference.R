# Times the genuine phylomap package on a case written by export_case.py (SURVEY 8(d): "the reference CPU phylomap.cpp path timed on the same
# box's host cores"; R is not installed in this repository's build image, so bench.py's cpu_baseline is the oracle port -- this script is for a
# machine that has R + phylomap + ape).   Usage: Rscript time_reference.R case_dir [N]
# Prints branch x site realisations per second (nrow(z$edge) * N / elapsed) for sumstatMCMC_bigtree and sumstatEXP, single-threaded as the package is.
library(ape); library(phylomap)
args <- commandArgs(trailingOnly = TRUE)
d <- args[1]
edge <- as.matrix(read.csv(file.path(d, "edge.csv"), header = FALSE)); storage.mode(edge) <- "integer"; dimnames(edge) <- NULL
el <- scan(file.path(d, "edge_length.csv"), quiet = TRUE)
states <- scan(file.path(d, "states.csv"), quiet = TRUE)
Q <- as.matrix(read.csv(file.path(d, "Q.csv"), header = FALSE)); dimnames(Q) <- NULL
pid <- scan(file.path(d, "pid.csv"), quiet = TRUE)
par <- read.csv(file.path(d, "params.csv"))
N <- if (length(args) > 1) as.integer(args[2]) else par$N
ntip <- length(states)
z <- list(edge = edge, Nnode = ntip - 1L, tip.label = paste0("t", 1:ntip), edge.length = el)
class(z) <- "phylo"; attr(z, "order") <- "cladewise"
lines <- readLines(file.path(d, "maps.csv"))
z$maps <- list(); z$mapnames <- list()
for (i in seq_along(lines)) {
  p <- strsplit(lines[i], ";")[[1]]
  dw <- as.numeric(strsplit(p[1], " ")[[1]]); st <- as.integer(strsplit(p[2], " ")[[1]])
  names(dw) <- st; z$maps[[i]] <- dw; z$mapnames[[i]] <- st
}
z$states <- states
z$node.states <- matrix(1L, nrow = nrow(edge), ncol = 2)
E <- nrow(edge)
# the R preamble (pruningwiseedgeorder / makenodelist: interpreted O(E^2) loops, R/sumstatMCMC.R:1-18) is part of what an R user waits for: timed apart
t_pre <- system.time({ nen <- pruningwiseedgeorder(z); nodelist <- makenodelist(z); root <- myreorder(z) })[["elapsed"]]
set.seed(par$seed)
t_mcmc <- system.time(ss <- sumstatMCMC_bigtree(z, Q, pid, par$Omega, N))[["elapsed"]]
cat(sprintf("tips %d, branches %d, N %d\n", ntip, E, N))
cat(sprintf("helper preamble (once per call, inside every wrapper): %.3f s\n", t_pre))
cat(sprintf("sumstatMCMC_bigtree: %.3f s = %.4g branch-site realisations/s (1 thread, preamble included)\n", t_mcmc, E * N / t_mcmc))
if (ntip <= 300) {   # makePLexp does not rescale: sumstatEXP underflows beyond a few hundred tips (src/phylomap.cpp:2899-2906)
  set.seed(par$seed)
  t_exp <- system.time(ss <- sumstatEXP(z, Q, pid, N))[["elapsed"]]
  cat(sprintf("sumstatEXP:          %.3f s = %.4g branch-site realisations/s\n", t_exp, E * N / t_exp))
}
