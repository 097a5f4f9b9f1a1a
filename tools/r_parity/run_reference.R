# Runs the genuine phylomap package on the case written by export_case.py (see README.md).
# Usage: Rscript run_reference.R case_dir
library(ape); library(phylomap)
d <- commandArgs(trailingOnly = TRUE)[1]
edge <- as.matrix(read.csv(file.path(d, "edge.csv"), header = FALSE)); storage.mode(edge) <- "integer"; dimnames(edge) <- NULL
el <- scan(file.path(d, "edge_length.csv"), quiet = TRUE)
states <- scan(file.path(d, "states.csv"), quiet = TRUE)
Q <- as.matrix(read.csv(file.path(d, "Q.csv"), header = FALSE)); dimnames(Q) <- NULL
pid <- scan(file.path(d, "pid.csv"), quiet = TRUE)
par <- read.csv(file.path(d, "params.csv"))
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
# the helper preamble every wrapper carries (R/sumstatMCMC.R:1-18)
nen <- pruningwiseedgeorder(z); nodelist <- makenodelist(z); root <- myreorder(z)
write.table(nen, file.path(d, "nen.csv"), row.names = FALSE, col.names = FALSE)
write.table(nodelist, file.path(d, "nodelist.csv"), row.names = FALSE, col.names = FALSE)
write.table(root, file.path(d, "root.csv"), row.names = FALSE, col.names = FALSE)
for (v in c("sumstatMCMC", "sumstatMCMC_bigtree", "SPARSEsumstatMCMC")) {
  set.seed(par$seed)
  ss <- do.call(v, list(z, Q, pid, par$Omega, par$N))
  write.table(format(ss, digits = 17), file.path(d, paste0(v, ".csv")), sep = ",", row.names = FALSE, col.names = FALSE, quote = FALSE)
}
# sumstatEXP (R/sumstatEXP.R:19-31): the eigen-decomposition R computes is written out too, so that the oracle exponentiates
# with R's own lefts / rights / d (LAPACK's eigenvector scaling differs from numpy's in the last bits)
uer <- eigen(Q); lefts <- uer$vectors; rights <- solve(lefts); dd <- diag(uer$values)
for (nm in c("lefts", "rights", "dd")) write.table(format(get(nm), digits = 17), file.path(d, paste0(nm, ".csv")), sep = ",", row.names = FALSE, col.names = FALSE, quote = FALSE)
z$edge.length <- el
set.seed(par$seed)
ss <- sumstatEXP(z, Q, pid, par$N)
write.table(format(ss, digits = 17), file.path(d, "sumstatEXP.csv"), sep = ",", row.names = FALSE, col.names = FALSE, quote = FALSE)
# The rate-updating drivers: Rf_rgamma + runif on R's stream after every sweep (src/phylomap.cpp:1299-1300, :1862-1866).  They edit
# the caller's Q in place (:1212-1217), so each call gets a fresh copy.
if (file.exists(file.path(d, "params_q.csv"))) {
  parq <- read.csv(file.path(d, "params_q.csv"))
  read_maps <- function(f, zz) {
    lines <- readLines(f); zz$maps <- list(); zz$mapnames <- list()
    for (i in seq_along(lines)) {
      p <- strsplit(lines[i], ";")[[1]]
      dw <- as.numeric(strsplit(p[1], " ")[[1]]); st <- as.integer(strsplit(p[2], " ")[[1]])
      names(dw) <- st; zz$maps[[i]] <- dw; zz$mapnames[[i]] <- st
    }
    zz
  }
  z2 <- read_maps(file.path(d, "maps2.csv"), z)
  z2$states <- scan(file.path(d, "states2.csv"), quiet = TRUE)
  Q2 <- as.matrix(read.csv(file.path(d, "Q2.csv"), header = FALSE)); dimnames(Q2) <- NULL
  prior_bf <- scan(file.path(d, "prior_bf.csv"), quiet = TRUE)
  set.seed(parq$seed)
  ss <- sumstatMCMCbf(z2, Q2 + 0, c(0.5, 0.5), parq$Omega, parq$N, prior_bf)
  write.table(format(ss, digits = 17), file.path(d, "sumstatMCMCbf.csv"), sep = ",", row.names = FALSE, col.names = FALSE, quote = FALSE)
  Q4 <- as.matrix(read.csv(file.path(d, "Q.csv"), header = FALSE)); dimnames(Q4) <- NULL
  prior_ks <- scan(file.path(d, "prior_ks.csv"), quiet = TRUE)
  set.seed(parq$seed)
  ss <- sumstatMCMCks(z, Q4 + 0, pid, parq$Omega, parq$N, prior_ks)
  write.table(format(ss, digits = 17), file.path(d, "sumstatMCMCks.csv"), sep = ",", row.names = FALSE, col.names = FALSE, quote = FALSE)
}
writeLines(paste(R.version$major, R.version$minor, sep = "."), file.path(d, "R_version.txt"))   # dpois_raw changed in R 4.1 (ebd0)
cat("done\n")
