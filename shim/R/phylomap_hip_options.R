# Reaching the GPU's replica axis from R without touching the sumstat* wrappers (INTEGRATION.md, "Reaching the throughput from R").
# The shim (shim/phylomap_shim.cpp, request_from_R) reads these options and one optional field of the tree object:
#   options(phylomap.hip.replicas = S)    S independent chains on the same data; the result is the N x cols matrix of statistics
#                                         SUMMED over the chains (divide by S for means)
#   options(phylomap.hip.reduce = FALSE)  ... or a list of S matrices, one per chain
#   options(phylomap.hip.device = d)      HIP device ordinal
#   options(phylomap.hip.devices = D)     D GPUs of the node (or a vector of ordinals): chains / sites / sumstatEXP samples are
#                                         sharded over them INSIDE the call; same counts as one GPU, dwell sums to rounding
#   options(phylomap.hip.rescale = TRUE)  row-rescaled pruning for sumstatMCMC / SPARSEsumstatMCMC / sumstatEXP on big trees
#                                         (the reference underflows there; sumstatMCMC_bigtree always rescales)
#   options(phylomap.hip.mapping = "auto" | "replicas" | "branches" | "tiles"),  options(phylomap.hip.cap_tail = p)
#   z$sites                               S x length(z$states) matrix of 1-based tip states: one chain per alignment site
# These helpers only set / clear them.

phylomap_hip_options <- function(replicas = NULL, reduce = NULL, device = NULL, devices = NULL, rescale = NULL, mapping = NULL,
                                 cap_tail = NULL) {
  old <- options(phylomap.hip.replicas = replicas, phylomap.hip.reduce = reduce, phylomap.hip.device = device,
                 phylomap.hip.devices = devices, phylomap.hip.rescale = rescale, phylomap.hip.mapping = mapping,
                 phylomap.hip.cap_tail = cap_tail)
  invisible(old)                                   # options(old) restores the previous settings
}

# sumstat over an alignment: `fn` is any fixed-Q wrapper of the package (sumstatMCMC, sumstatMCMC_bigtree, SPARSEsumstatMCMC) or a
# rate-updating one (sumstatMCMCbf, sumstatMCMCks: the sites then share one Q); `sites` is the S x tips matrix of 1-based states.
# Returns the N x cols matrix summed over sites, or (reduce = FALSE, fixed-Q wrappers only) a list of S matrices.
sumstat_sites <- function(fn, z, sites, ..., reduce = TRUE) {
  stopifnot(is.matrix(sites), ncol(sites) == length(z$states))
  z$sites <- sites
  old <- options(phylomap.hip.reduce = reduce)
  on.exit(options(old))
  fn(z, ...)
}

# S independent chains on the data of `z` (e.g. for between-chain diagnostics): list of S matrices
sumstat_chains <- function(fn, z, S, ...) {
  old <- options(phylomap.hip.replicas = as.integer(S), phylomap.hip.reduce = FALSE)
  on.exit(options(old))
  fn(z, ...)
}
