// phm_qupdate.h -- rate-matrix updates of the Q-updating drivers (host side); see phm_qupdate.cpp
#pragma once

#include <stdint.h>

namespace phm {

// Q: n x n column-major, edited in place.  row: one iteration's statistics, n dwell sums then n*n counts (row-major from,to).
// bf: prior[4] = (alpha01, beta01, alpha10, beta10)         src/phylomap.cpp:1189-1253
void bf_updates(double* Q, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter);
// ks: prior[6] = (alpha_lambda, beta_lambda, alpha_kappa, beta_kappa, alpha_gamma, beta_gamma), n = 2k+2 >= 4   :1435-1785
// mt = true: the multi-tree twins (prior[8]: l01, l10, kappa, gamma shape/rate pairs)                         :2371-2705
void ks_updates(double* Q, int n, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter, bool mt = false);
// two-state multi-tree updates (acceptance tested), prior[4]                                                  :2192-2262
void mt_updates(double* Q, double Omega, const double* prior, const double* row, uint64_t seed, uint32_t iter);
// index of the tree whose row drives this iteration's update (sampleOnce over unit weights, :2347-2348); n_trees = ran off the end
uint32_t pick_tree(int n_trees, uint64_t seed, uint32_t iter);

}  // namespace phm
