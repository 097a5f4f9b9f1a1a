// phm_rtc.h -- pruning through an UNSTRUCTURED sparse chain matrix (5..32 states): a kernel specialised for the matrix's pattern,
// compiled at model upload through hipRTC.
//
// SPARSEmakePLrcpp / spmmmmvFORpl (src/phylomap.cpp:490-501, :451-457) walk the non-zeros of an Armadillo sp_mat -- any pattern:
// an amino-acid neighbour structure is not banded.  With one lane per replica the lane's chain vector lives in registers, and
// a register file is indexed statically: the column index of a non-zero has to be known when the code is generated.  The band
// kernels of phm_wtiles.hip get that from the band's shape; for any other pattern the shape is only known once the caller's Q is
// there, so the kernel is generated THEN: for each row i one straight line
//     acc = +0;  acc = fma(c_k, x_j, acc)  for the non-zeros (i, j) of the row, j ascending
// -- the dense specification (n > 4, fused, j ascending from +0) with its exact-zero terms left out
// (fma(0, x_j, acc) = acc for the finite non-negative x of a partial likelihood): the same bits as the matrix-core kernel.
// Coefficient VALUES are read from a device buffer (scalar loads), so a rate update that keeps the pattern reuses the kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "phm_sched.h"

namespace phm {

// what the generated kernel reads; the same text is handed to the runtime compiler (phm_rtc.cpp: PHM_RTC_ABI)
#define PHM_RTC_ABI(X)                                                                                                                  \
  X(struct RtcUpStep { int parent; int child[2]; int edge[2]; };                                                                        \
    struct RtcUpParams {                                                                                                                \
      int n_states; int ldt; int n_tips; int n_node; int n_edge; int n_tiles; int normalise; int tips_per_replica; int tip_masks;       \
      int klong;                                                                                                                        \
      const RtcUpStep* up; const int* up_order; const double* colL; const double* maskL; const unsigned char* tips;                     \
      const unsigned short* mcount; double* PL; unsigned int* err; const double* coef;                                                  \
    };)
#define PHM_RTC_IDENT(...) __VA_ARGS__
PHM_RTC_ABI(PHM_RTC_IDENT)
#undef PHM_RTC_IDENT
static_assert(sizeof(RtcUpStep) == sizeof(UpStep), "the generated kernel reads the engine's UpStep array");

constexpr int RTC_SPARSE_NMAX = 32;          // the chain vectors of a step live in registers (two of them, plus the other child's result)
constexpr double RTC_SPARSE_MAX_FILL = 0.5;  // non-zeros / n^2 up to which the specialised kernel is generated (beyond: matrix cores)

struct SparseUpKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  int device = -1;
  int n = 0, nnz = 0;
  std::vector<int32_t> row_ptr, col;           // the pattern the kernel was generated for (CSR, columns ascending)
};

// The kernel for this pattern on the current device: from the per-process cache, or generated and compiled now (about a second).
// Returns nullptr and fills `err` when hipRTC is not usable; the caller then keeps the matrix-core kernel.
const SparseUpKernel* rtc_sparse_up_kernel(int n, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col, std::string& err);

// one launch per height level, a wave per (node, tile)
hipError_t launch_sparse_up(const SparseUpKernel& k, const RtcUpParams& p, const std::vector<int32_t>& up_off, hipStream_t stream);

// the generated source (tests / inspection)
std::string rtc_sparse_up_source(int n, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col);

}  // namespace phm
