// phm_exp.hip -- the matrix-exponentiation path (sumstatEXP, src/phylomap.cpp:2877-3051).
//
//   K1  expm_eigen_kernel : P_b = |L diag(exp(d_k t_b)) R|          matexp :2964-2968, abs :2980/:3042
//   K1' expm_pade_kernel  : P_b = expmat(Q t_b), Pade(6) + squarings arma::expmat at :3226,:3243,:3359,:3383
//   K2e exp_pl_kernel     : pruning with P(t_b)                      makePLold :2877-2895 / makePLexp :2899-2906
//   K5  exp_sample_kernel : per sample: node states top-down (sampleinternalnodesEXP :2910-2961) and the
//                           end-point-conditioned uniformisation sampler (newunifSample :93-208);
//                           one LANE per i.i.d. sample, topology wave-uniform, B^k e_j table in LDS.
// Summation orders follow the arithmetic spec, so results match the CPU oracle bit for bit.
#include "phm_exp.h"

namespace phm {

// ------------------------------------------------------------------------------------------------
// K1: eigen route.  Several small matrices share a workgroup; one thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EXP_BLOCK) void expm_eigen_kernel(int n, const double* __restrict__ L,
                                                               const double* __restrict__ R,
                                                               const double* __restrict__ dvals,
                                                               const double* __restrict__ t, int n_t, int mpb,
                                                               double* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* ex = reinterpret_cast<double*>(smem);          // [mpb][n]  exp(d_k t_b)
  const int nn = n * n;
  const int b0 = blockIdx.x * mpb;
  for (int i = threadIdx.x; i < mpb * n; i += EXP_BLOCK) {
    int q = i / n, k = i - q * n;
    if (b0 + q < n_t) ex[i] = phm_exp(dvals[k] * t[b0 + q]);               // :2966
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < mpb * nn; idx += EXP_BLOCK) {
    int q = idx / nn, e = idx - q * nn;
    if (b0 + q >= n_t) continue;
    int i = e / n, j = e - i * n;
    const double* eq = ex + q * n;
    double acc = (L[i * n] * eq[0]) * R[j];
    for (int k = 1; k < n; ++k) acc += (L[i * n + k] * eq[k]) * R[k * n + j];
    out[(size_t)(b0 + q) * nn + e] = fabs(acc);
  }
}

hipError_t launch_expm_eigen(int n, const double* L, const double* R, const double* dvals, const double* t, int n_t,
                             double* out, hipStream_t stream) {
  int mpb = EXP_BLOCK / (n * n);
  if (mpb < 1) mpb = 1;
  int grid = (n_t + mpb - 1) / mpb;
  hipLaunchKernelGGL(expm_eigen_kernel, dim3(grid), dim3(EXP_BLOCK), sizeof(double) * mpb * n, stream, n, L, R, dvals,
                     t, n_t, mpb, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K1': Pade(6) scaling-and-squaring, one workgroup per matrix, operands in a global workspace.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_gemm(const double* A, const double* B, double* C, int n) {
  for (int e = threadIdx.x; e < n * n; e += EXP_BLOCK) {
    int i = e / n, j = e - i * n;
    double acc = A[i * n] * B[j];
    for (int k = 1; k < n; ++k) acc += A[i * n + k] * B[k * n + j];
    C[e] = acc;
  }
  __syncthreads();
}

__global__ __launch_bounds__(EXP_BLOCK) void expm_pade_kernel(int n, const double* __restrict__ Q,
                                                              const double* __restrict__ t,
                                                              const int32_t* __restrict__ sq, double* __restrict__ work,
                                                              double* __restrict__ out, uint32_t* err) {
  __shared__ double fv[128];
  __shared__ int s_piv;
  const int nn = n * n;
  const int b = blockIdx.x;
  double* A = work + (size_t)b * 5 * nn;
  double *Em = A + nn, *Dm = A + 2 * nn, *X = A + 3 * nn, *T2 = A + 4 * nn;
  const double tb = t[b];
  const int s = sq[b];
  const double sc = ldexp(1.0, s);
  double c = 0.5;
  for (int e = threadIdx.x; e < nn; e += EXP_BLOCK) {
    double a = (Q[e] * tb) / sc;
    int i = e / n, j = e - i * n;
    A[e] = a; X[e] = a;
    double ca = c * a;
    Em[e] = (i == j) ? ca + 1.0 : ca;
    Dm[e] = (i == j) ? -ca + 1.0 : -ca;
  }
  __syncthreads();
  bool positive = true;
  for (int i = 2; i <= 6; ++i) {
    c = c * (double)(6 - i + 1) / (double)(i * (2 * 6 - i + 1));
    block_gemm(A, X, T2, n);
    for (int e = threadIdx.x; e < nn; e += EXP_BLOCK) {
      double x = T2[e];
      X[e] = x;
      Em[e] += c * x;
      if (positive) Dm[e] += c * x; else Dm[e] -= c * x;
    }
    __syncthreads();
    positive = !positive;
  }
  // solve Dm * Xsol = Em (partial pivoting), Xsol written into X
  for (int col = 0; col < n; ++col) {
    if (threadIdx.x == 0) {
      int piv = col; double best = fabs(Dm[col * n + col]);
      for (int r = col + 1; r < n; ++r) { double v = fabs(Dm[r * n + col]); if (v > best) { best = v; piv = r; } }
      if (!(best > 0.0)) atomicOr(err, DERR_ZERO_PROB);
      s_piv = piv;
    }
    __syncthreads();
    int piv = s_piv;
    if (piv != col) {
      for (int k = threadIdx.x; k < n; k += EXP_BLOCK) {
        double a = Dm[col * n + k]; Dm[col * n + k] = Dm[piv * n + k]; Dm[piv * n + k] = a;
        a = Em[col * n + k]; Em[col * n + k] = Em[piv * n + k]; Em[piv * n + k] = a;
      }
      __syncthreads();
    }
    for (int r = col + 1 + threadIdx.x; r < n; r += EXP_BLOCK) fv[r] = Dm[r * n + col] / Dm[col * n + col];
    __syncthreads();
    const int nr = n - col - 1;
    for (int e = threadIdx.x; e < nr * n; e += EXP_BLOCK) {
      int r = col + 1 + e / n, k = e % n;
      double f = fv[r];
      if (k >= col) Dm[r * n + k] -= f * Dm[col * n + k];
      Em[r * n + k] -= f * Em[col * n + k];
    }
    __syncthreads();
  }
  for (int r = n - 1; r >= 0; --r) {
    for (int k = threadIdx.x; k < n; k += EXP_BLOCK) {
      double acc = Em[r * n + k];
      for (int j = r + 1; j < n; ++j) acc -= Dm[r * n + j] * X[j * n + k];
      X[r * n + k] = acc / Dm[r * n + r];
    }
    __syncthreads();
  }
  for (int i = 0; i < s; ++i) {
    block_gemm(X, X, T2, n);
    for (int e = threadIdx.x; e < nn; e += EXP_BLOCK) X[e] = T2[e];
    __syncthreads();
  }
  for (int e = threadIdx.x; e < nn; e += EXP_BLOCK) out[(size_t)b * nn + e] = X[e];
}

hipError_t launch_expm_pade(int n, const double* Q, const double* t, const int32_t* s, int n_t, double* work,
                            double* out, uint32_t* err, hipStream_t stream) {
  hipLaunchKernelGGL(expm_pade_kernel, dim3(n_t), dim3(EXP_BLOCK), 0, stream, n, Q, t, s, work, out, err);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K2e: pruning with P(t_b); the result is shared by every sample, so one thread walks the tree once.
// ------------------------------------------------------------------------------------------------
__global__ void exp_pl_kernel(int n, int n_node, int n_tips, const UpStep* __restrict__ up, const double* __restrict__ P,
                              double* __restrict__ PL, int rescale) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int k = 0; k < n_node; ++k) {
    const UpStep st = up[k];
    const int ca = st.child[0] >= 0 ? st.child[0] + n_tips : ~st.child[0];
    const int cb = st.child[1] >= 0 ? st.child[1] + n_tips : ~st.child[1];
    const double* Pa = P + (size_t)st.edge[0] * n * n;
    const double* Pb = P + (size_t)st.edge[1] * n * n;
    const double* va = PL + (size_t)ca * n;
    const double* vb = PL + (size_t)cb * n;
    double* dst = PL + (size_t)(st.parent + n_tips) * n;
    for (int i = 0; i < n; ++i) {
      double a = Pa[i * n] * va[0];
      for (int j = 1; j < n; ++j) a += Pa[i * n + j] * va[j];
      double b = Pb[i * n] * vb[0];
      for (int j = 1; j < n; ++j) b += Pb[i * n + j] * vb[j];
      dst[i] = a * b;                                                           // :2903
    }
    if (rescale) {      // not in the reference: row / sum(row), the sampler-equivalent rescaling of makePLrcpp_bigtree :525
      double sum = dst[0];
      for (int i = 1; i < n; ++i) sum += dst[i];
      for (int i = 0; i < n; ++i) dst[i] = dst[i] / sum;
    }
  }
}

// the same pass level by level: a WAVE per node of one height level (children strictly below), lane i computing row i of both
// matrix-vector products with the sums in the order above -- identical values.  (One thread per node walked 2 n^2 dependent
// multiply-adds: 1.2 ms per level at 61 states, 21 of the 38 ms of a sumstatEXP call on 300 tips.)  n <= 64.
__global__ __launch_bounds__(64) void exp_pl_nodes_kernel(int n, int n_tips, const UpStep* __restrict__ up, const int32_t* __restrict__ order, int begin,
                                                           int end, const double* __restrict__ P, double* __restrict__ PL, int rescale) {
  __shared__ double s_row[64];
  __shared__ double s_sum;
  const int idx = begin + (int)blockIdx.x;
  if (idx >= end) return;
  const int i = threadIdx.x;
  const UpStep st = up[order[idx]];
  const int ca = st.child[0] >= 0 ? st.child[0] + n_tips : ~st.child[0];
  const int cb = st.child[1] >= 0 ? st.child[1] + n_tips : ~st.child[1];
  const double* Pa = P + (size_t)st.edge[0] * n * n;
  const double* Pb = P + (size_t)st.edge[1] * n * n;
  const double* va = PL + (size_t)ca * n;
  const double* vb = PL + (size_t)cb * n;
  double* dst = PL + (size_t)(st.parent + n_tips) * n;
  double r = 0.0;
  if (i < n) {
    double a = Pa[i * n] * va[0];
    for (int j = 1; j < n; ++j) a += Pa[i * n + j] * va[j];
    double b = Pb[i * n] * vb[0];
    for (int j = 1; j < n; ++j) b += Pb[i * n + j] * vb[j];
    r = a * b;                                                                  // :2903
  }
  if (rescale) {
    s_row[i] = r;
    __syncthreads();
    if (i == 0) {
      double sum = s_row[0];
      for (int q = 1; q < n; ++q) sum += s_row[q];
      s_sum = sum;
    }
    __syncthreads();
    r = r / s_sum;
  }
  if (i < n) dst[i] = r;
}

hipError_t launch_exp_pl_levels(int n, int n_tips, const UpStep* up, const int32_t* order, const std::vector<int32_t>& level_off,
                                const double* P, double* PL, int rescale, hipStream_t stream) {
  if (n > 64) return hipErrorInvalidValue;
  for (size_t l = 0; l + 1 < level_off.size(); ++l) {
    const int cnt = level_off[l + 1] - level_off[l];
    if (cnt > 0) hipLaunchKernelGGL(exp_pl_nodes_kernel, dim3(cnt), dim3(64), 0, stream, n, n_tips, up, order, level_off[l],
                                    level_off[l + 1], P, PL, rescale);
  }
  return hipGetLastError();
}

hipError_t launch_exp_pl(int n, int n_node, int n_tips, const UpStep* up, const double* P, double* PL, int rescale,
                         hipStream_t stream) {
  hipLaunchKernelGGL(exp_pl_kernel, dim3(1), dim3(64), 0, stream, n, n_node, n_tips, up, P, PL, rescale);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// log p(y|Q) by matrix exponentiation (the DIC drivers: PPmakePLD :3158-3178, PPmakePLksD :3268-3297): pruning with
// P(t_b), every internal row divided by its sum, the log scale factors accumulated in the caller's pruningwise order
// (`up` is built from nen), then log(sum_j PL[root,j] pid_j) + S.
// The tree is walked level by level (a lane per node of one height level; `order` lists positions of `up` grouped by
// level), each node leaving log(scale factor) in logs[position]; one thread then adds the logs in nen order, so the sum
// is the reference's left-to-right sum whatever the launch geometry.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void exp_pl_node(int n, int n_tips, const UpStep* __restrict__ up, const int32_t* __restrict__ order, int idx,
                                            const double* __restrict__ P, double* __restrict__ PL, double* __restrict__ logs) {
  const int k = order[idx];
  const UpStep st = up[k];
  const int ca = st.child[0] >= 0 ? st.child[0] + n_tips : ~st.child[0];
  const int cb = st.child[1] >= 0 ? st.child[1] + n_tips : ~st.child[1];
  const double* Pa = P + (size_t)st.edge[0] * n * n;
  const double* Pb = P + (size_t)st.edge[1] * n * n;
  const double* va = PL + (size_t)ca * n;
  const double* vb = PL + (size_t)cb * n;
  double* dst = PL + (size_t)(st.parent + n_tips) * n;
  double sm = 0.0;
  for (int i = 0; i < n; ++i) {
    double a = Pa[i * n] * va[0];
    for (int j = 1; j < n; ++j) a += Pa[i * n + j] * va[j];
    double b = Pb[i * n] * vb[0];
    for (int j = 1; j < n; ++j) b += Pb[i * n + j] * vb[j];
    const double r = a * b;
    dst[i] = r;
    sm = (i == 0) ? r : sm + r;
  }
  logs[k] = phm_log(sm);
  for (int i = 0; i < n; ++i) dst[i] = dst[i] / sm;
}

__global__ void exp_pl_level_kernel(int n, int n_tips, const UpStep* __restrict__ up, const int32_t* __restrict__ order,
                                    int begin, int end, const double* __restrict__ P, double* __restrict__ PL,
                                    double* __restrict__ logs) {
  const int idx = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= end) return;
  exp_pl_node(n, n_tips, up, order, idx, P, PL, logs);
}

// A run of consecutive levels of at most EXP_RUN_BLOCK nodes each in ONE launch (one workgroup, a workgroup-scope fence and a
// barrier between levels): the DIC drivers call this every iteration, and on a 1 000-tip tree 22 of the 23 levels are that narrow.
constexpr int EXP_RUN_BLOCK = 1024, EXP_RUN_LEVELS = 63;
struct ExpLevelRun { int32_t off[EXP_RUN_LEVELS + 1]; int32_t n_levels; };
__global__ __launch_bounds__(EXP_RUN_BLOCK) void exp_pl_run_kernel(int n, int n_tips, const UpStep* __restrict__ up,
                                                                   const int32_t* __restrict__ order, ExpLevelRun run,
                                                                   const double* P, double* PL, double* logs) {
  for (int l = 0; l < run.n_levels; ++l) {
    for (int idx = run.off[l] + (int)threadIdx.x; idx < run.off[l + 1]; idx += EXP_RUN_BLOCK) exp_pl_node(n, n_tips, up, order, idx, P, PL, logs);
    __threadfence_block();
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void exp_pl_logsum_kernel(int n, int n_node, const double* __restrict__ logs,
                                                            const double* __restrict__ PL, const double* __restrict__ pid,
                                                            int root_node, double* __restrict__ out_ll) {
  __shared__ double chunk[4096];
  double S = 0;
  for (int base = 0; base < n_node; base += 4096) {
    const int len = min(4096, n_node - base);
    for (int i = threadIdx.x; i < len; i += 256) chunk[i] = logs[base + i];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int i = 0; i < len; ++i) S = S + chunk[i];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double X = 0;
    for (int j = 0; j < n; ++j) X = X + PL[(size_t)root_node * n + j] * pid[j];
    *out_ll = phm_log(X) + S;
  }
}

hipError_t launch_exp_pl_loglik(int n, int n_node, int n_tips, const UpStep* up, const int32_t* order,
                                const std::vector<int32_t>& level_off, const double* P, double* PL, double* logs,
                                const double* pid, int root_node, double* out_ll, hipStream_t stream) {
  const int L = (int)level_off.size() - 1;
  auto narrow = [&](int l) { return level_off[l + 1] - level_off[l] <= EXP_RUN_BLOCK; };
  for (int l = 0; l < L;) {
    if (narrow(l)) {
      ExpLevelRun run;
      run.n_levels = 0;
      while (l < L && narrow(l) && run.n_levels < EXP_RUN_LEVELS) { run.off[run.n_levels++] = level_off[l]; ++l; }
      run.off[run.n_levels] = level_off[l];
      hipLaunchKernelGGL(exp_pl_run_kernel, dim3(1), dim3(EXP_RUN_BLOCK), 0, stream, n, n_tips, up, order, run, P, PL, logs);
      continue;
    }
    const int cnt = level_off[l + 1] - level_off[l];
    hipLaunchKernelGGL(exp_pl_level_kernel, dim3((cnt + 63) / 64), dim3(64), 0, stream, n, n_tips, up, order, level_off[l],
                       level_off[l + 1], P, PL, logs);
    ++l;
  }
  hipLaunchKernelGGL(exp_pl_logsum_kernel, dim3(1), dim3(256), 0, stream, n, n_node, logs, PL, pid, root_node, out_ll);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K5: one lane per i.i.d. sample (treesampleEXP :2977-2996)
// ------------------------------------------------------------------------------------------------
// sampleOnce, src/phylomap.cpp:81-90: no sort, per-element division, strict '<'
template <int NS>
__device__ __forceinline__ int sample_once(const double (&w)[NS], double u, uint32_t& err) {
  double total = w[0];
#pragma unroll
  for (int j = 1; j < NS; ++j) total += w[j];
  double cum = 0.0;
  int idx = NS;
  bool found = false;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    cum += w[j] / total;
    if (!found && u < cum) { idx = j; found = true; }
  }
  if (!found) { err |= DERR_SAMPLEONCE; idx = NS - 1; }
  return idx;
}

template <int NS>
__global__ __launch_bounds__(EXP_BLOCK) void exp_sample_kernel(ExpParams<NS> p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NCNT = NS * (NS - 1);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * (EXP_BLOCK / 64) + wave;
  double* s_col = reinterpret_cast<double*>(smem);                       // [UNIF_CAP+1][NS][NS]
  double* s_B2 = s_col + (UNIF_CAP + 1) * NS * NS;                       // [NS][NS]
  double* s_dw = s_B2 + NS * NS + (size_t)wave * NS * 64;                // [NS][64]
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_B2 + NS * NS + (size_t)(EXP_BLOCK / 64) * NS * 64) +
                    (size_t)wave * NCNT * 64;
  for (int i = threadIdx.x; i < (UNIF_CAP + 1) * NS * NS; i += EXP_BLOCK) s_col[i] = p.colpow[i];
  if (threadIdx.x < NS * NS) s_B2[threadIdx.x] = p.B2[threadIdx.x];
  __syncthreads();
  if (tile >= p.n_tiles) return;

  const int it = tile * 64 + lane;           // sample index = RNG iteration word
  const bool valid = it < p.N;
  uint32_t err = 0;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  double* __restrict__ tms = p.times + (size_t)tile * UNIF_CAP * 64;
#pragma unroll
  for (int c = 0; c < NS; ++c) s_dw[c * 64 + lane] = 0.0;
#pragma unroll
  for (int c = 0; c < NCNT; ++c) s_cnt[c * 64 + lane] = 0u;

  {
    double pr[NS];
    const double* plr = p.PL + (size_t)(p.root + p.n_tips) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) pr[c] = p.pid[c] * plr[c];                   // :2926
    double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
    nst[p.root * 64 + lane] = (uint8_t)sample_cat<NS>(pr, u, err);            // :2934
  }

  for (int k = 0; k < p.n_edge; ++k) {
    const DownStep ds = p.down[k];
    const int b = ds.edge;
    const int a = nst[ds.parent * 64 + lane];
    const double* Pb = p.P + (size_t)b * NS * NS;
    int e;
    if (ds.child >= 0) {
      double pr[NS];
      const double* plc = p.PL + (size_t)(ds.child + p.n_tips) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) pr[c] = Pb[a * NS + c] * plc[c];           // :2953
      double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(ds.child + p.n_tips), 0);
      e = sample_cat<NS>(pr, u, err);                                         // :2956
      nst[ds.child * 64 + lane] = (uint8_t)e;
    } else {
      e = p.tips[~ds.child];
    }

    // ---- newunifSample(a, e, t_b, P_b[a,e]) :93-208 ----
    const double tb = p.edge_length[b];
    const double transProb = Pb[a * NS + e];
    Stream sr;
    sr.open(ENT_BUNIF | (uint32_t)b, (uint32_t)(it + p.it0), p.replica, p.seed_lo, p.seed_hi);
    uint32_t dr = 0;
    const double rU = sr.draw(dr++);                                          // :103
    const double lam = p.poisson_rate * tb;
    double pk = phm_exp(-lam);
    double cum = 0.0;
    if (a == e) cum = pk / transProb;                                         // :107
    bool notExceed = !(cum > rU);
    int nj = 0;
    bool capped = false;
    while (notExceed) {
      nj++;
      if (nj > UNIF_CAP) { capped = true; break; }                            // :120
      pk = pk * lam / (double)nj;
      double nextProb = pk * s_col[(nj * NS + e) * NS + a] / transProb;       // :127-128
      cum += nextProb;
      if (cum > rU) notExceed = false;
    }
    if (capped) { err |= DERR_UNIF_CAP; continue; }
    if (nj == 0 || (nj == 1 && a == e)) {                                     // :138
      s_dw[a * 64 + lane] += tb - 0.0;
    } else if (nj == 1) {                                                     // :144
      double tj = tb * sr.draw(dr++);                                         // :147
      s_dw[a * 64 + lane] += tj - 0.0;
      s_dw[e * 64 + lane] += tb - tj;
      s_cnt[(a * (NS - 1) + (e > a ? e - 1 : e)) * 64 + lane] += 1u;
    } else {
      for (int i = 0; i < nj; ++i) {                                          // :151-152 jump times, ascending
        double v = tb * sr.draw(dr++);
        int j = i - 1;
        while (j >= 0) {
          double tj = tms[j * 64 + lane];
          if (!(tj > v)) break;
          tms[(j + 1) * 64 + lane] = tj;
          --j;
        }
        tms[(j + 1) * 64 + lane] = v;
      }
      int prev = a, sprev = a;
      double tprev = 0.0;
      for (int i = 1; i <= nj; ++i) {
        int di = e;
        if (i < nj) {                                                         // :158-160
          double pr[NS];
          const double* beta = s_col + ((nj - i) * NS + e) * NS;
          const double* row = s_B2 + prev * NS;
#pragma unroll
          for (int c = 0; c < NS; ++c) pr[c] = row[c] * beta[c];
          di = sample_once<NS>(pr, sr.draw(dr++), err);
        }
        if (prev != di) {                                                     // :168-173 drop virtual jumps
          double ti = tms[(i - 1) * 64 + lane];
          s_dw[sprev * 64 + lane] += ti - tprev;
          s_cnt[(sprev * (NS - 1) + (di > sprev ? di - 1 : di)) * 64 + lane] += 1u;
          tprev = ti; sprev = di;
        }
        prev = di;
      }
      s_dw[sprev * 64 + lane] += tb - tprev;
    }
  }

  if (valid) {
#pragma unroll
    for (int c = 0; c < NS; ++c) p.out[(size_t)c * p.N + it] = s_dw[c * 64 + lane];
#pragma unroll
    for (int c = 0; c < NCNT; ++c) p.out[(size_t)(NS + c) * p.N + it] = (double)s_cnt[c * 64 + lane];
    if (err) atomicOr(p.err, err);
  }
}

template <int NS>
hipError_t launch_exp_sample(const ExpParams<NS>& p, hipStream_t stream) {
  constexpr int W = EXP_BLOCK / 64;
  size_t lds = sizeof(double) * ((size_t)(UNIF_CAP + 1) * NS * NS + NS * NS + (size_t)W * NS * 64) +
               sizeof(uint32_t) * (size_t)W * NS * (NS - 1) * 64;
  hipLaunchKernelGGL(exp_sample_kernel<NS>, dim3((p.n_tiles + W - 1) / W), dim3(EXP_BLOCK), lds, stream, p);
  return hipGetLastError();
}

template hipError_t launch_exp_sample<2>(const ExpParams<2>&, hipStream_t);
template hipError_t launch_exp_sample<3>(const ExpParams<3>&, hipStream_t);
template hipError_t launch_exp_sample<4>(const ExpParams<4>&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// K5 for 5 <= n <= 64 states (the tutorial's 20-state tridiagonal Q, C4's 61 states): same algorithm and draw order as
// exp_sample_kernel, one LANE per i.i.d. sample, with the n-vectors streamed instead of held in registers: every
// categorical draw makes two passes over its weights (total, then running sum), the weights being recomputed from
// P_b[ps,:] (.) PL[child,:] or B[prev,:] (.) B^(k-i) e_end on the fly.  Statistics go straight to the result matrix
// (column-major N x cols, zeroed by the host): lanes are consecutive samples, so a column access is coalesced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EXP_BLOCK) void exp_wide_kernel(ExpWideParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = p.n_states;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * (EXP_BLOCK / 64) + wave;
  double* s_B2 = reinterpret_cast<double*>(smem);                            // [n][n]
  for (int i = threadIdx.x; i < n * n; i += EXP_BLOCK) s_B2[i] = p.B2[i];
  __syncthreads();
  if (tile >= p.n_tiles) return;

  const int it = tile * 64 + lane;
  const bool valid = it < p.N;
  const int itc = valid ? it : p.N - 1;          // padded lanes compute on a valid column and never store
  uint32_t err = 0;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  double* __restrict__ tms = p.times + (size_t)tile * UNIF_CAP * 64;
  double* __restrict__ orow = p.out + itc;                                    // + col * N
  auto stat_add = [&](int col, double v) { if (valid) orow[(size_t)col * p.N] += v; };

  // first j with u*sum(w) <= w_0+..+w_j, w_c = a[c]*b[c]   (sample_cat, index order)
  auto draw_node = [&](const double* a, const double* b, double u) -> int {
    double total = a[0] * b[0];
    for (int c = 1; c < n; ++c) total += a[c] * b[c];
    if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
    const double thr = u * total;
    double cum = a[0] * b[0];
    int idx = (thr <= cum) ? 0 : 1;
    for (int c = 1; c < n; ++c) { cum += a[c] * b[c]; idx += (thr <= cum) ? 0 : 1; }
    return idx < n ? idx : n - 1;
  };

  {
    const double* plr = p.PL + (size_t)(p.root + p.n_tips) * n;
    double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
    nst[p.root * 64 + lane] = (uint8_t)draw_node(p.pid, plr, u);                // :2926-2934
  }

  for (int k = 0; k < p.n_edge; ++k) {
    const DownStep ds = p.down[k];
    const int b = ds.edge;
    const int a = nst[ds.parent * 64 + lane];
    const double* Pb = p.P + (size_t)b * n * n;
    int e;
    if (ds.child >= 0) {
      double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(ds.child + p.n_tips), 0);
      e = draw_node(Pb + (size_t)a * n, p.PL + (size_t)(ds.child + p.n_tips) * n, u);   // :2953-2956
      nst[ds.child * 64 + lane] = (uint8_t)e;
    } else {
      e = p.tips[~ds.child];
    }

    // ---- newunifSample(a, e, t_b, P_b[a,e]) :93-208 ----
    const double tb = p.edge_length[b];
    const double transProb = Pb[(size_t)a * n + e];
    Stream sr;
    sr.open(ENT_BUNIF | (uint32_t)b, (uint32_t)(it + p.it0), p.replica, p.seed_lo, p.seed_hi);
    uint32_t dr = 0;
    const double rU = sr.draw(dr++);
    const double lam = p.poisson_rate * tb;
    double pk = phm_exp(-lam);
    double cum = 0.0;
    if (a == e) cum = pk / transProb;
    bool notExceed = !(cum > rU);
    int nj = 0;
    bool capped = false;
    while (notExceed) {
      nj++;
      if (nj > UNIF_CAP) { capped = true; break; }
      pk = pk * lam / (double)nj;
      double nextProb = pk * p.colpow[((size_t)nj * n + e) * n + a] / transProb;
      cum += nextProb;
      if (cum > rU) notExceed = false;
    }
    if (capped) { err |= DERR_UNIF_CAP; continue; }
    auto count = [&](int from, int to) { stat_add(n + from * (n - 1) + (to > from ? to - 1 : to), 1.0); };
    if (nj == 0 || (nj == 1 && a == e)) {
      stat_add(a, tb - 0.0);
    } else if (nj == 1) {
      double tj = tb * sr.draw(dr++);
      stat_add(a, tj - 0.0);
      stat_add(e, tb - tj);
      count(a, e);
    } else {
      for (int i = 0; i < nj; ++i) {
        double v = tb * sr.draw(dr++);
        int j = i - 1;
        while (j >= 0) {
          double tj = tms[j * 64 + lane];
          if (!(tj > v)) break;
          tms[(j + 1) * 64 + lane] = tj;
          --j;
        }
        tms[(j + 1) * 64 + lane] = v;
      }
      int prev = a, sprev = a;
      double tprev = 0.0;
      for (int i = 1; i <= nj; ++i) {
        int di = e;
        if (i < nj) {                                                         // sampleOnce :81-90, :158-160
          const double* beta = p.colpow + ((size_t)(nj - i) * n + e) * n;
          const double* row = s_B2 + prev * n;
          double total = row[0] * beta[0];
          for (int c = 1; c < n; ++c) total += row[c] * beta[c];
          const double u = sr.draw(dr++);
          double cw = 0.0;
          int pick = n;
          for (int c = 0; c < n; ++c) {
            cw += (row[c] * beta[c]) / total;
            if (pick == n && u < cw) pick = c;
          }
          if (pick == n) { err |= DERR_SAMPLEONCE; pick = n - 1; }
          di = pick;
        }
        if (prev != di) {
          double ti = tms[(i - 1) * 64 + lane];
          stat_add(sprev, ti - tprev);
          count(sprev, di);
          tprev = ti; sprev = di;
        }
        prev = di;
      }
      stat_add(sprev, tb - tprev);
    }
  }
  if (valid && err) atomicOr(p.err, err);
}

hipError_t launch_exp_wide(const ExpWideParams& p, hipStream_t stream) {
  constexpr int W = EXP_BLOCK / 64;
  size_t lds = sizeof(double) * (size_t)p.n_states * p.n_states;
  hipLaunchKernelGGL(exp_wide_kernel, dim3((p.n_tiles + W - 1) / W), dim3(EXP_BLOCK), lds, stream, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sumstatEXP, one wavefront per (tile of 64 samples, item): see ExpTilesParams (phm_exp.h).  Runtime state count (2..64),
// tables through L1/L2 as in exp_wide_kernel; arithmetic and draw order are those of exp_sample_kernel / exp_wide_kernel.
// ------------------------------------------------------------------------------------------------
namespace {

// first j with u*sum(w) <= w_0+..+w_j, w_c = a[c]*b[c]   (sample_cat, index order)
__device__ __forceinline__ int exp_draw_node(const double* __restrict__ a, const double* __restrict__ b, int n, double u, uint32_t& err) {
  double total = a[0] * b[0];
  for (int c = 1; c < n; ++c) total += a[c] * b[c];
  if (!(total > 0.0) || isinf(total)) err |= DERR_ZERO_PROB;
  const double thr = u * total;
  double cum = a[0] * b[0];
  int idx = (thr <= cum) ? 0 : 1;
  for (int c = 1; c < n; ++c) { cum += a[c] * b[c]; idx += (thr <= cum) ? 0 : 1; }
  return idx < n ? idx : n - 1;
}

__global__ __launch_bounds__(EXP_BLOCK) void exp_tiles_root_kernel(ExpTilesParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (EXP_BLOCK / 64) + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;
  const int it = tile * 64 + lane;
  uint32_t err = 0;
  const double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(p.root + p.n_tips), 0);
  p.nstate[((size_t)tile * p.n_node + p.root) * 64 + lane] =
      (uint8_t)exp_draw_node(p.pid, p.PL + (size_t)(p.root + p.n_tips) * p.n_states, p.n_states, u, err);      // :2926-2934
  if (err && it < p.N) atomicOr(p.err, err);
}

__global__ __launch_bounds__(EXP_BLOCK) void exp_tiles_node_kernel(ExpTilesParams p, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * (EXP_BLOCK / 64) + (threadIdx.x >> 6);
  const int n_lvl = end - begin;
  if (item >= n_lvl * p.n_tiles) return;
  const int n = p.n_states;
  const int tile = item / n_lvl;
  const DownStep ds = p.down[p.node_order[begin + item % n_lvl]];
  const int it = tile * 64 + lane;
  uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
  const int a = nst[ds.parent * 64 + lane];
  uint32_t err = 0;
  const double u = stream_u(p.seed_lo, p.seed_hi, p.replica, (uint32_t)(it + p.it0), ENT_NODE | (uint32_t)(ds.child + p.n_tips), 0);
  const int e = exp_draw_node(p.P + ((size_t)ds.edge * n + a) * n, p.PL + (size_t)(ds.child + p.n_tips) * n, n, u, err);   // :2953-2956
  nst[ds.child * 64 + lane] = (uint8_t)e;
  if (err && it < p.N) atomicOr(p.err, err);
}

// newunifSample(a, e, t_b, P_b[a,e]) :93-208 for the 64 samples of a tile; persistent waves over (tile, group of `group` consecutive
// branches in pre-order).
//  * Jump times (`t * runif(k)` sorted ascending, :147-152): up to EXP_KL per lane sit in LDS ([slot][lane], conflict-free); a lane
//    with more jumps on a branch uses its wave's scratch rows in global memory (the round-2/3 form for every lane: the insertion
//    sort was a chain of dependent global reads -- the kernel waited 0.8 of its time and moved 4x its algorithmic bytes).
//  * NS > 0 (2..4 states, compile-time): the statistics of the group's branches collect in registers (n dwell sums in 64-bit fixed
//    point, n(n-1) counters) and reach the per-sample accumulators once per group; NS = 0 (runtime n up to 64): one atomic per segment.
constexpr int EXP_KL = 12;

template <int NS>
__global__ __launch_bounds__(EXP_BLOCK) void exp_tiles_branch_kernel(ExpTilesParams p, int group) {
  __shared__ double s_tm[EXP_BLOCK / 64][EXP_KL][64];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int wslot = blockIdx.x * (EXP_BLOCK / 64) + wv;
  const int n = NS > 0 ? NS : p.n_states;
  const int n_groups = (p.n_edge + group - 1) / group;
  const int64_t items = (int64_t)n_groups * p.n_tiles;
  const size_t npad = (size_t)p.n_tiles * 64;
  double* __restrict__ gtm = p.times + (size_t)wslot * UNIF_CAP * 64;
  constexpr int NA = NS > 0 ? NS : 1, NC = NS > 0 ? NS * (NS - 1) : 1;
  uint32_t err = 0;
  for (int64_t item = wslot; item < items; item += (int64_t)gridDim.x * (EXP_BLOCK / 64)) {
    const int tile = (int)(item % p.n_tiles);
    const int q0 = (int)(item / p.n_tiles) * group, q1 = min(q0 + group, p.n_edge);
    const int it = tile * 64 + lane;
    const bool valid = it < p.N;
    const uint8_t* __restrict__ nst = p.nstate + (size_t)tile * p.n_node * 64;
    unsigned long long acc_dw[NA];
    uint32_t acc_ct[NC];
#pragma unroll
    for (int c = 0; c < NA; ++c) acc_dw[c] = 0ull;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc_ct[c] = 0u;
    auto stat_add = [&](int col, double v) {           // dwell: fixed point, exact in any order
      const unsigned long long fx = (unsigned long long)__double2ll_rn(v * p.fx_scale);
      if (NS > 0) {
#pragma unroll
        for (int c = 0; c < NA; ++c) acc_dw[c] += (col == c) ? fx : 0ull;
      } else if (valid) atomicAdd(p.dwfx + (size_t)col * npad + it, fx);
    };
    auto count = [&](int from, int to) {
      const int col = from * (n - 1) + (to > from ? to - 1 : to);
      if (NS > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc_ct[c] += (col == c) ? 1u : 0u;
      } else if (valid) atomicAdd(p.cnt + (size_t)col * npad + it, 1u);
    };
    // end states of the first branch of the group; those of the next branch are requested while the current one is sampled
    DownStep ds = p.down[q0];
    int a = nst[ds.parent * 64 + lane];
    int e = ds.child >= 0 ? (int)nst[ds.child * 64 + lane] : (int)p.tips[~ds.child];
    for (int q = q0; q < q1; ++q) {
      const int b = ds.edge;
      const int a_cur = a, e_cur = e;
      if (q + 1 < q1) {
        ds = p.down[q + 1];
        a = nst[ds.parent * 64 + lane];
        e = ds.child >= 0 ? (int)nst[ds.child * 64 + lane] : (int)p.tips[~ds.child];
      }
      const double* Pb = p.P + (size_t)b * n * n;
      const double tb = p.edge_length[b];
      const double transProb = Pb[(size_t)a_cur * n + e_cur];
      Stream sr;
      sr.open(ENT_BUNIF | (uint32_t)b, (uint32_t)(it + p.it0), p.replica, p.seed_lo, p.seed_hi);
      uint32_t dr = 0;
      const double rU = sr.draw(dr++);                                            // :103
      const double lam = p.poisson_rate * tb;
      double pk = phm_exp(-lam);
      double cum = 0.0;
      if (a_cur == e_cur) cum = pk / transProb;                                   // :107
      bool notExceed = !(cum > rU);
      int nj = 0;
      bool capped = false;
      const double* __restrict__ cp = p.colpow + (size_t)e_cur * n + a_cur;       // (B^k e_e)[a] at cp[k n n]
      double cnext = cp[(size_t)n * n];
      while (notExceed) {
        nj++;
        if (nj > UNIF_CAP) { capped = true; break; }                              // :120
        const double ck = cnext;
        if (nj < UNIF_CAP) cnext = cp[(size_t)(nj + 1) * n * n];                   // the next term's table entry is on its way
        pk = pk * lam / (double)nj;
        const double nextProb = pk * ck / transProb;                              // :127-128
        cum += nextProb;
        if (cum > rU) notExceed = false;
      }
      if (capped) { if (valid) err |= DERR_UNIF_CAP; continue; }
      if (nj == 0 || (nj == 1 && a_cur == e_cur)) {                               // :138
        stat_add(a_cur, tb - 0.0);
      } else if (nj == 1) {                                                       // :144
        const double tj = tb * sr.draw(dr++);
        stat_add(a_cur, tj - 0.0);
        stat_add(e_cur, tb - tj);
        count(a_cur, e_cur);
      } else {
        const bool in_lds = nj <= EXP_KL;
        double* __restrict__ ltm = &s_tm[wv][0][lane];
        auto tm_get = [&](int k) -> double { double v; if (in_lds) v = ltm[k * 64]; else v = gtm[k * 64 + lane]; return v; };
        auto tm_put = [&](int k, double v) { if (in_lds) ltm[k * 64] = v; else gtm[k * 64 + lane] = v; };
        for (int i = 0; i < nj; ++i) {                                            // :151-152 jump times, ascending
          const double v = tb * sr.draw(dr++);
          int j = i - 1;
          while (j >= 0) {
            const double tj = tm_get(j);
            if (!(tj > v)) break;
            tm_put(j + 1, tj);
            --j;
          }
          tm_put(j + 1, v);
        }
        int prev = a_cur, sprev = a_cur;
        double tprev = 0.0;
        for (int i = 1; i <= nj; ++i) {
          int di = e_cur;
          if (i < nj) {                                                           // sampleOnce :81-90, :158-160
            const double* beta = p.colpow + ((size_t)(nj - i) * n + e_cur) * n;
            const double* row = p.B2 + (size_t)prev * n;
            double total = row[0] * beta[0];
            for (int c = 1; c < n; ++c) total += row[c] * beta[c];
            const double u = sr.draw(dr++);
            double cw = 0.0;
            int pick = n;
            for (int c = 0; c < n; ++c) {
              cw += (row[c] * beta[c]) / total;
              if (pick == n && u < cw) pick = c;
            }
            if (pick == n) { if (valid) err |= DERR_SAMPLEONCE; pick = n - 1; }
            di = pick;
          }
          if (prev != di) {                                                       // :168-173 drop virtual jumps
            const double ti = tm_get(i - 1);
            stat_add(sprev, ti - tprev);
            count(sprev, di);
            tprev = ti; sprev = di;
          }
          prev = di;
        }
        stat_add(sprev, tb - tprev);
      }
    }
    if (NS > 0 && valid) {
#pragma unroll
      for (int c = 0; c < NA; ++c) if (acc_dw[c]) atomicAdd(p.dwfx + (size_t)c * npad + it, acc_dw[c]);
#pragma unroll
      for (int c = 0; c < NC; ++c) if (acc_ct[c]) atomicAdd(p.cnt + (size_t)c * npad + it, acc_ct[c]);
    }
  }
  if (err) atomicOr(p.err, err);
}

__global__ void exp_tiles_finish_kernel(ExpTilesParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = p.n_states, cols = n + n * (n - 1);
  if (gid >= (int64_t)cols * p.N) return;
  const int c = (int)(gid / p.N), it = (int)(gid % p.N);
  const size_t npad = (size_t)p.n_tiles * 64;
  p.out[(size_t)c * p.N + it] = c < n ? (double)(long long)p.dwfx[(size_t)c * npad + it] * p.fx_inv
                                      : (double)p.cnt[(size_t)(c - n) * npad + it];
}

}  // namespace

hipError_t launch_exp_tiles(const ExpTilesParams& p, const std::vector<int32_t>& level_off, int branch_blocks, hipStream_t stream) {
  constexpr int W = EXP_BLOCK / 64;
  hipLaunchKernelGGL(exp_tiles_root_kernel, dim3((p.n_tiles + W - 1) / W), dim3(EXP_BLOCK), 0, stream, p);
  for (size_t l = 0; l + 1 < level_off.size(); ++l) {
    const int cnt = level_off[l + 1] - level_off[l];
    if (cnt > 0) hipLaunchKernelGGL(exp_tiles_node_kernel, dim3((unsigned)(((int64_t)cnt * p.n_tiles + W - 1) / W)), dim3(EXP_BLOCK), 0, stream, p, level_off[l], level_off[l + 1]);
  }
  // branches per wave-item: as many as still leave every SIMD a few waves (an R call with N = 1 000 samples has 16 tiles)
  const int group = (int)std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)p.n_edge * p.n_tiles / 8192));
  if (p.n_states == 2) hipLaunchKernelGGL(exp_tiles_branch_kernel<2>, dim3(branch_blocks), dim3(EXP_BLOCK), 0, stream, p, group);
  else if (p.n_states == 3) hipLaunchKernelGGL(exp_tiles_branch_kernel<3>, dim3(branch_blocks), dim3(EXP_BLOCK), 0, stream, p, group);
  else if (p.n_states == 4) hipLaunchKernelGGL(exp_tiles_branch_kernel<4>, dim3(branch_blocks), dim3(EXP_BLOCK), 0, stream, p, group);
  else hipLaunchKernelGGL(exp_tiles_branch_kernel<0>, dim3(branch_blocks), dim3(EXP_BLOCK), 0, stream, p, group);
  const int64_t cells = (int64_t)(p.n_states + p.n_states * (p.n_states - 1)) * p.N;
  hipLaunchKernelGGL(exp_tiles_finish_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K1 on the matrix cores: P_b = |(L diag(exp(d t_b))) R| as a batched small GEMM with MFMA f64 16x16x4.
// For 16 < n <= 64 (C4's 61-state Q pads to 64).  R and L are the same for every branch, so each wave keeps its
// MFMA fragments of both in registers for the whole batch (R: its 16-column block; L: all four row blocks) and a
// matrix costs 64 exp() + 64 v_mfma per wave.  Fragment maps (cdna_hip_programming.md section 3, f64 form):
//   A: lane l holds A[row = l&15][k = l>>4];  B: lane l holds B[k = l>>4][col = l&15];
//   C/D: col = l&15, row = (l>>4) + 4*reg.
// The MFMA accumulates each 4-deep k-slice with fused multiply-adds, so results differ from the exact kernel
// (unfused left-to-right sums) in the last bits: use expm_eigen_kernel where bit-parity with the oracle matters
// (the sumstatEXP sampler does); this kernel is the throughput path of phm_expm_eigen_mfma.
// ------------------------------------------------------------------------------------------------
using d4_t = __attribute__((ext_vector_type(4))) double;

__global__ __launch_bounds__(256) void expm_eigen_mfma_kernel(int n, const double* __restrict__ L,
                                                              const double* __restrict__ R,
                                                              const double* __restrict__ dvals,
                                                              const double* __restrict__ t, int n_t,
                                                              double* __restrict__ out) {
  __shared__ double s_e[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rb = (n + 15) >> 4;                 // 16-wide row / column blocks in use (1..4)
  const int lr = lane & 15, lk = lane >> 4;
  double Lf[4][16], Rf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 4 * s + lk, col = 16 * w + lr;
    Rf[s] = (k < n && col < n) ? R[k * n + col] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * i + lr;
      Lf[i][s] = (row < n && k < n) ? L[row * n + k] : 0.0;
    }
  }
  for (int b = blockIdx.x; b < n_t; b += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < 64) s_e[threadIdx.x] = ((int)threadIdx.x < n) ? phm_exp(dvals[threadIdx.x] * t[b]) : 0.0;   // :2966
    __syncthreads();
    if (w < rb) {
      double ek[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) ek[s] = s_e[4 * s + lk];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < rb) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lf[i][s] * ek[s], Rf[s], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * i + lk + 4 * q, col = 16 * w + lr;
            if (row < n && col < n) out[(size_t)b * n * n + row * n + col] = fabs(acc[q]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K1' on the matrix cores: expmat(Q t_b) = Pade(6) + s squarings as a batched small GEMM chain (16 < n <= 64).
// One workgroup (4 waves) per matrix; wave w owns the 16-column block w of every intermediate.  With A as the left
// operand, the powers X <- A X never leave registers: an MFMA result tile (col = l&15, row = (l>>4) + 4 reg) is
// exactly the B-operand slice layout (k = 4 s + (l>>4)), so register `reg` of row block kb feeds k-step 4 kb + reg.
// Q is staged in LDS once per workgroup (every matrix of the batch is Q times a scalar); solve(D, E) is a block Gauss-Jordan
// elimination in registers (16 x 16 diagonal blocks inverted in LDS by one wave); the squarings re-read the result from LDS in
// both operand layouts.  78.8 KB of LDS and 256 registers: two workgroups per CU.
// Fused k-slices: agrees with the exact kernel to rounding (tests: <= 2e-13), not bit for bit.  A matrix with a small pivot
// in a diagonal block is flagged in `bad` and left to the pivoted kernel by the caller.
// ------------------------------------------------------------------------------------------------
constexpr int PADE_LDP = 66;      // LDS row stride (doubles): the A-operand reads (row = lane & 15, k = lane >> 4) hit 32 distinct banks per half wave

__global__ __launch_bounds__(256, 2) void expm_pade_mfma_kernel(int n, const double* __restrict__ Q,
                                                             const double* __restrict__ t,
                                                             const int32_t* __restrict__ sq, int n_t,
                                                             double* __restrict__ out, int32_t* __restrict__ bad, double piv_min) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* sQ = reinterpret_cast<double*>(smem);                  // [64][PADE_LDP]  Q, zero-padded to 64 x 64
  double* sE = sQ + 64 * PADE_LDP;                               // [64][PADE_LDP]  Y = solve(D, E), then its squares
  __shared__ int s_bad;
  __shared__ double sCol[64 * 17], sA[16 * 17], sW[16 * 17];     // block Gauss-Jordan: a column block of D, the diagonal block, its inverse
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
  const int rb = (n + 15) >> 4;
  const int lr = lane & 15, lk = lane >> 4;
  const int mycol = 16 * w + lr;

  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    sQ[r * PADE_LDP + c] = (r < n && c < n) ? Q[r * n + c] : 0.0;
  }
  __syncthreads();

  for (int b = blockIdx.x; b < n_t; b += gridDim.x) {
    const int s = sq[b];
    const double sc = t[b] * ldexp(1.0, -s);                     // A = Q t / 2^s (the power of two is exact: same bits as (Q t) / 2^s)
    d4_t X[4], Em[4], Dm[4], T[4];
    {   // every wave fills its column block (blocks beyond n carry the identity)
      // A^k = sc^k Q^k: the products run on the unscaled Q straight from LDS (A-operand fragments read as they are used, so no
      // 128 registers of fragments: the kernel stays within 256 and two workgroups share a CU), each result scaled once
      double c = 0.5;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * i + lk + 4 * q;
          const double a = sQ[row * PADE_LDP + mycol] * sc;
          const double dg = (row == mycol) ? 1.0 : 0.0;
          X[i][q] = a;
          Em[i][q] = c * a + dg;
          Dm[i][q] = dg - c * a;
        }
      bool positive = true;
      for (int pw = 2; pw <= 6; ++pw) {
        c = c * (double)(6 - pw + 1) / (double)(pw * (2 * 6 - pw + 1));
        int frag = lr * PADE_LDP + lk;
        asm volatile("" : "+v"(frag));                 // a fresh value every round: the fragment reads stay in the loop (hoisted, they are 128 registers)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
          if (i < rb && w < rb) {
#pragma unroll
            for (int sI = 0; sI < 16; ++sI)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[frag + 16 * i * PADE_LDP + 4 * sI], X[sI >> 2][sI & 3], acc, 0, 0, 0);
          }
          T[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          X[i] = sc * T[i];
          Em[i] += c * X[i];
          if (positive) Dm[i] += c * X[i]; else Dm[i] -= c * X[i];
        }
        positive = !positive;
      }
    }

    // ---- solve D Y = E: block Gauss-Jordan, 16 x 16 blocks, on the matrix cores ------------------------------------------
    // No row exchanges between blocks: D = I -+ A/2 + ... of a rate matrix times a time has its eigenvalues right of 1 and a
    // dominant diagonal; a pivot below piv_min (1e-3) flags the matrix for the pivoted kernel instead.
    // Step k: wave k publishes its column block of D and inverts the diagonal block (Gauss-Jordan in LDS, one wave) -> W; every
    // wave multiplies row block k of ITS column blocks of D and E by W (the result tile is a B-operand slice as it stands) and
    // subtracts D_ik times that from every other row block i -- all in registers, two barriers per step.  After the last step
    // E holds Y.  (Newton-Schulz iterations W <- W (2 I - D_kk W) on the matrix cores were tried for the inversion: with
    // arma's scaling, s = exponent(log2 ||A||) + 1, ||A / 2^s|| reaches 1 - 4, D_kk is far from the identity and the
    // iteration needs 6 - 8 steps of three dependent products from W = diag(D_kk)^-1 -- no faster than the elimination.)
    if (tid == 0) s_bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // unrolled: the register tiles are indexed by k
      if (k >= rb) continue;                           // uniform over the workgroup (the barriers below are too)
      __syncthreads();                                 // the previous step's (or matrix's) readers of sCol / sW are done
      if (w == k) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) sCol[(16 * i + lk + 4 * q) * 17 + lr] = Dm[i][q];
        // invert D_kk by Gauss-Jordan elimination in LDS: lane = (row r, four columns 4 cg .. 4 cg + 3) of [D_kk | I]
        const int r = lane & 15, cg = lane >> 4;
        double a[4], wv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a[c] = sCol[(16 * k + r) * 17 + 4 * cg + c];             // this wave's own writes, in order
          wv[c] = (r == 4 * cg + c) ? 1.0 : 0.0;
          sA[r * 17 + 4 * cg + c] = a[c];
          sW[r * 17 + 4 * cg + c] = wv[c];
        }
        bool bad_piv = false;
        for (int pv = 0; pv < 16; ++pv) {
          const double piv = sA[pv * 17 + pv];
          if (!(fabs(piv) >= piv_min)) bad_piv = true;
          const double inv = 1.0 / piv;
          const double f = sA[r * 17 + pv];
          double pa[4], pw[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { pa[c] = sA[pv * 17 + 4 * cg + c] * inv; pw[c] = sW[pv * 17 + 4 * cg + c] * inv; }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            a[c] = (r == pv) ? pa[c] : a[c] - f * pa[c];
            wv[c] = (r == pv) ? pw[c] : wv[c] - f * pw[c];
            sA[r * 17 + 4 * cg + c] = a[c];
            sW[r * 17 + 4 * cg + c] = wv[c];
          }
        }
        if (bad_piv) s_bad = 1;
      }
      __syncthreads();
      if (w < rb) {
        double aW[4];
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) aW[sI] = sW[lr * 17 + 4 * sI + lk];
        if (w > k) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[sI], Dm[k][sI], acc, 0, 0, 0);
          Dm[k] = acc;
        }
        {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[sI], Em[k][sI], acc, 0, 0, 0);
          Em[k] = acc;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i != k && i < rb) {
            double aD[4];
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) aD[sI] = -sCol[(16 * i + lr) * 17 + 4 * sI + lk];
            if (w > k) {
              d4_t acc = Dm[i];
#pragma unroll
              for (int sI = 0; sI < 4; ++sI) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aD[sI], Dm[k][sI], acc, 0, 0, 0);
              Dm[i] = acc;
            }
            d4_t acc = Em[i];
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aD[sI], Em[k][sI], acc, 0, 0, 0);
            Em[i] = acc;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sE[(16 * i + lk + 4 * q) * PADE_LDP + mycol] = Em[i][q];
    __syncthreads();
    if (s_bad) {                                       // uniform: left to the exact kernel
      if (tid == 0) bad[b] = 1;
      __syncthreads();
      continue;
    }

    // ---- s squarings: Y <- Y Y, operands re-read from LDS in A- and B-slice layouts ----
    for (int it = 0; it < s; ++it) {
      {
        d4_t Yb[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int q = 0; q < 4; ++q) Yb[kb][q] = sE[(16 * kb + lk + 4 * q) * PADE_LDP + mycol];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
          if (i < rb && w < rb) {
#pragma unroll
            for (int sI = 0; sI < 16; ++sI)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sE[(16 * i + lr) * PADE_LDP + 4 * sI + lk], Yb[sI >> 2][sI & 3], acc, 0, 0, 0);
          }
          T[i] = acc;
        }
      }
      __syncthreads();
      {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) sE[(16 * i + lk + 4 * q) * PADE_LDP + mycol] = T[i][q];
      }
      __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) {
      const int r = e / n, k = e - r * n;
      out[(size_t)b * n * n + e] = sE[r * PADE_LDP + k];
    }
    __syncthreads();
  }
}

hipError_t launch_expm_pade_mfma(int n, const double* Q, const double* t, const int32_t* s, int n_t, double* out,
                                 int32_t* bad, double piv_min, hipStream_t stream) {
  size_t lds = sizeof(double) * 2 * 64 * PADE_LDP;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(expm_pade_mfma_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int grid = n_t < 1024 ? n_t : 1024;
  hipLaunchKernelGGL(expm_pade_mfma_kernel, dim3(grid), dim3(256), lds, stream, n, Q, t, s, n_t, out, bad, piv_min);
  return hipGetLastError();
}

hipError_t launch_expm_eigen_mfma(int n, const double* L, const double* R, const double* dvals, const double* t, int n_t,
                                  double* out, hipStream_t stream) {
  int grid = n_t < 2048 ? n_t : 2048;
  hipLaunchKernelGGL(expm_eigen_mfma_kernel, dim3(grid), dim3(256), 0, stream, n, L, R, dvals, t, n_t, out);
  return hipGetLastError();
}

}  // namespace phm
