// phm_rtc.cpp -- generates, compiles (hipRTC) and caches the pattern-specialised pruning kernel of phm_rtc.h.
#include "phm_rtc.h"

#include <hip/hiprtc.h>

#include <memory>
#include <mutex>
#include <sstream>

namespace phm {

namespace {

#define PHM_RTC_STR(...) #__VA_ARGS__
const char* const kAbiText = PHM_RTC_ABI(PHM_RTC_STR);
#undef PHM_RTC_STR

}  // namespace

// The kernel is wt_up_band_kernel (phm_wtiles.hip) with the band loop replaced by one generated line per row of the matrix:
// same loads, same order of operations around the chain step, same normalisation sum (four interleaved partial sums).
std::string rtc_sparse_up_source(int n, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col) {
  const int np = (n + 3) / 4 * 4;
  std::ostringstream s;
  s << "#if !defined(__HIPCC_RTC__)\n#include <hip/hip_runtime.h>\n#endif\n";      // hipRTC brings the HIP device declarations itself; hipcc (tests) needs the header
  s << kAbiText << "\n";
  s << "#define N " << n << "\n#define NP " << np << "\n";
  s << R"(
extern "C" __global__ __launch_bounds__(256) void phm_sparse_up(RtcUpParams p, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int n_lvl = end - begin;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)n_lvl * p.n_tiles) return;
  const int ldt = p.ldt;
  const int tile = (int)(item % p.n_tiles), li = (int)(item / p.n_tiles);
  const RtcUpStep st = p.up[p.up_order[begin + li]];
  double* __restrict__ PLt = p.PL + (size_t)tile * p.n_node * N * 64;
  const unsigned short* __restrict__ mct = p.mcount + (size_t)tile * p.n_edge * 64;
  const unsigned char* __restrict__ tips_t = p.tips_per_replica ? p.tips + (size_t)tile * p.n_tips * 64 : p.tips;
  const double* __restrict__ C = p.coef;
  unsigned int err = 0;
  double R[2][NP];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {                     // ch 0: "first" = child[1] (:508); ch 1: "second" = child[0] (:509)
    const int child = st.child[1 - ch], edge = st.edge[1 - ch];
    int k = (int)mct[edge * 64 + lane] - 1;
    if (child < 0) {                                   // tip: a row of the chain table
      const int tip = ~child;
      const int ts = p.tips_per_replica ? tips_t[tip * 64 + lane] : p.tips[tip];
      if (k >= p.klong) { err |= 2u; k = p.klong - 1; }
      const double2* __restrict__ src = reinterpret_cast<const double2*>(p.tip_masks ? p.maskL + ((size_t)k * 2 + (ts & 1)) * ldt : p.colL + ((size_t)k * N + ts) * ldt);
#pragma unroll
      for (int i = 0; i < NP; i += 2) {                // rows are 16-byte aligned and padded to an even length with zeros
        double2 v = {0.0, 0.0};
        if (i < N) v = src[i >> 1];
        R[ch][i] = v.x; R[ch][i + 1] = v.y;
      }
    } else {
      double (&x)[NP] = R[ch];
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = (i < N) ? PLt[((size_t)child * N + i) * 64 + lane] : 0.0;
      int kmax = k;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
      for (int step = 1; step <= kmax; ++step) {
        if (step <= k) {                               // x <- Bc x for the lanes still inside their chain
          double y[NP];
#pragma unroll
          for (int i = N; i < NP; ++i) y[i] = 0.0;
)";
  for (int i = 0; i < n; ++i) {
    s << "          { double a = 0.0;";
    for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) s << " a = __builtin_fma(C[" << k << "], x[" << col[k] << "], a);";
    s << " y[" << i << "] = a; }\n";
  }
  s << R"(
#pragma unroll
          for (int i = 0; i < NP; ++i) x[i] = y[i];
        }
      }
    }
  }
  double P[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) P[i] = R[0][i] * R[1][i];                      // :510
  if (p.normalise) {                                                          // :525; four interleaved partial sums (DESIGN.md section 2)
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i & 3] += P[i];
    const double tt = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
    for (int i = 0; i < NP; ++i) P[i] = P[i] / tt;
  }
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (i < N) PLt[((size_t)st.parent * N + i) * 64 + lane] = P[i];
  if (err) atomicOr(p.err, err);
}
)";
  return s.str();
}

const SparseUpKernel* rtc_sparse_up_kernel(int n, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col, std::string& err) {
  static std::mutex mu;
  static std::vector<std::unique_ptr<SparseUpKernel>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { err = "hipGetDevice failed"; return nullptr; }
  std::lock_guard<std::mutex> guard(mu);
  for (const auto& k : cache)
    if (k->device == dev && k->n == n && k->row_ptr == row_ptr && k->col == col) return k.get();
  const std::string src = rtc_sparse_up_source(n, row_ptr, col);
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "phm_sparse_up.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return nullptr; }
  // -ffp-contract=off: no contraction beyond the fused multiply-adds that are written (the arithmetic spec of the whole library)
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  const hiprtcResult cr = hiprtcCompileProgram(prog, 4, opts);
  if (cr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    (void)hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) (void)hiprtcGetProgramLog(prog, &log[0]);
    (void)hiprtcDestroyProgram(&prog);
    err = std::string("hipRTC could not compile the pattern-specialised pruning kernel: ") + hiprtcGetErrorString(cr) + ": " + log.substr(0, 400);
    return nullptr;
  }
  size_t cs = 0;
  (void)hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  (void)hiprtcGetCode(prog, code.data());
  (void)hiprtcDestroyProgram(&prog);
  auto k = std::make_unique<SparseUpKernel>();
  if (hipModuleLoadData(&k->module, code.data()) != hipSuccess) { err = "hipModuleLoadData failed for the pattern-specialised pruning kernel"; return nullptr; }
  if (hipModuleGetFunction(&k->fn, k->module, "phm_sparse_up") != hipSuccess) { (void)hipModuleUnload(k->module); err = "hipModuleGetFunction failed"; return nullptr; }
  k->device = dev; k->n = n; k->nnz = (int)col.size(); k->row_ptr = row_ptr; k->col = col;
  cache.push_back(std::move(k));
  return cache.back().get();
}

hipError_t launch_sparse_up(const SparseUpKernel& k, const RtcUpParams& p, const std::vector<int32_t>& up_off, hipStream_t stream) {
  for (size_t l = 0; l + 1 < up_off.size(); ++l) {
    const int cnt = up_off[l + 1] - up_off[l];
    if (cnt <= 0) continue;
    const long long items = (long long)cnt * p.n_tiles;        // a wave per (node, tile)
    RtcUpParams pp = p;
    int begin = up_off[l], end = up_off[l + 1];
    void* args[] = {&pp, &begin, &end};
    const hipError_t e = hipModuleLaunchKernel(k.fn, (unsigned)((items + 3) / 4), 1, 1, 256, 1, 1, 0, stream, args, nullptr);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace phm
