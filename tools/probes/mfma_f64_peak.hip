// Measured rate of v_mfma_f64_16x16x4_f64 on this device: what the matrix-core pruning kernel (phm_wtiles.hip) can be priced against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_peak tools/probes/mfma_f64_peak.hip && /tmp/mfma_f64_peak
// CHAINS independent accumulators per wave, each a dependent chain of MFMAs (the pruning kernel: 4 accumulators, 16 dependent each);
// WAVES waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4_t = __attribute__((ext_vector_type(4))) double;

template <int CHAINS>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  d4_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4_t{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd, double* out) {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const int iters = 4000;
  const int blocks = cus * waves_per_simd;            // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 1e-3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * 4 * iters * 8 * CHAINS;
  const double tf = n_mfma * 2048 / (ms * 1e-3) / 1e12;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * CHAINS * waves_per_simd);
  std::printf("chains %d, waves/SIMD %d: %.1f TFLOP/s, %.1f cycles of a 2.4 GHz clock per MFMA and SIMD (%d CUs, %.2f ms)\n", CHAINS, waves_per_simd, tf, cyc, cus, ms);
}

int main() {
  double* out; (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
  for (int w = 1; w <= 2; ++w) { run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out); }
  return 0;
}
