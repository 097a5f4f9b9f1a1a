// Issue rates of the VALU instructions the branch kernels are made of (gfx950): clocks per wave-instruction on one SIMD, measured
// with 8 waves per SIMD and four independent dependency chains per wave.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, double dseed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x9e3779b9u, a3 = a1 + 77u;
  uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
  double d0 = dseed + threadIdx.x, d1 = d0 * 1.5, d2 = d0 + 3.0, d3 = d1 - 2.0;
  const uint32_t M = 0xD2511F53u;
  for (int i = 0; i < REP; ++i) {
    if (OP == 0) { a0 ^= a1 + i; a1 ^= a2 + i; a2 ^= a3 + i; a3 ^= a0 + i; }                    // 8 x (v_add / v_xor) b32
    if (OP == 1) { w0 = (uint64_t)(uint32_t)w0 * M + w1; w1 = (uint64_t)(uint32_t)w1 * M + w2; w2 = (uint64_t)(uint32_t)w2 * M + w3; w3 = (uint64_t)(uint32_t)w3 * M + w0; }   // 4 x v_mad_u64_u32
    if (OP == 2) { d0 = __builtin_fma(d0, 1.0000001, d1); d1 = __builtin_fma(d1, 0.9999999, d2); d2 = __builtin_fma(d2, 1.0000002, d3); d3 = __builtin_fma(d3, 0.9999998, d0); }   // 4 x v_fma_f64
    if (OP == 3) { d0 = d0 + d1; d1 = d1 + d2; d2 = d2 + d3; d3 = d3 + d0; }                  // 4 x v_add_f64
    if (OP == 4) { d0 = d0 * d1; d1 = d1 * d2; d2 = d2 * d3; d3 = d3 * d0; }                  // 4 x v_mul_f64
    if (OP == 5) { a0 = __umulhi(a0, M) ^ a1; a1 = __umulhi(a1, M) ^ a2; a2 = __umulhi(a2, M) ^ a3; a3 = __umulhi(a3, M) ^ a0; }   // 4 x (v_mul_hi_u32 + v_xor)
    if (OP == 6) { a0 = a0 * M ^ a1; a1 = a1 * M ^ a2; a2 = a2 * M ^ a3; a3 = a3 * M ^ a0; }   // 4 x (v_mul_lo_u32 + v_xor)
    if (OP == 7) { d0 = (d0 < d1) ? d2 : d0; d1 = (d1 < d2) ? d3 : d1; d2 = (d2 < d3) ? d0 : d2; d3 = (d3 < d0) ? d1 : d3; }   // 4 x (v_cmp_f64 + 2 v_cndmask)
    if (OP == 8) { d0 = __builtin_ldexp(d0, (int)(a0 & 1)); d1 = __builtin_ldexp(d1, (int)(a0 & 1)); d2 = __builtin_ldexp(d2, (int)(a0 & 1)); d3 = __builtin_ldexp(d3, (int)(a0 & 1)); }
    if (OP == 9) { d0 = (double)(uint32_t)(a0 + i) + d0; d1 = (double)(uint32_t)(a1 + i) + d1; d2 = (double)(uint32_t)(a2 + i) + d2; d3 = (double)(uint32_t)(a3 + i) + d3; }   // 4 x (v_add + v_cvt_f64_u32 + v_add_f64)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3) ^ (uint32_t)(int64_t)(d0 + d1 + d2 + d3);
}

template <int OP>
void run(const char* name, int per_iter, uint32_t* d) {
  const int blocks = 256 * 8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 1.0);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u, 1.0);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double wave_instr_per_simd = (double)REP * per_iter * 8;      // 8 waves per SIMD
  printf("%-34s %8.3f ms  -> %6.2f clk per wave-instruction (at %d MHz)\n", name, ms, ms * 1e-3 * clk_khz * 1e3 / wave_instr_per_simd, clk_khz / 1000);
}

int main() {
  uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32 / v_xor_b32 (8)", 8, d);
  run<1>("v_mad_u64_u32 (4)", 4, d);
  run<2>("v_fma_f64 (4)", 4, d);
  run<3>("v_add_f64 (4)", 4, d);
  run<4>("v_mul_f64 (4)", 4, d);
  run<5>("v_mul_hi_u32 + v_xor (4+4)", 8, d);
  run<6>("v_mul_lo_u32 + v_xor (4+4)", 8, d);
  run<7>("v_cmp_f64 + 2 v_cndmask (4+8)", 12, d);
  run<8>("v_ldexp_f64 (4)", 4, d);
  run<9>("v_add + v_cvt_f64_u32 + v_add_f64", 12, d);
  return 0;
}
