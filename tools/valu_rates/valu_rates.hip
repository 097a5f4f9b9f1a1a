// Issue cost of the VALU instructions the sweep kernels lean on, measured on the device: cycles per wave64 instruction on one
// SIMD with 4 resident waves and 8 independent chains per wave (throughput, not latency).  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, int iters, uint32_t seed) {
  uint64_t a[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 977u + i; d[i] = 1.0 + 1e-9 * (double)(a[i] & 1023); }
  const uint32_t m = 0xD2511F53u;
  const double c = 1.0000001;
  for (int it = 0; it < iters; ++it) {
#define MAD(i) if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"((uint32_t)a[i]), "s"(m) : "vcc");
#define XOR(i) if (OP == 1) { uint32_t t; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"((uint32_t)a[i]), "s"(m)); a[i] = t; }
#define MULF(i) if (OP == 2) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(c));
#define ADDF(i) if (OP == 3) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(c));
#define FMAF(i) if (OP == 4) asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(d[i]) : "v"(d[i]), "v"(c));
#define RCP(i) if (OP == 5) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define CVT(i) if (OP == 6) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"((uint32_t)a[i]));
#define LDEXP(i) if (OP == 7) asm volatile("v_ldexp_f64 %0, %1, 1" : "=v"(d[i]) : "v"(d[i]));
#define FREXP(i) if (OP == 8) asm volatile("v_frexp_mant_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define MULLO(i) if (OP == 9) { uint32_t t; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"((uint32_t)a[i]), "s"(m)); a[i] = t; }
#define MULHI(i) if (OP == 10) { uint32_t t; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(t) : "v"((uint32_t)a[i]), "s"(m)); a[i] = t; }
#define CMP(i) if (OP == 11) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(c) : "vcc");
#define CNDM(i) if (OP == 12) { uint32_t t; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"((uint32_t)a[i]), "v"(m) : "vcc"); a[i] = t; }
#define MUL24(i) if (OP == 13) { uint32_t t; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(t) : "v"((uint32_t)a[i]), "v"(m)); a[i] = t; }
#define DIVF(i) if (OP == 14) d[i] = c / d[i];
#define CVTI(i) if (OP == 15) { uint32_t t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(d[i])); a[i] = t; }
#pragma unroll 4
    for (int u = 0; u < 4; ++u) {
      REP8(MAD) REP8(XOR) REP8(MULF) REP8(ADDF) REP8(FMAF) REP8(RCP) REP8(CVT) REP8(LDEXP) REP8(FREXP) REP8(MULLO) REP8(MULHI)
      REP8(CMP) REP8(CNDM) REP8(MUL24) REP8(DIVF) REP8(CVTI)
    }
  }
  uint64_t acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + (uint64_t)__double_as_longlong(d[i]);
  if (acc == 0x1234567u) out[0] = acc;
}

template <int OP>
double run(const char* name, uint64_t* out, int instr_per_rep) {
  const int iters = 2000, blocks = 256 * 4;       // one 256-thread block per SIMD-quad x 4: four waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<OP><<<blocks, 256>>>(out, 10, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<OP><<<blocks, 256>>>(out, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double instr_per_wave = (double)iters * 4 * 8 * instr_per_rep;
  const double waves_per_simd = 4.0;             // 1024 blocks x 4 waves / 1024 SIMDs
  const double cycles = ms * 1e-3 * clk_khz * 1e3;
  const double cyc_per_instr = cycles / (instr_per_wave * waves_per_simd);
  printf("%-22s %8.3f ms  %6.2f cycles per wave64 instruction (at %d MHz)\n", name, ms, cyc_per_instr, clk_khz / 1000);
  return cyc_per_instr;
}

int main() {
  uint64_t* out;
  hipMalloc(&out, 64);
  run<1>("v_xor_b32", out, 1);
  run<0>("v_mad_u64_u32", out, 1);
  run<9>("v_mul_lo_u32", out, 1);
  run<10>("v_mul_hi_u32", out, 1);
  run<13>("v_mul_u32_u24", out, 1);
  run<2>("v_mul_f64", out, 1);
  run<3>("v_add_f64", out, 1);
  run<4>("v_fma_f64", out, 1);
  run<5>("v_rcp_f64", out, 1);
  run<6>("v_cvt_f64_u32", out, 1);
  run<15>("v_cvt_i32_f64", out, 1);
  run<7>("v_ldexp_f64", out, 1);
  run<8>("v_frexp_mant_f64", out, 1);
  run<11>("v_cmp_lt_f64", out, 1);
  run<12>("v_cndmask_b32", out, 1);
  run<14>("f64 division (sequence)", out, 1);
  return 0;
}
