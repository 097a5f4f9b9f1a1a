// Issue cost of single gfx950 VALU instructions (asm volatile, four independent destinations, 8 waves per SIMD), and of a
// Philox4x32-7 block built from v_mad_u64_u32 (what the compiler picks for a 32x32->64 product) or from v_mul_hi_u32 + v_mul_lo_u32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048
#define A4(S, C) asm volatile(S "\n" S "\n" S "\n" S :: C);

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, double dseed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x9e3779b9u, a3 = a1 + 77u;
  double d0 = dseed + threadIdx.x, d1 = d0 * 1.5;
  double r0 = 0, r1 = 0, r2 = 0, r3 = 0; uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  for (int i = 0; i < REP; ++i) {
    if (OP == 0) asm volatile("v_cmp_lt_f64 vcc, %0, %1\nv_cmp_lt_f64 vcc, %1, %0\nv_cmp_lt_f64 vcc, %0, %1\nv_cmp_lt_f64 vcc, %1, %0" :: "v"(d0), "v"(d1) : "vcc");
    if (OP == 1) asm volatile("v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0), "v"(a1) : "vcc");
    if (OP == 2) asm volatile("v_frexp_mant_f64 %0, %4\nv_frexp_mant_f64 %1, %4\nv_frexp_mant_f64 %2, %4\nv_frexp_mant_f64 %3, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0));
    if (OP == 3) asm volatile("v_frexp_exp_i32_f64 %0, %4\nv_frexp_exp_i32_f64 %1, %4\nv_frexp_exp_i32_f64 %2, %4\nv_frexp_exp_i32_f64 %3, %4" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(d0));
    if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %4\nv_cvt_i32_f64 %2, %4\nv_cvt_i32_f64 %3, %4" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(d0));
    if (OP == 5) asm volatile("v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %4\nv_cvt_f64_i32 %2, %4\nv_cvt_f64_i32 %3, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0));
    if (OP == 6) asm volatile("v_cvt_f64_u32 %0, %4\nv_cvt_f64_u32 %1, %4\nv_cvt_f64_u32 %2, %4\nv_cvt_f64_u32 %3, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0));
    if (OP == 7) asm volatile("v_mul_hi_u32 %0, %4, %5\nv_mul_hi_u32 %1, %4, %5\nv_mul_hi_u32 %2, %4, %5\nv_mul_hi_u32 %3, %4, %5" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0), "v"(a1));
    if (OP == 8) asm volatile("v_mul_lo_u32 %0, %4, %5\nv_mul_lo_u32 %1, %4, %5\nv_mul_lo_u32 %2, %4, %5\nv_mul_lo_u32 %3, %4, %5" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0), "v"(a1));
    if (OP == 9) asm volatile("v_fma_f64 %0, %4, %5, %5\nv_fma_f64 %1, %4, %5, %5\nv_fma_f64 %2, %4, %5, %5\nv_fma_f64 %3, %4, %5, %5" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0), "v"(d1));
    if (OP == 10) asm volatile("v_xor_b32 %0, %4, %5\nv_xor_b32 %1, %4, %5\nv_xor_b32 %2, %4, %5\nv_xor_b32 %3, %4, %5" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0), "v"(a1));
    if (OP == 11) asm volatile("v_ldexp_f64 %0, %4, %5\nv_ldexp_f64 %1, %4, %5\nv_ldexp_f64 %2, %4, %5\nv_ldexp_f64 %3, %4, %5" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0), "v"(a1));
    if (OP == 12) asm volatile("v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %0\nv_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %0" :: "v"(a0), "v"(a1) : "vcc");
    if (OP == 13) asm volatile("v_lshlrev_b64 %0, 3, %4\nv_lshlrev_b64 %1, 3, %4\nv_lshlrev_b64 %2, 3, %4\nv_lshlrev_b64 %3, 3, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0));
    if (OP == 14) asm volatile("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %0, 5\nv_readlane_b32 s22, %0, 7\nv_readlane_b32 s23, %0, 9" :: "v"(a0) : "s20", "s21", "s22", "s23");
    if (OP == 15) asm volatile("v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0));
    if (OP == 16) asm volatile("v_add_f64 %0, %4, %5\nv_add_f64 %1, %4, %5\nv_add_f64 %2, %4, %5\nv_add_f64 %3, %4, %5" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(d0), "v"(d1));
    if (OP == 17) asm volatile("v_lshl_add_u32 %0, %4, 3, %5\nv_lshl_add_u32 %1, %4, 3, %5\nv_lshl_add_u32 %2, %4, 3, %5\nv_lshl_add_u32 %3, %4, 3, %5" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a0), "v"(a1));
    if (OP == 20 || OP == 21) {      // one Philox4x32-7 block per iteration
      uint32_t c0 = a0, c1 = a1, c2 = a2, c3 = a3 + i, k0 = seed, k1 = 77u;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        uint32_t h0, l0, h1, l1;
        if (OP == 20) { uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2; h0 = p0 >> 32; l0 = (uint32_t)p0; h1 = p1 >> 32; l1 = (uint32_t)p1; }
        else {
          asm("v_mul_hi_u32 %0, %1, %2" : "=v"(h0) : "v"(c0), "s"(0xD2511F53u));
          asm("v_mul_lo_u32 %0, %1, %2" : "=v"(l0) : "v"(c0), "s"(0xD2511F53u));
          asm("v_mul_hi_u32 %0, %1, %2" : "=v"(h1) : "v"(c2), "s"(0xCD9E8D57u));
          asm("v_mul_lo_u32 %0, %1, %2" : "=v"(l1) : "v"(c2), "s"(0xCD9E8D57u));
        }
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
      }
      a0 ^= c0; a1 ^= c1; a2 ^= c2; a3 ^= c3;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ q0 ^ q1 ^ q2 ^ q3 ^ (uint32_t)(int64_t)(r0 + r1 + r2 + r3);
}

static uint32_t* d;
template <int OP>
void run(const char* name, int per_iter) {
  const int blocks = 256 * 8;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 1.0);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u, 1.0);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)REP * per_iter * 8;
  printf("%-28s %8.3f ms  -> %7.2f clk (2.4 GHz) per wave-%s\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd, per_iter == 1 ? "block" : "instruction");
}

int main() {
  (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  run<15>("v_mov_b32", 4); run<10>("v_xor_b32", 4); run<17>("v_lshl_add_u32", 4); run<7>("v_mul_hi_u32", 4); run<8>("v_mul_lo_u32", 4);
  run<12>("v_cmp_lt_u32", 4); run<1>("v_cndmask_b32", 4); run<14>("v_readlane_b32", 4);
  run<9>("v_fma_f64", 4); run<16>("v_add_f64", 4); run<0>("v_cmp_lt_f64", 4); run<2>("v_frexp_mant_f64", 4); run<3>("v_frexp_exp_i32_f64", 4);
  run<4>("v_cvt_i32_f64", 4); run<5>("v_cvt_f64_i32", 4); run<6>("v_cvt_f64_u32", 4); run<11>("v_ldexp_f64", 4); run<13>("v_lshlrev_b64", 4);
  run<20>("philox4x32-7 (v_mad_u64_u32)", 1); run<21>("philox4x32-7 (mul_hi + mul_lo)", 1);
  return 0;
}
