// Clocks per wave of the building blocks of the n <= 4 branch kernel (phm_tiles.hip), as the compiler emits them from
// phm_device.h: 8 waves per SIMD, every block fed by the previous one's result so nothing is hoisted.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "phm_device.h"
using namespace phm;

#define REP 1024
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, double dseed) {
  __shared__ __align__(16) double s_ltab[2 * PHM_LOGTAB_N];
  __shared__ double s_col[16 * 16], s_B2[16], s_scale[4];
  for (int i = threadIdx.x; i < 2 * PHM_LOGTAB_N; i += 256) s_ltab[i] = logtab_entry(i);
  s_col[threadIdx.x] = 0.1 + 0.001 * threadIdx.x; if (threadIdx.x < 16) s_B2[threadIdx.x] = 0.2 + 0.01 * threadIdx.x; if (threadIdx.x < 4) s_scale[threadIdx.x] = 0.5;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + seed, err = 0;
  double acc = dseed;
  int s = threadIdx.x & 3;
  for (int i = 0; i < REP; ++i) {
    if (OP == 0) { uint32_t o[4]; philox4x32(i, 5u, x, threadIdx.x, seed, 7u, o); x ^= o[0] ^ o[1] ^ o[2] ^ o[3]; }
    if (OP == 1) { acc += neglog_u32(x, s_ltab); x = x * 1664525u + 1013904223u; }
    if (OP == 2) { double p[4] = {acc, 0.3, 0.2 + dseed, 0.4}; s = sample_cat<4>(p, u01(x), err); acc += 0.125 * s; x = x * 1664525u + 1013904223u; }
    if (OP == 3) { acc += u01(x); x = x * 1664525u + 1013904223u; }
    if (OP == 4) {      // draw_state: LDS rows, 4 products, categorical draw
      const double* beta = s_col + (((x >> 8) & 15) * 4 + (s ^ 1)) * 4 % 240; double pr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) pr[c] = s_B2[s * 4 + c] * beta[c];
      s = sample_cat<4>(pr, u01(x), err); x = x * 1664525u + 1013904223u;
    }
    if (OP == 5) {      // pass-B step without the store: variate, compare, select
      const double len = 0.7 + dseed;
      double rl = s_scale[s] * neglog_u32(x, s_ltab);
      double piece; if ((acc + rl) < len) { piece = rl; } else { piece = len - acc; s = (s + 1) & 3; }
      acc += piece * 1e-3; x = x * 1664525u + 1013904223u;
    }
    if (OP == 6) { x = x * 1664525u + 1013904223u; }      // the feeder alone
  }
  out[blockIdx.x * 256 + threadIdx.x] = x ^ err ^ (uint32_t)s ^ (uint32_t)(int64_t)(acc * 1e6);
}

static uint32_t* d;
template <int OP>
double run(const char* name, double feeder) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 1u, 1.0);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 2u, 1.0);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double clk = ms * 1e-3 * 2.4e9 / ((double)REP * 8);
  printf("%-44s %8.3f ms  -> %7.1f clk (2.4 GHz) per wave-call (less feeder: %7.1f)\n", name, ms, clk, clk - feeder);
  return clk;
}

int main() {
  (void)hipMalloc(&d, 2048 * 256 * 4);
  const double f = run<6>("feeder (x = a x + c)", 0);
  run<0>("philox4x32-7 block", 0); run<3>("u01", f); run<1>("neglog_u32 (LDS table)", f); run<2>("sample_cat<4>", f);
  run<4>("draw_state: LDS rows + products + draw", f); run<5>("virtual-jump step without the store", f);
  return 0;
}
