// Measured device peaks to price the kernels against: HBM bandwidth from stream kernels (copy, triad,
// read-only, write-only) and FP64 throughput from a vector-FMA loop and a v_mfma_f64_16x16x4 loop.
//   hipcc --offload-arch=gfx950 -O3 -o device_peaks device_peaks.hip && ./device_peaks
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s, size_t n2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    double2 x = b[i], y = c[i];
    a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
  }
}
__global__ __launch_bounds__(256) void k_copy(double2* __restrict__ a, const double2* __restrict__ b, size_t n2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i];
}
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ b, double* out, size_t n2) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { double2 x = b[i]; acc += x.x + x.y; }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ a, double v, size_t n2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2(v, v);
}

// one pass, four independent 16-byte accesses per lane in flight
__global__ __launch_bounds__(256) void k_triad4(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s, size_t n2) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (base + 768 >= n2) return;
  double2 x[4], y[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { x[u] = b[base + u * 256]; y[u] = c[base + u * 256]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) a[base + u * 256] = make_double2(x[u].x + s * y[u].x, x[u].y + s * y[u].y);
}
__global__ __launch_bounds__(256) void k_read4(const double2* __restrict__ b, double* out, size_t n2) {
  const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
  if (base + 7 * 256 >= n2) return;
  double acc = 0.0;
  double2 x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = b[base + u * 256];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc += x[u].x + x[u].y;
  if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double m = 1.0000000001, a = 1e-12;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, a);
  }
  double acc = 0.0;
  for (int i = 0; i < 8; ++i) acc += x[i];
  if (acc == 1.2345e300) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  double acc = c0[0] + c1[1] + c2[2] + c3[3];
  if (acc == 1.2345e300) out[0] = acc;
}

template <class F>
float timed(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const size_t n = (size_t)1 << 29;            // 4 GiB per array: far beyond the 256 MB Infinity Cache
  const size_t n2 = n / 2;
  double *a, *b, *c, *out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8));
  const int grid = 256 * 32;
  float ms;
  ms = timed([&] { k_copy<<<grid, 256>>>((double2*)a, (const double2*)b, n2); }, 5);
  printf("copy   (read + write)      %8.1f GB/s\n", 2.0 * n * 8 / ms / 1e6);
  ms = timed([&] { k_triad<<<grid, 256>>>((double2*)a, (const double2*)b, (const double2*)c, 3.0, n2); }, 5);
  printf("triad  (2 reads + write)   %8.1f GB/s\n", 3.0 * n * 8 / ms / 1e6);
  ms = timed([&] { k_triad4<<<(unsigned)(n2 / 1024), 256>>>((double2*)a, (const double2*)b, (const double2*)c, 3.0, n2); }, 5);
  printf("triad, one pass x4         %8.1f GB/s\n", 3.0 * n * 8 / ms / 1e6);
  ms = timed([&] { k_read4<<<(unsigned)(n2 / 2048), 256>>>((const double2*)b, out, n2); }, 5);
  printf("read only, one pass x8     %8.1f GB/s\n", 1.0 * n * 8 / ms / 1e6);
  ms = timed([&] { k_read<<<grid, 256>>>((const double2*)b, out, n2); }, 5);
  printf("read only                  %8.1f GB/s\n", 1.0 * n * 8 / ms / 1e6);
  ms = timed([&] { k_write<<<grid, 256>>>((double2*)a, 1.0, n2); }, 5);
  printf("write only                 %8.1f GB/s\n", 1.0 * n * 8 / ms / 1e6);
  const int iters = 4000, blocks = 1024 * 2;     // 8 waves per SIMD
  ms = timed([&] { k_fma<<<blocks, 256>>>(out, iters); }, 3);
  printf("FP64 vector fma            %8.2f TFLOP/s\n", (double)blocks * 256 * iters * 32 * 2 / ms / 1e9);
  ms = timed([&] { k_mfma<<<blocks, 256>>>(out, iters); }, 3);
  printf("FP64 v_mfma_f64_16x16x4    %8.2f TFLOP/s\n", (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9);
  return 0;
}
