// Microbenchmark: does a chain of dependent kernels pay for the L2 write-back of what the previous kernel stored?
// y[i] = a * x[i] over n doubles, 200 launches back to back on one stream; plain stores against nontemporal stores.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/wbflush.hip -o scripts/micro/wbflush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool NTS>
__global__ __launch_bounds__(256) void scale_kernel(const double* __restrict__ x, double* __restrict__ y, long n, double a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = a * __builtin_nontemporal_load(x + i);
    if (NTS) __builtin_nontemporal_store(v, y + i);
    else y[i] = v;
  }
}

int main() {
  const long sizes[] = {1L << 14, 1L << 17, 1L << 20, 1L << 22, 10707315L, 1L << 25};
  double *x, *y;
  const long nmax = 1L << 25;
  (void)hipMalloc(&x, nmax * 8);
  (void)hipMalloc(&y, nmax * 8);
  (void)hipMemset(x, 0, nmax * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (long n : sizes) {
    for (int nts = 0; nts < 2; ++nts) {
      const int reps = 200;
      const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
      for (int w = 0; w < 2; ++w) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) {
          // ping-pong so that every launch depends on the previous one's output
          const double* in = r & 1 ? y : x;
          double* out = r & 1 ? x : y;
          if (nts) hipLaunchKernelGGL(scale_kernel<true>, dim3(grid), dim3(256), 0, 0, in, out, n, 1.0);
          else hipLaunchKernelGGL(scale_kernel<false>, dim3(grid), dim3(256), 0, 0, in, out, n, 1.0);
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = 1e3 * ms / reps;
      printf("n = %9ld doubles (%7.2f MB written per launch), %s stores: %8.2f us per launch, %6.0f GB/s\n", n, n * 8e-6,
             nts ? "nontemporal" : "plain      ", us, 16.0 * n / us * 1e-3);
    }
  }
  return 0;
}
