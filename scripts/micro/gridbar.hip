// micro-benchmark: cost of a grid-wide barrier inside one cooperative kernel vs a chain of dependent tiny kernels (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

// every phase: x[i] = x[perm-neighbour] + 1 (forces real cross-workgroup data flow), then barrier
__global__ void persistent(double* a, double* b, int n, int phases, unsigned* ctr) {
  const int G = gridDim.x;
  for (int p = 0; p < phases; ++p) {
    double* src = (p & 1) ? b : a;
    double* dst = (p & 1) ? a : b;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G * blockDim.x) dst[i] = src[(i + 4099) % n] + 1.0;
    grid_barrier(ctr, (unsigned)G * (p + 1));
  }
}
__global__ void step(double* dst, const double* src, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[(i + 4099) % n] + 1.0;
}

int main() {
  const int phases = 2000;
  for (int n : {4096, 65536, 1048576}) {
    double *a, *b;
    unsigned* ctr;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&ctr, 4));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {32, 64, 128, 256, 512, 1024}) {
      CK(hipMemsetAsync(ctr, 0, 4, s));
      int nn = n, ph = phases;
      void* args[] = {&a, &b, &nn, &ph, &ctr};
      CK(hipEventRecord(e0, s));
      CK(hipLaunchCooperativeKernel((const void*)persistent, dim3(G), dim3(256), args, 0, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<double> h(4); CK(hipMemcpy(h.data(), a, 32, hipMemcpyDeviceToHost));
      printf("n=%8d persistent G=%4d: %.2f us per phase (check %.0f)\n", n, G, 1e3 * ms / phases, h[0]);
    }
    for (int G : {64, 256, 1024}) {
      CK(hipEventRecord(e0, s));
      for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(step, dim3(G), dim3(256), 0, s, (p & 1) ? a : b, (p & 1) ? b : a, n);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("n=%8d kernel chain G=%4d: %.2f us per kernel\n", n, G, 1e3 * ms / phases);
    }
    // same chain replayed as a graph
    {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(step, dim3(256), dim3(256), 0, s, (p & 1) ? a : b, (p & 1) ? b : a, n);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("n=%8d graph of %d kernels G=256: %.2f us per kernel\n", n, phases, 1e3 * ms / phases);
    }
    hipFree(a); hipFree(b); hipFree(ctr);
  }
  return 0;
}
