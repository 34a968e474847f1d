// Micro-probe: does the time of a fixed serial dependent chain (one lane per wave) depend on how many waves run?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void chain(float* out, int iters, int lds_ops) {
  __shared__ float tab[256];
  tab[threadIdx.x] = threadIdx.x * 0.5f; tab[threadIdx.x + 64] = 1.f; tab[threadIdx.x + 128] = 2.f; tab[threadIdx.x + 192] = 3.f;
  __syncthreads();
  if (threadIdx.x != 0) return;
  float v = out[blockIdx.x & 255];
  unsigned idx = blockIdx.x & 255;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    v = v * 1.0001f + 0.5f;
    if (lds_ops) { idx = (idx * 5 + (unsigned)v) & 255; v += tab[idx]; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x & 255] = v;
  if (blockIdx.x == 0) ((unsigned long long*)(out + 1024))[0] = t1 - t0;
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 16);
  hipMemset(d, 0, 1 << 16);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int lds = 0; lds < 2; lds++)
    for (int grid : {1, 64, 256, 1024, 2048, 4096, 8192}) {
      hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, d, 1000, lds);
      hipEventRecord(a);
      hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, d, 200000, lds);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      unsigned long long cyc;
      hipMemcpy(&cyc, d + 1024, 8, hipMemcpyDeviceToHost);
      printf("lds=%d waves=%5d : %8.3f ms  (%.1f ns/iter, %.1f shader cycles/iter => %.2f GHz)\n", lds, grid, ms, ms * 1e6 / 200000, (double)cyc / 200000,
             (double)cyc / (ms * 1e6));
    }
  return 0;
}
