// Micro-probe: how long does an (almost) empty kernel take as a function of dynamic LDS size, VGPR budget and block size?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int kRegs>
__global__ __launch_bounds__(256) void probe(float* out, int n) {
  extern __shared__ float lds[];
  if (n == 12345) {   // never true: keeps kRegs live registers and the LDS symbol referenced
    float v[kRegs];
    for (int i = 0; i < kRegs; i++) v[i] = out[i + threadIdx.x];
    float s = 0;
    for (int i = 0; i < kRegs; i++) s += v[i] * v[(i * 7) % kRegs];
    lds[threadIdx.x] = s;
    __syncthreads();
    out[threadIdx.x] = lds[(threadIdx.x + 1) & 255];
  }
}
template <int kRegs>
float run(int grid, int block, size_t lds, float* d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)probe<kRegs>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(probe<kRegs>, dim3(grid), dim3(block), lds, 0, d, 0);
  hipEventRecord(a);
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(probe<kRegs>, dim3(grid), dim3(block), lds, 0, d, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 10 * 1000;
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  const int grid = 2040;
  size_t ldss[] = {0, 8192, 16384, 27648, 32768, 35456, 40960, 55552, 65536};
  for (size_t l : ldss) printf("regs=8   block=256 lds=%6zu : %8.1f us\n", l, run<8>(grid, 256, l, d));
  for (size_t l : {(size_t)0, (size_t)35456}) printf("regs=120 block=256 lds=%6zu : %8.1f us\n", l, run<120>(grid, 256, l, d));
  for (size_t l : {(size_t)0, (size_t)35456}) printf("regs=8   block=64  lds=%6zu : %8.1f us\n", l, run<8>(grid * 4, 64, l, d));
  return 0;
}
