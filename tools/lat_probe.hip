// Micro-probe: latency of one dependent step (one lane per wave, one wave) for the building blocks of a serial entropy decoder on
// gfx950: VALU op, LDS read, L1-hit global read, scalar (constant-cache) read, v_readlane, SALU op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(unsigned* buf, const unsigned* __restrict__ gtab, int iters, int mode, unsigned long long* cyc) {
  __shared__ unsigned tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = gtab[i];
  __syncthreads();
  unsigned idx = threadIdx.x * 7 & 1023;
  unsigned lanev = gtab[threadIdx.x];   // per-lane table entry for the readlane probe
  unsigned long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {          // VALU dependent chain: mad
    for (int i = 0; i < iters; i++) idx = idx * 5u + 1u;
  } else if (mode == 1) {   // LDS dependent read
    for (int i = 0; i < iters; i++) idx = tab[idx & 1023];
  } else if (mode == 2) {   // global dependent read (4 KB table: L1 / TCP hits)
    for (int i = 0; i < iters; i++) idx = gtab[idx & 1023];
  } else if (mode == 3) {   // scalar dependent read (uniform index -> s_load)
    unsigned s = __builtin_amdgcn_readfirstlane(idx);
    for (int i = 0; i < iters; i++) s = __builtin_amdgcn_readfirstlane(gtab[s & 1023]);
    idx = s;
  } else if (mode == 4) {   // v_readlane dependent chain (index in SGPR)
    unsigned s = __builtin_amdgcn_readfirstlane(idx);
    for (int i = 0; i < iters; i++) s = __builtin_amdgcn_readlane(lanev, s & 63);
    idx = s;
  } else if (mode == 5) {   // SALU dependent chain
    unsigned s = __builtin_amdgcn_readfirstlane(idx);
    for (int i = 0; i < iters; i++) s = s * 5u + 1u;
    idx = s;
  } else if (mode == 6) {   // ds_bpermute dependent chain
    for (int i = 0; i < iters; i++) idx = __builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)lanev);
  } else if (mode == 7) {   // LDS read + 8 dependent VALU ops (a token-like step)
    for (int i = 0; i < iters; i++) { unsigned e = tab[idx & 1023]; idx = ((e >> 3) * 5u + (e & 7u)) ^ (idx >> 2); idx = idx * 3u + (e >> 9); idx ^= idx >> 5; idx += e; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  buf[threadIdx.x] = idx;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  unsigned *d, *g; unsigned long long* c;
  hipMalloc(&d, 4096); hipMalloc(&g, 4096); hipMalloc(&c, 8);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = (i * 37u + 11u) & 1023;   // a permutation-ish chain inside the table
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  const char* names[] = {"VALU mad chain", "LDS read chain", "global read chain (L1 hit)", "scalar read chain (K$)", "v_readlane chain", "SALU mul-add chain", "ds_bpermute chain", "LDS read + ~10 VALU"};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 8; mode++) {
    const int iters = 100000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, 1000, mode, c);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, iters, mode, c);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long cyc; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    printf("%-30s %7.1f ns/step  %7.1f counter ticks/step\n", names[mode], ms * 1e6 / iters, (double)cyc / iters);
  }
  return 0;
}
