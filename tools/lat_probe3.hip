// Micro-probe 3: dependent / independent VALU issue and branch costs in one wavefront, no inline-asm barriers (chains the
// compiler cannot fold).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ __launch_bounds__(64) void probe(unsigned* buf, const unsigned* __restrict__ gtab, int iters, int mode, unsigned long long* cyc) {
  unsigned a = threadIdx.x * 7 + gtab[threadIdx.x & 3], c = gtab[5] | 1, b2 = a ^ 0x55, b3 = a + 77, b4 = a * 3;
  const unsigned lane_odd = threadIdx.x & 1;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (mode == 0) { REP16(a = __umul24(a, c) + 1u;) }                                                   // 16 dependent mads
    else if (mode == 1) { REP16(a = __umul24(a, c) + 1u; b2 = __umul24(b2, c) + 3u;) }                  // 2 chains
    else if (mode == 2) { REP16(a = __umul24(a, c) + 1u; b2 = __umul24(b2, c) + 3u; b3 = __umul24(b3, c) + 5u; b4 = __umul24(b4, c) + 7u;) }   // 4 chains
    else if (mode == 3) { REP16(a = (a ^ c) + (a >> 3);) }                                              // xor, shift, add: 3 dependent-ish ops
    else if (mode == 4) { REP16(if (a & 0x10000) a = __umul24(a, 3u) + 7u; a = __umul24(a, c) + 1u;) }   // data-dependent branch, lanes differ
    else if (mode == 5) { REP16(if (__builtin_amdgcn_readfirstlane(a) & 0x10000) a = __umul24(a, 3u) + 7u; a = __umul24(a, c) + 1u;) }   // uniform branch
    else if (mode == 6) { REP16(a = (a & 0x10000) ? __umul24(a, 3u) + 7u : a; a = __umul24(a, c) + 1u;) }   // select instead
    else if (mode == 7) { unsigned long long q = ((unsigned long long)a << 32) | b2; REP16(q = (q << (c & 3)) ^ (q >> 7);) a = (unsigned)q ^ (unsigned)(q >> 32); }   // 64-bit shifts
    else if (mode == 8) { REP16(if (lane_odd && (a & 0x10000)) a = __umul24(a, 3u) + 7u; a = __umul24(a, c) + 1u;) }   // branch that only odd lanes can take
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  buf[threadIdx.x] = a + b2 + b3 + b4;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  unsigned *d, *g; unsigned long long* c;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&g, 4096); (void)hipMalloc(&c, 8);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = (i * 37u + 11u) & 1023;
  (void)hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  const char* names[] = {"1 chain: mad24 (per mad)", "2 chains (per pair)", "4 chains (per quad)", "xor/shift/add step", "lane-divergent if + mad", "uniform if + mad", "select + mad", "64-bit shl/shr/xor step", "odd-lane if + mad"};
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int mode = 0; mode < 9; mode++) {
    const int iters = 20000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, 10, mode, c);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, iters, mode, c);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long cyc; (void)hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %7.2f ns/step  %7.2f ticks/step\n", names[mode], ms * 1e6 / iters / 16, (double)cyc / iters / 16);
  }
  return 0;
}
