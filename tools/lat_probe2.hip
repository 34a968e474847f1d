// Micro-probe 2: per-operation cost inside one wavefront (one wave on the chip), unrolled x32 so that loop overhead vanishes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X
__global__ __launch_bounds__(64) void probe(unsigned* buf, const unsigned* __restrict__ gtab, int iters, int mode, unsigned long long* cyc) {
  __shared__ unsigned tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = gtab[i];
  __syncthreads();
  unsigned a = threadIdx.x * 7 + gtab[threadIdx.x & 3], c = gtab[5] | 1;
  unsigned long long q = ((unsigned long long)gtab[7] << 32) | gtab[9];
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (mode == 0) { REP32(a = a + c; asm volatile("" : "+v"(a));) }                                  // dependent v_add
    else if (mode == 1) { REP32(a = __umul24(a, c) + 1; asm volatile("" : "+v"(a));) }               // dependent v_mad_u32_u24
    else if (mode == 2) { REP32(a = a * c; asm volatile("" : "+v"(a));) }                              // dependent v_mul_lo_u32
    else if (mode == 3) { REP32(q = q << (c & 7); asm volatile("" : "+v"(q));) }                       // dependent v_lshlrev_b64
    else if (mode == 4) { REP32(a = (a & 1) ? a + c : a ^ c; asm volatile("" : "+v"(a));) }           // compare + select
    else if (mode == 5) { REP32(if (a & 1) { a = a * 3 + c; asm volatile("" : "+v"(a)); } a += 1; asm volatile("" : "+v"(a));) }   // divergent-capable branch, all lanes same way? (lane-dependent)
    else if (mode == 6) { REP32(a = tab[a & 1023]; asm volatile("" : "+v"(a));) }                     // dependent LDS read
    else if (mode == 7) { REP32(a = tab[a & 1023] + c; a ^= a >> 3; asm volatile("" : "+v"(a));) }    // LDS read + 2 ops
    else if (mode == 8) { unsigned b2 = a ^ 5; REP32(a = a + c; b2 = b2 + c; asm volatile("" : "+v"(a), "+v"(b2));) a ^= b2; }  // two independent chains
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  buf[threadIdx.x] = a + (unsigned)q;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  unsigned *d, *g; unsigned long long* c;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&g, 4096); (void)hipMalloc(&c, 8);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = (i * 37u + 11u) & 1023;
  (void)hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  const char* names[] = {"v_add dependent", "v_mad_u32_u24 dependent", "v_mul_lo_u32 dependent", "v_lshlrev_b64 dependent", "cmp+select dependent", "if-branch (lane dependent) + add", "LDS read dependent", "LDS read + 2 ALU", "two independent v_add chains (per pair)"};
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int mode = 0; mode < 9; mode++) {
    const int iters = 4000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, 10, mode, c);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, g, iters, mode, c);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long cyc; (void)hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    printf("%-42s %7.2f ns/op  %7.2f ticks/op\n", names[mode], ms * 1e6 / iters / 32, (double)cyc / iters / 32);
  }
  return 0;
}
