// How v_cvt_pk_u8_f32 rounds, saturates and treats NaN (decides whether the output stage may use it: tile_kernels.hip PixelToRgba8).
// hipcc --offload-arch=gfx950 -O2 tools/cvt_probe.hip -o gpurun_out/cvt_probe && gpurun_out/cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(unsigned* o, const float* in, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  const float v[] = {-1.f, -0.4f, 0.f, 0.49f, 0.5f, 0.51f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 126.5f, 127.5f, 254.49f, 254.5f, 254.51f, 255.f, 255.4f, 255.6f, 300.f, 1e9f, NAN, INFINITY, -INFINITY};
  const int n = sizeof(v) / sizeof(v[0]);
  float* di; unsigned* dout; unsigned out[64];
  hipMalloc(&di, sizeof(v)); hipMalloc(&dout, n * 4);
  hipMemcpy(di, v, sizeof(v), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di, n);
  hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) printf("%12g -> %u\n", v[i], out[i]);
  return 0;
}
