// Diagnostic: what this device sustains on a bare stream of v_mfma_f32_32x32x2_f32 (4 independent accumulators per wave,
// 4 waves per workgroup, 512 workgroups = 2 per CU), no memory traffic at all.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void mfma_stream(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 54; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + k, b + i, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int v = 0; v < 16; ++v) s += acc[i][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 2880 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {512, 2880, 2560}) {
    const int iters = 16;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(mfma_stream, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 10;
      const double flops = 4096.0 * 216 * iters * 4 * grid;
      printf("grid %d: %.3f ms  %.1f TFLOP/s = %.3f of 157.3\n", grid, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
  }
  return 0;
}
