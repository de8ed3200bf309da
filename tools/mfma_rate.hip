// Diagnostic: what this device sustains on a stream of v_mfma_f32_32x32x2_f32 shaped like the convolution kernel's
// (4 independent accumulators per wave, 4 waves per workgroup, 2 workgroups per CU), with pieces of that kernel's
// structure added one at a time.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: bare stream.  1: + 78.8 KiB of dynamic LDS per workgroup (2 workgroups per CU by LDS, as the kernel).
// 2: + one barrier per 216 MFMAs.  3: + the epilogue's 64 strided stores per wave.  4: + LDS operand reads (1 + 4 per 4 MFMAs).
template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_stream(float* out, int iters, long long ovol) {
  extern __shared__ float lds[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
  if (MODE >= 4) {
    for (int k = threadIdx.x; k < 9856; k += 256) lds[k] = k * 1e-5f;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 54; ++k) {
      float av = a + k;
      if (MODE >= 4) av = lds[6400 + k * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float bv = b + i;
        if (MODE >= 4) bv = lds[((k % 9) * 10 + i) * 40 + (lane & 31) + (k % 3) + 3];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
      }
    }
    if (MODE >= 2) __syncthreads();
  }
  if (MODE >= 3) {
    float* yb = out + (blockIdx.x % 64) * 1024 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) yb[(8 * (v >> 2) + (v & 3) + 4 * (lane >> 5)) * ovol + i * 312] = acc[i][v];
  } else {
    float s = 0;
    for (int i = 0; i < 4; ++i)
      for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int MODE>
void run(float* out, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = MODE >= 1 ? 78848 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_stream<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 78848);
  for (int grid : {2560, 2880}) {
    const int iters = 16;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(mfma_stream<MODE>, dim3(grid), dim3(256), lds, 0, out, iters, 1437696LL);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms / 10 < best ? ms / 10 : best;
    }
    const double flops = 4096.0 * 216 * iters * 4 * grid;
    printf("mode %d grid %d: %.3f ms  %.1f TFLOP/s = %.3f of 157.3\n", MODE, grid, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 64LL * 1437696 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  run<0>(out, e0, e1);
  run<1>(out, e0, e1);
  run<2>(out, e0, e1);
  run<3>(out, e0, e1);
  run<4>(out, e0, e1);
  return 0;
}
