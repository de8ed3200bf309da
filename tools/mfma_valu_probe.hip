// Do float32 matrix instructions (v_mfma_f32_32x32x2_f32) and float32 vector instructions of the OTHER wave of a SIMD execute at the same time on
// gfx950?  A workgroup = 8 waves (two per SIMD): waves 0-3 issue NM matrix instructions (4 independent accumulators), waves 4-7 NV vector fmas
// (8 independent chains).  Timed: matrix waves alone, vector waves alone, both.  "both ~ max" = separate pipes, "both ~ sum" = one pipe.
// Also: the partner issuing LDS reads instead of vector work.   hipcc -O2 --offload-arch=gfx950 -o tools/_build/mfma_valu_probe tools/mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>      // bit 0: matrix waves work, bit 1: vector waves work, bit 2: partner waves read LDS instead, bit 3: the matrix waves issue bf16 matrix instructions
__global__ __launch_bounds__(512, 2) void probe(float* out, int nm, int nv) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  float r = 0.0f;
  if (wave < 4) {
    if ((MODE & 9) == 9) {      // v_mfma_f32_32x32x16_bf16: 32 cycles each - twice as many for the same pipe time
      f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
      bf16x8 x, y;
      for (int k = 0; k < 8; ++k) x[k] = static_cast<__bf16>(lane * 0.01f + k), y[k] = static_cast<__bf16>(1.0f + lane * 0.001f);
      for (int i = 0; i < 2 * nm; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (MODE & 1) {
      f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
      const float x = lane * 0.01f, y = 1.0f + lane * 0.001f;
      for (int i = 0; i < nm; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (MODE & 2) {
      float c[8];
      for (int k = 0; k < 8; ++k) c[k] = lane * 0.001f + k;
      const float m = 1.0001f, a = 0.0001f;
      for (int i = 0; i < nv; i += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = __builtin_fmaf(c[k], m, a);
      for (int k = 0; k < 8; ++k) r += c[k];
    }
    if (MODE & 4) {
      float s = 0.0f;
      int at = lane;
      for (int i = 0; i < nv; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds[(at + 64 * k) & 4095];
        at += 7;
      }
      r += s;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE>
float run(float* out, int nm, int nv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, nm, nv);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, nm, nv);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int nm = 8192, nv = 8192 * 8;      // 8192 x 64 cycles of matrix work; 65536 vector fmas (4 cycles of issue each for one wave)
  const float m = run<1>(out, nm, nv), v = run<2>(out, nm, nv), b = run<3>(out, nm, nv), l = run<4>(out, nm, nv), ml = run<5>(out, nm, nv);
  const float hm = run<9>(out, nm, nv), hb = run<11>(out, nm, nv);
  printf("{\"probe\": \"mfma_f32_vs_valu\", \"matrix_waves_alone_ms\": %.4f, \"vector_waves_alone_ms\": %.4f, \"both_ms\": %.4f, \"sum_ms\": %.4f, \"max_ms\": %.4f, "
         "\"lds_read_waves_alone_ms\": %.4f, \"matrix_plus_lds_reads_ms\": %.4f, \"matrix_cycles_per_instruction_at_2.4GHz\": %.1f, "
         "\"bf16_matrix_waves_alone_ms\": %.4f, \"bf16_matrix_plus_vector_ms\": %.4f, \"bf16_sum_ms\": %.4f, \"bf16_max_ms\": %.4f}\n",
         m, v, b, m + v, m > v ? m : v, l, ml, m * 2.4e6 / nm, hm, hb, hm + v, hm > v ? hm : v);
  return 0;
}
