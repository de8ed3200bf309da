// What does an LDS-DMA lane whose buffer offset fails the range check do to LDS (zeros, or nothing)?  Does a 16-byte LDS-DMA accept a source
// address that is only 4-byte aligned?  (questions behind csrc/wino4.hip's input staging; prints the answers)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const float* src, int nbytes, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[512];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = -7.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // lanes 0-31 in range (shifted by `shift` floats), lanes 32-47 beyond num_records, lanes 48-63 the 0xFFFFFF00 marker
  int vo = lane < 32 ? (lane * 4 + shift) * 4 : lane < 48 ? nbytes + (lane - 32) * 16 : static_cast<int>(0xFFFFFF00u);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, vo, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = 100.0f + i;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; shift += 3) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024 - 256, o, shift);
    float r[256];
    hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 0; i < 128; ++i) ok = ok && r[i] == 100.0f + i + shift;
    printf("{\"probe\": \"lds_dma\", \"source_shift_floats\": %d, \"in_range_lanes_correct\": %s, \"beyond_records_lane32_first\": %g, \"marker_lane48_first\": %g, \"error\": \"%s\"}\n",
           shift, ok ? "true" : "false", r[128], r[192], hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
