// Dense photometric box alignment for the Stereo R-CNN detect-under-attack scripts.
//
// The reference calls the UPSTREAM `dense_align.align_parallel(calib, scale, im_left, im_right, boxes, kpts, poses)`
// (attack/Stereo-RCNN/predict_and_save_pgd.py:381); that module lives in the Stereo R-CNN checkout, not in the reference
// tree.  What is built here is the published algorithm (Stereo R-CNN, Li et al., CVPR 2019, sec. 5 "dense 3D box
// alignment"), stated in include/advengine.h: for every object the centre depth z is searched by ENUMERATION so that the
// sum of squared photometric differences between the left pixels of the object's valid region and the right image, sampled
// at the disparity fb / (z + dz(u)) of each pixel column, is minimal.  Parity: bit-exact against the oracle
// (oracle.c: orc_dense_align_cost), UNPINNED against the upstream module.
//
// Shape of the work: n objects (a handful) x K candidate depths (50 coarse, then 20 fine) x a few thousand pixels x 3
// channels - latency-bound, nothing for the matrix cores.  One 256-lane workgroup per (candidate, object); lanes stride
// over the region's pixels (consecutive lanes = consecutive columns of one row: coalesced reads of both images), keep a
// float32 partial sum and a pixel count, and the workgroup reduces them with wave-level shuffles (__shfl_down over the 64
// lanes) followed by a four-entry LDS pass - a fixed order, so the result is reproducible and the oracle can restate it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

#pragma clang fp contract(off)

namespace {

constexpr int kAlignBlock = 256;

__global__ __launch_bounds__(kAlignBlock) void dense_align_cost_kernel(const float* __restrict__ left, const float* __restrict__ right, int h,
                                                                       int w, const int32_t* __restrict__ roi, const float* __restrict__ dz,
                                                                       int dz_stride, const float* __restrict__ z_center, float fb, float step,
                                                                       int k_cand, float* __restrict__ cost_out) {
  __shared__ float s_sum[kAlignBlock / 64];
  __shared__ int s_cnt[kAlignBlock / 64];
  const int k = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int u0 = roi[4 * b], v0 = roi[4 * b + 1], u1 = roi[4 * b + 2], v1 = roi[4 * b + 3];
  const int rw = u1 - u0, rh = v1 - v0;
  // candidate k of object b: z_center + (k - (K-1)/2) * step, rounded as written (float32)
  const float z = z_center[b] + (static_cast<float>(k) - 0.5f * static_cast<float>(k_cand - 1)) * step;
  const long long plane = static_cast<long long>(h) * w;
  float acc = 0.0f;
  int cnt = 0;
  const int total = rw > 0 && rh > 0 ? rw * rh : 0;
  for (int p = t; p < total; p += kAlignBlock) {
    const int r = p / rw;
    const int c = p - r * rw;
    const int u = u0 + c, v = v0 + r;
    const float depth = z + dz[static_cast<long long>(b) * dz_stride + c];
    if (!(depth > 0.0f)) continue;
    const float x = static_cast<float>(u) - fb / depth;  // column in the right image
    const float xf = floorf(x);
    if (!(xf >= 0.0f) || !(xf < static_cast<float>(w - 1))) continue;
    const int x0 = static_cast<int>(xf);
    const float wr = x - xf, wl = 1.0f - wr;
    const long long row = static_cast<long long>(v) * w;
    float e = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float l = left[ch * plane + row + u];
      const float a = wl * right[ch * plane + row + x0];
      const float bb = wr * right[ch * plane + row + x0 + 1];
      const float d = l - (a + bb);
      e = e + d * d;
    }
    acc = acc + e;
    ++cnt;
  }
  // wave-level reduction (64 lanes), then the four wave results in order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    acc = acc + __shfl_down(acc, off, 64);
    cnt = cnt + __shfl_down(cnt, off, 64);
  }
  if ((t & 63) == 0) {
    s_sum[t >> 6] = acc;
    s_cnt[t >> 6] = cnt;
  }
  __syncthreads();
  if (t == 0) {
    float s = s_sum[0];
    int n = s_cnt[0];
#pragma unroll
    for (int i = 1; i < kAlignBlock / 64; ++i) {
      s = s + s_sum[i];
      n = n + s_cnt[i];
    }
    // mean squared difference per pixel; a candidate that leaves fewer than a quarter of the region inside the right image
    // cannot win
    cost_out[static_cast<long long>(b) * k_cand + k] = (n > 0 && 4 * n >= total) ? s / static_cast<float>(n) : INFINITY;
  }
}

__global__ void dense_align_argmin_kernel(const float* __restrict__ cost, int n, int k_cand, const float* __restrict__ z_center, float step,
                                          float* __restrict__ z_out, float* __restrict__ cost_min_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  int best = -1;
  float best_cost = INFINITY;
  for (int k = 0; k < k_cand; ++k) {  // first minimum wins; NaN and inf never do
    const float c = cost[static_cast<long long>(b) * k_cand + k];
    if (c < best_cost) {
      best_cost = c;
      best = k;
    }
  }
  const float kk = best >= 0 ? static_cast<float>(best) : 0.5f * static_cast<float>(k_cand - 1);
  z_out[b] = z_center[b] + (kk - 0.5f * static_cast<float>(k_cand - 1)) * step;
  cost_min_out[b] = best >= 0 ? best_cost : INFINITY;
}

inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

}  // namespace

extern "C" {

int adv_dense_align_cost_f32(const float* left, const float* right, int h, int w, int n, const int32_t* roi, const float* dz, int dz_stride,
                             const float* z_center, float fb, float step, int k, float* cost_out, adv_stream_t stream) {
  if (left == nullptr || right == nullptr || roi == nullptr || dz == nullptr || z_center == nullptr || cost_out == nullptr) return ADV_EINVAL;
  if (h < 1 || w < 2 || n < 1 || n > 65535 || k < 1 || dz_stride < 1 || static_cast<long long>(h) * w > (1LL << 30)) return ADV_EINVAL;
  if (!aligned4(left) || !aligned4(right) || !aligned4(roi) || !aligned4(dz) || !aligned4(z_center) || !aligned4(cost_out)) return ADV_EALIGN;
  hipLaunchKernelGGL(dense_align_cost_kernel, dim3(k, n), dim3(kAlignBlock), 0, static_cast<hipStream_t>(stream), left, right, h, w, roi, dz,
                     dz_stride, z_center, fb, step, k, cost_out);
  return adv_internal_finish_launch();
}

int adv_dense_align_argmin_f32(const float* cost, int n, int k, const float* z_center, float step, float* z_out, float* cost_min_out,
                               adv_stream_t stream) {
  if (cost == nullptr || z_center == nullptr || z_out == nullptr || cost_min_out == nullptr || n < 1 || k < 1) return ADV_EINVAL;
  if (!aligned4(cost) || !aligned4(z_center) || !aligned4(z_out) || !aligned4(cost_min_out)) return ADV_EALIGN;
  hipLaunchKernelGGL(dense_align_argmin_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), cost, n, k, z_center, step,
                     z_out, cost_min_out);
  return adv_internal_finish_launch();
}

}  // extern "C"
