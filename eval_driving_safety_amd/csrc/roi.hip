// Stereo R-CNN RoI-path natives for gfx950: RoIAlign forward/backward and greedy NMS.
// Small, latency-bound kernels (a few hundred RoIs x 256 channels x 7x7 / 14x14 bins); the design points are
// coalescing along the bin index, wave64-wide suppression masks for NMS, and determinism where it is cheap.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;

struct Taps {
  int y_low, x_low, y_high, x_high;
  float w1, w2, w3, w4;
  bool valid;
};

// maskrcnn-benchmark bilinear_interpolate / bilinear_interpolate_gradient pre-computation
__device__ __forceinline__ Taps taps_at(int height, int width, float y, float x) {
  Taps t;
  t.valid = !(y < -1.0f || y > static_cast<float>(height) || x < -1.0f || x > static_cast<float>(width));
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  t.y_low = static_cast<int>(y);
  t.x_low = static_cast<int>(x);
  if (t.y_low >= height - 1) {
    t.y_high = t.y_low = height - 1;
    y = static_cast<float>(t.y_low);
  } else {
    t.y_high = t.y_low + 1;
  }
  if (t.x_low >= width - 1) {
    t.x_high = t.x_low = width - 1;
    x = static_cast<float>(t.x_low);
  } else {
    t.x_high = t.x_low + 1;
  }
  const float ly = y - static_cast<float>(t.y_low), lx = x - static_cast<float>(t.x_low);
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  t.w1 = hy * hx;
  t.w2 = hy * lx;
  t.w3 = ly * hx;
  t.w4 = ly * lx;
  return t;
}

struct Bin {
  float start_h, start_w, bin_h, bin_w;
  int grid_h, grid_w, batch;
};

__device__ __forceinline__ Bin bin_of(const float* roi, float scale, int ph, int pw, int sampling_ratio) {
  Bin b;
  b.batch = static_cast<int>(roi[0]);
  b.start_w = roi[1] * scale;
  b.start_h = roi[2] * scale;
  const float end_w = roi[3] * scale, end_h = roi[4] * scale;
  const float rw = fmaxf(end_w - b.start_w, 1.0f), rh = fmaxf(end_h - b.start_h, 1.0f);
  b.bin_h = rh / static_cast<float>(ph);
  b.bin_w = rw / static_cast<float>(pw);
  b.grid_h = sampling_ratio > 0 ? sampling_ratio : static_cast<int>(ceilf(rh / static_cast<float>(ph)));
  b.grid_w = sampling_ratio > 0 ? sampling_ratio : static_cast<int>(ceilf(rw / static_cast<float>(pw)));
  return b;
}

// one lane per output element (r, c, ph, pw); the bin index is fastest, so a wave reads neighbouring samples
__global__ __launch_bounds__(kBlock) void roi_align_fwd(const float* __restrict__ feat, const float* __restrict__ rois,
                                                        float* __restrict__ out, int C, int H, int W, long long total, int PH,
                                                        int PW, float scale, int sampling_ratio) {
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int pw = static_cast<int>(i % PW);
    const int ph = static_cast<int>((i / PW) % PH);
    const int c = static_cast<int>((i / (static_cast<long long>(PW) * PH)) % C);
    const long long r = i / (static_cast<long long>(PW) * PH * C);
    const Bin b = bin_of(rois + r * 5, scale, PH, PW, sampling_ratio);
    const float* plane = feat + (static_cast<long long>(b.batch) * C + c) * H * W;
    const float count = static_cast<float>(b.grid_h * b.grid_w);
    float acc = 0.0f;
    for (int iy = 0; iy < b.grid_h; ++iy) {
      const float y = b.start_h + static_cast<float>(ph) * b.bin_h + (static_cast<float>(iy) + 0.5f) * b.bin_h / static_cast<float>(b.grid_h);
      for (int ix = 0; ix < b.grid_w; ++ix) {
        const float x = b.start_w + static_cast<float>(pw) * b.bin_w + (static_cast<float>(ix) + 0.5f) * b.bin_w / static_cast<float>(b.grid_w);
        const Taps t = taps_at(H, W, y, x);
        float v = 0.0f;
        if (t.valid) {
          const float v1 = plane[t.y_low * W + t.x_low], v2 = plane[t.y_low * W + t.x_high];
          const float v3 = plane[t.y_high * W + t.x_low], v4 = plane[t.y_high * W + t.x_high];
          v = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
        }
        acc += v;
      }
    }
    out[i] = acc / count;
  }
}

__global__ __launch_bounds__(kBlock) void roi_align_bwd(const float* __restrict__ gout, const float* __restrict__ rois,
                                                        float* gfeat, int C, int H, int W, long long total, int PH, int PW,
                                                        float scale, int sampling_ratio) {
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int pw = static_cast<int>(i % PW);
    const int ph = static_cast<int>((i / PW) % PH);
    const int c = static_cast<int>((i / (static_cast<long long>(PW) * PH)) % C);
    const long long r = i / (static_cast<long long>(PW) * PH * C);
    const Bin b = bin_of(rois + r * 5, scale, PH, PW, sampling_ratio);
    float* plane = gfeat + (static_cast<long long>(b.batch) * C + c) * H * W;
    const float count = static_cast<float>(b.grid_h * b.grid_w);
    const float g = gout[i];
    for (int iy = 0; iy < b.grid_h; ++iy) {
      const float y = b.start_h + static_cast<float>(ph) * b.bin_h + (static_cast<float>(iy) + 0.5f) * b.bin_h / static_cast<float>(b.grid_h);
      for (int ix = 0; ix < b.grid_w; ++ix) {
        const float x = b.start_w + static_cast<float>(pw) * b.bin_w + (static_cast<float>(ix) + 0.5f) * b.bin_w / static_cast<float>(b.grid_w);
        const Taps t = taps_at(H, W, y, x);
        if (t.valid) {
          atomicAdd(plane + t.y_low * W + t.x_low, g * t.w1 / count);
          atomicAdd(plane + t.y_low * W + t.x_high, g * t.w2 / count);
          atomicAdd(plane + t.y_high * W + t.x_low, g * t.w3 / count);
          atomicAdd(plane + t.y_high * W + t.x_high, g * t.w4 / count);
        }
      }
    }
  }
}

// ---- NMS: wave64 suppression masks.  Block (row tile i, col tile j), 64 lanes: lane l owns box 64*i + l and
// tests it against the 64 boxes of tile j (staged in LDS); bit k of its mask = IoU(box_l, box_{64j+k}) > thresh.
__device__ __forceinline__ float iou_legacy(const float* a, const float* b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float w = fmaxf(right - left + 1.0f, 0.0f), h = fmaxf(bottom - top + 1.0f, 0.0f);
  const float inter = w * h;
  const float sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
  const float sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
  return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(64) void nms_mask(const float* __restrict__ boxes, int n, float thresh, unsigned long long* mask, int col_blocks) {
  const int row = blockIdx.y, col = blockIdx.x;
  if (row > col) return;  // only later (lower-score) boxes can be suppressed by earlier ones
  __shared__ float tile[64 * 4];
  const int col_n = min(n - col * 64, 64), row_n = min(n - row * 64, 64);
  if (static_cast<int>(threadIdx.x) < col_n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[threadIdx.x * 4 + k] = boxes[(col * 64 + threadIdx.x) * 4 + k];
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < row_n) {
    const int me = row * 64 + threadIdx.x;
    float mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[k] = boxes[me * 4 + k];
    unsigned long long bits = 0ULL;
    const int start = (row == col) ? static_cast<int>(threadIdx.x) + 1 : 0;
    for (int k = start; k < col_n; ++k)
      if (iou_legacy(mine, tile + k * 4) > thresh) bits |= 1ULL << k;
    mask[static_cast<long long>(me) * col_blocks + col] = bits;
  }
}

// sequential greedy scan by one wave: removed[] (col_blocks words) lives in LDS, lane j owns words j, j+64, ...
__global__ __launch_bounds__(64) void nms_scan(const unsigned long long* __restrict__ mask, int n, int col_blocks, long long* keep, int* num_keep) {
  extern __shared__ unsigned long long removed[];
  for (int j = threadIdx.x; j < col_blocks; j += 64) removed[j] = 0ULL;
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const int blk = i >> 6, bit = i & 63;
    const bool dead = (removed[blk] >> bit) & 1ULL;  // same address for all lanes: LDS broadcast
    if (!dead) {
      if (threadIdx.x == 0) keep[kept] = i;
      ++kept;
      for (int j = blk + threadIdx.x; j < col_blocks; j += 64) removed[j] |= mask[static_cast<long long>(i) * col_blocks + j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = kept;
}

inline int finish() { return adv_internal_finish_launch(); }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

inline int check_roi(const void* a, const void* b, const void* c, int B, int C, int H, int W, int R, int PH, int PW) {
  if (!a || !b || !c) return ADV_EINVAL;
  if (B < 1 || C < 1 || H < 1 || W < 1 || R < 0 || PH < 1 || PW < 1) return ADV_EINVAL;
  if (!aligned4(a) || !aligned4(b) || !aligned4(c)) return ADV_EALIGN;
  return ADV_OK;
}

inline unsigned grid_for(long long total) {
  long long g = (total + kBlock - 1) / kBlock;
  return static_cast<unsigned>(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" {

int adv_roi_align_fwd_f32(const float* feat, const float* rois, float* out, int b, int c, int h, int w, int r, int ph, int pw,
                          float spatial_scale, int sampling_ratio, adv_stream_t stream) {
  const int rc = check_roi(feat, rois, out, b, c, h, w, r, ph, pw);
  if (rc != ADV_OK) return rc;
  if (r == 0) return ADV_OK;
  const long long total = static_cast<long long>(r) * c * ph * pw;
  hipLaunchKernelGGL(roi_align_fwd, dim3(grid_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), feat, rois, out, c, h, w,
                     total, ph, pw, spatial_scale, sampling_ratio);
  return finish();
}

int adv_roi_align_bwd_f32(const float* grad_out, const float* rois, float* grad_feat, int b, int c, int h, int w, int r, int ph,
                          int pw, float spatial_scale, int sampling_ratio, adv_stream_t stream) {
  const int rc = check_roi(grad_out, rois, grad_feat, b, c, h, w, r, ph, pw);
  if (rc != ADV_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(grad_feat, 0, static_cast<size_t>(b) * c * h * w * sizeof(float), st) != hipSuccess) return ADV_ELAUNCH;
  if (r == 0) return ADV_OK;
  const long long total = static_cast<long long>(r) * c * ph * pw;
  hipLaunchKernelGGL(roi_align_bwd, dim3(grid_for(total)), dim3(kBlock), 0, st, grad_out, rois, grad_feat, c, h, w, total, ph, pw,
                     spatial_scale, sampling_ratio);
  return finish();
}

int adv_nms_f32(const float* boxes, int n, float thresh, int64_t* keep_out, int32_t* num_keep_out, uint64_t* workspace,
                adv_stream_t stream) {
  if (!num_keep_out || n < 0 || (n > 0 && (!boxes || !keep_out || !workspace))) return ADV_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) return hipMemsetAsync(num_keep_out, 0, sizeof(int32_t), st) == hipSuccess ? ADV_OK : ADV_ELAUNCH;
  const int col_blocks = (n + 63) / 64;
  if (static_cast<size_t>(col_blocks) * 8 > 64 * 1024) return ADV_EINVAL;  // removed[] must fit 64 KiB of LDS: n <= 524288
  if (hipMemsetAsync(workspace, 0, static_cast<size_t>(n) * col_blocks * sizeof(uint64_t), st) != hipSuccess) return ADV_ELAUNCH;
  hipLaunchKernelGGL(nms_mask, dim3(col_blocks, col_blocks), dim3(64), 0, st, boxes, n, thresh,
                     reinterpret_cast<unsigned long long*>(workspace), col_blocks);
  hipLaunchKernelGGL(nms_scan, dim3(1), dim3(64), static_cast<size_t>(col_blocks) * 8, st,
                     reinterpret_cast<const unsigned long long*>(workspace), n, col_blocks, reinterpret_cast<long long*>(keep_out),
                     reinterpret_cast<int*>(num_keep_out));
  return finish();
}

}  // extern "C"
