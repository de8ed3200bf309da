// K7: plane-sweep cost-volume build (forward + adjoint) for gfx950.
//
// Pure data movement: the forward writes B*2C*D*H*W floats (368 MB per DSGN sample) from two small
// feature maps, the backward reads as many and reduces over the D depth planes.  Both are HBM-bound;
// the design goal is that HBM sees each cost-volume byte exactly once, in whole aligned 16-byte
// lane accesses, and the feature maps once:
//   * a workgroup owns kRows consecutive feature rows of one (sample, channel): of one depth plane in the
//     forward (long contiguous write runs), of ALL planes in the backward (the D-sum stays in registers);
//   * the per-plane disparity shift s makes the right-eye access unaligned (x - s).  The rows are
//     staged in LDS next to a zero apron; a lane then fetches the two ALIGNED float4 that straddle
//     its shifted position (conflict-free ds_read_b128) and picks the 4 floats with a select on
//     s mod 4, which is uniform over the wave - no unaligned or byte-granular memory access at all;
//   * the backward keeps the D-sum in registers (one lane owns 4 x-positions of one row for all
//     planes, added in plane order), so there are no atomics and the float32 result is reproducible.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

#pragma clang fp contract(off)

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRows = 4;

__device__ __forceinline__ v4f zero4() { return (v4f){0.0f, 0.0f, 0.0f, 0.0f}; }

// 4 consecutive floats starting `rem` floats into the aligned pair (a, b); rem is wave-uniform
__device__ __forceinline__ v4f funnel(v4f a, v4f b, int rem) {
  switch (rem) {
    case 0: return a;
    case 1: return (v4f){a.y, a.z, a.w, b.x};
    case 2: return (v4f){a.z, a.w, b.x, b.y};
    default: return (v4f){a.w, b.x, b.y, b.z};
  }
}

// ---------------------------------------------------------------------------------------------------
// forward.  grid = (ceil(H/kRows), D, B*C): one workgroup per (row tile, depth plane, sample*channel).
// Every workgroup writes two contiguous runs of kRows*W floats and consecutive workgroups write
// consecutive runs of the same (c,d) slice; the feature rows are re-read once per plane, which the
// L2 / Infinity Cache absorbs (7.7 MB of features per DSGN sample).  Measured against the
// "one workgroup owns all planes of its rows" decomposition (gpurun_out/psv_sweep.log, round 1):
// 0.71 / 0.81 / 0.75 of HBM peak at B = 1 / 4 / 16 versus 0.64 / 0.67 / 0.70; non-temporal stores
// are worth +30 % here (0.54 without).
// LDS per row: [ W/4 zero float4 | W/4 float4 of the right-eye row | 1 spare ], so that index
// (W/4 + x4 - ceil(s/4) ...) never leaves the buffer for 0 <= s <= W.
// ---------------------------------------------------------------------------------------------------
// One depth plane's disparity.  Integer form: shift s.  Interpolating form (LERP): a float shift sf, split into
// s0 = floor(sf) and w1 = sf - s0; the right-eye value at x - sf is w0 * R[x - s0] + w1 * R[x - s0 - 1] with w0 = 1 - w1
// (R zero-extended), and both halves are zero for x < sc = ceil(sf).  With w1 == 0 it degenerates to the integer form.
struct PlaneShift {
  int s0, sc;
  float w0, w1;
};

template <bool LERP>
__device__ __forceinline__ PlaneShift plane_shift(const void* shift, int idx, int w) {
  PlaneShift p;
  if (LERP) {
    float sf = static_cast<const float*>(shift)[idx];
    sf = sf >= 0.0f ? sf : 0.0f;  // also maps NaN to 0
    sf = sf > static_cast<float>(w) ? static_cast<float>(w) : sf;
    const float fl = floorf(sf);
    p.s0 = static_cast<int>(fl);
    p.w1 = sf - fl;
    p.w0 = 1.0f - p.w1;
    p.sc = p.w1 > 0.0f ? p.s0 + 1 : p.s0;
  } else {
    int s = static_cast<const int32_t*>(shift)[idx];
    s = s < 0 ? 0 : (s > w ? w : s);
    p.s0 = p.sc = s;
    p.w0 = 1.0f;
    p.w1 = 0.0f;
  }
  return p;
}

__device__ __forceinline__ v4f mask_below(v4f v, int x, int sc) {
  if (x < sc) v.x = 0.0f;
  if (x + 1 < sc) v.y = 0.0f;
  if (x + 2 < sc) v.z = 0.0f;
  if (x + 3 < sc) v.w = 0.0f;
  return v;
}

template <bool NT, bool LERP>
__global__ void psv_fwd_plane(const v4f* __restrict__ left, const v4f* __restrict__ right, const void* __restrict__ shift,
                              v4f* __restrict__ cost, int C, int D, int H, int w4) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  v4f* lds = reinterpret_cast<v4f*>(lds_raw);
  const int row_pitch = 2 * w4 + 1;
  const int b = blockIdx.z / C;
  const int c = blockIdx.z - b * C;
  const int d = blockIdx.y;
  const int y0 = blockIdx.x * kRows;
  const int r = threadIdx.x / w4;
  const int x4 = threadIdx.x - r * w4;
  const bool active = r < kRows && (y0 + r) < H;
  const long long plane = static_cast<long long>(H) * w4;
  v4f lv = zero4();
  if (active) {
    const long long src = (static_cast<long long>(b) * C + c) * plane + static_cast<long long>(y0 + r) * w4 + x4;
    lv = left[src];
    lds[r * row_pitch + x4] = zero4();
    lds[r * row_pitch + w4 + x4] = right[src];
    if (x4 == 0) lds[r * row_pitch + 2 * w4] = zero4();
  }
  __syncthreads();
  if (!active) return;
  const PlaneShift ps = plane_shift<LERP>(shift, b * D + d, 4 * w4);
  const int x = x4 * 4;
  lv = mask_below(lv, x, ps.sc);
  const int p = 4 * w4 + x - ps.s0;
  v4f ro = funnel(lds[r * row_pitch + (p >> 2)], lds[r * row_pitch + (p >> 2) + 1], p & 3);
  if (LERP) {  // the neighbour one pixel further left, from the same LDS row: no extra HBM traffic
    const int p1 = p - 1;
    const v4f r1 = p1 >= 0 ? funnel(lds[r * row_pitch + (p1 >> 2)], lds[r * row_pitch + (p1 >> 2) + 1], p1 & 3) : zero4();
    ro = mask_below(ps.w0 * ro + ps.w1 * r1, x, ps.sc);
  }
  const long long out_l = ((static_cast<long long>(b) * 2 * C + c) * D + d) * plane + static_cast<long long>(y0 + r) * w4 + x4;
  const long long out_r = out_l + static_cast<long long>(C) * D * plane;
  if (NT) {
    __builtin_nontemporal_store(lv, cost + out_l);
    __builtin_nontemporal_store(ro, cost + out_r);
  } else {
    cost[out_l] = lv;
    cost[out_r] = ro;
  }
}

// ---------------------------------------------------------------------------------------------------
// backward.  Same ownership.  Left half: aligned loads, masked, summed in registers.  Right half: the
// plane's rows go through a 2-deep LDS ring ([row | zero apron]) so that the shifted read x + s is again
// two aligned float4 + a uniform select; one barrier per plane.
// ---------------------------------------------------------------------------------------------------
template <bool NT, bool LERP>
__global__ void psv_bwd_vec4(const v4f* __restrict__ gcost, const void* __restrict__ shift, v4f* __restrict__ gleft,
                             v4f* __restrict__ gright, int C, int D, int H, int w4) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  v4f* lds = reinterpret_cast<v4f*>(lds_raw);
  const int row_pitch = 2 * w4 + 1;
  const int buf_pitch = kRows * row_pitch;
  const int b = blockIdx.z;
  const int c = blockIdx.y;
  const int y0 = blockIdx.x * kRows;
  const int r = threadIdx.x / w4;
  const int x4 = threadIdx.x - r * w4;
  const bool active = r < kRows && (y0 + r) < H;
  const long long plane = static_cast<long long>(H) * w4;
  const long long in_l = ((static_cast<long long>(b) * 2 * C + c) * D) * plane + static_cast<long long>(y0 + r) * w4 + x4;
  const long long in_r = in_l + static_cast<long long>(C) * D * plane;
  const int x = x4 * 4;

  if (active) {  // zero aprons of both ring buffers, once
    lds[r * row_pitch + w4 + x4] = zero4();
    lds[buf_pitch + r * row_pitch + w4 + x4] = zero4();
    if (x4 == 0) {
      lds[r * row_pitch + 2 * w4] = zero4();
      lds[buf_pitch + r * row_pitch + 2 * w4] = zero4();
    }
  }
  v4f accl = zero4(), accr = zero4();
  v4f gl_next = zero4(), gr_next = zero4();
  if (active && D > 0) {
    gl_next = NT ? __builtin_nontemporal_load(gcost + in_l) : gcost[in_l];
    gr_next = NT ? __builtin_nontemporal_load(gcost + in_r) : gcost[in_r];
  }
  for (int d = 0; d < D; ++d) {
    const v4f gl = gl_next, gr = gr_next;
    v4f* buf = lds + (d & 1) * buf_pitch;
    const PlaneShift ps = plane_shift<LERP>(shift, b * D + d, 4 * w4);
    // (LERP: the forward zeroes its output below sc, so those gradient elements do not flow back)
    if (active) buf[r * row_pitch + x4] = LERP ? mask_below(gr, x, ps.sc) : gr;
    if (active && d + 1 < D) {  // prefetch the next plane while this one is exchanged through LDS
      gl_next = NT ? __builtin_nontemporal_load(gcost + in_l + static_cast<long long>(d + 1) * plane) : gcost[in_l + static_cast<long long>(d + 1) * plane];
      gr_next = NT ? __builtin_nontemporal_load(gcost + in_r + static_cast<long long>(d + 1) * plane) : gcost[in_r + static_cast<long long>(d + 1) * plane];
    }
    __syncthreads();  // plane d is in buf; the other buffer (plane d-1) is free to be overwritten next trip
    if (active) {
      const v4f m = mask_below(gl, x, ps.sc);
      accl = (d == 0) ? m : accl + m;
      const int p = x + ps.s0;  // floats [x + s0, x + s0 + 4) of the zero-extended plane row
      v4f t = funnel(buf[r * row_pitch + (p >> 2)], buf[r * row_pitch + (p >> 2) + 1], p & 3);
      if (LERP) {
        const int p1 = p + 1;
        const v4f t1 = funnel(buf[r * row_pitch + (p1 >> 2)], buf[r * row_pitch + (p1 >> 2) + 1], p1 & 3);
        t = ps.w0 * t + ps.w1 * t1;
      }
      accr = (d == 0) ? t : accr + t;
    }
  }
  if (active) {
    const long long dst = (static_cast<long long>(b) * C + c) * plane + static_cast<long long>(y0 + r) * w4 + x4;
    gleft[dst] = accl;
    gright[dst] = accr;
  }
}

// ---------------------------------------------------------------------------------------------------
// general widths (W % 4 != 0 or unaligned pointers): one lane per element, no LDS
// ---------------------------------------------------------------------------------------------------
template <bool LERP>
__global__ void psv_fwd_scalar(const float* __restrict__ left, const float* __restrict__ right, const void* __restrict__ shift,
                               float* __restrict__ cost, int B, int C, int D, int H, int W) {
  const long long total = static_cast<long long>(B) * C * D * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int x = static_cast<int>(i % W);
    long long t = i / W;
    const int y = static_cast<int>(t % H);
    t /= H;
    const int d = static_cast<int>(t % D);
    t /= D;
    const int c = static_cast<int>(t % C);
    const int b = static_cast<int>(t / C);
    const PlaneShift ps = plane_shift<LERP>(shift, b * D + d, W);
    const long long src = ((static_cast<long long>(b) * C + c) * H + y) * W;
    const long long dl = (((static_cast<long long>(b) * 2 * C + c) * D + d) * H + y) * W + x;
    const long long dr = dl + static_cast<long long>(C) * D * H * W;
    const bool in = x >= ps.sc;
    cost[dl] = in ? left[src + x] : 0.0f;
    float rv = in ? right[src + x - ps.s0] : 0.0f;
    if (LERP && in) {
      const float r1 = x - ps.s0 - 1 >= 0 ? right[src + x - ps.s0 - 1] : 0.0f;
      rv = ps.w0 * rv + ps.w1 * r1;
    }
    cost[dr] = rv;
  }
}

template <bool LERP>
__global__ void psv_bwd_scalar(const float* __restrict__ gcost, const void* __restrict__ shift, float* __restrict__ gleft,
                               float* __restrict__ gright, int B, int C, int D, int H, int W) {
  const long long total = static_cast<long long>(B) * C * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int x = static_cast<int>(i % W);
    long long t = i / W;
    const int y = static_cast<int>(t % H);
    t /= H;
    const int c = static_cast<int>(t % C);
    const int b = static_cast<int>(t / C);
    float al = 0.0f, ar = 0.0f;
    for (int d = 0; d < D; ++d) {
      const PlaneShift ps = plane_shift<LERP>(shift, b * D + d, W);
      const long long gl = (((static_cast<long long>(b) * 2 * C + c) * D + d) * H + y) * W;
      const long long gr = gl + static_cast<long long>(C) * D * H * W;
      const float vl = x >= ps.sc ? gcost[gl + x] : 0.0f;
      const int xo = x + ps.s0;
      float vr = (xo < W && xo >= ps.sc) ? gcost[gr + xo] : 0.0f;
      if (LERP) {
        const float v1 = (xo + 1 < W && xo + 1 >= ps.sc) ? gcost[gr + xo + 1] : 0.0f;
        vr = ps.w0 * vr + ps.w1 * v1;
      }
      al = d == 0 ? vl : al + vl;
      ar = d == 0 ? vr : ar + vr;
    }
    gleft[i] = al;
    gright[i] = ar;
  }
}

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

inline int finish() { return adv_internal_finish_launch(); }

inline int check(const void* a, const void* b, const void* c, const void* d, int B, int C, int D, int H, int W) {
  if (!a || !b || !c || !d) return ADV_EINVAL;
  if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1 || B > 65535 || C > 65535) return ADV_EINVAL;
  if (static_cast<long long>(H) * W > (1LL << 30)) return ADV_EINVAL;
  if (!aligned(a, 4) || !aligned(b, 4) || !aligned(c, 4) || !aligned(d, 4)) return ADV_EALIGN;
  return ADV_OK;
}

template <bool LERP>
int launch_psv_fwd(const float* left, const float* right, const void* shift, float* cost, int b, int c, int d, int h, int w, hipStream_t st) {
  const int w4 = w / 4;
  const int threads = ((kRows * w4 + 63) / 64) * 64;
  const size_t lds = static_cast<size_t>(kRows) * (2 * w4 + 1) * sizeof(v4f);
  const bool vec = (w % 4 == 0) && threads <= 1024 && lds <= 65536 && aligned(left, 16) && aligned(right, 16) && aligned(cost, 16);
  if (vec) {
    if (static_cast<long long>(b) * c > 65535 || d > 65535) return ADV_EINVAL;
    hipLaunchKernelGGL((psv_fwd_plane<true, LERP>), dim3((h + kRows - 1) / kRows, d, b * c), dim3(threads), lds, st,
                       reinterpret_cast<const v4f*>(left), reinterpret_cast<const v4f*>(right), shift, reinterpret_cast<v4f*>(cost), c, d, h, w4);
  } else {
    const long long total = static_cast<long long>(b) * c * d * h * w;
    long long blocks = (total + 255) / 256;
    if (blocks > 1 << 20) blocks = 1 << 20;
    hipLaunchKernelGGL((psv_fwd_scalar<LERP>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, left, right, shift, cost, b, c, d, h, w);
  }
  return finish();
}

template <bool LERP>
int launch_psv_bwd(const float* grad_cost, const void* shift, float* grad_left, float* grad_right, int b, int c, int d, int h, int w,
                   hipStream_t st) {
  const int w4 = w / 4;
  const int threads = ((kRows * w4 + 63) / 64) * 64;
  const size_t lds = 2 * static_cast<size_t>(kRows) * (2 * w4 + 1) * sizeof(v4f);
  const bool vec = (w % 4 == 0) && threads <= 1024 && lds <= 65536 && aligned(grad_cost, 16) && aligned(grad_left, 16) && aligned(grad_right, 16);
  if (vec) {
    hipLaunchKernelGGL((psv_bwd_vec4<true, LERP>), dim3((h + kRows - 1) / kRows, c, b), dim3(threads), lds, st,
                       reinterpret_cast<const v4f*>(grad_cost), shift, reinterpret_cast<v4f*>(grad_left), reinterpret_cast<v4f*>(grad_right), c, d, h, w4);
  } else {
    const long long total = static_cast<long long>(b) * c * h * w;
    long long blocks = (total + 255) / 256;
    if (blocks > 1 << 20) blocks = 1 << 20;
    hipLaunchKernelGGL((psv_bwd_scalar<LERP>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, grad_cost, shift, grad_left, grad_right, b, c, d, h, w);
  }
  return finish();
}

}  // namespace

extern "C" {

int adv_psv_build_f32(const float* left, const float* right, const int32_t* shift, float* cost, int b, int c, int d, int h,
                      int w, adv_stream_t stream) {
  const int rc = check(left, right, shift, cost, b, c, d, h, w);
  if (rc != ADV_OK) return rc;
  return launch_psv_fwd<false>(left, right, shift, cost, b, c, d, h, w, static_cast<hipStream_t>(stream));
}

int adv_psv_build_bwd_f32(const float* grad_cost, const int32_t* shift, float* grad_left, float* grad_right, int b, int c,
                          int d, int h, int w, adv_stream_t stream) {
  const int rc = check(grad_cost, shift, grad_left, grad_right, b, c, d, h, w);
  if (rc != ADV_OK) return rc;
  return launch_psv_bwd<false>(grad_cost, shift, grad_left, grad_right, b, c, d, h, w, static_cast<hipStream_t>(stream));
}

int adv_psv_build_lerp_f32(const float* left, const float* right, const float* shift, float* cost, int b, int c, int d, int h,
                           int w, adv_stream_t stream) {
  const int rc = check(left, right, shift, cost, b, c, d, h, w);
  if (rc != ADV_OK) return rc;
  return launch_psv_fwd<true>(left, right, shift, cost, b, c, d, h, w, static_cast<hipStream_t>(stream));
}

int adv_psv_build_lerp_bwd_f32(const float* grad_cost, const float* shift, float* grad_left, float* grad_right, int b, int c,
                               int d, int h, int w, adv_stream_t stream) {
  const int rc = check(grad_cost, shift, grad_left, grad_right, b, c, d, h, w);
  if (rc != ADV_OK) return rc;
  return launch_psv_bwd<true>(grad_cost, shift, grad_left, grad_right, b, c, d, h, w, static_cast<hipStream_t>(stream));
}

}  // extern "C"
