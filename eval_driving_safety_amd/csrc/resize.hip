// Bilinear up-sampling of a feature pyramid level (align_corners = False) and its adjoint - the `_upsample_add` of the FPN top-down path
// (attack/Stereo-RCNN/stereo_rcnn.py:92-108: F.upsample(x, size=(H, W), mode='bilinear') + y).
//
// Why own kernels for an element-wise operator: torch's BACKWARD of it scatters every output gradient into its (up to) four source
// pixels with atomicAdd - the sum a source pixel receives depends on the order the hardware happens to serve the atomics in, so the
// attack gradient (and with it, now and then, a sign, a pixel of the PNG and a box index) differed from run to run.  The adjoint here is
// a GATHER: one thread per source pixel walks the output pixels that read it in a fixed order (rows ascending, columns ascending) and
// adds their contributions one after the other.  Both kernels are HBM-bound streaming passes (the forward writes, the backward reads the
// large map exactly once; the small map lives in L2).
//
//   forward   src = max(fma(scale, o + 0.5, -0.5), 0), scale = float(in) / out;  i0 = int(src), i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1
//             out[oy][ox] = l0y * (l0x * x[y0][x0] + l1x * x[y0][x1]) + l1y * (l0x * x[y1][x0] + l1x * x[y1][x1])
//   backward  gin[iy][ix] = sum over oy ascending, ox ascending of (wy(oy, iy) * wx(ox, ix)) * g[oy][ox],
//             w(o, i) = (i0(o) == i ? l0(o) : 0) + (i1(o) == i ? l1(o) : 0); terms with a zero weight are skipped
// (oracle/oracle_np.py bilinear_up / bilinear_up_bwd restate both, operation for operation; -ffp-contract=off: no fused multiply-add but the explicit one of the source coordinate).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

namespace {

constexpr int kBlock = 256;

struct Src {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Src source_of(int o, int n_in, float scale) {
  float s = fmaf(scale, static_cast<float>(o) + 0.5f, -0.5f);      // ONE rounding, as the contracted expression of torch's GPU kernel
  s = s < 0.0f ? 0.0f : s;
  Src r;
  r.i0 = static_cast<int>(s);
  if (r.i0 > n_in - 1) r.i0 = n_in - 1;              // (cannot happen for out >= in; keeps every index inside the map for any size pair)
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  r.l1 = s - static_cast<float>(r.i0);
  r.l0 = 1.0f - r.l1;
  return r;
}

__device__ __forceinline__ float weight_of(int o, int i, int n_in, float scale) {
  const Src s = source_of(o, n_in, scale);
  return (s.i0 == i ? s.l0 : 0.0f) + (s.i1 == i ? s.l1 : 0.0f);
}

// the output indices whose source interval can touch input index i: src(o) in (i - 1, i + 1), two guard indices on either side
// (candidates outside the true range have weight 0 and are skipped - the range only bounds the walk)
__device__ __forceinline__ void candidates(int i, int n_out, float scale, int& lo, int& hi) {
  const float a = (static_cast<float>(i) - 0.5f) / scale - 0.5f, b = (static_cast<float>(i) + 1.5f) / scale - 0.5f;
  lo = static_cast<int>(floorf(a)) - 2;
  hi = static_cast<int>(ceilf(b)) + 2;
  lo = lo < 0 ? 0 : lo;
  hi = hi > n_out - 1 ? n_out - 1 : hi;
  if (i == 0) lo = 0;                                  // every output whose source coordinate was clamped to 0 reads index 0
}

__global__ __launch_bounds__(kBlock) void bilinear_up_fwd(const float* __restrict__ x, float* __restrict__ out, int h, int w, int ho, int wo, float sy,
                                                           float sx, long long total) {
  for (long long idx = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; idx < total; idx += static_cast<long long>(gridDim.x) * kBlock) {
    const int ox = static_cast<int>(idx % wo), oy = static_cast<int>((idx / wo) % ho);
    const long long nc = idx / (static_cast<long long>(wo) * ho);
    const Src ys = source_of(oy, h, sy), xs = source_of(ox, w, sx);
    const float* p = x + nc * h * w;
    const float top = xs.l0 * p[static_cast<long long>(ys.i0) * w + xs.i0] + xs.l1 * p[static_cast<long long>(ys.i0) * w + xs.i1];
    const float bot = xs.l0 * p[static_cast<long long>(ys.i1) * w + xs.i0] + xs.l1 * p[static_cast<long long>(ys.i1) * w + xs.i1];
    __builtin_nontemporal_store(ys.l0 * top + ys.l1 * bot, out + idx);
  }
}

// KX column weights are kept in registers (an up-sampling by 2 has at most 4 + guards candidates per axis); wider ranges - any size
// pair is legal - finish in the generic loop
template <int KX>
__global__ __launch_bounds__(kBlock) void bilinear_up_bwd(const float* __restrict__ g, float* __restrict__ gin, int h, int w, int ho, int wo, float sy,
                                                           float sx, long long total) {
  for (long long idx = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; idx < total; idx += static_cast<long long>(gridDim.x) * kBlock) {
    const int ix = static_cast<int>(idx % w), iy = static_cast<int>((idx / w) % h);
    const long long nc = idx / (static_cast<long long>(w) * h);
    int ylo, yhi, xlo, xhi;
    candidates(iy, ho, sy, ylo, yhi);
    candidates(ix, wo, sx, xlo, xhi);
    float wx[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) wx[k] = xlo + k <= xhi ? weight_of(xlo + k, ix, w, sx) : 0.0f;
    const float* gp = g + nc * ho * wo;
    float acc = 0.0f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = weight_of(oy, iy, h, sy);
      if (wy == 0.0f) continue;
      const float* row = gp + static_cast<long long>(oy) * wo;
#pragma unroll
      for (int k = 0; k < KX; ++k) {
        if (wx[k] != 0.0f) acc = acc + (wy * wx[k]) * __builtin_nontemporal_load(row + xlo + k);
      }
      for (int ox = xlo + KX; ox <= xhi; ++ox) {
        const float wv = weight_of(ox, ix, w, sx);
        if (wv != 0.0f) acc = acc + (wy * wv) * row[ox];
      }
    }
    gin[idx] = acc;
  }
}

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

unsigned blocks_for(long long total) {
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 32) blocks = 65535LL * 32;
  return static_cast<unsigned>(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" {

int adv_bilinear_up_f32(const float* x, float* out, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream) {
  if (!x || !out || x == out || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1) return ADV_EINVAL;
  if (!aligned4(x) || !aligned4(out)) return ADV_EALIGN;
  const long long total = static_cast<long long>(nc) * ho * wo;
  hipLaunchKernelGGL(bilinear_up_fwd, dim3(blocks_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, out, h, w, ho, wo,
                     static_cast<float>(h) / static_cast<float>(ho), static_cast<float>(w) / static_cast<float>(wo), total);
  return adv_internal_finish_launch();
}

int adv_bilinear_up_bwd_f32(const float* grad_out, float* grad_in, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream) {
  if (!grad_out || !grad_in || grad_out == grad_in || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1) return ADV_EINVAL;
  if (!aligned4(grad_out) || !aligned4(grad_in)) return ADV_EALIGN;
  const long long total = static_cast<long long>(nc) * h * w;
  hipLaunchKernelGGL(bilinear_up_bwd<8>, dim3(blocks_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), grad_out, grad_in, h, w, ho, wo,
                     static_cast<float>(h) / static_cast<float>(ho), static_cast<float>(w) / static_cast<float>(wo), total);
  return adv_internal_finish_launch();
}

}  // extern "C"
