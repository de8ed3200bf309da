// Bilinear up-sampling of a feature pyramid level (align_corners = False) and its adjoint - the `_upsample_add` of the FPN top-down path
// (attack/Stereo-RCNN/stereo_rcnn.py:92-108: F.upsample(x, size=(H, W), mode='bilinear') + y).
//
// Why own kernels for an element-wise operator: torch's BACKWARD of it scatters every output gradient into its (up to) four source
// pixels with atomicAdd - the sum a source pixel receives depends on the order the hardware happens to serve the atomics in, so the
// attack gradient (and with it, now and then, a sign, a pixel of the PNG and a box index) differed from run to run.  The adjoint here is
// a GATHER: one thread per source pixel walks the output pixels that read it in a fixed order (rows ascending, columns ascending) and
// adds their contributions one after the other.  By bytes both are streaming passes over the large map (153 MB at the pyramid's largest
// level); as measured they are bound by their vector instructions - the taps and weights are recomputed per element - at 85 us forward
// (torch 174) and 186 us backward (torch's scatter 178) for that level, 0.4 ms of a 52 ms step in all: left there.
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

// Work decomposition of both kernels: a workgroup is 4 rows x 64 columns of the map it WRITES (blockIdx.x: four rows of the nc * rows
// row list, blockIdx.y: the 64-column chunk) - no per-element division (the first version decoded a flat 64-bit index with two 64-bit
// divisions per element and was bound by those: 120 us for a pass that moves 190 MB).
constexpr int kRowsPerThread = 4;     // rows per thread of the forward: the column taps are computed once for four outputs (98 -> 85 us at the largest level)

__global__ __launch_bounds__(kBlock) void bilinear_up_fwd(const float* __restrict__ x, float* __restrict__ out, int h, int w, int ho, int wo, float sy,
                                                           float sx, long long rows) {
  const long long row0 = (static_cast<long long>(blockIdx.x) * 4 + threadIdx.y) * kRowsPerThread;
  const int ox = blockIdx.y * 64 + threadIdx.x;
  if (row0 >= rows || ox >= wo) return;
  const Src xs = source_of(ox, w, sx);
  float v[kRowsPerThread][4], l0[kRowsPerThread], l1[kRowsPerThread];
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {             // all sixteen loads first (a row past the end repeats the last one and is not stored)
    const long long row = row0 + r < rows ? row0 + r : rows - 1;
    const long long nc = row / ho;
    const Src ys = source_of(static_cast<int>(row - nc * ho), h, sy);
    const float* p0 = x + (nc * h + ys.i0) * w;
    const float* p1 = x + (nc * h + ys.i1) * w;
    v[r][0] = p0[xs.i0], v[r][1] = p0[xs.i1], v[r][2] = p1[xs.i0], v[r][3] = p1[xs.i1];
    l0[r] = ys.l0, l1[r] = ys.l1;
  }
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    const float top = xs.l0 * v[r][0] + xs.l1 * v[r][1];
    const float bot = xs.l0 * v[r][2] + xs.l1 * v[r][3];
    if (row0 + r < rows) __builtin_nontemporal_store(l0[r] * top + l1[r] * bot, out + (row0 + r) * wo + ox);
  }
}

// The candidate range trimmed to the outputs that really read input index i (the source coordinate is monotonic in o, so they are
// contiguous; empty when no output reads i - possible only when down-sampling)
__device__ __forceinline__ void readers(int i, int n_in, int n_out, float scale, int& lo, int& hi) {
  candidates(i, n_out, scale, lo, hi);
  while (lo <= hi && weight_of(lo, i, n_in, scale) == 0.0f) ++lo;
  while (hi >= lo && weight_of(hi, i, n_in, scale) == 0.0f) --hi;
}

// KX column weights are kept in registers (an up-sampling by at most 2 has at most 4 readers per axis, KX = 4; other ratios KX = 8 and
// what lies beyond finishes in the generic loop - any size pair is legal).  No branch around a load: a load inside a divergent branch is
// awaited before the next one is issued, and a thread's 16 dependent round trips to HBM made the first version 3x slower than the
// scattering kernel it replaces.  Slots beyond the range load a valid address and add a selected 0.0f (acc + 0.0f == acc bit for bit:
// acc starts at +0 and a sum is -0 only when both terms are, so it never is) - what the oracle skips stays skipped even for inf / nan.
template <int KX>
__global__ __launch_bounds__(kBlock) void bilinear_up_bwd(const float* __restrict__ g, float* __restrict__ gin, int h, int w, int ho, int wo, float sy,
                                                           float sx, long long rows) {
  constexpr int KY = KX;      // reader rows held in registers, as the columns
  const long long row = static_cast<long long>(blockIdx.x) * 4 + threadIdx.y;
  const int ix = blockIdx.y * 64 + threadIdx.x;
  if (row >= rows || ix >= w) return;
  const long long nc = row / h;
  const int iy = static_cast<int>(row - nc * h);
  int ylo, yhi, xlo, xhi;
  readers(iy, h, ho, sy, ylo, yhi);
  readers(ix, w, wo, sx, xlo, xhi);
  float wx[KX], wy[KY];
  int xc[KX], yc[KY];
#pragma unroll
  for (int k = 0; k < KX; ++k) {
    wx[k] = xlo + k <= xhi ? weight_of(xlo + k, ix, w, sx) : 0.0f;
    xc[k] = xlo + k <= xhi ? xlo + k : (xlo <= xhi ? xlo : 0);
  }
#pragma unroll
  for (int r = 0; r < KY; ++r) {
    wy[r] = ylo + r <= yhi ? weight_of(ylo + r, iy, h, sy) : 0.0f;
    yc[r] = ylo + r <= yhi ? ylo + r : (ylo <= yhi ? ylo : 0);
  }
  const float* gp = g + nc * ho * wo;
  float v[KY][KX];
#pragma unroll
  for (int r = 0; r < KY; ++r)
#pragma unroll
    for (int k = 0; k < KX; ++k) v[r][k] = gp[static_cast<long long>(yc[r]) * wo + xc[k]];
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < KY; ++r) {
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      const float term = (wy[r] * wx[k]) * v[r][k];
      acc = acc + ((wy[r] != 0.0f && wx[k] != 0.0f) ? term : 0.0f);
    }
    for (int ox = xlo + KX; ox <= xhi; ++ox) {             // (a ragged ratio's extra reader column: rare)
      const float wv = weight_of(ox, ix, w, sx);
      if (ylo + r <= yhi && wy[r] != 0.0f && wv != 0.0f) acc = acc + (wy[r] * wv) * gp[static_cast<long long>(ylo + r) * wo + ox];
    }
  }
  for (int oy = ylo + KY; oy <= yhi; ++oy) {               // reader rows beyond the registers: large ratios
    const float wyv = weight_of(oy, iy, h, sy);
    const float* rowp = gp + static_cast<long long>(oy) * wo;
    for (int ox = xlo; ox <= xhi; ++ox) {
      const float wv = weight_of(ox, ix, w, sx);
      if (wyv != 0.0f && wv != 0.0f) acc = acc + (wyv * wv) * rowp[ox];
    }
  }
  gin[row * w + ix] = acc;
}

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

bool grid_for(long long rows, int width, dim3& grid, int rows_per_thread) {
  const long long gx = (rows + 4 * rows_per_thread - 1) / (4 * rows_per_thread);
  if (gx > 0x7fffffffLL) return false;
  grid = dim3(static_cast<unsigned>(gx), static_cast<unsigned>((width + 63) / 64));
  return true;
}

}  // namespace

extern "C" {

int adv_bilinear_up_f32(const float* x, float* out, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream) {
  if (!x || !out || x == out || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1 || wo > 64 * 65535) return ADV_EINVAL;
  if (!aligned4(x) || !aligned4(out)) return ADV_EALIGN;
  dim3 grid;
  if (!grid_for(static_cast<long long>(nc) * ho, wo, grid, kRowsPerThread)) return ADV_EINVAL;
  hipLaunchKernelGGL(bilinear_up_fwd, grid, dim3(64, 4), 0, static_cast<hipStream_t>(stream), x, out, h, w, ho, wo,
                     static_cast<float>(h) / static_cast<float>(ho), static_cast<float>(w) / static_cast<float>(wo), static_cast<long long>(nc) * ho);
  return adv_internal_finish_launch();
}

int adv_bilinear_up_bwd_f32(const float* grad_out, float* grad_in, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream) {
  if (!grad_out || !grad_in || grad_out == grad_in || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1 || w > 64 * 65535) return ADV_EINVAL;
  if (!aligned4(grad_out) || !aligned4(grad_in)) return ADV_EALIGN;
  dim3 grid;
  if (!grid_for(static_cast<long long>(nc) * h, w, grid, 1)) return ADV_EINVAL;
  const float sy = static_cast<float>(h) / static_cast<float>(ho), sx = static_cast<float>(w) / static_cast<float>(wo);
  const long long rows = static_cast<long long>(nc) * h;
  if (wo <= 2 * w && ho <= 2 * h)        // at most four readers per axis (more - a ragged ratio's fifth - are taken by the generic loops)
    hipLaunchKernelGGL(bilinear_up_bwd<4>, grid, dim3(64, 4), 0, static_cast<hipStream_t>(stream), grad_out, grad_in, h, w, ho, wo, sy, sx, rows);
  else
    hipLaunchKernelGGL(bilinear_up_bwd<8>, grid, dim3(64, 4), 0, static_cast<hipStream_t>(stream), grad_out, grad_in, h, w, ho, wo, sy, sx, rows);
  return adv_internal_finish_launch();
}

}  // extern "C"
