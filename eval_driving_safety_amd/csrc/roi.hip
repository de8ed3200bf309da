// Stereo R-CNN RoI-path natives for gfx950: RoIAlign forward/backward and greedy NMS.
// Small, latency-bound kernels (a few hundred RoIs x 256 channels x 7x7 / 14x14 bins); the design points are
// coalescing along the bin index (forward), a gather-shaped, atomic-free and therefore bit-reproducible backward,
// and wave64-wide suppression masks for NMS.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdint>
#include <cstdlib>

#include "adv_internal.h"
#include "advengine.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;
typedef float v4f __attribute__((ext_vector_type(4)));

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

// one lane per output element (r, c, ph, pw); the bin index is fastest, so a wave reads neighbouring samples.
// The kernel is bound by the texture addresser's gather rate (16 single-dword gathers per output, each wave-load touching a
// dozen rows), not by arithmetic or HBM - two restructurings that remove the redundant tap arithmetic were measured and
// dropped: (a) one workgroup per (roi, 32 channels) with the roi's window and a tap table in LDS: bit-identical, 0.25 ms
// against 0.10 ms for 512 rois x 256 channels x 7x7 bins (windows of ~30 x 30 pixels: staging costs more than the L1/L2 hits
// it saves); (b) one lane per (roi, bin) with the taps in registers, streaming over 32 channels: 0.112 ms against 0.098
// (profiles/r02_roi_fwd_lds_attempt.jsonl, r02_roi_fwd_regs_attempt.jsonl).  PAIR: the two taps of a row are neighbours
// (x_high = x_low + 1, or the same pixel at the right edge) - one dword-aligned 8-byte load instead of two gathers.
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));

template <bool PAIR>
__global__ __launch_bounds__(kBlock) void roi_align_fwd(const float* __restrict__ feat, const float* __restrict__ rois,
                                                        float* __restrict__ out, int C, int H, int W, long long total, int PH,
                                                        int PW, float scale, int sampling_ratio) {
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int pw = static_cast<int>(i % PW);
    const int ph = static_cast<int>((i / PW) % PH);
    const int c = static_cast<int>((i / (static_cast<long long>(PW) * PH)) % C);
    const long long r = i / (static_cast<long long>(PW) * PH * C);
    const Bin b = bin_of(rois + r * 5, scale, PH, PW, sampling_ratio);
    if (b.batch < 0) continue;         // <round 4> a roi with a negative batch index is skipped: its output rows stay as they are
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
          float v1, v2, v3, v4;
          if (PAIR) {  // W >= 2 (host-checked): columns xs, xs + 1 cover x_low and x_high and never leave the row
            const int xs = t.x_low < W - 1 ? t.x_low : W - 2;
            const f32x2_u lo = *reinterpret_cast<const f32x2_u*>(plane + t.y_low * W + xs);
            const f32x2_u hi = *reinterpret_cast<const f32x2_u*>(plane + t.y_high * W + xs);
            v1 = t.x_low == xs ? lo.x : lo.y, v2 = t.x_high == xs ? lo.x : lo.y;
            v3 = t.x_low == xs ? hi.x : hi.y, v4 = t.x_high == xs ? hi.x : hi.y;
          } else {
            v1 = plane[t.y_low * W + t.x_low], v2 = plane[t.y_low * W + t.x_high];
            v3 = plane[t.y_high * W + t.x_low], v4 = plane[t.y_high * W + t.x_high];
          }
          v = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
        }
        acc += v;
      }
    }
    out[i] = acc / count;
  }
}

// ---- RoIAlign backward as a deterministic GATHER (no float atomics).
// The textbook backward scatters every sample's four weighted contributions with atomicAdd: the float32 sum order then
// depends on the schedule, and the gradient of the Stereo R-CNN attack is not reproducible from run to run.  Here every
// lane OWNS one feature pixel (for a block of kChanBlock channels, accumulators in registers) and collects, in one fixed
// order - roi index, then sample row, then sample column, then tap 1..4 - the contributions of exactly those samples
// whose bilinear taps touch its pixel.  Two things make that affordable:
//   * a first kernel writes, per 8 x 32-pixel tile, the ascending list of rois whose sample window can reach the tile
//     (wave ballot + prefix popcount keep the list ordered), so a lane only visits the handful of rois around it;
//   * bilinear weights are separable (w1 = hy*hx ...): per roi a lane classifies its few candidate sample rows and
//     columns (which of its two taps hits the pixel, with which weight) per axis, and the double loop only multiplies.
// The candidate ranges are a two-sample-wide superset of the samples that can touch the pixel; every candidate is tested
// with the forward's own tap arithmetic (taps_at), so membership is exact whatever the float rounding of the range.
constexpr int kTileY = 8, kTileX = 32;  // 256 lanes: one pixel each, x fastest (coalesced stores)
constexpr int kChanBlock = 8;   // most channels a lane carries; fewer on small maps (see roi_align_bwd_gather)

struct Axis1 {  // one sample coordinate against one pixel coordinate
  float w_low, w_high;  // weight when the sample's low / high tap is this pixel (0 otherwise)
  bool hit_low, hit_high, valid;
};

// the per-axis half of taps_at: clamp, low/high index, the two 1-D weights
__device__ __forceinline__ Axis1 axis_taps(int size, float v, int pixel) {
  Axis1 a;
  a.valid = !(v < -1.0f || v > static_cast<float>(size));
  if (v <= 0.0f) v = 0.0f;
  int low = static_cast<int>(v), high;
  if (low >= size - 1) {
    high = low = size - 1;
    v = static_cast<float>(low);
  } else {
    high = low + 1;
  }
  const float l = v - static_cast<float>(low), h = 1.0f - l;
  a.hit_low = low == pixel;
  a.hit_high = high == pixel;
  a.w_low = h;   // "hy" / "hx"
  a.w_high = l;  // "ly" / "lx"
  return a;
}

// window of feature pixels the samples of a roi can touch along one axis (conservative), clipped to the map
__device__ __forceinline__ void roi_window(float start, float extent, int size, int* lo, int* hi) {
  const float a = fmaxf(start, 0.0f) - 1.0f, b = start + extent + 1.0f;
  *lo = max(0, static_cast<int>(floorf(a)));
  *hi = min(size - 1, static_cast<int>(ceilf(b)));
}

__global__ __launch_bounds__(64) void roi_tile_lists(const float* __restrict__ rois, int R, int H, int W, int tiles_y, int tiles_x, int PH,
                                                     int PW, float scale, int sampling_ratio, int* __restrict__ lists, int G, int seg_len) {
  // one wave per (tile, image, roi segment): lists[..][0] = count, lists[..][1..] = the segment's roi indices ascending
  const int tile = blockIdx.x, img = blockIdx.y / G, seg = blockIdx.y % G;
  const int ty = (tile / tiles_x) * kTileY, tx = (tile % tiles_x) * kTileX;
  int* out = lists + ((static_cast<long long>(img) * G + seg) * tiles_y * tiles_x + tile) * (R + 1);
  int count = 0;
  const int r_end = min(R, (seg + 1) * seg_len);
  for (int r0 = seg * seg_len; r0 < r_end; r0 += 64) {
    const int r = r0 + threadIdx.x;
    bool hit = false;
    if (r < r_end) {
      const Bin b = bin_of(rois + static_cast<long long>(r) * 5, scale, PH, PW, sampling_ratio);
      if (b.batch == img) {
        int y_lo, y_hi, x_lo, x_hi;
        roi_window(b.start_h, b.bin_h * static_cast<float>(PH), H, &y_lo, &y_hi);
        roi_window(b.start_w, b.bin_w * static_cast<float>(PW), W, &x_lo, &x_hi);
        hit = y_lo < ty + kTileY && y_hi >= ty && x_lo < tx + kTileX && x_hi >= tx;
      }
    }
    const unsigned long long m = __ballot(hit);
    if (hit) out[1 + count + __popcll(m & ((1ULL << threadIdx.x) - 1ULL))] = r;
    count += __popcll(m);
  }
  if (threadIdx.x == 0) out[0] = count;
}

// candidate sample indices k (over PH*grid bins x samples) whose coordinate start + (k + 0.5) * step can touch `pixel`
__device__ __forceinline__ void cand_range(float start, float step, int n_samples, int pixel, int size, int* k_lo, int* k_hi) {
  float lo = static_cast<float>(pixel) - 1.0f, hi = static_cast<float>(pixel) + 1.0f;
  if (pixel == 0) lo = -1.0f;                                 // samples in [-1, 0] clamp onto pixel 0
  if (pixel == size - 1) hi = static_cast<float>(size);        // samples in [size-1, size] clamp onto the last pixel
  const float a = (lo - start) / step - 0.5f, b = (hi - start) / step - 0.5f;
  *k_lo = max(0, static_cast<int>(floorf(a)) - 2);
  *k_hi = min(n_samples - 1, static_cast<int>(ceilf(b)) + 2);
}

// v / count, the forward's "/ count" of a sample's weight.  The IEEE division costs ~12 instructions and sat 32 times in the innermost
// loop (4 taps x 8 channels per sample): 5 of the 6.5 ms of a P2 call with 510 clustered rois.  For a count that is a power of two
// (sampling grids 1x1, 1x2, 2x2, 2x4 ... - the common ones) multiplying by the exactly representable 1 / count gives the same
// correctly rounded result, subnormals included; any other count keeps the division.  Uniform per roi.
__device__ __forceinline__ float over_count(float v, float count, float inv, bool pow2) { return pow2 ? v * inv : v / count; }

// one roi's contributions to this lane's pixel, classified by the lane itself (the round-2 formulation: every lane tests its
// candidate rows x candidate columns) - the path of a lane whose row or column list does not fit the shared LDS lists
__host__ __device__ __forceinline__ int cpad(int c) { return (c + 7) & ~7; }     // channels of the channel-last copy: zero padded to 8

template <int CB>
__device__ __forceinline__ void roi_direct(float (&acc)[CB], const Bin& b, const float* __restrict__ gout, int r, int C, int c0, int H, int W, int PH,
                                           int PW, int py, int px) {
  const int cnt = b.grid_h * b.grid_w;
  const float count = static_cast<float>(cnt), inv = 1.0f / count;
  const bool pow2 = (cnt & (cnt - 1)) == 0;
  int ky_lo, ky_hi, kx_lo, kx_hi;
  cand_range(b.start_h, b.bin_h / static_cast<float>(b.grid_h), PH * b.grid_h, py, H, &ky_lo, &ky_hi);
  cand_range(b.start_w, b.bin_w / static_cast<float>(b.grid_w), PW * b.grid_w, px, W, &kx_lo, &kx_hi);
  for (int ky = ky_lo; ky <= ky_hi; ++ky) {
    const int ph = ky / b.grid_h, iy = ky - ph * b.grid_h;
    const float y = b.start_h + static_cast<float>(ph) * b.bin_h + (static_cast<float>(iy) + 0.5f) * b.bin_h / static_cast<float>(b.grid_h);
    const Axis1 ay = axis_taps(H, y, py);
    if (!ay.valid || !(ay.hit_low || ay.hit_high)) continue;
    for (int kx = kx_lo; kx <= kx_hi; ++kx) {
      const int pw = kx / b.grid_w, ix = kx - pw * b.grid_w;
      const float x = b.start_w + static_cast<float>(pw) * b.bin_w + (static_cast<float>(ix) + 0.5f) * b.bin_w / static_cast<float>(b.grid_w);
      const Axis1 ax = axis_taps(W, x, px);
      if (!ax.valid || !(ax.hit_low || ax.hit_high)) continue;
      // taps 1..4 of this sample = (y_low,x_low) (y_low,x_high) (y_high,x_low) (y_high,x_high), in that order
      const float w1 = ay.w_low * ax.w_low, w2 = ay.w_low * ax.w_high, w3 = ay.w_high * ax.w_low, w4 = ay.w_high * ax.w_high;
      const float* g = gout + ((static_cast<long long>(r) * PH + ph) * PW + pw) * cpad(C) + c0;      // channel-last copy of grad_out
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        if (c0 + c < C) {
          const float gv = g[c];
          if (ay.hit_low && ax.hit_low) acc[c] = acc[c] + over_count(gv * w1, count, inv, pow2);
          if (ay.hit_low && ax.hit_high) acc[c] = acc[c] + over_count(gv * w2, count, inv, pow2);
          if (ay.hit_high && ax.hit_low) acc[c] = acc[c] + over_count(gv * w3, count, inv, pow2);
          if (ay.hit_high && ax.hit_high) acc[c] = acc[c] + over_count(gv * w4, count, inv, pow2);
        }
      }
    }
  }
}

// <round 3> The classification is SEPARABLE and depends on the pixel's row or column only - 8 + 32 distinct problems per roi for
// the 256 lanes of a tile, which round 2's kernel solved 256 times each (and the column one again for every candidate row).  Now
// a batch of kRoiBatch rois is classified COOPERATIVELY: lane (roi k of the batch, axis pixel a) walks that pixel's candidate
// samples once and leaves, in LDS, the ascending list of the samples that touch it (bin index, which tap, both 1-D weights);
// after one barrier every lane runs its row list x its column list - the same contributions in the same order (roi, sample
// row, sample column, tap 1..4) with the same float operations, so the bits are those of the direct formulation and of the
// oracle.  Lists hold kListCap entries (a sample step is bin / ceil(bin) in (0.5, 1] pixels: 2-4 samples touch a pixel per axis);
// a longer one (rois smaller than their pooled grid) sends that lane to roi_direct for that roi.  CB channels per lane: 8 on
// large maps, fewer on the small pyramid levels, where 6-20 tiles x C/8 channel blocks left most of the chip idle behind a few
// hundred-roi lists (profiles/r03_r101_kernel_stats.csv: 9.9 ms per call before, 69 of the 157 ms of the R101-shaped step).
constexpr int kRoiBatch = 6;   // 6 x (8 rows + 32 columns) = 240 classifying lanes
constexpr int kListCap = 16;  // taps on one pixel per axis: 2 * pooled / roi_size + 1 - 16 covers a 14-bin grid on rois from 2 pixels up (8 sent the narrow proposals to roi_direct: 10x slower)

struct AxisList {
  int n;                       // tap hits on the pixel (may exceed kListCap: then the entries are not used)
  int bin[kListCap];           // ph / pw of the sample the tap belongs to
  float w[kListCap];           // the tap's 1-D weight: hy / hx for a sample's low tap, ly / lx for its high tap
};
// One entry per TAP that lands on the pixel, in the order (sample ascending; low tap before high tap).  A sample whose low AND high
// tap are the same pixel (clamped at the map's last row / column, where the high tap's weight is exactly 0) gives two entries.  A
// pixel's contributions are then (row entry) x (column entry), one multiply-add each and no conditions: the same products
// hy*hx, hy*lx, ly*hx, ly*lx as the forward's four taps, summed in the order (roi, row entry, column entry) - which is the order
// (roi, sample row, sample column, tap 1..4) except that the ZERO terms of a clamped sample come later; adding an exact zero
// at another position of the chain does not change a float32 sum, so the oracle's ordered sum is reproduced bit for bit.

template <int CB>
__device__ __forceinline__ void roi_bwd_body(AxisList (*s_list)[kTileY + kTileX], Bin* s_bin, int c0, const float* __restrict__ gout,
                                             const float* __restrict__ rois, const int* __restrict__ lists, float* __restrict__ gfeat, int C, int H, int W,
                                             int R, int tiles_y, int tiles_x, int PH, int PW, float scale, int sampling_ratio, int dbg) {
  const int tile = blockIdx.x, img = blockIdx.z;
  const int tid = static_cast<int>(threadIdx.x);
  const int ty0 = (tile / tiles_x) * kTileY, tx0 = (tile % tiles_x) * kTileX;
  const int ly = tid / kTileX, lx = tid % kTileX;
  const int py = ty0 + ly, px = tx0 + lx;
  const bool inside = py < H && px < W;
  const int* list = lists + (static_cast<long long>(img) * tiles_y * tiles_x + tile) * (R + 1);
  const int n_list = list[0];
  float acc[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) acc[c] = 0.0f;
  // classifying role of this lane: roi `cb` of the batch, axis pixel `ca` (0..7 = the tile's rows, 8..39 = its columns)
  const int cb = tid / (kTileY + kTileX), ca = tid % (kTileY + kTileX);
  for (int l0 = 0; l0 < n_list; l0 += kRoiBatch) {
    const int nb = min(kRoiBatch, n_list - l0);
    if (cb < nb && !(dbg & 2)) {
      const int r = list[1 + l0 + cb];
      const Bin b = bin_of(rois + static_cast<long long>(r) * 5, scale, PH, PW, sampling_ratio);
      if (ca == 0) s_bin[cb] = b;
      const bool is_y = ca < kTileY;
      const int pixel = is_y ? ty0 + ca : tx0 + (ca - kTileY), size = is_y ? H : W;
      const float start = is_y ? b.start_h : b.start_w, bin = is_y ? b.bin_h : b.bin_w;
      const int grid = is_y ? b.grid_h : b.grid_w, pooled = is_y ? PH : PW;
      AxisList& out = s_list[cb][ca];
      int n = 0;
      if (pixel < size) {
        int k_lo, k_hi;
        cand_range(start, bin / static_cast<float>(grid), pooled * grid, pixel, size, &k_lo, &k_hi);
        for (int k = k_lo; k <= k_hi; ++k) {
          const int pb = k / grid, ik = k - pb * grid;
          const float v = start + static_cast<float>(pb) * bin + (static_cast<float>(ik) + 0.5f) * bin / static_cast<float>(grid);
          const Axis1 a = axis_taps(size, v, pixel);
          if (!a.valid || !(a.hit_low || a.hit_high)) continue;
          if (a.hit_low) {
            if (n < kListCap) out.bin[n] = pb, out.w[n] = a.w_low;
            ++n;
          }
          if (a.hit_high) {
            if (n < kListCap) out.bin[n] = pb, out.w[n] = a.w_high;
            ++n;
          }
        }
      }
      out.n = n;
    }
    __syncthreads();
    if (inside && !(dbg & 1)) {
      for (int k = 0; k < nb; ++k) {
        const AxisList& yl = s_list[k][ly];
        const AxisList& xl = s_list[k][kTileY + lx];
        const int ny = yl.n, nx = xl.n;
        if (ny == 0 || nx == 0) continue;
        const int r = list[1 + l0 + k];
        const Bin& b = s_bin[k];
        if (ny > kListCap || nx > kListCap) {
          roi_direct<CB>(acc, b, gout, r, C, c0, H, W, PH, PW, py, px);
          continue;
        }
        const int cnt = b.grid_h * b.grid_w;
        const float count = static_cast<float>(cnt), inv = 1.0f / count;
        const bool pow2 = (cnt & (cnt - 1)) == 0;
        // The samples of this roi that touch the pixel, row-major (sample row, then sample column): FOUR at a time - their gathers are
        // issued together and only then consumed in order.  One sample per trip left every trip waiting a full L2 round trip for its
        // own gathers (one wave per SIMD on a hot tile, nothing else to run): 20 k cycles per roi on a tile that 500 clustered rois reach.
        const int ns = ny * nx;
        const int cp = cpad(C);
        const float* gbase = gout + static_cast<long long>(r) * PH * PW * cp + c0;      // channel-last copy: a lane's CB channels are ONE load
        // two copies of the loop, chosen per roi (uniform): with `v * inv` only, or with the IEEE division only - left as a select the
        // compiler evaluated BOTH for each of the 32 contributions of a sample (12 instructions of division each)
        auto samples = [&](auto pow2_c) {
        constexpr bool kPow2 = decltype(pow2_c)::value;
        for (int s0 = 0; s0 < ns; s0 += 4) {
          float gv[4][CB], wgt[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int sidx = min(s0 + u, ns - 1);
            const int iy = sidx / nx, ix = sidx - iy * nx;
            wgt[u] = yl.w[iy] * xl.w[ix];
            const float* g = gbase + (static_cast<long long>(yl.bin[iy]) * PW + xl.bin[ix]) * cp;
            if constexpr (CB % 4 == 0) {
#pragma unroll
              for (int q = 0; q < CB / 4; ++q) {
                const float4 v4 = *reinterpret_cast<const float4*>(g + 4 * q);
                gv[u][4 * q] = v4.x, gv[u][4 * q + 1] = v4.y, gv[u][4 * q + 2] = v4.z, gv[u][4 * q + 3] = v4.w;
              }
            } else {
#pragma unroll
              for (int c = 0; c < CB; ++c) gv[u][c] = g[c];      // (the padding channels hold zeros)
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (s0 + u < ns) {
#pragma unroll
              for (int c = 0; c < CB; ++c)
                if (c0 + c < C) acc[c] = acc[c] + over_count(gv[u][c] * wgt[u], count, inv, kPow2);
            }
          }
        }
        };
        if (pow2) samples(std::true_type{});
        else samples(std::false_type{});
      }
    }
    __syncthreads();
  }
  if (inside) {
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (c0 + c < C) gfeat[((static_cast<long long>(img) * C + c0 + c) * H + py) * W + px] = acc[c];
  }
}

// grad_out [R][C][S = PH*PW] -> channel-last [R][S][C'] (C' = C padded to 8 with zeros): the gather reads a lane's block of channels of
// one bin as one 16 / 32-byte load instead of CB loads S floats apart.  One workgroup per (roi, 32 channels): 32 x S contiguous floats in,
// S rows of 32 floats out.
__global__ __launch_bounds__(kBlock) void roi_gout_channel_last(const float* __restrict__ gout, float* __restrict__ out, int C, int S) {
  extern __shared__ float s_t[];                 // [32][S + 1]
  const int r = blockIdx.x, c0 = blockIdx.y * 32, cp = cpad(C);
  const int nc = min(32, C - c0);
  const float* src = gout + (static_cast<long long>(r) * C + c0) * S;
  for (int i = threadIdx.x; i < nc * S; i += kBlock) s_t[(i / S) * (S + 1) + i % S] = src[i];
  __syncthreads();
  float* dst = out + static_cast<long long>(r) * S * cp + c0;
  const int ncp = min(32, cp - c0);              // the zero padding belongs to the last block
  for (int i = threadIdx.x; i < S * 32; i += kBlock) {
    const int sidx = i / 32, c = i % 32;
    if (c < ncp) dst[static_cast<long long>(sidx) * cp + c] = c < nc ? s_t[c * (S + 1) + sidx] : 0.0f;
  }
}

template <int CB>
__global__ __launch_bounds__(kBlock) void roi_align_bwd_gather(const float* __restrict__ gout, const float* __restrict__ rois,
                                                               const int* __restrict__ lists, float* __restrict__ gfeat, int C, int H, int W,
                                                               int R, int tiles_y, int tiles_x, int PH, int PW, float scale,
                                                               int sampling_ratio, int dbg) {
  __shared__ AxisList s_list[kRoiBatch][kTileY + kTileX];
  __shared__ Bin s_bin[kRoiBatch];
  roi_bwd_body<CB>(s_list, s_bin, static_cast<int>(blockIdx.y) * CB, gout, rois, lists, gfeat, C, H, W, R, tiles_y, tiles_x, PH, PW, scale,
                   sampling_ratio, dbg);
}

// <round 3, second formulation> The gather above keeps a pixel's accumulators in the registers of the lane that owns the pixel: for the
// narrow proposals a detector really produces (4 x 13 pixels at the pyramid level) a roi touches ~10 of a wave's 64 pixels and the
// wave runs the whole sample loop at 15 % lane utilisation, once per block of 8 channels (and classifies the roi again for each of
// the 32 channel blocks).  Here the tile's accumulators live in LDS, [256 pixels][32 channels], and the lanes are handed WORK ITEMS
// = (pixel the roi touches, channel): a half-wave owns one pixel, its 32 lanes the 32 channels - every lane busy, the 32 channels of
// a bin one 128-byte load from the channel-last copy of grad_out, the classification once per 32 channels.  Rois are taken in list
// order with a barrier between them, an item adds its roi's samples to the pixel in (row entry, column entry) order: the same float
// operations in the same order as the register formulation and the oracle.
constexpr int kLdsBatch = 4;
constexpr int kStageBins = 196;     // grad_out blocks of up to 14 x 14 bins are staged in LDS (two buffers); larger grids gather from global memory

// STAGE: the roi's block of grad_out - [PH*PW bins][32 channels] of the channel-last copy, 6 KB for 7 x 7 bins, 25 KB for 14 x 14 - is
// copied into LDS by all 256 lanes at once (one round of coalesced 16-byte loads) while the PREVIOUS roi's items are being summed, and
// the sample loops read LDS.  Gathering straight from global memory left each half-wave waiting a full L2 round trip per four
// samples with only 8 pixels in flight per workgroup: 25-35 us per roi on the tile's serial chain (profiles/r03_roi_bwd.jsonl).
// <round 4> V4 (with STAGE): an item is (touched pixel, FOUR channels) - one 16-byte LDS read per sample instead of four, the sample's
// weight product and its bin address computed once for the four: the counters put this kernel at 2.8e8 vector instructions per call on
// 512 proposals, 0 matrix work, issue-bound (profiles/r04_conv_pmc.json) - the per-channel float operations and their order are unchanged.
template <bool STAGE, int kAccChan, bool V4 = false>     // kAccChan channels per workgroup: a roi's items are (touched pixel, channel), 256 / kAccChan pixels at a time
__global__ __launch_bounds__(kBlock) void roi_align_bwd_lds(const float* __restrict__ gcl, const float* __restrict__ rois,
                                                            const int* __restrict__ lists, float* __restrict__ gfeat, int C, int H, int W, int R,
                                                            int tiles_y, int tiles_x, int PH, int PW, float scale, int sampling_ratio, int dbg,
                                                            int G, long long seg_elems) {
  __shared__ AxisList s_list[kLdsBatch][kTileY + kTileX];
  __shared__ Bin s_bin[kLdsBatch];
  __shared__ int s_rect[kLdsBatch][4];            // first touched row, rows, first touched column, columns (tile-local)
  static_assert(!V4 || (STAGE && kAccChan % 4 == 0), "the four-channel items read the staged block");
  constexpr int kAccStride = V4 ? kAccChan + 4 : kAccChan + 1;      // (V4: 16-byte aligned rows)
  constexpr int kLanesPerPix = V4 ? kAccChan / 4 : kAccChan;
  __shared__ __attribute__((aligned(16))) float s_acc[kTileY * kTileX * kAccStride];
  extern __shared__ __attribute__((aligned(16))) float s_g[];     // STAGE: [2][PH*PW][kAccChan]
  // <round 4> G > 1: the rois are split into G segments of consecutive indices and blockIdx.z = image * G + segment sums ITS segment's
  // rois into its own copy of the map (gfeat + segment * seg_elems); roi_bwd_sum_segments adds the copies in segment order.  A hot tile's
  // serial chain (hundreds of rois on the few tiles where the proposals cluster, the rest of the chip idle behind them) is cut G-fold.
  const int tile = blockIdx.x, img = static_cast<int>(blockIdx.z) / G, seg = static_cast<int>(blockIdx.z) % G, c0 = blockIdx.y * kAccChan;
  const int tid = static_cast<int>(threadIdx.x);
  const int ty0 = (tile / tiles_x) * kTileY, tx0 = (tile % tiles_x) * kTileX;
  const int* list = lists + ((static_cast<long long>(img) * G + seg) * tiles_y * tiles_x + tile) * (R + 1);
  const int n_list = list[0];
  if (G > 1 && n_list == 0) return;               // this segment does not reach the tile: its copy is never read there (roi_bwd_sum_segments)
  gfeat += static_cast<long long>(seg) * seg_elems;
  for (int i = tid; i < kTileY * kTileX * kAccStride; i += kBlock) s_acc[i] = 0.0f;
  const int cb = tid / (kTileY + kTileX), ca = tid % (kTileY + kTileX);
  const int cp = cpad(C);
  const int lane_c = V4 ? 4 * (tid % kLanesPerPix) : tid % kAccChan;     // this lane's (first) channel of an item
  const bool chan_ok = V4 ? true : c0 + lane_c < C;   // (V4: channels past C are zeros of the staged block - they add nothing and are not written)
  const int bins = PH * PW, nf4 = bins * (kAccChan / 4);
  constexpr int kStageSlots = (kStageBins * (kAccChan / 4) + kBlock - 1) / kBlock;   // float4 per lane of a staged block (7)
  float4 stage[kStageSlots];
  // (c0 + 32 <= cp always: cp is a multiple of 8 and the block's channels beyond C are zero padding or, past cp, never loaded)
  auto stage_load = [&](int r) {
#pragma unroll
    for (int i = 0; i < kStageSlots; ++i) {
      const int f = tid + kBlock * i, bin = f / (kAccChan / 4), q = f % (kAccChan / 4);
      stage[i] = (f < nf4 && c0 + 4 * q < cp) ? *reinterpret_cast<const float4*>(gcl + (static_cast<long long>(r) * bins + bin) * cp + c0 + 4 * q)
                                               : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kStageSlots; ++i) {
      const int f = tid + kBlock * i;
      if (f < nf4) *reinterpret_cast<float4*>(s_g + buf * bins * kAccChan + 4 * f) = stage[i];
    }
  };
  for (int l0 = 0; l0 < n_list; l0 += kLdsBatch) {
    const int nb = min(kLdsBatch, n_list - l0);
    __syncthreads();                              // the previous batch's lists are no longer read (and the zeroing above is done)
    if (STAGE) stage_load(list[1 + l0]);          // the batch's first roi: its block travels while the lanes classify
    if (cb < nb) {                                // classification: as in roi_bwd_body
      const int r = list[1 + l0 + cb];
      const Bin b = bin_of(rois + static_cast<long long>(r) * 5, scale, PH, PW, sampling_ratio);
      if (ca == 0) s_bin[cb] = b;
      const bool is_y = ca < kTileY;
      const int pixel = is_y ? ty0 + ca : tx0 + (ca - kTileY), size = is_y ? H : W;
      const float start = is_y ? b.start_h : b.start_w, bin = is_y ? b.bin_h : b.bin_w;
      const int grid = is_y ? b.grid_h : b.grid_w, pooled = is_y ? PH : PW;
      AxisList& out = s_list[cb][ca];
      int n = 0;
      if (pixel < size && !(dbg & 2)) {
        int k_lo, k_hi;
        cand_range(start, bin / static_cast<float>(grid), pooled * grid, pixel, size, &k_lo, &k_hi);
        for (int k = k_lo; k <= k_hi; ++k) {
          const int pb = k / grid, ik = k - pb * grid;
          const float v = start + static_cast<float>(pb) * bin + (static_cast<float>(ik) + 0.5f) * bin / static_cast<float>(grid);
          const Axis1 a = axis_taps(size, v, pixel);
          if (!a.valid || !(a.hit_low || a.hit_high)) continue;
          if (a.hit_low) {
            if (n < kListCap) out.bin[n] = pb, out.w[n] = a.w_low;
            ++n;
          }
          if (a.hit_high) {
            if (n < kListCap) out.bin[n] = pb, out.w[n] = a.w_high;
            ++n;
          }
        }
      }
      out.n = n;
    }
    if (STAGE) stage_store(0);
    __syncthreads();
    if (tid < nb) {                               // the rectangle of tile pixels roi `tid` of the batch touches
      int y_lo = kTileY, y_hi = -1, x_lo = kTileX, x_hi = -1;
      for (int i = 0; i < kTileY; ++i)
        if (s_list[tid][i].n > 0) y_lo = min(y_lo, i), y_hi = i;
      for (int i = 0; i < kTileX; ++i)
        if (s_list[tid][kTileY + i].n > 0) x_lo = min(x_lo, i), x_hi = i;
      s_rect[tid][0] = y_lo, s_rect[tid][1] = y_hi - y_lo + 1, s_rect[tid][2] = x_lo, s_rect[tid][3] = x_hi - x_lo + 1;
    }
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
      const int rows = s_rect[k][1], cols = s_rect[k][3];
      if (STAGE && k + 1 < nb) stage_load(list[1 + l0 + k + 1]);       // the next roi's block, into registers
      if (rows > 0 && cols > 0 && !(dbg & 1)) {
        const int y_lo = s_rect[k][0], x_lo = s_rect[k][2];
        const int r = list[1 + l0 + k];
        const Bin& b = s_bin[k];
        const int cnt = b.grid_h * b.grid_w;
        const float count = static_cast<float>(cnt), inv = 1.0f / count;
        const bool pow2 = (cnt & (cnt - 1)) == 0;
        const float* gbase = gcl + static_cast<long long>(r) * bins * cp + c0 + lane_c;
        const float* sbase = s_g + (k & 1) * bins * kAccChan + lane_c;
        for (int p = tid / kLanesPerPix; p < rows * cols; p += kBlock / kLanesPerPix) {   // 256 / kLanesPerPix pixels at a time
          const int ly = y_lo + p / cols, lx = x_lo + p % cols;
          const AxisList& yl = s_list[k][ly];
          const AxisList& xl = s_list[k][kTileY + lx];
          const int ny = yl.n, nx = xl.n;
          if (ny == 0 || nx == 0 || !chan_ok) continue;
          float* slot = s_acc + (ly * kTileX + lx) * kAccStride + lane_c;
          if constexpr (V4) {
            v4f acc4 = *reinterpret_cast<v4f*>(slot);
            if (ny > kListCap || nx > kListCap) {                         // more taps than the lists hold: classify per sample
              float four[4] = {acc4.x, acc4.y, acc4.z, acc4.w};
              roi_direct<4>(four, b, gcl, r, C, c0 + lane_c, H, W, PH, PW, ty0 + ly, tx0 + lx);
              *reinterpret_cast<v4f*>(slot) = v4f{four[0], four[1], four[2], four[3]};
              continue;
            }
            if (!(dbg & 4)) {
              auto run4 = [&](auto pow2_c) {
                constexpr bool kPow2 = decltype(pow2_c)::value;
                int xb[kListCap];
                float xw[kListCap];
#pragma unroll
                for (int i = 0; i < kListCap; ++i) {
                  xb[i] = i < nx ? xl.bin[i] * kAccChan : 0;
                  xw[i] = i < nx ? xl.w[i] : 0.0f;
                }
                for (int iy = 0; iy < ny; ++iy) {
                  const float wy = yl.w[iy];
                  const float* row = sbase + yl.bin[iy] * PW * kAccChan;
#pragma unroll
                  for (int i0 = 0; i0 < kListCap; i0 += 4) {
                    if (i0 >= nx) break;
                    v4f gv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) gv[u] = *reinterpret_cast<const v4f*>(row + xb[i0 + u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                      if (i0 + u < nx) {
                        const float wgt = wy * xw[i0 + u];
                        acc4.x = acc4.x + over_count(gv[u].x * wgt, count, inv, kPow2);
                        acc4.y = acc4.y + over_count(gv[u].y * wgt, count, inv, kPow2);
                        acc4.z = acc4.z + over_count(gv[u].z * wgt, count, inv, kPow2);
                        acc4.w = acc4.w + over_count(gv[u].w * wgt, count, inv, kPow2);
                      }
                  }
                }
              };
              if (pow2) run4(std::true_type{});
              else run4(std::false_type{});
            }
            *reinterpret_cast<v4f*>(slot) = acc4;
            continue;
          }
          float acc = *slot;
          if (ny > kListCap || nx > kListCap) {                           // more taps than the lists hold: classify per sample
            float one[1] = {acc};
            roi_direct<1>(one, b, gcl, r, C, c0 + lane_c, H, W, PH, PW, ty0 + ly, tx0 + lx);
            *slot = one[0];
            continue;
          }
          const int ns = ny * nx;
          auto samples = [&](auto pow2_c) {
            constexpr bool kPow2 = decltype(pow2_c)::value;
            if (STAGE) {                                                  // LDS is close: plain nested loops, no index arithmetic
              // the column entries once per item, in registers (every row of samples reuses them); then four LDS reads in flight per
              // step and four ordered additions - a plain loop made every sample wait for two dependent LDS round trips
              int xb[kListCap];
              float xw[kListCap];
#pragma unroll
              for (int i = 0; i < kListCap; ++i) {
                xb[i] = i < nx ? xl.bin[i] * kAccChan : 0;
                xw[i] = i < nx ? xl.w[i] : 0.0f;
              }
              for (int iy = 0; iy < ny; ++iy) {
                const float wy = yl.w[iy];
                const float* row = sbase + yl.bin[iy] * PW * kAccChan;
#pragma unroll
                for (int i0 = 0; i0 < kListCap; i0 += 4) {
                  if (i0 >= nx) break;
                  float gv[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) gv[u] = row[xb[i0 + u]];
#pragma unroll
                  for (int u = 0; u < 4; ++u)
                    if (i0 + u < nx) acc = acc + over_count(gv[u] * (wy * xw[i0 + u]), count, inv, kPow2);
                }
              }
              return;
            }
            for (int s0 = 0; s0 < ns; s0 += 4) {                          // four gathers in flight, consumed in order
              float gv[4], wgt[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int sidx = min(s0 + u, ns - 1);
                const int iy = sidx / nx, ix = sidx - iy * nx;
                wgt[u] = yl.w[iy] * xl.w[ix];
                const int bin = yl.bin[iy] * PW + xl.bin[ix];
                gv[u] = STAGE ? sbase[bin * kAccChan] : gbase[static_cast<long long>(bin) * cp];
              }
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (s0 + u < ns) acc = acc + over_count(gv[u] * wgt[u], count, inv, kPow2);
            }
          };
          if (dbg & 4) {
          } else if (pow2) samples(std::true_type{});
          else samples(std::false_type{});
          *slot = acc;
        }
      }
      if (STAGE && k + 1 < nb) stage_store((k + 1) & 1);              // (buffer (k + 1) & 1 was last read by roi k - 1: a barrier ago)
      __syncthreads();                            // the next roi may touch the same pixels through other lanes
    }
  }
  __syncthreads();
  const int py = ty0 + tid / kTileX, px = tx0 + tid % kTileX;
  if (py < H && px < W) {
    for (int c = 0; c < kAccChan && c0 + c < C; ++c)
      gfeat[((static_cast<long long>(img) * C + c0 + c) * H + py) * W + px] = s_acc[tid * kAccStride + c];
  }
}

// gfeat[b,c,y,x] = P_0 + P_1 + ... + P_{G-1} in that order, P_s = segment s's copy where the segment reaches the pixel's tile (its list is
// not empty) and an exact 0 elsewhere - which leaves a float32 sum unchanged, so the copies of empty (tile, segment) pairs are neither
// written nor read.  One lane per element, x fastest.
__global__ __launch_bounds__(kBlock) void roi_bwd_sum_segments(const float* __restrict__ parts, const int* __restrict__ lists, float* __restrict__ gfeat,
                                                               int C, int H, int W, int R, int tiles_y, int tiles_x, int G, long long seg_elems) {
  const long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= seg_elems) return;
  const int x = static_cast<int>(i % W), y = static_cast<int>((i / W) % H);
  const int img = static_cast<int>(i / (static_cast<long long>(W) * H * C));
  const int tile = (y / kTileY) * tiles_x + x / kTileX;
  float acc = 0.0f;
  for (int s = 0; s < G; ++s)
    if (lists[((static_cast<long long>(img) * G + s) * tiles_y * tiles_x + tile) * (R + 1)] > 0) acc = acc + parts[s * seg_elems + i];
  gfeat[i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// <round 5> third formulation: PER-ROI AXIS TABLES + a wave per pixel.
// The LDS formulation above classifies every roi again for each (tile, block of 16 channels) and walks a tile's rois one after the other
// behind two barriers each: on 510 clustered P2 proposals 0.23 ms of classification, 0.36 ms of staging / barriers and 0.66 ms of item and
// sample loops (profiles/r04_roi_bwd_phases.json).  Here
//   * roi_axis_tables classifies ONCE per call: for roi r and every row y / column x of the map, the ascending list of the sample taps that
//     land on that row / column (bin index + 1-D weight: exactly the AxisList entries of the kernels above) - n_tab[r][H + W] counts and
//     e_tab[r][H + W][kListCap] entries, shared by all channels and all pixels;
//   * roi_align_bwd_tab gives a pixel to a WAVE (lanes = 64 channels; grad_out's channel-last rows are 256-byte loads).  The wave scans the
//     tile's roi list 64 rois at a time (lane j: do roi j's row and column lists both hold entries for my pixel? - one ballot), lays the
//     touching rois' ny x nx entries end to end (wave prefix sum), and its lanes fetch 64 entries of that sequence at once - whichever rois
//     they belong to - as (weight product, grad_out offset, sample count); the ordered sum then runs over the 64 entries with the loads
//     eight deep.  No barrier, no LDS accumulators, no serial walk over rois that do not touch the pixel, and a hot pixel's chain is a
//     stream of independent loads rather than a round trip per roi.
// Per pixel and channel the additions are (roi ascending; row entry; column entry) - the order of the formulations above and of the
// oracle's roi_align_bwd_ordered with one segment - with the same float operations (weight product, * 1/count or / count), so the bits
// are the same.  A roi whose list for the pixel overflows kListCap is summed by roi_direct in its place of the sequence.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kTabPix = 8;       // pixels per workgroup (two per wave): a 32-byte run per channel when the tile is written
constexpr int kTabChan = 64;     // channels per workgroup = the lanes of a wave

__global__ __launch_bounds__(kBlock) void roi_axis_tables(const float* __restrict__ rois, int B, int H, int W, int PH, int PW, float scale,
                                                          int sampling_ratio, int* __restrict__ n_tab, int2* __restrict__ e_tab,
                                                          float* __restrict__ cnt) {
  const int r = blockIdx.y, a = blockIdx.x * kBlock + static_cast<int>(threadIdx.x);
  if (a >= H + W) return;
  const Bin b = bin_of(rois + static_cast<long long>(r) * 5, scale, PH, PW, sampling_ratio);
  if (a == 0) cnt[r] = static_cast<float>(b.grid_h * b.grid_w);
  int n = 0;
  if (b.batch >= 0 && b.batch < B) {               // (a skipped roi - batch index -1 - is on no tile's list either)
    const bool is_y = a < H;
    const int pixel = is_y ? a : a - H, size = is_y ? H : W;
    const float start = is_y ? b.start_h : b.start_w, bin = is_y ? b.bin_h : b.bin_w;
    const int grid = is_y ? b.grid_h : b.grid_w, pooled = is_y ? PH : PW;
    int2* out = e_tab + (static_cast<long long>(r) * (H + W) + a) * kListCap;
    int k_lo, k_hi;
    cand_range(start, bin / static_cast<float>(grid), pooled * grid, pixel, size, &k_lo, &k_hi);
    for (int k = k_lo; k <= k_hi; ++k) {           // classification: as in roi_bwd_body
      const int pb = k / grid, ik = k - pb * grid;
      const float v = start + static_cast<float>(pb) * bin + (static_cast<float>(ik) + 0.5f) * bin / static_cast<float>(grid);
      const Axis1 t = axis_taps(size, v, pixel);
      if (!t.valid || !(t.hit_low || t.hit_high)) continue;
      if (t.hit_low) {
        if (n < kListCap) out[n] = int2{pb, __float_as_int(t.w_low)};
        ++n;
      }
      if (t.hit_high) {
        if (n < kListCap) out[n] = int2{pb, __float_as_int(t.w_high)};
        ++n;
      }
    }
  }
  n_tab[static_cast<long long>(r) * (H + W) + a] = n;
}

__device__ __forceinline__ int lane_read(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_read(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int lane_fetch(int v, int lane) { return __builtin_amdgcn_ds_bpermute(lane << 2, v); }

__global__ __launch_bounds__(kBlock) void roi_align_bwd_tab(const float* __restrict__ gcl, const float* __restrict__ rois,
                                                            const int* __restrict__ lists, const int* __restrict__ n_tab,
                                                            const int2* __restrict__ e_tab, const float* __restrict__ cnt,
                                                            float* __restrict__ gfeat, int C, int H, int W, int R, int tiles_y, int tiles_x,
                                                            int groups_x, int PH, int PW, float scale, int sampling_ratio) {
  __shared__ float s_out[kTabChan][kTabPix + 1];
  const int img = blockIdx.z, c0 = static_cast<int>(blockIdx.y) * kTabChan;
  const int gx = static_cast<int>(blockIdx.x) % groups_x, py = static_cast<int>(blockIdx.x) / groups_x;
  const int px0 = gx * kTabPix;
  const int tile = (py / kTileY) * tiles_x + px0 / kTileX;
  const int* list = lists + (static_cast<long long>(img) * tiles_y * tiles_x + tile) * (R + 1);
  const int n_list = list[0];
  const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cp = cpad(C), chan = c0 + lane, HW = H + W, bins = PH * PW;
  const bool chan_ok = chan < cp;                  // (channels C .. cp - 1 of the channel-last copy are zeros; beyond cp nothing is read)
  const float* gl = gcl + (chan_ok ? chan : 0);
#pragma unroll 1
  for (int i = 0; i < kTabPix / 4; ++i) {
    const int lx = wave * (kTabPix / 4) + i, px = px0 + lx;
    float acc = 0.0f;
    if (px < W) {
#pragma unroll 1
      for (int l0 = 0; l0 < n_list; l0 += 64) {
        const int r = l0 + lane < n_list ? list[1 + l0 + lane] : -1;
        int ny = 0, nx = 0;
        if (r >= 0) {
          ny = n_tab[static_cast<long long>(r) * HW + py];
          nx = n_tab[static_cast<long long>(r) * HW + H + px];
        }
        const bool touch = ny > 0 && nx > 0;
        const float count = touch ? cnt[r] : 1.0f;
        unsigned long long m = __ballot(touch), ov = __ballot(touch && (ny > kListCap || nx > kListCap));
        while (m) {
          // the touching rois up to the first one whose lists overflow, as one sequence of entries; then that roi by itself
          const int jo = ov ? __builtin_ctzll(ov) : 64;
          const unsigned long long grp = jo < 64 ? (m & ((1ULL << jo) - 1ULL)) : m;
          if (grp) {
            const int ns = ((grp >> lane) & 1ULL) ? ny * nx : 0;
            int incl = ns;                          // inclusive prefix sum over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
              const int up = __shfl_up(incl, d, 64);
              if (lane >= d) incl += up;
            }
            const int excl = incl - ns, total = lane_read(incl, 63);
#pragma unroll 1
            for (int t0 = 0; t0 < total; t0 += 64) {
              const int nk = min(64, total - t0);
              const int t = t0 + min(lane, nk - 1);
              int j = 0;                            // the lane (= roi of the batch) whose run of entries holds entry t
#pragma unroll
              for (int q = 0; q < 63; ++q) j += lane_read(incl, q) <= t ? 1 : 0;
              const int e = t - lane_fetch(excl, j), nxj = lane_fetch(nx, j), rj = lane_fetch(r, j);
              const float cj = __int_as_float(lane_fetch(__float_as_int(count), j));
              const int iy = e / nxj, ix = e - iy * nxj;
              const long long rb = static_cast<long long>(rj) * HW;
              const int2 ye = e_tab[(rb + py) * kListCap + iy], xe = e_tab[(rb + H + px) * kListCap + ix];
              const float wgt = __int_as_float(ye.y) * __int_as_float(xe.y);
              const int off = (rj * bins + ye.x * PW + xe.x) * cp;
              const bool p2 = (__float_as_int(cj) & 0x007fffff) == 0;       // the count is a power of two: * (1 / count) is the division
              const float inv = 1.0f / cj;
              auto run = [&](auto all_pow2) __attribute__((always_inline)) {
                constexpr bool kAllPow2 = decltype(all_pow2)::value;
#pragma unroll 1
                for (int k = 0; k < nk; k += 8) {
                  float g[8], w[8], c[8], v[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const int kk = min(k + u, nk - 1);
                    w[u] = lane_read(wgt, kk);
                    c[u] = lane_read(cj, kk);
                    v[u] = lane_read(inv, kk);
                    g[u] = gl[lane_read(off, kk)];
                  }
#pragma unroll
                  for (int u = 0; u < 8; ++u)
                    if (k + u < nk) {
                      const float prod = g[u] * w[u];
                      if (kAllPow2) acc = acc + prod * v[u];
                      else acc = acc + (((__float_as_int(c[u]) & 0x007fffff) == 0) ? prod * v[u] : prod / c[u]);
                    }
                }
              };
              if (__ballot(!p2) == 0ULL) run(std::true_type{});
              else run(std::false_type{});
            }
            m &= ~grp;
          }
          if (jo < 64) {
            const int rj = lane_read(r, jo);
            const Bin b = bin_of(rois + static_cast<long long>(rj) * 5, scale, PH, PW, sampling_ratio);
            float one[1] = {acc};
            roi_direct<1>(one, b, gcl, rj, C, chan, H, W, PH, PW, py, px);
            acc = one[0];
            ov &= ov - 1ULL;
            m &= ~(1ULL << jo);
          }
        }
      }
    }
    s_out[lane][lx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < kTabChan * kTabPix / kBlock; ++p) {
    const int ch = tid / kTabPix + p * (kBlock / kTabPix), lx = tid % kTabPix, px = px0 + lx;
    if (px < W && c0 + ch < C) gfeat[((static_cast<long long>(img) * C + c0 + ch) * H + py) * W + px] = s_out[ch][lx];
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

// the mask words start at zero (a kernel rather than hipMemsetAsync: inside a captured hipGraph every node of the suppression is then a
// kernel node)
__global__ __launch_bounds__(256) void nms_zero(unsigned long long* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) p[i] = 0ull;
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

// <round 5> the same greedy scan, 64 boxes at a time.  The scan above pays a barrier and a dependent global load per BOX (2000 proposals:
// 0.42 ms, 1 % of the detector step, profiles/r05_r101_step_profile.json); here a block of 64 boxes is settled among themselves in scalar
// registers - lane l holds row 64k + l's word of the DIAGONAL mask block, and only the boxes still alive are visited - while the other
// waves copy the NEXT block's 64 mask rows (contiguous in memory) into LDS; then every word of removed[] beyond the block takes the OR of
// the kept boxes' rows from LDS.  Two barriers per 64 boxes, no global latency on the chain.  Same kept indices in the same order.
constexpr int kNmsScanThreads = 1024;     // one workgroup: wave 0 settles the diagonal block, the other fifteen copy mask rows (one round trip per block)
__global__ __launch_bounds__(kNmsScanThreads) void nms_scan_blocks(const unsigned long long* __restrict__ mask, int n, int col_blocks, long long* keep,
                                                                   int* num_keep) {
  extern __shared__ unsigned long long s_nms[];      // removed[col_blocks] | rows[3][64 * col_blocks] | alive
  unsigned long long* removed = s_nms;
  unsigned long long* rows = s_nms + col_blocks;
  unsigned long long* s_alive = rows + 3 * 64 * col_blocks;
  const int tid = static_cast<int>(threadIdx.x), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = 64 * col_blocks;
  // rows 64k .. 64k + 63 (zeros beyond n) -> rows[k % 3], TWO blocks ahead of their use (a ring of three): a block takes ~1.5 us, a trip to
  // memory about as long - one block ahead the copy was what every block waited for (3.5 us per block, profiles/r05_r101_kernel_stats.csv)
  auto stage = [&](int k, int first, int step) {
    unsigned long long* dst = rows + (k % 3) * per;
    const unsigned long long* src = mask + static_cast<long long>(k) * per;
    const int have = min(64, n - 64 * k) * col_blocks;
#pragma unroll 6
    for (int i = first; i < per; i += step) dst[i] = i < have ? src[i] : 0ULL;      // (unrolled: the thread's loads in flight together)
  };
  for (int j = tid; j < col_blocks; j += kNmsScanThreads) removed[j] = 0ULL;
  stage(0, tid, kNmsScanThreads);
  if (col_blocks > 1) stage(1, tid, kNmsScanThreads);
  __syncthreads();
  int kept = 0;
  for (int k = 0; k < col_blocks; ++k) {
    const unsigned long long* cur = rows + (k % 3) * per;
    if (wave == 0) {
      const int nb = min(64, n - 64 * k);
      const unsigned long long diag = cur[lane * col_blocks + k];
      const unsigned long long valid = nb == 64 ? ~0ULL : ((1ULL << nb) - 1ULL);
      const unsigned long long rem = removed[k];
      unsigned long long alive = ((static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(~rem >> 32)))) << 32) |
                                  static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(~rem)))) & valid;
      const int dlo = static_cast<int>(diag), dhi = static_cast<int>(diag >> 32);
      unsigned long long todo = alive;
      while (todo) {
        const int l = __builtin_ctzll(todo);
        const unsigned long long d = (static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readlane(dhi, l))) << 32) |
                                     static_cast<unsigned>(__builtin_amdgcn_readlane(dlo, l));
        alive &= ~d;                                   // (row l's bits are those of the later boxes of the block only)
        todo = l == 63 ? 0ULL : (alive & ~((2ULL << l) - 1ULL));
      }
      if ((alive >> lane) & 1ULL) keep[kept + __popcll(alive & ((1ULL << lane) - 1ULL))] = 64LL * k + lane;
      if (lane == 0) *s_alive = alive;
    } else if (k + 2 < col_blocks) {
      stage(k + 2, tid - 64, kNmsScanThreads - 64);
    }
    __syncthreads();
    const unsigned long long alive = *s_alive;
    kept += __popcll(alive);
    // every later word of removed[] takes the OR of the kept boxes' rows: eight threads per word, eight rows each with their LDS reads in
    // flight together, combined by an LDS atomic OR (commutative: the result does not depend on the order) - a thread per word walking the
    // kept rows one dependent read after the other was 2-3 us of the ~4.4 us a block took
    for (int item = tid; item < 8 * (col_blocks - k - 1); item += kNmsScanThreads) {
      const int j = k + 1 + (item >> 3), part = item & 7;
      unsigned long long acc = 0ULL;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int l = 8 * part + u;
        const unsigned long long row = cur[l * col_blocks + j];
        acc |= ((alive >> l) & 1ULL) ? row : 0ULL;
      }
      if (acc) atomicOr(&removed[j], acc);
    }
    __syncthreads();
  }
  if (tid == 0) *num_keep = kept;
}
constexpr int kNmsBlockScanMaxWords = 96;             // col_blocks up to which the block scan's LDS image (three blocks of rows: 148 KB) fits (n <= 6144)

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
  if (w >= 2 && !adv_hook("ADV_ROI_FWD_DIRECT"))
    hipLaunchKernelGGL(roi_align_fwd<true>, dim3(grid_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), feat, rois, out, c, h, w,
                       total, ph, pw, spatial_scale, sampling_ratio);
  else  // a one-column map, or ADV_ROI_FWD_DIRECT=1 (test hook): four single-dword gathers per sample
    hipLaunchKernelGGL(roi_align_fwd<false>, dim3(grid_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), feat, rois, out, c, h, w,
                       total, ph, pw, spatial_scale, sampling_ratio);
  return finish();
}

// how many segments of consecutive roi indices the backward sums separately before adding them in segment order (1: one ordered sum).
// A function of the roi count ALONE - part of the documented float32 order, restated by the oracle (roi_align_bwd_ordered(segments=)).
int adv_roi_align_bwd_segments(int r) {
  if (const char* v = adv_hook_value("ADV_ROI_SEGMENTS")) return std::max(1, std::min(8, std::atoi(v)));     // test-hook build only
  // measured (profiles/r04_roi_bwd.jsonl): on 510 proposals the kernel is bound by its instruction count, not by a tile's serial chain -
  // eight segments cost 0.25 ms more than one (the extra pass) and gained nothing; segments only where the lists get very long
  return r <= 1024 ? 1 : std::min(8, (r + 511) / 512);
}

int64_t adv_roi_align_bwd_workspace_ints(int b, int c, int h, int w, int r, int ph, int pw) {
  if (b < 1 || c < 1 || h < 1 || w < 1 || r < 0 || ph < 1 || pw < 1) return 0;
  const long long tiles = static_cast<long long>((h + kTileY - 1) / kTileY) * ((w + kTileX - 1) / kTileX);
  const int g = adv_roi_align_bwd_segments(r);
  const long long lists = (static_cast<long long>(b) * g * tiles * (r + 1) + 3) & ~3LL;
  const long long gcl = (static_cast<long long>(ph) * pw * r * cpad(c) + 3) & ~3LL;            // the channel-last copy of grad_out
  // <round 5> one ordered sum (g == 1): the per-roi axis tables - counts [r][h + w], entries [r][h + w][kListCap] x 2 ints, sample counts [r]
  const long long tabs = g == 1 ? ((static_cast<long long>(r) * (h + w) + 3) & ~3LL) + static_cast<long long>(r) * (h + w) * kListCap * 2 + ((r + 3) & ~3LL) : 0;
  return lists + gcl + tabs + (g > 1 ? static_cast<long long>(g) * b * c * h * w : 0);         // + one copy of the map per segment
}

// gcl_in != NULL: the channel-last copy of grad_out made by the caller (adv_roi_gout_channel_last_f32) - a pyramid pools ONE roi list from four
// levels, and the backward of each level reads the same grad_out: transposed once instead of once per level
static int roi_bwd_impl(const float* grad_out, const float* gcl_in, const float* rois, float* grad_feat, int b, int c, int h, int w, int r, int ph,
                        int pw, float spatial_scale, int sampling_ratio, int32_t* workspace, adv_stream_t stream) {
  const int rc = check_roi(gcl_in ? gcl_in : grad_out, rois, grad_feat, b, c, h, w, r, ph, pw);
  if (rc != ADV_OK) return rc;
  if (workspace == nullptr || b > 65535) return ADV_EINVAL;
  // the channel-last copy of grad_out inside the workspace is read with 16-byte loads: the lists before it are padded to a multiple of
  // four ints, so the workspace itself must start on a 16-byte boundary
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return ADV_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles_y = (h + kTileY - 1) / kTileY, tiles_x = (w + kTileX - 1) / kTileX;
  if (static_cast<long long>(ph) * pw > 1500) return ADV_EINVAL;      // the transposing kernel stages 32 x ph*pw floats in LDS
  const int G = adv_hook("ADV_ROI_BWD_REGS") ? 1 : adv_roi_align_bwd_segments(r);
  if (static_cast<long long>(b) * G > 65535) return ADV_EINVAL;
  const int seg_len = (r + G - 1) / G;
  const long long seg_elems = static_cast<long long>(b) * c * h * w;
  float* gcl_ws = reinterpret_cast<float*>(workspace) + ((static_cast<long long>(b) * G * tiles_y * tiles_x * (r + 1) + 3) & ~3LL);
  float* parts = gcl_ws + ((static_cast<long long>(ph) * pw * r * cpad(c) + 3) & ~3LL);
  float* dest = G > 1 ? parts : grad_feat;
  if (gcl_in && (reinterpret_cast<uintptr_t>(gcl_in) & 15) != 0) return ADV_EALIGN;
  const float* gcl = gcl_in ? gcl_in : gcl_ws;
  if (r > 0 && !gcl_in)
    hipLaunchKernelGGL(roi_gout_channel_last, dim3(r, (c + 31) / 32), dim3(kBlock), sizeof(float) * 32 * (ph * pw + 1), st, grad_out, gcl_ws, c, ph * pw);
  // every element of grad_feat is written by its owning lane (zeros where no roi reaches): no memset needed
  hipLaunchKernelGGL(roi_tile_lists, dim3(tiles_y * tiles_x, b * G), dim3(64), 0, st, rois, r, h, w, tiles_y, tiles_x, ph, pw, spatial_scale,
                     sampling_ratio, reinterpret_cast<int*>(workspace), G, std::max(1, seg_len));
  // channels per lane: as many as still leave ~8 workgroups per compute unit (the small pyramid levels have 6-20 tiles)
  const long long tiles = static_cast<long long>(tiles_y) * tiles_x * b;
  // <round 5> the shipped route for one ordered sum: per-roi axis tables + a wave per pixel (ADV_ROI_BWD_LDS=1 or one of the older routes'
  // own hooks: the formulations below - the same bits)
  const bool older = adv_hook("ADV_ROI_BWD_LDS") || adv_hook("ADV_ROI_BWD_REGS") || adv_hook("ADV_ROI_BWD_SCALAR_ITEMS") || adv_hook("ADV_ROI_BWD_NO_STAGE") ||
                     adv_hook("ADV_ROI_ACC_CHAN") || adv_hook("ADV_ROI_DBG");
  if (G == 1 && !older && static_cast<long long>(r) * ph * pw * cpad(c) < (1LL << 31) && static_cast<long long>(r) * (h + w) * kListCap < (1LL << 31)) {
    int* n_tab = reinterpret_cast<int*>(parts);
    int2* e_tab = reinterpret_cast<int2*>(n_tab + ((static_cast<long long>(r) * (h + w) + 3) & ~3LL));
    float* cnt = reinterpret_cast<float*>(e_tab + static_cast<long long>(r) * (h + w) * kListCap);
    if (r > 0)
      hipLaunchKernelGGL(roi_axis_tables, dim3((h + w + kBlock - 1) / kBlock, r), dim3(kBlock), 0, st, rois, b, h, w, ph, pw, spatial_scale, sampling_ratio,
                         n_tab, e_tab, cnt);
    const int groups_x = (w + kTabPix - 1) / kTabPix;
    hipLaunchKernelGGL(roi_align_bwd_tab, dim3(groups_x * h, (c + kTabChan - 1) / kTabChan, b), dim3(kBlock), 0, st, gcl, rois,
                       reinterpret_cast<const int*>(workspace), n_tab, e_tab, cnt, grad_feat, c, h, w, r, tiles_y, tiles_x, groups_x, ph, pw, spatial_scale,
                       sampling_ratio);
    return finish();
  }
  const char* dbg_s = adv_hook_value("ADV_ROI_DBG");
  const int dbg = dbg_s ? dbg_s[0] - '0' : 0;
  if (!adv_hook("ADV_ROI_BWD_REGS")) {            // the shipped route: accumulators in LDS, lanes = (touched pixel, channel)
    const char* dbg2_s = adv_hook_value("ADV_ROI_DBG");
    const int dbg = dbg2_s ? dbg2_s[0] - '0' : 0;
    const char* ac_s = adv_hook_value("ADV_ROI_ACC_CHAN");
    const int ac = ac_s ? std::atoi(ac_s) : 16;      // 16 channels per workgroup: the fastest of 8 / 16 / 32 on the R101-shaped proposals (profiles/r03_roi_bwd.jsonl)
    const bool stage = ph * pw <= kStageBins && !adv_hook("ADV_ROI_BWD_NO_STAGE");
    const bool scalar_items = adv_hook("ADV_ROI_BWD_SCALAR_ITEMS");      // test hook: round 3's one-channel items (same bits)
#define ADV_LAUNCH_ROI_LDS(AC_)                                                                                                               \
  do {                                                                                                                                        \
    const dim3 grid(tiles_y * tiles_x, (c + AC_ - 1) / AC_, b * G);                                                                           \
    if (stage) {                                                                                                                              \
      const size_t dyn = 2 * sizeof(float) * static_cast<size_t>(ph) * pw * AC_;                                                              \
      if (!adv_internal_lds_limit<roi_align_bwd_lds<true, AC_>>(2 * sizeof(float) * kStageBins * AC_)) return ADV_ELAUNCH;                    \
      if (!adv_internal_lds_limit<roi_align_bwd_lds<true, AC_, true>>(2 * sizeof(float) * kStageBins * AC_)) return ADV_ELAUNCH;              \
      if (scalar_items)                                                                                                                     \
        hipLaunchKernelGGL((roi_align_bwd_lds<true, AC_>), grid, dim3(kBlock), dyn, st, gcl, rois, reinterpret_cast<const int*>(workspace),   \
                           dest, c, h, w, r, tiles_y, tiles_x, ph, pw, spatial_scale, sampling_ratio, dbg, G, seg_elems);                     \
      else                                                                                                                                  \
        hipLaunchKernelGGL((roi_align_bwd_lds<true, AC_, true>), grid, dim3(kBlock), dyn, st, gcl, rois,                                      \
                           reinterpret_cast<const int*>(workspace), dest, c, h, w, r, tiles_y, tiles_x, ph, pw, spatial_scale,                \
                           sampling_ratio, dbg, G, seg_elems);                                                                                \
    } else {                                                                                                                                  \
      hipLaunchKernelGGL((roi_align_bwd_lds<false, AC_>), grid, dim3(kBlock), 0, st, gcl, rois, reinterpret_cast<const int*>(workspace),      \
                         dest, c, h, w, r, tiles_y, tiles_x, ph, pw, spatial_scale, sampling_ratio, dbg, G, seg_elems);                       \
    }                                                                                                                                         \
  } while (0)
    if (ac == 32) ADV_LAUNCH_ROI_LDS(32);
    else if (ac == 16) ADV_LAUNCH_ROI_LDS(16);
    else ADV_LAUNCH_ROI_LDS(8);
#undef ADV_LAUNCH_ROI_LDS
    if (G > 1)
      hipLaunchKernelGGL(roi_bwd_sum_segments, dim3(static_cast<unsigned>((seg_elems + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, parts,
                         reinterpret_cast<const int*>(workspace), grad_feat, c, h, w, r, tiles_y, tiles_x, G, seg_elems);
    return finish();
  }
  int cb = kChanBlock;
  while (cb > 1 && tiles * ((c + cb - 1) / cb) < 2048) cb >>= 1;
  if (adv_hook("ADV_ROI_BWD_CB8")) cb = kChanBlock;
#define ADV_LAUNCH_ROI_BWD(CB_)                                                                                                             \
  hipLaunchKernelGGL(roi_align_bwd_gather<CB_>, dim3(tiles_y * tiles_x, (c + CB_ - 1) / CB_, b), dim3(kBlock), 0, st, gcl, rois,                \
                     reinterpret_cast<const int*>(workspace), grad_feat, c, h, w, r, tiles_y, tiles_x, ph, pw, spatial_scale, sampling_ratio, dbg)
  if (cb == 8) ADV_LAUNCH_ROI_BWD(8);
  else if (cb == 4) ADV_LAUNCH_ROI_BWD(4);
  else if (cb == 2) ADV_LAUNCH_ROI_BWD(2);
  else ADV_LAUNCH_ROI_BWD(1);
#undef ADV_LAUNCH_ROI_BWD
  return finish();
}

int adv_roi_align_bwd_f32(const float* grad_out, const float* rois, float* grad_feat, int b, int c, int h, int w, int r, int ph, int pw,
                          float spatial_scale, int sampling_ratio, int32_t* workspace, adv_stream_t stream) {
  return roi_bwd_impl(grad_out, nullptr, rois, grad_feat, b, c, h, w, r, ph, pw, spatial_scale, sampling_ratio, workspace, stream);
}

int64_t adv_roi_gout_channel_last_floats(int c, int r, int ph, int pw) {
  if (c < 1 || r < 0 || ph < 1 || pw < 1) return 0;
  return static_cast<int64_t>(ph) * pw * r * cpad(c);
}

int adv_roi_gout_channel_last_f32(const float* grad_out, float* gcl, int c, int r, int ph, int pw, adv_stream_t stream) {
  if (!grad_out || !gcl || c < 1 || r < 0 || ph < 1 || pw < 1 || static_cast<long long>(ph) * pw > 1500) return ADV_EINVAL;
  if (!aligned4(grad_out) || (reinterpret_cast<uintptr_t>(gcl) & 15) != 0) return ADV_EALIGN;
  if (r == 0) return ADV_OK;
  hipLaunchKernelGGL(roi_gout_channel_last, dim3(r, (c + 31) / 32), dim3(kBlock), sizeof(float) * 32 * (ph * pw + 1), static_cast<hipStream_t>(stream),
                     grad_out, gcl, c, ph * pw);
  return finish();
}

int adv_roi_align_bwd_cl_f32(const float* gcl, const float* rois, float* grad_feat, int b, int c, int h, int w, int r, int ph, int pw,
                             float spatial_scale, int sampling_ratio, int32_t* workspace, adv_stream_t stream) {
  if (!gcl) return ADV_EINVAL;
  return roi_bwd_impl(nullptr, gcl, rois, grad_feat, b, c, h, w, r, ph, pw, spatial_scale, sampling_ratio, workspace, stream);
}

int adv_nms_f32(const float* boxes, int n, float thresh, int64_t* keep_out, int32_t* num_keep_out, uint64_t* workspace,
                adv_stream_t stream) {
  if (!num_keep_out || n < 0 || (n > 0 && (!boxes || !keep_out || !workspace))) return ADV_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) return hipMemsetAsync(num_keep_out, 0, sizeof(int32_t), st) == hipSuccess ? ADV_OK : ADV_ELAUNCH;
  const int col_blocks = (n + 63) / 64;
  if (static_cast<size_t>(col_blocks) * 8 > 64 * 1024) return ADV_EINVAL;  // removed[] must fit 64 KiB of LDS: n <= 524288
  {
    const long long words = static_cast<long long>(n) * col_blocks;
    const unsigned zb = static_cast<unsigned>(words / 256 + 1 < 1024 ? words / 256 + 1 : 1024);
    hipLaunchKernelGGL(nms_zero, dim3(zb), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(workspace), words);
  }
  hipLaunchKernelGGL(nms_mask, dim3(col_blocks, col_blocks), dim3(64), 0, st, boxes, n, thresh,
                     reinterpret_cast<unsigned long long*>(workspace), col_blocks);
  if (col_blocks <= kNmsBlockScanMaxWords && !adv_hook("ADV_NMS_BOX_SCAN")) {
    const size_t lds = (static_cast<size_t>(col_blocks) * 193 + 1) * 8;
    if (!adv_internal_lds_limit<nms_scan_blocks>((static_cast<size_t>(kNmsBlockScanMaxWords) * 193 + 1) * 8)) return ADV_ELAUNCH;
    hipLaunchKernelGGL(nms_scan_blocks, dim3(1), dim3(kNmsScanThreads), lds, st, reinterpret_cast<const unsigned long long*>(workspace), n, col_blocks,
                       reinterpret_cast<long long*>(keep_out), reinterpret_cast<int*>(num_keep_out));
    return finish();
  }
  hipLaunchKernelGGL(nms_scan, dim3(1), dim3(64), static_cast<size_t>(col_blocks) * 8, st,
                     reinterpret_cast<const unsigned long long*>(workspace), n, col_blocks, reinterpret_cast<long long*>(keep_out),
                     reinterpret_cast<int*>(num_keep_out));
  return finish();
}

}  // extern "C"
