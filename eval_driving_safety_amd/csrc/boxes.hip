// Box arithmetic of the proposal / target stage of a Stereo R-CNN step, one kernel per chain of torch one-liners (round 5).
// The reference reaches this code inside the detector call of attack/Stereo-RCNN/pgd_attack.py:156 (upstream lib/model/rpn/
// bbox_transform.py: bbox_overlaps, bbox_transform, bbox_transform_inv, clip_boxes [UPSTREAM-UNVERIFIED paths]); this package's layer-list
// graph (surrogates.py) wrote them as ~15 element-wise torch operators each - 3-5 us per launch, a few hundred launches per step.
// Every kernel evaluates the SAME float32 expressions in the same order as those operators (no contraction: -ffp-contract=off; logf / expf
// are the device library's, as in torch's kernels), so the results are the operators' bit for bit on the device.  Also here: the
// proposal bookkeeping (stable size partition, roi sampling: single-workgroup kernels, exact), the RPN head's list packing with its backward,
// and the attack script's six-term objective chain (attack/Stereo-RCNN/pgd_attack.py:165-171) in its own order of additions.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

namespace {

constexpr int kBlock = 256;

struct Box {
  float x1, y1, x2, y2;
};
__device__ __forceinline__ Box load_box(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return Box{v.x, v.y, v.z, v.w};
}

// surrogates._iou: legacy +1 areas
__device__ __forceinline__ float iou_of(const Box& a, const Box& b) {
  const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1), rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
  const float w = fmaxf(rbx - ltx + 1.0f, 0.0f), h = fmaxf(rby - lty + 1.0f, 0.0f);
  const float inter = w * h;
  const float area_a = (a.x2 - a.x1 + 1.0f) * (a.y2 - a.y1 + 1.0f), area_b = (b.x2 - b.x1 + 1.0f) * (b.y2 - b.y1 + 1.0f);
  return inter / (area_a + area_b - inter);
}

// one lane per box of ``a``: its IoU with every box of ``b`` (written to iou[N][M] if asked for), the largest and the index of the FIRST largest
__global__ __launch_bounds__(kBlock) void box_iou_rows(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ iou,
                                                      float* __restrict__ best, long long* __restrict__ arg, long long n, int m) {
  const long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  if (i >= n) return;
  const Box bi = load_box(a + 4 * i);
  float bv = 0.0f;
  long long bj = 0;
  for (int j = 0; j < m; ++j) {
    const float v = iou_of(bi, load_box(b + 4 * j));
    if (iou) iou[i * m + j] = v;
    if (j == 0 || v > bv || (v != v && bv == bv)) bv = v, bj = j;      // (torch.max: the first maximum; a NaN wins)
  }
  best[i] = bv;
  arg[i] = bj;
}

struct Enc {
  float dx, dy, dw, dh;
};
// surrogates._encode
__device__ __forceinline__ Enc encode(const Box& s, const Box& d) {
  const float sw = s.x2 - s.x1 + 1.0f, sh = s.y2 - s.y1 + 1.0f;
  const float sx = s.x1 + 0.5f * sw, sy = s.y1 + 0.5f * sh;
  const float dw = d.x2 - d.x1 + 1.0f, dh = d.y2 - d.y1 + 1.0f;
  const float dx = d.x1 + 0.5f * dw, dy = d.y1 + 0.5f * dh;
  return Enc{(dx - sx) / sw, (dy - sy) / sh, logf(dw / sw), logf(dh / sh)};
}

// out[i] = (encode(src_i, gt_l[arg_i]), encode(src_r_i, gt_r[arg_i]).dx, .dw): the six regression targets of a stereo box pair (src_r = src for anchors)
__global__ __launch_bounds__(kBlock) void box_encode6(const float* __restrict__ src, const float* __restrict__ src_r, const float* __restrict__ gt_l, const float* __restrict__ gt_r,
                                                     const long long* __restrict__ arg, float* __restrict__ out, long long n, int m) {
  const long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  if (i >= n) return;
  long long j = arg[i];
  j = j < 0 ? 0 : (j >= m ? m - 1 : j);
  const Enc l = encode(load_box(src + 4 * i), load_box(gt_l + 4 * j)), r = encode(load_box(src_r + 4 * i), load_box(gt_r + 4 * j));
  float* o = out + 6 * i;
  o[0] = l.dx, o[1] = l.dy, o[2] = l.dw, o[3] = l.dh, o[4] = r.dx, o[5] = r.dw;
}

// surrogates._decode + the clip to the image
__device__ __forceinline__ Box decode_clip(const Box& s, float d0, float d1, float d2, float d3, float wmax, float hmax) {
  const float sw = s.x2 - s.x1 + 1.0f, sh = s.y2 - s.y1 + 1.0f;
  const float sx = s.x1 + 0.5f * sw, sy = s.y1 + 0.5f * sh;
  const float cx = d0 * sw + sx, cy = d1 * sh + sy;
  const float w = expf(fminf(d2, 4.0f)) * sw, h = expf(fminf(d3, 4.0f)) * sh;
  Box o{cx - 0.5f * w, cy - 0.5f * h, cx + 0.5f * w - 1.0f, cy + 0.5f * h - 1.0f};
  o.x1 = fminf(fmaxf(o.x1, 0.0f), wmax), o.x2 = fminf(fmaxf(o.x2, 0.0f), wmax);
  o.y1 = fminf(fmaxf(o.y1, 0.0f), hmax), o.y2 = fminf(fmaxf(o.y2, 0.0f), hmax);
  return o;
}

// left = decode(a, d[0..3]), right = decode(a, (d4, d1, d5, d3)), both clipped to [0, W-1] x [0, H-1]; big = both wide and the left one high enough
__global__ __launch_bounds__(kBlock) void box_decode_stereo(const float* __restrict__ anchors, const float* __restrict__ d, float* __restrict__ left,
                                                           float* __restrict__ right, long long* __restrict__ big, long long n, float wmax, float hmax,
                                                           float min_size) {
  const long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  if (i >= n) return;
  const Box s = load_box(anchors + 4 * i);
  const float* di = d + 6 * i;
  const Box l = decode_clip(s, di[0], di[1], di[2], di[3], wmax, hmax), r = decode_clip(s, di[4], di[1], di[5], di[3], wmax, hmax);
  *reinterpret_cast<float4*>(left + 4 * i) = float4{l.x1, l.y1, l.x2, l.y2};
  *reinterpret_cast<float4*>(right + 4 * i) = float4{r.x1, r.y1, r.x2, r.y2};
  if (big) big[i] = (l.x2 - l.x1 + 1.0f >= min_size && l.y2 - l.y1 + 1.0f >= min_size && r.x2 - r.x1 + 1.0f >= min_size) ? 1 : 0;
}

// ---- proposal bookkeeping of one image, each a single workgroup (a few thousand boxes at most): what the static forward did with two
//      sorts, a dozen gathers and concatenations.
constexpr int kOneBlock = 1024;

// block-wide count of ``flag`` and this thread's exclusive prefix of it (16 waves: ballots + one LDS round)
__device__ __forceinline__ int block_prefix(bool flag, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int in_wave = __popcll(m & ((1ULL << lane) - 1ULL));
  __syncthreads();                                   // (s_wave is reused from call to call)
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  int base = 0, sum = 0;
  for (int w = 0; w < kOneBlock / 64; ++w) {
    const int c = s_wave[w];
    if (w < wave) base += c;
    sum += c;
  }
  *total = sum;
  return base + in_wave;
}

// STABLE partition: the boxes with big != 0 first, in order, then the others, in order (= gathering with argsort(1 - big, stable)); nothing
// moves if no box is big.  nvalid = the number of big boxes (n if none is).
__global__ __launch_bounds__(kOneBlock) void box_partition_stereo(const float* __restrict__ left, const float* __restrict__ right,
                                                                  const long long* __restrict__ big, float* __restrict__ out_left,
                                                                  float* __restrict__ out_right, long long* __restrict__ nvalid, int n) {
  __shared__ int s_wave[kOneBlock / 64];
  int nbig = 0;
  for (int i0 = 0; i0 < n; i0 += kOneBlock) {        // first pass: how many are big
    int t;
    const int i = i0 + static_cast<int>(threadIdx.x);
    block_prefix(i < n && big[i] != 0, s_wave, &t);
    nbig += t;
  }
  int seen_big = 0;
  for (int i0 = 0; i0 < n; i0 += kOneBlock) {
    const int i = i0 + static_cast<int>(threadIdx.x);
    const bool in = i < n, flag = in && big[i] != 0;
    int t;
    const int before = block_prefix(flag, s_wave, &t) + seen_big;      // big boxes before box i
    if (in) {
      const int dst = nbig == 0 ? i : (flag ? before : nbig + (i - before));
      *reinterpret_cast<float4*>(out_left + 4LL * dst) = *reinterpret_cast<const float4*>(left + 4LL * i);
      *reinterpret_cast<float4*>(out_right + 4LL * dst) = *reinterpret_cast<const float4*>(right + 4LL * i);
    }
    seen_big += t;
  }
  if (threadIdx.x == 0) *nvalid = nbig == 0 ? n : nbig;
}

// The rois of one image: candidates = (ground truth boxes, then the kept proposals left[keep[j]] for the j with 0 <= keep[j] < nvalid - a
// prefix of ``keep``), sampled in order with replacement: roi i = candidate i % max(count, 1).  Writes (0, box) rows and the boxes alone.
__global__ __launch_bounds__(kOneBlock) void box_sample_rois(const long long* __restrict__ keep, int k, const long long* __restrict__ nvalid,
                                                             const float* __restrict__ left, const float* __restrict__ right,
                                                             const float* __restrict__ gt_l, const float* __restrict__ gt_r, int n_gt, int R,
                                                             float* __restrict__ rois_l, float* __restrict__ rois_r, float* __restrict__ out_left,
                                                             float* __restrict__ out_right) {
  __shared__ int s_wave[kOneBlock / 64];
  const long long nv = *nvalid;
  int nkeep = 0;
  for (int j0 = 0; j0 < k; j0 += kOneBlock) {
    const int j = j0 + static_cast<int>(threadIdx.x);
    int t;
    block_prefix(j < k && keep[j] >= 0 && keep[j] < nv, s_wave, &t);
    nkeep += t;
  }
  const int total = max(nkeep + n_gt, 1);
  for (int i = threadIdx.x; i < R; i += kOneBlock) {
    const int idx = i % total;
    float4 l, r;
    if (idx < n_gt) {
      l = *reinterpret_cast<const float4*>(gt_l + 4LL * idx), r = *reinterpret_cast<const float4*>(gt_r + 4LL * idx);
    } else {
      const long long src = max(keep[idx - n_gt], 0LL);
      l = *reinterpret_cast<const float4*>(left + 4 * src), r = *reinterpret_cast<const float4*>(right + 4 * src);
    }
    *reinterpret_cast<float4*>(out_left + 4LL * i) = l;
    *reinterpret_cast<float4*>(out_right + 4LL * i) = r;
    float* pl = rois_l + 5LL * i;
    float* pr = rois_r + 5LL * i;
    pl[0] = 0.0f, pl[1] = l.x, pl[2] = l.y, pl[3] = l.z, pl[4] = l.w;
    pr[0] = 0.0f, pr[1] = r.x, pr[2] = r.y, pr[3] = r.z, pr[4] = r.w;
  }
}

// ---- the RPN head's output of one pyramid level, [B][A + 6A][H*W] (A objectness maps, then six regression maps per anchor), into the
//      proposal stage's lists: scores [(b, pixel, a)] and deltas [(b, pixel, a)][6] (= permute(0, 2, 3, 1).reshape(...) of the two slices),
//      written at the level's offset of the lists of ALL levels; ``bounded``: deltas = 0.5 * tanh(raw) (surrogates.StereoRcnnR101).  And
//      back: the gradient w.r.t. the head's output from the gradients of the two lists (tanh's: g * 0.5 * (1 - y * y), torch's order).
__global__ __launch_bounds__(kBlock) void rpn_pack_fwd(const float* __restrict__ src, float* __restrict__ scores, float* __restrict__ deltas, int A,
                                                      long long hw, long long total, int bounded) {
  const long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;      // (b, pixel, a)
  if (i >= total) return;
  const int a = static_cast<int>(i % A);
  const long long p = (i / A) % hw, b = i / (A * hw);
  const float* sp = src + b * 7 * A * hw + p;
  scores[i] = sp[a * hw];
  float* d = deltas + 6 * i;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float raw = sp[(A + a * 6 + j) * hw];
    d[j] = bounded ? 0.5f * tanhf(raw) : raw;
  }
}

__global__ __launch_bounds__(kBlock) void rpn_pack_bwd(const float* __restrict__ src, const float* __restrict__ g_scores,
                                                      const float* __restrict__ g_deltas, float* __restrict__ g_src, int A, long long hw,
                                                      long long total, int bounded) {
  const long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  if (i >= total) return;
  const int a = static_cast<int>(i % A);
  const long long p = (i / A) % hw, b = i / (A * hw);
  const long long base = b * 7 * A * hw + p;
  g_src[base + a * hw] = g_scores ? g_scores[i] : 0.0f;
  const float* g = g_deltas ? g_deltas + 6 * i : nullptr;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const long long at = base + (A + a * 6 + j) * hw;
    float v = g ? g[j] : 0.0f;
    if (bounded) {
      const float y = tanhf(src[at]);
      v = (v * 0.5f) * fmaf(-y, y, 1.0f);      // torch's tanh_backward kernel, a * (1 - b * b), is built with contraction: 1 - b * b is one fma
    }
    g_src[at] = v;
  }
}

// ---- the Stereo R-CNN attack objective (attack/Stereo-RCNN/pgd_attack.py:165-171): loss = sum_k (term_k * exp(-u_k) + u_k), added in the
//      script's order - ((0 + t0 w0) + u0) + t1 w1 ... - by ONE thread; w_k = exp(-u_k) is kept for the backward (d loss / d term_k = w_k).
//      The script's six-fold loop is ~45 scalar launches forward and ~30 backward through torch.
__global__ __launch_bounds__(64) void objective_chain(const float* __restrict__ terms, const float* __restrict__ u, float* __restrict__ loss,
                                                     float* __restrict__ w, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = 0.0f;
  for (int k = 0; k < n; ++k) {
    const float wk = expf(-u[k]);
    w[k] = wk;
    acc = acc + terms[k] * wk;
    acc = acc + u[k];
  }
  loss[0] = acc;
}

// ---- masked-mean losses: loss = sum_i w_i * l(pred_i, target_i) / max(scale * sum_i w_i, 1) with l = smooth-L1 (beta 1) over the K columns of
//      row i, or binary cross-entropy with logits (K = 1) - the RPN / RCNN loss terms of the layer-list graph, which torch computes with
//      six or seven launches each (+ four or five backward).  Partial sums per block (a fixed chunk per block, a tree inside it), then one
//      thread adds the partials in order: deterministic; the VALUE differs from torch's own summation order in the last bits, the GRADIENT
//      does not depend on that order and is evaluated with torch's expressions in torch's order: bit-equal (tests/test_boxes.py).
constexpr int kLossBlock = 256, kLossMaxBlocks = 256;

template <bool BCE>
__device__ __forceinline__ float loss_value(float p, float t) {
  if (BCE) {                                        // (1 - y) x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
    const float ls = fminf(p, 0.0f) - log1pf(expf(-fabsf(p)));
    return (1.0f - t) * p - ls;
  }
  const float z = fabsf(p - t);                     // smooth-L1, beta = 1
  return z < 1.0f ? 0.5f * z * z : z - 0.5f;
}

template <bool BCE>
__global__ __launch_bounds__(kLossBlock) void masked_loss_partial(const float* __restrict__ pred, const float* __restrict__ target,
                                                                 const float* __restrict__ weight, float* __restrict__ partial, long long rows,
                                                                 int K, long long chunk) {
  __shared__ float s_l[kLossBlock], s_w[kLossBlock];
  const long long r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
  float sl = 0.0f, sw = 0.0f;
  for (long long r = r0 + threadIdx.x; r < r1; r += kLossBlock) {
    const float w = weight[r];
    sw = sw + w;
    for (int k = 0; k < K; ++k) sl = sl + loss_value<BCE>(pred[r * K + k], target[r * K + k]) * w;
  }
  s_l[threadIdx.x] = sl, s_w[threadIdx.x] = sw;
  __syncthreads();
  for (int d = kLossBlock / 2; d > 0; d >>= 1) {
    if (static_cast<int>(threadIdx.x) < d) s_l[threadIdx.x] = s_l[threadIdx.x] + s_l[threadIdx.x + d], s_w[threadIdx.x] = s_w[threadIdx.x] + s_w[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[2 * blockIdx.x] = s_l[0], partial[2 * blockIdx.x + 1] = s_w[0];
}

__global__ void masked_loss_finish(const float* __restrict__ partial, int blocks, float scale, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float sl = 0.0f, sw = 0.0f;
  for (int b = 0; b < blocks; ++b) sl = sl + partial[2 * b], sw = sw + partial[2 * b + 1];
  const float d = fmaxf(scale * sw, 1.0f);
  out[0] = sl / d;                                   // the loss
  out[1] = d;                                        // its denominator, for the backward
}

// grad_pred = l'(pred, target) * ((g / denominator) * w): torch's smooth_l1_loss_backward (x < -1 ? -go : x > 1 ? go : x * go) and
// binary_cross_entropy_with_logits backward ((sigmoid(x) - y) * go)
template <bool BCE>
__global__ __launch_bounds__(kLossBlock) void masked_loss_bwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                             const float* __restrict__ weight, const float* __restrict__ out,
                                                             const float* __restrict__ g, float* __restrict__ grad, long long total, int K) {
  const long long i = blockIdx.x * static_cast<long long>(kLossBlock) + threadIdx.x;
  if (i >= total) return;
  const float go = (g[0] / out[1]) * weight[i / K];
  const float p = pred[i], t = target[i];
  float r;
  if (BCE) {
    const float sg = 1.0f / (1.0f + expf(-p));
    r = (sg - t) * go;
  } else {
    const float x = p - t;
    r = x < -1.0f ? -go : (x > 1.0f ? go : x * go);
  }
  grad[i] = r;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
inline bool al4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
inline unsigned blocks_for(long long n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" {

int adv_box_iou_rows_f32(const float* a, const float* b, float* iou, float* best, int64_t* arg, int64_t n, int m, adv_stream_t stream) {
  if (!a || !b || !best || !arg || n < 0 || m < 1 || n > (1LL << 40)) return ADV_EINVAL;
  if (n == 0) return ADV_OK;
  if (!al16(a) || !al16(b) || !al4(best) || !al8(arg) || (iou && !al4(iou))) return ADV_EALIGN;
  hipLaunchKernelGGL(box_iou_rows, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a, b, iou, best,
                     reinterpret_cast<long long*>(arg), static_cast<long long>(n), m);
  return adv_internal_finish_launch();
}

int adv_box_encode6_f32(const float* src, const float* src_right, const float* gt_left, const float* gt_right, const int64_t* arg, float* out,
                        int64_t n, int m, adv_stream_t stream) {
  if (!src || !gt_left || !gt_right || !arg || !out || n < 0 || m < 1) return ADV_EINVAL;
  if (n == 0) return ADV_OK;
  if (!src_right) src_right = src;
  if (!al16(src) || !al16(src_right) || !al16(gt_left) || !al16(gt_right) || !al8(arg) || !al4(out)) return ADV_EALIGN;
  hipLaunchKernelGGL(box_encode6, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), src, src_right, gt_left, gt_right,
                     reinterpret_cast<const long long*>(arg), out, static_cast<long long>(n), m);
  return adv_internal_finish_launch();
}

int adv_box_decode_stereo_f32(const float* anchors, const float* deltas, float* left, float* right, int64_t* big, int64_t n, float width,
                              float height, float min_size, adv_stream_t stream) {
  if (!anchors || !deltas || !left || !right || n < 0 || left == right) return ADV_EINVAL;
  if (n == 0) return ADV_OK;
  if (!al16(anchors) || !al4(deltas) || !al16(left) || !al16(right) || (big && !al8(big))) return ADV_EALIGN;
  hipLaunchKernelGGL(box_decode_stereo, dim3(blocks_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), anchors, deltas, left, right,
                     reinterpret_cast<long long*>(big), static_cast<long long>(n), width - 1.0f, height - 1.0f, min_size);
  return adv_internal_finish_launch();
}

int64_t adv_masked_loss_workspace_floats(void) { return 2 * kLossMaxBlocks; }

static int masked_loss_blocks(long long rows, long long* chunk) {
  long long blocks = (rows + 4 * kLossBlock - 1) / (4 * kLossBlock);
  if (blocks < 1) blocks = 1;
  if (blocks > kLossMaxBlocks) blocks = kLossMaxBlocks;
  *chunk = (rows + blocks - 1) / blocks;
  return static_cast<int>((rows + *chunk - 1) / *chunk);
}

int adv_masked_loss_f32(const float* pred, const float* target, const float* weight, float* out2, float* workspace, int64_t rows, int k, float scale,
                        int bce, adv_stream_t stream) {
  if (!pred || !target || !weight || !out2 || !workspace || rows < 1 || k < 1 || (bce && k != 1)) return ADV_EINVAL;
  if (!al4(pred) || !al4(target) || !al4(weight) || !al4(out2) || !al4(workspace)) return ADV_EALIGN;
  long long chunk = 0;
  const int blocks = masked_loss_blocks(rows, &chunk);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bce) hipLaunchKernelGGL(masked_loss_partial<true>, dim3(blocks), dim3(kLossBlock), 0, st, pred, target, weight, workspace, static_cast<long long>(rows), k, chunk);
  else hipLaunchKernelGGL(masked_loss_partial<false>, dim3(blocks), dim3(kLossBlock), 0, st, pred, target, weight, workspace, static_cast<long long>(rows), k, chunk);
  hipLaunchKernelGGL(masked_loss_finish, dim3(1), dim3(64), 0, st, workspace, blocks, scale, out2);
  return adv_internal_finish_launch();
}

int adv_masked_loss_bwd_f32(const float* pred, const float* target, const float* weight, const float* out2, const float* grad_loss, float* grad_pred,
                            int64_t rows, int k, int bce, adv_stream_t stream) {
  if (!pred || !target || !weight || !out2 || !grad_loss || !grad_pred || rows < 1 || k < 1 || (bce && k != 1)) return ADV_EINVAL;
  if (!al4(pred) || !al4(target) || !al4(weight) || !al4(out2) || !al4(grad_loss) || !al4(grad_pred)) return ADV_EALIGN;
  const long long total = static_cast<long long>(rows) * k;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((total + kLossBlock - 1) / kLossBlock);
  if (bce) hipLaunchKernelGGL(masked_loss_bwd<true>, dim3(blocks), dim3(kLossBlock), 0, st, pred, target, weight, out2, grad_loss, grad_pred, total, k);
  else hipLaunchKernelGGL(masked_loss_bwd<false>, dim3(blocks), dim3(kLossBlock), 0, st, pred, target, weight, out2, grad_loss, grad_pred, total, k);
  return adv_internal_finish_launch();
}

int adv_objective_chain_f32(const float* terms, const float* u, float* loss, float* w, int n, adv_stream_t stream) {
  if (!terms || !u || !loss || !w || n < 1 || n > 64) return ADV_EINVAL;
  if (!al4(terms) || !al4(u) || !al4(loss) || !al4(w)) return ADV_EALIGN;
  hipLaunchKernelGGL(objective_chain, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), terms, u, loss, w, n);
  return adv_internal_finish_launch();
}

int adv_rpn_pack_fwd_f32(const float* head, float* scores, float* deltas, int b, int anchors, int64_t hw, int bounded, adv_stream_t stream) {
  if (!head || !scores || !deltas || b < 1 || anchors < 1 || hw < 1) return ADV_EINVAL;
  if (!al4(head) || !al4(scores) || !al4(deltas)) return ADV_EALIGN;
  const long long total = static_cast<long long>(b) * anchors * hw;
  hipLaunchKernelGGL(rpn_pack_fwd, dim3(blocks_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), head, scores, deltas, anchors,
                     static_cast<long long>(hw), total, bounded ? 1 : 0);
  return adv_internal_finish_launch();
}

int adv_rpn_pack_bwd_f32(const float* head, const float* grad_scores, const float* grad_deltas, float* grad_head, int b, int anchors, int64_t hw,
                         int bounded, adv_stream_t stream) {
  if (!head || !grad_head || head == grad_head || b < 1 || anchors < 1 || hw < 1) return ADV_EINVAL;
  if (!al4(head) || !al4(grad_head) || (grad_scores && !al4(grad_scores)) || (grad_deltas && !al4(grad_deltas))) return ADV_EALIGN;
  const long long total = static_cast<long long>(b) * anchors * hw;
  hipLaunchKernelGGL(rpn_pack_bwd, dim3(blocks_for(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), head, grad_scores, grad_deltas, grad_head,
                     anchors, static_cast<long long>(hw), total, bounded ? 1 : 0);
  return adv_internal_finish_launch();
}

int adv_box_partition_stereo_f32(const float* left, const float* right, const int64_t* big, float* out_left, float* out_right, int64_t* nvalid,
                                 int n, adv_stream_t stream) {
  if (!left || !right || !big || !out_left || !out_right || !nvalid || n < 1 || left == out_left || right == out_right) return ADV_EINVAL;
  if (!al16(left) || !al16(right) || !al16(out_left) || !al16(out_right) || !al8(big) || !al8(nvalid)) return ADV_EALIGN;
  hipLaunchKernelGGL(box_partition_stereo, dim3(1), dim3(kOneBlock), 0, static_cast<hipStream_t>(stream), left, right,
                     reinterpret_cast<const long long*>(big), out_left, out_right, reinterpret_cast<long long*>(nvalid), n);
  return adv_internal_finish_launch();
}

int adv_box_sample_rois_f32(const int64_t* keep, int k, const int64_t* nvalid, const float* left, const float* right, const float* gt_left,
                            const float* gt_right, int n_gt, int r, float* rois_left, float* rois_right, float* out_left, float* out_right,
                            adv_stream_t stream) {
  if (!keep || !nvalid || !left || !right || !rois_left || !rois_right || !out_left || !out_right || k < 1 || n_gt < 0 || r < 1) return ADV_EINVAL;
  if (n_gt > 0 && (!gt_left || !gt_right)) return ADV_EINVAL;
  if (!al16(left) || !al16(right) || !al16(out_left) || !al16(out_right) || !al4(rois_left) || !al4(rois_right) || !al8(keep) || !al8(nvalid) ||
      (n_gt > 0 && (!al16(gt_left) || !al16(gt_right))))
    return ADV_EALIGN;
  hipLaunchKernelGGL(box_sample_rois, dim3(1), dim3(kOneBlock), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(keep), k,
                     reinterpret_cast<const long long*>(nvalid), left, right, gt_left, gt_right, n_gt, r, rois_left, rois_right, out_left, out_right);
  return adv_internal_finish_launch();
}

}  // extern "C"
