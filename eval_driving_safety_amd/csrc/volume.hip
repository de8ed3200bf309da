// What a plane-sweep stereo detector does with its cost volume AFTER the 3D convolutions (reached through
// attack/DSGN/pgd_attack.py:308,324 - upstream DSGN code, SURVEY 2.2: "trilinear upsample", "grid_sample PSV->3DGV",
// "sigmoid focal loss CUDA op"), written for gfx950:
//
//   depth regression   cost [B,D,h,w] -> trilinear upsample to [Do,H,W] -> softmax over Do -> sum_k p_k z_k = depth [B,H,W],
//                      FUSED: the up-sampled volume (Do*H*W = 92 M floats per image at the DSGN size, 368 MB) and its softmax
//                      are never written - a lane owns one output pixel and streams over the planes; HBM traffic is the 5.75 MB
//                      cost volume in and the 1.9 MB depth map out instead of ~2 GB.  Backward in two atomic-free stages
//                      (depth-axis adjoint per pixel, then the bilinear adjoint as a gather per cost cell).
//   grid sample 3D     F.grid_sample(volume [B,C,D,H,W], grid [B,Z,Y,X,3], bilinear, zeros): the eight corner offsets and
//                      weights of an output voxel are worked out once and reused for every channel; the backward w.r.t. the
//                      volume is a GATHER over a per-grid plan (for every input cell the sorted list of the output voxels
//                      that sample it): deterministic float32 sums, no atomics in the data path.
//   sigmoid focal loss elementwise, forward and gradient w.r.t. the logits.
//
// Floating point: exp/log are the device's (ocml) - parity with a float32 torch / numpy reference is within the tolerance the
// tests state (1e-5 relative), not bit-exact; everything else is plain IEEE float32 with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

namespace {

constexpr int kBlock = 256;

// torch's linear-interpolation source index (aten/native/UpSample.h: area_pixel_compute_source_index), float32
struct Lin {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ float lin_scale(int in, int out, int align) {
  if (align) return out > 1 ? static_cast<float>(in - 1) / static_cast<float>(out - 1) : 0.0f;
  return static_cast<float>(in) / static_cast<float>(out);
}

__device__ __forceinline__ Lin lin_at(int dst, int in, float scale, int align) {
  float src = align ? scale * static_cast<float>(dst) : scale * (static_cast<float>(dst) + 0.5f) - 0.5f;
  if (!align && src < 0.0f) src = 0.0f;
  Lin r;
  r.i0 = static_cast<int>(src);
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = fminf(fmaxf(src - static_cast<float>(r.i0), 0.0f), 1.0f);  // guard_index_and_lambda
  r.l0 = 1.0f - r.l1;
  return r;
}

// bilinear value of source plane dd at the pixel's (y, x) taps: innermost along w, then h (torch's nesting)
struct Pix {
  Lin y, x;
};

__device__ __forceinline__ float plane_at(const float* __restrict__ p, int w, const Pix& px) {
  const float top = px.x.l0 * p[px.y.i0 * w + px.x.i0] + px.x.l1 * p[px.y.i0 * w + px.x.i1];
  const float bot = px.x.l0 * p[px.y.i1 * w + px.x.i0] + px.x.l1 * p[px.y.i1 * w + px.x.i1];
  return px.y.l0 * top + px.y.l1 * bot;
}

// Streams the Do up-sampled values of one pixel in plane order: the two source planes a value interpolates between are
// kept in registers and advanced as the source index moves (it is non-decreasing in k).
struct Column {
  const float* cost;  // [D,h,w] of this batch element
  int D, h, w, Do, align;
  float sd;
  Pix px;
  int cur;
  float va, vb;
  __device__ __forceinline__ void start() {
    cur = 0;
    va = plane_at(cost, w, px);
    vb = D > 1 ? plane_at(cost + static_cast<long long>(h) * w, w, px) : va;
  }
  __device__ __forceinline__ float at(int k, Lin& lk) {
    lk = lin_at(k, D, sd, align);
    while (cur < lk.i0) {
      ++cur;
      va = vb;
      vb = cur + 1 < D ? plane_at(cost + static_cast<long long>(cur + 1) * h * w, w, px) : va;
    }
    return lk.l0 * va + lk.l1 * (lk.i1 > lk.i0 ? vb : va);
  }
};

__global__ __launch_bounds__(kBlock) void depth_regress_fwd(const float* __restrict__ cost, const float* __restrict__ zval,
                                                            float* __restrict__ depth, float* __restrict__ stats, int D, int h, int w,
                                                            int Do, int H, int W, int align) {
  const int X = blockIdx.x * kBlock + threadIdx.x, Y = blockIdx.y, b = blockIdx.z;
  if (X >= W) return;
  Column col;
  col.cost = cost + static_cast<long long>(b) * D * h * w;
  col.D = D, col.h = h, col.w = w, col.Do = Do, col.align = align;
  col.sd = lin_scale(D, Do, align);
  col.px.y = lin_at(Y, h, lin_scale(h, H, align), align);
  col.px.x = lin_at(X, w, lin_scale(w, W, align), align);
  Lin lk;
  col.start();
  float m = -INFINITY;
  for (int k = 0; k < Do; ++k) m = fmaxf(m, col.at(k, lk));
  col.start();
  float s = 0.0f, e = 0.0f;
  for (int k = 0; k < Do; ++k) {
    const float p = expf(col.at(k, lk) - m);
    s += p;
    e += p * zval[k];
  }
  const long long o = (static_cast<long long>(b) * H + Y) * W + X;
  depth[o] = e / s;
  if (stats) {
    stats[(static_cast<long long>(b) * 2 * H + Y) * W + X] = m;
    stats[((static_cast<long long>(b) * 2 + 1) * H + Y) * W + X] = s;
  }
}

// backward, stage 1: per pixel, t_k = g p_k (z_k - depth) carried back along the depth axis only:
// T[b,d,Y,X] = sum_k [i0(k) == d] l0(k) t_k + [i1(k) == d] l1(k) t_k, accumulated in two rolling registers in plane order
__global__ __launch_bounds__(kBlock) void depth_regress_bwd_planes(const float* __restrict__ cost, const float* __restrict__ zval,
                                                                   const float* __restrict__ depth, const float* __restrict__ stats,
                                                                   const float* __restrict__ gdepth, float* __restrict__ T, int D, int h,
                                                                   int w, int Do, int H, int W, int align) {
  const int X = blockIdx.x * kBlock + threadIdx.x, Y = blockIdx.y, b = blockIdx.z;
  if (X >= W) return;
  Column col;
  col.cost = cost + static_cast<long long>(b) * D * h * w;
  col.D = D, col.h = h, col.w = w, col.Do = Do, col.align = align;
  col.sd = lin_scale(D, Do, align);
  col.px.y = lin_at(Y, h, lin_scale(h, H, align), align);
  col.px.x = lin_at(X, w, lin_scale(w, W, align), align);
  const long long o = (static_cast<long long>(b) * H + Y) * W + X;
  const float g = gdepth[o], dp = depth[o];
  const float m = stats[(static_cast<long long>(b) * 2 * H + Y) * W + X], s = stats[((static_cast<long long>(b) * 2 + 1) * H + Y) * W + X];
  float* Tp = T + (static_cast<long long>(b) * D * H + Y) * W + X;  // + d * H * W
  const long long plane = static_cast<long long>(H) * W;
  Lin lk;
  col.start();
  int at = 0;                  // accA belongs to source plane `at`, accB to `at + 1`
  float accA = 0.0f, accB = 0.0f;
  for (int k = 0; k < Do; ++k) {
    const float c = col.at(k, lk);
    const float t = g * (expf(c - m) / s) * (zval[k] - dp);
    while (at < lk.i0) {
      Tp[at * plane] = accA;
      accA = accB;
      accB = 0.0f;
      ++at;
    }
    accA += lk.l0 * t;
    if (lk.i1 > lk.i0)
      accB += lk.l1 * t;
    else
      accA += lk.l1 * t;
  }
  Tp[at * plane] = accA;
  if (at + 1 < D) Tp[(at + 1) * plane] = accB;
  for (int d = at + 2; d < D; ++d) Tp[d * plane] = 0.0f;
}

// the candidate output indices whose linear taps can touch source index i: a superset, every candidate is tested with lin_at
__device__ __forceinline__ void footprint(int i, int out, float scale, int align, int& lo, int& hi) {
  lo = 0, hi = out - 1;
  if (scale > 0.0f) {  // invert src(dst) at src = i -+ 1, two outputs of slack
    const float a = static_cast<float>(i) - 1.0f, c = static_cast<float>(i) + 1.0f;
    const float dlo = align ? a / scale : (a + 0.5f) / scale - 0.5f, dhi = align ? c / scale : (c + 0.5f) / scale - 0.5f;
    const int l = static_cast<int>(floorf(dlo)) - 2, u = static_cast<int>(ceilf(dhi)) + 2;
    lo = l < 0 ? 0 : l;
    hi = u > out - 1 ? out - 1 : u;
  }
}

__device__ __forceinline__ float tap_weight(const Lin& l, int i) { return (l.i0 == i ? l.l0 : 0.0f) + (l.i1 == i ? l.l1 : 0.0f); }

// backward, stage 2: the adjoint of the bilinear (h, w) up-sampling as a gather, one lane per cost cell, rows then columns
__global__ __launch_bounds__(kBlock) void depth_regress_bwd_cells(const float* __restrict__ T, float* __restrict__ gcost, int D, int h, int w,
                                                                  int H, int W, int align) {
  const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y % h, d = blockIdx.y / h, b = blockIdx.z;
  if (x >= w) return;
  const float sy = lin_scale(h, H, align), sx = lin_scale(w, W, align);
  int ylo, yhi, xlo, xhi;
  footprint(y, H, sy, align, ylo, yhi);
  footprint(x, W, sx, align, xlo, xhi);
  const float* Tp = T + (static_cast<long long>(b) * D + d) * H * W;
  float acc = 0.0f;
  for (int Y = ylo; Y <= yhi; ++Y) {
    const float wy = tap_weight(lin_at(Y, h, sy, align), y);
    if (wy == 0.0f) continue;
    float row = 0.0f;
    for (int X = xlo; X <= xhi; ++X) {
      const float wx = tap_weight(lin_at(X, w, sx, align), x);
      if (wx != 0.0f) row += wx * Tp[static_cast<long long>(Y) * W + X];
    }
    acc += wy * row;
  }
  gcost[((static_cast<long long>(b) * D + d) * h + y) * w + x] = acc;
}

// ---- sigmoid focal loss (maskrcnn-benchmark's SigmoidFocalLoss, the classification term of an FCOS-style head; upstream
//      code reached through RPN3DLoss at attack/DSGN/pgd_attack.py:324).  logits [N,K], targets int32 [N] in 0..K (0 = background,
//      class k matches column k-1); per element  t = 1: -alpha (1-p)^gamma log p;  t = 0: -(1-alpha) p^gamma log(1-p).
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }  // log sigma(x), stable

__global__ __launch_bounds__(kBlock) void focal_fwd_bwd(const float* __restrict__ logits, const int* __restrict__ targets,
                                                        float* __restrict__ loss, float* __restrict__ grad, long long total, int K, float gamma,
                                                        float alpha) {
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int k = static_cast<int>(i % K);
    const int t = targets[i / K];
    const float x = logits[i];
    const float p = 1.0f / (1.0f + expf(-x));
    const float lp = log_sigmoid(x), lq = log_sigmoid(-x);  // log p, log(1 - p)
    const bool pos = t == k + 1, neg = t >= 0 && !pos;       // t < 0: ignored element
    float l = 0.0f, g = 0.0f;
    if (pos) {
      const float f = powf(1.0f - p, gamma);
      l = -alpha * f * lp;
      // d/dx [-(1-p)^g log p] = (1-p)^g (g p log p - (1 - p))
      g = alpha * f * (gamma * p * lp - (1.0f - p));
    } else if (neg) {
      const float f = powf(p, gamma);
      l = -(1.0f - alpha) * f * lq;
      // d/dx [-p^g log(1-p)] = p^g (p - g (1-p) log(1-p))
      g = (1.0f - alpha) * f * (p - gamma * (1.0f - p) * lq);
    }
    if (loss) loss[i] = l;
    if (grad) grad[i] = g;
  }
}


// ---- grid_sample on a 5-D volume: torch.nn.functional.grid_sample(vol [B,C,D,H,W], grid [B,Z,Y,X,3], mode="bilinear",
//      padding_mode="zeros", align_corners) - DSGN's plane-sweep volume -> 3D geometric volume resampling.  The arithmetic is
//      aten's grid_sampler_3d (unnormalise, floor, eight corner weights as products of three differences, corners added in
//      the order tnw tne tsw tse bnw bne bsw bse, out-of-range corners skipped), so the forward is bit-identical to torch on
//      the CPU.  grid[..., 0] indexes W, 1 H, 2 D.
struct Corners {
  int off[8];     // element offset inside one channel of the volume; -1 = outside (contributes nothing)
  float wgt[8];
};

__device__ __forceinline__ float unnormalize(float c, int size, int align) {
  return align ? ((c + 1.0f) / 2.0f) * static_cast<float>(size - 1) : ((c + 1.0f) * static_cast<float>(size) - 1.0f) / 2.0f;
}

__device__ __forceinline__ Corners corners_at(const float* __restrict__ g, int D, int H, int W, int align) {
  const float ix = unnormalize(g[0], W, align), iy = unnormalize(g[1], H, align), iz = unnormalize(g[2], D, align);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float x1 = fx + 1.0f, y1 = fy + 1.0f, z1 = fz + 1.0f;
  // aten: tnw = (ix_bse - ix) * (iy_bse - iy) * (iz_bse - iz), tne = (ix - ix_bsw) * (iy_bsw - iy) * (iz_bsw - iz), ...
  const float ax = x1 - ix, bx = ix - fx, ay = y1 - iy, by = iy - fy, az = z1 - iz, bz = iz - fz;
  Corners c;
  c.wgt[0] = ax * ay * az, c.wgt[1] = bx * ay * az, c.wgt[2] = ax * by * az, c.wgt[3] = bx * by * az;
  c.wgt[4] = ax * ay * bz, c.wgt[5] = bx * ay * bz, c.wgt[6] = ax * by * bz, c.wgt[7] = bx * by * bz;
  // floats far outside the int range (or NaN) are outside the volume whatever they convert to: test in float first
  const bool okx0 = fx >= 0.0f && fx <= static_cast<float>(W - 1), okx1 = x1 >= 0.0f && x1 <= static_cast<float>(W - 1);
  const bool oky0 = fy >= 0.0f && fy <= static_cast<float>(H - 1), oky1 = y1 >= 0.0f && y1 <= static_cast<float>(H - 1);
  const bool okz0 = fz >= 0.0f && fz <= static_cast<float>(D - 1), okz1 = z1 >= 0.0f && z1 <= static_cast<float>(D - 1);
  const int x0 = okx0 ? static_cast<int>(fx) : 0, xx1 = okx1 ? static_cast<int>(x1) : 0;
  const int y0 = oky0 ? static_cast<int>(fy) : 0, yy1 = oky1 ? static_cast<int>(y1) : 0;
  const int z0 = okz0 ? static_cast<int>(fz) : 0, zz1 = okz1 ? static_cast<int>(z1) : 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool ok = ((k & 1) ? okx1 : okx0) && ((k & 2) ? oky1 : oky0) && ((k & 4) ? okz1 : okz0);
    const int xx = (k & 1) ? xx1 : x0, yy = (k & 2) ? yy1 : y0, zz = (k & 4) ? zz1 : z0;
    c.off[k] = ok ? (zz * H + yy) * W + xx : -1;
  }
  return c;
}

// one lane per output voxel (x fastest: coalesced stores); the corners are computed once and reused for every channel
__global__ __launch_bounds__(kBlock) void grid_sample3d_fwd(const float* __restrict__ vol, const float* __restrict__ grid, float* __restrict__ out,
                                                            int C, int D, int H, int W, long long ovol, int align) {
  const long long o = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  const int b = blockIdx.y;
  if (o >= ovol) return;
  const Corners cn = corners_at(grid + (static_cast<long long>(b) * ovol + o) * 3, D, H, W, align);
  const long long ivol = static_cast<long long>(D) * H * W;
  const float* vp = vol + static_cast<long long>(b) * C * ivol;
  float* op = out + static_cast<long long>(b) * C * ovol + o;
  // (pairing the two x-neighbours of a corner pair into one 8-byte load, which helps RoIAlign's forward, measured 0.148 vs 0.140 ms here: dropped)
  for (int c = 0; c < C; ++c, vp += ivol, op += ovol) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (cn.off[k] >= 0) acc += vp[cn.off[k]] * cn.wgt[k];
    __builtin_nontemporal_store(acc, op);
  }
}

// ---- the backward's plan: for every cell of the volume the list of (output voxel, weight) that sample it, sorted by output
//      voxel.  Built once per grid (the grid is a function of the camera calibration only) with integer atomics - counts and
//      slots are order-independent once every list is sorted - and then read by an atomic-free gather.
//      plan = int32 offsets[cells + 1] | int32 scratch[cells + nblocks + 1] | int32 idx[8 * B * ovol] | float wgt[8 * B * ovol]
constexpr int kScanPer = 1024;  // cells per scan block (256 lanes x 4)

struct PlanView {
  int* offsets;
  int* scratch;   // counts, then cursors
  int* sums;      // per scan block
  int* idx;
  float* wgt;
};

__host__ __device__ inline long long plan_blocks(long long cells) { return (cells + kScanPer - 1) / kScanPer; }

__host__ __device__ inline PlanView plan_view(void* plan, long long cells, long long entries) {
  PlanView v;
  v.offsets = reinterpret_cast<int*>(plan);
  v.scratch = v.offsets + cells + 1;
  v.sums = v.scratch + cells;
  v.idx = v.sums + plan_blocks(cells) + 1;
  v.wgt = reinterpret_cast<float*>(v.idx + entries);
  return v;
}

__global__ __launch_bounds__(kBlock) void plan_zero(int* p, long long n) {
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * kBlock) p[i] = 0;
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void plan_count_or_fill(const float* __restrict__ grid, PlanView pv, int D, int H, int W, long long ovol,
                                                             int align) {
  const long long o = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  const int b = blockIdx.y;
  if (o >= ovol) return;
  const Corners cn = corners_at(grid + (static_cast<long long>(b) * ovol + o) * 3, D, H, W, align);
  const long long ivol = static_cast<long long>(D) * H * W;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (cn.off[k] < 0) continue;
    const long long cell = b * ivol + cn.off[k];
    const int slot = atomicAdd(pv.scratch + cell, 1);
    if (FILL) {
      const int pos = pv.offsets[cell] + slot;
      pv.idx[pos] = static_cast<int>(o);
      pv.wgt[pos] = cn.wgt[k];
    }
  }
}

// exclusive scan of scratch[0..cells) into offsets, three small kernels: per-block totals, scan of the totals, write-back
__global__ __launch_bounds__(kBlock) void plan_scan_blocks(PlanView pv, long long cells) {
  __shared__ int part[kBlock];
  const long long base = blockIdx.x * static_cast<long long>(kScanPer) + threadIdx.x * 4;
  int s = 0;
  for (int j = 0; j < 4; ++j)
    if (base + j < cells) s += pv.scratch[base + j];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) pv.sums[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(kBlock) void plan_scan_sums(PlanView pv, long long nblocks) {  // one workgroup, serial over chunks of 256
  __shared__ int part[kBlock];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long long base = 0; base < nblocks; base += kBlock) {
    const long long i = base + threadIdx.x;
    const int v = i < nblocks ? pv.sums[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {  // Hillis-Steele inclusive scan
      const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) pv.sums[i] = carry + part[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == 0) carry += part[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) pv.sums[nblocks] = carry;
}

__global__ __launch_bounds__(kBlock) void plan_scan_write(PlanView pv, long long cells) {
  __shared__ int part[kBlock];
  const long long base = blockIdx.x * static_cast<long long>(kScanPer) + threadIdx.x * 4;
  int v[4], s = 0;
  for (int j = 0; j < 4; ++j) {
    v[j] = base + j < cells ? pv.scratch[base + j] : 0;
    s += v[j];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int run = pv.sums[blockIdx.x] + part[threadIdx.x] - s;
  for (int j = 0; j < 4; ++j) {
    if (base + j < cells) pv.offsets[base + j] = run;
    run += v[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) pv.offsets[cells] = pv.sums[gridDim.x];
}

__global__ __launch_bounds__(kBlock) void plan_sort(PlanView pv, long long cells) {  // insertion sort of each (short) list by output voxel
  const long long cell = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  if (cell >= cells) return;
  const int lo = pv.offsets[cell], hi = pv.offsets[cell + 1];
  for (int i = lo + 1; i < hi; ++i) {
    const int key = pv.idx[i];
    const float w = pv.wgt[i];
    int j = i - 1;
    while (j >= lo && pv.idx[j] > key) {
      pv.idx[j + 1] = pv.idx[j];
      pv.wgt[j + 1] = pv.wgt[j];
      --j;
    }
    pv.idx[j + 1] = key;
    pv.wgt[j + 1] = w;
  }
}

// grad_vol[b,c,cell] = sum over the cell's list, in list order, of grad_out[b,c,voxel] * weight.  One lane per cell (x fastest),
// kGsChan channels at a time so that a list entry is read once per channel block.
constexpr int kGsChan = 16;  // measured at the DSGN size: 8 -> 0.603 ms, 16 -> 0.541 ms, 32 -> 0.657 ms

__global__ __launch_bounds__(kBlock) void grid_sample3d_bwd(const float* __restrict__ gout, PlanView pv, float* __restrict__ gvol, int C,
                                                            long long ivol, long long ovol) {
  const long long cell_in = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x;
  const int b = blockIdx.z, c0 = blockIdx.y * kGsChan;
  if (cell_in >= ivol) return;
  const long long cell = b * ivol + cell_in;
  const int lo = pv.offsets[cell], hi = pv.offsets[cell + 1];
  const float* gp = gout + (static_cast<long long>(b) * C + c0) * ovol;
  float acc[kGsChan];
#pragma unroll
  for (int j = 0; j < kGsChan; ++j) acc[j] = 0.0f;
  for (int e = lo; e < hi; ++e) {
    const int o = pv.idx[e];
    const float w = pv.wgt[e];
#pragma unroll
    for (int j = 0; j < kGsChan; ++j)
      if (c0 + j < C) acc[j] += gp[j * ovol + o] * w;
  }
  float* op = gvol + (static_cast<long long>(b) * C + c0) * ivol + cell_in;
#pragma unroll
  for (int j = 0; j < kGsChan; ++j)
    if (c0 + j < C) __builtin_nontemporal_store(acc[j], op + j * ivol);
}

// The same gather with the gradient volume re-laid channels-last first.  In NCDHW every (list entry, channel) gather above pulls its own
// cache line - PMC: 3.03 GB fetched per launch for a 149 MB volume.  With [B, voxel, C] one list entry is ONE run of C consecutive floats:
//   1. gs_to_channels_last: [B,C,O] -> [B,O,C] through a 32 x 64 LDS tile (both sides coalesced);
//   2. grid_sample3d_bwd_cl: a half-wave (32 lanes = 32 channels) owns a cell at a time, all its lanes read the same (voxel, weight)
//      entry (one broadcast transaction) and then the 128-byte run of that voxel; a workgroup covers 64 consecutive cells and turns
//      its [32][64] results through LDS so that the NCDHW stores are 256-byte runs per channel.
// Per (cell, channel) the sum still runs over the list in list order: the same bits as the kernel above.
constexpr int kClCells = 64;   // cells per workgroup (8 half-waves x 8 cells)

__global__ __launch_bounds__(kBlock) void gs_to_channels_last(const float* __restrict__ src, float* __restrict__ dst, int C, long long O) {
  __shared__ float tile[32][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 32;
  const long long o0 = blockIdx.x * 64LL;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 voxels x 4 channel rows per pass
  const float* sp = src + static_cast<long long>(b) * C * O;
#pragma unroll
  for (int r = ty; r < 32; r += 4)
    tile[r][tx] = (c0 + r < C && o0 + tx < O) ? sp[(c0 + r) * O + o0 + tx] : 0.0f;
  __syncthreads();
  float* dp = dst + static_cast<long long>(b) * O * C;
  const int cx = threadIdx.x & 31, oy = threadIdx.x >> 5;   // 32 channels x 8 voxels per pass
#pragma unroll
  for (int r = oy; r < 64; r += 8)
    if (c0 + cx < C && o0 + r < O) dp[(o0 + r) * C + c0 + cx] = tile[cx][r];
}

__global__ __launch_bounds__(kBlock) void grid_sample3d_bwd_cl(const float* __restrict__ gcl, PlanView pv, float* __restrict__ gvol, int C,
                                                               long long ivol, long long ovol) {
  __shared__ float res[32][kClCells + 1];
  const int b = blockIdx.z, c0 = blockIdx.y * 32;
  const long long cell0 = blockIdx.x * static_cast<long long>(kClCells);
  const int lane_c = threadIdx.x & 31, hw = threadIdx.x >> 5;   // channel, half-wave 0..7
  const bool live_c = c0 + lane_c < C;
  const float* gp = gcl + static_cast<long long>(b) * ovol * C + c0 + lane_c;
#pragma unroll 1
  for (int k = 0; k < kClCells / 8; ++k) {
    const int cl = hw * (kClCells / 8) + k;
    const long long cell_in = cell0 + cl;
    float acc = 0.0f;
    if (cell_in < ivol) {
      const long long cell = b * ivol + cell_in;
      const int lo = pv.offsets[cell], hi = pv.offsets[cell + 1];
      for (int e = lo; e < hi; e += 4) {   // four entries in flight: the (voxel, weight) loads and the four runs are independent
        int o[4];
        float w[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = e + j < hi;
          o[j] = pv.idx[in ? e + j : lo];   // the same address in all 32 lanes: one broadcast transaction
          w[j] = pv.wgt[in ? e + j : lo];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = live_c ? gp[static_cast<long long>(o[j]) * C] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (e + j < hi) acc += v[j] * w[j];   // in list order
      }
    }
    res[lane_c][cl] = acc;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 cells x 4 channels per pass
  float* op = gvol + (static_cast<long long>(b) * C + c0) * ivol + cell0;
#pragma unroll
  for (int r = ty; r < 32; r += 4)
    if (c0 + r < C && cell0 + tx < ivol) __builtin_nontemporal_store(res[r][tx], op + r * ivol + tx);
}

// ---- the backward of a fused ReLU: out = y > 0 ? grad : 0 (torch's threshold_backward), one pass instead of a compare and a
//      multiply; the convolutions' autograd wrappers apply it to the incoming gradient before the adjoint convolution.
__global__ __launch_bounds__(kBlock) void relu_backward_kernel(const float* __restrict__ grad, const float* __restrict__ y, float* __restrict__ out,
                                                               long long n4, long long n) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const long long stride = static_cast<long long>(gridDim.x) * kBlock;
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < n4; i += stride) {
    const v4 g = __builtin_nontemporal_load(reinterpret_cast<const v4*>(grad) + i);
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(y) + i);
    v4 o;
    o.x = v.x > 0.0f ? g.x : 0.0f, o.y = v.y > 0.0f ? g.y : 0.0f, o.z = v.z > 0.0f ? g.z : 0.0f, o.w = v.w > 0.0f ? g.w : 0.0f;
    __builtin_nontemporal_store(o, reinterpret_cast<v4*>(out) + i);
  }
  for (long long i = 4 * n4 + blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < n; i += stride) out[i] = y[i] > 0.0f ? grad[i] : 0.0f;
}

// ---- the ResNet stem's tail in one pass each way: y = maxpool(3x3, stride 2, padding 1)(relu(t + bias)) straight from the convolution's
//      output t, and the gradient w.r.t. t straight from the gradient w.r.t. y.  torch runs a bias pass, a ReLU pass, the pooling, and
//      backward the pooling's gather (0.25 ms on [2,64,300,994]) and threshold_backward - five passes over a 305 MB tensor per step
//      (profiles/r05_r101_small_ops.json).  max(relu(v)) = relu(max(v)) and rounding is monotone, so the value is torch's; the window is
//      scanned row-major and a later element replaces the maximum only if it is greater (or NaN): torch's argmax.  The code byte is the
//      argmax's position in the window (0..8), or kNoGrad where the maximum is <= 0: relu's backward passes nothing there, so the backward
//      needs neither t nor y.  A pixel's gradient sums the outputs that chose it in (oy, ox) order, as torch's gather does.
constexpr unsigned char kNoGrad = 15;

__global__ __launch_bounds__(kBlock) void stem_pool_fwd(const float* __restrict__ t, const float* __restrict__ bias, float* __restrict__ y,
                                                        unsigned char* __restrict__ code, int C, int H, int W, int OH, int OW, long long total) {
  const long long stride = static_cast<long long>(gridDim.x) * kBlock;
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += stride) {
    const int ox = static_cast<int>(i % OW), oy = static_cast<int>((i / OW) % OH);
    const long long plane = i / (static_cast<long long>(OW) * OH);
    const float b = bias ? bias[plane % C] : 0.0f;
    const float* tp = t + plane * H * W;
    float m = -__builtin_inff();
    int arg = -1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const float v = tp[static_cast<long long>(iy) * W + ix] + b;
        if (arg < 0 || v > m || v != v) m = v, arg = ky * 3 + kx;
      }
    }
    const bool pass = m > 0.0f || m != m;
    y[i] = pass ? m : 0.0f;
    code[i] = pass ? static_cast<unsigned char>(arg) : kNoGrad;
  }
}

__global__ __launch_bounds__(kBlock) void stem_pool_bwd(const float* __restrict__ gy, const unsigned char* __restrict__ code, float* __restrict__ gt,
                                                        int H, int W, int OH, int OW, long long total) {
  const long long stride = static_cast<long long>(gridDim.x) * kBlock;
  for (long long i = blockIdx.x * static_cast<long long>(kBlock) + threadIdx.x; i < total; i += stride) {
    const int x = static_cast<int>(i % W), yy = static_cast<int>((i / W) % H);
    const long long plane = i / (static_cast<long long>(W) * H);
    const float* gp = gy + plane * OH * OW;
    const unsigned char* cp = code + plane * OH * OW;
    float acc = 0.0f;
    const int oy_hi = min((yy + 1) / 2, OH - 1), ox_hi = min((x + 1) / 2, OW - 1);
    for (int oy = yy / 2; oy <= oy_hi; ++oy)
      for (int ox = x / 2; ox <= ox_hi; ++ox) {
        const int k = (yy - (2 * oy - 1)) * 3 + (x - (2 * ox - 1));
        if (cp[oy * OW + ox] == k) acc = acc + gp[oy * OW + ox];
      }
    __builtin_nontemporal_store(acc, gt + i);
  }
}

inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

}  // namespace

// ---- bird's-eye-view fold: the 3D geometric volume [B,C,Z,Y,X] -> the 2D head's input [B, C * Y/P, Z, X]: the height is average-pooled by P
// and what remains of it folded into the channels (DSGN: F.avg_pool3d(v, (1, P, 1)) -> permute(0, 1, 3, 2, 4) -> reshape).  torch makes
// that a pooling kernel plus a permuting copy (and two more passes backward); it is one HBM-bound pass each way.  out[b, c*Yp + yy, z, x]
// = ((v[.., P*yy, x] + v[.., P*yy + 1, x]) + ...) / P in that order; rows past Yp * P (floor, as avg_pool3d) are dropped.
__global__ __launch_bounds__(kBlock) void bev_fold_fwd(const float* __restrict__ v, float* __restrict__ out, int C, int Z, int Y, int X, int P, int Yp,
                                                       long long total) {
  for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int x = static_cast<int>(i % X);
    long long r = i / X;
    const int z = static_cast<int>(r % Z);
    r /= Z;
    const int yy = static_cast<int>(r % Yp);
    r /= Yp;                                          // r = b * C + c
    const float* src = v + ((r * Z + z) * Y + static_cast<long long>(P) * yy) * X + x;
    float acc = __builtin_nontemporal_load(src);
    for (int k = 1; k < P; ++k) acc = acc + __builtin_nontemporal_load(src + static_cast<long long>(k) * X);
    out[i] = acc / static_cast<float>(P);
  }
}

// mask (the forward's input v, a ReLU output whose only consumer is this fold; or null): the gradient is zeroed where v <= 0 - the producer's
// ReLU backward without a pass of its own over the volume
__global__ __launch_bounds__(kBlock) void bev_fold_bwd(const float* __restrict__ gout, const float* __restrict__ mask, float* __restrict__ gv, int C, int Z,
                                                       int Y, int X, int P, int Yp, long long total) {
  for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int x = static_cast<int>(i % X);
    long long r = i / X;
    const int y = static_cast<int>(r % Y);
    r /= Y;
    const int z = static_cast<int>(r % Z);
    r /= Z;                                           // r = b * C + c
    const int yy = y / P;
    float g = 0.0f;
    if (yy < Yp) g = gout[((r * Yp + yy) * Z + z) * X + x] / static_cast<float>(P);
    if (mask != nullptr && !(__builtin_nontemporal_load(mask + i) > 0.0f)) g = 0.0f;
    __builtin_nontemporal_store(g, gv + i);
  }
}

// the same, four consecutive x per lane (X % 4 == 0, 16-byte aligned tensors): 16-byte loads and stores - the scalar version moved 0.67 GB in
// 0.31 ms on the [1,64,192,20,304] volume (2.1 TB/s, profiles/r05_dsgn_small_ops.json)
__global__ __launch_bounds__(kBlock) void bev_fold_bwd_vec4(const float* __restrict__ gout, const float* __restrict__ mask, float* __restrict__ gv, int C,
                                                            int Z, int Y, int X4, int P, int Yp, long long total4) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < total4; i += static_cast<long long>(gridDim.x) * kBlock) {
    const int x4 = static_cast<int>(i % X4);
    long long r = i / X4;
    const int y = static_cast<int>(r % Y);
    r /= Y;
    const int z = static_cast<int>(r % Z);
    r /= Z;                                           // r = b * C + c
    const int yy = y / P;
    v4 g = {0.0f, 0.0f, 0.0f, 0.0f};
    if (yy < Yp) {
      g = *(reinterpret_cast<const v4*>(gout) + ((r * Yp + yy) * Z + z) * X4 + x4);
      const float p = static_cast<float>(P);
      g.x = g.x / p, g.y = g.y / p, g.z = g.z / p, g.w = g.w / p;
    }
    if (mask != nullptr) {
      const v4 m = __builtin_nontemporal_load(reinterpret_cast<const v4*>(mask) + i);
      g.x = m.x > 0.0f ? g.x : 0.0f, g.y = m.y > 0.0f ? g.y : 0.0f, g.z = m.z > 0.0f ? g.z : 0.0f, g.w = m.w > 0.0f ? g.w : 0.0f;
    }
    __builtin_nontemporal_store(g, reinterpret_cast<v4*>(gv) + i);
  }
}

extern "C" {

int adv_depth_regress_f32(const float* cost, const float* depth_values, float* depth_out, float* stats_out, int b, int d, int h, int w,
                          int d_out, int h_out, int w_out, int align_corners, adv_stream_t stream) {
  if (!cost || !depth_values || !depth_out || b < 1 || d < 1 || h < 1 || w < 1 || d_out < 1 || h_out < 1 || w_out < 1) return ADV_EINVAL;
  if (h_out > 65535 || b > 65535) return ADV_EINVAL;
  if (!aligned4(cost) || !aligned4(depth_values) || !aligned4(depth_out) || !aligned4(stats_out)) return ADV_EALIGN;
  hipLaunchKernelGGL(depth_regress_fwd, dim3((w_out + kBlock - 1) / kBlock, h_out, b), dim3(kBlock), 0, static_cast<hipStream_t>(stream), cost,
                     depth_values, depth_out, stats_out, d, h, w, d_out, h_out, w_out, align_corners ? 1 : 0);
  return adv_internal_finish_launch();
}

int adv_depth_regress_bwd_f32(const float* cost, const float* depth_values, const float* depth, const float* stats, const float* grad_depth,
                              float* workspace, float* grad_cost, int b, int d, int h, int w, int d_out, int h_out, int w_out,
                              int align_corners, adv_stream_t stream) {
  if (!cost || !depth_values || !depth || !stats || !grad_depth || !workspace || !grad_cost) return ADV_EINVAL;
  if (b < 1 || d < 1 || h < 1 || w < 1 || d_out < 1 || h_out < 1 || w_out < 1) return ADV_EINVAL;
  if (h_out > 65535 || b > 65535 || static_cast<long long>(d) * h > 65535) return ADV_EINVAL;
  if (!aligned4(cost) || !aligned4(depth) || !aligned4(stats) || !aligned4(grad_depth) || !aligned4(workspace) || !aligned4(grad_cost)) return ADV_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int al = align_corners ? 1 : 0;
  hipLaunchKernelGGL(depth_regress_bwd_planes, dim3((w_out + kBlock - 1) / kBlock, h_out, b), dim3(kBlock), 0, st, cost, depth_values, depth, stats,
                     grad_depth, workspace, d, h, w, d_out, h_out, w_out, al);
  hipLaunchKernelGGL(depth_regress_bwd_cells, dim3((w + kBlock - 1) / kBlock, d * h, b), dim3(kBlock), 0, st, workspace, grad_cost, d, h, w, h_out,
                     w_out, al);
  return adv_internal_finish_launch();
}

int adv_sigmoid_focal_loss_f32(const float* logits, const int32_t* targets, float* loss_out, float* grad_out, int64_t n, int k, float gamma,
                               float alpha, adv_stream_t stream) {
  if (!logits || !targets || (!loss_out && !grad_out) || n < 0 || k < 1) return ADV_EINVAL;
  if (n == 0) return ADV_OK;
  if (!aligned4(logits) || !aligned4(targets) || !aligned4(loss_out) || !aligned4(grad_out)) return ADV_EALIGN;
  const long long total = static_cast<long long>(n) * k;
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  hipLaunchKernelGGL(focal_fwd_bwd, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), logits,
                     reinterpret_cast<const int*>(targets), loss_out, grad_out, total, k, gamma, alpha);
  return adv_internal_finish_launch();
}

int adv_grid_sample3d_f32(const float* vol, const float* grid, float* out, int b, int c, int d, int h, int w, int zo, int yo, int xo,
                          int align_corners, adv_stream_t stream) {
  if (!vol || !grid || !out || b < 1 || c < 1 || d < 1 || h < 1 || w < 1 || zo < 1 || yo < 1 || xo < 1) return ADV_EINVAL;
  const long long ovol = static_cast<long long>(zo) * yo * xo, ivol = static_cast<long long>(d) * h * w;
  if (b > 65535 || ivol >= (1LL << 31) || ovol >= (1LL << 31)) return ADV_EINVAL;
  if (!aligned4(vol) || !aligned4(grid) || !aligned4(out)) return ADV_EALIGN;
  hipLaunchKernelGGL(grid_sample3d_fwd, dim3(static_cast<unsigned>((ovol + kBlock - 1) / kBlock), b), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     vol, grid, out, c, d, h, w, ovol, align_corners ? 1 : 0);
  return adv_internal_finish_launch();
}

int64_t adv_grid_sample3d_plan_bytes(int b, int d, int h, int w, int zo, int yo, int xo) {
  if (b < 1 || d < 1 || h < 1 || w < 1 || zo < 1 || yo < 1 || xo < 1) return 0;
  const long long cells = static_cast<long long>(b) * d * h * w, entries = 8LL * b * zo * yo * xo;
  if (entries >= (1LL << 31) || cells >= (1LL << 31)) return 0;
  return static_cast<int64_t>(4) * ((cells + 1) + cells + (plan_blocks(cells) + 1) + 2 * entries);
}

int adv_grid_sample3d_plan_f32(const float* grid, void* plan, int b, int d, int h, int w, int zo, int yo, int xo, int align_corners,
                               adv_stream_t stream) {
  if (!grid || !plan || adv_grid_sample3d_plan_bytes(b, d, h, w, zo, yo, xo) == 0 || b > 65535) return ADV_EINVAL;
  if (!aligned4(grid) || !aligned4(plan)) return ADV_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long ivol = static_cast<long long>(d) * h * w, ovol = static_cast<long long>(zo) * yo * xo;
  const long long cells = b * ivol, entries = 8 * b * ovol, nblocks = plan_blocks(cells);
  const PlanView pv = plan_view(plan, cells, entries);
  const int al = align_corners ? 1 : 0;
  const dim3 og(static_cast<unsigned>((ovol + kBlock - 1) / kBlock), b);
  const unsigned zb = static_cast<unsigned>((cells + kBlock - 1) / kBlock > 65535 * 8 ? 65535 * 8 : (cells + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(plan_zero, dim3(zb), dim3(kBlock), 0, st, pv.scratch, cells);
  hipLaunchKernelGGL(plan_count_or_fill<false>, og, dim3(kBlock), 0, st, grid, pv, d, h, w, ovol, al);
  hipLaunchKernelGGL(plan_scan_blocks, dim3(static_cast<unsigned>(nblocks)), dim3(kBlock), 0, st, pv, cells);
  hipLaunchKernelGGL(plan_scan_sums, dim3(1), dim3(kBlock), 0, st, pv, nblocks);
  hipLaunchKernelGGL(plan_scan_write, dim3(static_cast<unsigned>(nblocks)), dim3(kBlock), 0, st, pv, cells);
  hipLaunchKernelGGL(plan_zero, dim3(zb), dim3(kBlock), 0, st, pv.scratch, cells);
  hipLaunchKernelGGL(plan_count_or_fill<true>, og, dim3(kBlock), 0, st, grid, pv, d, h, w, ovol, al);
  hipLaunchKernelGGL(plan_sort, dim3(static_cast<unsigned>((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, pv, cells);
  return adv_internal_finish_launch();
}

int adv_grid_sample3d_bwd_f32(const float* grad_out, const void* plan, float* grad_vol, int b, int c, int d, int h, int w, int zo, int yo,
                              int xo, adv_stream_t stream) {
  if (!grad_out || !plan || !grad_vol || c < 1 || adv_grid_sample3d_plan_bytes(b, d, h, w, zo, yo, xo) == 0) return ADV_EINVAL;
  if (b > 65535 || (c + kGsChan - 1) / kGsChan > 65535) return ADV_EINVAL;
  if (!aligned4(grad_out) || !aligned4(plan) || !aligned4(grad_vol)) return ADV_EALIGN;
  const long long ivol = static_cast<long long>(d) * h * w, ovol = static_cast<long long>(zo) * yo * xo;
  const PlanView pv = plan_view(const_cast<void*>(plan), b * ivol, 8 * b * ovol);
  hipLaunchKernelGGL(grid_sample3d_bwd, dim3(static_cast<unsigned>((ivol + kBlock - 1) / kBlock), (c + kGsChan - 1) / kGsChan, b), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), grad_out, pv, grad_vol, c, ivol, ovol);
  return adv_internal_finish_launch();
}

int64_t adv_grid_sample3d_bwd_workspace_floats(int b, int c, int zo, int yo, int xo) {
  if (b < 1 || c < 1 || zo < 1 || yo < 1 || xo < 1) return 0;
  return static_cast<int64_t>(b) * c * zo * yo * xo;
}

int adv_grid_sample3d_bwd_ws_f32(const float* grad_out, const void* plan, float* grad_vol, float* workspace, int b, int c, int d, int h, int w,
                                 int zo, int yo, int xo, adv_stream_t stream) {
  if (workspace == nullptr) return adv_grid_sample3d_bwd_f32(grad_out, plan, grad_vol, b, c, d, h, w, zo, yo, xo, stream);
  if (!grad_out || !plan || !grad_vol || c < 1 || adv_grid_sample3d_plan_bytes(b, d, h, w, zo, yo, xo) == 0) return ADV_EINVAL;
  if (b > 65535 || (c + 31) / 32 > 65535) return ADV_EINVAL;
  if (!aligned4(grad_out) || !aligned4(plan) || !aligned4(grad_vol) || !aligned4(workspace)) return ADV_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long ivol = static_cast<long long>(d) * h * w, ovol = static_cast<long long>(zo) * yo * xo;
  const PlanView pv = plan_view(const_cast<void*>(plan), b * ivol, 8 * b * ovol);
  const unsigned cg = static_cast<unsigned>((c + 31) / 32);
  hipLaunchKernelGGL(gs_to_channels_last, dim3(static_cast<unsigned>((ovol + 63) / 64), cg, b), dim3(kBlock), 0, st, grad_out, workspace, c, ovol);
  hipLaunchKernelGGL(grid_sample3d_bwd_cl, dim3(static_cast<unsigned>((ivol + kClCells - 1) / kClCells), cg, b), dim3(kBlock), 0, st, workspace, pv,
                     grad_vol, c, ivol, ovol);
  return adv_internal_finish_launch();
}

int adv_bev_fold_f32(const float* v, float* out, int b, int c, int z, int y, int x, int pool, adv_stream_t stream) {
  if (!v || !out || v == out || b < 1 || c < 1 || z < 1 || y < 1 || x < 1 || pool < 1 || pool > y) return ADV_EINVAL;
  if (!aligned4(v) || !aligned4(out)) return ADV_EALIGN;
  const int yp = y / pool;
  const long long total = static_cast<long long>(b) * c * yp * z * x;
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  hipLaunchKernelGGL(bev_fold_fwd, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), v, out, c, z, y, x, pool, yp, total);
  return adv_internal_finish_launch();
}

int adv_bev_fold_bwd_f32(const float* grad_out, const float* mask, float* grad_v, int b, int c, int z, int y, int x, int pool, adv_stream_t stream) {
  if (!grad_out || !grad_v || grad_out == grad_v || mask == grad_v || b < 1 || c < 1 || z < 1 || y < 1 || x < 1 || pool < 1 || pool > y) return ADV_EINVAL;
  if (!aligned4(grad_out) || !aligned4(grad_v) || !aligned4(mask)) return ADV_EALIGN;
  const long long total = static_cast<long long>(b) * c * z * y * x;
  if (x % 4 == 0 && ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_v) | reinterpret_cast<uintptr_t>(mask)) & 15u) == 0 &&
      !adv_hook("ADV_BEV_SCALAR")) {
    long long blocks4 = (total / 4 + kBlock - 1) / kBlock;
    if (blocks4 > 65535LL * 16) blocks4 = 65535LL * 16;
    hipLaunchKernelGGL(bev_fold_bwd_vec4, dim3(static_cast<unsigned>(blocks4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), grad_out, mask, grad_v, c, z,
                       y, x / 4, pool, y / pool, total / 4);
    return adv_internal_finish_launch();
  }
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  hipLaunchKernelGGL(bev_fold_bwd, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), grad_out, mask, grad_v, c, z, y, x,
                     pool, y / pool, total);
  return adv_internal_finish_launch();
}

int adv_stem_pool_fwd_f32(const float* t, const float* bias, float* y, uint8_t* code, int64_t planes, int c, int h, int w, adv_stream_t stream) {
  if (!t || !y || !code || planes < 0 || c < 1 || h < 1 || w < 1 || static_cast<const void*>(t) == static_cast<const void*>(y)) return ADV_EINVAL;
  if (planes == 0) return ADV_OK;
  if (!aligned4(t) || !aligned4(y) || (bias && !aligned4(bias))) return ADV_EALIGN;
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  const long long total = static_cast<long long>(planes) * oh * ow;
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  hipLaunchKernelGGL(stem_pool_fwd, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), t, bias, y, code, c, h, w, oh, ow,
                     total);
  return adv_internal_finish_launch();
}

int adv_stem_pool_bwd_f32(const float* grad_y, const uint8_t* code, float* grad_t, int64_t planes, int h, int w, adv_stream_t stream) {
  if (!grad_y || !code || !grad_t || planes < 0 || h < 1 || w < 1 || grad_y == grad_t) return ADV_EINVAL;
  if (planes == 0) return ADV_OK;
  if (!aligned4(grad_y) || !aligned4(grad_t)) return ADV_EALIGN;
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  const long long total = static_cast<long long>(planes) * h * w;
  long long blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  hipLaunchKernelGGL(stem_pool_bwd, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), grad_y, code, grad_t, h, w, oh, ow,
                     total);
  return adv_internal_finish_launch();
}

int adv_relu_backward_f32(const float* grad, const float* y, float* out, int64_t n, adv_stream_t stream) {
  if (!grad || !y || !out || n < 0) return ADV_EINVAL;
  if (n == 0) return ADV_OK;
  if (!aligned4(grad) || !aligned4(y) || !aligned4(out)) return ADV_EALIGN;
  const bool vec = ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  const long long n4 = vec ? n / 4 : 0;
  long long blocks = ((vec ? n4 : n) + kBlock - 1) / kBlock;
  if (blocks > 65535LL * 16) blocks = 65535LL * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(relu_backward_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), grad, y, out, n4,
                     static_cast<long long>(n));
  return adv_internal_finish_launch();
}

}  // extern "C"
