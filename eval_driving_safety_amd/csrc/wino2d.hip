// 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) with the sixteen element-wise products on the gfx950 float32 matrix
// cores (v_mfma_f32_16x16x4_f32) - 2.25x fewer multiply-adds than the direct kernel of conv2d.hip, everything in ONE kernel (no
// transformed tensor ever reaches HBM).
//
//   Y(2x2) = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      d_c: the 4x4 input patch of channel c around the 2x2 output block
//
// A 512-thread workgroup owns 8 rows x 32 columns of the output (4 x 16 = 64 patches) x 64 output channels.  Per stage of 8 input
// channels:
//   1. the input tile [8][10][40] and the transformed weights [16][8][64] go global -> registers -> LDS while the previous stage
//      computes (two buffers; a lane's fetch plan is computed once per tile, as in the direct kernel);
//   2. every thread transforms one (channel, patch): 16 LDS reads, 32 additions, 16 LDS writes into V[16][8][64 patches];
//   3. for each of the 16 transform positions k, a [64 ch_out x 8 ch_in] x [8 ch_in x 64 patches] product: wave w owns channel block
//      w & 3 (16 channels) and patch rows 2 (w >> 2), 2 (w >> 2) + 1 (two blocks of 16 patches): 2 x 16 accumulators of 4 registers.
//      A operand = U_k[co = lane & 15][c = lane >> 4], B operand = V_k[c = lane >> 4][patch = lane & 15]; rows of U and V are padded
//      to 80 floats so that the two 16-lane groups a ds_read serves per cycle fall into different banks.
//   All 16 values M_k of one (channel, patch) end up in the SAME lane and register index of the 16 accumulators, so the output
//   transform A^T M A is plain per-lane arithmetic; the epilogue (+ bias, + residual, ReLU, mask) follows it before the store.
//
// Order of operations (oracle/oracle.c orc_conv2d_wino restates it bit for bit): the transforms' additions as written below, the
// accumulation of each M_k one fmaf per input channel in ascending order starting from 0 (the matrix instruction is a k-ordered fmaf
// chain).  The backward w.r.t. the input is the same kernel on the transposed, flipped weights.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kWC = 8;       // input channels per stage
constexpr int kWCO = 64;     // output channels per workgroup
constexpr int kWStr = 80;    // LDS row of U (64 channels) and V (64 patches), padded
constexpr int kWRows = 10, kWLW = 40;
constexpr int kSXw = kWC * kWRows * kWLW;          // 3200 floats
constexpr int kSWw = 16 * kWC * kWStr;             // 10240 floats
constexpr int kXNw = kWC * kWRows * (kWLW / 4);    // 800 float4 per stage
constexpr int kWNw = 16 * kWC * (kWCO / 4);        // 2048 float4 per stage
constexpr int kXSl = (kXNw + 511) / 512, kWSl = kWNw / 512;

struct EpiW {
  const float* bias;
  const float* residual;
  const float* mask;
  int relu;
};

__global__ __launch_bounds__(512, 1) void conv2d_3x3_wino(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int Cin,
                                                         int Cout, int cinpad, int copad, int H, int W, int tiles_w, EpiW epi) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const sv = lds + 2 * (kSXw + kSWw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int cob = wave & 3, hf = wave >> 2;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 32, h0 = ht * 8, co0 = blockIdx.y * kWCO;
  const long long b = blockIdx.z;
  const long long HW = static_cast<long long>(H) * W;
  const float* xb = x + b * Cin * HW;

  v4f acc[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

  long long xoff[kXSl];
  int xkind[kXSl], xc[kXSl];
#pragma unroll
  for (int i = 0; i < kXSl; ++i) {
    const int sidx = tid + 512 * i;
    const int j = sidx % (kWLW / 4), r = (sidx / (kWLW / 4)) % kWRows, c = sidx / ((kWLW / 4) * kWRows);
    const int gh = h0 - 1 + r, gw = w0 - 4 + 4 * j;
    xkind[i] = 0, xoff[i] = 0, xc[i] = c;
    if (sidx < kXNw && gh >= 0 && gh < H && gw + 3 >= 0 && gw < W) {
      xoff[i] = static_cast<long long>(c) * HW + static_cast<long long>(gh) * W + gw;
      xkind[i] = (gw >= 0 && gw + 3 < W) ? 1 : 2;
    }
  }
  v4f rx[kXSl], rw[kWSl];
  auto fetch_x = [&](int c0) {
#pragma unroll
    for (int i = 0; i < kXSl; ++i) {
      v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (xkind[i] != 0 && c0 + xc[i] < Cin) {
        const float* src = xb + static_cast<long long>(c0) * HW + xoff[i];
        if (xkind[i] == 1) {
          v = *reinterpret_cast<const v4f_u*>(src);
        } else {
          const int gw = w0 - 4 + 4 * ((tid + 512 * i) % (kWLW / 4));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gw + e >= 0 && gw + e < W) v[e] = src[e];
        }
      }
      rx[i] = v;
    }
  };
  auto fetch_w = [&](int c0) {
#pragma unroll
    for (int i = 0; i < kWSl; ++i) {
      const int sidx = tid + 512 * i;
      const int q = sidx % (kWCO / 4), row = sidx / (kWCO / 4);      // row = k * 8 + c
      const int k = row / kWC, c = row % kWC;
      rw[i] = *reinterpret_cast<const v4f*>(wp + (static_cast<long long>(k) * cinpad + c0 + c) * copad + co0 + 4 * q);
    }
  };
  auto commit = [&](int buf) {
    float* sx = lds + buf * (kSXw + kSWw);
    float* sw = sx + kSXw;
#pragma unroll
    for (int i = 0; i < kXSl; ++i) {
      const int sidx = tid + 512 * i;
      if (sidx < kXNw) *reinterpret_cast<v4f*>(sx + 4 * sidx) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < kWSl; ++i) {
      const int sidx = tid + 512 * i;
      const int q = sidx % (kWCO / 4), row = sidx / (kWCO / 4);
      *reinterpret_cast<v4f*>(sw + row * kWStr + 4 * q) = rw[i];
    }
  };

  // the transform's (channel, patch) of this thread
  const int tc = tid >> 6, tp = tid & 63, tpr = tp >> 4, tpc = tp & 15;

  const int nstage = (Cin + kWC - 1) / kWC;
  fetch_x(0);
  fetch_w(0);
  commit(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const bool more = st + 1 < nstage;
    if (more) fetch_x((st + 1) * kWC);
    const float* sx = lds + (st & 1) * (kSXw + kSWw);
    const float* sw = sx + kSXw;
    {   // V = B^T d B
      const float* dp = sx + (tc * kWRows + 2 * tpr) * kWLW + 3 + 2 * tpc;
      float d[4][4], t[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = dp[i * kWLW + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
      }
      float* vp = sv + tc * kWStr + tp;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vp[(i * 4 + 0) * kWC * kWStr] = t[i][0] - t[i][2];
        vp[(i * 4 + 1) * kWC * kWStr] = t[i][1] + t[i][2];
        vp[(i * 4 + 2) * kWC * kWStr] = t[i][2] - t[i][1];
        vp[(i * 4 + 3) * kWC * kWStr] = t[i][1] - t[i][3];
      }
    }
    __syncthreads();
    if (more) fetch_w((st + 1) * kWC);
    {   // 32 steps (channel group cs = t >> 4, transform position k = t & 15), two matrix instructions each; the operands of step t + 4
        // are read from LDS before step t issues, and the scheduler may not move anything across a step (it would otherwise hoist all 96
        // reads and spill, or - not unrolled - reuse three registers and expose the LDS latency every four instructions)
      constexpr int kAhead = 4;
      float ra[32], rb0[32], rb1[32];
      const float* ap = sw + k4 * kWStr + cob * 16 + i16;
      const float* bp = sv + k4 * kWStr + (2 * hf) * 16 + i16;
      auto load = [&](int t) {
        const int row = ((t & 15) * kWC + (t >> 4) * 4) * kWStr;
        ra[t] = ap[row];
        rb0[t] = bp[row];
        rb1[t] = bp[row + 16];
      };
#pragma unroll
      for (int t = 0; t < kAhead; ++t) load(t);
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        if (t + kAhead < 32) load(t + kAhead);
        acc[0][t & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], rb0[t], acc[0][t & 15], 0, 0, 0);
        acc[1][t & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], rb1[t], acc[1][t & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) commit((st + 1) & 1);
    __syncthreads();
  }

  const long long MP = static_cast<long long>(Cout) * HW;
  float* yb = y + b * MP;
  const float* resb = epi.residual ? epi.residual + b * MP : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP : nullptr;
  const int gw = w0 + 2 * i16;
  if (gw >= W) return;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int gh = h0 + 2 * (2 * hf + blk);
    if (gh >= H) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int co = co0 + cob * 16 + 4 * k4 + reg;
      if (co >= Cout) continue;
      float s[2][4], o[2][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = (acc[blk][j][reg] + acc[blk][4 + j][reg]) + acc[blk][8 + j][reg];
        s[1][j] = (acc[blk][4 + j][reg] - acc[blk][8 + j][reg]) - acc[blk][12 + j][reg];
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        o[r][0] = (s[r][0] + s[r][1]) + s[r][2];
        o[r][1] = (s[r][1] - s[r][2]) - s[r][3];
      }
      const float bv = epi.bias ? epi.bias[co] : 0.0f;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (gh + r >= H) continue;
        const long long at = static_cast<long long>(co) * HW + static_cast<long long>(gh + r) * W + gw;
        const bool two = gw + 1 < W;
        float v0 = o[r][0], v1 = o[r][1];
        if (epi.bias) v0 = v0 + bv, v1 = v1 + bv;
        if (resb) {
          v0 = v0 + __builtin_nontemporal_load(resb + at);
          if (two) v1 = v1 + __builtin_nontemporal_load(resb + at + 1);
        }
        if (epi.relu) v0 = v0 > 0.0f ? v0 : 0.0f, v1 = v1 > 0.0f ? v1 : 0.0f;
        if (maskb) {
          v0 = __builtin_nontemporal_load(maskb + at) > 0.0f ? v0 : 0.0f;
          if (two) v1 = __builtin_nontemporal_load(maskb + at + 1) > 0.0f ? v1 : 0.0f;
        }
        yb[at] = v0;
        if (two) yb[at + 1] = v1;
      }
    }
  }
}

// U = G g G^T for every (output, input) channel pair, laid out [k = 4 i + j][c' ][m'] (zero rows / columns of padding).
// forward: m = co, c = ci, g = w[co][ci];  transpose (backward w.r.t. the input): m = ci, c = co, g = w[co][ci] rotated by 180 degrees
__global__ void conv2d_wino_prep_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int transpose, int kpad, int mpad) {
  const long long n = static_cast<long long>(kpad) * mpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    const int m = static_cast<int>(i % mpad), c = static_cast<int>(i / mpad);
    const int co = transpose ? c : m, ci = transpose ? m : c;
    float u[16];
    if (co < cout && ci < cin) {
      const float* gp = w + (static_cast<long long>(co) * cin + ci) * 9;
      float g[9], t[4][3];
#pragma unroll
      for (int q = 0; q < 9; ++q) g[q] = gp[transpose ? 8 - q : q];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0;
        t[1][j] = ((g0 + g1) + g2) * 0.5f;
        t[2][j] = ((g0 - g1) + g2) * 0.5f;
        t[3][j] = g2;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u[q * 4 + 0] = t[q][0];
        u[q * 4 + 1] = ((t[q][0] + t[q][1]) + t[q][2]) * 0.5f;
        u[q * 4 + 2] = ((t[q][0] - t[q][1]) + t[q][2]) * 0.5f;
        u[q * 4 + 3] = t[q][2];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) u[q] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[q * n + i] = u[q];
  }
}

int round_up_w(int v, int q) { return (v + q - 1) / q * q; }

}  // namespace

extern "C" {

int64_t adv_conv2d_wino_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 16LL * round_up_w(k, kWC) * round_up_w(m, kWCO);
}

int adv_conv2d_wino_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up_w(k, kWC), mpad = round_up_w(m, kWCO);
  const long long n = static_cast<long long>(kpad) * mpad;
  const unsigned blocks = static_cast<unsigned>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
  hipLaunchKernelGGL(conv2d_wino_prep_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, w_prep, cout, cin, transpose ? 1 : 0,
                     kpad, mpad);
  return adv_internal_finish_launch();
}

int adv_conv2d_wino_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                        int cout, int h, int w, int relu, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) return ADV_EINVAL;
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  EpiW epi{bias, residual, mask, relu ? 1 : 0};
  const int cinpad = round_up_w(cin, kWC), copad = round_up_w(cout, kWCO);
  const int tiles_w = (w + 31) / 32, tiles_h = (h + 7) / 8;
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + kWCO - 1) / kWCO;
  if (tiles > 0x7fffffffLL || cgroups > 65535 || b > 65535) return ADV_EINVAL;
  const size_t lds = sizeof(float) * static_cast<size_t>(2 * (kSXw + kSWw) + kSWw);
  if (!adv_internal_lds_limit<conv2d_3x3_wino>(lds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL(conv2d_3x3_wino, dim3(static_cast<unsigned>(tiles), cgroups, b), dim3(512), lds, static_cast<hipStream_t>(stream), x, w_prep, y,
                     cin, cout, cinpad, copad, h, w, tiles_w, epi);
  return adv_internal_finish_launch();
}

}  // extern "C"
