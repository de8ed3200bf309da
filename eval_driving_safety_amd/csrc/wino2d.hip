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
#include <cstdlib>
#include <type_traits>

#include "adv_internal.h"
#include "advengine.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kWC = 8;       // input channels per stage
constexpr int kWCO = 64;     // output channels per workgroup
constexpr int kWStr = 64;    // LDS row of U (64 channels) and V (64 patches); column ^ ((channel & 3) << 4): the four 16-lane groups of a read
                             // (four consecutive channels) fall into four different sets of 16 banks without padding
constexpr int kSWw = 16 * kWC * kWStr;             // 8192 floats (U of a stage; V of a stage has the same shape)
constexpr int kWNw = 16 * kWC * (kWCO / 4);        // 2048 float4 per stage
constexpr int kWSl = kWNw / 512;

template <int PR, int PC>     // patch rows x patch columns of a workgroup's tile (PR * PC = 64): 4 x 16 (8 x 32 outputs) or 8 x 8 (16 x 16)
struct WGeo {
  static_assert(PR * PC == 64, "64 patches per workgroup");
  static constexpr int kRows = 2 * PR + 2, kLW = 2 * PC + 8;     // input rows; LDS row: column = gw - (w0 - 5), so a patch row starts on an even column
  static constexpr int kSX = kWC * kRows * kLW, kXN = kSX / 4, kXSl = (kXN + 511) / 512;
};

struct EpiW {
  const float* bias;
  const float* residual;
  const float* mask;
  int relu;
};

template <int PR, int PC, bool DBG>
__global__ __launch_bounds__(512, 1) void conv2d_3x3_wino(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int Cin,
                                                         int Cout, int cinpad, int copad, int H, int W, int tiles_w, long long total, EpiW epi, int dbg_arg) {
  using G = WGeo<PR, PC>;
  const int dbg = DBG ? dbg_arg : 0;      // phase ablation for timing: compiled in only for the -DADV_TEST_HOOKS build's probe
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int cob = wave & 3, hf = wave >> 2;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 2 * PC, h0 = ht * 2 * PR, co0 = blockIdx.y * kWCO;
  const long long b = blockIdx.z;
  const long long HW = static_cast<long long>(H) * W;

  v4f acc[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

  // fetch plan of the input tile, once per tile: slot = tid + 512 i -> (channel c, tile row r, float4 group j).  The loads themselves are
  // UNCONDITIONAL (no divergent branch around a load: the compiler would have to wait for it at the join, i.e. before the matrix
  // instructions): a slot that lies outside the image loads from a clamped address and is zeroed, element by element, when it is committed
  // to LDS after the stage's matrix instructions.
  long long xflat[G::kXSl];                  // gh * W + gw of the group's first element (may lie outside the row / the image)
  int xc[G::kXSl];
  unsigned xvm[G::kXSl];                     // bit e: element e of the group is a pixel of the image
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) {
    const int sidx = tid + 512 * i;
    const int j = sidx % (G::kLW / 4), r = (sidx / (G::kLW / 4)) % G::kRows, c = sidx / ((G::kLW / 4) * G::kRows);
    const int gh = h0 - 1 + r, gw = w0 - 5 + 4 * j;
    xc[i] = c < kWC ? c : kWC - 1;
    xflat[i] = static_cast<long long>(gh) * W + gw;
    unsigned vm = 0;
    if (sidx < G::kXN && gh >= 0 && gh < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) vm |= (gw + e >= 0 && gw + e < W) ? (1u << e) : 0u;
    xvm[i] = vm;
  }
  const long long xlast = total - 4;         // the last float4 that lies inside the tensor
  v4f rx[G::kXSl], rw[kWSl];
  int rxs[G::kXSl];                          // how far the clamp moved the load (non-zero only in the tensor's first / last three floats)
  unsigned rxm[G::kXSl];
  // Everything below is written per slot / per row so that a stage's side work (global loads of the stages ahead, the input transform
  // of the next stage, the LDS commits) can be placed BETWEEN the matrix instructions of the current stage, one piece per step.
  auto fetch_x1 = [&](int i, int c0) {
    const int ch = c0 + xc[i];
    const long long at = (b * Cin + (ch < Cin ? ch : Cin - 1)) * HW + xflat[i];
    const long long cl = at < 0 ? 0 : (at > xlast ? xlast : at);
    rx[i] = *reinterpret_cast<const v4f_u*>(x + cl);
    rxs[i] = static_cast<int>(cl - at);
    rxm[i] = ch < Cin ? xvm[i] : 0u;
  };
  auto fetch_w1 = [&](int i, int c0) {
    const int sidx = tid + 512 * i;
    const int q = sidx % (kWCO / 4), row = sidx / (kWCO / 4);      // row = k * 8 + c
    const int k = row / kWC, c = row % kWC;
    rw[i] = *reinterpret_cast<const v4f*>(wp + (static_cast<long long>(k) * cinpad + c0 + c) * copad + co0 + 4 * q);
  };
  float* const sxb = lds;                               // [2][8][rows][LW]   input tiles
  float* const swb = lds + 2 * G::kSX;                   // [2][16][8][64]     U = G g G^T of the stage's channels
  float* const svb = swb + 2 * kSWw;                     // [2][16][8][64]     V = B^T d B of the stage's channels, 64 patches
  auto commit_x1 = [&](int i, int buf) {
    float* sx = sxb + buf * G::kSX;
    const int sidx = tid + 512 * i;
    const v4f t = rx[i];
    const int sh = rxs[i];
    const unsigned m = rxm[i];
    v4f v;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(sh != 0) != 0, 0)) {     // wave-uniform: the tensor's first / last float4 only
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = e - sh;
        v[e] = k == 0 ? t[0] : (k == 1 ? t[1] : (k == 2 ? t[2] : t[3]));
      }
    } else {
      v = t;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (m >> e) & 1u ? v[e] : 0.0f;
    if (sidx < G::kXN) *reinterpret_cast<v4f*>(sx + 4 * sidx) = v;
  };
  auto commit_w1 = [&](int i, int buf) {
    float* sw = swb + buf * kSWw;
    const int sidx = tid + 512 * i;
    const int q = sidx % (kWCO / 4), row = sidx / (kWCO / 4);
    *reinterpret_cast<v4f*>(sw + row * kWStr + ((4 * q) ^ ((row & 3) << 4))) = rw[i];
  };
  // the input transform: thread -> (channel tid >> 6, patch tid & 63); 8 LDS reads of two floats, 32 additions, 16 LDS writes
  const int tc = tid >> 6, tp = tid & 63;
  const int toff = (tc * G::kRows + 2 * (tp / PC)) * G::kLW + 2 * (tp % PC) + 4;
  const int voff = tc * kWStr + (tp ^ ((tc & 3) << 4));
  float td[4][4], tv[16];
  auto tr_read = [&](int i, int buf) {
    const float* dp = sxb + buf * G::kSX + toff + i * G::kLW;
    const v2f lo = *reinterpret_cast<const v2f*>(dp), hi = *reinterpret_cast<const v2f*>(dp + 2);
    td[i][0] = lo[0], td[i][1] = lo[1], td[i][2] = hi[0], td[i][3] = hi[1];
  };
  auto tr_compute = [&]() {
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = td[0][j] - td[2][j];
      t[1][j] = td[1][j] + td[2][j];
      t[2][j] = td[2][j] - td[1][j];
      t[3][j] = td[1][j] - td[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tv[i * 4 + 0] = t[i][0] - t[i][2];
      tv[i * 4 + 1] = t[i][1] + t[i][2];
      tv[i * 4 + 2] = t[i][2] - t[i][1];
      tv[i * 4 + 3] = t[i][1] - t[i][3];
    }
  };
  auto tr_write = [&](int k, int buf) { svb[buf * kSWw + voff + k * kWC * kWStr] = tv[k]; };

  // operands of the matrix instructions: A = U_k[co = lane & 15][c = lane >> 4], B = V_k[c = lane >> 4][patch = lane & 15]
  const int aoff = k4 * kWStr + ((cob * 16 + i16) ^ (k4 << 4));
  const int boff0 = k4 * kWStr + (((2 * hf) * 16 + i16) ^ (k4 << 4)), boff1 = k4 * kWStr + (((2 * hf + 1) * 16 + i16) ^ (k4 << 4));

  const int nstage = (Cin + kWC - 1) / kWC;
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, 0);
#pragma unroll
  for (int i = 0; i < kWSl; ++i) fetch_w1(i, 0);
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 0);
#pragma unroll
  for (int i = 0; i < kWSl; ++i) commit_w1(i, 0);
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, nstage > 1 ? kWC : 0);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) tr_read(i, 0);
  tr_compute();
#pragma unroll
  for (int k = 0; k < 16; ++k) tr_write(k, 0);
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 1);
  __syncthreads();

  // Stage st: 32 steps (channel group cs = t >> 4, transform position k = t & 15) of two matrix instructions each on V / U of stage st.
  // The operands of step t + 4 are read from LDS before step t issues.  Between the steps, one piece each: the global loads of the
  // inputs of stage st + 2 and the weights of stage st + 1 (steps 0..5), the input transform of stage st + 1 (LDS reads at steps 6..9,
  // the additions at 12, LDS writes at 14..21) and the commits of the loaded data to LDS (steps 24..29) - all of them touch buffers the
  // current stage's products do not read.  The scheduler may not move anything across a step (sched_barrier): it would hoist all reads
  // and spill.  ONE barrier per stage.
  auto products = [&](int st, auto fx_tag, auto fw_tag) {
    constexpr bool FX = decltype(fx_tag)::value, FW = decltype(fw_tag)::value;
    constexpr int kAhead = 4;
    float ra[32], rb0[32], rb1[32];
    const float* ap = swb + (st & 1) * kSWw + aoff;
    const float* bp = svb + (st & 1) * kSWw;
    auto load = [&](int t) {
      const int row = ((t & 15) * kWC + (t >> 4) * 4) * kWStr;
      ra[t] = ap[row];
      rb0[t] = bp[row + boff0];
      rb1[t] = bp[row + boff1];
    };
    const int nb = (st + 1) & 1;       // the buffers of stage st + 1 (V, U) - and of stage st + 2's inputs: st & 1
#pragma unroll
    for (int t = 0; t < kAhead; ++t) load(t);
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      if (t + kAhead < 32) load(t + kAhead);
      if (FX && t < G::kXSl && !(dbg & 4)) fetch_x1(t, (st + 2) * kWC);
      if (FW && t >= 2 && t < 2 + kWSl && !(dbg & 4)) fetch_w1(t - 2, (st + 1) * kWC);
      if (FW && t >= 6 && t < 10 && !(dbg & 1)) tr_read(t - 6, nb);
      if (FW && t == 12 && !(dbg & 1)) tr_compute();
      if (FW && t >= 14 && t < 22 && !(dbg & 1)) {
        tr_write(2 * (t - 14), nb);
        tr_write(2 * (t - 14) + 1, nb);
      }
      if (FX && t >= 24 && t < 24 + G::kXSl && !(dbg & 8)) commit_x1(t - 24, st & 1);
      if (FW && t >= 26 && t < 26 + kWSl && !(dbg & 8)) commit_w1(t - 26, nb);
      if (!(dbg & 2)) {
        acc[0][t & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], rb0[t], acc[0][t & 15], 0, 0, 0);
        acc[1][t & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t], rb1[t], acc[1][t & 15], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The loop is peeled by hand so that every global load and the LDS write that consumes it sit in the SAME straight-line block: with
  // "if (more) fetch ... if (more) commit" the compiler must assume a path on which a load is still in flight at the loop's head and
  // waits for ALL loads (the fresh ones too) before it may reuse the registers - i.e. before the matrix instructions.
  int st = 0;
  for (; st + 2 < nstage; ++st) {
    products(st, std::true_type{}, std::true_type{});
    if (!(dbg & 16)) __syncthreads();
  }
  if (st + 1 < nstage) {
    products(st, std::false_type{}, std::true_type{});
    __syncthreads();
    ++st;
  }
  products(st, std::false_type{}, std::false_type{});

  const long long MP = static_cast<long long>(Cout) * HW;
  float* yb = y + b * MP;
  const float* resb = epi.residual ? epi.residual + b * MP : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP : nullptr;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int t = (2 * hf + blk) * 16 + i16;
    const int gh = h0 + 2 * (t / PC), gw = w0 + 2 * (t % PC);
    if (gh >= H || gw >= W) continue;
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

template <int PR, int PC>
int launch_wino(const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int h, int w, const EpiW& epi, hipStream_t st) {
  using G = WGeo<PR, PC>;
  const int tiles_w = (w + 2 * PC - 1) / (2 * PC), tiles_h = (h + 2 * PR - 1) / (2 * PR);
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + kWCO - 1) / kWCO;
  if (tiles > 0x7fffffffLL || cgroups > 65535 || b > 65535) return ADV_EINVAL;
  const size_t lds = 2 * sizeof(float) * static_cast<size_t>(G::kSX + 2 * kSWw);
  const long long total = static_cast<long long>(b) * cin * h * w;
  const dim3 grid(static_cast<unsigned>(tiles), cgroups, b);
#ifdef ADV_TEST_HOOKS
  if (const char* dbg_s = adv_hook_value("ADV_WINO_DBG")) {      // phase ablation for timing (results are wrong)
    if (!adv_internal_lds_limit<conv2d_3x3_wino<PR, PC, true>>(lds)) return ADV_ELAUNCH;
    hipLaunchKernelGGL((conv2d_3x3_wino<PR, PC, true>), grid, dim3(512), lds, st, x, wp, y, cin, cout, cinpad, copad, h, w, tiles_w, total, epi,
                       std::atoi(dbg_s));
    return adv_internal_finish_launch();
  }
#endif
  if (!adv_internal_lds_limit<conv2d_3x3_wino<PR, PC, false>>(lds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv2d_3x3_wino<PR, PC, false>), grid, dim3(512), lds, st, x, wp, y, cin, cout, cinpad, copad, h, w, tiles_w, total, epi, 0);
  return adv_internal_finish_launch();
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
                        int cout, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || tile < -1 || tile > 1) return ADV_EINVAL;
  if (static_cast<long long>(b) * cin * h * w < 4) return ADV_EINVAL;      // the kernel loads whole float4s (clamped into the tensor)
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  EpiW epi{bias, residual, mask, relu ? 1 : 0};
  const int cinpad = round_up_w(cin, kWC), copad = round_up_w(cout, kWCO);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // 8 x 32 outputs per workgroup (128-byte store runs), or 16 x 16 where that wastes fewer patches (the 14 x 14 maps of the box heads:
  // one tile instead of two per image)
  auto waste = [&](int th, int tw) { return static_cast<long long>((h + th - 1) / th) * ((w + tw - 1) / tw); };
  const int t = tile >= 0 ? tile : (waste(16, 16) * 10 < waste(8, 32) * 8 ? 1 : 0);
  if (t == 1) return launch_wino<8, 8>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
  return launch_wino<4, 16>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
}

}  // extern "C"
