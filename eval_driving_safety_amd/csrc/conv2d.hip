// 2D convolutions of the detectors' backbones on the gfx950 float32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// 1x1 / stride 1 - a plain GEMM per image:  Y_b[co][p] = sum_ci W[co][ci] * X_b[ci][p],  p = h*W + w  (NCHW: X_b is a
// row-major [Cin][P] matrix as it stands, so no im2col and no transposition):
//   A operand = one weight per lane (row = output channel = lane & 31, k = lane >> 5), B operand = one input value per lane
//   (k = lane >> 5, column = pixel = lane & 31); every accumulator register holds 32 consecutive pixels of one output channel,
//   so the epilogue's loads (skip connection, mask) and stores are 128-byte runs.
//   A 256-thread workgroup owns BM output channels x BN pixels; per stage of kKC = 16 input channels the X rows [16][BN] and the
//   weight rows [16][BM] (prepared layout [Cin'][Cout']: a stage is 16 contiguous rows) go global -> registers -> LDS while the
//   MFMAs of the previous stage run (two LDS buffers, one barrier per stage).  Rows of X are dword-aligned only (P is odd for
//   most feature maps), hence register staging with dword-aligned float4 loads rather than LDS-DMA.
//   Five tile shapes (128x256 ... 64x64, 64x32 with two waves); the host picks the one that needs the fewest "MFMA rounds" for the layer's size.
//   Accumulation order: ci ascending, one fmaf per product starting from 0 (the MFMA is a k-ordered fmaf chain), then
//   + bias, + residual, ReLU, mask - the C oracle restates it bit for bit.
//   The backward w.r.t. the input is the same kernel on W^T (adv_conv2d_1x1_prep_weights_f32(transpose = 1)); `mask` (the layer's
//   own input, a ReLU output) turns its result into the gradient w.r.t. the previous layer's PRE-activation in the epilogue.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

#ifdef ADV_C2_STAMPS
// diagnostic builds only (tools/build_variant.sh c2stamps conv2d.hip -DADV_C2_STAMPS; tools/c2_stamps.py): s_memtime stamps of the 1x1 kernel -
// [workgroup < 16][wave < 4][stage < 62 | 62: life | 63: epilogue][5]
__device__ unsigned long long adv_c2_stamps[16][4][64][5];
extern "C" __attribute__((visibility("default"))) int adv_debug_c2_stamps(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(adv_c2_stamps), bytes < sizeof(adv_c2_stamps) ? bytes : sizeof(adv_c2_stamps)));
}
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));  // a float4 the compiler may not assume 16-byte aligned

constexpr int kKC = 16;  // input channels per LDS stage

struct Epi2 {
  const float* bias;      // [M] or null
  const float* residual;  // laid out like y, or null: added after the bias, before the ReLU
  const float* mask;      // laid out like y, or null: result = mask > 0 ? result : 0, applied last
  int relu;
};

template <int WM, int WN, int TM, int TN>
struct GemmGeo {
  static_assert(WM * WN == 4 || WM * WN == 2, "four waves, or two (the 64 x 32 tile)");
  static constexpr int kT = 64 * WM * WN;                                        // threads
  static constexpr int kBM = WM * TM * 32, kBN = WN * TN * 32;
  static constexpr int kXF4 = kKC * kBN / 4 / kT, kWF4 = kKC * kBM / 4 / kT;     // float4 per thread and stage
  static_assert(kXF4 >= 1 && kWF4 >= 1 && kXF4 * 4 * kT == kKC * kBN && kWF4 * 4 * kT == kKC * kBM, "the loader threads cover a stage in whole float4 rounds");
  static constexpr int kStage = kKC * (kBN + kBM);                               // floats per LDS buffer
};

// x [B][K][P], wp [Kpad][mpad] (zero padded), y [B][M][P].  One workgroup per (m tile, n tile, image); the linear block index is
// remapped so that each XCD (blocks i, i+8, ... share one) walks a contiguous range of the tile order - m fastest, so the m tiles
// that re-read one X tile sit in the same L2.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv2d_1x1_mfma(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int K, int M,
                                                       int mpad, long long P, int tiles_m, int tiles_n, long long ntiles, long long wbytes, Epi2 epi) {
  using G = GemmGeo<WM, WN, TM, TN>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

#ifdef ADV_C2_STAMPS
  const unsigned long long c2_entry = __builtin_amdgcn_s_memtime();
  const bool c2_on = (blockIdx.x < 8 || blockIdx.x + 8 >= gridDim.x) && (WM * WN == 4);      // the first eight and the LAST eight workgroups
  const int c2_slot = blockIdx.x < 8 ? blockIdx.x : 8 + static_cast<int>(blockIdx.x + 8 - gridDim.x);
  const int c2_wave = threadIdx.x >> 6;
#endif
  long long t = blockIdx.x;
  {
    const long long base = ntiles >> 3, rem = ntiles & 7, xcd = t & 7, j = t >> 3;
    t = xcd * base + (xcd < rem ? xcd : rem) + j;
  }
  const int mt = static_cast<int>(t % tiles_m);
  const long long r = t / tiles_m;
  const int nt = static_cast<int>(r % tiles_n);
  const long long b = r / tiles_n;
  const int m0 = mt * G::kBM;
  const long long n0 = static_cast<long long>(nt) * G::kBN;
  const long long KP = static_cast<long long>(K) * P;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.0f;

  // <round 5> Staging by BUFFER loads, two register sets deep.  The counters (profiles/r04_conv_pmc.json) had this kernel at ten vector
  // instructions per matrix instruction and 0.38-0.46 matrix-pipe busy: every stage recomputed 64-bit addresses, clamps and element masks,
  // and a stage is short (16 channels: 8-64 matrix instructions per wave), so a load requested at its start was not back at its end.
  // Now a slot's byte offset inside the image / the prepared weights is computed ONCE; a stage adds a wave-uniform offset (one v_add per
  // input slot; the scalar offset operand for the weights) and the hardware's range check returns zeros for rows >= K and behind the
  // tensor - no clamp, no mask (columns past a row's end read the next row: never stored).  Every load travels a whole stage longer: stage s
  // requests stage s + 2 into one register set and commits the other (requested during stage s - 1); the loop is uniform, unrolled by two
  // for the alternation, requests past the last stage are out of range (zeros) and their commits fill a buffer nobody reads.  Same bits.
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + b * KP), 0, static_cast<int>(static_cast<unsigned>(KP * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, static_cast<int>(static_cast<unsigned>(wbytes)), 0x00020000);
  int xvo[G::kXF4], wvo[G::kWF4];
#pragma unroll
  for (int i = 0; i < G::kXF4; ++i) {
    const int f = tid + G::kT * i, row = f / (G::kBN / 4), c4 = f % (G::kBN / 4);
    xvo[i] = static_cast<int>((static_cast<unsigned>(row) * static_cast<unsigned>(P) + static_cast<unsigned>(n0) + 4u * c4) * 4u);
  }
#pragma unroll
  for (int i = 0; i < G::kWF4; ++i) {
    const int f = tid + G::kT * i, row = f / (G::kBM / 4), c4 = f % (G::kBM / 4);
    wvo[i] = (row * mpad + m0 + c4 * 4) * 4;
  }
  struct Set {
    v4f x[G::kXF4], w[G::kWF4];
  };
  Set sa, sb;
  const int nstage = (K + kKC - 1) / kKC;
  auto fetch = [&](int k0, Set& st) {
    const int xo = static_cast<int>(static_cast<unsigned>(k0) * static_cast<unsigned>(P) * 4u);      // wave-uniform
#pragma unroll
    for (int i = 0; i < G::kXF4; ++i) st.x[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsx, xvo[i] + xo, 0, 0));
    // (the scalar offset takes no part in the range check: the two look-ahead stages past the last one re-read the last stage - in bounds,
    // into a register set nobody reads - instead of reaching past the prepared weights)
    const int kw = k0 < nstage * kKC ? k0 : (nstage - 1) * kKC;
#pragma unroll
    for (int i = 0; i < G::kWF4; ++i) st.w[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo[i], kw * mpad * 4, 0));
  };
  auto commit = [&](int buf, const Set& st) {
    float* sx = lds + buf * G::kStage;
    float* sw = sx + kKC * G::kBN;
#pragma unroll
    for (int i = 0; i < G::kXF4; ++i) *reinterpret_cast<v4f*>(sx + 4 * (tid + G::kT * i)) = st.x[i];         // [row][BN] row-major == float4 index
#pragma unroll
    for (int i = 0; i < G::kWF4; ++i) *reinterpret_cast<v4f*>(sw + 4 * (tid + G::kT * i)) = st.w[i];
  };
  auto products = [&](int s) {
    const float* sx = lds + (s & 1) * G::kStage;
    const float* sw = sx + kKC * G::kBN;
#pragma unroll
    for (int kk = 0; kk < kKC; kk += 2) {
      float a[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = sw[(kk + half) * G::kBM + (wm * TM + i) * 32 + l32];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[jn] = sx[(kk + half) * G::kBN + (wn * TN + jn) * 32 + l32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[jn], acc[i][jn], 0, 0, 0);
    }
  };

#ifdef ADV_C2_STAMPS
  const unsigned long long c2_setup = __builtin_amdgcn_s_memtime();
#endif
  if (tid < G::kBM) lds[2 * G::kStage + tid] = epi.bias ? epi.bias[m0 + tid < M ? m0 + tid : 0] : 0.0f;      // the tile's bias values: read back in the epilogue without a memory round trip between stores
  fetch(0, sa);
  fetch(kKC, sb);
  commit(0, sa);
  __syncthreads();
#ifdef ADV_C2_STAMPS
  const unsigned long long c2_loop = __builtin_amdgcn_s_memtime();
#endif
  auto stage = [&](int s, Set& fs, const Set& cs) {
#ifdef ADV_C2_STAMPS
    const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ADV_C2_NOFETCH      // (timing experiments, wrong results: no requests after the first two stages / no barrier / no LDS commit)
    fetch((s + 2) * kKC, fs);
#endif
    __builtin_amdgcn_sched_barrier(0);     // the loads stay AHEAD of the matrix instructions (the scheduler would sink them to their use)
#ifdef ADV_C2_STAMPS
    const unsigned long long q1 = __builtin_amdgcn_s_memtime();
#endif
    products(s);
#ifdef ADV_C2_STAMPS
    const unsigned long long q2 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ADV_C2_NOCOMMIT
    commit((s + 1) & 1, cs);
#endif
#ifdef ADV_C2_STAMPS
    const unsigned long long q3 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ADV_C2_NOBARRIER
    __syncthreads();
#endif
#ifdef ADV_C2_STAMPS
    const unsigned long long q4 = __builtin_amdgcn_s_memtime();
    if (c2_on && s < 62 && (threadIdx.x & 63) == 0) {
      unsigned long long* o = adv_c2_stamps[c2_slot][c2_wave][s];
      o[0] = q0, o[1] = q1, o[2] = q2, o[3] = q3, o[4] = q4;
    }
#endif
  };
  int s = 0;
  for (; s + 1 < nstage; s += 2) {
    stage(s, sa, sb);
    stage(s + 1, sb, sa);
  }
  if (s < nstage) products(s);
#ifdef ADV_C2_STAMPS
  const unsigned long long c2_loop_end = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: register v of a 32x32 accumulator = output channel (v & 3) + 8 * (v >> 2) + 4 * half, pixel = lane & 31
  const long long MP = static_cast<long long>(M) * P;
  float* yb = y + b * MP;
  const float* resb = epi.residual ? epi.residual + b * MP : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const long long p = n0 + (wn * TN + jn) * 32 + l32;
      const int cbase = m0 + (wm * TM + i) * 32 + 4 * half;
      if (p >= P) continue;
      // the skip connection and the mask of all sixteen outputs first (unconditional loads: an element past the last channel reads the
      // tensor's first float and is not stored), the stores after them - a load behind a store to y, or inside a divergent branch, is
      // waited for on the spot: a memory round trip per element
      float rv[16], mv[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) rv[v] = 0.0f, mv[v] = 1.0f;
      if (resb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int co = cbase + (v & 3) + 8 * (v >> 2);
          rv[v] = __builtin_nontemporal_load(co < M ? resb + static_cast<long long>(co) * P + p : epi.residual);
        }
      }
      if (maskb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int co = cbase + (v & 3) + 8 * (v >> 2);
          mv[v] = __builtin_nontemporal_load(co < M ? maskb + static_cast<long long>(co) * P + p : epi.mask);
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co = cbase + (v & 3) + 8 * (v >> 2);
        if (co >= M) continue;
        const long long at = static_cast<long long>(co) * P + p;
        float r = acc[i][jn][v];
        if (epi.bias) r = r + lds[2 * G::kStage + co - m0];
        if (resb) r = r + rv[v];
        if (epi.relu) r = r > 0.0f ? r : 0.0f;
        if (maskb) r = mv[v] > 0.0f ? r : 0.0f;
#ifdef ADV_C2_NOSTORE      // (timing experiment: the store only where the result is NaN - never, but the compiler keeps the arithmetic)
        if (r != r) yb[at] = r;
#else
        yb[at] = r;
#endif
      }
    }
  }
#ifdef ADV_C2_STAMPS
  const unsigned long long c2_issued = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c2_end = __builtin_amdgcn_s_memtime();
  if (c2_on && (threadIdx.x & 63) == 0) {
    unsigned long long* o = adv_c2_stamps[c2_slot][c2_wave][62];
    o[0] = c2_entry, o[1] = c2_setup, o[2] = c2_loop, o[3] = c2_loop_end, o[4] = c2_end;
    adv_c2_stamps[c2_slot][c2_wave][63][0] = c2_issued;
  }
#endif
}

// ---- the same GEMM in HALF-SIZE wave units (tile index 5): 64 output channels x 32 pixels per workgroup, four waves, each wave 32 channels x
// 16 pixels as two v_mfma_f32_16x16x4_f32 blocks that share their pixel operand.  The matrix pipes run the waves of a SIMD one after another,
// so a launch takes ceil(waves / SIMDs) wave-times: 1024 -> 256 on [2,1024,38,125] is 2.33 waves of 32 x 32 per SIMD = 3 wave-times; in
// half-size units it is 4.66 = 5 half wave-times (2.5).  The 16x16x4 instruction adds its four k in ascending order like the 32x32x2 one its
// two: the same fmaf chain per output as every other shape - the same bits (tests/test_conv2d.py runs all six shapes against one oracle).
// Stage = 32 input channels; LDS rows padded (pixels 32 -> 48 floats, channels 64 -> 80) so that the four 16-lane groups of an operand read
// (k = lane / 16) fall into different banks.
constexpr int kKC16 = 32, kXP16 = 48, kWP16 = 80, kStage16 = kKC16 * (kXP16 + kWP16);
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv2d_1x1_mfma16(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int K, int M, int mpad,
                                                            long long P, int tiles_m, int tiles_n, long long ntiles, long long wbytes, Epi2 epi) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  long long t = blockIdx.x;
  {
    const long long base = ntiles >> 3, rem = ntiles & 7, xcd = t & 7, j = t >> 3;
    t = xcd * base + (xcd < rem ? xcd : rem) + j;
  }
  const int mt = static_cast<int>(t % tiles_m);
  const long long r = t / tiles_m;
  const int nt = static_cast<int>(r % tiles_n);
  const long long b = r / tiles_n;
  const int m0 = mt * 64;
  const long long n0 = static_cast<long long>(nt) * 32;
  const long long KP = static_cast<long long>(K) * P;
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + b * KP), 0, static_cast<int>(static_cast<unsigned>(KP * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, static_cast<int>(static_cast<unsigned>(wbytes)), 0x00020000);
  // one float4 of the pixel slab [32 rows][32 pixels] and two of the weight slab [32 rows][64 channels] per thread and stage
  const int xrow = tid >> 3, xc4 = tid & 7;
  const int xvo = static_cast<int>((static_cast<unsigned>(xrow) * static_cast<unsigned>(P) + static_cast<unsigned>(n0) + 4u * xc4) * 4u);
  int wvo[2], wrow[2], wc4[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;
    wrow[i] = f >> 4, wc4[i] = f & 15;
    wvo[i] = (wrow[i] * mpad + m0 + 4 * wc4[i]) * 4;
  }
  struct Set {
    v4f x, w[2];
  };
  Set sa, sb;
  const int nstage = (K + kKC16 - 1) / kKC16;
  auto fetch = [&](int k0, Set& st) {
    const int xo = static_cast<int>(static_cast<unsigned>(k0) * static_cast<unsigned>(P) * 4u);      // wave-uniform; rows >= K fail the range check: zeros
    st.x = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsx, xvo + xo, 0, 0));
    // (look-ahead stages past the last one re-read the last stage into a set nobody reads; the stage's offset travels in the VECTOR offset,
    // which the range check covers - the prepared weights are padded to 16 rows, a 32-row stage may reach past them: zeros)
    const int kw = k0 < nstage * kKC16 ? k0 : (nstage - 1) * kKC16;
#pragma unroll
    for (int i = 0; i < 2; ++i) st.w[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo[i] + kw * mpad * 4, 0, 0));
  };
  auto commit = [&](int buf, const Set& st) {
    float* sx = lds + buf * kStage16;
    float* sw = sx + kKC16 * kXP16;
    *reinterpret_cast<v4f*>(sx + xrow * kXP16 + 4 * xc4) = st.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(sw + wrow[i] * kWP16 + 4 * wc4[i]) = st.w[i];
  };
  auto products = [&](int s) {
    const float* sx = lds + (s & 1) * kStage16 + g * kXP16 + wn * 16 + l16;
    const float* sw = lds + (s & 1) * kStage16 + kKC16 * kXP16 + g * kWP16 + wm * 32 + l16;
#pragma unroll
    for (int kk = 0; kk < kKC16; kk += 4) {
      const float a0 = sw[kk * kWP16], a1 = sw[kk * kWP16 + 16], bv = sx[kk * kXP16];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1, 0, 0, 0);
    }
  };
  if (tid < 64) lds[2 * kStage16 + tid] = epi.bias ? epi.bias[m0 + tid < M ? m0 + tid : 0] : 0.0f;
  fetch(0, sa);
  fetch(kKC16, sb);
  commit(0, sa);
  __syncthreads();
  auto stage = [&](int s, Set& fs, const Set& cs) {
    fetch((s + 2) * kKC16, fs);
    __builtin_amdgcn_sched_barrier(0);
    products(s);
    commit((s + 1) & 1, cs);
    __syncthreads();
  };
  int s = 0;
  for (; s + 1 < nstage; s += 2) {
    stage(s, sa, sb);
    stage(s + 1, sb, sa);
  }
  if (s < nstage) products(s);

  // ---- epilogue: register r of a 16 x 16 accumulator = output channel 4 (lane / 16) + r of its block, pixel = lane % 16
  const long long MP = static_cast<long long>(M) * P;
  float* yb = y + b * MP;
  const float* resb = epi.residual ? epi.residual + b * MP : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP : nullptr;
  const long long p = n0 + wn * 16 + l16;
  if (p >= P) return;
  const int cbase = m0 + wm * 32 + 4 * g;
  float rv[8], mv[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {      // the skip connection and the mask first, unconditionally (see the 32 x 32 kernel)
    const int co = cbase + 16 * (v >> 2) + (v & 3);
    rv[v] = resb ? __builtin_nontemporal_load(co < M ? resb + static_cast<long long>(co) * P + p : epi.residual) : 0.0f;
    mv[v] = maskb ? __builtin_nontemporal_load(co < M ? maskb + static_cast<long long>(co) * P + p : epi.mask) : 1.0f;
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int co = cbase + 16 * (v >> 2) + (v & 3);
    if (co >= M) continue;
    float o = v < 4 ? acc0[v & 3] : acc1[v & 3];
    if (epi.bias) o = o + lds[2 * kStage16 + co - m0];
    if (resb) o = o + rv[v];
    if (epi.relu) o = o > 0.0f ? o : 0.0f;
    if (maskb) o = mv[v] > 0.0f ? o : 0.0f;
    yb[static_cast<long long>(co) * P + p] = o;
  }
}

__global__ void conv2d_1x1_prep_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int transpose, int kpad, int mpad) {
  // forward:  out[k = ci][m = co] = w[co][ci];   transpose (backward w.r.t. the input): out[k = co][m = ci] = w[co][ci]
  const long long n = static_cast<long long>(kpad) * mpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    const int k = static_cast<int>(i / mpad), m = static_cast<int>(i % mpad);
    const int co = transpose ? k : m, ci = transpose ? m : k;
    out[i] = (co < cout && ci < cin) ? w[static_cast<long long>(co) * cin + ci] : 0.0f;
  }
}

int round_up(int v, int q) { return (v + q - 1) / q * q; }

int cu_count() {  // compute units of the current device (256 on MI355X); queried once per host thread
  static thread_local int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached = n;
  }
  return cached;
}

template <int WM, int WN, int TM, int TN>
int launch_1x1(const float* x, const float* wp, float* y, int b, int K, int M, int mpad, long long P, const Epi2& epi, hipStream_t st) {
  using G = GemmGeo<WM, WN, TM, TN>;
  const int tiles_m = (M + G::kBM - 1) / G::kBM;
  const long long tiles_n = (P + G::kBN - 1) / G::kBN;
  const long long ntiles = static_cast<long long>(tiles_m) * tiles_n * b;
  if (tiles_n > 0x7fffffffLL || ntiles > 0x7fffffffLL) return ADV_EINVAL;
  // an image and the prepared weights are addressed with 32-bit byte offsets (buffer loads): both below 4 GiB
  const long long wbytes = static_cast<long long>(round_up(K, kKC)) * mpad * 4;
  if ((static_cast<long long>(K) + kKC) * P * 4 >= 0xfff00000LL || wbytes >= 0xfff00000LL) return ADV_EINVAL;
  size_t lds = sizeof(float) * (2 * static_cast<size_t>(G::kStage) + G::kBM);      // two stage buffers + the tile's bias
  if (const char* occ = adv_hook_value("ADV_C2_OCC")) {      // test hook / A-B: at most this many workgroups per compute unit (LDS padding)
    const int r = std::atoi(occ);
    if (r >= 1 && r <= 8 && static_cast<size_t>(160 * 1024 / r) > lds) lds = static_cast<size_t>(160 * 1024 / r) & ~static_cast<size_t>(511);
  }
  if (lds > 64 * 1024 && !adv_internal_lds_limit<conv2d_1x1_mfma<WM, WN, TM, TN>>(lds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv2d_1x1_mfma<WM, WN, TM, TN>), dim3(static_cast<unsigned>(ntiles)), dim3(G::kT), lds, st, x, wp, y, K, M, mpad, P, tiles_m,
                     static_cast<int>(tiles_n), ntiles, wbytes, epi);
  return adv_internal_finish_launch();
}

int launch_1x1_m16(const float* x, const float* wp, float* y, int b, int K, int M, int mpad, long long P, const Epi2& epi, hipStream_t st) {
  const int tiles_m = (M + 63) / 64;
  const long long tiles_n = (P + 31) / 32;
  const long long ntiles = static_cast<long long>(tiles_m) * tiles_n * b;
  if (tiles_n > 0x7fffffffLL || ntiles > 0x7fffffffLL) return ADV_EINVAL;
  const long long wbytes = static_cast<long long>(round_up(K, kKC)) * mpad * 4;
  if ((static_cast<long long>(K) + 3 * kKC16) * P * 4 >= 0xfff00000LL || wbytes + 3LL * kKC16 * mpad * 4 >= 0xfff00000LL) return ADV_EINVAL;
  const size_t lds = sizeof(float) * (2 * static_cast<size_t>(kStage16) + 64);
  hipLaunchKernelGGL(conv2d_1x1_mfma16, dim3(static_cast<unsigned>(ntiles)), dim3(256), lds, st, x, wp, y, K, M, mpad, P, tiles_m, static_cast<int>(tiles_n), ntiles,
                     wbytes, epi);
  return adv_internal_finish_launch();
}

// which tile shape: 0 = 128x256, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 64x32 (output channels x pixels; 4: two waves), 5 = 64x32 in four half-size waves.  One wave's work
// is TM*TN accumulators over K; the matrix pipes run the waves of a SIMD one after another, so a launch takes about
// ceil(waves / SIMDs) * TM*TN / efficiency(shape) - bigger tiles reuse more per staged byte, smaller ones fill the chip.
int pick_1x1_tile(int b, int M, long long P, int simds) {
  static const int bm[6] = {128, 128, 64, 64, 64, 64}, bn[6] = {256, 128, 128, 64, 32, 32}, wg_waves[6] = {4, 4, 4, 4, 2, 4};
  static const double work[6] = {8, 4, 2, 1, 1, 0.5};
  // relative efficiency of a wave's MFMA stream per tile shape, fitted to the sweep of the R101 1x1 layers on MI355X (round 5, after the
  // buffer-load staging: profiles/r05_conv2d_1x1_tile_sweep.jsonl - the small tiles gained most: several workgroups per compute unit
  // cover each other's prologue and epilogue; 0.07 ms per R101 step over the per-layer optimum, the round-3 fit {0.8, 1.08, 0.94, 1} 0.26).
  // <round 6> the 64 x 32 shape (profiles/r06_conv2d_1x1_tile_sweep.jsonl): as fast as 64 x 64 where the pixels fill whole tiles, 7 % faster
  // on the RoI heads' 14 x 14 maps (196 pixels per image: 7 tiles of 32 instead of 4 of 64) - it wins where it saves more than 5 % of the waves
  // the half-size units of shape 5 (conv2d_1x1_mfma16: four waves of 32 x 16 on a 64 x 32 tile) where the whole-wave rounding of the others
  // hurts - 1024 -> 256 on [2,1024,38,125]: 2.33 waves per SIMD = 3 wave-times against 4.66 = 5 half ones: 0.0575 -> 0.0544 ms; 2048 -> 512
  // on [2,2048,19,63]: 2 against 3 halves: 0.0713 -> 0.0607 ms; both fit an efficiency of 0.88 (profiles/r06_conv2d_1x1_half_units.jsonl)
  static const double eff[6] = {0.75, 0.88, 0.99, 1.0, 0.95, 0.88};
  int best = 0;
  double best_t = 1e300;
  for (int c = 0; c < 6; ++c) {
    const double waves = static_cast<double>(wg_waves[c]) * ((M + bm[c] - 1) / bm[c]) * static_cast<double>((P + bn[c] - 1) / bn[c]) * b;
    const double rounds = static_cast<double>(static_cast<long long>((waves + simds - 1) / simds));
    const double tcost = rounds * work[c] / eff[c];
    if (tcost < best_t) best_t = tcost, best = c;
  }
  return best;
}


// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / padding = dilation (1 or 2): implicit GEMM  D[co][pixel] += W[co][ci][tap] * X[ci][pixel + tap offset], the 2D
// sibling of csrc/conv3d.hip's main kernel.  A workgroup (4 waves) owns TH output rows x 32 columns x CO output channels; wave v
// owns RW of the rows for all CO / 32 channel blocks (RW * CO / 32 = 4 accumulators per wave).  Per stage of kC3 = 8 input
// channels the input tile with its halo ([8][TH + 2 dil][40 floats]: columns w0 - 4 .. w0 + 35, every tap is a shift of the LDS
// column) and the weights [9][8][CO] are fetched global -> registers while the MFMAs of the previous stage run, then committed to
// the other LDS buffer (one barrier per stage).  A lane's fetch plan (element offsets, which float4 groups lie inside the image) is
// computed once per tile.  Order of accumulation: stage, tap, channel pair - orc_conv2d(chunk = 8) restates it bit for bit.
constexpr int kC3 = 8;
constexpr int kLW = 40;   // LDS row: 4 halo columns left, 32 outputs, 4 right

template <int RW, int CBK, int DIL>   // rows per wave, 32-channel blocks per workgroup, dilation
struct Geo3 {
  static constexpr int kTH = 4 * RW, kCO = 32 * CBK;
  static constexpr int kRows = kTH + 2 * DIL;
  static constexpr int kSX = kC3 * kRows * kLW, kSW = 9 * kC3 * kCO;       // floats per stage
  static constexpr int kXN = kC3 * kRows * (kLW / 4), kWN = 9 * kC3 * (kCO / 4);
  static constexpr int kXSlots = (kXN + 255) / 256, kWSlots = (kWN + 255) / 256;   // float4 fetches per thread and stage
};

template <int RW, int CBK, int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_3x3_mfma(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int Cin,
                                                          int Cout, int cinpad, int copad, int H, int W, int tiles_w, long long total, Epi2 epi) {
  using G = Geo3<RW, CBK, DIL>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 32, h0 = ht * G::kTH, co0 = blockIdx.y * G::kCO;
  const long long b = blockIdx.z;
  const long long HW = static_cast<long long>(H) * W;

  f32x16 acc[CBK][RW];
#pragma unroll
  for (int i = 0; i < CBK; ++i)
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][r][v] = 0.0f;

  // fetch plan, once per tile: slot s = tid + 256 i -> (channel c, tile row r, float4 group j).  The loads are UNCONDITIONAL (a load
  // inside a divergent branch makes the compiler wait for it at the join - before the matrix instructions): a slot outside the image
  // loads from an address clamped into the tensor and is zeroed element by element when it is committed to LDS.
  long long xflat[G::kXSlots];               // gh * W + gw of the group's first element (may lie outside the row / the image)
  int xc[G::kXSlots];
  unsigned xvm[G::kXSlots];                  // bit e: element e of the group is a pixel of the image
#pragma unroll
  for (int i = 0; i < G::kXSlots; ++i) {
    const int sidx = tid + 256 * i;
    const int j = sidx % (kLW / 4), r = (sidx / (kLW / 4)) % G::kRows, c = sidx / ((kLW / 4) * G::kRows);
    const int gh = h0 - DIL + r, gw = w0 - 4 + 4 * j;
    xc[i] = c < kC3 ? c : kC3 - 1;
    xflat[i] = static_cast<long long>(gh) * W + gw;
    unsigned vm = 0;
    if (sidx < G::kXN && gh >= 0 && gh < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) vm |= (gw + e >= 0 && gw + e < W) ? (1u << e) : 0u;
    xvm[i] = vm;
  }
  const long long xlast = total - 4;         // the last float4 that lies inside the tensor
  v4f rx[G::kXSlots], rw[G::kWSlots];
  int rxs[G::kXSlots];                       // how far the clamp moved the load (non-zero only in the tensor's first / last three floats)
  unsigned rxm[G::kXSlots];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < G::kXSlots; ++i) {
      const int ch = c0 + xc[i];
      const long long at = (b * Cin + (ch < Cin ? ch : Cin - 1)) * HW + xflat[i];
      const long long cl = at < 0 ? 0 : (at > xlast ? xlast : at);
      rx[i] = *reinterpret_cast<const v4f_u*>(x + cl);
      rxs[i] = static_cast<int>(cl - at);
      rxm[i] = ch < Cin ? xvm[i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < G::kWSlots; ++i) {
      int sidx = tid + 256 * i;
      if (sidx >= G::kWN) sidx -= 256;       // idle lanes of the last pass repeat a slot (same data to the same place)
      const int q = sidx % (G::kCO / 4), k = (sidx / (G::kCO / 4)) % kC3, tap = sidx / ((G::kCO / 4) * kC3);
      rw[i] = *reinterpret_cast<const v4f*>(wp + (static_cast<long long>(tap) * cinpad + c0 + k) * copad + co0 + 4 * q);
    }
  };
  auto commit = [&](int buf) {
    float* sx = lds + buf * (G::kSX + G::kSW);
    float* sw = sx + G::kSX;
#pragma unroll
    for (int i = 0; i < G::kXSlots; ++i) {
      const int sidx = tid + 256 * i;
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
      if (sidx < G::kXN) *reinterpret_cast<v4f*>(sx + 4 * sidx) = v;   // [c][r][40] row-major == float4 index sidx
    }
#pragma unroll
    for (int i = 0; i < G::kWSlots; ++i) {
      int sidx = tid + 256 * i;
      if (sidx >= G::kWN) sidx -= 256;
      *reinterpret_cast<v4f*>(sw + 4 * sidx) = rw[i];   // [tap][k][CO]
    }
  };
  auto products = [&](int st) {
    const float* sx = lds + (st & 1) * (G::kSX + G::kSW);
    const float* sw = sx + G::kSX;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int kk = 0; kk < kC3; kk += 2) {
        float a[CBK], bv[RW];
#pragma unroll
        for (int i = 0; i < CBK; ++i) a[i] = sw[(tap * kC3 + kk + half) * G::kCO + i * 32 + l32];
#pragma unroll
        for (int r = 0; r < RW; ++r) bv[r] = sx[((kk + half) * G::kRows + wave * RW + r + DIL * kh) * kLW + 4 + DIL * (kw - 1) + l32];
#pragma unroll
        for (int i = 0; i < CBK; ++i)
#pragma unroll
          for (int r = 0; r < RW; ++r) acc[i][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[r], acc[i][r], 0, 0, 0);
      }
    }
  };

  // the last stage is peeled off: a load and the LDS write that consumes it always sit in the same straight-line block (with
  // "if (more) fetch ... if (more) commit" the compiler assumes a path with a load still in flight at the loop's head and waits for
  // all loads before it reuses the registers)
  const int nstage = (Cin + kC3 - 1) / kC3;
  fetch(0);
  commit(0);
  __syncthreads();
  int st = 0;
  for (; st + 1 < nstage; ++st) {
    fetch((st + 1) * kC3);
    __builtin_amdgcn_sched_barrier(0);     // the loads stay AHEAD of the matrix instructions (the scheduler would sink them to their use)
    products(st);
    commit((st + 1) & 1);
    __syncthreads();
  }
  products(st);

  const long long MP = static_cast<long long>(Cout) * HW;
  float* yb = y + b * MP;
  const float* resb = epi.residual ? epi.residual + b * MP : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP : nullptr;
  const int gw = w0 + l32;
  if (gw >= W) return;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int gh = h0 + wave * RW + r;
    if (gh >= H) continue;
#pragma unroll
    for (int i = 0; i < CBK; ++i) {
      const int cbase = co0 + i * 32 + 4 * half;
      const long long px = static_cast<long long>(gh) * W + gw;
      float rv[16], mv[16];                       // skip connection and mask of the sixteen outputs first, unconditionally (see the 1x1 kernel)
#pragma unroll
      for (int v = 0; v < 16; ++v) rv[v] = 0.0f, mv[v] = 1.0f;
      if (resb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int co = cbase + (v & 3) + 8 * (v >> 2);
          rv[v] = __builtin_nontemporal_load(co < Cout ? resb + static_cast<long long>(co) * HW + px : epi.residual);
        }
      }
      if (maskb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int co = cbase + (v & 3) + 8 * (v >> 2);
          mv[v] = __builtin_nontemporal_load(co < Cout ? maskb + static_cast<long long>(co) * HW + px : epi.mask);
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co = cbase + (v & 3) + 8 * (v >> 2);
        if (co >= Cout) continue;
        const long long at = static_cast<long long>(co) * HW + px;
        float o = acc[i][r][v];
        if (epi.bias) o = o + epi.bias[co];
        if (resb) o = o + rv[v];
        if (epi.relu) o = o > 0.0f ? o : 0.0f;
        if (maskb) o = mv[v] > 0.0f ? o : 0.0f;
        yb[at] = o;
      }
    }
  }
}

__global__ void conv2d_3x3_prep_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int transpose, int kpad, int mpad) {
  // forward:  out[tap][k = ci][m = co] = w[co][ci][tap];  transpose (backward w.r.t. the input): out[tap][k = co][m = ci] = w[co][ci][8 - tap]
  const long long n = 9LL * kpad * mpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    const int m = static_cast<int>(i % mpad), k = static_cast<int>((i / mpad) % kpad), tap = static_cast<int>(i / (static_cast<long long>(mpad) * kpad));
    const int co = transpose ? k : m, ci = transpose ? m : k, t = transpose ? 8 - tap : tap;
    out[i] = (co < cout && ci < cin) ? w[(static_cast<long long>(co) * cin + ci) * 9 + t] : 0.0f;
  }
}

template <int RW, int CBK, int DIL>
int launch_3x3(const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int h, int w, const Epi2& epi, hipStream_t st) {
  using G = Geo3<RW, CBK, DIL>;
  const int tiles_w = (w + 31) / 32, tiles_h = (h + G::kTH - 1) / G::kTH;
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + G::kCO - 1) / G::kCO;
  if (tiles > 0x7fffffffLL || cgroups > 65535 || b > 65535) return ADV_EINVAL;
  const size_t lds = 2 * sizeof(float) * static_cast<size_t>(G::kSX + G::kSW);
  if (lds > 64 * 1024 && !adv_internal_lds_limit<conv2d_3x3_mfma<RW, CBK, DIL>>(lds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv2d_3x3_mfma<RW, CBK, DIL>), dim3(static_cast<unsigned>(tiles), cgroups, b), dim3(256), lds, st, x, wp, y, cin, cout, cinpad, copad,
                     h, w, tiles_w, static_cast<long long>(b) * cin * h * w, epi);
  return adv_internal_finish_launch();
}


// The epilogue of a convolution somebody else computed (MIOpen for the layer shapes where it is faster): y = [relu](y + bias[c] +
// residual) in ONE pass over the tensor, in place - torch runs it as up to three element-wise kernels (bias broadcast, skip add,
// clamp), each a full read + write.  The same float operations in the same order as the fused epilogues above.
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ residual, int C,
                                                       long long hw, int relu) {
  const long long plane = blockIdx.y;                       // b * C + c
  const float bv = bias ? bias[plane % C] : 0.0f;
  float* yp = y + plane * hw;
  const float* rp = residual ? residual + plane * hw : nullptr;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < hw; i += stride) {
    float v = yp[i];
    if (bias) v = v + bv;
    if (rp) v = v + rp[i];
    if (relu) v = v > 0.0f ? v : 0.0f;
    yp[i] = v;
  }
}

}  // namespace

extern "C" {

int64_t adv_conv2d_1x1_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return static_cast<int64_t>(round_up(k, kKC)) * round_up(m, 128);
}

int adv_conv2d_1x1_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up(k, kKC), mpad = round_up(m, 128);
  const long long n = static_cast<long long>(kpad) * mpad;
  const unsigned blocks = static_cast<unsigned>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
  hipLaunchKernelGGL(conv2d_1x1_prep_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, w_prep, cout, cin, transpose ? 1 : 0,
                     kpad, mpad);
  return adv_internal_finish_launch();
}

int adv_conv2d_1x1_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                       int cout, int64_t pixels, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || pixels < 1 || tile < -1 || tile > 5) return ADV_EINVAL;
  if (static_cast<long long>(b) * cin * pixels < 4) return ADV_EINVAL;      // the kernels load whole float4s (clamped into the tensor)
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  Epi2 epi{bias, residual, mask, relu ? 1 : 0};
  const int mpad = round_up(cout, 128);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int c = tile >= 0 ? tile : pick_1x1_tile(b, cout, pixels, 4 * cu_count());
  switch (c) {
    case 0: return launch_1x1<2, 2, 2, 4>(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);
    case 1: return launch_1x1<2, 2, 2, 2>(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);
    case 2: return launch_1x1<2, 2, 1, 2>(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);
    case 4: return launch_1x1<2, 1, 1, 1>(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);      // 64 x 32, two waves
    case 5: return launch_1x1_m16(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);              // 64 x 32, four waves of half-size units
    default: return launch_1x1<2, 2, 1, 1>(x, w_prep, y, b, cin, cout, mpad, pixels, epi, st);
  }
}

int64_t adv_conv2d_3x3_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 9LL * round_up(k, kC3) * round_up(m, 64);
}

int adv_conv2d_3x3_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up(k, kC3), mpad = round_up(m, 64);
  const long long n = 9LL * kpad * mpad;
  const unsigned blocks = static_cast<unsigned>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
  hipLaunchKernelGGL(conv2d_3x3_prep_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, w_prep, cout, cin, transpose ? 1 : 0,
                     kpad, mpad);
  return adv_internal_finish_launch();
}

int adv_conv2d_3x3_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                       int cout, int h, int w, int dilation, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || (dilation != 1 && dilation != 2) || tile < -1 || tile > 2)
    return ADV_EINVAL;
  if (static_cast<long long>(b) * cin * h * w < 4) return ADV_EINVAL;       // the kernels load whole float4s (clamped into the tensor)
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  Epi2 epi{bias, residual, mask, relu ? 1 : 0};
  const int cinpad = round_up(cin, kC3), copad = round_up(cout, 64);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // tile 0: 8 rows x 32 columns x 64 channels (two rows and two channel blocks per wave); tile 1: 16 rows x 32 columns x 32 channels
  // (four rows of one channel block per wave) - for layers of 32 output channels or fewer, where tile 0 would compute a zero block
  // tile 2: 4 rows x 32 columns x 64 channels (one row, two channel blocks per wave) - twice the workgroups of tile 0 for maps that
  // would leave compute units with one workgroup or none (64 channels at 96 x 312: 240 tiles of 8 rows for 256 CUs; a single
  // workgroup per CU has one wave per SIMD and nothing to overlap its LDS reads and barriers with)
  int t = tile;
  if (t < 0) {
    t = cout <= 32 ? 1 : 0;
    const long long n0 = static_cast<long long>((w + 31) / 32) * ((h + 7) / 8) * ((cout + 63) / 64) * b;
    if (t == 0 && n0 <= cu_count()) t = 2;       // (at 480 tiles the 8-row tile is still the faster one: measured on 128->128 at 96 x 312)
  }
  if (t == 2) {
    if (dilation == 1) return launch_3x3<1, 2, 1>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
    return launch_3x3<1, 2, 2>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
  }
  if (t == 0) {
    if (dilation == 1) return launch_3x3<2, 2, 1>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
    return launch_3x3<2, 2, 2>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
  }
  if (dilation == 1) return launch_3x3<4, 1, 1>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
  return launch_3x3<4, 1, 2>(x, w_prep, y, b, cin, cout, cinpad, copad, h, w, epi, st);
}

int adv_bias_act_f32(float* y, const float* bias, const float* residual, int64_t planes, int c, int64_t hw, int relu, adv_stream_t stream) {
  if (!y || planes < 1 || c < 1 || hw < 1 || planes > 0x7fffffffLL || residual == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(y) & 3) || (bias && (reinterpret_cast<uintptr_t>(bias) & 3)) || (residual && (reinterpret_cast<uintptr_t>(residual) & 3)))
    return ADV_EALIGN;
  if (planes > 65535) return ADV_EINVAL;
  const unsigned gx = static_cast<unsigned>((hw + 1023) / 1024 < 1 ? 1 : ((hw + 1023) / 1024 > 4096 ? 4096 : (hw + 1023) / 1024));
  hipLaunchKernelGGL(bias_act_kernel, dim3(gx, static_cast<unsigned>(planes)), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias, residual, c,
                     static_cast<long long>(hw), relu ? 1 : 0);
  return adv_internal_finish_launch();
}

}  // extern "C"
