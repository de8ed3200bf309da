// 3x3 (x3) / stride 1 / pad 1 convolutions by Winograd F(4x4, 3x3) on the gfx950 float32 matrix cores (v_mfma_f32_32x32x2_f32):
// 36 element-wise products per 4 x 4 outputs - 4x fewer multiply-adds than the direct kernels, 1.78x fewer than csrc/wino2d.hip's
// F(2x2, 3x3) - input transform, products, output transform and epilogue in ONE kernel (no transformed tensor reaches HBM).
//
//   Y(4x4) = A^T [ sum_q (G g_q G^T) .* (B^T d_q B) ] A      d_q: the 6 x 6 input patch of "channel" q around the 4 x 4 output block
//   2D layers: q = input channel.   3x3x3 layers: the transform in the (H, W) plane, the depth taps inside the contraction, q = (kd, c).
//
// Why the design differs from wino2d.hip.  Thirty-six products of a (channel block) x (patch block) tile need 36 accumulator tiles.  With
// 16x16 tiles and wino2d's scheme (every wave all positions of its own block) that is 144 registers per 16 x 16 block and two operand reads
// per matrix instruction - the LDS, not the matrix pipe, would set the pace.  Here the POSITIONS are dealt to the waves: a workgroup is
// EIGHT waves (two per SIMD, 256 registers each); with 64 output channels per workgroup (CB = 2) waves 0-3 carry four positions each
// (k = w + 4 n) and run the input transform, waves 4-7 five each; every wave holds its positions for ALL 64 channels x 32 patches of the
// tile (8 / 10 accumulators of 32 x 32 = 128 / 160 registers), one B operand shared by the two channel blocks: three LDS reads per two
// 64-cycle matrix instructions, a quarter of wino2d's operand traffic per matrix cycle.  With 32 output channels (CB = 1) the tile is 32
// channels x 64 patches and all eight waves transform.  (A first version - four waves, one per SIMD, nine positions = 288 accumulator
// registers each - kept part of the accumulators in AGPRs and shuffled them around every matrix instruction; two waves per SIMD also give
// the pipe something to run while the other wave transforms.)  The 36 values of one (channel, patch) sit in eight different waves after
// the contraction: they are exchanged through LDS in rounds of 16 channels and every thread transforms one or two (channel, patch) items
// per round (prologue + epilogue: 13 % of a tile at 256 channels, profiles/r05_wino4_phases.jsonl).
//
//   per stage of KC = 4 q:   input tile [4][4 PR + 2][LWP]   global -> registers (buffer loads, two sets deep) -> LDS
//                            U = G g G^T of the stage [36][4][64]   global -> LDS by LDS-DMA, each wave ITS OWN positions (1 KiB per instruction)
//                            V = B^T d B [36][4][32]   the transform waves: two threads per (q, patch), three of the six rows of V each
//   one "s_waitcnt; s_barrier" per stage (not __syncthreads(): its fence would drain the loads in flight), two buffers of everything, and
//   the side work woven between the matrix instructions in half-steps (one matrix instruction per scheduling step).
//   PAIR: two images of at most 15 columns side by side in one 32-column tile (the RoI heads' 14 x 14 maps).
//
// Order of float operations (oracle/oracle.c orc_conv_wino4 restates it bit for bit): the transforms' expressions as written below
// (explicit fmaf where a multiply feeds an add), each M_k one fmaf chain over q ascending starting from 0 (the matrix instruction is a
// k-ordered fmaf chain).  The backward w.r.t. the input is the same kernel on the transposed, flipped weights.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "adv_internal.h"
#include "advengine.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kKC = 4;        // "channels" q per stage
constexpr int kTransformWavePositions = 4;      // 64-channel workgroups: positions carried by each of the four transform waves (the others: 9 - this)

// PR x PC patches of 4 x 4 outputs per workgroup, CB blocks of 32 output channels: 32 patches x 64 channels (CB = 2) or 64 patches x 32
// channels (CB = 1: layers of 32 output channels or an odd number of 32-channel blocks); LWP: floats per LDS row of the input tile
template <int PR, int PC, int LWP, int CB>
struct W4Geo {
  static constexpr int kPB = 2 / CB, kNP = 32 * kPB, kCO = 32 * CB;   // patch blocks per wave tile, patches and output channels per workgroup
  static_assert((CB == 1 || CB == 2) && PR * PC == kNP, "32 patches x 64 channels or 64 patches x 32 channels per workgroup");
  static constexpr int kRows = 4 * PR + 2, kLW = 4 * PC + 8;      // input rows; loaded columns gw = w0 - 4 .. w0 + 4 PC + 3 (whole aligned float4 groups)
  static_assert(LWP >= kLW + 4 && LWP % 4 == 0, "row pitch (the tile sits one column to the right of the row's start)");
  static constexpr int kSX = kKC * kRows * LWP;                   // floats per input-tile buffer
  static constexpr int kXN = kKC * kRows * (kLW / 4);             // float4 groups per stage
  static constexpr int kXSl = (kXN + 511) / 512;
  static constexpr int kSW = 36 * kKC * kCO, kSV = 36 * kKC * kNP;
  static constexpr int kPairs = kKC * kNP;                        // (q, patch) pairs of the input transform per stage: two threads each
  static constexpr size_t kLds = 2 * sizeof(float) * (kSX + kSW + kSV);
  static_assert(kLds >= sizeof(float) * 36 * 16 * kNP && kLds <= 160 * 1024, "the exchange buffer of the epilogue fits; the CU's LDS holds the workgroup");
};

struct Epi4 {
  const float* bias;
  const float* residual;
  const float* mask;
  int relu;
};

// the six expressions of B^T applied to (d0 .. d5)
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float (&t)[6]) {
  t[0] = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
  t[1] = __builtin_fmaf(-4.0f, d1 + d2, d3 + d4);
  t[2] = __builtin_fmaf(4.0f, d1 - d2, d4 - d3);
  t[3] = __builtin_fmaf(2.0f, d3 - d1, d4 - d2);
  t[4] = __builtin_fmaf(2.0f, d1 - d3, d4 - d2);
  t[5] = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}

// A^T applied to (m0 .. m5) -> four outputs
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
  const float a = m1 + m2, b = m1 - m2, c = m3 + m4, e = m3 - m4;
  y[0] = (m0 + a) + c;
  y[1] = __builtin_fmaf(2.0f, e, b);
  y[2] = __builtin_fmaf(4.0f, c, a);
  y[3] = __builtin_fmaf(8.0f, e, b) + m5;
}

// ABL: phase ablation for timing (compile-time, so the schedule of the rest is the shipped one; results are wrong): bit 0 input transform,
// 1 matrix instructions, 2 input loads, 3 input commits, 4 the stage's wait + barrier, 5 operand reads, 6 weight DMA.
// The -DADV_TEST_HOOKS build's probe instantiates non-zero values, the shipped kernels are ABL = 0.
// PAIR (16 x 32-output tiles of 2D layers on maps of at most 15 x 16 pixels - the RoI heads' 14 x 14 maps): the tile's left and right halves are
// TWO IMAGES side by side (blockIdx.z = image pair): an image's right zero padding and its neighbour's left one coincide in the LDS tile, so
// a 14 x 14 map uses 77 % of the tile instead of 38 %.  Needs Cin % 4 == 0 (the range check covers the pair, not one image's channels).
template <int PR, int PC, int LWP, int CB, bool DEPTH, int ABL = 0, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void conv_wino4(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int Cin, int Cout,
                                                     int cinpad, int copad, int D, int H, int W, int tiles_w, long long wbytes, int nimg, int flags, Epi4 epi) {
  using G = W4Geo<PR, PC, LWP, CB>;
  constexpr int PB = G::kPB, NPT = G::kNP, CO = G::kCO;
  constexpr int dbg = ABL;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 4 * PC, h0 = ht * 4 * PR, co0 = blockIdx.y * CO;
  static_assert(!PAIR || (!DEPTH && PC == 8), "image pairs: the 16 x 32 tile of a 2D layer");
  const long long b = DEPTH ? blockIdx.z / D : (PAIR ? 2LL * blockIdx.z : blockIdx.z);      // PAIR: the first image of the pair
  const int od = DEPTH ? static_cast<int>(blockIdx.z % D) : 0;
  const long long HW = static_cast<long long>(H) * W;
  const long long DHW = HW * D;

  float* const sxb = lds;                          // [2][KC][rows][LWP]   input tiles
  float* const swb = lds + 2 * G::kSX;             // [2][36][KC][CO]      U of the stage
  float* const svb = swb + 2 * G::kSW;             // [2][36][KC][NPT]     V of the stage

  // ---- addressing: buffer loads, one 32-bit byte offset per slot computed once per tile, the hardware's range check = zero padding
  const float* const xb = x + b * Cin * DHW;
  const long long ximgs = PAIR ? (b + 1 < nimg ? 2 : 1) : 1;      // images behind the descriptor
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, static_cast<int>(static_cast<unsigned>(ximgs * Cin * DHW * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, static_cast<int>(wbytes), 0x00020000);
  int xvo[G::kXSl], xls[G::kXSl];
  unsigned xvm[G::kXSl];
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) {
    const int sidx = tid + 512 * i;
    const int j = sidx % (G::kLW / 4), r = (sidx / (G::kLW / 4)) % G::kRows, c = sidx / ((G::kLW / 4) * G::kRows);
    // groups start on multiples of four columns: none straddles the row's start.  PAIR: groups 0-4 are image A's columns -4 .. 15, groups
    // 5-9 image B's columns 0 .. 19 (LDS column = column + 5 / + 21: B's column -1 is A's column 15 - zero padding for both)
    const int gh = h0 - 1 + r, gw = PAIR ? (j < 5 ? 4 * j - 4 : 4 * (j - 5)) : w0 - 4 + 4 * j;
    xvo[i] = static_cast<int>(static_cast<unsigned>(c < kKC ? c : kKC - 1) * static_cast<unsigned>(DHW) * 4u) + (gh * W + gw) * 4 +
             (PAIR && j >= 5 ? static_cast<int>(static_cast<unsigned>(Cin * DHW) * 4u) : 0);
    xls[i] = ((c < kKC ? c : kKC - 1) * G::kRows + r) * LWP + 4 * j + 1;    // LDS column = gw - (w0 - 5): a patch's six columns start on a multiple of four
    unsigned vm = 0;
    if (sidx < G::kXN && gh >= 0 && gh < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) vm |= (gw + e >= 0 && gw + e < W) ? (1u << e) : 0u;
    xvm[i] = vm | (sidx < G::kXN ? 16u : 0u);                // bit 4: the slot exists
  }
  struct XSet {
    v4f v[G::kXSl];
  };
  auto stage_off = [&](int q0) -> unsigned {      // wave-uniform: bytes from (channel 0, plane 0) to (the stage's first channel, its plane)
    const int kd = DEPTH ? q0 / cinpad : 0;
    const int ch = DEPTH ? q0 - kd * cinpad : q0;
    return static_cast<unsigned>((static_cast<long long>(ch) * D + (DEPTH ? od + kd - 1 : 0)) * HW * 4);
  };
  auto fetch_x1 = [&](int i, int q0, XSet& set) {
    set.v[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, xvo[i] + static_cast<int>(stage_off(q0)), 0, 0));
  };
  auto commit_x1 = [&](int i, int buf, const XSet& set) {
    v4f v = set.v[i];
    const unsigned m = xvm[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (m >> e) & 1u ? v[e] : 0.0f;
    if (m & 16u) {        // the group lands one column off a 16-byte boundary (so that the transform reads whole aligned pieces): 4 + 8 + 4 bytes
      float* p = sxb + buf * G::kSX + xls[i];
      p[0] = v[0];
      *reinterpret_cast<v2f*>(p + 1) = v2f{v[1], v[2]};
      p[3] = v[3];
    }
  };
  // The 36 positions are dealt to the eight waves so that the two waves of a SIMD carry nine between them: waves 0-3 own NT each,
  // k = w + 4 n; waves 4-7 own 9 - NT each, k = 4 NT + (w - 4) + 4 n.  NT = 4 / 5 everywhere (64-channel workgroups: waves 0-3 also compute the
  // whole input transform; giving them three positions and the others six measured 2-5 % slower, profiles/r05_wino4_split36_negative.jsonl).
  constexpr int NT = CB == 2 ? kTransformWavePositions : 4;      // positions of a wave 0-3 (waves 4-7: 9 - NT)
  const int kbase = wave < 4 ? wave : 4 * NT - 4 + wave;
  // weights: a wave stages the positions it multiplies itself: U[k][q0 .. q0 + 3][co0 .. co0 + CO - 1] = 1 KiB (CO = 64) or 512 B, one
  // LDS-DMA instruction (lane L: row L / (CO / 4), float4 L % (CO / 4); the upper half of the wave idle at CO = 32), straight into the
  // stage buffer - no staging registers, no commit, no other wave involved
  const int wrows = DEPTH ? 3 * cinpad : cinpad;           // rows of U per transform position
  const int wvo = ((lane / (CO / 4)) * copad + co0 + 4 * (lane % (CO / 4))) * 4;
  auto dma_w1 = [&](int n, int q0, int buf) {
    const int k = kbase + 4 * n;
    float* dst = swb + buf * G::kSW + k * (kKC * CO);
    if (CB == 2 || lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwgt, (lds_void*)dst, 16, wvo, (k * wrows + q0) * copad * 4, 0, 0);
  };

  // ---- the input transform: two threads per (q, patch) pair, three of the six rows of V each (hs = 0: rows 0-2, 1: rows 3-5).
  // CB = 2 (128 pairs): waves 0, 1 / 2, 3; waves 4-7 do not transform (they carry five positions).  CB = 1 (256 pairs): waves 0-3 / 4-7.
  const int pair = tid % G::kPairs, tc = pair / NPT, tp = pair % NPT;
  const int hs = CB == 2 ? (wave >> 1) & 1 : wave >> 2;
  const int toff = (tc * G::kRows + 4 * (tp / PC) + hs) * LWP + 4 * (tp % PC) + 4;     // first needed row: 0 (rows 0-4) or 1 (rows 1-5)
  const int voff = tc * NPT + tp;
  float td[5][6], tt[3][6], tv[18];
  auto tr_read = [&](int i, int buf) {
    const float* p = sxb + buf * G::kSX + toff + i * LWP;          // six columns: one 16-byte and one 8-byte read, both aligned
    const v4f m = *reinterpret_cast<const v4f*>(p);
    const v2f n = *reinterpret_cast<const v2f*>(p + 4);
    td[i][0] = m[0], td[i][1] = m[1], td[i][2] = m[2], td[i][3] = m[3], td[i][4] = n[0], td[i][5] = n[1];
  };
  auto tr_cols = [&](auto hs_c, int j) {          // column pass, column j: the thread's three rows of T = B^T d
    constexpr bool kHi = decltype(hs_c)::value;
    const float e0 = td[0][j], e1 = td[1][j], e2 = td[2][j], e3 = td[3][j], e4 = td[4][j];
    if constexpr (!kHi) {       // rows 0, 1, 2 from d0 .. d4
      tt[0][j] = __builtin_fmaf(4.0f, e0, __builtin_fmaf(-5.0f, e2, e4));
      tt[1][j] = __builtin_fmaf(-4.0f, e1 + e2, e3 + e4);
      tt[2][j] = __builtin_fmaf(4.0f, e1 - e2, e4 - e3);
    } else {                    // rows 3, 4, 5 from d1 .. d5 (e0 = d1)
      tt[0][j] = __builtin_fmaf(2.0f, e2 - e0, e3 - e1);
      tt[1][j] = __builtin_fmaf(2.0f, e0 - e2, e3 - e1);
      tt[2][j] = __builtin_fmaf(4.0f, e0, __builtin_fmaf(-5.0f, e2, e4));
    }
  };
  auto tr_row = [&](int i) {                      // row pass of the thread's row i (0 .. 2)
    float o[6];
    bt6(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) tv[6 * i + j] = o[j];
  };
  auto tr_write = [&](int idx, int buf) {
    const int k = (3 * hs + idx / 6) * 6 + idx % 6;
    svb[buf * G::kSV + k * (kKC * NPT) + voff] = tv[idx];
  };

  // ---- operands of the matrix instructions: A = U_k[q = 2 kp + half][co = 32 cb + l32], B = V_k[q][patch = 32 pb + l32]
  const int aoff = (kbase * kKC + half) * CO + l32, boff = (kbase * kKC + half) * NPT + l32;

  const int q_lo = DEPTH ? (od == 0 ? cinpad : 0) : 0;
  const int q_hi = DEPTH ? (od == D - 1 ? 2 * cinpad : 3 * cinpad) : cinpad;
  const int nstage = (q_hi - q_lo) / kKC;
  auto qclamp = [&](int q) { return q < q_hi - kKC ? q : q_hi - kKC; };

  // epilogue geometry: the exchange rounds hand every thread (channel of the round's 16, patch) items - one at 32 patches, two at 64
  const long long MP = static_cast<long long>(Cout) * DHW;
  const long long plane0 = static_cast<long long>(od) * HW;
  float* const yb = y + b * MP + plane0;
  const float* const resb = epi.residual ? epi.residual + b * MP + plane0 : nullptr;
  const float* const maskb = epi.mask ? epi.mask + b * MP + plane0 : nullptr;
  const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(epi.residual) | reinterpret_cast<uintptr_t>(epi.mask)) & 15) == 0;
  const int ep = tid % NPT, eg = tid / NPT;               // patch; first channel of the round's 16 (then + 512 / NPT)
  const int gh0 = h0 + 4 * (ep / PC), gw0 = PAIR ? 4 * (ep % 4) : w0 + 4 * (ep % PC);
  const long long eimg = PAIR && (ep % PC) >= 4 ? MP : 0;      // PAIR: the right half of the tile is the pair's second image
  const bool eok = !PAIR || (ep % PC) < 4 || b + 1 < nimg;

  // body<NP, TR, HS>: a wave's whole life after the set-up - NP positions; TR: it also computes the input transform (half HS of it)
  auto body = [&](auto np_c, auto t_c, auto hs_c) __attribute__((always_inline)) {
    constexpr int NP = decltype(np_c)::value;
    constexpr bool TR = decltype(t_c)::value;
    if (CB == 2 && ((TR && (flags & 1)) || (!TR && (flags & 2)))) __builtin_amdgcn_s_setprio(1);      // static priority of one half of the waves (A/B: kWaveFlags)
    constexpr int NS = 2 * NP;                    // steps per stage: (position n, q pair kp), two matrix instructions each
    f32x16 acc[NP][2];                            // [n][cb] (CB = 2) or [n][pb] (CB = 1)
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[n][c][v] = 0.0f;
    XSet xsA, xsB;
    {
      // prologue: the first three input tiles and the first weights are requested together
      XSet xs0, xs1;
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, q_lo, xs0);
#pragma unroll
      for (int n = 0; n < NP; ++n) dma_w1(n, q_lo, 0);
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, qclamp(q_lo + kKC), xs1);
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, qclamp(q_lo + 2 * kKC), xsB);
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 0, xs0);
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 1, xs1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if constexpr (TR) {
#pragma unroll
        for (int i = 0; i < 5; ++i) tr_read(i, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) tr_cols(hs_c, j);
#pragma unroll
        for (int i = 0; i < 3; ++i) tr_row(i);
#pragma unroll
        for (int k = 0; k < 18; ++k) tr_write(k, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // Stage st: NS steps of two matrix instructions each (two channel blocks sharing the B operand, or two patch blocks sharing the A
    // operand) on U / V of stage st.  Woven between them, one piece per step: the LDS-DMA of the wave's weights of stage st + 1, the loads
    // of the input tile of stage st + 3, the commit of the input tile of stage st + 2 (requested during stage st - 1) and - transform
    // waves, eight steps - the input transform of stage st + 1.  The scheduler may not move anything across a step.  ONE barrier per stage;
    // before it the wave's LDS-DMAs have landed (vmcnt: all but the stage's own input loads, issued after them, which travel on).
    auto stage = [&](int st, XSet& fxs, const XSet& cxs) __attribute__((always_inline)) {
      constexpr int kAhead = 3;
      float r0[NS], r1[NS], r2[NS];               // CB = 2: A (block 0), A (block 1), B;  CB = 1: A, B (block 0), B (block 1)
      if constexpr ((dbg & 32) != 0) {
#pragma unroll
        for (int t = 0; t < NS; ++t) r0[t] = r1[t] = r2[t] = static_cast<float>(lane + t);
      }
      const float* ap = swb + (st & 1) * G::kSW + aoff;
      const float* bp = svb + (st & 1) * G::kSV + boff;
      auto load = [&](int t) {
        const int row = (4 * (t >> 1)) * kKC + 2 * (t & 1);      // (k - kbase) * KC + 2 kp
        r0[t] = ap[row * CO];
        r1[t] = CB == 2 ? ap[row * CO + 32] : bp[row * NPT];
        r2[t] = CB == 2 ? bp[row * NPT] : bp[row * NPT + 32];
      };
      const int nb = (st + 1) & 1;
      const int q1 = qclamp(q_lo + (st + 1) * kKC), q3 = qclamp(q_lo + (st + 3) * kKC);
#pragma unroll
      for (int t = 0; t < kAhead; ++t)
        if (!(dbg & 32)) load(t);
      // HALF-steps: exactly one matrix instruction each, the side work in front of it.  A wave issues in order: with both matrix
      // instructions of a step back to back the second waits out the first's 64 cycles with the wave stalled behind it, so only every
      // other matrix instruction had side work in its shadow (profiles/r05_wino4_phases.jsonl: the two did not overlap at all).
      // Schedule (transform waves): rows 0-1 | 2-3 | 4 | columns 0-1 | 2-3 | 4-5 | V row 0 | V row 1 + writes | V row 2 + writes | writes x 3,
      // the weight DMAs on half-steps 0 .. NP-1, the input loads on 9 .., the commits on the last ones.
#pragma unroll
      for (int h = 0; h < 2 * NS; ++h) {
        const int t = h >> 1;
        if ((h & 1) == 0 && t + kAhead < NS && !(dbg & 32)) load(t + kAhead);
        if (h < NP && !(dbg & 64)) dma_w1(h, q1, nb);
        if (h >= 9 && h < 9 + G::kXSl && !(dbg & 4)) fetch_x1(h - 9, q3, fxs);
        if (h >= 2 * NS - 1 - G::kXSl && h < 2 * NS - 1 && !(dbg & 8)) commit_x1(h - (2 * NS - 1 - G::kXSl), st & 1, cxs);
        if constexpr (TR && !(dbg & 1)) {
          if (h == 0) tr_read(0, nb), tr_read(1, nb);
          if (h == 1) tr_read(2, nb), tr_read(3, nb);
          if (h == 2) tr_read(4, nb);
          if (h >= 3 && h < 6) tr_cols(hs_c, 2 * (h - 3)), tr_cols(hs_c, 2 * (h - 3) + 1);
          if (h >= 6 && h < 9) tr_row(h - 6);
          if (h >= 7 && h < 12) {
            constexpr int kPer[5] = {3, 3, 4, 4, 4};      // 18 writes over half-steps 7-11 (row i's values exist from half-step 6 + i on)
            int first = 0;
#pragma unroll
            for (int u = 7; u < h; ++u) first += kPer[u - 7];
#pragma unroll
            for (int u = 0; u < kPer[h - 7]; ++u) tr_write(first + u, nb);
          }
        }
        if constexpr ((dbg & 2) != 0) {
        } else if constexpr (CB == 2) {
          if ((h & 1) == 0) acc[t >> 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0[t], r2[t], acc[t >> 1][0], 0, 0, 0);
          else acc[t >> 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1[t], r2[t], acc[t >> 1][1], 0, 0, 0);
        } else {
          if ((h & 1) == 0) acc[t >> 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0[t], r1[t], acc[t >> 1][0], 0, 0, 0);
          else acc[t >> 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0[t], r2[t], acc[t >> 1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the stage's LDS-DMAs have landed (issued before its input loads, which travel on: the counter retires in order), the LDS writes
      // are done; a plain __syncthreads() would wait for ALL memory operations - the input loads too
      if constexpr (!(dbg & 16)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::kXSl) : "memory");
    };
    {
      int st = 0;
      for (; st + 1 < nstage; st += 2) {
        stage(st, xsA, xsB);
        stage(st + 1, xsB, xsA);
      }
      if (st < nstage) stage(st, xsA, xsB);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (clamped requests of the last stages: nothing may land later)

    // ---- epilogue: the 36 values M_k of one (channel, patch) sit in eight waves - exchange through LDS, 16 channels per round:
    // E[k][co16][patch]; register v of a 32 x 32 accumulator = channel (v & 3) + 8 (v >> 2) + 4 half of its block, patch = lane & 31
    float* const se = lds;
#pragma unroll
    for (int round = 0; round < 2 * CB; ++round) {          // (unrolled: the accumulator registers are addressed by constants)
      if (co0 + 16 * round >= Cout) continue;               // (workgroup-uniform) nothing but padding from here on
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        const int k = kbase + 4 * n;
#pragma unroll
        for (int v8 = 0; v8 < 8; ++v8) {
          const int co16 = (v8 & 3) + 8 * (v8 >> 2) + 4 * half;
          if constexpr (CB == 2) {
            se[(k * 16 + co16) * NPT + l32] = acc[n][round >> 1][8 * (round & 1) + v8];
          } else {
            se[(k * 16 + co16) * NPT + l32] = acc[n][0][8 * round + v8];
            se[(k * 16 + co16) * NPT + 32 + l32] = acc[n][1][8 * round + v8];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int item = 0; item < PB; ++item) {
        const int co16 = eg + (512 / NPT) * item;
        const int co = co0 + 16 * round + co16;
        float s[4][6], o[4][4];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          float col[4];
          at6(se[((0 * 6 + j) * 16 + co16) * NPT + ep], se[((1 * 6 + j) * 16 + co16) * NPT + ep], se[((2 * 6 + j) * 16 + co16) * NPT + ep],
              se[((3 * 6 + j) * 16 + co16) * NPT + ep], se[((4 * 6 + j) * 16 + co16) * NPT + ep], se[((5 * 6 + j) * 16 + co16) * NPT + ep], col);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r][j] = col[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) at6(s[r][0], s[r][1], s[r][2], s[r][3], s[r][4], s[r][5], o[r]);
        if (co < Cout && gh0 < H && gw0 < W && eok) {
          const float bv = epi.bias ? epi.bias[co] : 0.0f;
          const long long at0 = eimg + static_cast<long long>(co) * DHW + static_cast<long long>(gh0) * W + gw0;
          // the skip connection and the mask of all sixteen outputs FIRST, the stores after them: a load placed behind a store to y may
          // not be moved ahead of it (the pointers could alias) - row by row that was a memory round trip per row (+25 % on the layers with
          // a residual); unconditional loads (a row / column outside the map reads the tensor's first floats and is never stored)
          float rv[4][4], mv[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool rok = gh0 + r < H;
            const long long at = at0 + static_cast<long long>(r) * W;
            if (vec4) {
              const v4f tr = resb ? *reinterpret_cast<const v4f*>(rok ? resb + at : epi.residual) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
              const v4f tm = maskb ? *reinterpret_cast<const v4f*>(rok ? maskb + at : epi.mask) : v4f{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
              for (int c = 0; c < 4; ++c) rv[r][c] = tr[c], mv[r][c] = tm[c];
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const bool ok = rok && gw0 + c < W;
                rv[r][c] = resb ? (ok ? resb[at + c] : epi.residual[0]) : 0.0f;
                mv[r][c] = maskb ? (ok ? maskb[at + c] : epi.mask[0]) : 1.0f;
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (gh0 + r >= H) continue;
            const long long at = at0 + static_cast<long long>(r) * W;
            float res[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float v = o[r][c];
              if (epi.bias) v = v + bv;
              if (resb) v = v + rv[r][c];
              if (epi.relu) v = v > 0.0f ? v : 0.0f;
              if (maskb) v = mv[r][c] > 0.0f ? v : 0.0f;
              res[c] = v;
            }
            if (vec4) {
              *reinterpret_cast<v4f*>(yb + at) = v4f{res[0], res[1], res[2], res[3]};
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (gw0 + c < W) yb[at + c] = res[c];
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  };
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;
  using IT = std::integral_constant<int, NT>;
  using IP = std::integral_constant<int, 9 - NT>;
  if constexpr (CB == 2) {
    if (wave >= 4) body(IP{}, std::false_type{}, std::false_type{});
    else if (hs) body(IT{}, std::true_type{}, std::true_type{});
    else body(IT{}, std::true_type{}, std::false_type{});
  } else {
    if (wave >= 4) body(I5{}, std::true_type{}, std::true_type{});
    else body(I4{}, std::true_type{}, std::false_type{});
  }
}

int round_up4(int v, int q) { return (v + q - 1) / q * q; }

constexpr int kWaveFlags = 0;      // bit 0: the transform waves of a 64-channel workgroup run at priority 1, bit 1: the product waves
int wave_flags() {
  if (const char* e = adv_hook_value("ADV_WINO4_FLAGS")) return std::atoi(e);      // test hook / A-B; same bits
  return kWaveFlags;
}

template <int PR, int PC, int LWP, int CB, bool DEPTH, bool PAIR = false>
int launch_wino4(const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int d, int h, int w, const Epi4& epi,
                 hipStream_t st) {
  using G = W4Geo<PR, PC, LWP, CB>;
  if (PAIR && (w > 15 || cin % kKC != 0)) return ADV_EINVAL;
  const int tiles_w = PAIR ? 1 : (w + 4 * PC - 1) / (4 * PC), tiles_h = (h + 4 * PR - 1) / (4 * PR);
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + G::kCO - 1) / G::kCO;
  const long long gz = PAIR ? (static_cast<long long>(b) + 1) / 2 : static_cast<long long>(b) * d;
  if (tiles > 0x7fffffffLL || cgroups > 65535 || gz > 65535) return ADV_EINVAL;
  const long long wbytes = 36LL * (DEPTH ? 3 : 1) * cinpad * copad * 4;
  if ((static_cast<long long>(cinpad) + 1) * d * h * w * 4 * (PAIR ? 2 : 1) >= 0xfff00000LL || wbytes >= 0x7ff00000LL) return ADV_EINVAL;
  const dim3 grid(static_cast<unsigned>(tiles), cgroups, static_cast<unsigned>(gz));
#ifdef ADV_TEST_HOOKS
  if (const char* dbg_s = adv_hook_value("ADV_WINO4_DBG")) {      // phase ablation for timing (results are wrong): the 8 x 64 x 64 2D shape and the 16 x 64 x 32 3D shape
    if constexpr (!PAIR && ((PR == 2 && PC == 16 && CB == 2 && !DEPTH) || (PR == 4 && PC == 16 && CB == 1 && DEPTH))) {
      const int abl = std::atoi(dbg_s);
#define ADV_WINO4_ABL(A_)                                                                                                                       \
  if (abl == A_) {                                                                                                                              \
    if (!adv_internal_lds_limit<conv_wino4<PR, PC, LWP, CB, DEPTH, A_>>(G::kLds)) return ADV_ELAUNCH;                                           \
    hipLaunchKernelGGL((conv_wino4<PR, PC, LWP, CB, DEPTH, A_>), grid, dim3(512), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, tiles_w, \
                       wbytes, b, wave_flags(), epi);                                                                                                            \
    return adv_internal_finish_launch();                                                                                                        \
  }
      ADV_WINO4_ABL(1) ADV_WINO4_ABL(2) ADV_WINO4_ABL(4) ADV_WINO4_ABL(8) ADV_WINO4_ABL(16) ADV_WINO4_ABL(32) ADV_WINO4_ABL(64) ADV_WINO4_ABL(12) ADV_WINO4_ABL(93) ADV_WINO4_ABL(125) ADV_WINO4_ABL(127)
#undef ADV_WINO4_ABL
    }
  }
#endif
  if (!adv_internal_lds_limit<conv_wino4<PR, PC, LWP, CB, DEPTH, 0, PAIR>>(G::kLds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv_wino4<PR, PC, LWP, CB, DEPTH, 0, PAIR>), grid, dim3(512), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, tiles_w, wbytes,
                     b, wave_flags(), epi);
  return adv_internal_finish_launch();
}

// tile: 0 = 16 x 32 outputs x 64 channels (4 x 8 patches), 1 = 8 x 64 x 64 (2 x 16 patches: maps of few rows), 2 = 32 x 32 outputs x 32
// channels (8 x 8 patches), 3 = 16 x 64 x 32 (4 x 16 patches).  The 32-channel shapes where 64 channels would compute a block of padding
// (an odd number of 32-channel blocks); the flat shapes where they need clearly fewer workgroups for the map.
int pick_wino4_tile(int cout, int h, int w, long long bz, bool pair_ok) {
  // every shape is one workgroup per compute unit with the same work: the launch takes ceil(workgroups / 256) rounds; among equal rounds the
  // shape with fewer workgroups (less padding), then the wide one (8 x 64 / 16 x 64: measured 2-7 % faster on large maps, profiles/r05_wino4_*.jsonl)
  auto wgs = [&](int th, int tw, int co) { return static_cast<long long>((h + th - 1) / th) * ((w + tw - 1) / tw) * ((cout + co - 1) / co) * bz; };
  const bool narrow = ((cout + 31) / 32) % 2 == 1;
  if (pair_ok && !narrow) return 4;      // maps of at most 15 columns (the RoI heads' 14 x 14): two images per tile
  const long long a = narrow ? wgs(32, 32, 32) : wgs(16, 32, 64), b = narrow ? wgs(16, 64, 32) : wgs(8, 64, 64);
  const long long ra = (a + 255) / 256, rb = (b + 255) / 256;
  const bool wide = rb < ra || (rb == ra && b <= a);
  return (narrow ? 2 : 0) + (wide ? 1 : 0);
}

template <bool DEPTH>
int launch_wino4_tile(int t, const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int d, int h, int w,
                      const Epi4& epi, hipStream_t st) {
  switch (t) {
    case 0: return launch_wino4<4, 8, 56, 2, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 1: return launch_wino4<2, 16, 80, 2, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 2: return launch_wino4<8, 8, 44, 1, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 3: return launch_wino4<4, 16, 80, 1, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    default:
      if constexpr (!DEPTH) return launch_wino4<4, 8, 56, 2, false, true>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
      return ADV_EINVAL;
  }
}

int check_wino4_args(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, const float* y) {
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  return ADV_OK;
}

constexpr int kCO = 64;       // the prepared weights are padded to multiples of 64 output channels (both workgroup shapes read them)

// U = G g G^T (6 x 6) for every (output, input) channel pair (and depth tap), laid out [k = 6 i + j][kd][c'][m'] (zero rows / columns of
// padding).  forward: m = co, c = ci, g = w[co][ci][kd];  transpose (backward w.r.t. the input): m = ci, c = co, g = w[co][ci] with all
// its taps reversed.  taps = 1: a 2D layer's [Cout][Cin][3][3] weights.
__device__ __forceinline__ void g6(float g0, float g1, float g2, float (&t)[6]) {
  const float w6 = -1.0f / 6.0f, w24 = 1.0f / 24.0f, w12 = 1.0f / 12.0f, w6p = 1.0f / 6.0f;
  t[0] = g0 * 0.25f;
  t[1] = ((g0 + g1) + g2) * w6;
  t[2] = ((g0 - g1) + g2) * w6;
  t[3] = __builtin_fmaf(g0, w24, __builtin_fmaf(g1, w12, g2 * w6p));
  t[4] = __builtin_fmaf(g0, w24, __builtin_fmaf(-g1, w12, g2 * w6p));
  t[5] = g2;
}

__global__ void conv_wino4_prep_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps, int transpose, int kpad,
                                       int mpad) {
  const long long n = static_cast<long long>(taps) * kpad * mpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    const int m = static_cast<int>(i % mpad), c = static_cast<int>((i / mpad) % kpad), kd = static_cast<int>(i / (static_cast<long long>(mpad) * kpad));
    const int co = transpose ? c : m, ci = transpose ? m : c;
    float u[36];
    if (co < cout && ci < cin) {
      const float* gp = w + ((static_cast<long long>(co) * cin + ci) * taps + (transpose ? taps - 1 - kd : kd)) * 9;
      float g[9], t[6][3];
#pragma unroll
      for (int q = 0; q < 9; ++q) g[q] = gp[transpose ? 8 - q : q];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float col[6];
        g6(g[j], g[3 + j], g[6 + j], col);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][j] = col[r];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float row[6];
        g6(t[r][0], t[r][1], t[r][2], row);
#pragma unroll
        for (int j = 0; j < 6; ++j) u[6 * r + j] = row[j];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 36; ++q) u[q] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) out[q * n + i] = u[q];
  }
}

int prep_wino4(const float* w, float* w_prep, int cout, int cin, int taps, int transpose, hipStream_t st) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up4(k, kKC), mpad = round_up4(m, kCO);
  const long long n = static_cast<long long>(taps) * kpad * mpad;
  const unsigned blocks = static_cast<unsigned>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
  hipLaunchKernelGGL(conv_wino4_prep_kernel, dim3(blocks), dim3(256), 0, st, w, w_prep, cout, cin, taps, transpose ? 1 : 0, kpad, mpad);
  return adv_internal_finish_launch();
}

}  // namespace

extern "C" {

int64_t adv_conv2d_wino4_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 36LL * round_up4(k, kKC) * round_up4(m, kCO);
}

int adv_conv2d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino4(w, w_prep, cout, cin, 1, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv2d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                         int cout, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || tile < -1 || tile > 4) return ADV_EINVAL;
  if (const int rc = check_wino4_args(x, w_prep, bias, residual, mask, y)) return rc;
  const Epi4 epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino4_tile<false>(tile >= 0 ? tile : pick_wino4_tile(cout, h, w, b, w <= 15 && cin % kKC == 0 && b >= 2), x, w_prep, y, b, cin, cout, round_up4(cin, kKC), round_up4(cout, kCO), 1, h, w,
                                  epi, static_cast<hipStream_t>(stream));
}

int64_t adv_conv3d_wino4_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 108LL * round_up4(k, kKC) * round_up4(m, kCO);
}

int adv_conv3d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino4(w, w_prep, cout, cin, 3, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv3d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                         int cout, int d, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1 || tile < -1 || tile > 3) return ADV_EINVAL;
  if (const int rc = check_wino4_args(x, w_prep, bias, residual, mask, y)) return rc;
  const Epi4 epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino4_tile<true>(tile >= 0 ? tile : pick_wino4_tile(cout, h, w, static_cast<long long>(b) * d, false), x, w_prep, y, b, cin, cout, round_up4(cin, kKC), round_up4(cout, kCO), d, h, w,
                                 epi, static_cast<hipStream_t>(stream));
}

}  // extern "C"
