// 3x3 (x3) / stride 1 / pad 1 convolutions by Winograd F(2x2, 3x3) with the sixteen element-wise products on the gfx950 float32 matrix
// cores (v_mfma_f32_16x16x4_f32) - 2.25x fewer multiply-adds than the direct kernels of conv2d.hip / conv3d.hip, everything in ONE
// kernel (no transformed tensor ever reaches HBM).
//
//   Y(2x2) = A^T [ sum_q (G g_q G^T) .* (B^T d_q B) ] A      d_q: the 4x4 input patch of "channel" q around the 2x2 output block
//
//   2D layers: q = input channel.   3D layers (3x3x3 kernels): the transform is applied in the (H, W) plane only and the depth taps are
//   part of the contraction, q = (kd, channel): output plane od reads the input planes od - 1, od, od + 1 (a missing plane is skipped:
//   its terms are exact zeros) - 48 products per 2x2 outputs instead of 108.
//
// A workgroup owns up to 64 patches (4 x 16 patches = 8 x 32 outputs, 8 x 8 = 16 x 16 outputs, or 5 x 12 = 10 x 24 for maps of 10 / 20 rows)
// of one image / output plane and
//   shape A: 64 output channels, 512 threads, stages of 8 q   (155 KB LDS, one workgroup per CU)
//   shape B: 32 output channels, 256 threads, stages of 4 q   ( 61 KB LDS, two workgroups per CU) - layers of 32 channels or fewer.
// Wave w owns channel block w % COB (16 channels) and two blocks of 16 patches: 2 x 16 accumulators of 4 registers; A operand =
// U_k[co = lane & 15][q = lane >> 4], B operand = V_k[q = lane >> 4][patch = lane & 15].  All 16 values M_k of one (channel, patch) end
// up in the SAME lane and register index of the 16 accumulators, so the output transform A^T M A is plain per-lane arithmetic; the
// epilogue (+ bias, + residual, ReLU, mask) follows it before the store.
//
// Per stage: the input tile [KC][rows][LW] and the transformed weights U [16][KC][CO] go global -> registers -> LDS, every thread
// transforms one (q, patch) - 8 LDS reads of two floats, 32 additions, 16 LDS writes into V [16][KC][64] - and the waves run the
// matrix instructions on the PREVIOUS stage's V and U.  All of that is software-pipelined by hand into one instruction stream (see
// `products`): one barrier per stage, two buffers of everything.  LDS rows of U and V are not padded: column ^ ((q & 3) << 4) puts
// the four 16-lane groups of a read (four consecutive q) into different banks.
//
// Order of operations (oracle/oracle.c orc_conv2d_wino / orc_conv3d_wino restate it bit for bit): the transforms' additions as written
// below, the accumulation of each M_k one fmaf per q in ascending order starting from 0 (the matrix instruction is a k-ordered fmaf
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

// PR x PC patches per workgroup (4 x 16, 8 x 8, or 5 x 12 for maps of 10 / 20 rows); COB blocks of 16 output channels; KC "channels" per stage
template <int PR, int PC, int COB, int KC>
struct WGeo {
  static_assert(PR * PC <= 64 && PR * PC > 48, "up to 64 patches per workgroup (5 x 12 leaves four idle)");
  static_assert(KC * 64 == COB * 128, "one (q, patch) of the input transform per thread");
  static constexpr int kNT = COB * 128;                            // threads: COB channel blocks x 2 patch halves, one wave each
  static constexpr int kCO = 16 * COB;
  static constexpr int kRows = 2 * PR + 2, kLW = 2 * PC + 8;       // input rows; LDS row: column = gw - (w0 - 5), so a patch row starts on an even column
  static constexpr int kSX = KC * kRows * kLW, kXN = kSX / 4, kXSl = (kXN + kNT - 1) / kNT;
  static constexpr int kSW = 16 * KC * kCO, kWN = kSW / 4, kWSl = kWN / kNT;
  static constexpr int kSV = 16 * KC * 64;
  static constexpr int kSteps = (KC / 4) * 16;                     // pairs of matrix instructions per stage
  // where a stage's side work sits among the steps: loads first, then the next stage's input transform, the LDS commits last
  static constexpr int kLoadW = kXSl, kRead = kXSl + kWSl, kComp = kRead + 4 + (kSteps == 32 ? 2 : 0), kWrite = kComp + (kSteps == 32 ? 2 : 1);
  static constexpr int kWritesPerStep = kSteps == 32 ? 2 : 4;
  static constexpr int kCommitX = kSteps - kXSl - kWSl - (kSteps == 32 ? 2 : 0), kCommitW = kCommitX + kXSl;
  static_assert(kWrite + 16 / kWritesPerStep <= kCommitX + 1, "the schedule fits");
  static constexpr size_t kLds = 2 * sizeof(float) * (kSX + kSW + kSV);
};

constexpr int kDeep64Stages = 8;      // 64-channel workgroups: deep staging from this many stages per tile up (profiles/r05_wino_deep64.jsonl)

struct EpiW {
  const float* bias;
  const float* residual;
  const float* mask;
  int relu;
};

// Cin, Cout: channels; cinpad / copad: the prepared weights' padding; D: planes (1 for a 2D layer); DEPTH: 3x3x3 kernel
template <int PR, int PC, int COB, int KC, bool DEPTH, int ABL, int DEEP = 0>
__global__ __launch_bounds__(COB * 128, COB == 4 ? 1 : 2) void conv_wino(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                                          int Cin, int Cout, int cinpad, int copad, int D, int H, int W, int tiles_w,
                                                                          long long wbytes, EpiW epi) {
  using G = WGeo<PR, PC, COB, KC>;
  constexpr int NT = G::kNT;
  constexpr int dbg = ABL;                // phase ablation for timing (compile-time, so the schedule of the rest is the shipped one): the
                                          // -DADV_TEST_HOOKS build's probe instantiates non-zero values, the shipped kernels are ABL = 0
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int cob = wave % COB, hf = wave / COB;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 2 * PC, h0 = ht * 2 * PR, co0 = blockIdx.y * G::kCO;
  const long long b = DEPTH ? blockIdx.z / D : blockIdx.z;
  const int od = DEPTH ? static_cast<int>(blockIdx.z % D) : 0;
  const long long HW = static_cast<long long>(H) * W;
  const long long DHW = HW * D;

  v4f acc[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

  // <round 5> Addressing by BUFFER loads: the per-stage side work of a wave is issued between its matrix instructions, and the counters
  // (profiles/r04_conv_pmc.json) showed the vector ALU, not the matrix pipe, setting the pace - 80 vector instructions per stage of 32 matrix
  // instructions, 45 of them 64-bit address arithmetic (quarter-rate 32-bit multiplies among them) and clamps.  Now every slot's byte offset
  // inside the IMAGE (input) or inside the prepared weights is computed ONCE per tile as a 32-bit number; a stage adds one wave-uniform
  // offset to it (one v_add per input slot, nothing per weight slot: the scalar offset operand of the instruction), and the hardware's range
  // check against the image's byte count returns zeros for what lies before the image, behind it, or in a channel >= Cin (padding of the
  // contraction) - no clamp, no per-stage mask.  Same loads, same bits.
  //
  // fetch plan of the input tile, once per tile: slot = tid + NT i -> (channel c, tile row r, float4 group j).  The loads themselves are
  // UNCONDITIONAL (no divergent branch around a load: the compiler would have to wait for it at the join, i.e. before the matrix
  // instructions): a slot that lies outside the image row loads whatever the address holds (or zeros, out of range) and is zeroed, element
  // by element, when it is committed to LDS after the stage's matrix instructions.
  const float* const xb = x + b * Cin * DHW;                       // this image (wave-uniform)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, static_cast<int>(static_cast<unsigned>(Cin * DHW * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, static_cast<int>(wbytes), 0x00020000);
  int xvo[G::kXSl];                          // byte offset of the group's first element inside the image, for channel c of a stage at channel 0, plane 0
  unsigned xvm[G::kXSl];                     // bit e: element e of the group is a pixel of the image
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) {
    const int sidx = tid + NT * i;
    const int j = sidx % (G::kLW / 4), r = (sidx / (G::kLW / 4)) % G::kRows, c = sidx / ((G::kLW / 4) * G::kRows);
    const int gh = h0 - 1 + r, gw = w0 - 5 + 4 * j;
    // (32-bit wrap-around is part of the scheme: a negative offset is a huge unsigned one - out of range, zeros - unless the stage's
    //  offset brings it back inside the image, where it addresses the end of the previous channel: masked)
    xvo[i] = static_cast<int>(static_cast<unsigned>(c < KC ? c : KC - 1) * static_cast<unsigned>(DHW) * 4u) + (gh * W + gw) * 4;
    unsigned vm = 0;
    if (sidx < G::kXN && gh >= 0 && gh < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) vm |= (gw + e >= 0 && gw + e < W) ? (1u << e) : 0u;
    xvm[i] = vm;
  }
  struct XSet {                              // staging registers of an input tile
    v4f v[G::kXSl];
  };
  XSet xs;
  v4f rw[G::kWSl];
  // Everything below is written per slot / per row so that a stage's side work (global loads of the stages ahead, the input transform
  // of the next stage, the LDS commits) can be placed BETWEEN the matrix instructions of the current stage, one piece per step.
  // q0 = the stage's first "channel": 2D: the channel itself; 3D: q = kd * cinpad + c, input plane od + kd - 1
  auto stage_off = [&](int q0) -> unsigned {      // wave-uniform: bytes from (channel 0, plane 0) to (the stage's first channel, its plane)
    const int kd = DEPTH ? q0 / cinpad : 0;
    const int ch = DEPTH ? q0 - kd * cinpad : q0;
    return static_cast<unsigned>((static_cast<long long>(ch) * D + (DEPTH ? od + kd - 1 : 0)) * HW * 4);
  };
  auto fetch_x1 = [&](int i, int q0, XSet& set) {
    set.v[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, xvo[i] + static_cast<int>(stage_off(q0)), 0, 0));
  };
  const int wrows = DEPTH ? 3 * cinpad : cinpad;       // rows of U per transform position
  int wvo[G::kWSl];                          // byte offset of the slot's float4 in the first stage's rows of U
#pragma unroll
  for (int i = 0; i < G::kWSl; ++i) {
    const int sidx = tid + NT * i;
    const int q = sidx % (G::kCO / 4), row = sidx / (G::kCO / 4);      // row = k * KC + c
    const int k = row / KC, c = row % KC;
    wvo[i] = ((k * wrows + c) * copad + co0 + 4 * q) * 4;
  }
  auto fetch_w1 = [&](int i, int q0, v4f (&dst)[G::kWSl]) {
    dst[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rwgt, wvo[i], q0 * copad * 4, 0));
  };
  float* const sxb = lds;                               // [2][KC][rows][LW]   input tiles
  float* const swb = lds + 2 * G::kSX;                   // [2][16][KC][CO]     U = G g G^T of the stage
  float* const svb = swb + 2 * G::kSW;                   // [2][16][KC][64]     V = B^T d B of the stage, 64 patches
  auto commit_x1 = [&](int i, int buf, const XSet& set) {
    float* sx = sxb + buf * G::kSX;
    const int sidx = tid + NT * i;
    v4f v = set.v[i];
    const unsigned m = xvm[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (m >> e) & 1u ? v[e] : 0.0f;
    if (sidx < G::kXN) *reinterpret_cast<v4f*>(sx + 4 * sidx) = v;
  };
  // U rows are CO floats, unpadded; a read's four 16-lane groups (q = 0..3 of a step) must hit different banks: 64-float rows are
  // rotated by 16 (q & 3); 32-float rows alternate between the two halves of a 64-bank line by themselves, so q = 1, 2 swap their halves
  // (distinct modulo 32 within each half-wave and modulo 64 across the wave)
  auto uswz = [](int q) { return (G::kCO == 64 ? (q & 3) : ((q ^ (q >> 1)) & 1)) << 4; };
  auto commit_w1 = [&](int i, int buf, const v4f (&src)[G::kWSl]) {
    float* sw = swb + buf * G::kSW;
    const int sidx = tid + NT * i;
    const int q = sidx % (G::kCO / 4), row = sidx / (G::kCO / 4);
    *reinterpret_cast<v4f*>(sw + row * G::kCO + ((4 * q) ^ uswz(row % KC))) = src[i];
  };
  // the input transform: thread -> (q = tid >> 6, patch = tid & 63)
  const int tc = tid >> 6, tp = tid & 63;
  const int tpc = tp < PR * PC ? tp : PR * PC - 1;      // idle patches (5 x 12 tile) repeat the last one
  const int toff = (tc * G::kRows + 2 * (tpc / PC)) * G::kLW + 2 * (tpc % PC) + 4;
  const int voff = tc * 64 + (tp ^ ((tc & 3) << 4));
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
  auto tr_write = [&](int k, int buf) { svb[buf * G::kSV + voff + k * KC * 64] = tv[k]; };

  // operands of the matrix instructions: A = U_k[co = lane & 15][q = lane >> 4], B = V_k[q = lane >> 4][patch = lane & 15]
  const int aoff = k4 * G::kCO + ((cob * 16 + i16) ^ uswz(k4));
  const int boff0 = k4 * 64 + (((2 * hf) * 16 + i16) ^ (k4 << 4)), boff1 = k4 * 64 + (((2 * hf + 1) * 16 + i16) ^ (k4 << 4));

  // the stages: 2D: q = 0, KC, ... < cinpad; 3D: the depth taps whose input plane exists (a contiguous range of q)
  const int q_lo = DEPTH ? (od == 0 ? cinpad : 0) : 0;
  const int q_hi = DEPTH ? (od == D - 1 ? 2 * cinpad : 3 * cinpad) : cinpad;
  const int nstage = (q_hi - q_lo) / KC;
  // <round 4> DEEP staging (32-channel workgroups on layers with many stages): a stage of 4 q is 16 steps - ~0.85 us, less than a round
  // trip to HBM - so a load issued at a stage's start is not always back at its end, where it is committed.  A second set of staging
  // registers lets every load travel for a whole stage more: stage st requests the inputs of stage st + 3 and the weights of stage st + 2
  // into one set and commits the other one (requested during stage st - 1).  The stage loop is then UNIFORM - no peeled tail: requests
  // past the last stage are clamped to it and their commits / transforms fill buffers nobody reads - unrolled by two for the alternation
  // of the register sets.  Worth 3-8 % from 24 (3D) / 32 (2D) stages up, a loss of as much on short contractions (three stages of wasted
  // side work per tile): the launch picks (launch_wino).  Same bits.
  // <round 5> DEEP = 2: the INPUT tiles only (the 64-channel shape has no registers for a second set of weight registers - with both the
  // compiler spills 7-16 registers; its weights keep the one-stage trip: they are L2-resident, the input tiles come from HBM)
  constexpr bool kDeep = DEEP != 0, kDeepW = DEEP == 1;
  XSet xsB;
  v4f rwB[G::kWSl];
  auto qclamp = [&](int q) { return q < q_hi - KC ? q : q_hi - KC; };
  {
    // prologue: the first two input tiles and the first weights are requested TOGETHER (a second, short-lived register set for the
    // second tile) - one memory round trip before the first products instead of two in a row; with 24 short stages per tile (the
    // 32-channel 3D layers) or 8 (64-channel 2D layers) the prologue is a fifth to a third of a tile's time
    XSet xs1;
#pragma unroll
    for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, q_lo, xs);
#pragma unroll
    for (int i = 0; i < G::kWSl; ++i) fetch_w1(i, q_lo, rw);
#pragma unroll
    for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, nstage > 1 ? q_lo + KC : q_lo, xs1);
    if constexpr (kDeep) {       // the second register sets: the input tile of stage 2 and the weights of stage 1 (committed during stage 0)
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) fetch_x1(i, qclamp(q_lo + 2 * KC), xsB);
      if constexpr (kDeepW) {
#pragma unroll
        for (int i = 0; i < G::kWSl; ++i) fetch_w1(i, qclamp(q_lo + KC), rwB);
      }
    }
    if (blockIdx.x == 0 && stage_off(q_lo) == 0) {      // wave-uniform: the tile at the image's origin, its first stage at channel 0, plane 0
      // the float4 that holds columns -1 .. 2 of row 0 starts four bytes BEFORE the image: out of range as a whole (zeros) - its three
      // pixels are fetched one by one.  (Everywhere else such a group starts in the previous row, channel or plane: inside the image.)
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i)
        if (tid + NT * i == G::kLW / 4 + 1)              // channel 0, tile row 1 (image row 0), group 1 (columns -1 .. 2)
          xs.v[i] = v4f{0.0f, xb[0], W > 1 ? xb[1] : 0.0f, W > 2 ? xb[2] : 0.0f};
    }
#pragma unroll
    for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 0, xs);
#pragma unroll
    for (int i = 0; i < G::kWSl; ++i) commit_w1(i, 0, rw);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) tr_read(i, 0);
    tr_compute();
#pragma unroll
    for (int k = 0; k < 16; ++k) tr_write(k, 0);
#pragma unroll
    for (int i = 0; i < G::kXSl; ++i) commit_x1(i, 1, xs1);
    __syncthreads();
  }

  // Stage st: kSteps steps (q group cs = t >> 4, transform position k = t & 15) of two matrix instructions each on V / U of stage st.
  // The operands of step t + 4 are read from LDS before step t issues.  Between the steps, one piece each: the global loads of the
  // inputs of stage st + 2 and the weights of stage st + 1, the input transform of stage st + 1 (LDS reads, the additions, LDS writes)
  // and the commits of the loaded data to LDS - all of them touch buffers the current stage's products do not read.  The scheduler may
  // not move anything across a step (sched_barrier): it would hoist all reads and spill.  ONE barrier per stage.
  auto products = [&](int st, auto fx_tag, auto fw_tag, XSet& fxs, XSet& cxs, v4f (&frw)[G::kWSl], v4f (&crw)[G::kWSl]) {
    constexpr bool FX = decltype(fx_tag)::value, FW = decltype(fw_tag)::value;
    constexpr int kAhead = 4, NS = G::kSteps;
    float ra[NS], rb0[NS], rb1[NS];
    const float* ap = swb + (st & 1) * G::kSW + aoff;
    const float* bp = svb + (st & 1) * G::kSV;
    auto load = [&](int t) {
      const int row = (t & 15) * KC + (t >> 4) * 4;
      ra[t] = ap[row * G::kCO];
      rb0[t] = bp[row * 64 + boff0];
      rb1[t] = bp[row * 64 + boff1];
    };
    const int nb = (st + 1) & 1;       // the buffers of stage st + 1 (V, U) - and of stage st + 2's inputs: st & 1
    const int q1 = q_lo + (st + 1) * KC;
    // what this stage requests: the inputs of stage st + 2 and the weights of stage st + 1 - or, deep, one stage further each
    const int qfx = kDeep ? qclamp(q1 + 2 * KC) : q1 + KC, qfw = kDeepW ? qclamp(q1 + KC) : (kDeep ? qclamp(q1) : q1);
#pragma unroll
    for (int t = 0; t < kAhead; ++t) load(t);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      if (t + kAhead < NS) load(t + kAhead);
      // (input-only deep staging: the weights are requested FIRST - they are committed in this very stage, and the memory counter retires
      //  loads in order: a weight commit would otherwise wait for the stage's input loads too, which are meant to stay in flight)
      constexpr int kFx0 = DEEP == 2 ? G::kWSl : 0, kFw0 = DEEP == 2 ? 0 : G::kLoadW;
      if (FX && t >= kFx0 && t < kFx0 + G::kXSl && !(dbg & 4)) fetch_x1(t - kFx0, qfx, fxs);
      if (FW && t >= kFw0 && t < kFw0 + G::kWSl && !(dbg & 4)) fetch_w1(t - kFw0, qfw, frw);
      if (FW && t >= G::kRead && t < G::kRead + 4 && !(dbg & 1)) tr_read(t - G::kRead, nb);
      if (FW && t == G::kComp && !(dbg & 1)) tr_compute();
      if (FW && t >= G::kWrite && t < G::kWrite + 16 / G::kWritesPerStep && !(dbg & 1)) {
#pragma unroll
        for (int u = 0; u < G::kWritesPerStep; ++u) tr_write(G::kWritesPerStep * (t - G::kWrite) + u, nb);
      }
      if (FX && t >= G::kCommitX && t < G::kCommitX + G::kXSl && !(dbg & 8)) commit_x1(t - G::kCommitX, st & 1, cxs);
      if (FW && t >= G::kCommitW && t < G::kCommitW + G::kWSl && !(dbg & 8)) commit_w1(t - G::kCommitW, nb, crw);
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
  if constexpr (kDeep) {
    for (; st + 1 < nstage; st += 2) {
      if constexpr (kDeepW) {
        products(st, std::true_type{}, std::true_type{}, xs, xsB, rw, rwB);
        if (!(dbg & 16)) __syncthreads();
        products(st + 1, std::true_type{}, std::true_type{}, xsB, xs, rwB, rw);
      } else {
        products(st, std::true_type{}, std::true_type{}, xs, xsB, rw, rw);
        if (!(dbg & 16)) __syncthreads();
        products(st + 1, std::true_type{}, std::true_type{}, xsB, xs, rw, rw);
      }
      if (!(dbg & 16)) __syncthreads();
    }
    if (st < nstage) products(st, std::false_type{}, std::false_type{}, xs, xs, rw, rw);      // an odd count's last stage: products only
  } else {
    for (; st + 2 < nstage; ++st) {
      products(st, std::true_type{}, std::true_type{}, xs, xs, rw, rw);
      if (!(dbg & 16)) __syncthreads();
    }
    if (st + 1 < nstage) {
      products(st, std::false_type{}, std::true_type{}, xs, xs, rw, rw);
      __syncthreads();
      ++st;
    }
    products(st, std::false_type{}, std::false_type{}, xs, xs, rw, rw);
  }

  const long long MP = static_cast<long long>(Cout) * DHW;
  const long long plane0 = static_cast<long long>(od) * HW;
  float* yb = y + b * MP + plane0;
  const float* resb = epi.residual ? epi.residual + b * MP + plane0 : nullptr;
  const float* maskb = epi.mask ? epi.mask + b * MP + plane0 : nullptr;
  const bool vec2 = (W & 1) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(epi.residual) | reinterpret_cast<uintptr_t>(epi.mask)) & 7) == 0;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int t = (2 * hf + blk) * 16 + i16;
    const int gh = h0 + 2 * (t / PC), gw = w0 + 2 * (t % PC);
    if (t >= PR * PC || gh >= H || gw >= W) continue;
    const bool two = gw + 1 < W, row2 = gh + 1 < H;
    const long long at0 = static_cast<long long>(co0 + cob * 16 + 4 * k4) * DHW + static_cast<long long>(gh) * W + gw;
    // the skip connection and the mask of ALL sixteen outputs of this block first, the stores after them: a load placed behind a store
    // to y may not be moved ahead of it (the pointers could alias), so element-by-element code waits a memory round trip per element.
    // Unconditional loads - an element outside the tensor reads the tensor's first float(s) instead and is never stored: a load inside
    // a divergent branch is awaited at the join.  vec2: W even and 8-byte aligned tensors - a patch row is one 8-byte access.
    float rv[4][2][2], mv[4][2][2];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) rv[reg][r][c] = 0.0f, mv[reg][r][c] = 1.0f;
    auto gather = [&](const float* base, const float* safe, float (&dst)[4][2][2]) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const bool rok = co0 + cob * 16 + 4 * k4 + reg < Cout && (r == 0 || row2);
          const float* p = base + (at0 + reg * DHW + r * W);
          if (vec2) {
            const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(rok ? p : safe));
            dst[reg][r][0] = v[0], dst[reg][r][1] = v[1];
          } else {
            dst[reg][r][0] = __builtin_nontemporal_load(rok ? p : safe);
            dst[reg][r][1] = __builtin_nontemporal_load(rok && two ? p + 1 : safe);
          }
        }
    };
    if (resb) gather(resb, epi.residual, rv);
    if (maskb) gather(maskb, epi.mask, mv);
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
        if (r == 1 && !row2) continue;
        const long long at = at0 + reg * DHW + r * W;
        float v0 = o[r][0], v1 = o[r][1];
        if (epi.bias) v0 = v0 + bv, v1 = v1 + bv;
        if (resb) v0 = v0 + rv[reg][r][0], v1 = v1 + rv[reg][r][1];
        if (epi.relu) v0 = v0 > 0.0f ? v0 : 0.0f, v1 = v1 > 0.0f ? v1 : 0.0f;
        if (maskb) v0 = mv[reg][r][0] > 0.0f ? v0 : 0.0f, v1 = mv[reg][r][1] > 0.0f ? v1 : 0.0f;
        if (vec2) {
          *reinterpret_cast<v2f*>(yb + at) = v2f{v0, v1};
        } else {
          yb[at] = v0;
          if (two) yb[at + 1] = v1;
        }
      }
    }
  }
}

template <int PR, int PC, int COB, int KC, bool DEPTH>
int launch_wino(const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int d, int h, int w, const EpiW& epi,
                hipStream_t st) {
  using G = WGeo<PR, PC, COB, KC>;
  const int tiles_w = (w + 2 * PC - 1) / (2 * PC), tiles_h = (h + 2 * PR - 1) / (2 * PR);
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + G::kCO - 1) / G::kCO;
  const long long gz = static_cast<long long>(b) * d;
  if (tiles > 0x7fffffffLL || cgroups > 65535 || gz > 65535) return ADV_EINVAL;
  // the kernel addresses an image and the prepared weights with 32-bit byte offsets (buffer loads): both must stay below 4 GiB
  const long long wbytes = 16LL * (DEPTH ? 3 : 1) * cinpad * copad * 4;
  if ((static_cast<long long>(cinpad) + 1) * d * h * w * 4 >= 0xfff00000LL || wbytes >= 0xfff00000LL) return ADV_EINVAL;
  const dim3 grid(static_cast<unsigned>(tiles), cgroups, static_cast<unsigned>(gz));
#ifdef ADV_TEST_HOOKS
  if (const char* dbg_s = adv_hook_value("ADV_WINO_DBG")) {      // phase ablation for timing (results are wrong); the 8 x 32 x 64 2D shape and the 8 x 32 x 32 3D shape
    if constexpr (PR == 4 && PC == 16 && ((COB == 4 && !DEPTH) || (COB == 2 && DEPTH))) {
      const int abl = std::atoi(dbg_s);
#define ADV_WINO_ABL(A_)                                                                                                                  \
  if (abl == A_) {                                                                                                                        \
    if (!adv_internal_lds_limit<conv_wino<PR, PC, COB, KC, DEPTH, A_>>(G::kLds)) return ADV_ELAUNCH;                                      \
    hipLaunchKernelGGL((conv_wino<PR, PC, COB, KC, DEPTH, A_>), grid, dim3(G::kNT), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, \
                       tiles_w, wbytes, epi);                                                                                              \
    return adv_internal_finish_launch();                                                                                                  \
  }
      ADV_WINO_ABL(1) ADV_WINO_ABL(2) ADV_WINO_ABL(4) ADV_WINO_ABL(8) ADV_WINO_ABL(16) ADV_WINO_ABL(3) ADV_WINO_ABL(13) ADV_WINO_ABL(29) ADV_WINO_ABL(31)
#undef ADV_WINO_ABL
    }
  }
#endif
  {
    // deep staging (a second set of staging registers: every load travels a stage longer): 32-channel workgroups from 24 (3D) / 32 (2D)
    // stages up (round 4); <round 5> the 64-channel shape too - the buffer-load addressing freed the registers - from ADV_DEEP64 stages up
    const int stages = (DEPTH ? 3 : 1) * cinpad / KC;       // (interior planes)
    bool deep = COB == 2 ? stages >= (DEPTH ? 24 : 32) : stages >= kDeep64Stages;
    if (const char* e = adv_hook_value("ADV_WINO_DEEP")) deep = e[0] == '1';      // test hook / A-B; same bits
    if (deep) {
      constexpr int kMode = COB == 2 ? 1 : 2;      // 32-channel workgroups: inputs and weights; 64-channel: the inputs only
      if (!adv_internal_lds_limit<conv_wino<PR, PC, COB, KC, DEPTH, 0, kMode>>(G::kLds)) return ADV_ELAUNCH;
      hipLaunchKernelGGL((conv_wino<PR, PC, COB, KC, DEPTH, 0, kMode>), grid, dim3(G::kNT), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w,
                         tiles_w, wbytes, epi);
      return adv_internal_finish_launch();
    }
  }
  if (!adv_internal_lds_limit<conv_wino<PR, PC, COB, KC, DEPTH, 0>>(G::kLds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv_wino<PR, PC, COB, KC, DEPTH, 0>), grid, dim3(G::kNT), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, tiles_w,
                     wbytes, epi);
  return adv_internal_finish_launch();
}

template <bool DEPTH>
int launch_wino_tile(int t, const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int d, int h, int w,
                     const EpiW& epi, hipStream_t st) {
  switch (t) {
    case 0: return launch_wino<4, 16, 4, 8, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 1: return launch_wino<8, 8, 4, 8, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 2: return launch_wino<4, 16, 2, 4, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 3: return launch_wino<8, 8, 2, 4, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 4: return launch_wino<5, 12, 4, 8, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 5: return launch_wino<5, 12, 2, 4, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 6: return launch_wino<3, 20, 4, 8, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    default: return launch_wino<3, 20, 2, 4, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
  }
}

// tile: 0 = 8 x 32 outputs x 64 channels, 1 = 16 x 16 x 64, 2 = 8 x 32 x 32 channels, 3 = 16 x 16 x 32, 4 = 10 x 24 x 64, 5 = 10 x 24 x 32,
// <round 4> 6 = 6 x 40 x 64, 7 = 6 x 40 x 32 (3 x 20 patches): the 5-row maps of the 3D geometric volume's quarter resolution ([48,5,76]: two
// tiles of 6 x 40 per plane instead of three of 8 x 32, 1.6x fewer padded outputs) and the 24 x 78 planes of the cost volume's
int pick_wino_tile(int cout, int h, int w, long long bz) {
  // 8 x 32 outputs per workgroup (128-byte store runs) unless another shape needs clearly fewer workgroups for the map: 16 x 16 for the
  // 14 x 14 maps of the box heads (one tile instead of two per image), 10 x 24 for the bird's-eye volumes of 10 / 20 rows;
  // 32-channel workgroups where 64 would compute a zero block
  auto tiles = [&](int th, int tw) { return static_cast<long long>((h + th - 1) / th) * ((w + tw - 1) / tw); };
  const long long t0 = tiles(8, 32) * 10, t1 = tiles(16, 16) * 12, t2 = tiles(10, 24) * 11, t3 = tiles(6, 40) * 11;   // x 1.2, x 1.1: the handicaps of the other shapes
  int shape = (t1 < t0 && t1 <= t2) ? 1 : (t2 < t0 ? 4 : 0);
  if (t3 < t0 && t3 < t1 && t3 < t2) shape = 6;
  int narrow = ((cout + 31) / 32) % 2 == 1 ? ((shape == 4 || shape == 6) ? 1 : 2) : 0;      // an odd number of 32-channel blocks
  // <round 4> ... and where 64-channel workgroups (one per compute unit) would leave more than half of the chip idle: the 19 x 63 and
  // 48 x 76 maps at one pair per step (48-96 workgroups).  Two 32-channel workgroups per unit transform every input tile twice, but
  // they run everywhere: 1.22-1.24x on those layers, slower from ~160 workgroups up (profiles/r04_wino_tiles.jsonl).  Same bits.
  const long long t = shape == 1 ? tiles(16, 16) : (shape == 4 ? tiles(10, 24) : (shape == 6 ? tiles(6, 40) : tiles(8, 32)));
  // <round 5> the same question as a makespan estimate over the chip's 256 compute units (profiles/r05_wino_tiles.jsonl): 64-channel workgroups
  // run one per unit, ceil(n64 / 256) rounds; 32-channel workgroups run two per unit at ~7 % more work per output (every input tile is
  // transformed twice) but their LAST round is finer-grained - up to 256 left-over workgroups sit alone on a unit and finish in half a
  // round.  The quantisation of a 64-channel launch of 1.25 or 2.5 rounds (the 75 x 249 maps: 320 / 640 workgroups) costs 17-24 %; launches of
  // many rounds gain nothing from the finer tail and keep the 64-channel shape.  (n64 <= 128 is the round-4 rule: half of the chip idle.)
  if (narrow == 0) {
    const long long n64 = t * ((cout + 63) / 64) * bz, n32 = t * ((cout + 31) / 32) * bz;
    const double t64 = static_cast<double>((n64 + 255) / 256);
    const long long rem = n32 % 512;
    const double t32 = 1.07 * (static_cast<double>(n32 / 512) + (rem == 0 ? 0.0 : (rem <= 256 ? 0.5 : 1.0)));
    if (t32 < t64) narrow = (shape == 4 || shape == 6) ? 1 : 2;
  }
  return shape + narrow;
}

int check_wino_args(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, const float* y) {
  if (residual == y || mask == y || x == y) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(y) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 3)) || (mask && (reinterpret_cast<uintptr_t>(mask) & 3)) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return ADV_EALIGN;
  return ADV_OK;
}

// U = G g G^T for every (output, input) channel pair (and depth tap), laid out [k = 4 i + j][kd][c'][m'] (zero rows / columns of
// padding).  forward: m = co, c = ci, g = w[co][ci][kd];  transpose (backward w.r.t. the input): m = ci, c = co, g = w[co][ci] with all
// its taps reversed (2 - kd, and the 3x3 slice rotated by 180 degrees).  taps = 1: a 2D layer's [Cout][Cin][3][3] weights.
__global__ void conv_wino_prep_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps, int transpose, int kpad,
                                      int mpad) {
  const long long n = static_cast<long long>(taps) * kpad * mpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
    const int m = static_cast<int>(i % mpad), c = static_cast<int>((i / mpad) % kpad), kd = static_cast<int>(i / (static_cast<long long>(mpad) * kpad));
    const int co = transpose ? c : m, ci = transpose ? m : c;
    float u[16];
    if (co < cout && ci < cin) {
      const float* gp = w + ((static_cast<long long>(co) * cin + ci) * taps + (transpose ? taps - 1 - kd : kd)) * 9;
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

int prep_wino(const float* w, float* w_prep, int cout, int cin, int taps, int transpose, hipStream_t st) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up_w(k, 8), mpad = round_up_w(m, 64);
  const long long n = static_cast<long long>(taps) * kpad * mpad;
  const unsigned blocks = static_cast<unsigned>(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
  hipLaunchKernelGGL(conv_wino_prep_kernel, dim3(blocks), dim3(256), 0, st, w, w_prep, cout, cin, taps, transpose ? 1 : 0, kpad, mpad);
  return adv_internal_finish_launch();
}

}  // namespace

extern "C" {

int64_t adv_conv2d_wino_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 16LL * round_up_w(k, 8) * round_up_w(m, 64);
}

int adv_conv2d_wino_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino(w, w_prep, cout, cin, 1, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv2d_wino_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                        int cout, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || tile < -1 || tile > 7) return ADV_EINVAL;
  if (static_cast<long long>(b) * cin * h * w < 4) return ADV_EINVAL;      // the kernel loads whole float4s (clamped into the tensor)
  if (const int rc = check_wino_args(x, w_prep, bias, residual, mask, y)) return rc;
  const EpiW epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino_tile<false>(tile >= 0 ? tile : pick_wino_tile(cout, h, w, b), x, w_prep, y, b, cin, cout, round_up_w(cin, 8), round_up_w(cout, 64), 1,
                                 h, w, epi, static_cast<hipStream_t>(stream));
}

int64_t adv_conv3d_wino_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 48LL * round_up_w(k, 8) * round_up_w(m, 64);
}

int adv_conv3d_wino_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino(w, w_prep, cout, cin, 3, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv3d_wino_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                        int cout, int d, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1 || tile < -1 || tile > 7) return ADV_EINVAL;
  if (static_cast<long long>(b) * cin * d * h * w < 4) return ADV_EINVAL;
  if (const int rc = check_wino_args(x, w_prep, bias, residual, mask, y)) return rc;
  const EpiW epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino_tile<true>(tile >= 0 ? tile : pick_wino_tile(cout, h, w, static_cast<long long>(b) * d), x, w_prep, y, b, cin, cout, round_up_w(cin, 8), round_up_w(cout, 64), d,
                                h, w, epi, static_cast<hipStream_t>(stream));
}

}  // extern "C"
