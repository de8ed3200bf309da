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
// EIGHT waves (two per SIMD, 256 registers each) over a tile of 32 patches x 64 output channels (CB = 2) or 64 patches x 32 channels
// (CB = 1), and every wave holds its positions for the WHOLE tile in 32 x 32 accumulators.
//
// <round 6> What the machine does with such a kernel, measured with in-kernel time stamps and two probes (profiles/r06_wino4_stamps*.jsonl,
// r06_mfma_valu_probe.json, r06_wino4_stagger_negative.*), and what this version does about it:
//   * a float32 matrix instruction and the OTHER wave's float32 vector instructions do not execute at the same time on a SIMD (matrix waves
//     alone 0.254 ms, vector waves alone 0.190 ms, together 0.420 ms): the input transform's ~93 vector instructions per thread and stage come
//     out of the matrix pipe's time whichever wave issues them.  Only LDS, memory and scalar instructions hide behind matrix instructions.
//     The matrix-pipe counter of this kernel is therefore bounded by matrix / (matrix + vector) cycles = 0.86 (CB = 2) / 0.75 (CB = 1).
//   * the matrix pipe serves the OLDER wave of a SIMD whenever both are ready, and a wave issues in order: round 5's split (waves 0-3: four
//     positions and the whole transform, waves 4-7: five positions) left waves 4-7 idle for ~800 of 3 500 cycles per stage at the barrier.
//     Now every wave is alike: the nine positions of a SIMD's two waves are four whole ones each and the ninth split by channel (patch) block -
//     NINE accumulators and 18 matrix instructions per four q for every wave - and every thread transforms half a (q, patch) pair per stage
//     (a stage is 8 q at CB = 2: half the barriers of round 5 per matrix instruction).
//   * U (the transformed weights) never touches LDS: every wave multiplies its own positions only, so the prepared layout hands each lane its
//     A operands as one 16-byte buffer load per (position, four q), requested a stage (two at CB = 1) ahead into the registers just consumed:
//     37 KB per stage less through LDS, no weight DMA to issue or to wait for at the barrier.
//   * the input tile arrives by LDS-DMA (1 KiB per wave instruction, three buffers, two stages ahead; the range check writes the zero
//     padding): no staging registers, no per-element masks, no commit stores (round 5's were 75 % of the kernel's LDS bank conflicts).
//   Tried and dropped: fixed roles in time (waves 0-3 all their matrix instructions first while waves 4-7 do their side work, then the
//   other way round) - 7 % SLOWER: the side-work wave makes no progress while its partner streams matrix instructions (first bullet).
//
//   per stage of KC = 4 CB q:   input tile [KC][4 PR + 2][groups of 4 columns]   global -> LDS by LDS-DMA
//                               U = G g G^T: global -> registers, each wave its own positions
//                               V = B^T d B [36][KC][patches]   every thread: half a (q, patch) pair, three of the six rows of V
//   one "s_waitcnt; s_barrier" per stage (not __syncthreads(): its fence would drain the loads in flight), the side work woven between the
//   matrix instructions in half-steps (one matrix instruction per scheduling step).  The 36 values of one (channel, patch) sit in eight
//   waves after the contraction: they are exchanged through LDS in rounds of 16 channels and every thread transforms one or two (channel,
//   patch) items per round.  PAIR: two images of at most 15 columns side by side in one 32-column tile (the RoI heads' 14 x 14 maps).
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

#ifdef ADV_WINO4_STAMPS
// diagnostic builds only (tools/build_variant.sh ... -DADV_WINO4_STAMPS; tools/wino4_stamps.py): s_memtime stamps of the stage loop of the first
// workgroups - [workgroup < 8][wave][stage < 64][5]: stage start, start of half-step (stage % half-steps), last matrix instruction issued,
// LDS drained, barrier passed.  The stamps go to a buffer of their own and feed nothing.
__device__ unsigned long long adv_wino4_stamps[8][8][64][5];
extern "C" __attribute__((visibility("default"))) int adv_debug_wino4_stamps(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(adv_wino4_stamps), bytes < sizeof(adv_wino4_stamps) ? bytes : sizeof(adv_wino4_stamps)));
}
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));      // four floats at any float address (global memory takes unaligned 16-byte accesses)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kKQ = 4;        // "channels" q per SUB-stage (two matrix-instruction steps of two q each); a stage is CB sub-stages

// PR x PC patches of 4 x 4 outputs per workgroup, CB blocks of 32 output channels: 32 patches x 64 channels (CB = 2) or 64 patches x 32
// channels (CB = 1: layers of 32 output channels or an odd number of 32-channel blocks).  A stage covers KC = 4 CB "channels" q, so that the
// input transform of a stage is 256 (q, patch) pairs for either shape: every one of the 512 threads transforms half a pair per stage.
template <int PR, int PC, int CB>
struct W4Geo {
  static constexpr int kPB = 2 / CB, kNP = 32 * kPB, kCO = 32 * CB, kKC = kKQ * CB;
  static_assert((CB == 1 || CB == 2) && PR * PC == kNP, "32 patches x 64 channels or 64 patches x 32 channels per workgroup");
  static constexpr int kRows = 4 * PR + 2;                        // input rows h0 - 1 .. h0 + 4 PR
  // Input tiles arrive by LDS-DMA (16 bytes per lane, 64 consecutive groups of four columns per wave instruction): the LDS image is
  // [q][row][group], the groups of a row start at column w0 - 4 (whole aligned groups: none straddles the map's left edge; one entirely
  // outside the map is a lane whose offset fails the range check = zeros).  Groups per row: PC + 2 needed, padded so that the patch rows
  // read together by the transform fall into different banks (pitch = 8 mod 16 floats at 8 patches per row, 0 mod 16 at 16).
  static constexpr int kGPR = PC == 16 ? 20 : PC + 2;
  static_assert(PC == 16 || (PC == 8 && kGPR % 4 == 2), "row pitch against bank conflicts");
  static constexpr int kPitch = 4 * kGPR;
  static constexpr int kXN = kKC * kRows * kGPR;                  // groups per stage
  static constexpr int kPieces = (kXN + 63) / 64;                 // wave instructions per stage (1 KiB each)
  static constexpr int kXSl = (kPieces + 7) / 8, kXMin = kPieces / 8;      // pieces per wave: at most / at least
  static constexpr int kSX = kPieces * 256;                       // floats per input-tile buffer
  static constexpr int kNXB = 3;                                  // input-tile buffers: a tile is requested two stages before the transform reads it
  static constexpr int kSV = 36 * kKC * kNP;                      // floats per V buffer (two of them)
  static_assert(kKC * kNP == 256, "one (q, patch) pair per two threads");
  static constexpr size_t kMain = sizeof(float) * (kNXB * kSX + 2 * kSV), kExch = sizeof(float) * 36 * 16 * kNP;      // stage buffers; the epilogue's exchange buffer
  static constexpr size_t kLds = kMain > kExch ? kMain : kExch;
  static_assert(kLds <= 160 * 1024, "the CU's LDS holds the workgroup");
};

struct Epi4 {
  const float* bias;
  const float* residual;
  const float* mask;
  int relu;
  int ksplit, kchunk;      // 2D layers on small maps: the contraction dealt to ksplit workgroups per tile, kchunk channels each (0 / 1: whole)
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
// 1 matrix instructions, 2 input-tile DMA, 3 the edge fix-up, 4 the stage's wait + barrier, 5 operand reads, 6 weight loads.
// The -DADV_TEST_HOOKS build's probe instantiates non-zero values, the shipped kernels are ABL = 0.
// PAIR (16 x 32-output tiles of 2D layers on maps of at most 15 x 16 pixels - the RoI heads' 14 x 14 maps): the tile's left and right halves are
// TWO IMAGES side by side (blockIdx.z = image pair): an image's right zero padding and its neighbour's left one coincide in the LDS tile, so
// a 14 x 14 map uses 77 % of the tile instead of 38 %.  Needs Cin % 8 == 0 (the range check covers the pair, not one image's channels).
template <int PR, int PC, int CB, bool DEPTH, int ABL = 0, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void conv_wino4(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, int Cin, int Cout,
                                                     int cinpad, int copad, int D, int H, int W, int tiles_w, long long wbytes, int nimg, int flags, Epi4 epi) {
  using G = W4Geo<PR, PC, CB>;
  constexpr int PB = G::kPB, NPT = G::kNP, CO = G::kCO, KC = G::kKC, NSUB = CB, NH = 18 * CB;      // NH: matrix instructions (= half-steps) per wave and stage
  constexpr int dbg = ABL;
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef ADV_WINO4_STAMPS
  const unsigned long long stamp_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * 4 * PC, h0 = ht * 4 * PR, co0 = blockIdx.y * CO;
  static_assert(!PAIR || (!DEPTH && PC == 8), "image pairs: the 16 x 32 tile of a 2D layer");
  // K-split (2D, one image per tile): blockIdx.z = image * ksplit + part; a part multiplies kchunk input channels and writes its raw
  // F(4x4,3x3) outputs to plane `part` of a scratch tensor [ksplit][B][Cout][H][W] (y here) - wino4_ksplit_sum adds the planes in order
  const int ksn = (!DEPTH && !PAIR && epi.ksplit > 1) ? epi.ksplit : 1;
  const int ks = ksn > 1 ? static_cast<int>(blockIdx.z % ksn) : 0;
  const long long b = DEPTH ? blockIdx.z / D : (PAIR ? 2LL * blockIdx.z : (ksn > 1 ? blockIdx.z / ksn : blockIdx.z));      // PAIR: the first image of the pair
  const int od = DEPTH ? static_cast<int>(blockIdx.z % D) : 0;
  const long long HW = static_cast<long long>(H) * W;
  const long long DHW = HW * D;

  float* const sxb = lds;                          // [3][KC][rows][groups][4]   input tiles (+ the pad of the last 1 KiB piece)
  float* const svb = lds + G::kNXB * G::kSX;       // [2][36][KC][NPT]           V of the stage

  // ---- input tiles: LDS-DMA, one 32-bit byte offset per piece and lane computed once per tile; the descriptor's base moves with the
  // stage (scalar arithmetic), the hardware's range check writes the zeros of the padding: a lane outside the map (row, whole group, pad)
  // carries an offset no descriptor admits.  Left: groups start on multiples of four columns, none straddles column 0.  Right (W % 4 != 0):
  // the group that straddles column W brings the next row's first columns along - the lane that requested it overwrites them with zeros
  // when its piece has landed (xfm: the elements to clear; the rare lanes of the right-edge tiles only).
  const float* const xb = x + b * Cin * DHW;
  const unsigned xtotal = static_cast<unsigned>((PAIR ? (b + 1 < nimg ? 2LL : 1LL) : 1LL) * Cin * DHW * 4);      // bytes behind the descriptor (whole tensor of the image / the pair)
  const __amdgpu_buffer_rsrc_t rwgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, static_cast<int>(wbytes), 0x00020000);
  int xvo[G::kXSl];
  unsigned xfm[G::kXSl];
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) {
    const int sidx = (wave + 8 * i) * 64 + lane;
    const int j = sidx % G::kGPR, r = (sidx / G::kGPR) % G::kRows, c = sidx / (G::kGPR * G::kRows);
    // PAIR: groups 0-4 are image A's columns -4 .. 15, groups 5-9 image B's columns 0 .. 19 (B's column -1 is A's column 15: zero for both)
    const int gh = h0 - 1 + r, gw = PAIR ? (j < 5 ? 4 * j - 4 : 4 * (j - 5)) : w0 - 4 + 4 * j;
    const bool in = wave + 8 * i < G::kPieces && c < KC && j < PC + 2 && gh >= 0 && gh < H && gw >= 0 && gw < W;
    xvo[i] = in ? static_cast<int>(static_cast<unsigned>(c) * static_cast<unsigned>(DHW) * 4u + static_cast<unsigned>(gh * W + gw) * 4u +
                                   (PAIR && j >= 5 ? static_cast<unsigned>(Cin * DHW) * 4u : 0u))
                : static_cast<int>(0xFFFFFF00u);
    unsigned fm = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e) fm |= (in && gw + e >= W) ? (1u << e) : 0u;
    xfm[i] = fm;
  }
  unsigned anyfm = 0;
#pragma unroll
  for (int i = 0; i < G::kXSl; ++i) anyfm |= xfm[i];
  const bool wave_fixes = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(anyfm != 0) != 0 ? 1 : 0) != 0;      // any straddling group in this wave's pieces?
  auto stage_off = [&](int q0) -> long long {      // wave-uniform: bytes from (channel 0, plane 0) to (the stage's first channel, its plane)
    const int kd = DEPTH ? q0 / cinpad : 0;
    const int ch = DEPTH ? q0 - kd * cinpad : q0;
    return (static_cast<long long>(ch) * D + (DEPTH ? od + kd - 1 : 0)) * HW * 4;
  };
  auto dma_x = [&](int i, long long so, int buf) {      // piece wave + 8 i of the tile of the stage so bytes into the image -> input buffer buf
    const long long left = static_cast<long long>(xtotal) - so;      // (channels past Cin: nothing left - zeros)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xb) + so), 0,
                                                                         static_cast<int>(static_cast<unsigned>(left > 0 ? left : 0)), 0x00020000);
    float* dst = sxb + buf * G::kSX + (wave + 8 * i) * 256;
    const int vo = xvo[i];      // (a copy: the host pass of hipcc cannot compile the array element as this builtin's argument)
    if (i < G::kXMin || wave + 8 * i < G::kPieces) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)dst, 16, vo, 0, 0, 0);
  };
  auto fix_x = [&](int i, int buf) {               // the straddling groups of piece wave + 8 i: the columns past the map's right edge -> 0
    if (xfm[i]) {
      float* p = sxb + buf * G::kSX + ((wave + 8 * i) * 64 + lane) * 4;
#pragma unroll
      for (int e = 1; e < 4; ++e)
        if ((xfm[i] >> e) & 1u) p[e] = 0.0f;
    }
  };

  // ---- positions.  The two waves of a SIMD (w and w + 4) carry nine of the 36 positions between them, k = sp + 4 j (sp = w & 3): wave w the
  // positions j = 0 .. 3 whole and block 0 of position 4 (channel block at CB = 2, patch block at CB = 1), wave w + 4 block 1 of position 4 and
  // j = 5 .. 8 whole: NINE 32 x 32 accumulators and 18 matrix instructions per sub-stage for every wave, and every wave computes its
  // 1/512th of the input transform - no wave waits at the stage's barrier for a partner with more to do (an earlier version gave four /
  // five whole positions and the whole transform to waves 0-3: those finished 800 cycles after the others, profiles/r06_wino4_stamps.jsonl).
  const int sp = wave & 3;
  // weights: every wave multiplies ITS OWN positions and nobody else reads them, so U never touches LDS: the prepared layout
  // [k][q / 4][co / 64][lane][cb, kp] hands lane (half, l32) the four A operands of one (position, sub-stage) - q = 4 s + 2 kp + half,
  // co = 64 g + 32 cb + l32 - as ONE 16-byte buffer load (1 KiB per wave instruction, whole lines); the shared position and 32-channel
  // workgroups take one block's half (8 bytes).  The wave-uniform part of the address travels in the scalar offset.  Register ring: the load
  // of (sub-stage + 2, slot) is issued right behind the last matrix instruction of (sub-stage, slot), into the registers that one read.
  const int wqs = (DEPTH ? 3 * cinpad : cinpad) / kKQ;     // sub-stages of U per transform position
  const int wg64 = copad / 64;
  const int wgrp = CB == 2 ? blockIdx.y : blockIdx.y >> 1;
  const int wvo16 = lane * 16, wcb = CB == 2 ? 0 : 8 * (blockIdx.y & 1);

  // ---- the input transform: two threads per (q, patch) pair, three of the six rows of V each (waves 0-3: rows 0-2, waves 4-7: rows 3-5)
  const int pair = tid & 255, tc = pair / NPT, tp = pair % NPT;
  const int toff = (tc * G::kRows + 4 * (tp / PC)) * G::kPitch + 4 * (tp % PC) + 2;     // LDS column of the patch's first column - 1 (8-byte aligned)
  float td[5][6], tt[3][6], tv[18];
  auto tr_read = [&](int i, const float* tile) {           // six columns at LDS columns 4 p + 3 .. 4 p + 8: 8 + 16 + 8 bytes, all aligned
    const float* p = tile + i * G::kPitch;
    const v2f a = *reinterpret_cast<const v2f*>(p);
    const v4f m = *reinterpret_cast<const v4f*>(p + 2);
    const v2f n = *reinterpret_cast<const v2f*>(p + 6);
    td[i][0] = a[1], td[i][1] = m[0], td[i][2] = m[1], td[i][3] = m[2], td[i][4] = m[3], td[i][5] = n[0];
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

  const int q_lo = DEPTH ? (od == 0 ? cinpad : 0) : ks * epi.kchunk;
  const int q_hi = DEPTH ? (od == D - 1 ? 2 * cinpad : 3 * cinpad) : (ksn > 1 && q_lo + epi.kchunk < cinpad ? q_lo + epi.kchunk : cinpad);
  const int nstage = (q_hi - q_lo) / KC;
  auto qclamp = [&](int q) { return q < q_hi - KC ? q : q_hi - KC; };      // (stage starts beyond the last one: the last one again)

  // epilogue geometry: the exchange rounds hand every thread (channel of the round's 16, patch) items - one at 32 patches, two at 64
  const long long MP = static_cast<long long>(Cout) * DHW;
  const long long plane0 = static_cast<long long>(od) * HW;
  float* const yb = y + (static_cast<long long>(ks) * nimg + b) * MP + plane0;
  const float* const resb = epi.residual ? epi.residual + b * MP + plane0 : nullptr;
  const float* const maskb = epi.mask ? epi.mask + b * MP + plane0 : nullptr;
  const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(epi.residual) | reinterpret_cast<uintptr_t>(epi.mask)) & 15) == 0;
  const int ep = tid % NPT, eg = tid / NPT;               // patch; first channel of the round's 16 (then + 512 / NPT)
  const int gh0 = h0 + 4 * (ep / PC), gw0 = PAIR ? 4 * (ep % 4) : w0 + 4 * (ep % PC);
  const long long eimg = PAIR && (ep % PC) >= 4 ? MP : 0;      // PAIR: the right half of the tile is the pair's second image
  const bool eok = !PAIR || (ep % PC) < 4 || b + 1 < nimg;
  const bool wide = gw0 + 3 < W;      // (per thread) the patch's four columns are inside the map

  // body<HI>: a wave's whole life after the set-up (HI: waves 4-7)
  auto body = [&](auto hi_c) __attribute__((always_inline)) {
    constexpr bool HI = decltype(hi_c)::value;
    constexpr int JB = HI ? 5 : 0;                // whole positions j = JB + slot (slot 0 .. 3); slot 4 = position 4, block HI
    if ((!HI && (flags & 1)) || (HI && (flags & 2))) __builtin_amdgcn_s_setprio(1);      // static priority of one half of the waves (A/B: kWaveFlags)
    f32x16 acc[9];                                // [2 slot + block] for the whole positions, [8] the shared one
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[u][v] = 0.0f;
    auto kpos = [&](int slot) { return sp + 4 * (slot < 4 ? JB + slot : 4); };
    using WF = std::conditional_t<CB == 2, v4f, v2f>;      // whole position: [cb][kp] or [kp]
    struct WSet {
      WF f[NSUB][4];
      v2f s[NSUB];
    };
    WSet wsA, wsB;
    // byte offset of (position, sub-stage s, the workgroup's channel group): wbase[slot] + s * wstep - one scalar add per load in the loop
    const int wstep = wg64 * 1024;
    int wbase[5];
#pragma unroll
    for (int slot = 0; slot < 5; ++slot) wbase[slot] = (kpos(slot) * wqs * wg64 + wgrp) * 1024;
    auto load_w = [&](int sub, int slot, int qoff, WSet& set) {      // qoff: (first q of the sub-stage / 4) * wstep
      const int so = wbase[slot] + qoff;          // wave-uniform
      if (slot < 4) {
        if constexpr (CB == 2) set.f[sub][slot & 3] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rwgt, wvo16, so, 0));
        else set.f[sub][slot & 3] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rwgt, wvo16 + wcb, so, 0));
      } else {
        set.s[sub] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rwgt, wvo16 + (CB == 2 ? (HI ? 8 : 0) : wcb), so, 0));
      }
    };
    auto tr_write = [&](int idx, float* vdst) {   // vdst: the V buffer + the pair's offset
      const int k = (3 * (HI ? 1 : 0) + idx / 6) * 6 + idx % 6;
      vdst[k * 256] = tv[idx];
    };
#ifdef ADV_WINO4_STAMPS
    unsigned long long pst[5] = {0, 0, 0, 0, 0};      // prologue: set-up done, requests issued, tiles landed, first barrier passed
    pst[0] = __builtin_amdgcn_s_memtime();
#endif
    {
      // prologue: the first three input tiles and the weights of the first two sub-stages are requested together
#pragma unroll
      for (int t = 0; t < G::kNXB; ++t)
#pragma unroll
        for (int i = 0; i < G::kXSl; ++i) dma_x(i, stage_off(qclamp(q_lo + t * KC)), t);
#pragma unroll
      for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int slot = 0; slot < 5; ++slot) load_w(s, slot, (q_lo / kKQ + s) * wstep, wsA);
      if constexpr (CB == 1) {
#pragma unroll
        for (int slot = 0; slot < 5; ++slot) load_w(0, slot, qclamp(q_lo + KC) / kKQ * wstep, wsB);
      }
#ifdef ADV_WINO4_STAMPS
      pst[1] = __builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // the tiles have landed (the weight loads behind them travel on)
#ifdef ADV_WINO4_STAMPS
      pst[2] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
      for (int i = 0; i < G::kXSl; ++i) fix_x(i, 0), fix_x(i, 1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ADV_WINO4_STAMPS
      pst[3] = __builtin_amdgcn_s_memtime();
#endif
      {
        const float* tile = sxb + toff + (HI ? G::kPitch : 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) tr_read(i, tile);
#pragma unroll
        for (int j = 0; j < 6; ++j) tr_cols(hi_c, j);
#pragma unroll
        for (int i = 0; i < 3; ++i) tr_row(i);
#pragma unroll
        for (int k = 0; k < 18; ++k) tr_write(k, svb + pair);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#ifdef ADV_WINO4_STAMPS
    unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_loop_start = stamp0;
    const bool stamped = blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0;
#endif
    // Stage st: NH matrix instructions on U (registers) / V (LDS) of stage st, one per HALF-step, the side work in front of them (a wave
    // issues in order: side work behind a matrix instruction runs in its 64-cycle shadow).  Woven in, one piece per half-step: the wave's
    // pieces of the input tile of stage st + 3 (LDS-DMA), its eighteen-step share of the input transform of stage st + 1 (row reads,
    // columns, rows, writes), a weight load behind the last matrix instruction of every slot.  The scheduler may not move anything across a
    // half-step.  At the end: the pieces of the tile of stage st + 2 (requested a stage ago) have landed - counted wait, the younger loads
    // travel on - their right-edge columns are cleared, then ONE barrier behind the wave's LDS writes.
    int xr = 0;                                   // st % 3: the input buffer of tiles st, st + 3
    // the tile the next stage requests (stage st: the tile of stage st + 3), kept incrementally: its first q, that q's channel, its byte
    // offset - a division by the channel count per stage (3x3x3 layers) was ~50 scalar instructions in the issue stream of every wave
    int xq = qclamp(q_lo + 3 * KC);
    int xch = DEPTH ? xq % cinpad : xq;
    long long xso = stage_off(xq);
    const long long xstep = static_cast<long long>(KC) * D * HW * 4;                                   // KC channels on
    const long long xwrap = (static_cast<long long>(1) - static_cast<long long>(cinpad - KC) * D) * HW * 4;      // 3x3x3: first channel of the next depth tap
    auto stage = [&](int st, WSet& ws) __attribute__((always_inline)) {
      constexpr int kAhead = 6;                   // B operands: read this many half-steps ahead
      float bv[NH];
      if constexpr ((dbg & 32) != 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) bv[h] = static_cast<float>(lane + h);
      }
#ifdef ADV_WINO4_STAMPS
      unsigned long long stampx = 0;
      const int hsel = st % NH;
#endif
      const int x1 = xr == 2 ? 0 : xr + 1, x2 = xr == 0 ? 2 : xr - 1;      // (st + 1) % 3, (st + 2) % 3
      const float* bp = svb + (st & 1) * G::kSV + half * NPT + l32;          // V_k[q = 2 kp + half (+ 4 sub)][patch = 32 pb + l32]
      float* const vdst = svb + ((st + 1) & 1) * G::kSV + pair;
      const float* const tile = sxb + x1 * G::kSX + toff + (HI ? G::kPitch : 0);
      // half-step h = 18 sub + m:  m < 16: slot m / 4, kp = (m / 2) % 2, block m % 2;  m = 16, 17: the shared position, kp = m - 16
      auto need_b = [](int h) { const int m = h % 18; return CB == 1 || m >= 16 || (m & 1) == 0; };       // does this half-step start a new B operand?
      auto load_b = [&](int h) {
        const int sub = h / 18, m = h % 18;
        const int slot = m < 16 ? m / 4 : 4, kp = m < 16 ? (m >> 1) & 1 : m - 16, blk = m < 16 ? m & 1 : (HI ? 1 : 0);
        bv[h] = bp[kpos(slot) * 256 + (kKQ * sub + 2 * kp) * NPT + (CB == 1 ? 32 * blk : 0)];
      };
      const int qnoff = (CB == 2 ? qclamp(q_lo + (st + 1) * KC) : qclamp(q_lo + (st + 2) * KC)) / kKQ * wstep;      // the stage whose weights this stage requests
#pragma unroll
      for (int h = 0; h < kAhead; ++h)
        if (need_b(h) && !(dbg & 32)) load_b(h);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int sub = h / 18, m = h % 18;
#ifdef ADV_WINO4_STAMPS
        if (h == hsel) stampx = __builtin_amdgcn_s_memtime();
#endif
        if (h + kAhead < NH && need_b(h + kAhead) && !(dbg & 32)) load_b(h + kAhead);
        if (h >= 1 && h < 1 + G::kXSl && !(dbg & 4)) dma_x(h - 1, xso, xr);
        if constexpr (!(dbg & 1)) {
          if (h % CB == 0) {                      // transform event e = h / CB (0 .. 17)
            const int e = h / CB;
            if (e < 5) tr_read(e, tile);
            if (e >= 5 && e < 11) tr_cols(hi_c, e - 5);
            if (e >= 11 && e < 14) tr_row(e - 11);
            if (e >= 12) {                        // row r's six values exist from event 11 + r on: three writes per event, 12 .. 17
#pragma unroll
              for (int u = 0; u < 3; ++u) tr_write(3 * (e - 12) + u, vdst);
            }
          }
        }
        if constexpr ((dbg & 2) != 0) {
        } else if constexpr (CB == 2) {
          if (m < 16) {
            const int slot = m / 4, kp = (m >> 1) & 1, cb = m & 1;
            acc[2 * slot + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.f[sub][slot][2 * cb + kp], bv[h & ~1], acc[2 * slot + cb], 0, 0, 0);
          } else {
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.s[sub][m - 16], bv[h], acc[8], 0, 0, 0);
          }
        } else {
          if (m < 16) {
            const int slot = m / 4, kp = (m >> 1) & 1, pb = m & 1;
            acc[2 * slot + pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.f[sub][slot][kp], bv[h], acc[2 * slot + pb], 0, 0, 0);
          } else {
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.s[sub][m - 16], bv[h], acc[8], 0, 0, 0);
          }
        }
        if (!(dbg & 64)) {
          if (m < 16 && (m & 3) == 3) load_w(sub, m / 4, qnoff + sub * wstep, ws);
          if (m == 17) load_w(sub, 4, qnoff + sub * wstep, ws);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef ADV_WINO4_STAMPS
      const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();
#endif
      // the wave's pieces of the tile of stage st + 2 (requested during stage st - 1) have landed: younger than they are at most this
      // stage's pieces and the weight loads of two stages (a plain __syncthreads() would wait for every load in flight)
      if constexpr (!(dbg & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::kXMin + 10 * NSUB - 2) : "memory");
      if constexpr (!(dbg & 8)) {
        if (wave_fixes) {
#pragma unroll
          for (int i = 0; i < G::kXSl; ++i) fix_x(i, x2);
        }
      }
#ifdef ADV_WINO4_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const unsigned long long stamp4 = __builtin_amdgcn_s_memtime();
      if (stamped && st < 64 && lane == 0) {
        unsigned long long* o = adv_wino4_stamps[blockIdx.x][wave][st];
        o[0] = stamp0, o[1] = stampx, o[2] = stamp2, o[3] = stamp3, o[4] = stamp4;
      }
      stamp0 = stamp4;
#else
      if constexpr (!(dbg & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      xr = x1;
      if (xq < q_hi - KC) {                       // the next tile (beyond the last stage: the last one again)
        xq += KC, xch += KC;
        if (DEPTH && xch == cinpad) xch = 0, xso += xwrap;
        else xso += xstep;
      }
    };
    if constexpr (CB == 2) {
      for (int st = 0; st < nstage; ++st) stage(st, wsA);
    } else {
      int st = 0;
      for (; st + 1 < nstage; st += 2) {
        stage(st, wsA);
        stage(st + 1, wsB);
      }
      if (st < nstage) stage(st, wsA);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (clamped requests of the last stages: nothing may land later)
#ifdef ADV_WINO4_STAMPS
    const unsigned long long stamp_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: the 36 values M_k of one (channel, patch) sit in eight waves - exchange through LDS, 16 channels per round:
    // E[k][co16][patch]; register v of a 32 x 32 accumulator = channel (v & 3) + 8 (v >> 2) + 4 half of its block, patch = lane & 31
    float* const se = lds;
    auto eat = [&](int k, int co16, int patch) { return (k * 16 + co16) * NPT + patch; };      // E[k][co16][patch]
#ifdef ADV_WINO4_STAMPS
    unsigned long long est[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // round 0: start, writes issued, barrier passed, transformed, stores issued; then each round's end
    est[0] = __builtin_amdgcn_s_memtime();
#endif
    // every unit's (round, item) bias and the first unit's skip connection / mask are requested HERE, ahead of the exchange: in front of
    // a round's use they would wait for the round before's stores (vmcnt is in-order) - and a cold bias line alone is ~2 us.
    // Unconditional loads (a row / column / channel outside the tensor reads its first floats and is never stored)
    float bvu[2 * CB * PB], rvb[2][4][4], mvb[2][4][4];
    auto load_rm = [&](int unit, float (&rv)[4][4], float (&mv)[4][4]) {
      const int co = co0 + 16 * (unit / PB) + eg + (512 / NPT) * (unit % PB);
      const bool uok = co < Cout && gh0 < H && gw0 < W && eok;
      const long long at0 = eimg + static_cast<long long>(co) * DHW + static_cast<long long>(gh0) * W + gw0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rok = uok && gh0 + r < H;
        const long long at = at0 + static_cast<long long>(r) * W;
        if (vec4) {
          const v4f tr = resb ? *reinterpret_cast<const v4f*>(rok ? resb + at : epi.residual) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
          const v4f tm = maskb ? *reinterpret_cast<const v4f*>(rok ? maskb + at : epi.mask) : v4f{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
          for (int c = 0; c < 4; ++c) rv[r][c] = tr[c], mv[r][c] = tm[c];
        } else if (wide) {      // W % 4 != 0: the rows start at any float - one unaligned 16-byte access where the patch's four columns are in the map
          const v4fu tr = resb ? *reinterpret_cast<const v4fu*>(rok ? resb + at : epi.residual) : v4fu{0.0f, 0.0f, 0.0f, 0.0f};
          const v4fu tm = maskb ? *reinterpret_cast<const v4fu*>(rok ? maskb + at : epi.mask) : v4fu{1.0f, 1.0f, 1.0f, 1.0f};
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
    };
#pragma unroll
    for (int u = 0; u < 2 * CB * PB; ++u) {
      const int co = co0 + 16 * (u / PB) + eg + (512 / NPT) * (u % PB);
      bvu[u] = epi.bias ? epi.bias[co < Cout ? co : 0] : 0.0f;
    }
    if (resb || maskb) load_rm(0, rvb[0], mvb[0]);
#pragma unroll
    for (int round = 0; round < 2 * CB; ++round) {          // (unrolled: the accumulator registers are addressed by constants)
      if (co0 + 16 * round >= Cout) continue;               // (workgroup-uniform) nothing but padding from here on
#pragma unroll
      for (int slot = 0; slot < 5; ++slot) {
        const int k = kpos(slot);
#pragma unroll
        for (int v8 = 0; v8 < 8; ++v8) {
          const int co16 = (v8 & 3) + 8 * (v8 >> 2) + 4 * half;
          // (a register's two lane halves are channels co16 and co16 + 4, 128 or 256 floats apart; sending one half to the other 32 banks
          // - row parity flipped / patch column ^ 32 - measured SLOWER: 1 200 -> 1 500 cycles per round of writes, the lane-dependent
          // address costs more than it saves: profiles/r06_wino4_epilogue_stamps.jsonl)
          if constexpr (CB == 2) {
            if (slot < 4) se[eat(k, co16, l32)] = acc[2 * (slot & 3) + (round >> 1)][8 * (round & 1) + v8];
            else if ((round >> 1) == (HI ? 1 : 0)) se[eat(k, co16, l32)] = acc[8][8 * (round & 1) + v8];
          } else {
            if (slot < 4) {
              se[eat(k, co16, l32)] = acc[2 * (slot & 3)][8 * round + v8];
              se[eat(k, co16, 32 + l32)] = acc[2 * (slot & 3) + 1][8 * round + v8];
            } else {
              se[eat(k, co16, (HI ? 32 : 0) + l32)] = acc[8][8 * round + v8];
            }
          }
        }
      }
#ifdef ADV_WINO4_STAMPS
      if (round == 0) est[1] = __builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ADV_WINO4_STAMPS
      if (round == 0) est[2] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
      for (int item = 0; item < PB; ++item) {
        const int co16 = eg + (512 / NPT) * item;
        const int co = co0 + 16 * round + co16;
        constexpr int kUnits = 2 * CB * PB;
        const int unit = round * PB + item;      // (a constant after unrolling)
        float s[4][6], o[4][4];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          float col[4];
          at6(se[eat(0 * 6 + j, co16, ep)], se[eat(1 * 6 + j, co16, ep)], se[eat(2 * 6 + j, co16, ep)], se[eat(3 * 6 + j, co16, ep)],
              se[eat(4 * 6 + j, co16, ep)], se[eat(5 * 6 + j, co16, ep)], col);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r][j] = col[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) at6(s[r][0], s[r][1], s[r][2], s[r][3], s[r][4], s[r][5], o[r]);
#ifdef ADV_WINO4_STAMPS
        if (round == 0 && item == 0) {
          asm volatile("" ::"v"(o[0][0]), "v"(o[3][3]));
          est[3] = __builtin_amdgcn_s_memtime();
        }
#endif
        // the NEXT unit's skip connection / mask are requested before this unit's stores: vmcnt retires in order and counts stores, a load
        // behind them waits ~5 000 cycles for their acknowledgement (profiles/r06_wino4_epilogue_stamps.jsonl)
        if (unit + 1 < kUnits && (resb || maskb)) load_rm(unit + 1, rvb[(unit + 1) & 1], mvb[(unit + 1) & 1]);
        if (co < Cout && gh0 < H && gw0 < W && eok) {
          const float bv = bvu[unit];
          const long long at0 = eimg + static_cast<long long>(co) * DHW + static_cast<long long>(gh0) * W + gw0;
          const auto& rv = rvb[unit & 1];
          const auto& mv = mvb[unit & 1];
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
            } else if (wide) {
              *reinterpret_cast<v4fu*>(yb + at) = v4fu{res[0], res[1], res[2], res[3]};
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (gw0 + c < W) yb[at + c] = res[c];
            }
          }
        }
      }
#ifdef ADV_WINO4_STAMPS
      if (round == 0) est[4] = __builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ADV_WINO4_STAMPS
      est[5 + round] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef ADV_WINO4_STAMPS
    if (stamped && lane == 0) {
      unsigned long long* o1 = adv_wino4_stamps[blockIdx.x][wave][61];
      unsigned long long* o2 = adv_wino4_stamps[blockIdx.x][wave][62];
      unsigned long long* o0 = adv_wino4_stamps[blockIdx.x][wave][60];
#pragma unroll
      for (int i = 0; i < 5; ++i) o0[i] = pst[i];
#pragma unroll
      for (int i = 0; i < 5; ++i) o1[i] = est[i], o2[i] = est[5 + i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left
    const unsigned long long stamp_end = __builtin_amdgcn_s_memtime();
    if (stamped && lane == 0) {                           // [.][.][63]: kernel entry, first stage's start, loop end, kernel end
      unsigned long long* o = adv_wino4_stamps[blockIdx.x][wave][63];
      o[0] = stamp_entry, o[1] = stamp_loop_start, o[2] = stamp_loop_end, o[3] = stamp_end, o[4] = 0;
    }
#endif
  };
#ifdef ADV_WINO4_ONEBODY      // timing experiment only (wrong results): every wave runs the waves-0-3 body - half the code, the same work
  body(std::false_type{});
#else
  if (wave >= 4) body(std::true_type{});
  else body(std::false_type{});
#endif
}

int round_up4(int v, int q) { return (v + q - 1) / q * q; }

constexpr int kWaveFlags = 0;      // bit 0: the transform waves of a 64-channel workgroup run at priority 1, bit 1: the product waves
int wave_flags() {
  if (const char* e = adv_hook_value("ADV_WINO4_FLAGS")) return std::atoi(e);      // test hook / A-B; same bits
  return kWaveFlags;
}

constexpr int kKPad = 8;      // the contraction is padded to whole stages of the 64-channel shape (prepared weights and kernel alike)

template <int PR, int PC, int CB, bool DEPTH, bool PAIR = false>
int launch_wino4(const float* x, const float* wp, float* y, int b, int cin, int cout, int cinpad, int copad, int d, int h, int w, const Epi4& epi,
                 hipStream_t st) {
  using G = W4Geo<PR, PC, CB>;
  if (PAIR && (w > 15 || cin % kKPad != 0)) return ADV_EINVAL;
  const int tiles_w = PAIR ? 1 : (w + 4 * PC - 1) / (4 * PC), tiles_h = (h + 4 * PR - 1) / (4 * PR);
  const long long tiles = static_cast<long long>(tiles_w) * tiles_h;
  const int cgroups = (cout + G::kCO - 1) / G::kCO;
  if (epi.ksplit > 1 && (DEPTH || PAIR)) return ADV_EINVAL;
  const long long gz = PAIR ? (static_cast<long long>(b) + 1) / 2 : static_cast<long long>(b) * d * (epi.ksplit > 1 ? epi.ksplit : 1);
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
    if (!adv_internal_lds_limit<conv_wino4<PR, PC, CB, DEPTH, A_>>(G::kLds)) return ADV_ELAUNCH;                                           \
    hipLaunchKernelGGL((conv_wino4<PR, PC, CB, DEPTH, A_>), grid, dim3(512), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, tiles_w, \
                       wbytes, b, wave_flags(), epi);                                                                                                            \
    return adv_internal_finish_launch();                                                                                                        \
  }
      ADV_WINO4_ABL(1) ADV_WINO4_ABL(2) ADV_WINO4_ABL(4) ADV_WINO4_ABL(8) ADV_WINO4_ABL(16) ADV_WINO4_ABL(32) ADV_WINO4_ABL(64) ADV_WINO4_ABL(12) ADV_WINO4_ABL(93) ADV_WINO4_ABL(125) ADV_WINO4_ABL(127)
#undef ADV_WINO4_ABL
    }
  }
#endif
  if (!adv_internal_lds_limit<conv_wino4<PR, PC, CB, DEPTH, 0, PAIR>>(G::kLds)) return ADV_ELAUNCH;
  hipLaunchKernelGGL((conv_wino4<PR, PC, CB, DEPTH, 0, PAIR>), grid, dim3(512), G::kLds, st, x, wp, y, cin, cout, cinpad, copad, d, h, w, tiles_w, wbytes,
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
    case 0: return launch_wino4<4, 8, 2, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 1: return launch_wino4<2, 16, 2, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 2: return launch_wino4<8, 8, 1, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    case 3: return launch_wino4<4, 16, 1, DEPTH>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
    default:
      if constexpr (!DEPTH) return launch_wino4<4, 8, 2, false, true>(x, wp, y, b, cin, cout, cinpad, copad, d, h, w, epi, st);
      return ADV_EINVAL;
  }
}

// ---- K-split, second pass: y = epilogue(((p0 + p1) + p2) + ...) over the scratch planes the parts wrote (fixed order: reproducible bits)
template <int V>
__global__ void wino4_ksplit_sum(const float* __restrict__ part, float* __restrict__ y, long long n, int splits, int cout, long long hw, Epi4 epi) {
  using vf = std::conditional_t<V == 1, float, v4f>;
  for (long long i = (blockIdx.x * 256LL + threadIdx.x) * V; i < n; i += 256LL * V * gridDim.x) {
    vf v = *reinterpret_cast<const vf*>(part + i);
    for (int s = 1; s < splits; ++s) v = v + *reinterpret_cast<const vf*>(part + s * n + i);
    // (V = 4 needs only n % 4 == 0: four consecutive floats may straddle a channel boundary - the bias per element)
    const long long pl = i / hw;
    const int co = static_cast<int>(pl % cout), left = static_cast<int>(hw - (i - pl * hw));      // floats of this channel from i on
    const int co1 = co + 1 < cout ? co + 1 : 0;
    const float bv = epi.bias ? epi.bias[co] : 0.0f, bv1 = (V > 1 && epi.bias) ? epi.bias[co1] : 0.0f;
    vf r, m;
    if (epi.residual) r = *reinterpret_cast<const vf*>(epi.residual + i);
    if (epi.mask) m = *reinterpret_cast<const vf*>(epi.mask + i);
    vf o;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      float t;
      if constexpr (V == 1) t = v; else t = v[c];
      if (epi.bias) t = t + (c < left ? bv : bv1);
      if (epi.residual) { if constexpr (V == 1) t = t + r; else t = t + r[c]; }
      if (epi.relu) t = t > 0.0f ? t : 0.0f;
      if (epi.mask) { if constexpr (V == 1) t = m > 0.0f ? t : 0.0f; else t = m[c] > 0.0f ? t : 0.0f; }
      if constexpr (V == 1) o = t; else o[c] = t;
    }
    *reinterpret_cast<vf*>(y + i) = o;
  }
}

// channels per part: whole stages of the tile's shape (8 channels at 64 output channels per workgroup, 4 at 32), as even as the stages divide
int ksplit_chunk(int cinpad, int tile, int splits) {
  const int kc = (tile == 2 || tile == 3) ? 4 : 8;
  const int nst = cinpad / kc;
  return (nst + splits - 1) / splits * kc;
}

// auto rule: the most parts that still give every workgroup a compute unit of its own (one round of at most 256 workgroups with the
// flatter of the two 64-channel tiles), each part at least 4 stages, at most 8 parts.  Measured on the ResNet-101 stage-4 / 5 layers
// (profiles/r06_wino4_ksplit.jsonl): a workgroup's fixed cost (first tiles, output transform, stores) is ~12 us of its life - parts that
// double up on a compute unit cost more than they save, parts that fill idle ones are nearly free.
int pick_wino4_ksplit(int b, int cin, int cout, int h, int w) {
  const long long cg = (cout + 63) / 64;
  const long long t0 = static_cast<long long>((h + 15) / 16) * ((w + 31) / 32) * cg * b, t1 = static_cast<long long>((h + 7) / 8) * ((w + 63) / 64) * cg * b;
  const long long wgs = t0 < t1 ? t0 : t1;
  const int nst = round_up4(cin, 8) / 8;
  int s = wgs > 0 && wgs <= 128 ? static_cast<int>(256 / wgs) : 1;
  if (s > nst / 4) s = nst / 4;
  if (s > 8) s = 8;
  return s < 2 ? 1 : s;
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

// U = G g G^T (6 x 6) for every (output, input) channel pair (and depth tap), laid out [k = 6 i + j][(kd, c') / 4][m' / 64][lane][4] as the
// kernel's waves load it (zero rows / columns of padding).  forward: m = co, c = ci, g = w[co][ci][kd];  transpose (backward w.r.t. the input): m = ci, c = co, g = w[co][ci] with all
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
    // element (k, q = kd kpad + c, m) -> [k][q / 4][m / 64][lane = 32 (q & 1) + m % 32][2 (m / 32 % 2) + (q / 2 % 2)]: the float4 of lane
    // (half, l32) holds the A operands (cb, kp) of the stage's two matrix-instruction steps
    const int q = kd * kpad + c;
    const long long at = ((static_cast<long long>(q >> 2) * (mpad >> 6) + (m >> 6)) * 64 + 32 * (q & 1) + (m & 31)) * 4 + 2 * ((m >> 5) & 1) + ((q >> 1) & 1);
#pragma unroll
    for (int k = 0; k < 36; ++k) out[k * n + at] = u[k];
  }
}

int prep_wino4(const float* w, float* w_prep, int cout, int cin, int taps, int transpose, hipStream_t st) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(w_prep) & 15)) return ADV_EALIGN;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  const int kpad = round_up4(k, kKPad), mpad = round_up4(m, kCO);
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
  return 36LL * round_up4(k, kKPad) * round_up4(m, kCO);
}

int adv_conv2d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino4(w, w_prep, cout, cin, 1, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv2d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                         int cout, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || tile < -1 || tile > 4) return ADV_EINVAL;
  if (const int rc = check_wino4_args(x, w_prep, bias, residual, mask, y)) return rc;
  const Epi4 epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino4_tile<false>(tile >= 0 ? tile : pick_wino4_tile(cout, h, w, b, w <= 15 && cin % kKPad == 0 && b >= 2), x, w_prep, y, b, cin, cout, round_up4(cin, kKPad), round_up4(cout, kCO), 1, h, w,
                                  epi, static_cast<hipStream_t>(stream));
}

int adv_conv2d_wino4_ksplit_pick(int b, int cin, int cout, int h, int w) {
  if (b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1) return ADV_EINVAL;
  return pick_wino4_ksplit(b, cin, cout, h, w);
}

int adv_conv2d_wino4_ksplit_chunk(int cin, int tile, int splits) {
  if (cin < 1 || tile < 0 || tile > 3 || splits < 1) return ADV_EINVAL;
  return ksplit_chunk(round_up4(cin, kKPad), tile, splits);
}

int adv_conv2d_wino4_ksplit_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y,
                                float* scratch, int b, int cin, int cout, int h, int w, int relu, int tile, int splits, adv_stream_t stream) {
  if (!x || !w_prep || !y || !scratch || b < 1 || cin < 1 || cout < 1 || h < 1 || w < 1 || tile < -1 || tile > 3 || splits < 2 || splits > 64) return ADV_EINVAL;
  if (const int rc = check_wino4_args(x, w_prep, bias, residual, mask, y)) return rc;
  if ((reinterpret_cast<uintptr_t>(scratch) & 15) || scratch == y || scratch == x) return ADV_EALIGN;
  const int cinpad = round_up4(cin, kKPad);
  const int t = tile >= 0 ? tile : pick_wino4_tile(cout, h, w, static_cast<long long>(b) * splits, false);
  const int chunk = ksplit_chunk(cinpad, t, splits);
  const int parts = (cinpad + chunk - 1) / chunk;      // (no empty part)
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (parts < 2) {
    const Epi4 whole{bias, residual, mask, relu ? 1 : 0, 0, 0};
    return launch_wino4_tile<false>(t, x, w_prep, y, b, cin, cout, cinpad, round_up4(cout, kCO), 1, h, w, whole, st);
  }
  const Epi4 raw{nullptr, nullptr, nullptr, 0, parts, chunk};
  if (const int rc = launch_wino4_tile<false>(t, x, w_prep, scratch, b, cin, cout, cinpad, round_up4(cout, kCO), 1, h, w, raw, st)) return rc;
  const long long hw = static_cast<long long>(h) * w, n = hw * cout * b;
  const Epi4 epi{bias, residual, mask, relu ? 1 : 0, 0, 0};
  const bool v4 = n % 4 == 0 && hw >= 4 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0;
  const long long items = v4 ? n / 4 : n;
  long long blocks = (items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (v4) hipLaunchKernelGGL(wino4_ksplit_sum<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, scratch, y, n, parts, cout, hw, epi);
  else hipLaunchKernelGGL(wino4_ksplit_sum<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, scratch, y, n, parts, cout, hw, epi);
  return adv_internal_finish_launch();
}

int64_t adv_conv3d_wino4_prep_floats(int cout, int cin, int transpose) {
  if (cout < 1 || cin < 1) return ADV_EINVAL;
  const int k = transpose ? cout : cin, m = transpose ? cin : cout;
  return 108LL * round_up4(k, kKPad) * round_up4(m, kCO);
}

int adv_conv3d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  return prep_wino4(w, w_prep, cout, cin, 3, transpose, static_cast<hipStream_t>(stream));
}

int adv_conv3d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask, float* y, int b, int cin,
                         int cout, int d, int h, int w, int relu, int tile, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1 || tile < -1 || tile > 3) return ADV_EINVAL;
  if (const int rc = check_wino4_args(x, w_prep, bias, residual, mask, y)) return rc;
  const Epi4 epi{bias, residual, mask, relu ? 1 : 0};
  return launch_wino4_tile<true>(tile >= 0 ? tile : pick_wino4_tile(cout, h, w, static_cast<long long>(b) * d, false), x, w_prep, y, b, cin, cout, round_up4(cin, kKPad), round_up4(cout, kCO), d, h, w,
                                 epi, static_cast<hipStream_t>(stream));
}

}  // extern "C"
