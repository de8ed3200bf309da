// 3x3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the gfx950 float32 matrix cores.
//
//   D[cout][voxel] += W[cout][k] * X[k][voxel],   k = (input channel, tap)
// with v_mfma_f32_32x32x2_f32: the A operand is one weight per lane (row = output channel = lane & 31, k = lane >> 5),
// the B operand one input value per lane (k = lane >> 5, column = voxel = lane & 31), 16 accumulators per lane whose
// column index is the lane - so the epilogue writes, per accumulator register, 32 consecutive floats of one output
// channel (coalesced 128 B), and no transposition is ever needed.
//
// A workgroup (4 waves) owns an output tile of kTD x kTH rows of 32 voxels and 32 output channels; per chunk of kCK = 4
// input channels the input tile with its one-voxel halo and the 27 x kCK x 32 weights are staged in LDS, then every
// wave runs 27 * kCK/2 MFMAs on each of its kNB rows.  LDS reads are conflict-free by construction: the 32 lanes of a
// half-wave read 32 consecutive floats (the tap only shifts the start), the two halves read different channels.
// The accumulation order is fixed (chunk, tap, channel pair; the MFMA itself is a k-ordered fmaf chain), so results
// are reproducible and the oracle can match them bit for bit.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "adv_internal.h"
#include "advengine.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Where and how a workgroup's results are written: optional per-channel bias (folded batch-norm shift), optional ReLU,
// and an output lattice out[(i * os + oo)] per axis - the identity for an ordinary convolution, stride 2 with offset
// (pd, ph, pw) for one parity class of a transposed convolution.  tap_mask: bit t set = tap t = kd*9 + kh*3 + kw is used.
struct Epi {
  const float* bias;
  int relu;
  unsigned tap_mask;
  unsigned class_masks[8];  // class_channels > 0: input channels [k*class_channels, (k+1)*class_channels) use class_masks[k]
  int class_channels;       //                      (the eight parity sub-volumes of a space-to-depth input), else tap_mask
  int od, oh, ow;     // output tensor dims
  int sd, sh, sw;     // output lattice stride
  int fd, fh, fw;     // output lattice offset
  // nclass > 0: ONE launch computes nclass masked convolutions of the same input - class k with its own weights, tap mask and
  // output lattice offset (the eight output parity classes of a transposed convolution); tap_mask / fd,fh,fw are then unused
  int nclass;
  const float* cls_wp[8];
  unsigned cls_mask[8];
  int cls_off[8][3];
  // added to every result after the bias and before the ReLU: a tensor laid out like y (an hourglass's skip connection); may be y itself
  const float* residual;
  // <round 3> a tensor laid out like y, or null: the result is zeroed where mask <= 0, last of all (main stride-1 kernel only).  A backward
  // call with mask = the layer's own input (a ReLU output with this layer as its only consumer) returns the gradient w.r.t. the
  // producer's PRE-activation - no relu_backward pass over the volume.
  const float* mask;
  // <round 6> cls_wp[0..7] as ONE window for buffer addressing (the transposed kernel's weight requests): the lowest of the eight pointers
  // and the bytes from it to the end of the highest tensor, or null when they do not lie within 2 GiB of each other
  const float* cls_lo;
  unsigned cls_span;
};

constexpr unsigned kAllTaps = (1u << 27) - 1u;

__device__ __forceinline__ unsigned chunk_mask(const Epi& e, int c0) {
  return e.class_channels > 0 ? e.class_masks[c0 / e.class_channels] : e.tap_mask;
}

__device__ __forceinline__ void epi_store(const Epi& e, float* y, long long b, int Cout, int co, int gd, int gh, int gw, float r) {
  const int zd = gd * e.sd + e.fd, zh = gh * e.sh + e.fh, zw = gw * e.sw + e.fw;
  if (zd >= e.od || zh >= e.oh || zw >= e.ow) return;
  if (e.bias) r = r + e.bias[co];
  const long long at = ((b * Cout + co) * e.od + zd) * (static_cast<long long>(e.oh) * e.ow) + static_cast<long long>(zh) * e.ow + zw;
  if (e.residual) r = r + e.residual[at];
  if (e.relu) r = r > 0.0f ? r : 0.0f;
  if (e.mask) r = e.mask[at] > 0.0f ? r : 0.0f;      // <round 5> (the narrow-input kernel's route: the adjoint of a 32 -> 1 score layer behind a ReLU)
  y[at] = r;
}

constexpr int kCK = 4;    // input channels per LDS stage (both kernels: one accumulation order)
constexpr int kTD = 2;    // tile depth
constexpr int kTH = 8;    // tile height
constexpr int kTW = 32;   // tile width = MFMA N
constexpr int kNB = (kTD * kTH) / 4;  // rows per wave
constexpr int kSW = 27 * kCK * 32;

// input tile of the generic kernel for convolution stride S: an output tile of kTD x kTH x 32 voxels reads
// (S*(kTD-1)+3) x (S*(kTH-1)+3) x (S*31+3) inputs per channel
template <int S>
struct GenGeo {
  static constexpr int kID = S * (kTD - 1) + 3, kIH = S * (kTH - 1) + 3, kIW = S * (kTW - 1) + 3;
  static constexpr int kRow = kIW + 2;                                   // padded LDS row
  static constexpr int kSX = kCK * kID * kIH * kRow;
};

// S = 1: any W / alignment (the main kernel below handles W % 4 == 0); S = 2: the strided convolution of an hourglass
// (out = ceil(in / 2) per axis, padding 1).  D, H, W are the OUTPUT grid of the tile decomposition; iD, iH, iW the input dims.
template <int S>
__global__ __launch_bounds__(256) void conv3d_k3_mfma_generic(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                      int Cin, int Cout, int cout_pad, int D, int H, int W, int iD, int iH, int iW,
                                                      int tiles_w, int cblocks, Epi epi) {
  using GG = GenGeo<S>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sx = lds;
  float* sw = lds + GG::kSX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * kTW, h0 = ht * kTH, d0 = blockIdx.y * kTD;
  const int b = blockIdx.z / cblocks, cob = blockIdx.z - b * cblocks;
  const long long plane = static_cast<long long>(iH) * iW;
  const long long vol = plane * iD;

  f32x16 acc[kNB];
#pragma unroll
  for (int i = 0; i < kNB; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;

  for (int c0 = 0; c0 < Cin; c0 += kCK) {
    // ---- stage the input tile (+ halo, zero padded) of kCK channels
    for (int idx = tid; idx < kCK * GG::kID * GG::kIH * GG::kIW; idx += 256) {
      const int ww = idx % GG::kIW;
      int t = idx / GG::kIW;
      const int hh = t % GG::kIH;
      t /= GG::kIH;
      const int dd = t % GG::kID;
      const int c = t / GG::kID;
      const int gd = S * d0 + dd - 1, gh = S * h0 + hh - 1, gw = S * w0 + ww - 1;
      float v = 0.0f;
      if (gd >= 0 && gd < iD && gh >= 0 && gh < iH && gw >= 0 && gw < iW)
        v = x[(static_cast<long long>(b) * Cin + c0 + c) * vol + gd * plane + static_cast<long long>(gh) * iW + gw];
      sx[((c * GG::kID + dd) * GG::kIH + hh) * GG::kRow + ww] = v;
    }
    // ---- stage the weights of this chunk: [27][kCK][32]
    for (int idx = tid; idx < kSW; idx += 256) {
      const int n = idx & 31;
      const int c = (idx >> 5) % kCK;
      const int tap = idx / (32 * kCK);
      sw[idx] = wp[(static_cast<long long>(tap) * Cin + c0 + c) * cout_pad + cob * 32 + n];
    }
    __syncthreads();
    const unsigned mask = chunk_mask(epi, c0);
#pragma unroll 1
    for (int tap = 0; tap < 27; ++tap) {
      if (!((mask >> tap) & 1u)) continue;  // uniform: an unused tap of a parity class
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
      for (int kk = 0; kk < kCK / 2; ++kk) {
        const int c = 2 * kk + half;
        const float a = sw[(tap * kCK + c) * 32 + l32];
#pragma unroll
        for (int i = 0; i < kNB; ++i) {
          const int row = wave * kNB + i;
          const int td = row / kTH, th = row - td * kTH;
          const float bv = sx[((c * GG::kID + S * td + kd) * GG::kIH + S * th + kh) * GG::kRow + S * l32 + kw];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- epilogue: accumulator register v of lane l is D[cout = 8*(v/4) + 4*(l/32) + v%4][voxel = l%32]
  const int gw = w0 + l32;
#pragma unroll
  for (int i = 0; i < kNB; ++i) {
    const int row = wave * kNB + i;
    const int td = row / kTH, th = row - td * kTH;
    const int gd = d0 + td, gh = h0 + th;
    if (gd >= D || gh >= H || gw >= W) continue;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int co = cob * 32 + 8 * (v >> 2) + 4 * half + (v & 3);
      if (co < Cout) epi_store(epi, y, b, Cout, co, gd, gh, gw, acc[i][v]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// main kernel (W % 4 == 0): the same tiling with
//   * 16-byte global loads and ds_write_b128 for the interior of the input tile (the LDS row is laid out so that the
//     interior starts 16-byte aligned: [3 pad | left halo | 32 interior | right halo | 3 pad]), scalars only for
//     the two halo columns;
//   * the NEXT chunk's input tile and weights fetched into registers before the MFMA loop of the current chunk (a dozen
//     unconditional loads at offsets precomputed once per tile - struct Plan) and written to the other LDS buffer two thirds
//     of the way through it (global latency and the LDS stores hidden behind ~27k cycles of matrix work per chunk);
//   * the tap loop unrolled over kw so that LDS operand reads run ahead of the MFMAs that consume them.
// ---------------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_u __attribute__((aligned(4)));  // a float4 the compiler may not assume 16-byte aligned

constexpr int kFC = 4;      // input channels per stage of the main kernel
constexpr int kP = 40;      // padded LDS row, interior at column 4
constexpr int kFSW = 27 * kFC * 32;  // weights per stage (floats)
constexpr int kWF4 = kFSW / 4;       // 864 float4

// compile-time geometry of the main kernel for a tile depth TD (waves = 2*TD, 4 rows of 32 voxels per wave)
template <int TD, int TH = 8>
struct Geo {
  static constexpr int kThreads = 256;                                              // four waves, whatever the tile shape
  static constexpr int kRows = kFC * (TD + 2) * (TH + 2);                          // tile rows per stage
  static constexpr int kRowsPerPass = kThreads / 8;
  static constexpr int kXPass = (kRows + kRowsPerPass - 1) / kRowsPerPass;          // float4 per lane
  static constexpr int kHaloPerLane = (2 * kRows + kThreads - 1) / kThreads;
  static constexpr int kWPass = (kWF4 + kThreads - 1) / kThreads;
  static constexpr int kSX = kRows * kP;                                            // one input buffer (floats)
  static constexpr int kStageFloats = kSX + kFSW;                                   // one LDS stage; two are resident
};

template <int TD, int TH = 8>
struct Stage {
  v4f xi[Geo<TD, TH>::kXPass];
  float xh[Geo<TD, TH>::kHaloPerLane];
  v4f wv[Geo<TD, TH>::kWPass];
};

// What a lane fetches for every stage, worked out ONCE per tile: only the channel base moves from stage to stage, so the
// per-stage fetch is a handful of unconditional loads at precomputed offsets (no index arithmetic, no branches - the compiler
// is free to sink them into the MFMA stream), and everything conditional (zero padding, rows that end inside a float4) is a
// select at commit time, when the data has long arrived.
//   xo[p]   element offset of the float4 of tile row p*kRowsPerPass + tid/8, columns 4*(tid%8) .. +3, relative to the stage's
//           first channel; clamped so that the 16 bytes never leave the row (a row that ends inside the group is loaded
//           from W-4 and shifted left by k = 1..3 at commit); 0 for a group that is entirely padding
//   xk/xok  2 bits / 1 bit per pass: the shift, and "not padding"
//   ho/hok  the same for the two halo columns (one float each); lanes beyond 2*kRows park their value in an unused pad column
//   wo[p]   element offset of the lane's weight float4 relative to wp + c0*cout_pad + cob*32
template <int TD, int TH = 8>
struct Plan {
  int xo[Geo<TD, TH>::kXPass];
  int ho[Geo<TD, TH>::kHaloPerLane];
  int wo[Geo<TD, TH>::kWPass];
  unsigned xk, xok, hok;
};

template <int TD, bool WEIGHTS = true, int TH = 8>
__device__ __forceinline__ void make_plan(Plan<TD, TH>& pl, int tid, int Cin, int cout_pad, int D, int H, int W, int d0, int h0, int w0,
                                          int plane, int vol) {
  using G = Geo<TD, TH>;
  static_assert(G::kRows % G::kRowsPerPass == 0, "every pass of the interior fetch is full");
  constexpr int kPerC = (TD + 2) * (TH + 2);
  const int j = tid & 7, r0 = tid >> 3;
  pl.xk = 0, pl.xok = 0, pl.hok = 0;
#pragma unroll
  for (int p = 0; p < G::kXPass; ++p) {
    const int row = p * G::kRowsPerPass + r0;
    const int c = row / kPerC, rem = row - c * kPerC;
    const int dd = rem / (TH + 2), hh = rem - dd * (TH + 2);
    const int gd = d0 + dd - 1, gh = h0 + hh - 1, gw = w0 + 4 * j;
    const bool ok = gd >= 0 && gd < D && gh >= 0 && gh < H && gw < W;
    const int gws = gw + 3 < W ? gw : W - 4;
    pl.xo[p] = ok ? c * vol + gd * plane + gh * W + gws : 0;
    pl.xok |= (ok ? 1u : 0u) << p;
    pl.xk |= (ok ? static_cast<unsigned>(gw - gws) : 0u) << (2 * p);
  }
#pragma unroll
  for (int p = 0; p < G::kHaloPerLane; ++p) {
    const int s = p * G::kThreads + tid;
    const bool real = s < 2 * G::kRows;
    const int row = (real ? s : s - 2 * G::kRows) >> 1, side = s & 1;
    const int c = row / kPerC, rem = row - c * kPerC;
    const int dd = rem / (TH + 2), hh = rem - dd * (TH + 2);
    const int gd = d0 + dd - 1, gh = h0 + hh - 1, gw = side ? w0 + kTW : w0 - 1;
    const bool ok = real && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
    pl.ho[p] = ok ? c * vol + gd * plane + gh * W + gw : 0;
    pl.hok |= (ok ? 1u : 0u) << p;
  }
  if (!WEIGHTS) return;
#pragma unroll
  for (int p = 0; p < G::kWPass; ++p) {
    int f = p * G::kThreads + tid;
    if (f >= kWF4) f -= G::kThreads;           // the last pass is partial: its idle lanes repeat an earlier float4 (same data, same slot)
    const int n4 = f & 7, c = (f >> 3) & (kFC - 1), tap = f / (8 * kFC);
    pl.wo[p] = (tap * Cin + c) * cout_pad + 4 * n4;
  }
}

template <int TD, bool WEIGHTS = true, int TH = 8>
__device__ __forceinline__ void stage_fetch(Stage<TD, TH>& st, const Plan<TD, TH>& pl, const float* __restrict__ xc, const float* __restrict__ wc) {
  using G = Geo<TD, TH>;
#pragma unroll
  for (int p = 0; p < G::kXPass; ++p) st.xi[p] = *reinterpret_cast<const v4f_u*>(xc + pl.xo[p]);  // dword-aligned float4 (any W)
#pragma unroll
  for (int p = 0; p < G::kHaloPerLane; ++p) st.xh[p] = xc[pl.ho[p]];
  if (!WEIGHTS) return;
#pragma unroll
  for (int p = 0; p < G::kWPass; ++p) st.wv[p] = *reinterpret_cast<const v4f*>(wc + pl.wo[p]);
}

template <int TD, bool WEIGHTS = true, int TH = 8>
__device__ __forceinline__ void stage_commit(const Stage<TD, TH>& st, const Plan<TD, TH>& pl, float* sx, float* sw, int tid) {
  using G = Geo<TD, TH>;
  const int j = tid & 7, r0 = tid >> 3;
#pragma unroll
  for (int p = 0; p < G::kXPass; ++p) {
    const int row = p * G::kRowsPerPass + r0;
    const v4f t = st.xi[p];
    const unsigned k = (pl.xk >> (2 * p)) & 3u;
    const bool ok = (pl.xok >> p) & 1u;
    v4f v;
    v.x = k == 0 ? t.x : (k == 1 ? t.y : (k == 2 ? t.z : t.w));
    v.y = k == 0 ? t.y : (k == 1 ? t.z : (k == 2 ? t.w : 0.0f));
    v.z = k == 0 ? t.z : (k == 1 ? t.w : 0.0f);
    v.w = k == 0 ? t.w : 0.0f;
    if (!ok) v = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<v4f*>(sx + row * kP + 4 + 4 * j) = v;
  }
#pragma unroll
  for (int p = 0; p < G::kHaloPerLane; ++p) {
    const int s = p * G::kThreads + tid;
    const bool real = s < 2 * G::kRows;
    const int row = (real ? s : s - 2 * G::kRows) >> 1, side = s & 1;
    sx[row * kP + (real ? (side ? 4 + kTW : 3) : side)] = ((pl.hok >> p) & 1u) ? st.xh[p] : 0.0f;   // idle lanes: pad columns 0 / 1
  }
  if (!WEIGHTS) return;
#pragma unroll
  for (int p = 0; p < G::kWPass; ++p) {
    int f = p * G::kThreads + tid;
    if (f >= kWF4) f -= G::kThreads;
    *reinterpret_cast<v4f*>(sw + 4 * f) = st.wv[p];
  }
}

// ---- staging by LDS-DMA (global_load_lds_dwordx4: global memory -> LDS with no register in between), W % 4 == 0.
// An LDS row of the input tile is 40 floats = the 40 floats [w0-4, w0+36) of the global row (the halo columns w0-1 and w0+32 sit at
// LDS columns 3 and 36, exactly where the register path puts them), so a stage's input tile is kRows*10 float4 that are contiguous
// in LDS, and its weights 27*4*8 float4 likewise: one wave-instruction moves 64 consecutive float4 (1 KiB), the lanes supplying
// their own global addresses.  Groups that lie outside the volume (zero padding in d/h, columns < 0 or >= W - whole groups,
// because W and w0 are multiples of 4) read a 16-byte zero page instead.  No staging registers, no selects, no ds_write.
__device__ __attribute__((aligned(16))) const float g_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

typedef __attribute__((address_space(3))) void lds_void;
// the same through a buffer descriptor: lane l's 16 bytes from base + voff (a 32-bit byte offset held in a register for the whole tile; a
// lane whose offset fails the range check - the marker kNoLane, channels past the last one - writes zeros).  The per-stage part of the
// address lives in the descriptor's base (scalar arithmetic): NO vector instruction per request.  Why that matters: a workgroup's waves
// issue their requests while the co-resident workgroup's waves stream matrix instructions, and on gfx950 a vector instruction of one wave
// and a matrix instruction of the other do not execute together (profiles/r06_mfma_valu_probe.json) - the 64-bit select + add per request
// of the pointer form made the ten requests of a stage take 4 600 of its 15 000 cycles (profiles/r06_s2_stamps.jsonl).
constexpr int kNoLane = static_cast<int>(0xFFFFFF00u);
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t r, int voff, float* lds_piece) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_piece, 16, voff, 0, 0, 0);
#else
  (void)r, (void)voff, (void)lds_piece;
#endif
}
__device__ __forceinline__ void glds16(const float* src, float* lds_piece) {  // lane l's 16 bytes land at lds_piece + 4*l floats
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, lds_piece, 16, 0, 0);
#else
  (void)src, (void)lds_piece;
#endif
}

template <int TD, int TH>
struct DmaGeo {
  using G = Geo<TD, TH>;
  static constexpr int kXF4 = G::kRows * 10;                    // float4 of the input tile
  static constexpr int kXInstr = (kXF4 + 63) / 64;              // 25 (2x8 rows) / 15 (2x4) / 23 (4x4: the last piece half used) wave-instructions
  static constexpr int kSXp = kXInstr * 256;                    // input area rounded up to whole 1 KiB pieces (= G::kSX for the 2-deep tiles)
  static constexpr int kWInstr = (kWF4 + 63) / 64;              // 14, the last one half used (its tail lands in the pad below)
  static constexpr int kWaves = G::kThreads / 64;
  static constexpr int kXPer = (kXInstr + kWaves - 1) / kWaves; // pieces a wave may own
  static constexpr int kWPer = (kWInstr + kWaves - 1) / kWaves;
  static constexpr int kStageFloats = kSXp + kWInstr * 256;     // weights area rounded up to whole pieces
};

template <int TD, int TH>
struct DmaPlan {   // per lane: element offsets of its float4 of every piece its wave owns; -1 = the zero page
  int xo[DmaGeo<TD, TH>::kXPer];
  int wo[DmaGeo<TD, TH>::kWPer];
};

template <int TD, int TH>
__device__ __forceinline__ void make_dma_plan(DmaPlan<TD, TH>& pl, int tid, int Cin, int cout_pad, int D, int H, int W, int d0, int h0, int w0,
                                              int plane, int vol) {
  using DG = DmaGeo<TD, TH>;
  constexpr int kPerC = (TD + 2) * (TH + 2);
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int p = 0; p < DG::kXPer; ++p) {
    const int q = 64 * (wave + DG::kWaves * p) + lane;   // float4 index inside the tile (pieces beyond kXInstr are never issued)
    const int row = q / 10, j = q - row * 10;
    const int c = row / kPerC, rem = row - c * kPerC;
    const int dd = rem / (TH + 2), hh = rem - dd * (TH + 2);
    const int gd = d0 + dd - 1, gh = h0 + hh - 1, gw = w0 - 4 + 4 * j;
    const bool ok = q < DG::kXF4 && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw + 3 < W;
    pl.xo[p] = ok ? c * vol + gd * plane + gh * W + gw : -1;
  }
#pragma unroll
  for (int p = 0; p < DG::kWPer; ++p) {
    const int f = 64 * (wave + DG::kWaves * p) + lane;
    const int n4 = f & 7, c = (f >> 3) & (kFC - 1), tap = f / (8 * kFC);
    pl.wo[p] = f < kWF4 ? (tap * Cin + c) * cout_pad + 4 * n4 : -1;
  }
}

template <int TD, int TH>
__device__ __forceinline__ void dma_issue(const DmaPlan<TD, TH>& pl, const float* __restrict__ xc, const float* __restrict__ wc, float* stage,
                                          int wave) {
  using DG = DmaGeo<TD, TH>;
#pragma unroll
  for (int p = 0; p < DG::kXPer; ++p) {
    const int k = wave + DG::kWaves * p;   // wave-uniform
    if (k < DG::kXInstr) {
      const float* src = pl.xo[p] >= 0 ? xc + pl.xo[p] : g_zero16;
      glds16(src, stage + k * 256);
    }
  }
#pragma unroll
  for (int p = 0; p < DG::kWPer; ++p) {
    const int k = wave + DG::kWaves * p;
    if (k < DG::kWInstr) {
      const float* src = pl.wo[p] >= 0 ? wc + pl.wo[p] : g_zero16;
      glds16(src, stage + DG::kSXp + k * 256);
    }
  }
}

// Tiles of the launch, linearised: (row tile, column tile) fastest, then the depth pair, then (batch element, block of 32 output
// channels).  PERSISTENT workgroups walk them: gridDim.x = min(tiles, 2 per CU), workgroup i takes a contiguous share of the
// tile range of "its" XCD (workgroups are dealt to the eight XCDs round-robin, so i % 8 is the XCD and each XCD's private L2 sees
// neighbouring tiles - which share halo rows and weights - at about the same time), and the software pipeline runs ACROSS tiles:
// the last stage of a tile fetches the first stage of the next one, so a workgroup pays the plan, the first global round trip and
// the launch itself once, not once per tile.
struct TileGrid {
  int tiles_w, tiles_hw, nd, cblocks, nclass;
  long long ntiles;
};

struct TilePos {
  int w0, h0, d0, b, cob, cls;
};

template <int TD, int TH = 8>
__device__ __forceinline__ TilePos tile_at(const TileGrid& tg, long long t) {
  TilePos p;
  p.cls = 0;
  if (tg.nclass > 0) {
    // the class is the fastest index, rotated every 64 tiles: a workgroup walks tiles first + k*step with step a multiple of 8,
    // and must not end up with one class only (a class with eight taps costs eight times a class with one)
    p.cls = static_cast<int>((t % tg.nclass + t / 64) % tg.nclass);
    t /= tg.nclass;
  }
  const int xy = static_cast<int>(t % tg.tiles_hw);
  t /= tg.tiles_hw;
  p.d0 = static_cast<int>(t % tg.nd) * TD;
  const int z = static_cast<int>(t / tg.nd);
  p.w0 = (xy % tg.tiles_w) * kTW;
  p.h0 = (xy / tg.tiles_w) * TH;
  p.b = z / tg.cblocks;
  p.cob = z - p.b * tg.cblocks;
  return p;
}

// MASKED = false: every tap, no branch in the unrolled tap loop (the ordinary convolution: operand reads run ahead of the
// MFMAs); MASKED = true: taps are skipped by epi.tap_mask (one parity class of a transposed convolution).
template <int TD, bool MASKED, int TH = 8, bool DMA = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                              float* __restrict__ y, int Cin, int Cout, int cout_pad, int D, int H, int W,
                                                              TileGrid tg, Epi epi) {
  using G = Geo<TD, TH>;
  // before which (kd,kh) group of the tap loop the next stage is written to the other LDS buffer - in the shadow of this
  // stage's MFMAs instead of after them (measured: 64->32 1.468 -> 1.437 ms, 32->64 1.395 -> 1.32 ms at 6; the masked
  // classes, with fewer MFMAs per stage, 0.60 -> 0.55 ms at 3; profiles/r02_conv3d_staging.jsonl)
  constexpr int kCommitAt = MASKED ? 3 : 6;
  constexpr int NB = TD * TH / 4;  // rows of 32 voxels per wave (four waves)
  constexpr int kSXo = DMA ? DmaGeo<TD, TH>::kSXp : G::kSX;   // where a stage's weights start
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int plane = H * W, vol = plane * D;  // < 2^31 (host-checked)

  // this workgroup's tiles: t = first + k * step for k < count
  long long first, step;
  int count;
  if ((gridDim.x & 7) == 0 && tg.ntiles >= 8) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const long long share = (tg.ntiles + 7) / 8, lo = xcd * share;
    const long long mine = lo + share <= tg.ntiles ? share : (tg.ntiles > lo ? tg.ntiles - lo : 0);
    first = lo + slot, step = per;
    count = slot < mine ? static_cast<int>((mine - slot + per - 1) / per) : 0;
  } else {
    first = blockIdx.x, step = gridDim.x;
    count = blockIdx.x < tg.ntiles ? static_cast<int>((tg.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
  }
  if (count == 0) return;

  // two LDS stages: while the waves run the MFMAs of stage `cur`, the next chunk travels global -> registers ->
  // the other stage; ONE barrier per chunk
  constexpr int kStage = DMA ? DmaGeo<TD, TH>::kStageFloats : G::kStageFloats;
  // (the register-staged path exists for the 2-deep tiles only: TDr keeps its templates instantiable when DMA is the path taken)
  constexpr int TDr = DMA ? 2 : TD;
  Stage<TDr, TH> st;
  Plan<TDr, TH> pl;
  DmaPlan<TD, TH> dpl;
  TilePos tp = tile_at<TD, TH>(tg, first);
  if constexpr (DMA)
    make_dma_plan<TD, TH>(dpl, tid, Cin, cout_pad, D, H, W, tp.d0, tp.h0, tp.w0, plane, vol);
  else
    make_plan<TDr, true, TH>(pl, tid, Cin, cout_pad, D, H, W, tp.d0, tp.h0, tp.w0, plane, vol);
  const float* xb = x + static_cast<long long>(tp.b) * Cin * vol;
  const float* wb = ((MASKED && epi.nclass > 0) ? epi.cls_wp[tp.cls] : wp) + tp.cob * 32;
  if constexpr (DMA) {
    dma_issue<TD, TH>(dpl, xb, wb, lds, wave);
  } else {
    stage_fetch<TDr, true, TH>(st, pl, xb, wb);
    stage_commit<TDr, true, TH>(st, pl, lds, lds + G::kSX, tid);
  }
  __syncthreads();   // (waits for the DMA too: the compiler drains vmcnt before a barrier)
  int cur = 0;
  for (int k = 0; k < count; ++k) {
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;
    const TilePos cp = tp;  // the tile being accumulated (its results are stored after the stage loop)
    for (int c0 = 0; c0 < Cin; c0 += kFC) {
      // what the idle buffer receives during this stage: the tile's next stage; on its last stage the FIRST stage of the
      // workgroup's next tile (new plan); on the very last stage of the workgroup the stage itself again (harmless) - so that the
      // body below has no branch and the loads, the commit's selects and LDS stores and the MFMAs are one basic block
      int cn = c0 + kFC;
      if (cn >= Cin) {
        if (k + 1 < count) {
          tp = tile_at<TD, TH>(tg, first + (k + 1) * step);
          if constexpr (DMA)
            make_dma_plan<TD, TH>(dpl, tid, Cin, cout_pad, D, H, W, tp.d0, tp.h0, tp.w0, plane, vol);
          else
            make_plan<TDr, true, TH>(pl, tid, Cin, cout_pad, D, H, W, tp.d0, tp.h0, tp.w0, plane, vol);
          xb = x + static_cast<long long>(tp.b) * Cin * vol;
          wb = ((MASKED && epi.nclass > 0) ? epi.cls_wp[tp.cls] : wp) + tp.cob * 32;
          cn = 0;
        } else {
          cn = c0;
        }
      }
      const float* sxc = lds + cur * kStage;
      const float* swc = sxc + kSXo;
      if constexpr (DMA)   // straight into the idle buffer; complete before the barrier that ends this stage
        dma_issue<TD, TH>(dpl, xb + static_cast<long long>(cn) * vol, wb + static_cast<long long>(cn) * cout_pad, lds + (cur ^ 1) * kStage, wave);
      else
        stage_fetch<TDr, true, TH>(st, pl, xb + static_cast<long long>(cn) * vol, wb + static_cast<long long>(cn) * cout_pad);
      // the loads are issued HERE, a hundred MFMAs before the commit that consumes them: without the fence the scheduler sinks
      // them next to their use and every wave sits in s_waitcnt vmcnt for a global-memory round trip per stage (PMC: 20 % of
      // the wave cycles parked, 10 % without the fetch)
      __builtin_amdgcn_sched_barrier(0);
      const unsigned mask = MASKED ? (epi.nclass > 0 ? epi.cls_mask[cp.cls] : chunk_mask(epi, c0)) : kAllTaps;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {  // (kd, kh); fully unrolled so that LDS operand reads run ahead of their MFMAs
        const int kd = t9 / 3, kh = t9 - kd * 3;
        if constexpr (!DMA) {
          if (t9 == kCommitAt) {  // the next stage goes to the OTHER buffer in the shadow of this stage's MFMAs
            float* nx = lds + (cur ^ 1) * kStage;
            stage_commit<TDr, true, TH>(st, pl, nx, nx + G::kSX, tid);
          }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = t9 * 3 + kw;
          if (MASKED && !((mask >> tap) & 1u)) continue;  // scalar branch, parity-class convolutions only
#pragma unroll
          for (int kk = 0; kk < kFC / 2; ++kk) {
            const int c = 2 * kk + half;
            const float a = swc[(tap * kFC + c) * 32 + l32];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
              const int row = wave * NB + i;
              const int td = row / TH, th = row - td * TH;
              const float bv = sxc[((c * (TD + 2) + td + kd) * (TH + 2) + th + kh) * kP + l32 + kw + 3];
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[i], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();
      cur ^= 1;
    }
    // ---- epilogue of tile cp: accumulator register v of lane l is D[cout = 8*(v/4) + 4*(l/32) + v%4][voxel = l%32].  The bias
    // values of the lane's 16 channels are fetched once, the lattice position once per row; a store is then an add, a max and
    // a pointer step.
    const int co0 = cp.cob * 32 + 4 * half;
    const bool has_bias = epi.bias != nullptr;
    float bz[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int co = co0 + 8 * (v >> 2) + (v & 3);
      bz[v] = (has_bias && co < Cout) ? epi.bias[co] : 0.0f;
    }
    const bool by_class = MASKED && epi.nclass > 0;
    const int fd = by_class ? epi.cls_off[cp.cls][0] : epi.fd, fh = by_class ? epi.cls_off[cp.cls][1] : epi.fh;
    const int fw = by_class ? epi.cls_off[cp.cls][2] : epi.fw;
    const int gw = cp.w0 + l32, zw = gw * epi.sw + fw;
    const long long ovol = static_cast<long long>(epi.od) * epi.oh * epi.ow;
    float* yb = y + (static_cast<long long>(cp.b) * Cout + co0) * ovol;
    const bool has_res = epi.residual != nullptr, has_mask = epi.mask != nullptr;
    const float* rb = epi.residual + (static_cast<long long>(cp.b) * Cout + co0) * ovol;
    const float* mb = epi.mask + (static_cast<long long>(cp.b) * Cout + co0) * ovol;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = wave * NB + i;
      const int td = row / TH, th = row - td * TH;
      const int gd = cp.d0 + td, gh = cp.h0 + th;
      const int zd = gd * epi.sd + fd, zh = gh * epi.sh + fh;
      if (gd >= D || gh >= H || gw >= W || zd >= epi.od || zh >= epi.oh || zw >= epi.ow) continue;
      const long long at = (static_cast<long long>(zd) * epi.oh + zh) * epi.ow + zw;
      float* yr = yb + at;
      if (cp.cob * 32 + 32 <= Cout) {  // workgroup-uniform: all 32 output channels of the block exist - no per-store predicate
        if (has_res) {                 // the skip connection's 16 values of this row first (loads in flight together), then the stores
          const float* rr = rb + at;
          float sk[16];
#pragma unroll
          for (int v = 0; v < 16; ++v) sk[v] = __builtin_nontemporal_load(rr + (8 * (v >> 2) + (v & 3)) * ovol);
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            float r = acc[i][v];
            if (has_bias) r = r + bz[v];
            r = r + sk[v];
            if (epi.relu) r = r > 0.0f ? r : 0.0f;
            if (has_mask) r = mb[at + (8 * (v >> 2) + (v & 3)) * ovol] > 0.0f ? r : 0.0f;
            yr[(8 * (v >> 2) + (v & 3)) * ovol] = r;
          }
          continue;
        }
        if (has_mask) {                // the layer's own input as a ReLU mask: its 16 values first, then the stores
          const float* mr = mb + at;
          float mk[16];
#pragma unroll
          for (int v = 0; v < 16; ++v) mk[v] = __builtin_nontemporal_load(mr + (8 * (v >> 2) + (v & 3)) * ovol);
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            float r = acc[i][v];
            if (has_bias) r = r + bz[v];
            if (epi.relu) r = r > 0.0f ? r : 0.0f;
            yr[(8 * (v >> 2) + (v & 3)) * ovol] = mk[v] > 0.0f ? r : 0.0f;
          }
          continue;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          float r = acc[i][v];
          if (has_bias) r = r + bz[v];
          if (epi.relu) r = r > 0.0f ? r : 0.0f;
          yr[(8 * (v >> 2) + (v & 3)) * ovol] = r;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int cr = 8 * (v >> 2) + (v & 3);
          float r = acc[i][v];
          if (has_bias) r = r + bz[v];
          if (has_res && co0 + cr < Cout) r = r + rb[at + cr * ovol];
          if (epi.relu) r = r > 0.0f ? r : 0.0f;
          if (has_mask && co0 + cr < Cout) r = mb[at + cr * ovol] > 0.0f ? r : 0.0f;
          if (co0 + cr < Cout) yr[cr * ovol] = r;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Transposed convolution (kernel 3, stride 2, padding 1, output_padding 1) with ALL EIGHT output parity classes per tile.
// The class-as-tile-index launch above stages an input tile once per class and uses it for that class's 1-8 taps; here a
// workgroup stages the tile ONCE per 4-channel stage and runs all 27 kernel taps on it, each into the accumulator of the
// class it belongs to (per axis: kernel tap 1 -> even outputs from input offset 0; tap 2 -> odd outputs from offset 0; tap 0 ->
// odd outputs from offset +1) - 2.3x fewer staged floats per MFMA.  Tile = 1 x kTT x 32 INPUT voxels (a wave per row) -> a
// 2 x 2*kTT x 64 block of outputs x 32 channels; eight 32x32 accumulators per wave (128 VGPRs); the input tile needs a halo on
// the high side only.  Staging by LDS-DMA (W % 4 == 0), two stages.  A class's accumulation order is the one of its masked
// convolution (stage, input offset in lexicographic order = its taps in ascending order, channel pair), so the result has the
// same bits as the class launches and the oracle.  The weights are read from the eight per-class prepared tensors the entry
// point already takes: kernel tap k of class c sits at tap t(k) of cls_wp[c].
// ---------------------------------------------------------------------------------------------------------------
// <round 4> TD = input planes per tile, spread over the waves like the rows: a tile is TD x TT input voxels-rows with TD * TT waves
// (4 x 1, 2 x 2 or 1 x 4 for the flat hourglass volumes: 5 and 10 rows pad to 8 and 12 in 4-row tiles).  Same footprint per stage
// (2 planes x 5 rows, 3 x 3, 5 x 2 per channel), same accumulation order per output, same bits.
template <int TT, int TD = 1>                             // TT = input rows, TD = input planes per tile; TT * TD waves per workgroup
struct TGeo {
  static constexpr int kNW = TT * TD;                    // waves
  static constexpr int kRows = kFC * (TD + 1) * (TT + 1);  // tile rows per stage: 4 channels x (TD + 1) planes x (TT + 1) rows
  static constexpr int kXF4 = kRows * 10;                // 400 float4 (TT = 4)
  static constexpr int kXInstr = (kXF4 + 63) / 64;       // 7 wave-instructions (the last one partly pad)
  static constexpr int kSX = kXInstr * 256;              // floats reserved for the input tile
  static constexpr int kWInstr = (kWF4 + 63) / 64;       // 14
  static constexpr int kStage = kSX + kWInstr * 256;     // 5376 floats = 21 KiB per stage (TT = 4)
  static constexpr int kXPer = (kXInstr + kNW - 1) / kNW, kWPer = (kWInstr + kNW - 1) / kNW;
};

__device__ __forceinline__ constexpr int tp_off(int a) { return a == 2 ? 1 : 0; }     // input offset of per-axis choice a
__device__ __forceinline__ constexpr int tp_par(int a) { return a == 0 ? 0 : 1; }     // output parity
__device__ __forceinline__ constexpr int tp_tap(int a) { return a == 2 ? 2 : 1; }     // tap of the CLASS convolution (offset + 1)

// <round 4> DMA = false: the stage goes global -> registers -> LDS (dword-aligned 16-byte loads, the row's last group masked element by
// element) - any width, any 4-byte aligned input; W = 78 (the cost volume at 1/16 resolution) used to fall back to the class-as-tile-index
// launch of the stride-1 kernel at 0.22 of the matrix peak.  Same LDS image of the tile, same accumulation order, same bits.
template <int kTT, int kTDt = 1, bool DMA = true>
__global__ __launch_bounds__(64 * kTT * kTDt, kTT * kTDt == 4 ? 2 : 1) void convt3d_k3_s2_mfma(const float* __restrict__ x, float* __restrict__ y, int Cin, int Cout,
                                                                  int cout_pad, int D, int H, int W, int tiles_w, int tiles_h, int cblocks,
                                                                  Epi epi) {
  using TG = TGeo<kTT, kTDt>;
  constexpr int kTXPer = TG::kXPer, kTWPer = TG::kWPer, kTXF4 = TG::kXF4, kTXInstr = TG::kXInstr, kTWInstr = TG::kWInstr, kTSX = TG::kSX,
                kTStage = TG::kStage, kNW = TG::kNW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wd = wave / kTT, wh = wave % kTT;              // this wave's input plane and row of the tile
  const int half = lane >> 5, l32 = lane & 31;
  const int plane = H * W, vol = plane * D;
  const int tiles_d = (D + kTDt - 1) / kTDt;
  int t = blockIdx.x;
  const int w0 = (t % tiles_w) * kTW;
  t /= tiles_w;
  const int h0 = (t % tiles_h) * kTT;
  t /= tiles_h;
  const int d0 = (t % tiles_d) * kTDt;
  t /= tiles_d;
  const int b = t / cblocks, cob = t - b * cblocks;
  const float* xb = x + static_cast<long long>(b) * Cin * vol;

  // what this lane moves per stage: float4 q = 64*k + lane of piece k (pieces wave, wave + 4, ...)
  int xo[kTXPer], xn[kTXPer];
  const float* wsrc[kTWPer];
#pragma unroll
  for (int p = 0; p < kTXPer; ++p) {
    const int q = 64 * (wave + kNW * p) + lane;
    const int row = q / 10, j = q - row * 10;
    const int c = row / ((kTDt + 1) * (kTT + 1)), rem = row - c * (kTDt + 1) * (kTT + 1);
    const int dd = rem / (kTT + 1), hh = rem - dd * (kTT + 1);
    const int gd = d0 + dd, gh = h0 + hh, gw = w0 - 4 + 4 * j;
    const bool ok = q < kTXF4 && j > 0 && gd < D && gh < H && (DMA ? gw + 3 < W : gw < W);   // j = 0 (columns left of the tile) is never read
    xo[p] = ok ? c * vol + gd * plane + gh * W + gw : -1;
    xn[p] = ok ? (W - gw < 4 ? W - gw : 4) : 0;                                    // valid floats of the group (register path: the row's end)
  }
  int xob[kTXPer], wob[kTWPer];                          // DMA: the groups' byte offsets (buffer form of the requests; weights: from epi.cls_lo)
#pragma unroll
  for (int p = 0; p < kTXPer; ++p) xob[p] = xo[p] >= 0 ? xo[p] * 4 : kNoLane;
#pragma unroll
  for (int p = 0; p < kTWPer; ++p) {
    const int f = 64 * (wave + kNW * p) + lane;
    const int n4 = f & 7, c = (f >> 3) & (kFC - 1), slot = f / (8 * kFC);           // slot = (ad*3 + ah)*3 + aw
    const int ad = slot / 9, ah = (slot / 3) % 3, aw = slot % 3;
    const int cls = (tp_par(ad) * 2 + tp_par(ah)) * 2 + tp_par(aw), tap = (tp_tap(ad) * 3 + tp_tap(ah)) * 3 + tp_tap(aw);
    const float* base = epi.cls_wp[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) base = cls == k ? epi.cls_wp[k] : base;
    wsrc[p] = f < kWF4 ? base + (static_cast<long long>(tap) * Cin + c) * cout_pad + cob * 32 + 4 * n4 : nullptr;
    wob[p] = (f < kWF4 && epi.cls_lo) ? static_cast<int>((base - epi.cls_lo) * 4 + ((static_cast<long long>(tap) * Cin + c) * cout_pad + cob * 32 + 4 * n4) * 4) : kNoLane;
  }
  auto issue = [&](int c0, float* stage) {
    // (DMA only: the launch keeps the image below 4 GiB) requests through buffer descriptors whose bases carry the stage's channel offset:
    // no vector instruction per request (see bdma16); the eight class tensors as one window where the caller allocated them together
    const long long xoff = static_cast<long long>(c0) * vol * 4;
    const __amdgpu_buffer_rsrc_t rxd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xb) + xoff), 0,
                                                                          static_cast<int>(static_cast<unsigned>(static_cast<long long>(Cin) * vol * 4 - xoff)), 0x00020000);
#pragma unroll
    for (int p = 0; p < kTXPer; ++p) {
      const int k = wave + kNW * p, vo = xob[p];
      if (k < kTXInstr) bdma16(rxd, vo, stage + k * 256);
    }
    if (epi.cls_lo) {
      const long long woff = static_cast<long long>(c0) * cout_pad * 4;
      const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(epi.cls_lo) + woff), 0,
                                                                            static_cast<int>(static_cast<unsigned>(static_cast<long long>(epi.cls_span) - woff)), 0x00020000);
#pragma unroll
      for (int p = 0; p < kTWPer; ++p) {
        const int k = wave + kNW * p, vo = wob[p];
        if (k < kTWInstr) bdma16(rwd, vo, stage + kTSX + k * 256);
      }
    } else {
#pragma unroll
      for (int p = 0; p < kTWPer; ++p) {
        const int k = wave + kNW * p;
        if (k < kTWInstr) glds16(wsrc[p] ? wsrc[p] + static_cast<long long>(c0) * cout_pad : g_zero16, stage + kTSX + k * 256);
      }
    }
  };

  typedef float tv4 __attribute__((ext_vector_type(4)));
  typedef tv4 tv4_u __attribute__((aligned(4)));
  tv4 rx[kTXPer], rw[kTWPer];                           // DMA = false: a stage's groups on their way to LDS
  auto fetch = [&](int c0) {
#pragma unroll
    for (int p = 0; p < kTXPer; ++p) {
      tv4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (wave + kNW * p < kTXInstr && xn[p] > 0) {
        const float* src = xb + static_cast<long long>(c0) * vol + xo[p];
        if (xn[p] == 4) {
          v = *reinterpret_cast<const tv4_u*>(src);
        } else {                                        // the row's last, partial group: its floats one by one, zeros beyond the row
          v.x = src[0];
          if (xn[p] > 1) v.y = src[1];
          if (xn[p] > 2) v.z = src[2];
        }
      }
      rx[p] = v;
    }
#pragma unroll
    for (int p = 0; p < kTWPer; ++p) {
      tv4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (wave + kNW * p < kTWInstr && wsrc[p]) v = *reinterpret_cast<const tv4*>(wsrc[p] + static_cast<long long>(c0) * cout_pad);
      rw[p] = v;
    }
  };
  auto commit = [&](float* stage) {
#pragma unroll
    for (int p = 0; p < kTXPer; ++p) {
      const int k = wave + kNW * p;
      if (k < kTXInstr) *reinterpret_cast<tv4*>(stage + k * 256 + 4 * lane) = rx[p];
    }
#pragma unroll
    for (int p = 0; p < kTWPer; ++p) {
      const int k = wave + kNW * p;
      if (k < kTWInstr) *reinterpret_cast<tv4*>(stage + kTSX + k * 256 + 4 * lane) = rw[p];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.0f;
  if constexpr (DMA) {
    issue(0, lds);
  } else {
    fetch(0);
    commit(lds);
  }
  __syncthreads();
  int cur = 0;
  for (int c0 = 0; c0 < Cin; c0 += kFC) {
    const int cn = c0 + kFC < Cin ? c0 + kFC : c0;      // the last stage fetches itself again (harmless): no branch in the body
    if constexpr (DMA) issue(cn, lds + (cur ^ 1) * kTStage);
    else fetch(cn);
    __builtin_amdgcn_sched_barrier(0);
    const float* sxc = lds + cur * kTStage;
    const float* swc = sxc + kTSX;
#pragma unroll
    for (int o = 0; o < 8; ++o) {                       // input offset (od, oh, ow), lexicographic
      const int od = o >> 2, oh = (o >> 1) & 1, ow = o & 1;
#pragma unroll
      for (int kk = 0; kk < kFC / 2; ++kk) {
        const int c = 2 * kk + half;
        const float bv = sxc[((c * (kTDt + 1) + wd + od) * (kTT + 1) + wh + oh) * kP + 4 + l32 + ow];
#pragma unroll
        for (int slot = 0; slot < 27; ++slot) {         // the kernel taps that read this offset: one per class they feed
          const int ad = slot / 9, ah = (slot / 3) % 3, aw = slot % 3;
          if (tp_off(ad) != od || tp_off(ah) != oh || tp_off(aw) != ow) continue;
          const int cls = (tp_par(ad) * 2 + tp_par(ah)) * 2 + tp_par(aw);
          const float a = swc[(slot * kFC + c) * 32 + l32];
          acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[cls], 0, 0, 0);
        }
      }
    }
    if constexpr (!DMA) commit(lds + (cur ^ 1) * kTStage);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: accumulator register v of lane l is D[cout = 8*(v/4) + 4*(l/32) + v%4][input voxel l%32]; the classes pw = 0 / 1
  // of an output row are neighbours in memory: one 8-byte store per lane, 256 contiguous bytes per half-wave
  const int gh = h0 + wh, gw = w0 + l32, gdi = d0 + wd;
  if (gh >= H || gw >= W || gdi >= D) return;
  const int co0 = cob * 32 + 4 * half;
  const long long ovol = static_cast<long long>(epi.od) * epi.oh * epi.ow;
  typedef float v2f __attribute__((ext_vector_type(2)));
  if (epi.bias == nullptr && epi.residual == nullptr && epi.mask == nullptr) {      // the adjoint of a strided convolution: nothing to fetch
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        float* yr = y + (static_cast<long long>(b) * Cout + co0) * ovol + (static_cast<long long>(2 * gdi + pd) * epi.oh + 2 * gh + ph) * epi.ow + 2 * gw;
        const int c0i = (pd * 2 + ph) * 2;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int cr = 8 * (v >> 2) + (v & 3);
          if (co0 + cr >= Cout) continue;
          v2f r = {acc[c0i][v], acc[c0i + 1][v]};
          if (epi.relu) r.x = r.x > 0.0f ? r.x : 0.0f, r.y = r.y > 0.0f ? r.y : 0.0f;
          *reinterpret_cast<v2f*>(yr + cr * ovol) = r;
        }
      }
    return;
  }
  float bz[16];      // the lane's 16 bias values, fetched once (a load inside the store loop would be a round trip per store)
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int co = co0 + 8 * (v >> 2) + (v & 3);
    bz[v] = (epi.bias != nullptr && co < Cout) ? epi.bias[co] : 0.0f;
  }
  const bool has_bias = epi.bias != nullptr, has_res = epi.residual != nullptr, has_mask = epi.mask != nullptr;
#pragma unroll
  for (int pd = 0; pd < 2; ++pd)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const long long at = (static_cast<long long>(b) * Cout + co0) * ovol + (static_cast<long long>(2 * gdi + pd) * epi.oh + 2 * gh + ph) * epi.ow + 2 * gw;
      const int c0i = (pd * 2 + ph) * 2;
      v2f sk[16];    // the skip connection's values of this output row, all in flight before the first store
      if (has_res) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int cr = 8 * (v >> 2) + (v & 3);
          sk[v] = co0 + cr < Cout ? __builtin_nontemporal_load(reinterpret_cast<const v2f*>(epi.residual + at + cr * ovol)) : (v2f){0.0f, 0.0f};
        }
      }
      v2f mk[16];    // <round 4> the mask (a ReLU output laid out like y): the result is zeroed where it is <= 0, last of all
      if (has_mask) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int cr = 8 * (v >> 2) + (v & 3);
          mk[v] = co0 + cr < Cout ? __builtin_nontemporal_load(reinterpret_cast<const v2f*>(epi.mask + at + cr * ovol)) : (v2f){1.0f, 1.0f};
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int cr = 8 * (v >> 2) + (v & 3);
        v2f r = {acc[c0i][v], acc[c0i + 1][v]};
        if (has_bias) r.x = r.x + bz[v], r.y = r.y + bz[v];
        if (has_res) r.x = r.x + sk[v].x, r.y = r.y + sk[v].y;
        if (epi.relu) r.x = r.x > 0.0f ? r.x : 0.0f, r.y = r.y > 0.0f ? r.y : 0.0f;
        if (has_mask) r.x = mk[v].x > 0.0f ? r.x : 0.0f, r.y = mk[v].y > 0.0f ? r.y : 0.0f;
        if (co0 + cr < Cout) *reinterpret_cast<v2f*>(y + at + cr * ovol) = r;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Strided convolution (kernel 3, stride 2, padding 1) DIRECTLY on the input.
// The space-to-depth route above runs 8 x Cin/4 short stages per tile (1-8 taps each) and every stage waits for its own round
// trip; here a stage is TWO input channels of the raw input tile - 3 planes x 9 rows x 72 columns for a 1 x 4 x 32 tile of
// outputs - and all 27 taps run on it for BOTH blocks of 32 output channels (with 32 channels or fewer: for two output rows per
// wave of a 1 x 8 x 32 tile): 54 MFMAs per wave and stage, as in the transposed kernel, the operand of output voxel w being read at
// LDS column 2w + kw: a stride-2 ds_read (2-way bank conflict, the LDS
// has the time).  No permuted copy of the input, no workspace.  Stages of two channels change the ACCUMULATION ORDER to
// (channel pair, tap, channel) - the oracle takes the stage size as a parameter; results differ from the 4-channel order in
// the last bits only (ops.conv3d_k3_s2_stage_channels tells which one a call uses).
// ---------------------------------------------------------------------------------------------------------------
#ifdef ADV_S2_STAMPS
// diagnostic builds only (tools/build_variant.sh s2stamps conv3d.hip -DADV_S2_STAMPS; tools/s2_stamps.py): s_memtime stamps of the strided kernel's
// stage loop - [workgroup < 8][wave][stage < 64][4]: stage start, requests issued, last matrix instruction issued, barrier passed
__device__ unsigned long long adv_s2_stamps[8][4][64][4];
extern "C" __attribute__((visibility("default"))) int adv_debug_s2_stamps(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(adv_s2_stamps), bytes < sizeof(adv_s2_stamps) ? bytes : sizeof(adv_s2_stamps)));
}
#endif
constexpr int kSC = 2;                                    // input channels per stage
constexpr int kSRow = 72;                                 // LDS row = global columns [2*w0 - 4, 2*w0 + 68)
// CB = blocks of 32 output channels per workgroup: 2 (cout > 32: a wave owns one output row, both blocks) or 1 (a wave owns two output
// rows of the one block) - either way two accumulators and 54 MFMAs per wave and stage
// <round 3> PD = output depth planes per workgroup.  A stage of the PD = 1 tile moves 30 KiB (3 input planes + 27 x 2 x 64 weights) for
// 54 MFMAs per wave - 8.9 bytes per cycle and workgroup, i.e. the kernel is bound by global -> LDS staging (two workgroups per CU ask for
// more than the ~10 B/clk/CU the path delivers), not by its 2-way LDS conflicts.  PD = 2 stages 5 input planes for 2 output planes and
// the same weights: 40 KiB for 108 MFMAs per wave, 5.8 bytes per cycle.  Only with two channel blocks per workgroup (cout > 32), where
// two 40 KiB stages x two workgroups are exactly the CU's 160 KiB, and only where the halved tile count still fills the chip.
// <round 4> WD = output depth planes spread over the four WAVES (1, 2 or 4): the tile is WD planes x 4 / WD rows (x kRW rows per wave)
// instead of 1 x 4.  The hourglass volumes are flat - 10 and 5 output rows at the 3D geometric volume's half and quarter resolution -
// and a 4-row tile pads them to 12 and 8 (17 % / 38 % of the matrix work on rows that do not exist); 2 x 2 and 4 x 1 tiles cover them
// with 10 and 5 (6 at 2 x 2).  The staged footprint is the same (3 planes x 9 rows, 5 x 5, 9 x 3 per channel) and so is every output's
// accumulation order (stage, tap, channel): the same bits whichever tile shape the launch picks.
template <int CB, int PD = 1, int WD = 1>
struct SGeo {
  static constexpr int kRW = 3 - CB;                      // output rows per wave
  static constexpr int kWR = 4 / WD;                      // waves along the rows
  static constexpr int kTH = kWR * kRW;                   // output rows per tile
  static constexpr int kTDp = PD * WD;                    // output planes per tile
  static constexpr int kRowsIn = 2 * kTH + 1;             // input rows per plane of the tile
  static constexpr int kPlanes = 2 * kTDp + 1;            // input planes of the tile
  static constexpr int kRows = kSC * kPlanes * kRowsIn;
  static constexpr int kXF4 = kRows * (kSRow / 4);        // 972 (CB = 2) / 1836 float4
  static constexpr int kXInstr = (kXF4 + 63) / 64;
  static constexpr int kSX = kXInstr * 256;
  static constexpr int kWF4 = 27 * kSC * 8 * CB;          // weights of a stage: 27 taps x 2 channels x 32*CB output channels
  static constexpr int kWInstr = (kWF4 + 63) / 64;
  static constexpr int kStage = kSX + kWInstr * 256;      // 30 KiB (CB = 2) / 36 KiB per stage; two stages resident
  static constexpr int kXPer = (kXInstr + 3) / 4, kWPer = (kWInstr + 3) / 4;
};

template <int CB, int PD = 1, int WD = 1>
__global__ __launch_bounds__(256, 2) void conv3d_k3_s2_mfma(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                            int Cin, int Cout, int cout_pad, int D, int H, int W, int gD, int gH, int gW,
                                                            int tiles_w, int tiles_h, int cgroups, Epi epi) {
  using SG = SGeo<CB, PD, WD>;
  constexpr int kRW = SG::kRW, kRowsIn = SG::kRowsIn, kCO = 32 * CB, kPlanes = SG::kPlanes;
  const int gDt = (gD + SG::kTDp - 1) / SG::kTDp;          // depth tiles
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wd = wave / SG::kWR, wr = wave % SG::kWR;      // this wave's plane (of WD) and row group (of 4 / WD) of the tile
  const int half = lane >> 5, l32 = lane & 31;
  const int plane = H * W, vol = plane * D;
  int t = blockIdx.x;
  const int w0 = (t % tiles_w) * kTW;
  t /= tiles_w;
  const int h0 = (t % tiles_h) * SG::kTH;
  t /= tiles_h;
  const int d0 = (t % gDt) * SG::kTDp;                     // first output plane of the tile
  t /= gDt;
  const int b = t / cgroups, cp = t - b * cgroups;
  const float* xb = x + static_cast<long long>(b) * Cin * vol;

  int xo[SG::kXPer], wo[SG::kWPer];
#pragma unroll
  for (int p = 0; p < SG::kXPer; ++p) {
    const int q = 64 * (wave + 4 * p) + lane;
    const int row = q / 18, j = q - row * 18;
    const int c = row / (kPlanes * kRowsIn), rem = row - c * kPlanes * kRowsIn;
    const int kd = rem / kRowsIn, r9 = rem - kd * kRowsIn;
    const int gd = 2 * d0 + kd - 1, gh = 2 * h0 + r9 - 1, gw = 2 * w0 - 4 + 4 * j;
    const bool ok = q < SG::kXF4 && j < 17 && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw + 3 < W;   // columns 68..71 are never read
    xo[p] = ok ? (c * vol + gd * plane + gh * W + gw) * 4 : kNoLane;            // byte offsets from the stage's first channel (the launch keeps the image below 4 GiB)
  }
#pragma unroll
  for (int p = 0; p < SG::kWPer; ++p) {
    const int f = 64 * (wave + 4 * p) + lane;
    const int n4 = f % (8 * CB), c = (f / (8 * CB)) & 1, tap = f / (16 * CB);
    const int co = cp * kCO + 4 * n4;
    wo[p] = (f < SG::kWF4 && co < cout_pad) ? ((tap * Cin + c) * cout_pad + co) * 4 : kNoLane;
  }
  const long long xbytes = static_cast<long long>(Cin) * vol * 4, wbytes = 27LL * Cin * cout_pad * 4;
  auto issue = [&](int c0, float* stage) {
    // the stage's channel pair moves the descriptors' bases; what is left behind them bounds the range check (a channel past Cin: zeros)
    const long long xoff = static_cast<long long>(c0) * vol * 4, woff = static_cast<long long>(c0) * cout_pad * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(xb) + xoff), 0,
                                                                         static_cast<int>(static_cast<unsigned>(xbytes - xoff)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(wp) + woff), 0,
                                                                         static_cast<int>(static_cast<unsigned>(wbytes - woff)), 0x00020000);
#pragma unroll
    for (int p = 0; p < SG::kXPer; ++p) {
      const int k = wave + 4 * p, vo = xo[p];
      if (k < SG::kXInstr) bdma16(rx, vo, stage + k * 256);
    }
#pragma unroll
    for (int p = 0; p < SG::kWPer; ++p) {
      const int k = wave + 4 * p, vo = wo[p];
      if (k < SG::kWInstr) bdma16(rw, vo, stage + SG::kSX + k * 256);
    }
  };

  f32x16 acc[PD][2];
#pragma unroll
  for (int pd = 0; pd < PD; ++pd)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[pd][i][v] = 0.0f;
  issue(0, lds);
  __syncthreads();
  int cur = 0;
  for (int c0 = 0; c0 < Cin; c0 += kSC) {
    const int cn = c0 + kSC < Cin ? c0 + kSC : c0;
#ifdef ADV_S2_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    issue(cn, lds + (cur ^ 1) * SG::kStage);
#ifdef ADV_S2_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_sched_barrier(0);
    const float* sxc = lds + cur * SG::kStage;
    const float* swc = sxc + SG::kSX;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      float a[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) a[cb] = swc[(tap * kSC + half) * kCO + cb * 32 + l32];
#pragma unroll
      for (int pd = 0; pd < PD; ++pd) {
        float bv[kRW];
#pragma unroll
        for (int r = 0; r < kRW; ++r)
          bv[r] = sxc[((half * kPlanes + 2 * (wd * PD + pd) + kd) * kRowsIn + 2 * (wr * kRW + r) + kh) * kSRow + 3 + 2 * l32 + kw];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[pd][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[CB == 2 ? i : 0], bv[CB == 2 ? 0 : i], acc[pd][i], 0, 0, 0);
      }
    }
#ifdef ADV_S2_STAMPS
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef ADV_S2_STAMPS
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x < 8 && lane == 0 && c0 / kSC < 64) {
      unsigned long long* o = adv_s2_stamps[blockIdx.x][wave][c0 / kSC];
      o[0] = st0, o[1] = st1, o[2] = st2, o[3] = st3;
    }
#endif
    cur ^= 1;
  }

  const int gw = w0 + l32;
  if (gw >= gW) return;
  const long long ovol = static_cast<long long>(gD) * gH * gW;
  const bool has_bias = epi.bias != nullptr, has_res = epi.residual != nullptr;
#pragma unroll
  for (int pd = 0; pd < PD; ++pd)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cb = CB == 2 ? i : 0, r = CB == 2 ? 0 : i;
    const int gh = h0 + wr * kRW + r, gd = d0 + wd * PD + pd;
    const int co0 = cp * kCO + cb * 32 + 4 * half;
    if (gh >= gH || gd >= gD || cp * kCO + cb * 32 >= Cout) continue;
    const long long at = (static_cast<long long>(b) * Cout + co0) * ovol + (static_cast<long long>(gd) * gH + gh) * gW + gw;
    float bz[16], sk[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int cr = 8 * (v >> 2) + (v & 3);
      bz[v] = (has_bias && co0 + cr < Cout) ? epi.bias[co0 + cr] : 0.0f;
      sk[v] = (has_res && co0 + cr < Cout) ? __builtin_nontemporal_load(epi.residual + at + cr * ovol) : 0.0f;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int cr = 8 * (v >> 2) + (v & 3);
      float rv = acc[pd][i][v];
      if (has_bias) rv = rv + bz[v];
      if (has_res) rv = rv + sk[v];
      if (epi.relu) rv = rv > 0.0f ? rv : 0.0f;
      if (co0 + cr < Cout) y[at + cr * ovol] = rv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow layers on the vector ALUs.  The matrix kernel pads the output channels to 32 rows, so the LAST layer of a
// cost-volume network (32 -> 1: per-plane scores) would spend 31/32 of its MFMA work on zeros, and its adjoint (1 -> 32,
// Cin = 1) has K = 27: both are cheaper as plain fmaf chains.  Same accumulation order as the matrix kernel and the oracle
// (stage of 4 channels, tap, channel), so the results are bit-identical to what the padded matrix kernel produces.
//   narrow_out<CO>: Cout <= CO <= 8.  Input tile staged exactly as in the main kernel (two LDS stages); the stage's weights sit
//                   beside it as [tap][o][4 channels], read with one broadcast ds_read_b128 per (tap, o) (scalar loads were
//                   tried first: SMEM shares lgkmcnt with LDS and returns out of order, so every weight use drained the LDS
//                   queue - 0.84 ms against 0.65 ms for the padded matrix kernel); a thread owns the two depth slices of one
//                   (row, column) of the 2 x 8 x 32 tile and reuses every LDS operand for both.
//   narrow_in<CI> : Cin = CI < 4.  One thread per voxel and 32 output channels in registers: per tap one cached global load of
//                   the (small) input and 32 fmaf with the weights broadcast from LDS, then one coalesced store per output
//                   channel.  HBM-bound on writing the Cout-channel result.
// ---------------------------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256, 2) void conv3d_k3_narrow_out(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                               int Cin, int Cout, int cout_pad, int D, int H, int W, int tiles_w, Epi epi) {
  using G = Geo<2>;
  constexpr int kNW = 27 * kFC * CO;                 // weights per stage, LDS layout [tap][o][c]: one ds_read_b128 = the 4 channels
  constexpr int kWPer = (kNW + 255) / 256;
  constexpr int kStage = G::kSX + kNW;
  constexpr int kUnrollDD = 1;   // a runtime loop over the four input depth slices keeps the hoisted LDS reads inside the register budget
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, th = tid >> 5, l32 = tid & 31;
  const int wt = blockIdx.x % tiles_w, ht = blockIdx.x / tiles_w;
  const int w0 = wt * kTW, h0 = ht * kTH, d0 = blockIdx.y * 2;
  const int b = blockIdx.z;
  const long long plane = static_cast<long long>(H) * W;
  const long long vol = plane * D;
  float acc0[CO], acc1[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc0[o] = 0.0f, acc1[o] = 0.0f;
  Stage<2> st;
  float wn[kWPer];
  auto fetch_w = [&](int c0) {
#pragma unroll
    for (int p = 0; p < kWPer; ++p) {
      const int f = p * 256 + tid;                   // f = (tap*CO + o)*4 + c
      const int c = f & 3, o = (f >> 2) % CO, tap = f / (4 * CO);
      wn[p] = f < kNW ? wp[(static_cast<long long>(tap) * Cin + c0 + c) * cout_pad + o] : 0.0f;
    }
  };
  auto commit_w = [&](float* sw) {
#pragma unroll
    for (int p = 0; p < kWPer; ++p) {
      const int f = p * 256 + tid;
      if (f < kNW) sw[f] = wn[p];
    }
  };
  Plan<2> pl;
  make_plan<2, false>(pl, tid, Cin, cout_pad, D, H, W, d0, h0, w0, static_cast<int>(plane), static_cast<int>(vol));
  const float* xb = x + static_cast<long long>(b) * Cin * vol;
  stage_fetch<2, false>(st, pl, xb, nullptr);
  fetch_w(0);
  stage_commit<2, false>(st, pl, lds, nullptr, tid);
  commit_w(lds + G::kSX);
  __syncthreads();
  int cur = 0;
  for (int c0 = 0; c0 < Cin; c0 += kFC) {
    const bool more = c0 + kFC < Cin;
    const float* sxc = lds + cur * kStage;
    const float* swc = sxc + G::kSX;
    if (more) {
      stage_fetch<2, false>(st, pl, xb + (c0 + kFC) * vol, nullptr);
      fetch_w(c0 + kFC);
    }
#pragma unroll kUnrollDD
    for (int dd = 0; dd < 4; ++dd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float v[kFC];
#pragma unroll
          for (int c = 0; c < kFC; ++c) v[c] = sxc[((c * 4 + dd) * (kTH + 2) + th + kh) * kP + l32 + kw + 3];
          if (dd < 3) {
#pragma unroll
            for (int o = 0; o < CO; ++o) {
              const v4f w4 = *reinterpret_cast<const v4f*>(swc + ((dd * 9 + kh * 3 + kw) * CO + o) * 4);  // one address for the wave: LDS broadcast
              acc0[o] = __builtin_fmaf(w4.x, v[0], acc0[o]);
              acc0[o] = __builtin_fmaf(w4.y, v[1], acc0[o]);
              acc0[o] = __builtin_fmaf(w4.z, v[2], acc0[o]);
              acc0[o] = __builtin_fmaf(w4.w, v[3], acc0[o]);
            }
          }
          if (dd > 0) {
#pragma unroll
            for (int o = 0; o < CO; ++o) {
              const v4f w4 = *reinterpret_cast<const v4f*>(swc + (((dd - 1) * 9 + kh * 3 + kw) * CO + o) * 4);
              acc1[o] = __builtin_fmaf(w4.x, v[0], acc1[o]);
              acc1[o] = __builtin_fmaf(w4.y, v[1], acc1[o]);
              acc1[o] = __builtin_fmaf(w4.z, v[2], acc1[o]);
              acc1[o] = __builtin_fmaf(w4.w, v[3], acc1[o]);
            }
          }
        }
    if (more) {
      float* nx = lds + (cur ^ 1) * kStage;
      stage_commit<2, false>(st, pl, nx, nullptr, tid);
      commit_w(nx + G::kSX);
    }
    __syncthreads();
    cur ^= 1;
  }
  const int gh = h0 + th, gw = w0 + l32;
  if (gh >= H || gw >= W) return;
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    if (o < Cout && d0 < D) epi_store(epi, y, b, Cout, o, d0, gh, gw, acc0[o]);
    if (o < Cout && d0 + 1 < D) epi_store(epi, y, b, Cout, o, d0 + 1, gh, gw, acc1[o]);
  }
}

template <int CI>
__global__ __launch_bounds__(256, CI == 1 ? 4 : 2) void conv3d_k3_narrow_in(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                              int Cout, int cout_pad, int D, int H, int W, long long voxels, Epi epi) {
  __shared__ __attribute__((aligned(16))) float sw[27 * CI * 32];  // one block of 32 output channels: [tap][c][32]
  constexpr int kUnrollKW = CI == 1 ? 3 : 1;  // more than one channel: keep the hoisted weight reads inside the register budget
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  const int b = blockIdx.y;
  const bool live = i < voxels;
  const int gw = static_cast<int>(i % W);
  const int gh = static_cast<int>((i / W) % H);
  const int gd = static_cast<int>(i / (static_cast<long long>(W) * H));
  const long long plane = static_cast<long long>(H) * W;
  const float* xb = x + static_cast<long long>(b) * CI * D * plane;
#pragma unroll 1
  for (int cb = 0; cb < cout_pad; cb += 32) {
    __syncthreads();
    for (int f = threadIdx.x; f < 27 * CI * 8; f += 256)
      *reinterpret_cast<v4f*>(sw + 4 * f) = *reinterpret_cast<const v4f*>(wp + static_cast<long long>(f >> 3) * cout_pad + cb + 4 * (f & 7));
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.0f;
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
        const int zd = gd + kd - 1, zh = gh + kh - 1;
        const bool row = live && zd >= 0 && zd < D && zh >= 0 && zh < H;
        const float* xr = xb + zd * plane + static_cast<long long>(zh) * W;
#pragma unroll kUnrollKW
        for (int kw = 0; kw < 3; ++kw) {
          const int zw = gw + kw - 1;
          const bool in = row && zw >= 0 && zw < W;
#pragma unroll
          for (int c = 0; c < CI; ++c) {
            const float v = in ? xr[c * D * plane + zw] : 0.0f;
            const float* wk = sw + (((kd * 3 + kh) * 3 + kw) * CI + c) * 32;  // the same address in every lane: LDS broadcast
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const v4f w4 = *reinterpret_cast<const v4f*>(wk + 4 * q);
              acc[4 * q + 0] = __builtin_fmaf(w4.x, v, acc[4 * q + 0]);
              acc[4 * q + 1] = __builtin_fmaf(w4.y, v, acc[4 * q + 1]);
              acc[4 * q + 2] = __builtin_fmaf(w4.z, v, acc[4 * q + 2]);
              acc[4 * q + 3] = __builtin_fmaf(w4.w, v, acc[4 * q + 3]);
            }
          }
        }
      }
    if (live) {
#pragma unroll
      for (int o = 0; o < 32; ++o)
        if (cb + o < Cout) epi_store(epi, y, b, Cout, cb + o, gd, gh, gw, acc[o]);
    }
  }
}

// xs[b, p*C + c, jd, jh, jw] = x[b, c, 2jd+pd, 2jh+ph, 2jw+pw], p = (pd*2+ph)*2+pw, zero beyond the input: the eight parity
// sub-volumes of x side by side in the channel dimension.  A stride-2 3x3x3 convolution of x is then a stride-1 convolution
// of xs in which sub-volume p uses only the taps its parity allows (27 taps over the eight classes - no wasted MFMA work),
// so the strided layers of an hourglass run on the tuned stride-1 kernel.  HBM-bound permute, writes coalesced.
__global__ __launch_bounds__(256) void space_to_depth2(const float* __restrict__ x, float* __restrict__ xs, int B, int C, int D, int H, int W,
                                                       int D2, int H2, int W2) {
  // one lane per (b, c, pd, ph, jd, jh, jw): it reads the x-pair (2jw, 2jw+1) of input row (2jd+pd, 2jh+ph) - consecutive lanes read
  // consecutive pairs, whole cache lines - and writes one float to each of the two sub-volumes pw = 0, 1 (both coalesced)
  const long long total = static_cast<long long>(B) * C * 4 * D2 * H2 * W2;
  const long long sub = static_cast<long long>(D2) * H2 * W2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int jw = static_cast<int>(i % W2);
    long long t = i / W2;
    const int jh = static_cast<int>(t % H2);
    t /= H2;
    const int jd = static_cast<int>(t % D2);
    t /= D2;
    const int q = static_cast<int>(t % 4);   // (pd, ph)
    t /= 4;
    const int c = static_cast<int>(t % C);
    const long long b = t / C;
    const int gd = 2 * jd + (q >> 1), gh = 2 * jh + (q & 1), gw = 2 * jw;
    float v0 = 0.0f, v1 = 0.0f;
    if (gd < D && gh < H) {
      const float* src = x + ((b * C + c) * D + gd) * (static_cast<long long>(H) * W) + static_cast<long long>(gh) * W + gw;
      if (gw + 1 < W) {
        if ((reinterpret_cast<uintptr_t>(src) & 7u) == 0) {
          const float2 v = *reinterpret_cast<const float2*>(src);
          v0 = v.x, v1 = v.y;
        } else {
          v0 = src[0], v1 = src[1];
        }
      } else if (gw < W) {
        v0 = src[0];
      }
    }
    float* dst = xs + ((b * 8 + 2 * q) * C + c) * sub + (static_cast<long long>(jd) * H2 + jh) * W2 + jw;  // sub-volume p = 2q + pw
    dst[0] = v0;
    dst[static_cast<long long>(C) * sub] = v1;
  }
}

__global__ void conv3d_k3_prep(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int transpose, int cin_p,
                               int cout_p, int cout_pad) {
  const long long total = 27LL * cin_p * cout_pad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int n = static_cast<int>(i % cout_pad);
    const int c = static_cast<int>((i / cout_pad) % cin_p);
    const int tap = static_cast<int>(i / (static_cast<long long>(cout_pad) * cin_p));
    float v = 0.0f;
    if (n < cout_p) {
      if (!transpose)
        v = w[(static_cast<long long>(n) * cin + c) * 27 + tap];         // W[co = n][ci = c][tap]
      else
        v = w[(static_cast<long long>(c) * cin + n) * 27 + (26 - tap)];  // W[co = c][ci = n][flipped tap]
    }
    wp[i] = v;
  }
}

}  // namespace

extern "C" {

int adv_conv3d_k3_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream) {
  if (!w || !w_prep || cout < 1 || cin < 1) return ADV_EINVAL;
  const int cin_p = transpose ? cout : cin, cout_p = transpose ? cin : cout;
  const int cout_pad = ((cout_p + 31) / 32) * 32;
  const long long total = 27LL * cin_p * cout_pad;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3d_k3_prep, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), w, w_prep, cout,
                     cin, transpose, cin_p, cout_p, cout_pad);
  return adv_internal_finish_launch();
}

static int cu_count() {  // compute units of the current device (256 on MI355X); queried once per thread
  static thread_local int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached = n;
  }
  return cached;
}

// the direct strided kernel (two-channel stages): rows of whole 16-byte groups, the convolution's own
// output grid, every tap.  adv_conv3d_k3_s2_stage_channels reports the choice (it fixes the accumulation order).
static bool conv3d_k3_s2_direct(int cout, int w, const float* x, const Epi& epi) {
  return cout >= 1 && w % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && epi.tap_mask == kAllTaps && epi.class_channels == 0 &&
         epi.nclass == 0 && epi.sd == 1 && epi.sh == 1 && epi.sw == 1 && epi.fd == 0 && epi.fh == 0 && epi.fw == 0 &&
         !adv_hook("ADV_CONV_GENERIC") && !adv_hook("ADV_CONV_S2_GENERIC");
}

static int launch_conv(const float* x, const float* w_prep, float* y, int b, int cin, int cout, int d, int h, int w, int stride,
                       const Epi& epi, hipStream_t st) {
  // (d, h, w) = input dims; the tile grid runs over the convolution's own output grid gd x gh x gw
  const int gd = stride == 2 ? (d + 1) / 2 : d, gh = stride == 2 ? (h + 1) / 2 : h, gw = stride == 2 ? (w + 1) / 2 : w;
  const int tiles_w = (gw + kTW - 1) / kTW, tiles_h = (gh + kTH - 1) / kTH;
  const int cblocks = (cout + 31) / 32;
  if (static_cast<long long>(b) * cblocks > 65535) return ADV_EINVAL;
  const bool plain = stride == 1 && epi.tap_mask == kAllTaps && epi.class_channels == 0;
  // the LDS-staged kernels' fetch plan holds 32-bit element offsets inside one batch element and loads 4 floats at a time:
  // w >= 4, cin*d*h*w and 27*cin*cout_pad below 2^31 - anything else takes the scalar-staging kernel
  const bool fits = w >= 4 && static_cast<long long>(cin) * d * h * w < (1LL << 31) && 27LL * cin * cblocks * 32 < (1LL << 31);
  const bool narrow_ok = !adv_hook("ADV_CONV_NO_NARROW");  // test hook: the padded matrix kernel instead
  if (epi.mask != nullptr && !(cin < kCK && plain)) {   // the main stride-1 matrix kernel and the narrow-input kernel have the mask epilogue: refuse every other route
    const bool main_route = stride == 1 && fits && cin >= kCK && !(plain && cout <= 8 && narrow_ok) && (reinterpret_cast<uintptr_t>(w_prep) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(x) & 3) == 0 && !adv_hook("ADV_CONV_GENERIC");
    if (!main_route) return ADV_EINVAL;
  }
  if (cin < kCK) {  // 1..3 input channels (the adjoint of a layer with 1..3 outputs): vector-ALU kernel, HBM-bound on the result
    if (!plain) return ADV_EINVAL;
    const long long voxels = static_cast<long long>(d) * h * w;
    const dim3 grid(static_cast<unsigned>((voxels + 255) / 256), b);
    if (cin == 1)
      hipLaunchKernelGGL((conv3d_k3_narrow_in<1>), grid, dim3(256), 0, st, x, w_prep, y, cout, cblocks * 32, d, h, w, voxels, epi);
    else if (cin == 2)
      hipLaunchKernelGGL((conv3d_k3_narrow_in<2>), grid, dim3(256), 0, st, x, w_prep, y, cout, cblocks * 32, d, h, w, voxels, epi);
    else
      hipLaunchKernelGGL((conv3d_k3_narrow_in<3>), grid, dim3(256), 0, st, x, w_prep, y, cout, cblocks * 32, d, h, w, voxels, epi);
    return adv_internal_finish_launch();
  }
  if (plain && fits && cout <= 8 && narrow_ok && (reinterpret_cast<uintptr_t>(x) & 3) == 0) {
    const dim3 grid(tiles_w * tiles_h, (d + 1) / 2, b);
    const int co_t = cout == 1 ? 1 : (cout <= 4 ? 4 : 8);
    const size_t lds = 2 * static_cast<size_t>(Geo<2>::kSX + 27 * kFC * co_t) * sizeof(float);
    if (cout == 1)
      hipLaunchKernelGGL((conv3d_k3_narrow_out<1>), grid, dim3(256), lds, st, x, w_prep, y, cin, cout, cblocks * 32, d, h, w, tiles_w, epi);
    else if (cout <= 4)
      hipLaunchKernelGGL((conv3d_k3_narrow_out<4>), grid, dim3(256), lds, st, x, w_prep, y, cin, cout, cblocks * 32, d, h, w, tiles_w, epi);
    else
      hipLaunchKernelGGL((conv3d_k3_narrow_out<8>), grid, dim3(256), lds, st, x, w_prep, y, cin, cout, cblocks * 32, d, h, w, tiles_w, epi);
    return adv_internal_finish_launch();
  }
  const bool fits32 = static_cast<long long>(cin) * d * h * w * 4 < 0xfff00000LL && 27LL * cin * cblocks * 32 * 4 < 0x7ff00000LL;   // byte offsets in 32 bits
  if (stride == 2 && conv3d_k3_s2_direct(cout, w, x, epi) && fits && fits32 && (reinterpret_cast<uintptr_t>(w_prep) & 15) == 0 && epi.od == gd &&
      epi.oh == gh && epi.ow == gw) {
    const bool two = cout > 32;   // both 32-channel blocks of a pair per workgroup, or two rows per wave of the one block
    const int tw = (gw + kTW - 1) / kTW, th = two ? (gh + SGeo<2>::kTH - 1) / SGeo<2>::kTH : (gh + SGeo<1>::kTH - 1) / SGeo<1>::kTH;
    const int cgroups = two ? (cblocks + 1) / 2 : cblocks;
    const long long ntiles = static_cast<long long>(tw) * th * gd * b * cgroups;
    if (ntiles < (1LL << 31)) {
      if (two) {
        // two output planes per workgroup (less staged per MFMA: measured 0.93 of the time per output on 64->128 at [192,20,304]) where
        // half as many tiles do not cost more rounds than that saves.  Rounds are counted per COMPUTE UNIT, not per resident workgroup:
        // two co-resident workgroups share the matrix pipes, so 576 tiles take three rounds of 256 CUs, and 288 double tiles two
        // double-length ones (a first version counted 512 slots and made the 16-GFLOP layers 15 % slower).
        // (ADV_CONV_S2_PD=1|2 forces one - test hook / A-B; same bits)
        // <round 4> flat volumes: spread the tile's four waves over 2 or 4 output planes where that pads fewer rows (same bits)
        // <round 6> tile shape by a makespan estimate instead of by padded volume alone (profiles/r06_s2_t2_tiles.jsonl: the old rule was 3-9 %
        // off the best shape on three of the four hourglass layers).  Two workgroups share a CU and its matrix pipes: a full "slot round" of
        // 2 x CUs tiles costs two tile times at ~0.9 (the pair keeps the pipe busy); a last partial round costs the same if some CU still
        // holds two tiles, and one tile time at ~1 / 0.65 if every CU holds at most one (a lone wave per SIMD cannot hide its operand
        // reads: 102 cycles per matrix instruction in the stamps, profiles/r06_s2_stamps.jsonl).  A tile's time = its output planes per
        // wave (PD) + 0.12 of fixed cost.  Ties go to the shape with more planes per tile (measured: never slower).
        const long long cus = cu_count();
        auto tiles_of = [&](int wdp, int pdp) {
          return static_cast<long long>(tw) * ((gh + 4 / wdp - 1) / (4 / wdp)) * ((gd + wdp * pdp - 1) / (wdp * pdp)) * b * cgroups;
        };
        auto estimate = [&](int wdp, int pdp) {
          const long long n = tiles_of(wdp, pdp), full = n / (2 * cus), rest = n - full * 2 * cus;
          const double t = pdp + 0.12;
          return static_cast<double>(full) * 2.0 * t * 0.9 + (rest == 0 ? 0.0 : (rest <= cus ? t / 0.65 : 2.0 * t * 0.9));
        };
        int wdv = 1;
        bool pd2 = false;
        {
          double best = estimate(1, 1);
          const int cand[4][2] = {{1, 2}, {2, 1}, {2, 2}, {4, 1}};
          for (const auto& c : cand) {
            const double e = estimate(c[0], c[1]);
            if (e <= best * 1.0001) best = e < best ? e : best, wdv = c[0], pd2 = c[1] == 2;
          }
        }
        if (const char* e = adv_hook_value("ADV_CONV_S2_WD")) wdv = e[0] == '4' ? 4 : (e[0] == '2' ? 2 : 1);
        const int thw = (gh + 4 / wdv - 1) / (4 / wdv);                   // row tiles of the chosen shape
        if (const char* e = adv_hook_value("ADV_CONV_S2_PD")) pd2 = e[0] == '2';
        pd2 = pd2 && wdv <= 2;
        const long long n1 = tiles_of(wdv, 1), n2 = tiles_of(wdv, 2);
#define ADV_LAUNCH_S2(PD_, WD_, N_)                                                                                                           \
  do {                                                                                                                                        \
    const size_t lds_ = 2 * sizeof(float) * static_cast<size_t>(SGeo<2, PD_, WD_>::kStage);                                                   \
    if (!adv_internal_lds_limit<conv3d_k3_s2_mfma<2, PD_, WD_>>(lds_)) return ADV_ELAUNCH;                                                    \
    hipLaunchKernelGGL((conv3d_k3_s2_mfma<2, PD_, WD_>), dim3(static_cast<unsigned>(N_)), dim3(256), lds_, st, x, w_prep, y, cin, cout,       \
                       cblocks * 32, d, h, w, gd, gh, gw, tw, thw, cgroups, epi);                                                             \
    return adv_internal_finish_launch();                                                                                                      \
  } while (0)
        if (n1 < (1LL << 31)) {
          if (pd2 && wdv == 1) ADV_LAUNCH_S2(2, 1, n2);
          if (pd2 && wdv == 2) ADV_LAUNCH_S2(2, 2, n2);
          if (wdv == 2) ADV_LAUNCH_S2(1, 2, n1);
          if (wdv == 4) ADV_LAUNCH_S2(1, 4, n1);
        }
#undef ADV_LAUNCH_S2
        hipLaunchKernelGGL(conv3d_k3_s2_mfma<2>, dim3(static_cast<unsigned>(ntiles)), dim3(256), 2 * sizeof(float) * static_cast<size_t>(SGeo<2>::kStage),
                           st, x, w_prep, y, cin, cout, cblocks * 32, d, h, w, gd, gh, gw, tw, th, cgroups, epi);
      } else {
        const size_t lds = 2 * sizeof(float) * static_cast<size_t>(SGeo<1>::kStage);   // 72 KiB: beyond the default dynamic-LDS limit
        if (!adv_internal_lds_limit<conv3d_k3_s2_mfma<1>>(lds)) return ADV_ELAUNCH;
        hipLaunchKernelGGL(conv3d_k3_s2_mfma<1>, dim3(static_cast<unsigned>(ntiles)), dim3(256), lds, st, x, w_prep, y, cin, cout, cblocks * 32, d, h, w,
                           gd, gh, gw, tw, th, cgroups, epi);
      }
      return adv_internal_finish_launch();
    }
  }
  // the main kernel takes every width (rows that are not 16-byte aligned are loaded as dword-aligned float4);
  // ADV_CONV_GENERIC=1 forces the scalar-staging kernel (kept as the reference implementation of the tiling)
  const bool fast = stride == 1 && fits && (reinterpret_cast<uintptr_t>(w_prep) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0 &&
                    !adv_hook("ADV_CONV_GENERIC");
  // tile depth 2 (4 waves) and 4 (8 waves, one workgroup per CU) measured the same within 1-2 % (profiles/r01_conv3d_mfma.jsonl)
  if (fast) {
    // persistent workgroups: two per CU (the LDS budget), a multiple of 8 so that each XCD walks its own contiguous tile range;
    // ADV_CONV_ONE_TILE_PER_WG=1 launches one workgroup per tile instead (test hook / A-B)
    const long long slots = 2LL * cu_count();
    const long long per_row_tile = static_cast<long long>(tiles_w) * ((d + 1) / 2) * b * cblocks * (epi.nclass > 0 ? epi.nclass : 1);
    // Tile height 8 (2 x 8 x 32 voxels, 4 rows per wave) or 4 (2 x 4 x 32, 2 rows per wave).  The half- and quarter-resolution
    // layers of an hourglass have 0.2-1.4 rounds of 8-row tiles for the resident workgroups; a 4-row tile costs 0.51 of an 8-row
    // one with all taps (measured at full resolution: 5760 tiles in 1.462 ms vs 2880 in 1.43) and ~0.75 with masked taps
    // (staging-dominated), and fills the tail - pick whichever needs less time by that estimate (ADV_CONV_TH=4|8 forces one:
    // test hook / A-B).  Same bits either way.
    const bool all_taps = epi.tap_mask == kAllTaps && epi.class_channels == 0 && epi.nclass == 0;
    const long long n8 = per_row_tile * ((h + 7) / 8), n4 = per_row_tile * ((h + 3) / 4);
    // rounds are counted per compute unit (two co-resident workgroups share its matrix pipes: 288 tiles are two rounds, not one)
    const long long cus = slots / 2;
    const double t8 = static_cast<double>((n8 + cus - 1) / cus);
    const double t4 = (all_taps ? 0.52 : 0.75) * static_cast<double>((n4 + cus - 1) / cus);
    // staging by LDS-DMA needs whole float4 groups inside the rows: w % 4 == 0 and a 16-byte aligned x (ADV_CONV_NO_DMA=1: the
    // register-staged path instead - test hook / A-B; same bits)
    const bool dma = w % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !adv_hook("ADV_CONV_NO_DMA");
    // <round 3> a third shape, 4 planes x 4 rows x 32 (the same 512 voxels as 2 x 8 x 32, 36 instead of 40 staged rows per channel):
    // for volumes whose height is a multiple of 4 but not of 8 - the 3D geometric volume's 20 / 10 rows, where 8-row tiles compute
    // 17-37 % padding and 4-row tiles stage half the voxels per tile (0.51 of peak on 32->64 at [192,20,304], 0.63 on 128->128 at
    // [96,10,152]: profiles/r03_conv3d_layers.jsonl).  All taps + LDS-DMA only.  ADV_CONV_TH=44 forces it.
    const long long n44 = static_cast<long long>(tiles_w) * ((d + 3) / 4) * b * cblocks * ((h + 3) / 4);
    const double t44 = (all_taps && dma) ? 0.98 * static_cast<double>((n44 + cus - 1) / cus) : 1e300;
    int tsel = t4 < t8 ? 4 : 8;
    if (t44 < (tsel == 4 ? t4 : t8)) tsel = 44;
    if (const char* e = adv_hook_value("ADV_CONV_TH")) tsel = (e[0] == '4' && e[1] == '4' && all_taps && dma) ? 44 : (e[0] == '4' ? 4 : 8);
    const bool th4 = tsel == 4;
    TileGrid tg;
    tg.tiles_w = tiles_w, tg.tiles_hw = tiles_w * (tsel == 8 ? (h + 7) / 8 : (h + 3) / 4), tg.nd = tsel == 44 ? (d + 3) / 4 : (d + 1) / 2, tg.cblocks = cblocks;
    tg.nclass = epi.nclass;
    tg.ntiles = tsel == 44 ? n44 : (th4 ? n4 : n8);
    long long wgs = slots;
    if (adv_hook("ADV_CONV_ONE_TILE_PER_WG") || tg.ntiles < wgs) wgs = tg.ntiles;
    if (wgs > 0x7fffffffLL) return ADV_EINVAL;
    const dim3 grid(static_cast<unsigned>(wgs));
    const int cpad = cblocks * 32;
#define ADV_LAUNCH_MFMA(TD_, MASKED_, TH_, DMA_)                                                                                              \
  do {                                                                                                                                        \
    const size_t lds_ = 2 * sizeof(float) * static_cast<size_t>(DMA_ ? DmaGeo<TD_, TH_>::kStageFloats : Geo<2, TH_>::kStageFloats);             \
    if (!adv_internal_lds_limit<conv3d_k3_mfma<TD_, MASKED_, TH_, DMA_>>(lds_)) return ADV_ELAUNCH;                                             \
    hipLaunchKernelGGL((conv3d_k3_mfma<TD_, MASKED_, TH_, DMA_>), grid, dim3(256), lds_, st, x, w_prep, y, cin, cout, cpad, d, h, w, tg, epi);  \
  } while (0)
    if (tsel == 44) {
      ADV_LAUNCH_MFMA(4, false, 4, true);
    } else if (!th4) {
      if (all_taps) { if (dma) ADV_LAUNCH_MFMA(2, false, 8, true); else ADV_LAUNCH_MFMA(2, false, 8, false); }
      else          { if (dma) ADV_LAUNCH_MFMA(2, true, 8, true); else ADV_LAUNCH_MFMA(2, true, 8, false); }
    } else {
      if (all_taps) { if (dma) ADV_LAUNCH_MFMA(2, false, 4, true); else ADV_LAUNCH_MFMA(2, false, 4, false); }
      else          { if (dma) ADV_LAUNCH_MFMA(2, true, 4, true); else ADV_LAUNCH_MFMA(2, true, 4, false); }
    }
#undef ADV_LAUNCH_MFMA
  } else if (stride == 1) {
    const dim3 grid(tiles_w * tiles_h, (gd + kTD - 1) / kTD, b * cblocks);
    hipLaunchKernelGGL((conv3d_k3_mfma_generic<1>), grid, dim3(256), static_cast<size_t>(GenGeo<1>::kSX + kSW) * sizeof(float), st, x, w_prep, y,
                       cin, cout, cblocks * 32, gd, gh, gw, d, h, w, tiles_w, cblocks, epi);
  } else {
    const dim3 grid(tiles_w * tiles_h, (gd + kTD - 1) / kTD, b * cblocks);
    const size_t lds = static_cast<size_t>(GenGeo<2>::kSX + kSW) * sizeof(float);
    // more than 64 KiB of dynamic LDS needs the attribute (raised once per kernel and device: adv_internal.h)
    if (!adv_internal_lds_limit<conv3d_k3_mfma_generic<2>>(lds)) return ADV_ELAUNCH;
    hipLaunchKernelGGL((conv3d_k3_mfma_generic<2>), grid, dim3(256), lds, st, x, w_prep, y, cin, cout, cblocks * 32, gd, gh, gw, d, h, w,
                       tiles_w, cblocks, epi);
  }
  return adv_internal_finish_launch();
}

int adv_conv3d_k3_f32(const float* x, const float* w_prep, float* y, int b, int cin, int cout, int d, int h, int w, int relu,
                      adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1) return ADV_EINVAL;
  if (cin % kCK != 0 && cin > kCK) return ADV_EINVAL;
  const Epi epi{nullptr, relu, kAllTaps, {0, 0, 0, 0, 0, 0, 0, 0}, 0, d, h, w, 1, 1, 1, 0, 0, 0};
  return launch_conv(x, w_prep, y, b, cin, cout, d, h, w, 1, epi, static_cast<hipStream_t>(stream));
}

int adv_conv3d_k3_masked_f32(const float* x, const float* w_prep, const float* mask, float* y, int b, int cin, int cout, int d, int h, int w,
                             adv_stream_t stream) {
  if (!x || !w_prep || !mask || !y || mask == y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1) return ADV_EINVAL;
  if (cin % kCK != 0 && cin > kCK) return ADV_EINVAL;
  if (reinterpret_cast<uintptr_t>(mask) & 3) return ADV_EALIGN;
  Epi epi{nullptr, 0, kAllTaps, {0, 0, 0, 0, 0, 0, 0, 0}, 0, d, h, w, 1, 1, 1, 0, 0, 0};
  epi.mask = mask;
  return launch_conv(x, w_prep, y, b, cin, cout, d, h, w, 1, epi, static_cast<hipStream_t>(stream));
}

int adv_conv3d_k3_s2_stage_channels(const float* x, int cout, int w) {
  Epi plain{nullptr, 0, kAllTaps, {0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0};
  return conv3d_k3_s2_direct(cout, w, x, plain) && w >= 4 ? kSC : kCK;
}

int adv_space_to_depth2_f32(const float* x, float* xs, int b, int c, int d, int h, int w, adv_stream_t stream) {
  if (!x || !xs || b < 1 || c < 1 || d < 1 || h < 1 || w < 1) return ADV_EINVAL;
  const int d2 = (d + 1) / 2, h2 = (h + 1) / 2, w2 = (w + 1) / 2;
  const long long total = static_cast<long long>(b) * 4 * c * d2 * h2 * w2;   // one lane per pair of output elements
  long long blocks = (total + 255) / 256;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(space_to_depth2, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), x, xs, b, c, d, h, w, d2,
                     h2, w2);
  return adv_internal_finish_launch();
}

int adv_conv3d_k3_ex_f32(const float* x, const float* w_prep, const float* bias, const float* residual, float* y, int b, int cin, int cout, int d, int h, int w,
                         int stride, int relu, uint32_t tap_mask, const uint32_t* class_masks, int class_channels, const int32_t* out_dims,
                         const int32_t* out_stride, const int32_t* out_offset, adv_stream_t stream) {
  if (!x || !w_prep || !y || b < 1 || cin < 1 || cout < 1 || d < 1 || h < 1 || w < 1) return ADV_EINVAL;
  if ((cin % kCK != 0 && cin > kCK) || (stride != 1 && stride != 2) || (tap_mask & ~kAllTaps)) return ADV_EINVAL;
  if ((out_dims == nullptr) != (out_stride == nullptr) || (out_dims == nullptr) != (out_offset == nullptr) || residual == y) return ADV_EINVAL;
  const int gd = stride == 2 ? (d + 1) / 2 : d, gh = stride == 2 ? (h + 1) / 2 : h, gw = stride == 2 ? (w + 1) / 2 : w;
  Epi epi{bias, relu, tap_mask, {0, 0, 0, 0, 0, 0, 0, 0}, 0, gd, gh, gw, 1, 1, 1, 0, 0, 0};
  epi.residual = residual;
  if (class_masks != nullptr) {
    if (class_channels < kCK || class_channels % kCK != 0 || cin != 8 * class_channels) return ADV_EINVAL;
    for (int k = 0; k < 8; ++k) {
      if (class_masks[k] & ~kAllTaps) return ADV_EINVAL;
      epi.class_masks[k] = class_masks[k];
    }
    epi.class_channels = class_channels;
  }
  if (out_dims) {
    for (int k = 0; k < 3; ++k)
      if (out_dims[k] < 1 || out_stride[k] < 1 || out_offset[k] < 0) return ADV_EINVAL;
    epi.od = out_dims[0], epi.oh = out_dims[1], epi.ow = out_dims[2];
    epi.sd = out_stride[0], epi.sh = out_stride[1], epi.sw = out_stride[2];
    epi.fd = out_offset[0], epi.fh = out_offset[1], epi.fw = out_offset[2];
  }
  return launch_conv(x, w_prep, y, b, cin, cout, d, h, w, stride, epi, static_cast<hipStream_t>(stream));
}

// mask != nullptr: only the all-classes kernel applies it (the other routes return ADV_EINVAL: the caller masks in a pass of its own)
static int convt3d_launch(const float* x, const float* const* w_prep_classes, const uint32_t* tap_masks, const float* bias,
                          const float* residual, const float* mask, float* y,
                          int b, int cin, int cout, int d, int h, int w, int relu, adv_stream_t stream) {
  if (!x || !w_prep_classes || !tap_masks || !y || residual == y || mask == y || b < 1 || cin < kCK || cin % kCK != 0 || cout < 1 || d < 1 || h < 1 || w < 1)
    return ADV_EINVAL;
  for (int k = 0; k < 8; ++k)
    if (w_prep_classes[k] == nullptr || (tap_masks[k] & ~kAllTaps) || tap_masks[k] == 0) return ADV_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Epi epi{bias, relu, tap_masks[0], {0, 0, 0, 0, 0, 0, 0, 0}, 0, 2 * d, 2 * h, 2 * w, 2, 2, 2, 0, 0, 0};
  epi.residual = residual;
  epi.mask = mask;
  const int cblocks = (cout + 31) / 32;
  const bool fits = w >= 4 && static_cast<long long>(cin) * d * h * w < (1LL << 31) && 27LL * cin * cblocks * 32 < (1LL << 31);
  bool aligned_w = (reinterpret_cast<uintptr_t>(x) & 3) == 0;
  for (int k = 0; k < 8; ++k) aligned_w = aligned_w && (reinterpret_cast<uintptr_t>(w_prep_classes[k]) & 15) == 0;
  if (fits && aligned_w && !adv_hook("ADV_CONV_GENERIC") && !adv_hook("ADV_CONV_CLASS_LAUNCHES")) {
    constexpr int tt = 4;   // rows (= waves) per tile; 8 rows / 512 threads / one workgroup per CU measured the same (0.200-0.204 against 0.205-0.207 ms)
    const int tiles_w = (w + kTW - 1) / kTW, tiles_h = (h + tt - 1) / tt;
    const long long ntiles = static_cast<long long>(tiles_w) * tiles_h * d * b * cblocks;
    // <round 4> flat volumes: 2 x 2 or 4 x 1 (planes x rows) tiles where they pad fewer rows than 1 x 4 (same bits)
    auto padded = [&](int tdp) { return static_cast<long long>((h + 4 / tdp - 1) / (4 / tdp)) * (4 / tdp) * ((d + tdp - 1) / tdp) * tdp; };
    int tdv = 1;
    if (padded(2) < padded(tdv)) tdv = 2;
    if (padded(4) < padded(tdv)) tdv = 4;
    if (const char* e = adv_hook_value("ADV_CONV_T_TD")) tdv = e[0] == '4' ? 4 : (e[0] == '2' ? 2 : 1);
    const bool dma = w % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && static_cast<long long>(cin) * d * h * w * 4 < 0xfff00000LL &&
                     !adv_hook("ADV_CONV_T_NO_DMA");     // else: register-staged (any width, any size)
    {   // the eight class tensors as one window of less than 2 GiB (ops.conv_transpose3d_k3_s2_prep allocates them as one slab)
      const float *lo = w_prep_classes[0], *hi = w_prep_classes[0];
      for (int k = 1; k < 8; ++k) lo = w_prep_classes[k] < lo ? w_prep_classes[k] : lo, hi = w_prep_classes[k] > hi ? w_prep_classes[k] : hi;
      const long long span = (hi - lo) * 4 + 27LL * cin * cblocks * 32 * 4;
      epi.cls_lo = span < 0x7ff00000LL && !adv_hook("ADV_CONV_T_POINTER_WEIGHTS") ? lo : nullptr;
      epi.cls_span = epi.cls_lo ? static_cast<unsigned>(span) : 0u;
    }
    const bool all_classes = (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(residual) & 7) == 0 &&
                             (reinterpret_cast<uintptr_t>(mask) & 7) == 0 && ntiles < (1LL << 31) && !adv_hook("ADV_CONV_T_CLASS_TILES");
    if (mask != nullptr && !all_classes) return ADV_EINVAL;
    if (all_classes) {   // every class from one staging of the input tile (convt3d_k3_s2_mfma)
      for (int k = 0; k < 8; ++k) epi.cls_wp[k] = w_prep_classes[k];
#define ADV_LAUNCH_T2(TT_, TD_, N_, TH_)                                                                                                     \
  do {                                                                                                                                        \
    const size_t lds_ = 2 * sizeof(float) * static_cast<size_t>(TGeo<TT_, TD_>::kStage);                                                      \
    if (dma)                                                                                                                                  \
      hipLaunchKernelGGL((convt3d_k3_s2_mfma<TT_, TD_, true>), dim3(static_cast<unsigned>(N_)), dim3(256), lds_, st, x, y, cin, cout,         \
                         cblocks * 32, d, h, w, tiles_w, TH_, cblocks, epi);                                                                  \
    else                                                                                                                                      \
      hipLaunchKernelGGL((convt3d_k3_s2_mfma<TT_, TD_, false>), dim3(static_cast<unsigned>(N_)), dim3(256), lds_, st, x, y, cin, cout,        \
                         cblocks * 32, d, h, w, tiles_w, TH_, cblocks, epi);                                                                  \
    return adv_internal_finish_launch();                                                                                                      \
  } while (0)
      if (tdv > 1) {
        const int th2 = (h + 4 / tdv - 1) / (4 / tdv);
        const long long nt = static_cast<long long>(tiles_w) * th2 * ((d + tdv - 1) / tdv) * b * cblocks;
        if (nt < (1LL << 31)) {
          if (tdv == 2) ADV_LAUNCH_T2(2, 2, nt, th2);
          ADV_LAUNCH_T2(1, 4, nt, th2);
        }
      }
      ADV_LAUNCH_T2(4, 1, ntiles, tiles_h);
#undef ADV_LAUNCH_T2
    }
    if (mask != nullptr) return ADV_EINVAL;
    epi.nclass = 8;   // one launch: the class is a tile index (launch_conv's persistent masked kernel)
    for (int k = 0; k < 8; ++k) {
      epi.cls_wp[k] = w_prep_classes[k];
      epi.cls_mask[k] = tap_masks[k];
      epi.cls_off[k][0] = (k >> 2) & 1, epi.cls_off[k][1] = (k >> 1) & 1, epi.cls_off[k][2] = k & 1;
    }
    epi.tap_mask = 0;  // never the all-taps kernel
    return launch_conv(x, w_prep_classes[0], y, b, cin, cout, d, h, w, 1, epi, st);
  }
  if (mask != nullptr) return ADV_EINVAL;
  for (int k = 0; k < 8; ++k) {  // eight masked launches: the same bits (ADV_CONV_CLASS_LAUNCHES=1: test hook / A-B)
    epi.tap_mask = tap_masks[k];
    epi.fd = (k >> 2) & 1, epi.fh = (k >> 1) & 1, epi.fw = k & 1;
    const int rc = launch_conv(x, w_prep_classes[k], y, b, cin, cout, d, h, w, 1, epi, st);
    if (rc != ADV_OK) return rc;
  }
  return ADV_OK;
}

int adv_conv_transpose3d_k3_s2_f32(const float* x, const float* const* w_prep_classes, const uint32_t* tap_masks, const float* bias,
                                   const float* residual, float* y,
                                   int b, int cin, int cout, int d, int h, int w, int relu, adv_stream_t stream) {
  return convt3d_launch(x, w_prep_classes, tap_masks, bias, residual, nullptr, y, b, cin, cout, d, h, w, relu, stream);
}

int adv_conv_transpose3d_k3_s2_dgrad_f32(const float* x, const float* const* w_prep_classes, const uint32_t* tap_masks, const float* residual,
                                         const float* mask, float* y, int b, int cin, int cout, int d, int h, int w, adv_stream_t stream) {
  if (mask == nullptr) return ADV_EINVAL;
  return convt3d_launch(x, w_prep_classes, tap_masks, nullptr, residual, mask, y, b, cin, cout, d, h, w, 0, stream);
}

}  // extern "C"
