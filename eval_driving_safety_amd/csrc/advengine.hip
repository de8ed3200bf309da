// libadvengine.so - perturbation inner loop of the stereo-detector attacks, for gfx950 (MI355X).
//
// Every kernel here is HBM-bound float32 streaming work; none of it is GEMM-shaped, so no MFMA.
// What matters (cdna_hip_programming.md G2/G11/G13, Appendix B "Element-wise"): 16-byte
// accesses per lane, fully coalesced 1 KiB wave transactions, enough independent loads in
// flight per lane, and one pass over memory per PGD step instead of the reference's ~12.
//
// Arithmetic contract: bit-identical to torch-CPU float32 (see include/advengine.h).  This file
// is compiled with -ffp-contract=off and the pragma below; HIP's default float division is the
// correctly rounded one (-fhip-fp32-correctly-rounded-divide-sqrt) and f32 denormals are kept.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "adv_internal.h"
#include "advengine.h"

#pragma clang fp contract(off)

namespace {

thread_local int g_last_hip_error = 0;

constexpr int kBlock = 256;  // small / irregular kernels: 4 waves of 64, one per SIMD of a CU

// Streaming kernels (measured on MI355X with tools/k1_tune.hip, 512 KITTI-shaped images resident,
// profiles/r01_k1_tuning.md): ONE wave per workgroup, every lane owning kUnroll pixel groups half an
// image apart, one workgroup per such pair of tiles (no capped grid, no multi-trip loop), and
// non-temporal loads AND stores (nothing is re-read before the next detector pass) ran at
// 6.0-6.3 TB/s algorithmic against 4.65 TB/s for 256-thread blocks on a grid capped at 4096.
constexpr int kWave = 64;
constexpr int kUnroll = 2;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));

__device__ __forceinline__ v4f ld_stream(const v4f* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(v4f* p, v4f v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(v3u* p, v3u v) { __builtin_nontemporal_store(v, p); }

struct SpaceK {  // by-value kernel argument (lives in the kernarg segment -> scalar loads)
  float scale[3];
  float shift[3];
  float lo[3];
  float hi[3];
  double export_add[3];
  float rcp[3];  // 1.0f / scale, float32
  int use_rcp;   // ADV_SPACE_AFFINE_RCP: re-normalise with (y - shift) * rcp, as torch ON A GPU evaluates tensor / python_scalar
};

// ------------------------------------------------------------------------------------------
// scalar building blocks, each the float32 operation torch performs
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float t_sign(float g) {  // torch.sign: (g > 0) - (g < 0)
  return (g > 0.0f ? 1.0f : 0.0f) - (g < 0.0f ? 1.0f : 0.0f);
}

__device__ __forceinline__ float t_clamp(float x, float lo, float hi) {  // NaN stays NaN
  return x < lo ? lo : (x > hi ? hi : x);
}

// (y - shift) / scale: a true IEEE division (torch-CPU), or - uniform branch on a kernel argument - the multiplication by the
// float32 reciprocal that torch's CUDA/ROCm kernels substitute when the divisor is a Python scalar
__device__ __forceinline__ float renorm(float y, float sc, float sh, float rcp, int use_rcp) {
  const float t = y - sh;
  return use_rcp ? t * rcp : t / sc;
}

template <int KIND>
__device__ __forceinline__ float pgd_elem(float x, float g, float cl, float sc, float sh, float lo,
                                          float hi, float alpha, float eps, float rcp = 0.0f, int use_rcp = 0) {
  float d = x;
  if (KIND == ADV_SPACE_AFFINE) {
    d = x * sc;
    d = d + sh;
  }
  const float a = d + alpha * t_sign(g);
  const float eta = t_clamp(a - cl, -eps, eps);
  const float y = t_clamp(cl + eta, lo, hi);
  if (KIND == ADV_SPACE_AFFINE) return renorm(y, sc, sh, rcp, use_rcp);
  return y;
}

// numpy's float32 -> uint8 astype on x86-64 is cvttss2si + low byte: NaN and |v| >= 2^31 give
// 0x80000000 whose low byte is 0.
__device__ __forceinline__ uint32_t trunc_low_byte(float v) {
  if (!(fabsf(v) < 2147483648.0f)) return 0u;
  return static_cast<uint32_t>(static_cast<int>(v)) & 0xffu;
}

// OpenCV saturate_cast<uchar>(float): cvRound (round half to even, 0x80000000 when not
// representable) then clip to [0,255].
__device__ __forceinline__ uint32_t round_sat_byte(float v) {
  if (!(fabsf(v) < 2147483648.0f)) return 0u;
  const int i = static_cast<int>(rintf(v));
  return i < 0 ? 0u : (i > 255 ? 255u : static_cast<uint32_t>(i));
}

template <int KIND>
__device__ __forceinline__ uint32_t export_byte(float xo, float sc, float sh, double add) {
  if (KIND == ADV_SPACE_AFFINE) {
    float v = xo * sc;
    v = v + sh;
    v = v * 255.0f;
    return trunc_low_byte(v);
  }
  const float f = static_cast<float>(static_cast<double>(xo) + add);
  return round_sat_byte(f);
}

// U8_ROWS_DWORD: w % 4 == 0 and whole rows -> a pixel group never leaves its row, one aligned 12-byte store;
// U8_FLAT_DWORD: dense uncropped image (pitch 3*w, all h rows; Stereo R-CNN's 600 x 1987, which is never cropped) ->
//                the HWC bytes of pixel group q are the 12 bytes at offset 12*q whatever w is;
// U8_BYTES: anything else, byte stores.
enum U8Mode { U8_NONE = 0, U8_ROWS_DWORD = 1, U8_BYTES = 2, U8_FLAT_DWORD = 3 };

struct U8Dst {
  uint8_t* base;
  long long row_stride;
  long long image_stride;
  int crop_h;
  int ncols;  // columns stored per row: w when the pitch holds whole rows, else crop_w
};

// 4 pixels x 3 channels -> 12 interleaved bytes (HWC) in three dwords
template <int KIND>
__device__ __forceinline__ v3u pack_hwc4(const v4f o[3], const SpaceK& sp) {
  uint32_t b[12];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int c = 0; c < 3; ++c) b[j * 3 + c] = export_byte<KIND>(o[c][j], sp.scale[c], sp.shift[c], sp.export_add[c]);
  }
  v3u r;
  r[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
  r[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
  r[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
  return r;
}

// the export bytes of one pixel group q of image img (shared by the fused and stand-alone kernels)
template <int KIND, int U8>
__device__ __forceinline__ void store_u8_group(const v4f O[3], long long img, int q, int w, const SpaceK& sp, const U8Dst& u8) {
  if (U8 == U8_ROWS_DWORD) {  // w % 4 == 0, whole rows, every address 4-byte aligned
    const int p = q * 4;
    const int row = p / w;
    if (row < u8.crop_h) {
      const int col = p - row * w;
      st_stream(reinterpret_cast<v3u*>(u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL), pack_hwc4<KIND>(O, sp));
    }
  } else if (U8 == U8_FLAT_DWORD) {
    st_stream(reinterpret_cast<v3u*>(u8.base + img * u8.image_stride + 12LL * q), pack_hwc4<KIND>(O, sp));
  } else if (U8 == U8_BYTES) {  // any pitch / crop: per-pixel byte stores
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = q * 4 + j;
      const int row = p / w;
      const int col = p - row * w;
      if (row < u8.crop_h && col < u8.ncols) {
        uint8_t* dst = u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          dst[c] = static_cast<uint8_t>(export_byte<KIND>(O[c][j], sp.scale[c], sp.shift[c], sp.export_add[c]));
      }
    }
  }
}

// the same for values that are ALREADY in [0,1] pixel space (the denormalised clean image): byte = trunc(v * 255),
// which is what tensor2im computes from the normalised image (attack/DSGN/pgd_attack.py:174-178)
template <int U8>
__device__ __forceinline__ void store_u8_pixel_group(const v4f O[3], long long img, int q, int w, const U8Dst& u8) {
  if (U8 == U8_NONE) return;
  const int p = q * 4;
  const int row = p / w;
  if (row >= u8.crop_h) return;
  const int col = p - row * w;
  uint8_t* dst = u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL;
  uint32_t b[12];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c) b[j * 3 + c] = trunc_low_byte(O[c][j] * 255.0f);
  if (U8 == U8_ROWS_DWORD) {  // w % 4 == 0 (required by the index build): the group never leaves its row
    v3u r;
    r[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    r[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    r[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    st_stream(reinterpret_cast<v3u*>(dst), r);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col + j < u8.ncols) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[j * 3 + c] = static_cast<uint8_t>(b[j * 3 + c]);
      }
  }
}

// ------------------------------------------------------------------------------------------
// K1/K2 (+K5 fused): one PGD step.  A lane owns 4 consecutive pixels of one image in all three
// channel planes (kUnroll such groups, half an image apart): 9 independent 16-byte loads per
// group, three 16-byte stores, and (optionally) the 12 interleaved export bytes, which for
// consecutive lanes are consecutive in memory.  grid.y = images, grid.x = tiles of 64 groups.
// x_out may alias x: every lane reads its own elements before it writes them.
// ------------------------------------------------------------------------------------------
template <int KIND, int U8>
__global__ __launch_bounds__(kWave) void pgd_step_vec4(const v4f* x, const v4f* __restrict__ g, const v4f* __restrict__ cl,
                                                       v4f* xo, long long n_img, int hw4, int w, SpaceK sp, float alpha,
                                                       float eps, U8Dst u8) {
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    for (int q0 = blockIdx.x * kWave + threadIdx.x; q0 < hw4; q0 += stride * kUnroll) {
      v4f X[kUnroll][3], G[kUnroll][3], C[kUnroll][3];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
            X[u][c] = ld_stream(x + i);
            G[u][c] = ld_stream(g + i);
            C[u][c] = ld_stream(cl + i);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
          v4f O[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              O[c][j] = pgd_elem<KIND>(X[u][c][j], G[u][c][j], C[u][c][j], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps, sp.rcp[c], sp.use_rcp);
            st_stream(xo + plane0 + static_cast<long long>(c) * hw4 + q, O[c]);
          }
          store_u8_group<KIND, U8>(O, img, q, w, sp, u8);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1 with the clean image held as an 8-bit INDEX (AFFINE spaces).  The clean image of an attack is re-read by
// every one of its N steps, and it is not an arbitrary float image: it came from 8-bit pixels v through
//     t = v / 255;  x0 = (t - shift) / scale;  clean = x0 * scale + shift            (all float32)
// i.e. ToTensor, Normalize, then the script's denormalize (attack/DSGN/pgd_attack.py:196-200,297-298), and the
// loader zero-pads the normalised image to the network size, so that outside the valid_h x valid_w corner the
// clean value is exactly shift_c (0 * scale + shift).  clean_index_build VERIFIES both facts per image, element
// by element and bit by bit, against a 3 x 256 table T_c[v] of that chain; the step kernel then reads the
// 1-byte index and looks the float up in an LDS copy of the table (13 instead of 16 bytes per element and
// launch, no division, bit-identical results).  An image with a single failing element (resized or otherwise
// processed inputs) keeps ok[img] == 0 and the same launch reads its float32 clean buffer instead - the flags
// are read on the device (one scalar load per workgroup), no host round trip, no effect on the other images.
// Measured on 512 KITTI-shaped images (tools/k1_idx_tune.hip, profiles/r02_k1_idx_tuning.md): table in LDS
// 1.573 ms, table gathered from global memory 1.600 ms, round 1's two IEEE divisions 1.592 ms; 4-wave
// workgroups or more than one trip per workgroup are 3-15 % slower.
// ------------------------------------------------------------------------------------------
template <int DIR>
__device__ __forceinline__ float affine_elem(float x, float sc, float sh, float rcp = 0.0f, int use_rcp = 0);

__device__ __forceinline__ float clean_from_index(uint32_t v, float sc, float sh) {
  float t = static_cast<float>(v) / 255.0f;
  t = (t - sh) / sc;
  t = t * sc;
  return t + sh;
}

constexpr int kLutSize = 3 * 256;

struct IdxK {  // by-value kernel argument: the device side of adv_clean_index_t
  uint32_t* idx;         // [n,3,hw4] words of 4 index bytes
  int* ok;               // [n]: bit 0 = table A reproduces the image, bit 1 = table B does (identity spaces only)
  float* lut;            // [2][3*256]: table A, table B
  const int* valid_hw;   // [n,2] or nullptr
  int vh, vw;
};

// AFFINE: one table (B repeats A).  IDENTITY (pixel value minus a per-channel mean, Stereo R-CNN): the loader's subtraction may
// have run in float32 (x = float(v) - float(mean)) or in float64 rounded once (numpy's `im -= pixel_means` with float64 means:
// x = float(double(v) - mean)) - the two differ in the last bit for many v, so both tables are kept and every image is
// verified against each; export_add[c] carries the mean as a double.
__global__ __launch_bounds__(256) void clean_lut_kernel(float* lut, SpaceK sp, int identity) {
  for (int e = threadIdx.x; e < kLutSize; e += 256) {
    const int c = e >> 8, v = e & 255;
    float a, b;
    if (identity) {
      a = static_cast<float>(v) - static_cast<float>(sp.export_add[c]);
      b = static_cast<float>(static_cast<double>(v) - sp.export_add[c]);
    } else {
      a = b = clean_from_index(v, sp.scale[c], sp.shift[c]);
    }
    lut[e] = a;
    lut[kLutSize + e] = b;
  }
}

__device__ __forceinline__ void stage_lut(float* lds, const float* __restrict__ lut_g) {
  for (int k = threadIdx.x; k < kLutSize / 4; k += kWave) reinterpret_cast<v4f*>(lds)[k] = reinterpret_cast<const v4f*>(lut_g)[k];
  __syncthreads();
}

template <int U8>
__global__ __launch_bounds__(kWave) void pgd_step_vec4_idx(const v4f* x, const v4f* __restrict__ g, const v4f* __restrict__ cl, IdxK ik,
                                                           v4f* xo, long long n_img, int hw4, int w, SpaceK sp, float alpha, float eps,
                                                           U8Dst u8) {
  __shared__ float lut[kLutSize];
  stage_lut(lut, ik.lut);
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const bool use_idx = ik.ok[img] != 0;  // uniform: scalar loads
    const int vh = ik.valid_hw ? ik.valid_hw[2 * img] : ik.vh;
    const int vw = ik.valid_hw ? ik.valid_hw[2 * img + 1] : ik.vw;
    const long long plane0 = img * 3LL * hw4;
    for (int q0 = blockIdx.x * kWave + threadIdx.x; q0 < hw4; q0 += stride * kUnroll) {
      v4f X[kUnroll][3], G[kUnroll][3], C[kUnroll][3];
      uint32_t I[kUnroll][3];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
            X[u][c] = ld_stream(x + i);
            G[u][c] = ld_stream(g + i);
            if (use_idx)
              I[u][c] = __builtin_nontemporal_load(ik.idx + i);
            else
              C[u][c] = ld_stream(cl + i);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
          const int p = q * 4;  // w % 4 == 0: the 4 pixels of a group share their row
          const int row = p / w;
          const int col = p - row * w;
          v4f O[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if (use_idx) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float t = lut[c * 256 + ((I[u][c] >> (8 * j)) & 0xffu)];
                C[u][c][j] = (row < vh && col + j < vw) ? t : sp.shift[c];
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
              O[c][j] = pgd_elem<ADV_SPACE_AFFINE>(X[u][c][j], G[u][c][j], C[u][c][j], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps, sp.rcp[c], sp.use_rcp);
            st_stream(xo + plane0 + static_cast<long long>(c) * hw4 + q, O[c]);
          }
          store_u8_group<ADV_SPACE_AFFINE, U8>(O, img, q, w, sp, u8);
        }
      }
    }
  }
}

// a1 fused with the index build, its verification and (optionally) the 8-bit export of iterate 0:
// clean = x*scale+shift; inside the valid corner v = rint(clean*255) clamped to 0..255 and T_c[v] must equal clean
// bit for bit, outside it clean must equal shift_c bit for bit; ok[img] is cleared otherwise (the host side sets
// it to 1 first).  The export byte of iterate 0 is trunc(clean*255): tensor2im applies the same x*scale+shift.
template <int U8>
__global__ __launch_bounds__(kWave) void clean_index_build_vec4(const v4f* x, v4f* clean, IdxK ik, long long n_img, int hw4, int w, SpaceK sp,
                                                                U8Dst u8) {
  __shared__ float lut[kLutSize];
  stage_lut(lut, ik.lut);
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const int vh = ik.valid_hw ? ik.valid_hw[2 * img] : ik.vh;
    const int vw = ik.valid_hw ? ik.valid_hw[2 * img + 1] : ik.vw;
    const long long plane0 = img * 3LL * hw4;
    bool bad = false;
    for (int q = blockIdx.x * kWave + threadIdx.x; q < hw4; q += stride) {
      const int p = q * 4;
      const int row = p / w;
      const int col = p - row * w;
      v4f O[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
        const v4f X = ld_stream(x + i);
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          O[c][j] = affine_elem<0>(X[j], sp.scale[c], sp.shift[c]);
          float r = rintf(O[c][j] * 255.0f);
          r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);  // NaN compares false twice and converts to 0 below
          const uint32_t v = (r == r) ? static_cast<uint32_t>(r) : 0u;
          const bool inside = row < vh && col + j < vw;
          const float want = inside ? lut[c * 256 + v] : sp.shift[c];
          bad |= __float_as_uint(want) != __float_as_uint(O[c][j]);
          word |= (inside ? v : 0u) << (8 * j);
        }
        st_stream(clean + i, O[c]);
        __builtin_nontemporal_store(word, ik.idx + i);
      }
      store_u8_pixel_group<U8>(O, img, q, w, u8);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) ik.ok[img] = 0;
  }
}

// The loader's work on the device (the inverse of the 8-bit export): 8-bit RGB pixels, HWC, -> what the DSGN loader hands over,
// x = ((v/255) - shift)/scale zero-padded in normalised space to the network frame (ToTensor, Normalize, pad: true float32
// divisions, as torch computes them on the CPU), and with it - for free, by construction instead of by verification - the attack's
// clean image clean = x*scale + shift and its 8-bit index (the pixels themselves).  One quarter of the PCIe bytes of a float
// upload and no conversion on the host.  A lane owns 4 pixels: 12 source bytes, three float4 per output array.
template <bool ALIGNED>
__global__ __launch_bounds__(kWave) void import_u8_vec4(const uint8_t* __restrict__ u8, long long row_stride, long long image_stride, v4f* x,
                                                        v4f* clean, IdxK ik, long long n_img, int hw4, int w, SpaceK sp) {
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const int vh = ik.valid_hw ? ik.valid_hw[2 * img] : ik.vh;
    const int vw = ik.valid_hw ? ik.valid_hw[2 * img + 1] : ik.vw;
    const long long plane0 = img * 3LL * hw4;
    for (int q = blockIdx.x * kWave + threadIdx.x; q < hw4; q += stride) {
      const int p = q * 4;  // w % 4 == 0: the 4 pixels of a group share their row
      const int row = p / w;
      const int col = p - row * w;
      uint32_t b[3] = {0u, 0u, 0u};  // 12 bytes: p0c0 p0c1 p0c2 p1c0 | p1c1 p1c2 p2c0 p2c1 | p2c2 p3c0 p3c1 p3c2
      if (row < vh && col < vw) {
        const uint8_t* src = u8 + img * image_stride + row * row_stride + col * 3LL;
        if (ALIGNED && col + 3 < vw) {
          const v3u t = *reinterpret_cast<const v3u*>(src);
          b[0] = t[0], b[1] = t[1], b[2] = t[2];
        } else {
          const int nb = 3 * ((vw - col) < 4 ? (vw - col) : 4);
          for (int k = 0; k < nb; ++k) b[k >> 2] |= static_cast<uint32_t>(src[k]) << (8 * (k & 3));
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v4f X, C;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 3 * j + c;
          const uint32_t v = (b[k >> 2] >> (8 * (k & 3))) & 0xffu;
          const bool inside = row < vh && col + j < vw;
          float t = static_cast<float>(v) / 255.0f;
          t = (t - sp.shift[c]) / sp.scale[c];
          X[j] = inside ? t : 0.0f;
          C[j] = inside ? t * sp.scale[c] + sp.shift[c] : sp.shift[c];
          word |= (inside ? v : 0u) << (8 * j);
        }
        const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
        st_stream(x + i, X);
        if (clean != nullptr) st_stream(clean + i, C);
        if (ik.idx != nullptr) __builtin_nontemporal_store(word, ik.idx + i);
      }
    }
  }
}

// The index of an IDENTITY-space batch (Stereo R-CNN: x = 8-bit pixel - mean_c, attack/Stereo-RCNN/pgd_attack.py:122-123 clones it
// as the clean pair): v = rint(x + mean_c), verified against both tables; no padding rule (the whole frame is image), any row
// length (hw % 4 == 0 is all the float4 path needs).  clean_out = x when the caller wants a separate copy.
__global__ __launch_bounds__(kWave) void clean_index_build_identity(const v4f* x, v4f* clean, IdxK ik, long long n_img, int hw4, SpaceK sp,
                                                                    int copy) {
  __shared__ float lut[2 * kLutSize];
  for (int k = threadIdx.x; k < 2 * kLutSize / 4; k += kWave) reinterpret_cast<v4f*>(lut)[k] = reinterpret_cast<const v4f*>(ik.lut)[k];
  __syncthreads();
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    bool bad_a = false, bad_b = false;
    for (int q = blockIdx.x * kWave + threadIdx.x; q < hw4; q += stride) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
        const v4f X = ld_stream(x + i);
        const float mean = static_cast<float>(sp.export_add[c]);
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float r = rintf(X[j] + mean);
          r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
          const uint32_t v = (r == r) ? static_cast<uint32_t>(r) : 0u;
          bad_a |= __float_as_uint(lut[c * 256 + v]) != __float_as_uint(X[j]);
          bad_b |= __float_as_uint(lut[kLutSize + c * 256 + v]) != __float_as_uint(X[j]);
          word |= v << (8 * j);
        }
        if (copy) st_stream(clean + i, X);
        __builtin_nontemporal_store(word, ik.idx + i);
      }
    }
    const int clear = (__any(bad_a) ? 1 : 0) | (__any(bad_b) ? 2 : 0);
    if (clear != 0 && (threadIdx.x & 63) == 0) atomicAnd(ik.ok + img, ~clear);
  }
}

// ------------------------------------------------------------------------------------------
// K1/K2 for planes that are NOT a whole number of 128-byte cache lines (Stereo R-CNN: 600 x 1987 floats
// = 37 256.25 lines).  There the three channel planes of an image start at different offsets within a
// line, so no common pixel tiling is line-aligned in all of them: the plain kernel's 1 KiB wave accesses
// straddle 9 lines instead of 8 (PMC: 1.094 x the algorithmic read bytes, 0.65 of peak instead of 0.76).
// Here every channel gets its OWN tile origin, shifted by its misalignment m_c (in float4, 0..7), so that
// every wave access of every plane is line-aligned: workgroup `tile` owns pixel groups
// [tile - m_c, tile - m_c + 256) of channel c.  The fused export needs the three channels of the SAME pixels:
// each lane packs its 4 pixels of a channel into one 32-bit word, the words are exchanged through LDS, and the
// groups all three channels of this workgroup cover - [tile - m_min, tile - m_max + 256), all but <= 7 - leave as
// aligned 12-byte stores.  The few groups at the two ends, whose channels are split between this workgroup and a
// neighbour, are written by BOTH as single bytes, each storing the channels it owns (distinct bytes, no race).
// No lane ever reads an element it does not own, so x_out may alias x (in place) with the export on - round 1's
// version re-read neighbours' x in "halo" lanes and could not.
// Workgroup = ONE wave = 64 pixel groups per channel, grid = (tiles, images): measured on 128 images of 600 x 1987 with the
// export fused and the update in place (tools/sb_sweep.sh, profiles/r02_srcnn_shape_sweep.log): 64 lanes x 1 sub-tile
// 1.267 ms = 0.768 of peak; 64 x 2: 0.732; 128 x 1: 0.729; 256 x 1 (round 1): 0.724; 256 x 2: 0.704; 512 x 1: 0.691.
// ------------------------------------------------------------------------------------------
#ifndef ADV_SHIFT_BLOCK
#define ADV_SHIFT_BLOCK 64
#endif
#ifndef ADV_SHIFT_UNROLL
#define ADV_SHIFT_UNROLL 1
#endif
constexpr int kShiftBlock = ADV_SHIFT_BLOCK;      // lanes per workgroup
constexpr int kShiftUnroll = ADV_SHIFT_UNROLL;    // consecutive sub-tiles of kShiftBlock groups per workgroup
constexpr int kShiftTile = kShiftBlock * kShiftUnroll;

template <int KIND>
__device__ __forceinline__ uint32_t pack_channel4(const v4f& o, float sc, float sh, double add) {
  return export_byte<KIND>(o[0], sc, sh, add) | (export_byte<KIND>(o[1], sc, sh, add) << 8) |
         (export_byte<KIND>(o[2], sc, sh, add) << 16) | (export_byte<KIND>(o[3], sc, sh, add) << 24);
}

// the 4 bytes of channel c of pixel group q, one by one (any layout)
template <int U8>
__device__ __forceinline__ void store_channel_bytes(uint32_t word, int c, long long img, int q, int w, const U8Dst& u8) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = q * 4 + k;
    uint8_t* dst;
    if (U8 == U8_FLAT_DWORD) {
      dst = u8.base + img * u8.image_stride + 3LL * p + c;
    } else {
      const int row = p / w;
      const int col = p - row * w;
      if (row >= u8.crop_h || col >= u8.ncols) continue;
      dst = u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL + c;
    }
    *dst = static_cast<uint8_t>((word >> (8 * k)) & 0xffu);
  }
}

// IDX: images whose clean-image index was verified (ik.ok[img] != 0) read one index word per pixel group and look the four clean
// values up in an LDS copy of the table instead of loading a float4 of `cl` (13 instead of 16 bytes per element); the table is
// staged once per workgroup, which walks two images (grid.y = n/2; measured optimum, see step_indexed_identity).
template <int KIND, int U8, bool IDX = false>
__global__ __launch_bounds__(kShiftBlock) void pgd_step_shifted(const v4f* x, const v4f* __restrict__ g,
                                                                const v4f* __restrict__ cl, v4f* xo, long long n_img, int hw4,
                                                                int w, int base_f4, SpaceK sp, float alpha, float eps, U8Dst u8,
                                                                IdxK ik = IdxK{nullptr, nullptr, nullptr, nullptr, 0, 0}) {
  constexpr bool kExchange = (U8 == U8_ROWS_DWORD || U8 == U8_FLAT_DWORD);
  __shared__ uint32_t words[kExchange ? 3 : 1][kExchange ? kShiftTile : 1];
  __shared__ float lut[IDX ? 2 * kLutSize : 4];
  const int j = threadIdx.x;
  if (IDX) {
    for (int k = j; k < 2 * kLutSize / 4; k += kShiftBlock) reinterpret_cast<v4f*>(lut)[k] = reinterpret_cast<const v4f*>(ik.lut)[k];
    __syncthreads();
  }
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    const int sel = IDX ? ik.ok[img] : 0;                                   // uniform: scalar load
    const float* tb = lut + ((sel & 1) ? 0 : kLutSize);
    int m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = static_cast<int>((base_f4 + plane0 + static_cast<long long>(c) * hw4) & 7);
    const int m_min = min(m[0], min(m[1], m[2])), m_max = max(m[0], max(m[1], m[2]));
    const int tile = blockIdx.x * kShiftTile;
    const int common_lo = tile - m_min, common_hi = min(tile - m_max + kShiftTile, hw4);  // groups all 3 channels cover here
    // all loads first (3 channels x kShiftUnroll groups x 3 arrays in flight per lane), then the arithmetic
    v4f X[kShiftUnroll][3], G[kShiftUnroll][3], C[kShiftUnroll][3];
    uint32_t I[kShiftUnroll][3];
#pragma unroll
    for (int u = 0; u < kShiftUnroll; ++u) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int q = tile - m[c] + u * kShiftBlock + j;  // this lane's pixel group in channel c: line-aligned per wave
        if (q >= 0 && q < hw4) {
          const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
          X[u][c] = ld_stream(x + i);
          G[u][c] = ld_stream(g + i);
          if (IDX && sel != 0)
            I[u][c] = __builtin_nontemporal_load(ik.idx + i);
          else
            C[u][c] = ld_stream(cl + i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kShiftUnroll; ++u) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int q = tile - m[c] + u * kShiftBlock + j;
        if (q >= 0 && q < hw4) {
          const long long i = plane0 + static_cast<long long>(c) * hw4 + q;
          v4f O;
          if (IDX && sel != 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) C[u][c][k] = tb[c * 256 + ((I[u][c] >> (8 * k)) & 0xffu)];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            O[k] = pgd_elem<KIND>(X[u][c][k], G[u][c][k], C[u][c][k], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps, sp.rcp[c], sp.use_rcp);
          st_stream(xo + i, O);
          if (U8 != U8_NONE) {
            const uint32_t word = pack_channel4<KIND>(O, sp.scale[c], sp.shift[c], sp.export_add[c]);
            if (kExchange) {
              words[c][u * kShiftBlock + j] = word;
              if (q < common_lo || q >= common_hi) store_channel_bytes<U8>(word, c, img, q, w, u8);  // <= 7 groups per end
            } else {
              store_channel_bytes<U8>(word, c, img, q, w, u8);
            }
          }
        }
      }
    }
    if (kExchange) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kShiftUnroll; ++u) {
        const int jj = u * kShiftBlock + j;
        const int q0 = common_lo + jj;
        if (q0 >= 0 && q0 < common_hi) {
          const uint32_t w0 = words[0][jj + m[0] - m_min];
          const uint32_t w1 = words[1][jj + m[1] - m_min];
          const uint32_t w2 = words[2][jj + m[2] - m_min];
          v3u r;  // bytes: p0c0 p0c1 p0c2 p1c0 | p1c1 p1c2 p2c0 p2c1 | p2c2 p3c0 p3c1 p3c2
          r[0] = (w0 & 0xffu) | ((w1 & 0xffu) << 8) | ((w2 & 0xffu) << 16) | (((w0 >> 8) & 0xffu) << 24);
          r[1] = ((w1 >> 8) & 0xffu) | (((w2 >> 8) & 0xffu) << 8) | (((w0 >> 16) & 0xffu) << 16) | (((w1 >> 16) & 0xffu) << 24);
          r[2] = ((w2 >> 16) & 0xffu) | (((w0 >> 24) & 0xffu) << 8) | (((w1 >> 24) & 0xffu) << 16) | (((w2 >> 24) & 0xffu) << 24);
          if (U8 == U8_FLAT_DWORD) {
            st_stream(reinterpret_cast<v3u*>(u8.base + img * u8.image_stride + 12LL * q0), r);
          } else {
            const int p = q0 * 4;
            const int row = p / w;
            if (row < u8.crop_h) {
              const int col = p - row * w;
              st_stream(reinterpret_cast<v3u*>(u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL), r);
            }
          }
        }
      }
      __syncthreads();  // words[] is reused by the next image of this workgroup
    }
  }
}

// general shapes (HW % 4 != 0 or pointers not 16-byte aligned): one pixel (3 channels) per lane
template <int KIND>
__global__ __launch_bounds__(kBlock) void pgd_step_scalar(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ cl, float* xo, long long n_img,
                                                          int hw, int w, SpaceK sp, float alpha, float eps, U8Dst u8) {
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < hw; p += gridDim.x * kBlock) {
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long i = plane0 + static_cast<long long>(c) * hw + p;
        o[c] = pgd_elem<KIND>(x[i], g[i], cl[i], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps, sp.rcp[c], sp.use_rcp);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) xo[plane0 + static_cast<long long>(c) * hw + p] = o[c];
      if (u8.base != nullptr) {
        const int row = p / w;
        const int col = p - row * w;
        if (row < u8.crop_h && col < u8.ncols) {
          uint8_t* dst = u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL;
#pragma unroll
          for (int c = 0; c < 3; ++c)
            dst[c] = static_cast<uint8_t>(export_byte<KIND>(o[c], sp.scale[c], sp.shift[c], sp.export_add[c]));
        }
      }
    }
  }
}

// K5 alone: read the image once, write the HWC bytes
template <int KIND, int U8>
__global__ __launch_bounds__(kWave) void export_u8_vec4(const v4f* __restrict__ x, long long n_img, int hw4, int w, SpaceK sp,
                                                        U8Dst u8) {
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    for (int q = blockIdx.x * kWave + threadIdx.x; q < hw4; q += stride) {
      if (U8 == U8_ROWS_DWORD && (q * 4) / w >= u8.crop_h) break;  // rows only grow with q
      v4f O[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) O[c] = ld_stream(x + plane0 + static_cast<long long>(c) * hw4 + q);
      store_u8_group<KIND, U8>(O, img, q, w, sp, u8);
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void export_u8_scalar(const float* __restrict__ x, long long n_img, int hw, int w, SpaceK sp,
                                                           U8Dst u8) {
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < hw; p += gridDim.x * kBlock) {
      const int row = p / w;
      const int col = p - row * w;
      if (row < u8.crop_h && col < u8.ncols) {
        uint8_t* dst = u8.base + img * u8.image_stride + row * u8.row_stride + col * 3LL;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          dst[c] = static_cast<uint8_t>(
              export_byte<KIND>(x[plane0 + static_cast<long long>(c) * hw + p], sp.scale[c], sp.shift[c], sp.export_add[c]));
      }
    }
  }
}

// a1 / a2: per-channel affine maps.  DIR 0: x*scale+shift, DIR 1: (x-shift)/scale
template <int DIR>
__device__ __forceinline__ float affine_elem(float x, float sc, float sh, float rcp, int use_rcp) {
  if (DIR == 0) {
    float d = x * sc;
    return d + sh;
  }
  return renorm(x, sc, sh, rcp, use_rcp);
}

template <int DIR>
__global__ __launch_bounds__(kWave) void affine_vec4(const v4f* x, v4f* out, long long n_img, int hw4, SpaceK sp) {
  const int stride = gridDim.x * kWave;
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    for (int q0 = blockIdx.x * kWave + threadIdx.x; q0 < hw4; q0 += stride * kUnroll) {
      v4f X[kUnroll][3];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
#pragma unroll
          for (int c = 0; c < 3; ++c) X[u][c] = ld_stream(x + plane0 + static_cast<long long>(c) * hw4 + q);
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            v4f o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = affine_elem<DIR>(X[u][c][j], sp.scale[c], sp.shift[c], sp.rcp[c], sp.use_rcp);
            st_stream(out + plane0 + static_cast<long long>(c) * hw4 + q, o);
          }
        }
      }
    }
  }
}

template <int DIR>
__global__ __launch_bounds__(kBlock) void affine_scalar(const float* x, float* out, long long n_img, int hw, SpaceK sp) {
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < hw; p += gridDim.x * kBlock) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long i = plane0 + static_cast<long long>(c) * hw + p;
        out[i] = affine_elem<DIR>(x[i], sp.scale[c], sp.shift[c], sp.rcp[c], sp.use_rcp);
      }
    }
  }
}

// a7: the disc mask as a tensor (for callers that still want one)
__global__ __launch_bounds__(kBlock) void disc_mask_kernel(float* mask, int h, int w, int cy, int cx, long long r2) {
  const int total = h * w;
  for (int p = blockIdx.x * kBlock + threadIdx.x; p < total; p += gridDim.x * kBlock) {
    const int y = p / w;
    const int x = p - y * w;
    const long long dy = y - cy, dx = x - cx;
    mask[p] = (dy * dy + dx * dx <= r2) ? 1.0f : 0.0f;
  }
}

// ------------------------------------------------------------------------------------------
// K3: patch paste.  Only the d x d bounding square is touched; the mask is analytic.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void paste_elem(float* img_plane, const float* patch_plane, int w, int d, int r, int i, int j,
                                           int y, int x) {
  const int dy = i - r, dx = j - r;
  const float m = (dy * dy + dx * dx <= r * r) ? 1.0f : 0.0f;  // mask_l, patch_attack.py:245-248
  const float one_minus = 1.0f - m;                             // (1 - mask_l)
  float* px = img_plane + static_cast<long long>(y) * w + x;
  const float a = one_minus * (*px);                            // torch.mul((1 - mask_l), imgL.data)
  const float b = m * patch_plane[i * d + j];                   // torch.mul(mask_l, patch_l)
  *px = a + b;
}

__global__ __launch_bounds__(kBlock) void patch_paste_kernel(float* img, const float* __restrict__ patch, long long n_img,
                                                             int h, int w, int d, int r, int cy0, int cx0,
                                                             const int32_t* __restrict__ centers) {
  const int dd = d * d;
  for (long long b = blockIdx.y; b < n_img; b += gridDim.y) {
    const int cy = centers ? centers[2 * b] : cy0;
    const int cx = centers ? centers[2 * b + 1] : cx0;
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < 3 * dd; e += gridDim.x * kBlock) {
      const int c = e / dd;
      const int ij = e - c * dd;
      const int i = ij / d;
      const int j = ij - i * d;
      const int y = cy - r + i, x = cx - r + j;
      if (y < 0 || y >= h || x < 0 || x >= w) continue;
      paste_elem(img + (b * 3 + c) * static_cast<long long>(h) * w, patch + static_cast<long long>(c) * dd, w, d, r, i, j, y, x);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K4: patch delta / update.  One lane per patch element; the n image pairs are folded in
// index order so the float32 sum is reproducible.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float delta_elem(const float* gl, const float* gr, long long b, int c, int h, int w, int i,
                                            int j, int cy, int cxl, int cxr, int r, float half_alpha, float eps) {
  const int y = cy - r + i;
  const long long plane = (b * 3 + c) * static_cast<long long>(h) * w + static_cast<long long>(y) * w;
  const float s = gl[plane + (cxl - r + j)] + gr[plane + (cxr - r + j)];  // imgL_grad + imgR_grad
  return t_clamp(half_alpha * s, -eps, eps);                              // clamp(0.5*alpha*(...), -eps, eps)
}

struct Lim3 {
  float lo[3];
  float hi[3];
  int on;
};

// MODE 0: patch update for one pair (scalar centres); MODE 1: summed delta of n pairs (device centres)
template <int MODE>
__global__ __launch_bounds__(kBlock) void patch_delta_kernel(float* patch, const float* __restrict__ gl,
                                                             const float* __restrict__ gr, long long n_pairs, int h, int w,
                                                             int d, int r, int cy0, int cxl0, int cxr0,
                                                             const int32_t* __restrict__ centers, float half_alpha,
                                                             float eps, Lim3 lim, float* delta_out) {
  const int dd = d * d;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= 3 * dd) return;
  const int c = e / dd;
  const int ij = e - c * dd;
  const int i = ij / d;
  const int j = ij - i * d;
  if (MODE == 0) {
    const float dl = delta_elem(gl, gr, 0, c, h, w, i, j, cy0, cxl0, cxr0, r, half_alpha, eps);
    float p = patch[e] - dl;  // patch -= ...
    if (lim.on) p = t_clamp(p, lim.lo[c], lim.hi[c]);
    patch[e] = p;
    if (delta_out) delta_out[e] = dl;
  } else {
    float acc = 0.0f;
    for (long long b = 0; b < n_pairs; ++b) {
      const int cy = centers[3 * b], cxl = centers[3 * b + 1], cxr = centers[3 * b + 2];
      float dl = 0.0f;
      const bool ok = cy - r >= 0 && cy + r < h && cxl - r >= 0 && cxl + r < w && cxr - r >= 0 && cxr + r < w;
      if (ok) dl = delta_elem(gl, gr, b, c, h, w, i, j, cy, cxl, cxr, r, half_alpha, eps);
      acc = (b == 0) ? dl : acc + dl;
    }
    delta_out[e] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void patch_apply_kernel(float* patch, const float* __restrict__ delta, int d, Lim3 lim) {
  const int dd = d * d;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= 3 * dd) return;
  float p = patch[e] - delta[e];
  if (lim.on) p = t_clamp(p, lim.lo[e / dd], lim.hi[e / dd]);
  patch[e] = p;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

inline SpaceK to_kernel_space(const adv_space_t* s) {
  SpaceK k;
  for (int c = 0; c < 3; ++c) {
    k.scale[c] = s->scale[c];
    k.shift[c] = s->shift[c];
    k.lo[c] = s->lo[c];
    k.hi[c] = s->hi[c];
    k.export_add[c] = s->export_add[c];
    // torch's GPU division by a Python scalar b multiplies by float(1.0 / b) with the reciprocal taken in DOUBLE from the
    // double b (measured on torch 2.10 + ROCm: float(1/0.224) = 4.46428585, whereas 1.0f / 0.224f = 4.46428537); for
    // AFFINE_RCP spaces export_add[c] carries that double divisor
    k.rcp[c] = s->kind == ADV_SPACE_AFFINE_RCP ? static_cast<float>(1.0 / s->export_add[c]) : 1.0f / s->scale[c];
  }
  k.use_rcp = s->kind == ADV_SPACE_AFFINE_RCP ? 1 : 0;
  return k;
}

inline int finish_launch() { return adv_internal_finish_launch(); }

// Vector streaming kernels: one 64-lane workgroup per `unroll` tiles of 64 pixel groups, grid.y = images.
inline dim3 wave_grid(long long groups_per_image, long long n_img, int unroll) {
  long long bx = (groups_per_image + static_cast<long long>(kWave) * unroll - 1) / (static_cast<long long>(kWave) * unroll);
  if (bx < 1) bx = 1;
  long long by = n_img < 1 ? 1 : (n_img > 65535 ? 65535 : n_img);
  return dim3(static_cast<unsigned>(bx), static_cast<unsigned>(by), 1);
}

// Scalar fall-back kernels (irregular shapes): 256-thread blocks, grid-stride beyond ~16 blocks per CU.
inline dim3 stream_grid(long long work_items_per_image, long long n_img) {
  long long bx = (work_items_per_image + kBlock - 1) / kBlock;
  if (bx < 1) bx = 1;
  long long by = n_img < 1 ? 1 : n_img;
  const long long kTarget = 256LL * 16;
  if (by > 65535) by = 65535;
  if (bx * by > kTarget) {
    long long want_x = kTarget / by;
    if (want_x < 1) want_x = 1;
    if (want_x < bx) bx = want_x;
  }
  return dim3(static_cast<unsigned>(bx), static_cast<unsigned>(by), 1);
}

struct U8Plan {
  U8Dst dst;
  int mode;
};

inline int plan_u8(uint8_t* u8, int h, int w, int crop_h, int crop_w, long long row_stride, long long image_stride,
                   bool vec_ok, U8Plan* out, bool allow_flat = true) {
  out->dst = U8Dst{nullptr, 0, 0, 0, 0};
  out->mode = U8_NONE;
  if (u8 == nullptr) return ADV_OK;
  if (crop_h < 1 || crop_h > h || crop_w < 1 || crop_w > w) return ADV_EINVAL;
  if (row_stride < 3LL * crop_w) return ADV_EINVAL;
  if (image_stride < row_stride * (crop_h - 1) + 3LL * crop_w) return ADV_EINVAL;
  const bool whole_rows = row_stride >= 3LL * w;
  out->dst = U8Dst{u8, row_stride, image_stride, crop_h, whole_rows ? w : crop_w};
  const bool dword_ok = vec_ok && whole_rows && (w % 4 == 0) && (row_stride % 4 == 0) && (image_stride % 4 == 0) &&
                        aligned(u8, 4) && image_stride >= row_stride * (crop_h - 1) + 3LL * w;
  const bool flat_ok = allow_flat && vec_ok && row_stride == 3LL * w && crop_h == h && crop_w == w && (image_stride % 4 == 0) && aligned(u8, 4) &&
                       image_stride >= 3LL * h * w;
  out->mode = dword_ok ? U8_ROWS_DWORD : (flat_ok ? U8_FLAT_DWORD : U8_BYTES);
  return ADV_OK;
}

inline bool is_affine(const adv_space_t* s) { return s->kind == ADV_SPACE_AFFINE || s->kind == ADV_SPACE_AFFINE_RCP; }

inline int check_space(const adv_space_t* s) {
  if (s == nullptr) return ADV_EINVAL;
  if (s->kind != ADV_SPACE_AFFINE && s->kind != ADV_SPACE_IDENTITY && s->kind != ADV_SPACE_AFFINE_RCP) return ADV_EINVAL;
  if (s->kind == ADV_SPACE_AFFINE_RCP)
    for (int c = 0; c < 3; ++c)
      if (!(s->export_add[c] > 0.0)) return ADV_EINVAL;  // the double divisor must be given
  return ADV_OK;
}

template <int KIND>
int launch_pgd(const float* x, const float* g, const float* cl, float* xo, long long n, int h, int w, const SpaceK& sp,
               float alpha, float eps, uint8_t* u8, int crop_h, int crop_w, long long rs, long long is, hipStream_t st) {
  const long long hw = static_cast<long long>(h) * w;
  const bool vec = (hw % 4 == 0) && aligned(x, 16) && aligned(g, 16) && aligned(cl, 16) && aligned(xo, 16);
  U8Plan plan;
  const int rc = plan_u8(u8, h, w, crop_h, crop_w, rs, is, vec, &plan);
  if (rc != ADV_OK) return rc;
  const uintptr_t res = reinterpret_cast<uintptr_t>(x) & 127;
  const bool same_residue = (reinterpret_cast<uintptr_t>(g) & 127) == res && (reinterpret_cast<uintptr_t>(cl) & 127) == res &&
                            (reinterpret_cast<uintptr_t>(xo) & 127) == res;
  if (vec && same_residue && (((hw / 4) & 7) != 0 || res != 0)) {
    // planes are not whole cache lines (or the buffers start inside one): per-channel shifted tiles
    const int hw4 = static_cast<int>(hw / 4);
    const int tiles = (hw4 + 7 + kShiftTile - 1) / kShiftTile;
    const dim3 grid(tiles, static_cast<unsigned>(n > 65535 ? 65535 : n), 1);
    const int base_f4 = static_cast<int>(res / 16);
    const v4f* x4 = reinterpret_cast<const v4f*>(x);
    const v4f* g4 = reinterpret_cast<const v4f*>(g);
    const v4f* c4 = reinterpret_cast<const v4f*>(cl);
    v4f* o4 = reinterpret_cast<v4f*>(xo);
#define ADV_LAUNCH_SHIFTED(MODE) \
  hipLaunchKernelGGL((pgd_step_shifted<KIND, MODE>), grid, dim3(kShiftBlock), 0, st, x4, g4, c4, o4, n, hw4, w, base_f4, sp, alpha, eps, plan.dst)
    switch (plan.mode) {
      case U8_NONE: ADV_LAUNCH_SHIFTED(U8_NONE); break;
      case U8_ROWS_DWORD: ADV_LAUNCH_SHIFTED(U8_ROWS_DWORD); break;
      case U8_FLAT_DWORD: ADV_LAUNCH_SHIFTED(U8_FLAT_DWORD); break;
      default: ADV_LAUNCH_SHIFTED(U8_BYTES); break;
    }
#undef ADV_LAUNCH_SHIFTED
  } else if (vec) {
    const int hw4 = static_cast<int>(hw / 4);
    const dim3 grid = wave_grid(hw4, n, kUnroll);
    const v4f* x4 = reinterpret_cast<const v4f*>(x);
    const v4f* g4 = reinterpret_cast<const v4f*>(g);
    const v4f* c4 = reinterpret_cast<const v4f*>(cl);
    v4f* o4 = reinterpret_cast<v4f*>(xo);
#define ADV_LAUNCH_VEC4(MODE) \
  hipLaunchKernelGGL((pgd_step_vec4<KIND, MODE>), grid, dim3(kWave), 0, st, x4, g4, c4, o4, n, hw4, w, sp, alpha, eps, plan.dst)
    switch (plan.mode) {
      case U8_NONE: ADV_LAUNCH_VEC4(U8_NONE); break;
      case U8_ROWS_DWORD: ADV_LAUNCH_VEC4(U8_ROWS_DWORD); break;
      case U8_FLAT_DWORD: ADV_LAUNCH_VEC4(U8_FLAT_DWORD); break;
      default: ADV_LAUNCH_VEC4(U8_BYTES); break;
    }
#undef ADV_LAUNCH_VEC4
  } else {
    const dim3 grid = stream_grid(hw, n);
    hipLaunchKernelGGL((pgd_step_scalar<KIND>), grid, dim3(kBlock), 0, st, x, g, cl, xo, n, static_cast<int>(hw), w, sp, alpha, eps, plan.dst);
  }
  return finish_launch();
}

// a13 with the indexed clean image, in the line-aligned kernel (planes that are not whole cache lines - the Stereo R-CNN frame);
// every other layout steps through the float32 kernels, which give the same bits.  Images per workgroup (= per staged copy of
// the table), 128 images of 600 x 1987, in place + export: 1 -> 1.17 ms, 2 -> 1.08, 4 -> 1.14, 8 -> 1.17, 16 -> 1.28; all-float32
// kernel 1.28-1.30 (profiles/r02_srcnn_index_sweep.log).
constexpr long long kIdxImagesPerWorkgroup = 2;

inline int step_indexed_identity(const float* x, const float* g, const float* cl, const IdxK& ik, float* xo, uint8_t* u8, long long n, int h,
                                 int w, const adv_space_t* space, float alpha, float eps, int crop_h, int crop_w, long long rs, long long is,
                                 hipStream_t st) {
  const SpaceK sp = to_kernel_space(space);
  const long long hw = static_cast<long long>(h) * w;
  const bool vec = (hw % 4 == 0) && aligned(x, 16) && aligned(g, 16) && aligned(cl, 16) && aligned(xo, 16);
  const uintptr_t res = reinterpret_cast<uintptr_t>(x) & 127;
  const bool same_residue = (reinterpret_cast<uintptr_t>(g) & 127) == res && (reinterpret_cast<uintptr_t>(cl) & 127) == res &&
                            (reinterpret_cast<uintptr_t>(xo) & 127) == res;
  const bool shifted = vec && same_residue && (((hw / 4) & 7) != 0 || res != 0);
  if (!shifted || n < kIdxImagesPerWorkgroup)
    return launch_pgd<ADV_SPACE_IDENTITY>(x, g, cl, xo, n, h, w, sp, alpha, eps, u8, crop_h, crop_w, rs, is, st);
  U8Plan plan;
  const int rc = plan_u8(u8, h, w, crop_h, crop_w, rs, is, vec, &plan);
  if (rc != ADV_OK) return rc;
  const int hw4 = static_cast<int>(hw / 4);
  const int tiles = (hw4 + 7 + kShiftTile - 1) / kShiftTile;
  long long gy = (n + kIdxImagesPerWorkgroup - 1) / kIdxImagesPerWorkgroup;
  if (gy > 65535) gy = 65535;
  const dim3 grid(tiles, static_cast<unsigned>(gy), 1);
  const int base_f4 = static_cast<int>(res / 16);
  const v4f* x4 = reinterpret_cast<const v4f*>(x);
  const v4f* g4 = reinterpret_cast<const v4f*>(g);
  const v4f* c4 = reinterpret_cast<const v4f*>(cl);
  v4f* o4 = reinterpret_cast<v4f*>(xo);
#define ADV_LAUNCH_SHIFTED_IDX(MODE)                                                                                                          \
  hipLaunchKernelGGL((pgd_step_shifted<ADV_SPACE_IDENTITY, MODE, true>), grid, dim3(kShiftBlock), 0, st, x4, g4, c4, o4, n, hw4, w, base_f4, sp, \
                     alpha, eps, plan.dst, ik)
  switch (plan.mode) {
    case U8_NONE: ADV_LAUNCH_SHIFTED_IDX(U8_NONE); break;
    case U8_ROWS_DWORD: ADV_LAUNCH_SHIFTED_IDX(U8_ROWS_DWORD); break;
    case U8_FLAT_DWORD: ADV_LAUNCH_SHIFTED_IDX(U8_FLAT_DWORD); break;
    default: ADV_LAUNCH_SHIFTED_IDX(U8_BYTES); break;
  }
#undef ADV_LAUNCH_SHIFTED_IDX
  return finish_launch();
}

inline Lim3 make_lim(const float* lo, const float* hi) {
  Lim3 l;
  l.on = (lo != nullptr && hi != nullptr) ? 1 : 0;
  for (int c = 0; c < 3; ++c) {
    l.lo[c] = l.on ? lo[c] : 0.0f;
    l.hi[c] = l.on ? hi[c] : 0.0f;
  }
  return l;
}

inline bool window_inside(int h, int w, int cy, int cx, int r) {
  return r >= 0 && cy - r >= 0 && cx - r >= 0 && cy + r < h && cx + r < w;
}

constexpr long long kMaxPlane = 1LL << 30;  // h*w fits comfortably in int

}  // namespace

void adv_internal_set_last_hip_error(int e) { g_last_hip_error = e; }

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int adv_abi_version(void) { return ADV_ABI_VERSION; }

int adv_build_has_test_hooks(void) {
#ifdef ADV_TEST_HOOKS
  return 1;
#else
  return 0;
#endif
}

int adv_last_hip_error(void) { return g_last_hip_error; }

const char* adv_strerror(int code) {
  switch (code) {
    case ADV_OK: return "ok";
    case ADV_EINVAL: return "invalid argument";
    case ADV_EALIGN: return "float pointer not 4-byte aligned";
    case ADV_ELAUNCH: return "HIP kernel launch failed (see adv_last_hip_error)";
    default: return "unknown advengine error";
  }
}

void adv_space_dsgn(adv_space_t* s) {
  // attack/DSGN/pgd_attack.py:153-154; Python doubles rounded to float32 as torch does
  const double mean[3] = {0.485, 0.456, 0.406};
  const double stdv[3] = {0.229, 0.224, 0.225};
  s->kind = ADV_SPACE_AFFINE;
  for (int c = 0; c < 3; ++c) {
    s->scale[c] = static_cast<float>(stdv[c]);
    s->shift[c] = static_cast<float>(mean[c]);
    s->lo[c] = 0.0f;  // torch.clamp(..., min=0, max=1), pgd_attack.py:349-350
    s->hi[c] = 1.0f;
    s->export_add[c] = 0.0;
  }
}

void adv_space_dsgn_gpu_reference(adv_space_t* s) {
  const double stdv[3] = {0.229, 0.224, 0.225};
  adv_space_dsgn(s);
  s->kind = ADV_SPACE_AFFINE_RCP;
  for (int c = 0; c < 3; ++c) s->export_add[c] = stdv[c];  // the divisor as the Python double the script divides by
}

void adv_space_srcnn(adv_space_t* s) {
  // attack/Stereo-RCNN/pgd_attack.py:189-207: min=(0 - m_c), max=(255 - m_c) as Python doubles
  const double m[3] = {102.9801, 115.9465, 122.7717};
  s->kind = ADV_SPACE_IDENTITY;
  for (int c = 0; c < 3; ++c) {
    s->scale[c] = 1.0f;
    s->shift[c] = 0.0f;
    s->lo[c] = static_cast<float>(0 - m[c]);
    s->hi[c] = static_cast<float>(255 - m[c]);
    s->export_add[c] = m[c];
  }
}

static int check_image_args(const void* a, const void* b, long long n, int h, int w) {
  if (a == nullptr || b == nullptr) return ADV_EINVAL;
  if (n < 1 || h < 1 || w < 1 || static_cast<long long>(h) * w > kMaxPlane) return ADV_EINVAL;
  if (!aligned(a, 4) || !aligned(b, 4)) return ADV_EALIGN;
  return ADV_OK;
}

static int launch_affine(int dir, const float* x, float* out, int64_t n, int h, int w, const adv_space_t* space,
                         adv_stream_t stream) {
  int rc = check_image_args(x, out, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_space(space);
  if (rc != ADV_OK) return rc;
  if (!is_affine(space)) return ADV_EINVAL;
  const long long hw = static_cast<long long>(h) * w;
  const bool vec = (hw % 4 == 0) && aligned(x, 16) && aligned(out, 16);
  const SpaceK sp = to_kernel_space(space);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long nn = n;
  if (vec) {
    const int hw4 = static_cast<int>(hw / 4);
    const dim3 grid = wave_grid(hw4, nn, kUnroll);
    const v4f* x4 = reinterpret_cast<const v4f*>(x);
    v4f* o4 = reinterpret_cast<v4f*>(out);
    if (dir == 0)
      hipLaunchKernelGGL((affine_vec4<0>), grid, dim3(kWave), 0, st, x4, o4, nn, hw4, sp);
    else
      hipLaunchKernelGGL((affine_vec4<1>), grid, dim3(kWave), 0, st, x4, o4, nn, hw4, sp);
  } else {
    const dim3 grid = stream_grid(hw, nn);
    if (dir == 0)
      hipLaunchKernelGGL((affine_scalar<0>), grid, dim3(kBlock), 0, st, x, out, nn, static_cast<int>(hw), sp);
    else
      hipLaunchKernelGGL((affine_scalar<1>), grid, dim3(kBlock), 0, st, x, out, nn, static_cast<int>(hw), sp);
  }
  return finish_launch();
}

int adv_denormalize_f32(const float* x, float* out, int64_t n, int h, int w, const adv_space_t* space, adv_stream_t stream) {
  return launch_affine(0, x, out, n, h, w, space, stream);
}

int adv_normalize_f32(const float* x, float* out, int64_t n, int h, int w, const adv_space_t* space, adv_stream_t stream) {
  return launch_affine(1, x, out, n, h, w, space, stream);
}

int adv_pgd_step_f32(const float* x, const float* grad, const float* clean, float* x_out, uint8_t* u8_out, int64_t n,
                     int h, int w, const adv_space_t* space, float alpha, float eps, int crop_h, int crop_w,
                     int64_t u8_row_stride, int64_t u8_image_stride, adv_stream_t stream) {
  int rc = check_image_args(x, x_out, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_image_args(grad, clean, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_space(space);
  if (rc != ADV_OK) return rc;
  if (!(eps >= 0.0f)) return ADV_EINVAL;  // also rejects NaN
  const SpaceK sp = to_kernel_space(space);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (is_affine(space))
    return launch_pgd<ADV_SPACE_AFFINE>(x, grad, clean, x_out, n, h, w, sp, alpha, eps, u8_out, crop_h, crop_w, u8_row_stride, u8_image_stride, st);
  return launch_pgd<ADV_SPACE_IDENTITY>(x, grad, clean, x_out, n, h, w, sp, alpha, eps, u8_out, crop_h, crop_w, u8_row_stride, u8_image_stride, st);
}

static int check_clean_index(const adv_clean_index_t* ci, int h, int w, IdxK* out) {
  if (ci == nullptr || ci->index == nullptr || ci->ok == nullptr || ci->lut == nullptr) return ADV_EINVAL;
  if (!aligned(ci->index, 4) || !aligned(ci->ok, 4) || !aligned(ci->lut, 16) || (ci->valid_hw && !aligned(ci->valid_hw, 4))) return ADV_EALIGN;
  if (ci->valid_hw == nullptr && (ci->valid_h < 0 || ci->valid_h > h || ci->valid_w < 0 || ci->valid_w > w)) return ADV_EINVAL;
  *out = IdxK{reinterpret_cast<uint32_t*>(ci->index), reinterpret_cast<int*>(ci->ok), ci->lut, reinterpret_cast<const int*>(ci->valid_hw),
              ci->valid_h, ci->valid_w};
  return ADV_OK;
}

int adv_clean_index_build_f32(const float* x, float* clean_out, const adv_clean_index_t* ci, uint8_t* u8_out, int64_t n, int h, int w,
                              const adv_space_t* space, int crop_h, int crop_w, int64_t u8_row_stride, int64_t u8_image_stride,
                              adv_stream_t stream) {
  int rc = check_image_args(x, clean_out, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_space(space);
  if (rc != ADV_OK) return rc;
  IdxK ik;
  rc = check_clean_index(ci, h, w, &ik);
  if (rc != ADV_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const SpaceK sp = to_kernel_space(space);
  if (!is_affine(space)) {
    // identity space (Stereo R-CNN): the whole frame is image, rows of any length; clean_out = x (may alias it)
    if (ci->valid_hw != nullptr || ci->valid_h != h || ci->valid_w != w) return ADV_EINVAL;
    if ((static_cast<long long>(h) * w) % 4 != 0 || !aligned(x, 16) || !aligned(clean_out, 16)) return ADV_EALIGN;
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ci->ok), 3, static_cast<size_t>(n), st) != hipSuccess) return ADV_ELAUNCH;
    hipLaunchKernelGGL(clean_lut_kernel, dim3(1), dim3(256), 0, st, ci->lut, sp, 1);
    const int hw4i = static_cast<int>(static_cast<long long>(h) * w / 4);
    hipLaunchKernelGGL(clean_index_build_identity, wave_grid(hw4i, n, 4), dim3(kWave), 0, st, reinterpret_cast<const v4f*>(x),
                       reinterpret_cast<v4f*>(clean_out), ik, static_cast<long long>(n), hw4i, sp, clean_out != x ? 1 : 0);
    rc = finish_launch();
    if (rc != ADV_OK || u8_out == nullptr) return rc;
    return adv_export_u8_f32(x, u8_out, n, h, w, space, crop_h, crop_w, u8_row_stride, u8_image_stride, stream);  // iterate 0
  }
  if (w % 4 != 0 || !aligned(x, 16) || !aligned(clean_out, 16)) return ADV_EALIGN;
  U8Plan plan;
  rc = plan_u8(u8_out, h, w, crop_h, crop_w, u8_row_stride, u8_image_stride, true, &plan, false);
  if (rc != ADV_OK) return rc;
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ci->ok), 1, static_cast<size_t>(n), st) != hipSuccess) return ADV_ELAUNCH;
  hipLaunchKernelGGL(clean_lut_kernel, dim3(1), dim3(256), 0, st, ci->lut, sp, 0);
  const int hw4 = static_cast<int>(static_cast<long long>(h) * w / 4);
  const dim3 grid = wave_grid(hw4, n, 1);
  const v4f* x4 = reinterpret_cast<const v4f*>(x);
  v4f* c4 = reinterpret_cast<v4f*>(clean_out);
  const long long nn = n;
  if (plan.mode == U8_NONE)
    hipLaunchKernelGGL((clean_index_build_vec4<U8_NONE>), grid, dim3(kWave), 0, st, x4, c4, ik, nn, hw4, w, sp, plan.dst);
  else if (plan.mode == U8_ROWS_DWORD)
    hipLaunchKernelGGL((clean_index_build_vec4<U8_ROWS_DWORD>), grid, dim3(kWave), 0, st, x4, c4, ik, nn, hw4, w, sp, plan.dst);
  else
    hipLaunchKernelGGL((clean_index_build_vec4<U8_BYTES>), grid, dim3(kWave), 0, st, x4, c4, ik, nn, hw4, w, sp, plan.dst);
  return finish_launch();
}

int adv_import_u8_f32(const uint8_t* u8_hwc, int64_t u8_row_stride, int64_t u8_image_stride, float* x_out, float* clean_out,
                      const adv_clean_index_t* ci, int valid_h, int valid_w, int64_t n, int h, int w, const adv_space_t* space,
                      adv_stream_t stream) {
  if (u8_hwc == nullptr || x_out == nullptr || n < 1 || h < 1 || w < 1 || static_cast<long long>(h) * w > kMaxPlane) return ADV_EINVAL;
  int rc = check_space(space);
  if (rc != ADV_OK) return rc;
  if (!is_affine(space) || space->kind == ADV_SPACE_AFFINE_RCP) return ADV_EINVAL;   // the loader runs on the CPU: true divisions
  if (w % 4 != 0 || !aligned(x_out, 16) || (clean_out && !aligned(clean_out, 16))) return ADV_EALIGN;
  IdxK ik{nullptr, nullptr, nullptr, nullptr, valid_h, valid_w};
  if (ci != nullptr) {
    rc = check_clean_index(ci, h, w, &ik);
    if (rc != ADV_OK) return rc;
  } else if (valid_h < 0 || valid_h > h || valid_w < 0 || valid_w > w) {
    return ADV_EINVAL;
  }
  // every image's rows must lie inside its own buffer: valid_h rows of at least 3*valid_w bytes (per-image sizes are device data:
  // the caller guarantees them against the strides it passes)
  if (u8_row_stride < 3LL * (ci && ci->valid_hw ? 1 : ik.vw) || u8_image_stride < 0) return ADV_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const SpaceK sp = to_kernel_space(space);
  if (ci != nullptr) {
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ci->ok), 1, static_cast<size_t>(n), st) != hipSuccess) return ADV_ELAUNCH;
    hipLaunchKernelGGL(clean_lut_kernel, dim3(1), dim3(256), 0, st, ci->lut, sp, 0);
  }
  const int hw4 = static_cast<int>(static_cast<long long>(h) * w / 4);
  const dim3 grid = wave_grid(hw4, n, 2);
  const bool al = aligned(u8_hwc, 4) && u8_row_stride % 4 == 0 && u8_image_stride % 4 == 0;
  v4f* x4 = reinterpret_cast<v4f*>(x_out);
  v4f* c4 = reinterpret_cast<v4f*>(clean_out);
  const long long nn = n;
  if (al)
    hipLaunchKernelGGL((import_u8_vec4<true>), grid, dim3(kWave), 0, st, u8_hwc, u8_row_stride, u8_image_stride, x4, c4, ik, nn, hw4, w, sp);
  else
    hipLaunchKernelGGL((import_u8_vec4<false>), grid, dim3(kWave), 0, st, u8_hwc, u8_row_stride, u8_image_stride, x4, c4, ik, nn, hw4, w, sp);
  return finish_launch();
}

int adv_pgd_step_indexed_f32(const float* x, const float* grad, const float* clean, const adv_clean_index_t* ci, float* x_out,
                             uint8_t* u8_out, int64_t n, int h, int w, const adv_space_t* space, float alpha, float eps, int crop_h,
                             int crop_w, int64_t u8_row_stride, int64_t u8_image_stride, adv_stream_t stream) {
  int rc = check_image_args(x, x_out, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_image_args(grad, clean, n, h, w);
  if (rc != ADV_OK) return rc;
  rc = check_space(space);
  if (rc != ADV_OK) return rc;
  if (!(eps >= 0.0f)) return ADV_EINVAL;
  IdxK ik;
  rc = check_clean_index(ci, h, w, &ik);
  if (rc != ADV_OK) return rc;
  if (!is_affine(space)) return step_indexed_identity(x, grad, clean, ik, x_out, u8_out, n, h, w, space, alpha, eps, crop_h, crop_w, u8_row_stride,
                                                       u8_image_stride, static_cast<hipStream_t>(stream));
  if (w % 4 != 0 || !aligned(x, 16) || !aligned(grad, 16) || !aligned(clean, 16) || !aligned(x_out, 16)) return ADV_EALIGN;
  U8Plan plan;
  rc = plan_u8(u8_out, h, w, crop_h, crop_w, u8_row_stride, u8_image_stride, true, &plan, false);
  if (rc != ADV_OK) return rc;
  const SpaceK sp = to_kernel_space(space);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int hw4 = static_cast<int>(static_cast<long long>(h) * w / 4);
  const dim3 grid = wave_grid(hw4, n, kUnroll);
  const v4f* x4 = reinterpret_cast<const v4f*>(x);
  const v4f* g4 = reinterpret_cast<const v4f*>(grad);
  const v4f* c4 = reinterpret_cast<const v4f*>(clean);
  v4f* o4 = reinterpret_cast<v4f*>(x_out);
  const long long nn = n;
  if (plan.mode == U8_NONE)
    hipLaunchKernelGGL((pgd_step_vec4_idx<U8_NONE>), grid, dim3(kWave), 0, st, x4, g4, c4, ik, o4, nn, hw4, w, sp, alpha, eps, plan.dst);
  else if (plan.mode == U8_ROWS_DWORD)
    hipLaunchKernelGGL((pgd_step_vec4_idx<U8_ROWS_DWORD>), grid, dim3(kWave), 0, st, x4, g4, c4, ik, o4, nn, hw4, w, sp, alpha, eps, plan.dst);
  else
    hipLaunchKernelGGL((pgd_step_vec4_idx<U8_BYTES>), grid, dim3(kWave), 0, st, x4, g4, c4, ik, o4, nn, hw4, w, sp, alpha, eps, plan.dst);
  return finish_launch();
}

int adv_export_u8_f32(const float* x, uint8_t* u8_out, int64_t n, int h, int w, const adv_space_t* space, int crop_h,
                      int crop_w, int64_t u8_row_stride, int64_t u8_image_stride, adv_stream_t stream) {
  int rc = check_image_args(x, u8_out, n, h, w);
  if (rc == ADV_EALIGN && aligned(x, 4)) rc = ADV_OK;  // the byte destination needs no alignment
  if (rc != ADV_OK) return rc;
  rc = check_space(space);
  if (rc != ADV_OK) return rc;
  const long long hw = static_cast<long long>(h) * w;
  const bool vec = (hw % 4 == 0) && aligned(x, 16);
  U8Plan plan;
  rc = plan_u8(u8_out, h, w, crop_h, crop_w, u8_row_stride, u8_image_stride, vec, &plan);
  if (rc != ADV_OK) return rc;
  const SpaceK sp = to_kernel_space(space);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long nn = n;
  if (plan.mode == U8_ROWS_DWORD || plan.mode == U8_FLAT_DWORD) {
    const int hw4 = static_cast<int>(hw / 4);
    const dim3 grid = wave_grid(plan.mode == U8_ROWS_DWORD ? static_cast<long long>(crop_h) * w / 4 : hw4, nn, 1);
    const v4f* x4 = reinterpret_cast<const v4f*>(x);
    const bool rows = plan.mode == U8_ROWS_DWORD;
    if (is_affine(space)) {
      if (rows) hipLaunchKernelGGL((export_u8_vec4<ADV_SPACE_AFFINE, U8_ROWS_DWORD>), grid, dim3(kWave), 0, st, x4, nn, hw4, w, sp, plan.dst);
      else hipLaunchKernelGGL((export_u8_vec4<ADV_SPACE_AFFINE, U8_FLAT_DWORD>), grid, dim3(kWave), 0, st, x4, nn, hw4, w, sp, plan.dst);
    } else {
      if (rows) hipLaunchKernelGGL((export_u8_vec4<ADV_SPACE_IDENTITY, U8_ROWS_DWORD>), grid, dim3(kWave), 0, st, x4, nn, hw4, w, sp, plan.dst);
      else hipLaunchKernelGGL((export_u8_vec4<ADV_SPACE_IDENTITY, U8_FLAT_DWORD>), grid, dim3(kWave), 0, st, x4, nn, hw4, w, sp, plan.dst);
    }
  } else {
    const dim3 grid = stream_grid(hw, nn);
    if (is_affine(space))
      hipLaunchKernelGGL((export_u8_scalar<ADV_SPACE_AFFINE>), grid, dim3(kBlock), 0, st, x, nn, static_cast<int>(hw), w, sp, plan.dst);
    else
      hipLaunchKernelGGL((export_u8_scalar<ADV_SPACE_IDENTITY>), grid, dim3(kBlock), 0, st, x, nn, static_cast<int>(hw), w, sp, plan.dst);
  }
  return finish_launch();
}

int adv_disc_mask_f32(float* mask_out, int h, int w, int cy, int cx, int r, adv_stream_t stream) {
  if (mask_out == nullptr || h < 1 || w < 1 || r < 0 || static_cast<long long>(h) * w > kMaxPlane) return ADV_EINVAL;
  if (!aligned(mask_out, 4)) return ADV_EALIGN;
  const dim3 grid = stream_grid(static_cast<long long>(h) * w, 1);
  hipLaunchKernelGGL(disc_mask_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), mask_out, h, w, cy, cx,
                     static_cast<long long>(r) * r);
  return finish_launch();
}

static int check_patch_args(const void* a, const void* b, int h, int w, int d, int r) {
  if (a == nullptr || b == nullptr) return ADV_EINVAL;
  if (h < 1 || w < 1 || r < 0 || d != 2 * r + 1 || d > h || d > w || static_cast<long long>(h) * w > kMaxPlane) return ADV_EINVAL;
  if (!aligned(a, 4) || !aligned(b, 4)) return ADV_EALIGN;
  return ADV_OK;
}

int adv_patch_paste_f32(float* img, const float* patch, int h, int w, int d, int cy, int cx, int r, adv_stream_t stream) {
  const int rc = check_patch_args(img, patch, h, w, d, r);
  if (rc != ADV_OK) return rc;
  if (!window_inside(h, w, cy, cx, r)) return ADV_EINVAL;
  const dim3 grid((3 * d * d + kBlock - 1) / kBlock, 1, 1);
  hipLaunchKernelGGL(patch_paste_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), img, patch, 1LL, h, w, d, r, cy, cx,
                     static_cast<const int32_t*>(nullptr));
  return finish_launch();
}

int adv_patch_paste_batch_f32(float* img, const float* patch, int64_t n, int h, int w, int d, const int32_t* centers, int r,
                              adv_stream_t stream) {
  const int rc = check_patch_args(img, patch, h, w, d, r);
  if (rc != ADV_OK) return rc;
  if (centers == nullptr || n < 1) return ADV_EINVAL;
  if (!aligned(centers, 4)) return ADV_EALIGN;
  const dim3 grid((3 * d * d + kBlock - 1) / kBlock, static_cast<unsigned>(n > 65535 ? 65535 : n), 1);
  hipLaunchKernelGGL(patch_paste_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), img, patch, static_cast<long long>(n), h, w,
                     d, r, 0, 0, centers);
  return finish_launch();
}

int adv_patch_update_f32(float* patch, const float* grad_l, const float* grad_r, int h, int w, int d, int cy, int cx_l,
                         int cx_r, int r, float half_alpha, float eps, const float* lo, const float* hi, float* delta_out,
                         adv_stream_t stream) {
  int rc = check_patch_args(patch, grad_l, h, w, d, r);
  if (rc != ADV_OK) return rc;
  if (grad_r == nullptr || (lo == nullptr) != (hi == nullptr) || !(eps >= 0.0f)) return ADV_EINVAL;
  if (!aligned(grad_r, 4) || (delta_out && !aligned(delta_out, 4))) return ADV_EALIGN;
  if (!window_inside(h, w, cy, cx_l, r) || !window_inside(h, w, cy, cx_r, r)) return ADV_EINVAL;
  const dim3 grid((3 * d * d + kBlock - 1) / kBlock, 1, 1);
  hipLaunchKernelGGL((patch_delta_kernel<0>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), patch, grad_l, grad_r, 1LL, h, w, d,
                     r, cy, cx_l, cx_r, static_cast<const int32_t*>(nullptr), half_alpha, eps, make_lim(lo, hi), delta_out);
  return finish_launch();
}

int adv_patch_delta_batch_f32(const float* grad_l, const float* grad_r, int64_t n, int h, int w, int d, const int32_t* centers,
                              int r, float half_alpha, float eps, float* delta_out, adv_stream_t stream) {
  int rc = check_patch_args(grad_l, grad_r, h, w, d, r);
  if (rc != ADV_OK) return rc;
  if (centers == nullptr || delta_out == nullptr || n < 1 || !(eps >= 0.0f)) return ADV_EINVAL;
  if (!aligned(centers, 4) || !aligned(delta_out, 4)) return ADV_EALIGN;
  const dim3 grid((3 * d * d + kBlock - 1) / kBlock, 1, 1);
  hipLaunchKernelGGL((patch_delta_kernel<1>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), static_cast<float*>(nullptr), grad_l,
                     grad_r, static_cast<long long>(n), h, w, d, r, 0, 0, 0, centers, half_alpha, eps, make_lim(nullptr, nullptr), delta_out);
  return finish_launch();
}

int adv_patch_apply_f32(float* patch, const float* delta, int d, const float* lo, const float* hi, adv_stream_t stream) {
  if (patch == nullptr || delta == nullptr || d < 1 || (lo == nullptr) != (hi == nullptr)) return ADV_EINVAL;
  if (!aligned(patch, 4) || !aligned(delta, 4)) return ADV_EALIGN;
  const dim3 grid((3 * d * d + kBlock - 1) / kBlock, 1, 1);
  hipLaunchKernelGGL(patch_apply_kernel, grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), patch, delta, d, make_lim(lo, hi));
  return finish_launch();
}

}  // extern "C"
