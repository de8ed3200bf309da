// Tuning harness for the INDEXED PGD-step kernel (round 2).  Not part of the product.  It times ways of turning the
// 1-byte clean-image index back into the float32 clean value inside the step kernel:
//   DIV   : the round-1 kernel (two IEEE divisions per element)
//   LDS   : a 3x256 float table staged in LDS per workgroup, ds_read_b32 gathers
//   GLB   : the same table gathered straight from global memory (L1/L2 resident)
// with/without the zero-padding rule, 1-wave vs 4-wave workgroups, 1 or 2 trips, and (FASTDIV) the final
// (y - shift)/scale as a 5-instruction Markstein sequence with a precomputed reciprocal.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o k1_idx_tune tools/k1_idx_tune.hip && ./k1_idx_tune [pairs]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#pragma clang fp contract(off)

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));

struct Sp {
  float scale[3], shift[3], lo[3], hi[3], rcp[3];
};

__device__ __forceinline__ float t_sign(float g) { return (g > 0.0f ? 1.0f : 0.0f) - (g < 0.0f ? 1.0f : 0.0f); }
__device__ __forceinline__ float t_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <bool FASTDIV>
__device__ __forceinline__ float div_const(float a, float sc, float r) {
  if (!FASTDIV) return a / sc;
  float q = a * r;
  float e = __builtin_fmaf(-sc, q, a);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-sc, q, a);
  return __builtin_fmaf(e, r, q);
}

template <bool FASTDIV>
__device__ __forceinline__ float elem(float x, float g, float cl, float sc, float sh, float lo, float hi, float alpha, float eps, float r) {
  float d = x * sc;
  d = d + sh;
  const float a = d + alpha * t_sign(g);
  const float eta = t_clamp(a - cl, -eps, eps);
  const float y = t_clamp(cl + eta, lo, hi);
  return div_const<FASTDIV>(y - sh, sc, r);
}
__device__ __forceinline__ uint32_t byte_of(float xo, float sc, float sh) {
  float v = xo * sc;
  v = v + sh;
  v = v * 255.0f;
  if (!(fabsf(v) < 2147483648.0f)) return 0u;
  return static_cast<uint32_t>(static_cast<int>(v)) & 0xffu;
}
__device__ __host__ __forceinline__ float clean_from_index(uint32_t v, float sc, float sh) {
  float t = static_cast<float>(v) / 255.0f;
  t = (t - sh) / sc;
  t = t * sc;
  return t + sh;
}

enum { M_FLOAT = 0, M_DIV = 1, M_LDS = 2, M_GLB = 3 };

// MODE: how clean is obtained; PAD: apply the "outside valid_h x valid_w the clean value is shift_c" rule;
// BLOCK threads per workgroup; TRIPS tiles-of-2 per workgroup
template <int MODE, bool PAD, int BLOCK, int TRIPS, bool FASTDIV>
__global__ __launch_bounds__(BLOCK) void k1i(const v4f* x, const v4f* __restrict__ g, const v4f* __restrict__ cl,
                                             const uint32_t* __restrict__ idx, const int* __restrict__ ok,
                                             const float* __restrict__ lut_g, v4f* xo, uint8_t* u8, long long n_img, int hw4,
                                             int w, int crop_h, int vh, int vw, Sp sp, float alpha, float eps) {
  __shared__ float lut[MODE == M_LDS ? 768 : 4];
  if (MODE == M_LDS) {
    for (int k = threadIdx.x; k < 192; k += BLOCK) reinterpret_cast<v4f*>(lut)[k] = reinterpret_cast<const v4f*>(lut_g)[k];
    __syncthreads();
  }
  constexpr int UN = 2;
  const long long img = blockIdx.y;
  const bool use_idx = MODE != M_FLOAT && ok[img] != 0;
  const long long plane0 = img * 3LL * hw4;
  const int stride = gridDim.x * BLOCK;
#pragma unroll 1
  for (int t = 0; t < TRIPS; ++t) {
    const int q0 = (t * UN) * stride + blockIdx.x * BLOCK + threadIdx.x;
    v4f X[UN][3], G[UN][3], C[UN][3];
    uint32_t I[UN][3];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + u * stride;
      if (q < hw4) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const long long i = plane0 + (long long)c * hw4 + q;
          X[u][c] = __builtin_nontemporal_load(x + i);
          G[u][c] = __builtin_nontemporal_load(g + i);
          if (use_idx)
            I[u][c] = __builtin_nontemporal_load(idx + i);
          else
            C[u][c] = __builtin_nontemporal_load(cl + i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + u * stride;
      if (q < hw4) {
        const int p = q * 4;
        const int row = p / w;
        const int col = p - row * w;
        v4f O[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (use_idx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t v = (I[u][c] >> (8 * j)) & 0xffu;
              float cv;
              if (MODE == M_DIV) cv = clean_from_index(v, sp.scale[c], sp.shift[c]);
              if (MODE == M_LDS) cv = lut[c * 256 + v];
              if (MODE == M_GLB) cv = lut_g[c * 256 + v];
              if (PAD) cv = (row < vh && col + j < vw) ? cv : sp.shift[c];
              C[u][c][j] = cv;
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            O[c][j] = elem<FASTDIV>(X[u][c][j], G[u][c][j], C[u][c][j], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps, sp.rcp[c]);
          __builtin_nontemporal_store(O[c], xo + plane0 + (long long)c * hw4 + q);
        }
        if (row < crop_h) {
          uint32_t b[12];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) b[j * 3 + c] = byte_of(O[c][j], sp.scale[c], sp.shift[c]);
          v3u r;
          r[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
          r[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
          r[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
          __builtin_nontemporal_store(r, reinterpret_cast<v3u*>(u8 + (img * crop_h + row) * (3LL * w) + col * 3LL));
        }
      }
    }
  }
}

// exhaustive check of the 5-instruction division against the IEEE one over a float range
__global__ void check_fastdiv(float sc, float r, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long b = lo_bits + blockIdx.x * 256ULL + threadIdx.x; b <= hi_bits; b += gridDim.x * 256ULL) {
    const float a = __uint_as_float(static_cast<uint32_t>(b));
    const float q0 = a / sc;
    const float q1 = div_const<true>(a, sc, r);
    if (__float_as_uint(q0) != __float_as_uint(q1) && !(q0 != q0 && q1 != q1)) ++n;
  }
  if (n) atomicAdd(bad, n);
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 256;
  const int H = 384, W = 1248, CROP_H = 375, CROP_W = 1242;
  const long long n_img = 2LL * pairs;
  const int hw4 = H * W / 4;
  const long long per = 3LL * H * W;
  const long long elems = n_img * per;
  const size_t bytes = elems * 4;
  Sp sp = {{0.229f, 0.224f, 0.225f}, {0.485f, 0.456f, 0.406f}, {0, 0, 0}, {1, 1, 1}, {0, 0, 0}};
  for (int c = 0; c < 3; ++c) sp.rcp[c] = 1.0f / sp.scale[c];
  float *x, *g, *cl, *xo, *lut;
  uint8_t *u8, *idx_smooth, *idx_rand;
  int* ok;
  CK(hipMalloc(&x, bytes));
  CK(hipMalloc(&g, bytes));
  CK(hipMalloc(&cl, bytes));
  CK(hipMalloc(&xo, bytes));
  CK(hipMalloc(&u8, n_img * (size_t)CROP_H * W * 3));
  CK(hipMalloc(&idx_smooth, elems));
  CK(hipMalloc(&idx_rand, elems));
  CK(hipMalloc(&ok, n_img * sizeof(int)));
  CK(hipMalloc(&lut, 768 * 4));
  {
    std::vector<float> hl(768);
    for (int c = 0; c < 3; ++c)
      for (int v = 0; v < 256; ++v) hl[c * 256 + v] = clean_from_index(v, sp.scale[c], sp.shift[c]);
    CK(hipMemcpy(lut, hl.data(), 768 * 4, hipMemcpyHostToDevice));
    std::vector<int> hok(n_img, 1);
    CK(hipMemcpy(ok, hok.data(), n_img * sizeof(int), hipMemcpyHostToDevice));
    // two images worth of content, repeated: smooth 8-bit field (+-8 noise) and white noise, zero padding outside the crop
    std::vector<uint8_t> hs(2 * per), hr(2 * per);
    std::vector<float> hx(2 * per), hg(2 * per), hc(2 * per);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int im = 0; im < 2; ++im)
      for (int c = 0; c < 3; ++c)
        for (int y = 0; y < H; ++y)
          for (int xx = 0; xx < W; ++xx) {
            const long long i = (im * 3LL + c) * H * W + (long long)y * W + xx;
            const bool in = y < CROP_H && xx < CROP_W;
            int sm = (int)(128 + 100 * sinf(0.013f * xx + 0.7f * c + im) * cosf(0.021f * y)) + (int)(rnd() % 17) - 8;
            sm = sm < 0 ? 0 : (sm > 255 ? 255 : sm);
            hs[i] = in ? (uint8_t)sm : 0;
            hr[i] = in ? (uint8_t)(rnd() & 255) : 0;
            const float cv = in ? clean_from_index(hs[i], sp.scale[c], sp.shift[c]) : sp.shift[c];
            hc[i] = cv;
            hx[i] = (cv - sp.shift[c]) / sp.scale[c];
            hg[i] = ((int)(rnd() & 0xffff) - 32768) / 65536.0f;
          }
    for (long long i = 0; i < n_img; i += 2) {
      CK(hipMemcpy(x + i * per, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(g + i * per, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(cl + i * per, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(idx_smooth + i * per, hs.data(), hs.size(), hipMemcpyHostToDevice));
      CK(hipMemcpy(idx_rand + i * per, hr.data(), hr.size(), hipMemcpyHostToDevice));
    }
  }
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const float alpha = 1.0f / 255.0f, eps = 0.03f;
  const double alg = (double)n_img * (16.0 * 3 * H * W + 3.0 * CROP_H * CROP_W);
  const double real_idx = (double)n_img * (13.0 * 3 * H * W + 3.0 * CROP_H * W);
  const double real_f = (double)n_img * (16.0 * 3 * H * W + 3.0 * CROP_H * W);
  printf("pairs %d images %lld ; algorithmic %.3f GB, moved (index) %.3f GB, moved (float) %.3f GB\n", pairs, n_img, alg / 1e9, real_idx / 1e9, real_f / 1e9);

  {  // exhaustive check of the short division over the range y - shift can take: [-0.5, 0.6] (all finite floats in it)
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    for (int c = 0; c < 3; ++c) {
      unsigned long long tot = 0;
      // positive floats 0 .. 0.6 and negative floats -0 .. -0.5, plus the whole float range for information
      const uint32_t ranges[3][2] = {{0x00000000u, 0x3f19999au}, {0x80000000u, 0xbf000000u}, {0x00000000u, 0xffffffffu}};
      for (int k = 0; k < 3; ++k) {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(check_fastdiv, dim3(4096), dim3(256), 0, s, sp.scale[c], sp.rcp[c], ranges[k][0], ranges[k][1], bad);
        CK(hipMemcpyAsync(&tot, bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("fastdiv c%d range %d [%08x,%08x]: %llu mismatches\n", c, k, ranges[k][0], ranges[k][1], tot);
      }
    }
  }

#define RUN(MODE, PAD, BL, TR, FD, IDXP, TAG)                                                                                   \
  {                                                                                                                          \
    const int gx_ = (hw4 + BL * 2 * TR - 1) / (BL * 2 * TR);                                                                 \
    dim3 grid(gx_, (unsigned)n_img);                                                                                         \
    float ms = time_ms(s, 20, [&] {                                                                                          \
      hipLaunchKernelGGL((k1i<MODE, PAD, BL, TR, FD>), grid, dim3(BL), 0, s, (const v4f*)xin, (const v4f*)g, (const v4f*)cl, \
                         (const uint32_t*)(IDXP), ok, lut, (v4f*)xout, u8, n_img, hw4, W, CROP_H, CROP_H, CROP_W, sp, alpha, eps); \
    });                                                                                                                      \
    const double moved = MODE == M_FLOAT ? real_f : real_idx;                                                                \
    printf("%-6s mode%d pad%d block%3d trips%d fastdiv%d gx%4d : %7.3f ms  alg %7.1f GB/s  moved %7.1f GB/s (%.3f of 8 TB/s)\n", TAG, MODE, PAD, BL, TR, \
           FD, gx_, ms, alg / ms / 1e6, moved / ms / 1e6, moved / ms / 1e6 / 8000.0);                                        \
  }

  const int REPS = argc > 2 ? atoi(argv[2]) : 20;
#undef RUN
#define RUN(MODE, PAD, BL, TR, FD, IDXP, TAG)                                                                                \
  {                                                                                                                          \
    const int gx_ = (hw4 + BL * 2 * TR - 1) / (BL * 2 * TR);                                                                 \
    dim3 grid(gx_, (unsigned)n_img);                                                                                         \
    int flip = 0;                                                                                                            \
    float ms = time_ms(s, REPS, [&] {                                                                                        \
      const float* xin_ = bufmode == 2 ? (flip ? xo : x) : x;                                                                \
      float* xout_ = bufmode == 0 ? x : (bufmode == 1 ? xo : (flip ? x : xo));                                               \
      flip ^= 1;                                                                                                             \
      hipLaunchKernelGGL((k1i<MODE, PAD, BL, TR, FD>), grid, dim3(BL), 0, s, (const v4f*)xin_, (const v4f*)g, (const v4f*)cl, \
                         (const uint32_t*)(IDXP), ok, lut, (v4f*)xout_, u8, n_img, hw4, W, CROP_H, CROP_H, CROP_W, sp, alpha, eps); \
    });                                                                                                                      \
    const double moved = MODE == M_FLOAT ? real_f : real_idx;                                                                \
    printf("%-6s mode%d pad%d block%3d trips%d gx%4d reps%4d : %7.3f ms  alg %7.1f GB/s  moved %7.1f GB/s (%.3f of 8 TB/s)\n", TAG, MODE, PAD, BL, TR, \
           gx_, REPS, ms, alg / ms / 1e6, moved / ms / 1e6, moved / ms / 1e6 / 8000.0);                                      \
  }
  const char* names[3] = {"in place (x -> x)", "fixed out of place (x -> xo, x never written)", "alternating (x -> xo -> x ...)"};
  for (int bufmode = 0; bufmode < 3; ++bufmode) {
    printf("---- %s\n", names[bufmode]);
    for (int rep = 0; rep < 2; ++rep) {
      RUN(M_FLOAT, false, 64, 1, false, idx_smooth, "float");
      RUN(M_DIV, true, 64, 1, false, idx_smooth, "div");
      RUN(M_LDS, true, 64, 1, false, idx_smooth, "lds-s");
      RUN(M_LDS, true, 64, 1, false, idx_rand, "lds-r");
      RUN(M_GLB, true, 64, 1, false, idx_smooth, "glb-s");
    }
  }
  return 0;
}
