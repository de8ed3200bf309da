// Stand-alone tuning harness for the PGD-step kernel (K1).  Not part of the product: it times
// launch-shape / cache-policy variants of the same arithmetic on a KITTI-shaped resident batch and
// prints algorithmic GB/s, so that the winner can be folded into csrc/advengine.hip.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o k1_tune tools/k1_tune.hip && ./k1_tune [pairs]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(off)

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));

struct Sp {
  float scale[3], shift[3], lo[3], hi[3];
};

__device__ __forceinline__ float t_sign(float g) { return (g > 0.0f ? 1.0f : 0.0f) - (g < 0.0f ? 1.0f : 0.0f); }
__device__ __forceinline__ float t_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float elem(float x, float g, float cl, float sc, float sh, float lo, float hi, float alpha, float eps) {
  float d = x * sc;
  d = d + sh;
  const float a = d + alpha * t_sign(g);
  const float eta = t_clamp(a - cl, -eps, eps);
  const float y = t_clamp(cl + eta, lo, hi);
  return (y - sh) / sc;
}
__device__ __forceinline__ uint32_t byte_of(float xo, float sc, float sh) {
  float v = xo * sc;
  v = v + sh;
  v = v * 255.0f;
  if (!(fabsf(v) < 2147483648.0f)) return 0u;
  return static_cast<uint32_t>(static_cast<int>(v)) & 0xffu;
}

template <bool NT>
__device__ __forceinline__ v4f ld(const v4f* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st(v4f* p, v4f v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// grid.y over images, grid.x strides over pixel groups; UNROLL groups per lane per trip
template <bool NTL, bool NTS, int UNROLL, int BLOCK, bool U8, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void k1(const v4f* x, const v4f* __restrict__ g, const v4f* __restrict__ cl,
                                            v4f* xo, uint8_t* u8, long long n_img, int hw4, int w, int crop_h, Sp sp,
                                            float alpha, float eps) {
  for (long long img = blockIdx.y; img < n_img; img += gridDim.y) {
    const long long plane0 = img * 3LL * hw4;
    // CONTIG: a block owns UNROLL consecutive tiles of BLOCK groups; else tiles are grid-strided
    const int stride = CONTIG ? BLOCK : gridDim.x * BLOCK;
    const int first = CONTIG ? blockIdx.x * BLOCK * UNROLL + threadIdx.x : blockIdx.x * BLOCK + threadIdx.x;
    const int trip = CONTIG ? gridDim.x * BLOCK * UNROLL : stride * UNROLL;
    for (int q0 = first; q0 < hw4; q0 += trip) {
      v4f X[UNROLL][3], G[UNROLL][3], C[UNROLL][3];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const long long i = plane0 + (long long)c * hw4 + q;
            X[u][c] = ld<NTL>(x + i);
            G[u][c] = ld<NTL>(g + i);
            C[u][c] = ld<NTL>(cl + i);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int q = q0 + u * stride;
        if (q < hw4) {
          v4f O[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              O[c][j] = elem(X[u][c][j], G[u][c][j], C[u][c][j], sp.scale[c], sp.shift[c], sp.lo[c], sp.hi[c], alpha, eps);
            st<NTS>(xo + plane0 + (long long)c * hw4 + q, O[c]);
          }
          if (U8) {
            const int p = q * 4;
            const int row = p / w;
            if (row < crop_h) {
              const int col = p - row * w;
              uint32_t b[12];
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) b[j * 3 + c] = byte_of(O[c][j], sp.scale[c], sp.shift[c]);
              v3u r;
              r[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
              r[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
              r[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
              v3u* dst = reinterpret_cast<v3u*>(u8 + (img * crop_h + row) * (3LL * w) + col * 3LL);
              if (NTS)
                __builtin_nontemporal_store(r, dst);
              else
                *dst = r;
            }
          }
        }
      }
    }
  }
}

// calibration: plain float4 copy and 3-read-1-write stream with no arithmetic
template <bool NT>
__global__ __launch_bounds__(256) void copy_k(const v4f* __restrict__ a, v4f* b, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) st<NT>(b + i, ld<NT>(a + i));
}
template <bool NT>
__global__ __launch_bounds__(256) void r3w1_k(const v4f* __restrict__ a, const v4f* __restrict__ b, const v4f* __restrict__ c,
                                              v4f* o, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    v4f va = ld<NT>(a + i), vb = ld<NT>(b + i), vc = ld<NT>(c + i);
    st<NT>(o + i, va + vb + vc);
  }
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
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
  const int H = argc > 2 ? atoi(argv[2]) : 384, W = argc > 3 ? atoi(argv[3]) : 1248;
  const int CROP_H = argc > 2 ? H : 375, CROP_W = argc > 2 ? W : 1242;
  const bool u8ok = (W % 4) == 0;
  const long long n_img = 2LL * pairs;
  const int hw4 = H * W / 4;
  const long long elems = n_img * 3LL * H * W;
  const size_t bytes = elems * 4;
  float *x, *g, *cl, *xo;
  uint8_t* u8;
  CK(hipMalloc(&x, bytes));
  CK(hipMalloc(&g, bytes));
  CK(hipMalloc(&cl, bytes));
  CK(hipMalloc(&xo, bytes));
  CK(hipMalloc(&u8, n_img * (size_t)CROP_H * W * 3));
  {  // deterministic fill on the host once (values do not matter for time; avoid denormal/NaN paths)
    std::vector<float> h(3LL * H * W * 2);
    uint32_t s = 12345;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = ((s >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f;
    }
    for (long long i = 0; i < n_img; i += 2) {
      CK(hipMemcpy(x + i * 3LL * H * W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(g + i * 3LL * H * W, h.data() + 7, (h.size() - 7) * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(cl + i * 3LL * H * W, h.data() + 3, (h.size() - 3) * 4, hipMemcpyHostToDevice));
    }
  }
  hipStream_t s;
  CK(hipStreamCreate(&s));
  Sp sp = {{0.229f, 0.224f, 0.225f}, {0.485f, 0.456f, 0.406f}, {0, 0, 0}, {1, 1, 1}};
  const float alpha = 1.0f / 255.0f, eps = 0.03f;
  const double alg = (double)n_img * (16.0 * 3 * H * W + 3.0 * CROP_H * CROP_W);
  const double alg_nou8 = (double)n_img * 16.0 * 3 * H * W;
  printf("pairs %d  images %lld  working set %.2f GB  algorithmic bytes/launch %.3f GB\n", pairs, n_img, 4.0 * bytes / 1e9, alg / 1e9);

  const long long n4 = elems / 4;
  for (int gridk : {2048, 8192, 65536}) {
    float ms = time_ms(s, 10, [&] { hipLaunchKernelGGL(copy_k<false>, dim3(gridk), dim3(256), 0, s, (const v4f*)x, (v4f*)xo, n4); });
    printf("copy      grid %6d        : %8.3f ms  %7.1f GB/s (r+w)\n", gridk, ms, 2.0 * bytes / ms / 1e6);
    ms = time_ms(s, 10, [&] { hipLaunchKernelGGL(copy_k<true>, dim3(gridk), dim3(256), 0, s, (const v4f*)x, (v4f*)xo, n4); });
    printf("copy nt   grid %6d        : %8.3f ms  %7.1f GB/s (r+w)\n", gridk, ms, 2.0 * bytes / ms / 1e6);
    ms = time_ms(s, 10, [&] { hipLaunchKernelGGL(r3w1_k<false>, dim3(gridk), dim3(256), 0, s, (const v4f*)x, (const v4f*)g, (const v4f*)cl, (v4f*)xo, n4); });
    printf("r3w1      grid %6d        : %8.3f ms  %7.1f GB/s\n", gridk, ms, 4.0 * bytes / ms / 1e6);
    ms = time_ms(s, 10, [&] { hipLaunchKernelGGL(r3w1_k<true>, dim3(gridk), dim3(256), 0, s, (const v4f*)x, (const v4f*)g, (const v4f*)cl, (v4f*)xo, n4); });
    printf("r3w1 nt   grid %6d        : %8.3f ms  %7.1f GB/s\n", gridk, ms, 4.0 * bytes / ms / 1e6);
  }

#define RUN(NTL, NTS, UN, BL, U8F, CONTIG, GX)                                                                                     \
  {                                                                                                                        \
    int gx_ = (GX);                                                                                                        \
    int maxgx = (hw4 + BL * UN - 1) / (BL * UN);                                                                           \
    if (gx_ > maxgx || gx_ <= 0) gx_ = maxgx;                                                                               \
    dim3 grid(gx_, (unsigned)n_img);                                                                                       \
    float ms = time_ms(s, 10, [&] {                                                                                        \
      hipLaunchKernelGGL((k1<NTL, NTS, UN, BL, U8F, CONTIG>), grid, dim3(BL), 0, s, (const v4f*)x, (const v4f*)g, (const v4f*)cl, \
                         (v4f*)xo, u8, n_img, hw4, W, CROP_H, sp, alpha, eps);                                            \
    });                                                                                                                    \
    printf("k1 ntl%d nts%d unroll%d block%4d u8%d contig%d gx%4d : %8.3f ms  %7.1f GB/s algorithmic\n", NTL, NTS, UN, BL, U8F, CONTIG, gx_,  \
           ms, (U8F ? alg : alg_nou8) / ms / 1e6);                                                                         \
  }

  float* xo_sep = xo;
  if (!u8ok) {  // Stereo R-CNN-like shapes: no aligned u8 rows; study plane misalignment with u8 off
    for (int inplace = 0; inplace < 2; ++inplace) {
      xo = inplace ? x : xo_sep;
      printf("---- %s\n", inplace ? "in place (x_out == x)" : "out of place");
      RUN(false, false, 1, 64, false, false, 0);
      RUN(false, true, 1, 64, false, false, 0);
      RUN(true, true, 1, 64, false, false, 0);
      RUN(true, false, 1, 64, false, false, 0);
      RUN(false, false, 2, 64, false, false, 0);
      RUN(false, true, 2, 64, false, false, 0);
      RUN(true, true, 2, 64, false, false, 0);
      RUN(true, true, 2, 256, false, false, 0);
      RUN(false, false, 2, 256, false, false, 0);
    }
    return 0;
  }
  for (int inplace = 0; inplace < 2; ++inplace) {
    xo = inplace ? x : xo_sep;
    printf("---- %s\n", inplace ? "in place (x_out == x)" : "out of place");
    for (int rep = 0; rep < 2; ++rep) {
      RUN(false, true, 1, 256, true, false, 0);
      RUN(true, true, 1, 256, true, false, 0);
      RUN(false, true, 1, 128, true, false, 0);
      RUN(true, true, 1, 128, true, false, 0);
      RUN(false, true, 1, 64, true, false, 0);
      RUN(true, true, 1, 64, true, false, 0);
      RUN(false, false, 1, 64, true, false, 0);
      RUN(false, true, 2, 64, true, false, 0);
      RUN(true, true, 2, 64, true, false, 0);
      RUN(false, true, 2, 64, true, true, 0);
      RUN(true, true, 2, 64, true, true, 0);
      RUN(true, true, 4, 64, true, true, 0);
      RUN(true, true, 1, 64, false, false, 0);
    }
  }
  return 0;
}
