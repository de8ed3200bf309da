/*
 * advengine.h - C ABI of libadvengine.so, the MI355X (gfx950) perturbation engine.
 *
 * The reference (DexterJZ/eval_driving_safety) has no FFI: its hot path is inline
 * torch code inside four scripts.  Each entry point below replaces one such block and
 * cites it (paths relative to the reference root).  INTEGRATION.md shows the ctypes
 * stub a maintainer would add to the reference scripts to call them.
 *
 * Conventions (all entry points)
 *   - every image pointer is DEVICE memory owned by the caller, float32, contiguous
 *     NCHW with C = ADV_CHANNELS = 3;  `n` counts [3,H,W] images, so a stereo batch of
 *     B pairs is n = 2B (or two calls with n = B);
 *   - enqueue-only on `stream` (a hipStream_t; NULL = the null stream): no allocation,
 *     no synchronisation, no host read-back, safe to capture in a hipGraph;
 *   - no global state; safe to call from several host threads on different streams;
 *   - returns 0 or a negative ADV_E* code; nothing is launched when an error is
 *     returned;  adv_last_hip_error() gives the hipError_t behind ADV_ELAUNCH;
 *   - the entry points that replace reference code (a1-a13, a5) are bit-identical to the reference's torch-CPU
 *     arithmetic (float32, every operation rounded separately, true division, NaN-propagating clamp); the detector-side
 *     operators further down (upstream code, not in the reference tree) state their own contract: bit-identical to torch's
 *     CPU operator or to the oracle's fixed summation order where no transcendental is involved, 1e-5 relative otherwise;
 *   - the convolution entry points read the device's CU count once per host thread and raise the kernels' dynamic-LDS limit once
 *     per process (host-side calls outside any capture, nothing allocated, nothing synchronised); no environment variable is read.
 */
#ifndef ADVENGINE_H
#define ADVENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADV_ABI_VERSION 11
#define ADV_API __attribute__((visibility("default"))) /* the library is built with -fvisibility=hidden */
#define ADV_CHANNELS 3

#define ADV_OK 0
#define ADV_EINVAL (-22)  /* bad argument: null pointer, non-positive size, window outside the image ... */
#define ADV_EALIGN (-14)  /* a float pointer is not 4-byte aligned */
#define ADV_ELAUNCH (-5)  /* hipLaunchKernel failed; see adv_last_hip_error() */

typedef void* adv_stream_t; /* hipStream_t */

/* Pixel space of the stored image, per channel.
 *   ADV_SPACE_AFFINE    stored = (pixel - shift) / scale with pixel in [lo, hi]
 *                       (DSGN: ImageNet-normalised RGB, pixel in [0,1];
 *                        attack/DSGN/pgd_attack.py:153-154,196-207)
 *   ADV_SPACE_IDENTITY  stored = pixel, pixel in [lo, hi]
 *                       (Stereo R-CNN: BGR minus PIXEL_MEANS on the 0..255 scale;
 *                        attack/Stereo-RCNN/pgd_attack.py:189-207) */
/*   ADV_SPACE_AFFINE_RCP as AFFINE, but the re-normalisation is (pixel - shift) * (1.0f / scale): what torch's CUDA / ROCm
 *                       kernels compute for `tensor / python_scalar`: a multiplication by float(1.0 / b), the reciprocal taken in
 *                       double from the Python double b - export_add[c] must hold that double divisor (0.229, 0.224, 0.225).  The
 *                       reference's scripts call normalize() on CUDA tensors (attack/DSGN/pgd_attack.py:203-207,353-354), so
 *                       THIS kind reproduces a GPU run of the reference bit for bit, AFFINE a CPU run (the north_star's
 *                       "reference CPU path"); the two stay within a few ulp of each other. */
enum { ADV_SPACE_AFFINE = 0, ADV_SPACE_IDENTITY = 1, ADV_SPACE_AFFINE_RCP = 2 };

typedef struct adv_space {
  int32_t kind;
  float scale[3];
  float shift[3];
  float lo[3];
  float hi[3];
  double export_add[3]; /* IDENTITY only: added in float64 before the 8-bit conversion
                           (cfg.PIXEL_MEANS, attack/Stereo-RCNN/pgd_attack.py:236) */
} adv_space_t;

/* Fill `s` with the reference's constants. */
ADV_API void adv_space_dsgn(adv_space_t* s);  /* mean/std of attack/DSGN/pgd_attack.py:153-154, range [0,1] */
ADV_API void adv_space_dsgn_gpu_reference(adv_space_t* s); /* the same constants, kind ADV_SPACE_AFFINE_RCP */
ADV_API void adv_space_srcnn(adv_space_t* s); /* range [-m_c, 255-m_c], m = (102.9801,115.9465,122.7717) */

ADV_API int adv_abi_version(void);
/* 0 for the shipped library: no entry point reads the environment, kernel selection depends on the arguments alone.  1 for the
 * -DADV_TEST_HOOKS build (libadvengine_hooks.so, opened by tests/ and tools/ only), whose convolution / RoIAlign entry points read
 * the ADV_* route switches of DESIGN.md 5 at each launch. */
ADV_API int adv_build_has_test_hooks(void);
ADV_API const char* adv_strerror(int code);
ADV_API int adv_last_hip_error(void); /* thread-local */

/* a1  denormalize(), attack/DSGN/pgd_attack.py:196-200 :  out = x * scale_c + shift_c.
 *     (The reference touches batch element 0 only; here all n images.)  out may alias x.
 *     AFFINE spaces only. */
ADV_API int adv_denormalize_f32(const float* x, float* out, int64_t n, int h, int w,
                        const adv_space_t* space, adv_stream_t stream);

/* a2  normalize(), attack/DSGN/pgd_attack.py:203-207 :  out = (x - shift_c) / scale_c. */
ADV_API int adv_normalize_f32(const float* x, float* out, int64_t n, int h, int w,
                      const adv_space_t* space, adv_stream_t stream);

/* a3 / a13 (+ a5)  one PGD / FGSM step for n images in ONE pass over memory.
 *   AFFINE   attack/DSGN/pgd_attack.py:339-354
 *            d = x*scale+shift; a = d + alpha*sign(grad); eta = clamp(a - clean, +-eps);
 *            y = clamp(clean + eta, lo, hi); x_out = (y - shift)/scale
 *   IDENTITY attack/Stereo-RCNN/pgd_attack.py:177-217
 *            a = x + alpha*sign(grad); eta = clamp(a - clean, +-eps);
 *            x_out = clamp(clean + eta, lo_c, hi_c)            (eps already times 255, :57)
 *   x, grad, clean, x_out : [n,3,h,w].  `clean` is in pixel space (the DENORMALISED clean
 *   image for AFFINE, pgd_attack.py:297-298).  x_out may alias x (in place).
 *   sign(nan) = sign(+-0) = 0 as torch.sign; clamp propagates NaN as torch.clamp.
 *   u8_out (nullable): the 8-bit HWC image the reference writes to PNG for this iterate,
 *     AFFINE   tensor2im, attack/DSGN/pgd_attack.py:157-179: trunc(((x_out*scale+shift)*255))
 *     IDENTITY attack/Stereo-RCNN/pgd_attack.py:233-237: sat_u8(rint(float32(x_out + export_add)))
 *   rows [0,crop_h) and columns [0,crop_w) of image i go to
 *     u8_out + i*u8_image_stride + row*u8_row_stride + col*3      (strides in bytes).
 *   With u8_row_stride >= 3*w whole rows (all w columns) are stored, which keeps every
 *   store aligned; the columns >= crop_w are then padding the consumer crops away
 *   (save_img's crop, pgd_attack.py:192). */
ADV_API int adv_pgd_step_f32(const float* x, const float* grad, const float* clean, float* x_out,
                     uint8_t* u8_out, int64_t n, int h, int w, const adv_space_t* space,
                     float alpha, float eps, int crop_h, int crop_w, int64_t u8_row_stride,
                     int64_t u8_image_stride, adv_stream_t stream);

/* The clean image of an attack held as ONE BYTE per element.  A clean image is re-read by every PGD step, and it normally
 *     stems from 8-bit pixels v through
 *         t = v/255;  x0 = (t - shift)/scale;  clean = x0*scale + shift        (float32: ToTensor, Normalize, denormalize)
 *     zero-padded IN NORMALISED SPACE to the network size, so that outside the image's valid_h x valid_w corner the clean
 *     value is exactly shift_c.  IDENTITY spaces (Stereo R-CNN: x0 = v - mean_c, the clean pair is a clone of x0,
 *     attack/Stereo-RCNN/pgd_attack.py:122-123): v = rint(x0 + mean_c), two candidate tables - the subtraction in float32,
 *     A_c[v] = float(v) - float(mean_c), or in float64 rounded once, B_c[v] = float(double(v) - mean_c) (numpy's
 *     `im -= pixel_means`), mean_c = export_add[c]; the whole frame is image (valid = (h, w)), rows of any length.
 *     All pointers are DEVICE memory owned by the caller; nothing here is read by the host. */
typedef struct adv_clean_index {
  uint8_t* index;          /* [n,3,h,w]  v = rint(clean*255) inside the valid corner, 0 outside                       */
  int32_t* ok;             /* [n]        != 0: every element of image i verified bit for bit (bit 0: table A, bit 1:
                                          table B); 0 = the steps read its float clean image                           */
  float* lut;              /* [2][3*256] table A, table B (AFFINE: both = the chain above applied to v); 16-byte aligned */
  const int32_t* valid_hw; /* [n,2] = (valid_h, valid_w) per image, or NULL: the two fields below for every image     */
  int32_t valid_h, valid_w;
} adv_clean_index_t;

/* a1 + index (+ a5 for iterate 0): clean_out = x*scale+shift as adv_denormalize_f32, AND fill ci->lut, ci->index, ci->ok:
 *     ok[i] = 1 iff for every element of image i   clean == T_c[index]  (inside the valid corner)
 *                                                 clean == shift_c     (outside it)            BIT FOR BIT.
 *     u8_out (nullable) receives the 8-bit export of x (= iterate 0, attack/DSGN/pgd_attack.py:279-294) exactly as
 *     adv_export_u8_f32 would write it.  AFFINE spaces: requires w % 4 == 0 and 16-byte aligned images.
 *     IDENTITY spaces: clean_out = x (may be the same pointer), requires h*w % 4 == 0, 16-byte aligned images and
 *     ci->valid = (h, w); ok[i] = 1 or 3 (table A), 2 (table B only) or 0. */
ADV_API int adv_clean_index_build_f32(const float* x, float* clean_out, const adv_clean_index_t* ci, uint8_t* u8_out,
                                      int64_t n, int h, int w, const adv_space_t* space, int crop_h, int crop_w,
                                      int64_t u8_row_stride, int64_t u8_image_stride, adv_stream_t stream);

/* The loader's work on the device (upstream DSGN test-time transform: ToTensor, ImageNet Normalize, zero padding to the network
 *     frame - the values the attack scripts receive as imgL / imgR, attack/DSGN/pgd_attack.py:262-263):
 *         x_out [n,3,h,w] = ((v/255) - shift)/scale inside image i's valid_h x valid_w corner, 0 outside      (true float32 divisions)
 *     from 8-bit RGB pixels u8_hwc (byte (row, col, c) of image i at u8_hwc + i*u8_image_stride + row*u8_row_stride + 3*col + c; DEVICE),
 *     and optionally, in the same pass: clean_out = x_out*scale + shift (shift outside the corner) = what adv_denormalize_f32 would
 *     give, and *ci filled as adv_clean_index_build_f32 would fill it - by construction, ok[i] = 1, no verification pass (the
 *     index IS the pixels).  The valid corner comes from ci (per image or common) or, with ci == NULL, from valid_h / valid_w.
 *     A quarter of the PCIe bytes of a float upload and no conversion on the host.  AFFINE spaces; w % 4 == 0. */
ADV_API int adv_import_u8_f32(const uint8_t* u8_hwc, int64_t u8_row_stride, int64_t u8_image_stride, float* x_out, float* clean_out,
                              const adv_clean_index_t* ci, int valid_h, int valid_w, int64_t n, int h, int w,
                              const adv_space_t* space, adv_stream_t stream);

/* a3 with the indexed clean image: identical results to adv_pgd_step_f32.  Images with ok[i] != 0 (read on the device,
 *     no host round trip) are stepped reading the 1-byte index and an LDS copy of the table instead of the float32
 *     `clean` (13 instead of 16 bytes per element, no division); the others read `clean` as usual, in the same launch.
 *     Same requirements as adv_clean_index_build_f32, which must have filled *ci for this batch.  IDENTITY spaces use the
 *     index in the line-aligned kernel (planes that are not whole cache lines: the 600 x 1987 Stereo R-CNN frame, two
 *     images or more); other layouts step through adv_pgd_step_f32's kernels - the same bits either way. */
ADV_API int adv_pgd_step_indexed_f32(const float* x, const float* grad, const float* clean, const adv_clean_index_t* ci,
                                     float* x_out, uint8_t* u8_out, int64_t n, int h, int w, const adv_space_t* space,
                                     float alpha, float eps, int crop_h, int crop_w, int64_t u8_row_stride,
                                     int64_t u8_image_stride, adv_stream_t stream);

/* a5  the 8-bit export alone (iterate 0 = the clean image, pgd_attack.py:279-294). */
ADV_API int adv_export_u8_f32(const float* x, uint8_t* u8_out, int64_t n, int h, int w,
                      const adv_space_t* space, int crop_h, int crop_w, int64_t u8_row_stride,
                      int64_t u8_image_stride, adv_stream_t stream);

/* a7  generate_round_mask's mask, attack/DSGN/patch_attack.py:245-248: float32 [h,w],
 *     1.0 where (y-cy)^2 + (x-cx)^2 <= r^2 (equal to the reference's float64 sqrt test). */
ADV_API int adv_disc_mask_f32(float* mask_out, int h, int w, int cy, int cx, int r, adv_stream_t stream);

/* a8  patch paste, attack/DSGN/patch_attack.py:326-333,369-376 (Stereo R-CNN :178-185,221-230):
 *     img = (1-M)*img + M*pad0(patch), M the disc of radius r centred (cy,cx), patch [3,d,d],
 *     d = 2r+1.  Evaluated literally on the d x d bounding square; outside it the reference
 *     computes 1*img + 0*0, i.e. leaves img as it is (a -0.0 there would become +0.0 in the
 *     reference and stays -0.0 here - the only deviation).  In place.  One image. */
ADV_API int adv_patch_paste_f32(float* img, const float* patch, int h, int w, int d, int cy, int cx,
                        int r, adv_stream_t stream);

/* a8 batched: image i of [n,3,h,w] gets the patch at (centers[2i], centers[2i+1]) = (cy,cx);
 *     `centers` is DEVICE int32 [n,2].  Elements whose target falls outside the image are
 *     skipped. */
ADV_API int adv_patch_paste_batch_f32(float* img, const float* patch, int64_t n, int h, int w, int d,
                              const int32_t* centers, int r, adv_stream_t stream);

/* a11 / a12  the reference's per-image patch update in one launch,
 *     attack/DSGN/patch_attack.py:416-430, attack/Stereo-RCNN/patch_attack.py:257-281:
 *     delta = clamp(half_alpha * (gradL[win(cy,cxL)] + gradR[win(cy,cxR)]), +-eps);
 *     patch = patch - delta;  if lo/hi given (HOST float[3]):  patch_c = clamp(patch_c, lo_c, hi_c).
 *     win = the (2r+1)^2 bounding SQUARE (not the disc).  half_alpha = 0.5*alpha = 500 in the
 *     reference.  delta_out (nullable, [3,d,d]) receives delta. */
ADV_API int adv_patch_update_f32(float* patch, const float* grad_l, const float* grad_r, int h, int w,
                         int d, int cy, int cx_l, int cx_r, int r, float half_alpha, float eps,
                         const float* lo, const float* hi, float* delta_out, adv_stream_t stream);

/* data-parallel form of a11: the clamped deltas of n image pairs evaluated against one patch
 *     snapshot, summed in index order:  delta_out = d_0 + d_1 + ... + d_{n-1}  ([3,d,d]).
 *     centers: DEVICE int32 [n,3] = (cy, cxL, cxR).  This buffer is what the RCCL all-reduce
 *     of the universal-patch attack carries. */
ADV_API int adv_patch_delta_batch_f32(const float* grad_l, const float* grad_r, int64_t n, int h, int w,
                              int d, const int32_t* centers, int r, float half_alpha, float eps,
                              float* delta_out, adv_stream_t stream);

/* second half of a11/a12:  patch = patch - delta;  optional per-channel clamp (HOST float[3]). */
ADV_API int adv_patch_apply_f32(float* patch, const float* delta, int d, const float* lo,
                        const float* hi, adv_stream_t stream);

/* K7  plane-sweep cost-volume build of a DSGN-style detector (the op the reference reaches through
 *     `model(imgL, imgR, ...)`, attack/DSGN/pgd_attack.py:308; it lives in the upstream DSGN repository,
 *     not in the reference tree, so this follows the published algorithm - PSMNet-style concatenation
 *     volume with one integer disparity shift per depth plane - and its parity is pinned against the
 *     oracle only, NOT against upstream code).
 *       left, right : [B,C,H,W] float32 feature maps (1/4 resolution in DSGN: C=32, 96 x 312)
 *       shift       : DEVICE int32 [B,D], s >= 0 = disparity of depth plane d in feature pixels
 *                     (fu * baseline / depth_d / downsample, rounded by the caller)
 *       cost        : [B,2C,D,H,W];  for x >= s:  cost[b,c,d,y,x]   = left[b,c,y,x]
 *                                                  cost[b,C+c,d,y,x] = right[b,c,y,x-s]
 *                                    for x <  s:  both 0.
 *     Every element of `cost` is written (no memset needed). */
ADV_API int adv_psv_build_f32(const float* left, const float* right, const int32_t* shift, float* cost, int b,
                      int c, int d, int h, int w, adv_stream_t stream);

/* K7 backward: the exact adjoint.  grad_left[b,c,y,x]  = sum_d [x >= s_d]     grad_cost[b,c,d,y,x]
 *                                  grad_right[b,c,y,x] = sum_d [x + s_d < W]  grad_cost[b,C+c,d,y,x+s_d]
 *     summed over d = 0..D-1 in that order in float32 (reproducible; no atomics). */
ADV_API int adv_psv_build_bwd_f32(const float* grad_cost, const int32_t* shift, float* grad_left,
                          float* grad_right, int b, int c, int d, int h, int w, adv_stream_t stream);

/* K7, interpolating form: the per-plane disparity fu * baseline / depth_d / downsample is fractional for almost every
 *     plane, so `shift` is DEVICE float32 [B,D], sf >= 0.  With s0 = floor(sf), w1 = sf - s0, w0 = 1 - w1, sc = ceil(sf):
 *         for x >= sc:  cost[b,c,d,y,x]   = left[b,c,y,x]
 *                       cost[b,C+c,d,y,x] = w0 * right[b,c,y,x-s0] + w1 * right[b,c,y,x-s0-1]     (right zero-extended)
 *         for x <  sc:  both 0.
 *     The two neighbours come from the same LDS-staged row (no extra HBM traffic); w1 == 0 reproduces
 *     adv_psv_build_f32.  Like the integer form it follows the published construction and is UNPINNED against upstream. */
ADV_API int adv_psv_build_lerp_f32(const float* left, const float* right, const float* shift, float* cost, int b, int c,
                                   int d, int h, int w, adv_stream_t stream);

/* its exact adjoint (w.r.t. left and right; the shifts are constants of the attack), per plane
 *     w0 * g[x+s0] + w1 * g[x+s0+1], summed over d = 0..D-1 in that order in float32 (no atomics). */
ADV_API int adv_psv_build_lerp_bwd_f32(const float* grad_cost, const float* shift, float* grad_left, float* grad_right,
                                       int b, int c, int d, int h, int w, adv_stream_t stream);

/* ---- Stereo R-CNN RoI path natives.  The ops are `from model.roi_layers import ROIAlign, nms`
 *      (attack/Stereo-RCNN/stereo_rcnn.py:18,44-45,132-134; predict_and_save_pgd.py:26,300) - compiled
 *      extensions of the upstream Stereo R-CNN checkout, absent from the reference tree.  These follow the
 *      published maskrcnn-benchmark algorithms the upstream extension is built from; parity is pinned against
 *      the oracle only (UNPINNED against upstream). */

/* RoIAlign forward (legacy, non-"aligned" pixel model): rois [R,5] = (batch index, x1, y1, x2, y2) in image
 *     coordinates, scaled by spatial_scale; sampling_ratio <= 0 -> ceil(roi_size / pooled_size) samples per bin
 *     (the reference constructs ROIAlign(..., 1/16, 0) and passes the FPN level's scale per call);
 *     out [R,C,PH,PW] = mean of the bilinear samples of each bin.  A roi whose batch index is NEGATIVE is skipped: the forward leaves
 *     its rows of out untouched, the backward adds nothing for it - one roi list can so be handed to every level of a feature pyramid,
 *     each level owning some of the rois and all of them writing one output (stereo_rcnn.py:110-141 without compaction). */
ADV_API int adv_roi_align_fwd_f32(const float* feat, const float* rois, float* out, int b, int c, int h, int w, int r,
                          int ph, int pw, float spatial_scale, int sampling_ratio, adv_stream_t stream);

/* RoIAlign backward, deterministic: every element of grad_feat [B,C,H,W] is written (no memset needed) as the float32 sum,
 *     in ONE fixed order - roi index, sample row, sample column, tap 1..4 - of the contributions
 *     grad_out[r,c,ph,pw] * weight / count of the samples whose bilinear taps touch it.  No float atomics: two runs give
 *     the same bits, and the result equals the oracle's ordered sum bit for bit.
 *     With more than 1024 rois the roi indices are split into G = adv_roi_align_bwd_segments(r) = min(8, ceil(r / 512)) segments of
 *     ceil(r / G) consecutive indices; each segment is summed in the order above into its own copy of the map (in the workspace) and the
 *     copies are added in segment order: grad_feat = ((P_0 + P_1) + ...) + P_{G-1}.  The segments run in parallel, which cuts the serial
 *     chain of the tiles where the proposals cluster G-fold; the order is a function of r alone (the oracle takes G as a parameter).
 *     workspace: DEVICE,
 *     adv_roi_align_bwd_workspace_ints(b, c, h, w, r, ph, pw) 4-byte elements (per 8 x 32-pixel tile the ascending list of rois that
 *     reach it; a channel-last copy of grad_out, whose 32 channels of one bin are one coalesced load; for G > 1 the G copies of the map), starting on a 16-BYTE boundary
 *     (the copy is read with 16-byte loads; ADV_EALIGN otherwise).  ph * pw <= 1500 (the transposing pass stages 32 x ph*pw floats in
 *     LDS; ADV_EINVAL beyond - the reference's pooled grids are 7 x 7 and 14 x 14). */
ADV_API int adv_roi_align_bwd_segments(int r);
ADV_API int64_t adv_roi_align_bwd_workspace_ints(int b, int c, int h, int w, int r, int ph, int pw);
ADV_API int adv_roi_align_bwd_f32(const float* grad_out, const float* rois, float* grad_feat, int b, int c, int h, int w,
                          int r, int ph, int pw, float spatial_scale, int sampling_ratio, int32_t* workspace,
                          adv_stream_t stream);
/* The same backward for SEVERAL feature maps pooled with one roi list (a feature pyramid: stereo_rcnn.py:110-141): grad_out is transposed to
 *   channel-last once - gcl, adv_roi_gout_channel_last_floats(c, r, ph, pw) floats, 16-byte aligned - and handed to each level's launch
 *   (_cl: same arguments and workspace as adv_roi_align_bwd_f32 otherwise; the same bits). */
ADV_API int64_t adv_roi_gout_channel_last_floats(int c, int r, int ph, int pw);
ADV_API int adv_roi_gout_channel_last_f32(const float* grad_out, float* gcl, int c, int r, int ph, int pw, adv_stream_t stream);
ADV_API int adv_roi_align_bwd_cl_f32(const float* gcl, const float* rois, float* grad_feat, int b, int c, int h, int w, int r, int ph, int pw,
                                     float spatial_scale, int sampling_ratio, int32_t* workspace, adv_stream_t stream);

/* Greedy NMS over n boxes [n,4] = (x1,y1,x2,y2) ALREADY SORTED by descending score, legacy "+1" areas,
 *     suppress when IoU > thresh.  keep_out [n] int64 receives the kept indices in order, num_keep_out [1] int32
 *     their count (both DEVICE); workspace: DEVICE, n * ceil(n/64) uint64.  Deterministic (bit-exact indices). */
ADV_API int adv_nms_f32(const float* boxes, int n, float thresh, int64_t* keep_out, int32_t* num_keep_out,
                uint64_t* workspace, adv_stream_t stream);

/* Dense photometric box alignment - `dense_align.align_parallel` of the upstream Stereo R-CNN checkout, called at
 *     attack/Stereo-RCNN/predict_and_save_pgd.py:381 (absent from the reference tree: this follows the published algorithm,
 *     Stereo R-CNN sec. 5, and is pinned against the oracle only).  For object b and candidate k
 *         z   = z_center[b] + (k - (K-1)/2) * step
 *         cost[b,k] = mean over the pixels (u,v) of roi[b] = (u0, v0, u1, v1) (half-open, network-scale pixels) of
 *                     sum_c ( left[c,v,u] - lerp(right[c,v,.], u - fb / (z + dz[b, u-u0])) )^2
 *     pixels whose right-image column falls outside [0, w-1) or whose depth is <= 0 are skipped; a candidate keeping fewer
 *     than a quarter of the region is +inf.  left/right [3,h,w] float32, roi DEVICE int32 [n,4], dz DEVICE [n,dz_stride]
 *     (depth offset of every pixel column from the box centre, metres), z_center DEVICE [n], fb = f * baseline * scale.
 *     Summation order is fixed (lane-strided partial sums, 64-lane shuffle tree, then the four waves in order). */
ADV_API int adv_dense_align_cost_f32(const float* left, const float* right, int h, int w, int n, const int32_t* roi,
                                     const float* dz, int dz_stride, const float* z_center, float fb, float step, int k,
                                     float* cost_out, adv_stream_t stream);

/* The depth of the cheapest candidate per object (first minimum; inf/NaN never win; no finite candidate: z_center is kept
 *     and cost_min_out = +inf).  Chain: coarse cost -> argmin -> fine cost around it -> argmin, all enqueue-only. */
ADV_API int adv_dense_align_argmin_f32(const float* cost, int n, int k, const float* z_center, float step, float* z_out,
                                       float* cost_min_out, adv_stream_t stream);

/* ---- dense 3x3x3 convolution on the matrix cores (float32 MFMA), the contraction a plane-sweep detector applies to
 *      the K7 cost volume (DSGN's 3D hourglass, reached through attack/DSGN/pgd_attack.py:308; upstream code - the
 *      semantics here are those of torch.nn.functional.conv3d(stride 1, padding 1, no bias); floating point: parity
 *      within 1e-4 relative of a float32 reference, and bit-exact against the oracle's k-ordered fmaf chain). */

/* Re-layout conv weights [Cout,Cin,3,3,3] for the kernel: w_prep [27][Cin'][32*ceil(Cout'/32)] (zero padded).
 *     transpose = 0: Cin' = cin, Cout' = cout                       (forward)
 *     transpose = 1: Cin' = cout, Cout' = cin, taps flipped        (backward w.r.t. the input: the adjoint conv) */
ADV_API int adv_conv3d_k3_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose,
                                           adv_stream_t stream);

/* y [B,Cout,D,H,W] = conv3d(x [B,Cin,D,H,W], w_prep), stride 1, zero padding 1; relu != 0 fuses max(y, 0).
 *     Cin must be a multiple of 4, or 1..3.  Implicit GEMM on v_mfma_f32_32x32x2_f32: M = 32 output channels, N = 32
 *     consecutive voxels along W, K = (input channel, tap); input tile + halo and the weights staged in LDS.
 *     Narrow layers leave the matrix cores: cout <= 8 (a network's last "score" layer, 32 -> 1) and cin <= 3 (its adjoint)
 *     run as fmaf chains on the vector ALUs in the SAME accumulation order - bit-identical to the padded matrix kernel. */
ADV_API int adv_conv3d_k3_f32(const float* x, const float* w_prep, float* y, int b, int cin, int cout, int d, int h,
                              int w, int relu, adv_stream_t stream);

/* y = conv3d(x, w_prep) zeroed where mask <= 0 (mask laid out like y, must not be y): the backward w.r.t. the input of a layer whose
 *     input is a ReLU output with that layer as its ONLY consumer - call it with x = grad_out, w_prep = the transposed weights and
 *     mask = the layer's input, and the result is the gradient w.r.t. the producer's pre-activation: no relu-backward pass over the
 *     volume.  Main matrix kernel only: cin % 4 == 0, cout > 8, 16-byte aligned w_prep - anything else returns ADV_EINVAL (callers
 *     then mask with adv_relu_backward_f32). */
ADV_API int adv_conv3d_k3_masked_f32(const float* x, const float* w_prep, const float* mask, float* y, int b, int cin, int cout, int d,
                                     int h, int w, adv_stream_t stream);

/* The same kernel family with what a plane-sweep detector's 3D hourglass needs around the plain convolution:
 *     bias       DEVICE [cout] or NULL, added after the accumulation (a folded batch-norm shift; its scale goes into the weights)
 *     residual   DEVICE tensor laid out like y (same dims, read where the result is written) or NULL: added after the bias and
 *                before the ReLU - an hourglass's skip connection, y = relu(conv(x) + bias + skip), without a pass of its own.
 *                Must not be y itself (-EINVAL).  Same float operations in the same order as conv, add, add, max done apart.
 *     stride     1, or 2 = the strided 3x3x3 convolution (padding 1): output grid ceil(d/2) x ceil(h/2) x ceil(w/2)
 *                (w % 4 == 0, x and w_prep 16-byte aligned, no masks / lattice: the direct strided matrix kernel, 0.52 of the
 *                float32 matrix peak on 32 -> 64 at the cost-volume size; otherwise a scalar-staging kernel, or the space-to-depth route below)
 *     tap_mask   bit t set = tap t = kd*9 + kh*3 + kw takes part (0x7ffffff = all)
 *     out_dims / out_stride / out_offset (HOST int32[3] each, or all NULL): result voxel i of the convolution's own grid is
 *                written to y[.., i*out_stride + out_offset] of a [b,cout,out_dims] tensor.
 *     A transposed convolution (kernel 3, stride 2, padding 1, output_padding 1 - the adjoint of the strided one, and the
 *     hourglass's up-sampling layer) is EIGHT such calls, one per output parity class (pd,ph,pw): stride 1, the class's 1-8
 *     taps in tap_mask, out_stride 2, out_offset (pd,ph,pw) - every tap of every class is used exactly once, so the matrix
 *     cores do 27 multiply-adds per INPUT voxel, not 27 per output voxel (ops.conv_transpose3d_k3_s2 prepares the classes).
 *     class_masks (HOST uint32[8] or NULL) + class_channels: input channels [k*class_channels, (k+1)*class_channels) use
 *                class_masks[k] instead of tap_mask (cin must be 8*class_channels).  With adv_space_to_depth2_f32 this is the
 *                FAST strided convolution: conv(x, stride 2) == conv(space_to_depth2(x), stride 1) where parity sub-volume p
 *                keeps only the taps its parity allows (27 taps over the eight sub-volumes), on the tuned stride-1 kernel. */
ADV_API int adv_conv3d_k3_ex_f32(const float* x, const float* w_prep, const float* bias, const float* residual, float* y, int b,
                                 int cin, int cout, int d, int h, int w, int stride, int relu, uint32_t tap_mask, const uint32_t* class_masks,
                                 int class_channels, const int32_t* out_dims, const int32_t* out_stride,
                                 const int32_t* out_offset, adv_stream_t stream);

/* y [b,cout,2d,2h,2w] = conv_transpose3d(x [b,cin,d,h,w], kernel 3, stride 2, padding 1, output_padding 1) (+ bias, + residual
 *     [b,cout,2d,2h,2w] or NULL, + ReLU - as in adv_conv3d_k3_ex_f32) in ONE
 *     launch: the eight output parity classes k = (pd*2+ph)*2+pw as one tile index of the persistent masked kernel, class k with
 *     its own prepared weights w_prep_classes[k] (HOST array of 8 DEVICE pointers) and tap mask tap_masks[k] (HOST uint32[8]) -
 *     ops.conv_transpose3d_k3_s2_prep builds both.  A workgroup stages an input tile once and runs all 27 kernel
 *     taps on it, each into the accumulator of the class it feeds (0.61 of the float32 matrix peak on 64 -> 32 at the half-resolution
 *     cost-volume size; LDS-DMA staging with w % 4 == 0 and a 16-byte aligned x, register staging for any other width); with y or
 *     residual not 8-byte aligned the classes' tiles (1 to 8 taps each) are interleaved over the workgroups of one launch.  Same bits
 *     as eight adv_conv3d_k3_ex_f32 calls with out_stride 2 / out_offset (pd,ph,pw). */
ADV_API int adv_conv_transpose3d_k3_s2_f32(const float* x, const float* const* w_prep_classes, const uint32_t* tap_masks,
                                           const float* bias, const float* residual, float* y, int b, int cin, int cout, int d,
                                           int h, int w, int relu, adv_stream_t stream);

/* <round 4> The same transposed convolution as the BACKWARD of a strided convolution (kernel 3, stride 2, padding 1) whose input x_in is
 *     a ReLU output with two consumers - that convolution and one skip path (an hourglass: the layer before a down-sampling layer also feeds
 *     the matching up-sampling layer's skip connection):  y = (conv_transpose(grad) + residual) zeroed where mask <= 0, with residual = the
 *     gradient arriving over the skip path (or NULL) and mask = x_in (laid out like y, must not be y) - i.e. the gradient w.r.t. the
 *     producer's PRE-activation, with neither an addition pass nor a relu-backward pass over the volume.  Same float operations in the same
 *     order as the three done apart.  Only the all-classes kernel applies the mask (y / residual / mask 8-byte aligned, w >= 4):
 *     anything else returns ADV_EINVAL and the caller does the two passes itself. */
ADV_API int adv_conv_transpose3d_k3_s2_dgrad_f32(const float* x, const float* const* w_prep_classes, const uint32_t* tap_masks,
                                                 const float* residual, const float* mask, float* y, int b, int cin, int cout, int d,
                                                 int h, int w, adv_stream_t stream);

/* How many input channels a STAGE of the plain strided convolution (adv_conv3d_k3_ex_f32, stride 2, every tap, own output grid,
 *     16-byte aligned w_prep) holds for this input pointer, output-channel count and width - the float32 accumulation order is
 *     (stage, tap, channel within the stage): 2 = the direct strided matrix kernel (w % 4 == 0, x 16-byte aligned: 54 MFMAs
 *     per wave and stage on the raw input tile, operands read from LDS at stride 2, no permuted copy), 4 = the scalar-staging kernel.
 *     Host-side, no launch.  Results of the two orders differ in the last bits only; the oracle takes the stage size as a parameter. */
ADV_API int adv_conv3d_k3_s2_stage_channels(const float* x, int cout, int w);

/* xs [b, 8c, ceil(d/2), ceil(h/2), ceil(w/2)]:  xs[b, p*c + ch, jd, jh, jw] = x[b, ch, 2jd+pd, 2jh+ph, 2jw+pw], p = (pd*2+ph)*2+pw,
 *     zero beyond the input.  HBM-bound permute (one pass); xs is caller-owned workspace. */
ADV_API int adv_space_to_depth2_f32(const float* x, float* xs, int b, int c, int d, int h, int w, adv_stream_t stream);

/* ---- after the 3D convolutions: what a plane-sweep detector does with its cost volume (upstream DSGN code reached through
 *      attack/DSGN/pgd_attack.py:308 "outputs = model(...)" and :324 "RPN3DLoss"; SURVEY 2.2 lists them as native ops the
 *      build must supply: trilinear upsample, grid_sample PSV -> 3D geometric volume, sigmoid focal loss).  UNPINNED against
 *      DSGN itself (not in the reference tree); pinned against torch's own operators (F.interpolate + softmax, F.grid_sample)
 *      on the CPU and against the oracle. */

/* depth_out [b,h_out,w_out] = sum_k softmax_k(U)[k] * depth_values[k],  U = trilinear up-sampling of cost [b,d,h,w] to
 *     [d_out,h_out,w_out] (torch.nn.functional.interpolate(mode="trilinear", align_corners)): the depth-regression head of a
 *     cost-volume network, FUSED - U and its softmax are never written (368 MB per image at the DSGN size); a lane owns one
 *     output pixel and streams over the planes.  depth_values DEVICE [d_out].  stats_out [b,2,h_out,w_out] (nullable): the
 *     per-pixel softmax maximum and denominator, which the backward reads.  exp is the device's: 1e-5 relative parity. */
ADV_API int adv_depth_regress_f32(const float* cost, const float* depth_values, float* depth_out, float* stats_out, int b, int d,
                                  int h, int w, int d_out, int h_out, int w_out, int align_corners, adv_stream_t stream);

/* grad_cost [b,d,h,w] from grad_depth [b,h_out,w_out].  Two atomic-free stages: per pixel the depth-axis adjoint into
 *     workspace [b,d,h_out,w_out] floats (caller-owned), then the bilinear adjoint as a gather per cost cell (rows, then
 *     columns, ascending): deterministic float32. */
ADV_API int adv_depth_regress_bwd_f32(const float* cost, const float* depth_values, const float* depth, const float* stats,
                                      const float* grad_depth, float* workspace, float* grad_cost, int b, int d, int h, int w,
                                      int d_out, int h_out, int w_out, int align_corners, adv_stream_t stream);

/* out [b,c,zo,yo,xo] = torch.nn.functional.grid_sample(vol [b,c,d,h,w], grid [b,zo,yo,xo,3], mode="bilinear",
 *     padding_mode="zeros", align_corners): aten's grid_sampler_3d arithmetic (corner order, weight products, skipped
 *     out-of-range corners) - bit-identical to torch on the CPU.  grid[...,0] indexes w, 1 h, 2 d, in [-1,1]. */
ADV_API int adv_grid_sample3d_f32(const float* vol, const float* grid, float* out, int b, int c, int d, int h, int w, int zo, int yo,
                                  int xo, int align_corners, adv_stream_t stream);

/* The gradient w.r.t. vol as a deterministic GATHER.  adv_grid_sample3d_plan_f32 builds, once per grid (the grid depends on
 *     the camera calibration only), for every cell of vol the list of (output voxel, weight) that sample it, sorted by output
 *     voxel, into plan (DEVICE, adv_grid_sample3d_plan_bytes bytes, caller-owned; 0 = arguments out of range).
 *     adv_grid_sample3d_bwd_f32 then sums grad_out * weight over each list in list order: no float atomics, the same bits on
 *     every run (torch's backward scatters with atomicAdd). */
ADV_API int64_t adv_grid_sample3d_plan_bytes(int b, int d, int h, int w, int zo, int yo, int xo);
ADV_API int adv_grid_sample3d_plan_f32(const float* grid, void* plan, int b, int d, int h, int w, int zo, int yo, int xo,
                                       int align_corners, adv_stream_t stream);
ADV_API int adv_grid_sample3d_bwd_f32(const float* grad_out, const void* plan, float* grad_vol, int b, int c, int d, int h, int w,
                                      int zo, int yo, int xo, adv_stream_t stream);

/* The same gradient through a channels-last copy of grad_out in workspace (DEVICE, adv_grid_sample3d_bwd_workspace_floats floats,
 *     caller-owned): a list entry then reads ONE run of c floats instead of c cache lines (PMC at the DSGN size: 3.0 GB fetched per
 *     launch by the function above for a 149 MB gradient volume).  Same bits.  workspace == NULL falls back to the function above. */
ADV_API int64_t adv_grid_sample3d_bwd_workspace_floats(int b, int c, int zo, int yo, int xo);
ADV_API int adv_grid_sample3d_bwd_ws_f32(const float* grad_out, const void* plan, float* grad_vol, float* workspace, int b, int c, int d,
                                         int h, int w, int zo, int yo, int xo, adv_stream_t stream);

/* Sigmoid focal loss, element-wise (maskrcnn-benchmark SigmoidFocalLoss semantics: the classification term of an FCOS-style
 *     head).  logits [n,k]; targets DEVICE int32 [n]: 0 = background, c in 1..k = class (column c-1), < 0 = ignored.
 *     loss_out / grad_out [n,k] (either may be NULL): per-element loss and its derivative w.r.t. the logit. */
ADV_API int adv_sigmoid_focal_loss_f32(const float* logits, const int32_t* targets, float* loss_out, float* grad_out, int64_t n,
                                       int k, float gamma, float alpha, adv_stream_t stream);

/* Bird's-eye-view fold (DSGN: F.avg_pool3d(v, (1, pool, 1)).permute(0, 1, 3, 2, 4).reshape(B, C * Y/pool, Z, X), reached at
 *     attack/DSGN/pgd_attack.py:308): v [b,c,z,y,x] -> out [b, c * (y / pool), z, x], out[b, ch*Yp + yy, z, x] = the float32 sum of
 *     v[b, ch, z, pool*yy + k, x] over k ascending, divided by pool (rows past Yp * pool dropped, as avg_pool3d floors).  One pass
 *     instead of a pooling kernel and a permuting copy; _bwd writes every element of grad_v (grad_out / pool, zero in dropped rows); its
 *     mask (laid out like v, or NULL) = the forward's input when that is a ReLU output consumed by the fold alone: grad_v is zeroed where
 *     mask <= 0, i.e. it is the gradient w.r.t. the producer's pre-activation. */
ADV_API int adv_bev_fold_f32(const float* v, float* out, int b, int c, int z, int y, int x, int pool, adv_stream_t stream);
ADV_API int adv_bev_fold_bwd_f32(const float* grad_out, const float* mask, float* grad_v, int b, int c, int z, int y, int x, int pool,
                                 adv_stream_t stream);

/* Bilinear up-sampling of a pyramid level, align_corners = False (the FPN top-down path's _upsample_add,
 *     attack/Stereo-RCNN/stereo_rcnn.py:92-108: F.upsample(x, size=(H, W), mode='bilinear') + y; also DSGN's pooled feature branches):
 *     x [nc,h,w] -> out [nc,ho,wo] for any size pair, source coordinate max(fmaf(in / out, o + 0.5, -0.5), 0) in float32.
 *     _bwd is its adjoint as a GATHER - every grad_in element is a sum in a fixed order (output rows, then columns, ascending; zero
 *     weights skipped), so the result is reproducible bit for bit, where a scattering backward with atomics is not. */
ADV_API int adv_bilinear_up_f32(const float* x, float* out, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream);
ADV_API int adv_bilinear_up_bwd_f32(const float* grad_out, float* grad_in, int64_t nc, int h, int w, int ho, int wo, adv_stream_t stream);

/* ---- 2D convolutions of the detectors' backbones on the matrix cores (float32 MFMA).  Upstream code reached through
 *      attack/Stereo-RCNN/pgd_attack.py:156 (ResNet-101-FPN: attack/Stereo-RCNN/stereo_rcnn.py:157-187) and attack/DSGN/pgd_attack.py:308
 *      (DSGN's 2D feature extractor and bird's-eye-view head); semantics = torch.nn.functional.conv2d; bit-exact against the
 *      oracle's ci-ordered fmaf chain, within 1e-4 relative of torch's float32 result. */

/* 1x1 / stride 1 convolution = one GEMM per image, Y_b [cout][pixels] = W [cout][cin] . X_b [cin][pixels] (NCHW as it stands).
 *     w_prep   from adv_conv2d_1x1_prep_weights_f32: [cin'][cout'] zero padded to multiples of 16 x 128
 *              (adv_conv2d_1x1_prep_floats floats).  transpose = 1 prepares W^T: the SAME entry point then computes the backward
 *              w.r.t. the input - call it with x = grad_out, cin = the layer's cout, cout = the layer's cin.
 *     bias     DEVICE [cout] or NULL (a folded batch-norm shift), added after the accumulation;
 *     residual DEVICE tensor laid out like y or NULL, added after the bias (a bottleneck's skip connection; in a backward the
 *              gradient arriving over the skip path);
 *     relu     != 0: max(., 0) after that;
 *     mask     DEVICE tensor laid out like y or NULL: the result is zeroed where mask <= 0, last of all.  In a backward call with
 *              mask = the layer's own input (a ReLU output) the result is the gradient w.r.t. the previous layer's PRE-activation:
 *              no separate ReLU-backward pass over the tensor;
 *     tile     -1 = pick by size; 0..5 force 128x256 / 128x128 / 64x128 / 64x64 / 64x32 / 64x32 in four half-size waves (output channels x
 *              pixels per workgroup; 5 = 16x16x4 matrix instructions: finer whole-wave rounding on few-workgroup layers) - the
 *              result does not depend on it (one fmaf chain per output element, ci ascending).
 *     x, y, residual, mask need 4-byte alignment only (rows of odd length are the rule: 150 x 497, 38 x 125 ...). */
ADV_API int64_t adv_conv2d_1x1_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv2d_1x1_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv2d_1x1_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                               float* y, int b, int cin, int cout, int64_t pixels, int relu, int tile, adv_stream_t stream);

/* 3x3 / stride 1 / padding = dilation (1 or 2) convolution - the residual blocks, FPN smoothing, RPN and head towers of the detectors.
 *     Implicit GEMM on v_mfma_f32_32x32x2_f32: a workgroup owns 8 rows x 32 columns x 64 output channels (or 16 x 32 x 32 for
 *     layers of 32 channels or fewer); per stage of 8 input channels the input tile with its halo and the weights [9][8][64] are
 *     staged in LDS; accumulation order (stage of 8 channels, tap ascending, channel ascending), one fmaf per product.
 *     w_prep from adv_conv2d_3x3_prep_weights_f32 ([9][cin'][cout'], zero padded to multiples of 8 x 64); transpose = 1 prepares the
 *     flipped W^T: the same entry point then computes the backward w.r.t. the input (x = grad_out, cin <-> cout swapped).
 *     bias / residual / relu / mask: as adv_conv2d_1x1_f32.  tile: -1 = by cout and map size, 0 = 8x32x64, 1 = 16x32x32, 2 = 4x32x64 (rows x columns x channels per
 *     workgroup; same result). */
ADV_API int64_t adv_conv2d_3x3_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv2d_3x3_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv2d_3x3_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                               float* y, int b, int cin, int cout, int h, int w, int dilation, int relu, int tile, adv_stream_t stream);

/* The same 3x3 / stride 1 / padding 1 convolution by Winograd F(2x2, 3x3): 2.25x fewer multiply-adds; input transform, the sixteen
 *     element-wise products (v_mfma_f32_16x16x4_f32, a workgroup owns 8 rows x 32 columns x 64 channels) and output transform in ONE
 *     kernel.  A different order of float operations than adv_conv2d_3x3_f32 (results agree to float32 rounding, ~1e-6 relative; the
 *     oracle's orc_conv2d_wino restates this order bit for bit): the caller chooses the route explicitly.
 *     w_prep from adv_conv2d_wino_prep_weights_f32 ([16][cin'][cout'] = G g G^T per channel pair, zero padded to multiples of 8 x 64);
 *     transpose = 1: the backward w.r.t. the input.  bias / residual / relu / mask: as adv_conv2d_1x1_f32.
 *     tile: -1 = by map size and cout, 0 = 8 x 32 outputs x 64 channels per workgroup, 1 = 16 x 16 x 64, 2 = 8 x 32 x 32 channels (256
 *     threads, two workgroups per CU), 3 = 16 x 16 x 32, 4 = 10 x 24 x 64, 5 = 10 x 24 x 32, 6 = 6 x 40 x 64, 7 = 6 x 40 x 32 (same result).  Tensors of fewer than four floats: ADV_EINVAL. */
ADV_API int64_t adv_conv2d_wino_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv2d_wino_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv2d_wino_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                                float* y, int b, int cin, int cout, int h, int w, int relu, int tile, adv_stream_t stream);

/* 3x3x3 / stride 1 / padding 1 convolution (the cost-volume and hourglass layers) by the same Winograd kernel: the transform is applied
 *     in the (H, W) plane, the three depth taps are part of the contraction (output plane od reads input planes od - 1, od, od + 1; a
 *     plane outside the volume is skipped) - 48 multiply-adds per 2 x 2 outputs instead of 108.  x [B,cin,D,H,W], y [B,cout,D,H,W];
 *     w_prep from adv_conv3d_wino_prep_weights_f32 (w [cout][cin][3][3][3] -> [16][3][cin'][cout'], padded to multiples of 8 x 64;
 *     transpose = 1: all taps reversed, the backward w.r.t. the input).  Its own order of float operations (oracle: orc_conv3d_wino);
 *     agrees with adv_conv3d_k3_f32 to float32 rounding.  bias / residual / relu / mask / tile: as adv_conv2d_wino_f32.  b * d <= 65535. */
ADV_API int64_t adv_conv3d_wino_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv3d_wino_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv3d_wino_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                                float* y, int b, int cin, int cout, int d, int h, int w, int relu, int tile, adv_stream_t stream);

/* The same 3x3 / stride 1 / padding 1 convolutions (2D, and 3x3x3 with the depth taps inside the contraction) by Winograd F(4x4, 3x3):
 *     36 element-wise products per 4 x 4 outputs - 4x fewer multiply-adds than the direct kernels, 1.78x fewer than F(2x2, 3x3) - on
 *     v_mfma_f32_32x32x2_f32; a workgroup of eight waves owns 16 x 32 (or 8 x 64) outputs x 64 channels, each wave nine 32 x 32 accumulators (four
 *     whole positions of the 36 and one block of a fifth), the input transform, products, output transform (positions exchanged through LDS) and epilogue in ONE kernel
 *     (csrc/wino4.hip).  Its own order of float operations (oracle: orc_conv_wino4, bit for bit); against the direct kernels and torch the
 *     results agree to ~1e-5 of the output's magnitude (the transform's constants reach 8 and 1/24: about ten times F(2x2, 3x3)'s error,
 *     inside the 1e-4 band) - the caller chooses the route explicitly.
 *     w_prep from adv_conv{2,3}d_wino4_prep_weights_f32 (G g G^T as [36][taps * cin' / 4][cout' / 64][lane][4]: the float4 a lane of the
 *     kernel loads per position and four input channels; zero padded to multiples of 8 x 64 - an opaque layout, sized by *_prep_floats; transpose
 *     = 1: the backward w.r.t. the input).  bias / residual / relu / mask: as adv_conv2d_1x1_f32.  tile: -1 = by map size and cout, 0 = 16 x 32
 *     outputs x 64 channels per workgroup, 1 = 8 x 64 x 64, 2 = 32 x 32 outputs x 32 channels, 3 = 16 x 64 x 32, 4 (2D, w <= 15, cin % 8 == 0) = two images per 16 x 32 tile (same result).  An image (cin x d x h x w floats) and the prepared weights must stay below 4 GiB / 2 GiB. */
ADV_API int64_t adv_conv2d_wino4_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv2d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv2d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                                 float* y, int b, int cin, int cout, int h, int w, int relu, int tile, adv_stream_t stream);
/* The same layer with the contraction dealt to `splits` workgroups per tile ("K-split"), for maps too small to fill the chip with one
 * workgroup per tile (a 256 -> 256 layer on 2 x 38 x 125 is 80 tiles for 256 compute units): part p multiplies input channels
 * [p * chunk, (p + 1) * chunk) and writes its raw F(4x4,3x3) outputs to plane p of `scratch` ([splits][b][cout][h][w] floats, 16-byte
 * aligned, the caller's); a second kernel adds the planes in order ((p0 + p1) + p2 ...), then bias, residual, ReLU, mask as above -
 * reproducible bits, the oracle's orc_conv_wino4 with `chunk`.  chunk = adv_conv2d_wino4_ksplit_chunk(cin, tile, splits) (whole stages
 * of the tile: tile must be 0..3 there); adv_conv2d_wino4_ksplit_pick: the number of parts this library would use for the layer
 * (1: do not split).  tile -1: chosen for b * splits workgroups per tile position (then read the chunk back through the oracle-side
 * helper with the same rule - tests pass explicit tiles).  No reference counterpart (the reference has no native code). */
ADV_API int adv_conv2d_wino4_ksplit_pick(int b, int cin, int cout, int h, int w);
ADV_API int adv_conv2d_wino4_ksplit_chunk(int cin, int tile, int splits);
ADV_API int adv_conv2d_wino4_ksplit_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                                        float* y, float* scratch, int b, int cin, int cout, int h, int w, int relu, int tile, int splits,
                                        adv_stream_t stream);
ADV_API int64_t adv_conv3d_wino4_prep_floats(int cout, int cin, int transpose);
ADV_API int adv_conv3d_wino4_prep_weights_f32(const float* w, float* w_prep, int cout, int cin, int transpose, adv_stream_t stream);
ADV_API int adv_conv3d_wino4_f32(const float* x, const float* w_prep, const float* bias, const float* residual, const float* mask,
                                 float* y, int b, int cin, int cout, int d, int h, int w, int relu, int tile, adv_stream_t stream);

/* y [planes = b*c][hw] <- [relu](y + bias[plane % c] + residual), in place, one pass: the epilogue of a convolution computed by another
 *     library (bias / residual NULL = skipped; residual laid out like y, must not be y).  planes <= 65535. */
ADV_API int adv_bias_act_f32(float* y, const float* bias, const float* residual, int64_t planes, int c, int64_t hw, int relu,
                             adv_stream_t stream);

/* The ResNet stem's tail (replaces relu + nn.MaxPool2d(kernel_size=3, stride=2, padding=1) behind the 7x7 convolution, upstream
 *     lib/model/stereo_rcnn/resnet.py [UPSTREAM-UNVERIFIED path]; torch: F.max_pool2d(F.relu(t + bias), 3, 2, 1)):
 *     y [planes][oh][ow] = maxpool(relu(t [planes][h][w] + bias[plane % c])), oh = (h - 1) / 2 + 1, ow likewise; bias NULL = none.
 *     code [planes][oh][ow] bytes: the argmax's position in the 3x3 window (row-major 0..8, first maximum), 15 where relu passes no gradient.
 *   _bwd: grad_t [planes][h][w] from grad_y and the code alone (each pixel sums the outputs that chose it, in (oy, ox) order - torch's). */
ADV_API int adv_stem_pool_fwd_f32(const float* t, const float* bias, float* y, uint8_t* code, int64_t planes, int c, int h, int w,
                                  adv_stream_t stream);
ADV_API int adv_stem_pool_bwd_f32(const float* grad_y, const uint8_t* code, float* grad_t, int64_t planes, int h, int w, adv_stream_t stream);

/* Box arithmetic of the proposal / target stage (csrc/boxes.hip; upstream lib/model/rpn/bbox_transform.py: bbox_overlaps, bbox_transform,
 *   bbox_transform_inv + clip_boxes [UPSTREAM-UNVERIFIED paths], reached inside the detector call of attack/Stereo-RCNN/pgd_attack.py:156).
 *   Boxes are (x1, y1, x2, y2) rows of four floats, 16-byte aligned; widths are x2 - x1 + 1 (the legacy convention).
 *   _iou_rows: a [n][4] against b [m][4]: iou [n][m] (NULL = not wanted), best [n] = the row maximum, arg [n] = index of its first occurrence.
 *   _encode6:  out [n][6] = (dx, dy, log dw, log dh) from src_i onto gt_left[arg_i], then (dx, log dw) from src_right_i (NULL: src_i) onto
 *              gt_right[arg_i].
 *   _decode_stereo: left = src moved by deltas [n][6] columns (0,1,2,3), right by (4,1,5,3) (log-sizes clamped at 4), both clipped to
 *              [0, width-1] x [0, height-1]; big [n] (NULL = not wanted) = 1 where both widths and the left height are >= min_size.
 *   _partition_stereo: STABLE partition of n box pairs: those with big != 0 first (the proposal layer's minimum-size filter without
 *              compaction: the small ones move behind the others, where NMS lets them suppress nothing); nvalid [1] = how many are big
 *              (n, and nothing moves, if none is).  One workgroup.
 *   _sample_rois: the proposal-target layer's sampling with replacement, in order: candidates = the n_gt ground-truth pairs, then
 *              left/right[keep[j]] for the entries 0 <= keep[j] < nvalid of keep [k] (a prefix: adv_nms_f32's padded list); roi i =
 *              candidate i % max(count, 1): rois_* [r][5] = (0, box), out_* [r][4] = the boxes.  One workgroup. */
/* Masked-mean loss terms of the detector's proposal / RoI stages (upstream: the smooth-L1 and objectness losses of stereo_rpn.py / stereo_rcnn.py
 *   [UPSTREAM-UNVERIFIED paths], reached inside the detector call of attack/Stereo-RCNN/pgd_attack.py:156):
 *   out2[0] = sum_i weight[i] * l(pred[i][:], target[i][:]) / max(scale * sum_i weight[i], 1), out2[1] = that denominator;
 *   l = smooth-L1 (beta 1) summed over the k columns, or (bce != 0, k == 1) binary cross-entropy with logits (target = the 0 / 1 label).
 *   workspace: adv_masked_loss_workspace_floats() floats.  Deterministic (fixed chunks, ordered partials).
 *   _bwd: grad_pred [rows][k] = l' * ((grad_loss[0] / out2[1]) * weight[i]) with torch's backward expressions in torch's order. */
ADV_API int64_t adv_masked_loss_workspace_floats(void);
ADV_API int adv_masked_loss_f32(const float* pred, const float* target, const float* weight, float* out2, float* workspace, int64_t rows, int k,
                                float scale, int bce, adv_stream_t stream);
ADV_API int adv_masked_loss_bwd_f32(const float* pred, const float* target, const float* weight, const float* out2, const float* grad_loss,
                                    float* grad_pred, int64_t rows, int k, int bce, adv_stream_t stream);

/* The Stereo R-CNN attack objective (attack/Stereo-RCNN/pgd_attack.py:165-171): loss[0] = sum over k < n of (terms[k] * exp(-u[k]) + u[k]),
 *   added in the script's order (product, then u_k, term after term, starting from 0); w[k] = exp(-u[k]) = d loss / d terms[k].  n <= 64. */
ADV_API int adv_objective_chain_f32(const float* terms, const float* u, float* loss, float* w, int n, adv_stream_t stream);

/* The RPN head's output of one pyramid level, head [b][7 * anchors][hw] (objectness maps, then six regression maps per anchor:
 *   stereo_rpn.py:32-40's RPN_cls_score / RPN_bbox_pred side by side), into the proposal stage's lists in (image, pixel, anchor) order:
 *   scores [b * hw * anchors], deltas [b * hw * anchors][6] (pointers AT this level's offset in the lists of all levels); bounded != 0:
 *   deltas = 0.5 * tanh(raw).  _bwd: grad_head from the gradients of the two lists (NULL = zero); needs head again for tanh's derivative. */
ADV_API int adv_rpn_pack_fwd_f32(const float* head, float* scores, float* deltas, int b, int anchors, int64_t hw, int bounded, adv_stream_t stream);
ADV_API int adv_rpn_pack_bwd_f32(const float* head, const float* grad_scores, const float* grad_deltas, float* grad_head, int b, int anchors,
                                 int64_t hw, int bounded, adv_stream_t stream);
ADV_API int adv_box_partition_stereo_f32(const float* left, const float* right, const int64_t* big, float* out_left, float* out_right,
                                         int64_t* nvalid, int n, adv_stream_t stream);
ADV_API int adv_box_sample_rois_f32(const int64_t* keep, int k, const int64_t* nvalid, const float* left, const float* right,
                                    const float* gt_left, const float* gt_right, int n_gt, int r, float* rois_left, float* rois_right,
                                    float* out_left, float* out_right, adv_stream_t stream);
ADV_API int adv_box_iou_rows_f32(const float* a, const float* b, float* iou, float* best, int64_t* arg, int64_t n, int m, adv_stream_t stream);
ADV_API int adv_box_encode6_f32(const float* src, const float* src_right, const float* gt_left, const float* gt_right, const int64_t* arg,
                                float* out, int64_t n, int m, adv_stream_t stream);
ADV_API int adv_box_decode_stereo_f32(const float* anchors, const float* deltas, float* left, float* right, int64_t* big, int64_t n, float width,
                                      float height, float min_size, adv_stream_t stream);

/* out[i] = y[i] > 0 ? grad[i] : 0  (the backward of a ReLU fused into a convolution's epilogue; out may alias grad). */
ADV_API int adv_relu_backward_f32(const float* grad, const float* y, float* out, int64_t n, adv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVENGINE_H */
