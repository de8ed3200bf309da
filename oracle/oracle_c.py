"""ctypes face of oracle/oracle.c with the same function names as oracle_np (the subset the
CPU baseline and the cross-check tests use).  TEST INFRASTRUCTURE ONLY - see oracle.c."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("ADV_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")     # ADV_ORACLE_LIB: the sanitizer build (Makefile)
if not os.path.exists(_PATH):
    raise ImportError("oracle/_build/liboracle.so not built (make -C oracle)")
_lib = ctypes.CDLL(_PATH)
_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_lib.orc_num_threads.restype = ctypes.c_int
_lib.orc_denormalize.argtypes = [_fp, _fp, ctypes.c_long, ctypes.c_long]
_lib.orc_normalize.argtypes = [_fp, _fp, ctypes.c_long, ctypes.c_long]
_lib.orc_pgd_step_norm01.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_long, ctypes.c_long, ctypes.c_float, ctypes.c_float]
_lib.orc_pgd_step_meansub255.argtypes = _lib.orc_pgd_step_norm01.argtypes
_lib.orc_tensor2im_u8.argtypes = [_fp, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
_lib.orc_patch_paste.argtypes = [_fp, _fp] + [ctypes.c_int] * 5
_lib.orc_patch_update.argtypes = [_fp, _fp, _fp] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_float,
                                                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
_lib.orc_srcnn_export.argtypes = [_fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
_lib.orc_srcnn_export.restype = None
_lib.orc_disc_mask.argtypes = [_fp] + [ctypes.c_int] * 5
_lib.orc_disc_mask.restype = None
_lib.orc_conv3d_k3.argtypes = [_fp, _fp, _fp] + [ctypes.c_int] * 8
_lib.orc_conv3d_k3.restype = None
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lib.orc_dense_align_cost.argtypes = [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _fp, ctypes.c_int, _fp, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_int, _fp]
_lib.orc_dense_align_cost.restype = None
_lib.orc_conv3d_k3_ex.argtypes = [_fp, _fp, ctypes.c_void_p, _fp] + [ctypes.c_int] * 8 + [ctypes.c_uint, _i32p, _i32p, _i32p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
_lib.orc_conv3d_k3_ex.restype = None
_lib.orc_conv2d.argtypes = [_fp, _fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _fp] + [ctypes.c_int] * 12
_lib.orc_conv2d.restype = None
_lib.orc_conv2d_wino.argtypes = [_fp, _fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _fp] + [ctypes.c_int] * 7
_lib.orc_conv2d_wino.restype = None
_lib.orc_conv3d_wino.argtypes = [_fp, _fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _fp] + [ctypes.c_int] * 8
_lib.orc_conv3d_wino.restype = None
_lib.orc_conv_wino4.argtypes = [_fp, _fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _fp] + [ctypes.c_int] * 9
_lib.orc_conv_wino4.restype = None
_lib.orc_conv_wino4_ksplit.argtypes = [_fp, _fp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _fp] + [ctypes.c_int] * 8
_lib.orc_conv_wino4_ksplit.restype = None
for _f in ("orc_denormalize", "orc_normalize", "orc_pgd_step_norm01", "orc_pgd_step_meansub255", "orc_tensor2im_u8",
           "orc_patch_paste", "orc_patch_update"):
    getattr(_lib, _f).restype = None


def num_threads():
    return _lib.orc_num_threads()


def _nhw(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 4 and a.shape[1] == 3
    return a, a.shape[0], a.shape[2] * a.shape[3]


def denormalize(im):
    a, n, hw = _nhw(im)
    out = np.empty_like(a)
    _lib.orc_denormalize(a, out, n, hw)
    return out


def normalize(im):
    a, n, hw = _nhw(im)
    out = np.empty_like(a)
    _lib.orc_normalize(a, out, n, hw)
    return out


def pgd_step_norm01(x, grad, clean, alpha, eps):
    a, n, hw = _nhw(x)
    out = np.empty_like(a)
    _lib.orc_pgd_step_norm01(a, np.ascontiguousarray(grad), np.ascontiguousarray(clean), out, n, hw, alpha, eps)
    return out


def pgd_step_meansub255(x, grad, clean, alpha, eps255):
    a, n, hw = _nhw(x)
    out = np.empty_like(a)
    _lib.orc_pgd_step_meansub255(a, np.ascontiguousarray(grad), np.ascontiguousarray(clean), out, n, hw, alpha, eps255)
    return out


def tensor2im_u8(x_norm, crop_h, crop_w):
    a = np.ascontiguousarray(x_norm, dtype=np.float32)
    assert a.ndim == 3 and a.shape[0] == 3
    out = np.empty((crop_h, crop_w, 3), np.uint8)
    _lib.orc_tensor2im_u8(a, out, a.shape[1], a.shape[2], crop_h, crop_w)
    return out


def patch_paste(img, patch, cy, cx, radius):
    out = np.array(img, dtype=np.float32, copy=True)
    assert out.shape[0] == 1
    _lib.orc_patch_paste(out, np.ascontiguousarray(patch, dtype=np.float32), out.shape[2], out.shape[3], cy, cx, radius)
    return out


def patch_update(patch, grad_l, grad_r, cy, cx_l, cx_r, radius, eps, alpha=1e3, lo=None, hi=None):
    out = np.array(patch, dtype=np.float32, copy=True)
    gl, gr = np.ascontiguousarray(grad_l, dtype=np.float32), np.ascontiguousarray(grad_r, dtype=np.float32)
    assert gl.shape[0] == 1
    lo_a = None if lo is None else (ctypes.c_float * 3)(*[float(v) for v in lo])
    hi_a = None if hi is None else (ctypes.c_float * 3)(*[float(v) for v in hi])
    _lib.orc_patch_update(out, gl, gr, gl.shape[2], gl.shape[3], cy, cx_l, cx_r, radius, 0.5 * alpha, eps,
                          None if lo_a is None else ctypes.cast(lo_a, ctypes.c_void_p),
                          None if hi_a is None else ctypes.cast(hi_a, ctypes.c_void_p), None)
    return out


def srcnn_hwc_plus_means(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((a.shape[1], a.shape[2], 3), np.float32)
    _lib.orc_srcnn_export(a, out.ctypes.data_as(ctypes.c_void_p), None, a.shape[1], a.shape[2])
    return out


def srcnn_export_u8(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((a.shape[1], a.shape[2], 3), np.uint8)
    _lib.orc_srcnn_export(a, None, out.ctypes.data_as(ctypes.c_void_p), a.shape[1], a.shape[2])
    return out


def disc_mask(h, w, cy, cx, radius):
    out = np.empty((h, w), np.float32)
    _lib.orc_disc_mask(out, h, w, cy, cx, radius)
    return out


def conv3d_k3(x, w, relu=False, transpose=False):
    """x [B,C,D,H,W], w [Cout,Cin,3,3,3] -> conv3d(stride 1, pad 1) or, with transpose, its adjoint applied to x"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[:2]
    b, c, d, h, ww = x.shape
    assert c == (cout if transpose else cin)
    y = np.empty((b, cin if transpose else cout, d, h, ww), np.float32)
    _lib.orc_conv3d_k3(x, w, y, b, cin, cout, d, h, ww, int(relu), int(transpose))
    return y


def dense_align_cost(left, right, roi, dz, z_center, fb, step, k):
    """csrc/align.hip's cost in the kernel's own summation order: left/right [3,H,W], roi int32 [n,4], dz [n,stride],
    z_center [n] -> cost [n,k]"""
    left = np.ascontiguousarray(left, dtype=np.float32)
    right = np.ascontiguousarray(right, dtype=np.float32)
    roi = np.ascontiguousarray(roi, dtype=np.int32)
    dz = np.ascontiguousarray(dz, dtype=np.float32)
    zc = np.ascontiguousarray(z_center, dtype=np.float32)
    n = roi.shape[0]
    cost = np.empty((n, k), np.float32)
    _lib.orc_dense_align_cost(left, right, left.shape[1], left.shape[2], n, roi, dz, dz.shape[1], zc, float(fb), float(step), int(k), cost)
    return cost


def dense_align_argmin(cost, z_center, step):
    """first minimum per object; inf/NaN never win (csrc/align.hip: dense_align_argmin_kernel)"""
    cost = np.asarray(cost, np.float32)
    n, k = cost.shape
    z = np.empty(n, np.float32)
    cmin = np.empty(n, np.float32)
    half = np.float32(0.5) * np.float32(k - 1)
    for b in range(n):
        best, bc = -1, np.float32(np.inf)
        for j in range(k):
            if cost[b, j] < bc:
                best, bc = j, cost[b, j]
        kk = np.float32(best) if best >= 0 else half
        z[b] = np.float32(z_center[b]) + (kk - half) * np.float32(step)
        cmin[b] = bc
    return z, cmin


def conv3d_k3_ex(x, w, bias=None, stride=1, relu=False, tap_mask=(1 << 27) - 1, out=None, out_stride=(1, 1, 1), out_offset=(0, 0, 0),
                 class_masks=None, chunk=4):
    """csrc/conv3d.hip's extended entry point: x [B,Cin,D,H,W], w [Cout,Cin,3,3,3] (ordinary conv layout); ``chunk`` = input
    channels per stage of the kernel being checked (the accumulation order is stage, tap, channel: 4 everywhere except the direct
    strided matrix kernel, which stages 2 - ops.conv3d_k3_s2_stage_channels)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[:2]
    b, c, d, h, ww = x.shape
    assert c == cin
    grid = tuple((v + 1) // 2 for v in (d, h, ww)) if stride == 2 else (d, h, ww)
    if out is None:
        out = np.zeros((b, cout) + grid, np.float32)
    bp = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)
    cm = None if class_masks is None else np.array(class_masks, np.uint32)
    _lib.orc_conv3d_k3_ex(x, w, bp, out, b, cin, cout, d, h, ww, int(stride), int(relu), int(tap_mask),
                          np.array(out.shape[2:], np.int32), np.array(out_stride, np.int32), np.array(out_offset, np.int32),
                          None if cm is None else cm.ctypes.data_as(ctypes.c_void_p), 0 if cm is None else cin // 8, int(chunk))
    return out


def conv2d(x, w, bias=None, residual=None, mask=None, stride=1, padding=0, relu=False, transpose=False, chunk=16, dilation=1):
    """csrc/conv2d.hip in its accumulation order: x [B,Cin,H,W], w [Cout,Cin,k,k] -> conv2d (+ bias, + residual, ReLU, mask);
    transpose=True: x is grad_out [B,Cout,H,W] -> the gradient w.r.t. the input [B,Cin,H,W] (stride 1)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    b, c, h, ww = x.shape
    assert c == (cout if transpose else cin) and w.shape[3] == k and (stride == 1 or not transpose)
    ho, wo = (h, ww) if transpose else ((h + 2 * padding - dilation * (k - 1) - 1) // stride + 1, (ww + 2 * padding - dilation * (k - 1) - 1) // stride + 1)
    y = np.empty((b, cin if transpose else cout, ho, wo), np.float32)

    def opt(a, shape):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == shape, (a.shape, shape)
        return a, a.ctypes.data_as(ctypes.c_void_p)

    keep = [opt(bias, (y.shape[1],)), opt(residual, y.shape), opt(mask, y.shape)]
    _lib.orc_conv2d(x, w, keep[0][1], keep[1][1], keep[2][1], y, b, cin, cout, h, ww, k, int(stride), int(padding), int(dilation), int(relu),
                    int(transpose), int(chunk))
    return y


def conv2d_wino(x, w, bias=None, residual=None, mask=None, relu=False, transpose=False):
    """csrc/wino2d.hip in its order of operations: the 3x3 stride-1 pad-1 convolution by Winograd F(2x2,3x3) (+ bias, + residual, ReLU,
    mask); transpose=True: x is grad_out -> the gradient w.r.t. the input"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[0], w.shape[1]
    b, c, h, ww = x.shape
    assert w.shape[2:] == (3, 3) and c == (cout if transpose else cin)
    y = np.empty((b, cin if transpose else cout, h, ww), np.float32)

    def opt(a, shape):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == shape, (a.shape, shape)
        return a, a.ctypes.data_as(ctypes.c_void_p)

    keep = [opt(bias, (y.shape[1],)), opt(residual, y.shape), opt(mask, y.shape)]
    _lib.orc_conv2d_wino(x, w, keep[0][1], keep[1][1], keep[2][1], y, b, cin, cout, h, ww, int(relu), int(transpose))
    return y


def conv3d_wino(x, w, bias=None, residual=None, mask=None, relu=False, transpose=False):
    """csrc/wino2d.hip on a 3x3x3 layer in its order of operations (Winograd F(2x2,3x3) in the (H, W) plane, the depth taps inside the
    contraction): x [B,Cin,D,H,W], w [Cout,Cin,3,3,3] -> [B,Cout,D,H,W] (+ bias, + residual, ReLU, mask); transpose=True: x is grad_out
    -> the gradient w.r.t. the input"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[0], w.shape[1]
    b, c, d, h, ww = x.shape
    assert w.shape[2:] == (3, 3, 3) and c == (cout if transpose else cin)
    y = np.empty((b, cin if transpose else cout, d, h, ww), np.float32)

    def opt(a, shape):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == shape, (a.shape, shape)
        return a, a.ctypes.data_as(ctypes.c_void_p)

    keep = [opt(bias, (y.shape[1],)), opt(residual, y.shape), opt(mask, y.shape)]
    _lib.orc_conv3d_wino(x, w, keep[0][1], keep[1][1], keep[2][1], y, b, cin, cout, d, h, ww, int(relu), int(transpose))
    return y


def space_to_depth2(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    b, c, d, h, w = x.shape
    d2, h2, w2 = (d + 1) // 2, (h + 1) // 2, (w + 1) // 2
    pad = np.zeros((b, c, 2 * d2, 2 * h2, 2 * w2), np.float32)
    pad[:, :, :d, :h, :w] = x
    out = np.empty((b, 8 * c, d2, h2, w2), np.float32)
    for p in range(8):
        out[:, p * c:(p + 1) * c] = pad[:, :, (p >> 2)::2, ((p >> 1) & 1)::2, (p & 1)::2]
    return out


def conv3d_k3_s2(x, w, bias=None, relu=False):
    """the strided convolution as ops.conv3d_k3_s2 computes it: space-to-depth + a stride-1 convolution over 8*Cin channels
    in which parity sub-volume p keeps only the taps its parity allows (same accumulation order as the kernel)"""
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[:2]
    pairs = {0: ((1, 1),), 1: ((0, 0), (1, 2))}
    w8 = np.zeros((cout, 8 * cin, 3, 3, 3), np.float32)
    masks = []
    for p in range(8):
        mask = 0
        for td, kd in pairs[p >> 2]:
            for th, kh in pairs[(p >> 1) & 1]:
                for tw, kw in pairs[p & 1]:
                    w8[:, p * cin:(p + 1) * cin, td, th, tw] = w[:, :, kd, kh, kw]
                    mask |= 1 << (td * 9 + th * 3 + tw)
        masks.append(mask)
    return conv3d_k3_ex(space_to_depth2(x), w8, bias=bias, relu=relu, class_masks=masks)


def conv_transpose3d_k3_s2(x, weight_t, bias=None, relu=False):
    """kernel 3, stride 2, padding 1, output_padding 1, as the eight parity-class convolutions of ops.conv_transpose3d_k3_s2"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    wt = np.ascontiguousarray(weight_t, dtype=np.float32)
    cin, cout = wt.shape[:2]
    b, _, d, h, w = x.shape
    out = np.zeros((b, cout, 2 * d, 2 * h, 2 * w), np.float32)
    pairs = {0: ((1, 1),), 1: ((1, 2), (2, 0))}
    for pd in (0, 1):
        for ph in (0, 1):
            for pw in (0, 1):
                wc = np.zeros((cout, cin, 3, 3, 3), np.float32)
                mask = 0
                for td, kd in pairs[pd]:
                    for th, kh in pairs[ph]:
                        for tw, kw in pairs[pw]:
                            wc[:, :, td, th, tw] = wt[:, :, kd, kh, kw].T
                            mask |= 1 << (td * 9 + th * 3 + tw)
                conv3d_k3_ex(x, wc, bias, 1, relu, mask, out, (2, 2, 2), (pd, ph, pw))
    return out


def conv_wino4(x, w, bias=None, residual=None, mask=None, relu=False, transpose=False, chunk=0):
    """csrc/wino4.hip in its order of operations: Winograd F(4x4,3x3) in the (H, W) plane - x [B,C,H,W] with w [Cout,Cin,3,3], or x
    [B,C,D,H,W] with w [Cout,Cin,3,3,3] (the depth taps inside the contraction) - + bias, + residual, ReLU, mask; transpose=True: x is
    grad_out -> the gradient w.r.t. the input.  ``chunk`` > 0 (2D): the K-split launch's order - parts of ``chunk`` input channels,
    each through its own output transform, added in order"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[0], w.shape[1]
    three_d = x.ndim == 5
    assert not (chunk and three_d)
    assert w.shape[2:] == ((3, 3, 3) if three_d else (3, 3)) and x.shape[1] == (cout if transpose else cin)
    b, d, h, ww = x.shape[0], (x.shape[2] if three_d else 1), x.shape[-2], x.shape[-1]
    y = np.empty((b, cin if transpose else cout) + tuple(x.shape[2:]), np.float32)

    def opt(a, shape):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape == shape, (a.shape, shape)
        return a, a.ctypes.data_as(ctypes.c_void_p)

    keep = [opt(bias, (y.shape[1],)), opt(residual, y.shape), opt(mask, y.shape)]
    if chunk:
        _lib.orc_conv_wino4_ksplit(x, w, keep[0][1], keep[1][1], keep[2][1], y, b, cin, cout, h, ww, int(relu), int(transpose), int(chunk))
        return y
    _lib.orc_conv_wino4(x, w, keep[0][1], keep[1][1], keep[2][1], y, b, cin, cout, d, h, ww, 3 if three_d else 1, int(relu), int(transpose))
    return y
