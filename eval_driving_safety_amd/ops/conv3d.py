"""3x3x3 convolutions on the float32 matrix cores: direct, strided, transposed, Winograd (csrc/conv3d.hip, csrc/wino2d.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)
from .elementwise import relu_backward
from .conv2d import WINO4_DIRECT_EQUIV_FLOPS, WINO_DIRECT_EQUIV_FLOPS, _Conv2dChoice, _like

# --------------------------------------------------------------------------------------------
# dense 3x3x3 convolution on the float32 matrix cores (the contraction applied to the K7 cost volume)
def conv3d_k3_prep(weight, transpose=False, out=None):
    """[Cout,Cin,3,3,3] -> the kernel's layout [27, Cin', 32*ceil(Cout'/32)]; transpose=True prepares the adjoint
    (gradient w.r.t. the input).  Do it once per weight tensor - the attacks never change the weights.  ``out``: a contiguous
    float32 tensor of that shape to fill (the eight class tensors of a transposed layer share one allocation)."""
    wt = _feat(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight must be [Cout,Cin,3,3,3]")
    cout, cin = wt.shape[:2]
    cin_p, cout_p = (cout, cin) if transpose else (cin, cout)
    shape = (27, cin_p, 32 * ((cout_p + 31) // 32))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=wt.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != wt.device:
        raise ValueError("out must be a contiguous float32 tensor of shape %s on the weight's device" % (shape,))
    with _on(wt):
        _lib.call("adv_conv3d_k3_prep_weights_f32", _ptr(wt), _ptr(out), cout, cin, int(transpose), _stream(wt))
    return out


ALL_TAPS = (1 << 27) - 1


def _i3(v):
    return None if v is None else (ctypes.c_int32 * 3)(*[int(a) for a in v])


def _residual(residual, out):
    if residual is None:
        return None
    r = _feat(residual, "residual")
    if r.shape != out.shape or r.data_ptr() == out.data_ptr():
        raise ValueError("residual must have the result's shape %s and must not be the result" % (tuple(out.shape),))
    return _ptr(r)


def _conv3d_ex(x, w_prep, cout, stride=1, relu=False, bias=None, tap_mask=ALL_TAPS, out=None, out_stride=None, out_offset=None,
               class_masks=None, residual=None):
    xi, wp = _feat(x, "x"), _feat(w_prep, "w_prep")
    if xi.dim() != 5 or wp.dim() != 3 or wp.shape[0] != 27 or wp.shape[1] != xi.shape[1] or wp.shape[2] < cout:
        raise ValueError("x must be [B,Cin,D,H,W] and w_prep [27,Cin,>=cout]")
    b, cin, d, h, w = xi.shape
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    grid = tuple((v + 1) // 2 for v in (d, h, w)) if stride == 2 else (d, h, w)
    if out is None:
        out = torch.empty((b, cout) + grid, dtype=torch.float32, device=xi.device)
        dims = None
    else:
        out = _feat(out, "out")
        if out.dim() != 5 or out.shape[0] != b or out.shape[1] != cout:
            raise ValueError("out must be [B,cout,D',H',W']")
        dims = tuple(out.shape[2:])
    cm = None if class_masks is None else (ctypes.c_uint32 * 8)(*[int(m) for m in class_masks])
    with _on(xi):
        _lib.call("adv_conv3d_k3_ex_f32", _ptr(xi), _ptr(wp), None if bias is None else _ptr(bias), _residual(residual, out), _ptr(out), b, cin,
                  cout, d, h, w, int(stride),
                  int(relu), int(tap_mask), cm, 0 if cm is None else cin // 8, _i3(dims), _i3(out_stride if dims else None),
                  _i3(out_offset if dims else None), _stream(xi))
    return out


def conv3d_k3(x, w_prep, cout, relu=False, bias=None, residual=None):
    """conv3d(x [B,Cin,D,H,W], stride 1, padding 1) (+ bias [cout]) (+ residual [B,cout,D,H,W]) (+ ReLU) with prepared weights
    -> [B,cout,D,H,W]"""
    if bias is not None or residual is not None:
        return _conv3d_ex(x, w_prep, cout, 1, relu, bias, residual=residual)
    xi, wp = _feat(x, "x"), _feat(w_prep, "w_prep")
    if xi.dim() != 5 or wp.dim() != 3 or wp.shape[0] != 27 or wp.shape[1] != xi.shape[1] or wp.shape[2] < cout:
        raise ValueError("x must be [B,Cin,D,H,W] and w_prep [27,Cin,>=cout]")
    b, cin, d, h, w = xi.shape
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    with _on(xi):
        _lib.call("adv_conv3d_k3_f32", _ptr(xi), _ptr(wp), _ptr(y), b, cin, cout, d, h, w, int(relu), _stream(xi))
    return y


def conv3d_k3_masked(x, w_prep, cout, mask):
    """conv3d(x, w_prep) zeroed where ``mask`` <= 0 - the backward of a layer whose input (``mask``) is a ReLU output it alone consumes;
    returns None when the shape is not one the main matrix kernel takes (the caller then masks with relu_backward)"""
    xi, wp, mk = _feat(x, "x"), _feat(w_prep, "w_prep"), _feat(mask, "mask")
    b, cin, d, h, w = xi.shape
    if tuple(mk.shape) != (b, cout, d, h, w):
        raise ValueError("mask must be [B,cout,D,H,W]")
    if cin >= 4 and (cin % 4 or cout <= 8):      # (fewer than four input channels: the narrow-input kernel, which masks too)
        return None
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    with _on(xi):
        rc = _lib.load().adv_conv3d_k3_masked_f32(_ptr(xi), _ptr(wp), _ptr(mk), _ptr(y), b, cin, cout, d, h, w, _stream(xi))
    if rc == _lib.ADV_EINVAL:
        return None
    _lib.check("adv_conv3d_k3_masked_f32", rc)
    return y


class Conv3dWinoPrep:
    """a 3x3x3 / stride 1 layer's weights transformed for the Winograd kernel (csrc/wino2d.hip, adv_conv3d_wino_f32): G g G^T per channel
    pair and depth tap, for the forward and for the backward w.r.t. the input; each made on first use, once"""

    def __init__(self, weight):
        wt = _feat(weight.detach().contiguous(), "weight")
        if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
            raise ValueError("weight must be [Cout,Cin,3,3,3]")
        self.cout, self.cin, self._wt, self._u, self._w4 = int(wt.shape[0]), int(wt.shape[1]), wt, {}, None

    def wino4(self):
        """the same layer as a ConvWino4Prep (Winograd F(4x4,3x3) in the plane, csrc/wino4.hip), made on first use"""
        if self._w4 is None:
            from .wino4 import ConvWino4Prep
            self._w4 = ConvWino4Prep(self._wt)
        return self._w4

    def u(self, transpose):
        t = self._u.get(bool(transpose))
        if t is None:
            n = int(_lib.load().adv_conv3d_wino_prep_floats(self.cout, self.cin, int(transpose)))
            t = torch.empty((n,), dtype=torch.float32, device=self._wt.device)
            with _on(self._wt):
                _lib.call("adv_conv3d_wino_prep_weights_f32", _ptr(self._wt), _ptr(t), self.cout, self.cin, int(transpose), _stream(self._wt))
            self._u[bool(transpose)] = t
        return t


def _conv3d_wino_call(x, u, cin, cout, bias, residual, relu, mask, tile):
    xi = _feat(x, "x")
    if xi.dim() != 5 or xi.shape[1] != cin:
        raise ValueError("x must be [B,%d,D,H,W]" % cin)
    b, _, d, h, w = xi.shape
    y = torch.empty((b, cout, d, h, w), dtype=torch.float32, device=xi.device)
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    with _on(xi):
        _lib.call("adv_conv3d_wino_f32", _ptr(xi), _ptr(u), None if bias is None else _ptr(bias), _like(residual, y, "residual"), _like(mask, y, "mask"),
                  _ptr(y), b, cin, cout, d, h, w, int(bool(relu)), int(tile), _stream(xi))
    return y


def conv3d_wino(x, prep, bias=None, residual=None, relu=False, mask=None, tile=-1):
    """conv3d(x [B,Cin,D,H,W], 3x3x3, stride 1, padding 1) (+ bias) (+ residual) (ReLU) (zeroed where mask <= 0) by the Winograd kernel:
    2.25x fewer multiply-adds than conv3d_k3, its own order of float operations (oracle: conv3d_wino)"""
    return _conv3d_wino_call(x, prep.u(False), prep.cin, prep.cout, bias, residual, relu, mask, tile)


def conv3d_wino_dgrad(grad, prep, residual=None, mask=None, tile=-1):
    """the backward w.r.t. the input of the same layer: grad [B,Cout,D,H,W] -> [B,Cin,D,H,W] (+ residual) (zeroed where mask <= 0)"""
    return _conv3d_wino_call(grad, prep.u(True), prep.cout, prep.cin, None, residual, False, mask, tile)


def space_to_depth2(x, out=None):
    """[B,C,D,H,W] -> [B,8C,ceil(D/2),ceil(H/2),ceil(W/2)]: the eight parity sub-volumes side by side in the channel dimension"""
    xi = _feat(x, "x")
    b, c, d, h, w = xi.shape
    shape = (b, 8 * c, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2)
    out = torch.empty(shape, dtype=torch.float32, device=xi.device) if out is None else _feat(out, "out")
    if tuple(out.shape) != shape:
        raise ValueError("out must be %s" % (shape,))
    with _on(xi):
        _lib.call("adv_space_to_depth2_f32", _ptr(xi), _ptr(out), b, c, d, h, w, _stream(xi))
    return out


def conv3d_k3_s2_prep(weight):
    """[Cout,Cin,3,3,3] -> (w_prep over the 8*Cin space-to-depth channels, the eight class tap masks, the plain w_prep for the
    direct route).
    out[o] = W0 x[2o-1] + W1 x[2o] + W2 x[2o+1] per axis = W1 e[o] + W0 odd[o-1] + W2 odd[o] with e[j] = x[2j], odd[j] = x[2j+1]:
    the even sub-volume uses tap index 1 (offset 0) with kernel element 1; the odd one tap 0 (offset -1) with element 0 and
    tap 1 with element 2."""
    wt = _feat(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight must be [Cout,Cin,3,3,3]")
    cout, cin = wt.shape[:2]
    pairs = {0: ((1, 1),), 1: ((0, 0), (1, 2))}                                   # parity -> ((tap index, kernel element), ...)
    w8 = torch.zeros((cout, 8 * cin, 3, 3, 3), dtype=torch.float32, device=wt.device)
    masks = []
    for p in range(8):
        mask = 0
        for td, kd in pairs[p >> 2]:
            for th, kh in pairs[(p >> 1) & 1]:
                for tw, kw in pairs[p & 1]:
                    w8[:, p * cin:(p + 1) * cin, td, th, tw] = wt[:, :, kd, kh, kw]
                    mask |= 1 << (td * 9 + th * 3 + tw)
        masks.append(mask)
    return conv3d_k3_prep(w8), tuple(masks), conv3d_k3_prep(wt)


def conv3d_k3_s2_stage_channels(x, cout):
    """2 or 4: the input channels per stage (= the float32 accumulation order) ``conv3d_k3_s2(x, conv3d_k3_prep(w), cout)`` will use -
    2 is the direct strided matrix kernel (W % 4 == 0, 16-byte aligned x), 4 the scalar-staging kernel; the oracle takes it as ``chunk``"""
    xi = _feat(x, "x")
    return int(_lib.load().adv_conv3d_k3_s2_stage_channels(_ptr(xi), int(cout), int(xi.shape[4])))


def conv3d_k3_s2(x, prep, cout, relu=False, bias=None, route="auto"):
    """the strided 3x3x3 convolution of an hourglass: stride 2, padding 1 -> [B,cout,ceil(D/2),ceil(H/2),ceil(W/2)].
    ``prep`` = ``conv3d_k3_s2_prep(weight)`` holds the weights for both routes:
      "direct"  the strided matrix kernel on the raw input (W % 4 == 0: two-channel stages, no permuted copy - the fastest
                route; otherwise the scalar-staging kernel, slow);
      "s2d"     space-to-depth + the stride-1 MFMA kernel with per-class tap masks (any shape; no wasted matrix work);
      "auto"    direct where the matrix kernel takes it, else s2d.
    The routes accumulate in different orders (last-bit differences; each is bit-exact against the oracle run its way).
    A plain ``conv3d_k3_prep(weight)`` tensor is accepted too (direct only)."""
    if isinstance(prep, torch.Tensor):
        return _conv3d_ex(x, prep, cout, 2, relu, bias)
    w_prep8, masks = prep[0], prep[1]
    plain = prep[2] if len(prep) > 2 else None
    if route not in ("auto", "direct", "s2d"):
        raise ValueError("route must be auto, direct or s2d")
    if plain is not None and (route == "direct" or (route == "auto" and conv3d_k3_s2_stage_channels(x, cout) == 2)):
        return _conv3d_ex(x, plain, cout, 2, relu, bias)
    if route == "direct":
        raise ValueError("this prep holds no weights for the direct route")
    return _conv3d_ex(space_to_depth2(x), w_prep8, cout, 1, relu, bias, class_masks=masks)


def conv_transpose3d_k3_s2_prep(weight_t):
    """ConvTranspose3d weights [Cin,Cout,3,3,3] (kernel 3, stride 2, padding 1, output_padding 1) -> the eight output
    parity classes [(w_prep, tap_mask, (pd,ph,pw))].  Output voxel o = 2j + p takes input j + (t - 1) through kernel tap k:
    p = 0: (t, k) = (1, 1);  p = 1: (t, k) = (1, 2), (2, 0)  per axis - so class (pd,ph,pw) is an ordinary convolution over
    the INPUT grid with 1-8 taps, written to every second output voxel."""
    wt = _feat(weight_t, "weight_t")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError("weight_t must be [Cin,Cout,3,3,3]")
    pairs = {0: ((1, 1),), 1: ((1, 2), (2, 0))}
    out = []
    # ONE allocation for the eight class tensors: the transposed kernel then requests them through one buffer descriptor (32-bit offsets
    # from the lowest of the eight pointers - csrc/conv3d.hip; separately allocated tensors still work, through 64-bit pointers)
    slab = torch.empty((8, 27, wt.shape[0], 32 * ((wt.shape[1] + 31) // 32)), dtype=torch.float32, device=wt.device)
    for pd in (0, 1):
        for ph in (0, 1):
            for pw in (0, 1):
                wc = torch.zeros((wt.shape[1], wt.shape[0], 3, 3, 3), dtype=torch.float32, device=wt.device)
                mask = 0
                for td, kd in pairs[pd]:
                    for th, kh in pairs[ph]:
                        for tw, kw in pairs[pw]:
                            wc[:, :, td, th, tw] = wt[:, :, kd, kh, kw].t()
                            mask |= 1 << (td * 9 + th * 3 + tw)
                out.append((conv3d_k3_prep(wc.contiguous(), out=slab[len(out)]), mask, (pd, ph, pw)))
    return out


def conv_transpose3d_k3_s2(x, classes, cout, relu=False, bias=None, residual=None):
    """conv_transpose3d(x [B,Cin,D,H,W], kernel 3, stride 2, padding 1, output_padding 1) -> [B,cout,2D,2H,2W]: the eight
    output parity classes (1-8 taps each) as ONE launch of the persistent masked MFMA kernel, the class being a tile index.
    ``residual`` [B,cout,2D,2H,2W] (a skip connection) is added in the epilogue, after the bias and before the ReLU."""
    xi = _feat(x, "x")
    b, cin, d, h, w = xi.shape
    if len(classes) != 8 or [tuple(c[2]) for c in classes] != [(k >> 2 & 1, k >> 1 & 1, k & 1) for k in range(8)]:
        raise ValueError("classes must be the eight (w_prep, tap_mask, (pd,ph,pw)) of conv_transpose3d_k3_s2_prep, in its order")
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    out = torch.empty((b, cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=xi.device)
    wps = (ctypes.c_void_p * 8)(*[_feat(c[0], "w_prep").data_ptr() for c in classes])
    masks = (ctypes.c_uint32 * 8)(*[int(c[1]) for c in classes])
    with _on(xi):
        _lib.call("adv_conv_transpose3d_k3_s2_f32", _ptr(xi), wps, masks, None if bias is None else _ptr(bias), _residual(residual, out),
                  _ptr(out), b, cin, cout, d, h, w, int(relu), _stream(xi))
    return out


def conv_transpose3d_k3_s2_dgrad(grad, classes, cout, residual=None, mask=None):
    """(conv_transpose3d(grad) + residual) zeroed where mask <= 0, in ONE launch: the backward of a strided 3x3x3 convolution whose input
    (``mask``, a ReLU output) also feeds a skip path whose gradient is ``residual``.  Returns None where the library has no fused kernel
    for the call (misaligned tensors, W < 4): the caller then adds / masks in passes of its own."""
    xi, mk = _feat(grad, "grad"), _feat(mask, "mask")
    b, cin, d, h, w = xi.shape
    out = torch.empty((b, cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=xi.device)
    if tuple(mk.shape) != tuple(out.shape):
        raise ValueError("mask must be laid out like the result %s" % (tuple(out.shape),))
    wps = (ctypes.c_void_p * 8)(*[_feat(c[0], "w_prep").data_ptr() for c in classes])
    masks = (ctypes.c_uint32 * 8)(*[int(c[1]) for c in classes])
    with _on(xi):
        rc = _lib.load().adv_conv_transpose3d_k3_s2_dgrad_f32(_ptr(xi), wps, masks, _residual(residual, out), _ptr(mk), _ptr(out), b, cin, cout, d, h, w,
                                                             _stream(xi))
    if rc == _lib.ADV_EINVAL:
        return None
    _lib.check("adv_conv_transpose3d_k3_s2_dgrad_f32", rc)
    return out


class Conv3dK3(torch.autograd.Function):
    """y = [relu](conv3d(x, weight) [+ bias]); gradient flows to x only (the attacks differentiate w.r.t. the images, the
    detector's weights are constants), through the same kernel family with the transposed / flipped weights (cout a
    multiple of 4 -> the matrix kernel, cout 1..3 -> the narrow vector-ALU kernel).  Any other cout: pass
    ``w_prep_t=None`` and the original ``weight``: the backward then uses torch's conv3d_input.  With ``relu`` the
    activation is fused into the forward's epilogue and its mask is applied to the incoming gradient; ``residual`` (a skip
    connection, [B,cout,D,H,W]) is added in the same epilogue and receives that gradient unchanged."""

    @staticmethod
    def forward(ctx, x, w_prep, w_prep_t, cout, weight=None, bias=None, relu=False, residual=None, mask_input=False, wino=None, skip_out=False):
        """<round 3> chains, as ops.Conv2dAuto: ``mask_input`` - x is a ReLU output this layer alone consumes, the gradient returned for it
        is already masked with x > 0 (in the dgrad kernel's epilogue); ``relu="consumer"`` - this layer's ReLU mask is applied by its only
        consumer's backward, not here.
        ``wino`` (a Conv3dWinoPrep of the same weights): each direction is computed by the direct kernel or by the Winograd kernel
        (csrc/wino2d.hip: 2.25x fewer multiply-adds, same epilogues), whichever measured faster for the layer shape at first use.
        <round 5> ``skip_out`` (as ops.Conv2dAuto / Conv3dK3S2): returns (y, x_skip), x_skip an alias of x - a residual block hands it to
        its last layer as the ``residual``; this layer's backward then receives BOTH gradients of x and adds the skip path's in the dgrad
        kernel's epilogue, before the mask of ``mask_input`` (which then covers both)."""
        x = x.contiguous()
        ctx.skip_out = bool(skip_out)
        ctx.has_t, ctx.has_res = w_prep_t is not None, residual is not None
        ctx.mask_own, ctx.mask_input = bool(relu) and relu != "consumer", bool(mask_input)
        ctx.xshape, ctx.wino = tuple(x.shape), wino
        res = None if residual is None else residual.contiguous()
        direct = lambda: conv3d_k3(x, w_prep, cout, relu=bool(relu), bias=bias, residual=res)       # noqa: E731
        if wino is not None and cout >= 4:
            by_wino = lambda: conv3d_wino(x, wino, bias, res, bool(relu))                            # noqa: E731

            def by_wino4():
                from .wino4 import conv_wino4
                return conv_wino4(x, wino.wino4(), bias, res, bool(relu))

            key = ("f3", x.shape[1], cout, tuple(x.shape), res is not None, bool(relu))
            took = _Conv2dChoice.pick(key, {"direct": direct, "wino": by_wino, "wino4": by_wino4})
            y = by_wino() if took == "wino" else (by_wino4() if took == "wino4" else direct())
            if took == "wino":
                WINO_DIRECT_EQUIV_FLOPS[0] += 54 * y.numel() * x.shape[1]
            elif took == "wino4":
                WINO4_DIRECT_EQUIV_FLOPS[0] += 54 * y.numel() * x.shape[1]
        else:
            y = direct()
        ctx.save_for_backward(w_prep_t if ctx.has_t else weight, y if ctx.mask_own else None, x if mask_input else None)
        if skip_out:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, grad_y, grad_skip=None):
        w, y, x_in = ctx.saved_tensors
        if ctx.mask_own:
            grad_y = relu_backward(grad_y, y)
        gres = grad_y if ctx.has_res else None      # the skip connection receives the (masked) gradient as it is
        skip = grad_skip.contiguous() if (ctx.skip_out and grad_skip is not None) else None
        if ctx.has_t:
            g = grad_y.contiguous()

            def direct():
                if skip is not None:                # the direct kernel's epilogue adds the skip gradient; the mask (if any) is a pass behind it
                    gx = conv3d_k3(g, w, ctx.xshape[1], residual=skip)
                    return relu_backward(gx, x_in) if ctx.mask_input else gx
                gx = conv3d_k3_masked(g, w, ctx.xshape[1], x_in) if ctx.mask_input else None
                if gx is None:
                    gx = conv3d_k3(g, w, ctx.xshape[1])
                    if ctx.mask_input:
                        gx = relu_backward(gx, x_in)
                return gx

            if ctx.wino is not None and ctx.xshape[1] >= 4:
                by_wino = lambda: conv3d_wino_dgrad(g, ctx.wino, residual=skip, mask=x_in if ctx.mask_input else None)      # noqa: E731

                def by_wino4():
                    from .wino4 import conv_wino4_dgrad
                    return conv_wino4_dgrad(g, ctx.wino.wino4(), residual=skip, mask=x_in if ctx.mask_input else None)

                key = ("b3", ctx.xshape[1], g.shape[1], ctx.xshape, ctx.mask_input)
                took = _Conv2dChoice.pick(key, {"direct": direct, "wino": by_wino, "wino4": by_wino4})
                gx = by_wino() if took == "wino" else (by_wino4() if took == "wino4" else direct())
                if took == "wino":
                    WINO_DIRECT_EQUIV_FLOPS[0] += 54 * gx.numel() * g.shape[1]
                elif took == "wino4":
                    WINO4_DIRECT_EQUIV_FLOPS[0] += 54 * gx.numel() * g.shape[1]
            else:
                gx = direct()
        else:
            gx = torch.nn.grad.conv3d_input(ctx.xshape, w, grad_y, padding=1)
            if skip is not None:
                gx = gx + skip
            if ctx.mask_input:
                gx = relu_backward(gx, x_in)
        return gx, None, None, None, None, None, None, gres, None, None, None


class Conv3dK3S2(torch.autograd.Function):
    """y = conv3d(x, weight, stride 2, padding 1); the gradient w.r.t. x is the transposed convolution of grad_y with the
    same weights (``classes_t = conv_transpose3d_k3_s2_prep(weight)``), cropped to x's size when a dimension is odd.

    <round 4> An hourglass's down-sampling layer reads a tensor that ALSO feeds the matching up-sampling layer's skip connection.  Two
    flags take the addition of the two gradients and the producer's ReLU mask into this layer's backward launch (as ops.Conv2dAuto's):
      ``skip_out=True``    returns (y, x_skip), x_skip an alias of x: hand it to the up-sampling layer as its ``residual``; this layer's backward
                           then receives both gradients of x and adds the skip path's in the transposed kernel's epilogue;
      ``mask_input=True``  x is a ReLU output whose only consumers are this layer and that skip path, and its producer left the mask to us
                           (relu="consumer"): the gradient returned is already multiplied by (x > 0).
    The caller vouches for the topology; same float operations in the same order as autograd's addition and a relu-backward pass."""

    @staticmethod
    def forward(ctx, x, w_prep, classes_t, cout, bias=None, relu=False, mask_input=False, skip_out=False):
        ctx.classes_t, ctx.xshape = classes_t, tuple(x.shape)
        ctx.relu = bool(relu) and relu != "consumer"        # "consumer": the only consumer's backward applies this layer's ReLU mask (Conv3dK3 mask_input)
        ctx.mask_input = bool(mask_input)
        x = x.contiguous()
        y = conv3d_k3_s2(x, w_prep, cout, relu=bool(relu), bias=bias)
        ctx.save_for_backward(y if ctx.relu else None, x if mask_input else None)
        return (y, x) if skip_out else y

    @staticmethod
    def backward(ctx, grad_y, grad_skip=None):
        y, x_in = ctx.saved_tensors
        if ctx.relu:
            grad_y = relu_backward(grad_y, y)
        d, h, w = ctx.xshape[2:]
        gy = grad_y.contiguous()
        even = (2 * gy.shape[2], 2 * gy.shape[3], 2 * gy.shape[4]) == (d, h, w)
        skip = None if grad_skip is None else grad_skip.contiguous()
        g = None
        if even and ctx.mask_input:                         # transposed convolution + skip gradient + mask in one launch
            g = conv_transpose3d_k3_s2_dgrad(gy, ctx.classes_t, ctx.xshape[1], residual=skip, mask=x_in)
        if g is None:
            g = conv_transpose3d_k3_s2(gy, ctx.classes_t, ctx.xshape[1], residual=skip if even else None)
            if tuple(g.shape[2:]) != (d, h, w):
                g = g[:, :, :d, :h, :w].contiguous()
            if skip is not None and not even:
                g = g + skip
            if ctx.mask_input:
                g = relu_backward(g, x_in)
        return g, None, None, None, None, None, None, None


class ConvTranspose3dK3S2(torch.autograd.Function):
    """y = conv_transpose3d(x, weight_t, stride 2, padding 1, output_padding 1); the gradient w.r.t. x is the strided
    convolution of grad_y with weight_t read as [out = Cin, in = Cout] (``w_prep_fwd = conv3d_k3_s2_prep(weight_t)``)."""

    @staticmethod
    def forward(ctx, x, classes, w_prep_fwd, cout, bias=None, relu=False, residual=None):
        """relu="consumer": the ReLU is applied here, its backward mask by this layer's only consumer (ops.BevFold mask_input)"""
        ctx.w_prep_fwd, ctx.cin, ctx.has_res = w_prep_fwd, x.shape[1], residual is not None
        ctx.relu = bool(relu) and relu != "consumer"
        y = conv_transpose3d_k3_s2(x.contiguous(), classes, cout, relu=bool(relu), bias=bias, residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(y if ctx.relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        if ctx.relu:
            grad_y = relu_backward(grad_y, y)
        return (conv3d_k3_s2(grad_y.contiguous(), ctx.w_prep_fwd, ctx.cin), None, None, None, None, None,
                grad_y if ctx.has_res else None)        # the skip connection receives the (masked) gradient as it is


__all__ = [n for n in dir() if not n.startswith("__")]
