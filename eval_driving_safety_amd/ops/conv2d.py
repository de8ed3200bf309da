"""2D convolutions on the float32 matrix cores: 1x1, 3x3, Winograd, the route choice; fused bias / ReLU pass (csrc/conv2d.hip, csrc/wino2d.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)
from .elementwise import relu_backward

# --------------------------------------------------------------------------------------------
# 2D convolutions of the detectors' backbones on the float32 matrix cores (csrc/conv2d.hip)
def conv2d_supported(x, weight, stride=1, padding=0, dilation=1):
    """does libadvengine have a kernel for this layer?  1x1 / stride 1 / no padding (a GEMM per image) and 3x3 / stride 1 /
    padding = dilation in (1, 2); strided layers and other kernel sizes stay on MIOpen"""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2] == weight.shape[3] and stride == 1):
        return False
    if x.dim() != 4 or x.numel() < 4 or x.shape[0] * weight.shape[0] * x.shape[2] * x.shape[3] < 4:
        return False            # the kernels load whole float4s (clamped into the tensor): fewer than four floats are refused
    k = weight.shape[2]
    return (k == 1 and padding == 0 and dilation == 1) or (k == 3 and dilation in (1, 2) and padding == dilation)


class Conv2dPrep:
    """the weights of one layer in the kernels' layout, for the forward and for the backward w.r.t. the input - prepared once (the
    attacks never change the weights)"""

    def __init__(self, weight, stride=1, padding=0, dilation=1):
        wt = _feat(weight.detach().contiguous(), "weight")
        if wt.dim() != 4 or wt.shape[2] != wt.shape[3] or stride != 1:
            raise ValueError("weight must be [Cout,Cin,k,k], stride 1")
        self.cout, self.cin, self.k, self.stride, self.padding, self.dilation = int(wt.shape[0]), int(wt.shape[1]), int(wt.shape[2]), 1, int(padding), int(dilation)
        if not ((self.k == 1 and padding == 0 and dilation == 1) or (self.k == 3 and dilation in (1, 2) and padding == dilation)):
            raise ValueError("only 1x1 / padding 0 and 3x3 / padding = dilation in (1, 2) layers have a kernel here")
        self.device = wt.device
        self.fwd, self.bwd = self._prep(wt, False), self._prep(wt, True)
        self.has_wino = self.k == 3 and self.dilation == 1      # Winograd F(2x2,3x3) route (csrc/wino2d.hip): prepared on first use
        self._wt, self._wino, self._wino4 = (wt if self.has_wino else None), {}, None

    def wino4(self):
        """the layer as a ConvWino4Prep (Winograd F(4x4,3x3), csrc/wino4.hip), made on first use"""
        if not self.has_wino:
            raise ValueError("the Winograd routes need a 3x3 / dilation 1 layer")
        if self._wino4 is None:
            from .wino4 import ConvWino4Prep
            self._wino4 = ConvWino4Prep(self._wt)
        return self._wino4

    def wino(self, transpose):
        """the layer's transformed weights G g G^T for adv_conv2d_wino_f32 (forward / backward w.r.t. the input), made once"""
        if not self.has_wino:
            raise ValueError("the Winograd route needs a 3x3 / dilation 1 layer")
        t = self._wino.get(bool(transpose))
        if t is None:
            t = self._wino[bool(transpose)] = self._prep(self._wt, transpose, "wino")
        return t

    def _prep(self, wt, transpose, kind=None):
        floats, prep = {1: ("adv_conv2d_1x1_prep_floats", "adv_conv2d_1x1_prep_weights_f32"),
                        3: ("adv_conv2d_3x3_prep_floats", "adv_conv2d_3x3_prep_weights_f32"),
                        "wino": ("adv_conv2d_wino_prep_floats", "adv_conv2d_wino_prep_weights_f32")}[kind or self.k]
        n = int(getattr(_lib.load(), floats)(self.cout, self.cin, int(transpose)))
        out = torch.empty((n,), dtype=torch.float32, device=wt.device)
        with _on(wt):
            _lib.call(prep, _ptr(wt), _ptr(out), self.cout, self.cin, int(transpose), _stream(wt))
        return out


def _like(t, out, name):
    if t is None:
        return None
    t = _feat(t, name)
    if t.shape != out.shape or t.data_ptr() == out.data_ptr():
        raise ValueError("%s must have the result's shape %s and must not be the result" % (name, tuple(out.shape)))
    return _ptr(t)


def _conv2d_call(x, prep, w_prep, cin, cout, bias, residual, relu, mask, tile, wino=False):
    xi = _feat(x, "x")
    if xi.dim() != 4 or xi.shape[1] != cin:
        raise ValueError("x must be [B,%d,H,W]" % cin)
    b, _, h, w = xi.shape
    y = torch.empty((b, cout, h, w), dtype=torch.float32, device=xi.device)
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    bp = None if bias is None else _ptr(bias)
    with _on(xi):
        if wino:
            _lib.call("adv_conv2d_wino_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin,
                      cout, h, w, int(bool(relu)), int(tile), _stream(xi))
        elif prep.k == 1:
            _lib.call("adv_conv2d_1x1_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout,
                      h * w, int(bool(relu)), int(tile), _stream(xi))
        else:
            _lib.call("adv_conv2d_3x3_f32", _ptr(xi), _ptr(w_prep), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout,
                      h, w, prep.dilation, int(bool(relu)), int(min(tile, 2)), _stream(xi))
    return y


def conv2d(x, prep, bias=None, residual=None, relu=False, mask=None, tile=-1, wino=False):
    """conv2d(x [B,Cin,H,W], prep) (+ bias [Cout]) (+ residual [B,Cout,H,W]) (ReLU) (zeroed where mask <= 0) -> [B,Cout,H,W].
    wino=True (3x3 / dilation 1 layers): the Winograd F(2x2,3x3) kernel - 2.25x fewer multiply-adds, its own order of float operations
    (oracle: conv2d_wino)"""
    if wino:
        return _conv2d_call(x, prep, prep.wino(False), prep.cin, prep.cout, bias, residual, relu, mask, tile, True)
    return _conv2d_call(x, prep, prep.fwd, prep.cin, prep.cout, bias, residual, relu, mask, tile)


def conv2d_dgrad(grad, prep, hw=None, residual=None, mask=None, tile=-1, wino=False):
    """the backward w.r.t. the input of the same layer: grad [B,Cout,H,W] -> [B,Cin,H,W] (+ residual: a gradient arriving over a
    skip path) (zeroed where mask <= 0: with mask = the layer's own input, a ReLU output, this is the gradient w.r.t. the previous
    layer's pre-activation)"""
    if wino:
        return _conv2d_call(grad, prep, prep.wino(True), prep.cout, prep.cin, None, residual, False, mask, tile, True)
    return _conv2d_call(grad, prep, prep.bwd, prep.cout, prep.cin, None, residual, False, mask, tile)


class Conv2d(torch.autograd.Function):
    """y = [relu](conv2d(x, prep) [+ bias] [+ residual]); gradients flow to x and to the residual only (the detector's weights are
    constants in an attack).  With ``relu`` the incoming gradient is masked with y > 0 first (ops.relu_backward)."""

    @staticmethod
    def forward(ctx, x, prep, bias=None, residual=None, relu=False):
        ctx.prep, ctx.relu, ctx.has_res = prep, bool(relu), residual is not None
        y = conv2d(x.contiguous(), prep, bias, None if residual is None else residual.contiguous(), relu)
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        g = grad_y.contiguous()
        if ctx.relu:
            g = relu_backward(g, y)
        return conv2d_dgrad(g, ctx.prep), None, None, (g if ctx.has_res else None), None


def bias_act_(y, bias=None, residual=None, relu=False):
    """y <- [relu](y + bias[c] + residual) in place, one pass (the epilogue of a convolution another library computed)"""
    yi = _feat(y, "y")
    if yi.dim() < 2:
        raise ValueError("y must be [B,C,...]")
    b, c = yi.shape[0], yi.shape[1]
    hw = yi.numel() // max(1, b * c)
    if yi.numel() == 0 or (bias is None and residual is None and not relu):
        return yi
    if b * c > 65535:                     # beyond the launch's plane index: torch's own operators
        if bias is not None:
            yi += bias.view(1, -1, *([1] * (yi.dim() - 2)))
        if residual is not None:
            yi += residual
        return yi.relu_() if relu else yi
    if bias is not None:
        bias = _feat(bias, "bias")
    with _on(yi):
        _lib.call("adv_bias_act_f32", _ptr(yi), None if bias is None else _ptr(bias), _like(residual, yi, "residual"), b * c, c, hw, int(bool(relu)),
                  _stream(yi))
    return yi


class BiasAct(torch.autograd.Function):
    """``[relu](y + bias[c] + residual)`` in place as ONE pass behind a convolution torch computed (adopt.py's layers without a kernel here,
    the bird's-eye view's transposed 2D layers), with the backward autograd cannot derive from a raw kernel call: the incoming gradient
    masked with (result > 0) when the ReLU is fused; the residual (a skip connection) receives that masked gradient as it is"""

    @staticmethod
    def forward(ctx, y, bias=None, relu=False, residual=None):
        out = bias_act_(y, bias, None if residual is None else residual.contiguous(), relu)
        ctx.mark_dirty(y)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, grad):
        (out,) = ctx.saved_tensors
        g = relu_backward(grad.contiguous(), out) if ctx.relu else grad
        return g, None, None, (g if ctx.has_res else None)


# direct-convolution FLOPs (2 x MACs) of the calls that took a Winograd route since the last reset: such a call EXECUTES 2.25x fewer
# multiply-adds on the matrix cores than it is credited with (F(2x2,3x3): 16 products per 4 outputs instead of 36) - tools/bench_end_to_end.py
# turns this into executed FLOPs per step and a matrix-pipe utilisation bound next to the direct-equivalent roofline fraction
WINO_DIRECT_EQUIV_FLOPS = [0]
WINO4_DIRECT_EQUIV_FLOPS = [0]       # the same for the F(4x4,3x3) route (csrc/wino4.hip): 4x fewer multiply-adds than it is credited with


class _Conv2dChoice:
    """Per layer shape and direction: who computes it - this package's direct kernel ("hip"), its Winograd kernel ("wino") or torch's
    operator ("", MIOpen / rocBLAS + one fused element-wise pass).  The three sum in different float orders, so the choice is part of
    the result: it comes from the committed table ``routes_gfx950.json`` (routes.py; a fixed rule for shapes the table does not hold),
    the same in every process and on every rank.  Only ``ADV_ROUTES=measure`` (tools/make_routes.py, the per-layer benches) times the
    candidates, which is how the table is generated (round 3 timed at first use in every run: results differed run to run)."""

    @staticmethod
    def pick(key, fns):
        """the route name for ``key`` among ``fns`` (name -> callable)"""
        return routes.choose(key, fns)

    @staticmethod
    def get(key, hip_fn, torch_fn, wino_fn=None, wino4_fn=None):
        """-> "hip" (direct implicit GEMM), "wino" / "wino4" (Winograd F(2x2,3x3) / F(4x4,3x3) on the matrix cores, 3x3 layers) or "" (torch / MIOpen)"""
        fns = {"hip": hip_fn, "": torch_fn}
        if wino_fn is not None:
            fns["wino"] = wino_fn
        if wino4_fn is not None:
            fns["wino4"] = wino4_fn
        return routes.choose(key, fns)


class Conv2dAuto(torch.autograd.Function):
    """y = [relu](conv2d(x) [+ bias] [+ residual]) and its backward w.r.t. x and the residual, each direction computed by whichever of
    {csrc/conv2d.hip, torch} measured faster for this layer shape (see _Conv2dChoice).  ``prep`` = Conv2dPrep of the layer, ``weight``
    the original tensor for torch's side.

    The ReLU's backward without a pass of its own, for CHAINS (a -> this layer is a's only consumer):
      ``mask_input=True``   x is a ReLU output consumed by this layer alone: the gradient returned for x is already multiplied by
                            (x > 0) - in the dgrad kernel's epilogue when libadvengine computes it - i.e. it is the gradient w.r.t.
                            the producer's PRE-activation;
      ``relu="consumer"``   this layer applies the ReLU in its forward but does not mask in its backward: its only consumer was
                            called with mask_input=True and has done it (y > 0 there is the same mask).
    The caller vouches for the topology; the result is the same gradient, bit for bit (the mask multiplies the same float once).

    Skip connections without the autograd engine's addition:
      ``skip_out=True``     returns (y, x_skip) with x_skip an alias of x.  A residual block hands x_skip to its last layer as the
                            ``residual``; this layer's backward then receives BOTH gradients of x - through its own convolution and over
                            the skip path - and adds the second in the dgrad kernel's epilogue (before the mask of ``mask_input``, which
                            then covers both: x may be a ReLU output whose only consumers are this layer and its block's skip path)."""

    @staticmethod
    def forward(ctx, x, prep, weight, bias=None, residual=None, relu=False, mask_input=False, skip_out=False):
        x = x.contiguous()
        res = None if residual is None else residual.contiguous()
        pad, dil = prep.padding, prep.dilation
        do_relu = bool(relu)
        key = ("f", prep.k, prep.cin, prep.cout, dil, tuple(x.shape), res is not None, do_relu)

        def by_torch():                  # MIOpen's convolution + ONE element-wise pass (bias, skip connection, ReLU)
            return bias_act_(F.conv2d(x, weight, None, 1, pad, dil), bias, res, do_relu)

        def by_wino4():
            from .wino4 import conv_wino4
            return conv_wino4(x, prep.wino4(), bias, res, do_relu, splits=0)       # (small maps: the contraction dealt to several workgroups per tile)

        use = _Conv2dChoice.get(key, lambda: conv2d(x, prep, bias, res, do_relu), by_torch,
                                (lambda: conv2d(x, prep, bias, res, do_relu, wino=True)) if prep.has_wino else None, by_wino4 if prep.has_wino else None)
        y = by_wino4() if use == "wino4" else (conv2d(x, prep, bias, res, do_relu, wino=(use == "wino")) if use else by_torch())
        if use == "wino":
            WINO_DIRECT_EQUIV_FLOPS[0] += 18 * y.numel() * prep.cin
        elif use == "wino4":
            WINO4_DIRECT_EQUIV_FLOPS[0] += 18 * y.numel() * prep.cin
        ctx.prep, ctx.has_res, ctx.xshape = prep, res is not None, tuple(x.shape)
        ctx.mask_own = do_relu and relu != "consumer"          # mask the incoming gradient with y > 0 here
        ctx.mask_input = bool(mask_input)
        ctx.save_for_backward(weight, y if ctx.mask_own else None, x if mask_input else None)
        return (y, x) if skip_out else y

    @staticmethod
    def backward(ctx, grad_y, grad_skip=None):
        weight, y, x_in = ctx.saved_tensors
        prep = ctx.prep
        g = grad_y.contiguous()
        if ctx.mask_own:
            g = relu_backward(g, y)
        skip = None if grad_skip is None else grad_skip.contiguous()      # the gradient of x over the block's skip path
        key = ("b", prep.k, prep.cin, prep.cout, prep.dilation, ctx.xshape, ctx.mask_input, skip is not None)

        def by_torch():
            gx = torch.ops.aten.convolution_backward(g, _shape_only(ctx.xshape, g), weight, None, [1, 1], [prep.padding, prep.padding],
                                                     [prep.dilation, prep.dilation], False, [0, 0], 1, [True, False, False])[0]
            if skip is not None:
                gx = gx + skip
            return relu_backward(gx, x_in) if ctx.mask_input else gx

        hip = lambda wino=False: conv2d_dgrad(g, prep, residual=skip, mask=x_in if ctx.mask_input else None, wino=wino)      # noqa: E731

        def by_wino4():
            from .wino4 import conv_wino4_dgrad
            return conv_wino4_dgrad(g, prep.wino4(), residual=skip, mask=x_in if ctx.mask_input else None, splits=0)

        use = _Conv2dChoice.get(key, hip, by_torch, (lambda: hip(True)) if prep.has_wino else None, by_wino4 if prep.has_wino else None)
        gx = by_wino4() if use == "wino4" else (hip(use == "wino") if use else by_torch())
        if use == "wino":
            WINO_DIRECT_EQUIV_FLOPS[0] += 18 * gx.numel() * prep.cout
        elif use == "wino4":
            WINO4_DIRECT_EQUIV_FLOPS[0] += 18 * gx.numel() * prep.cout
        return gx, None, None, None, (g if ctx.has_res else None), None, None, None


_SHAPE_DUMMY = {}


def _shape_only(shape, like):
    """convolution_backward wants the input TENSOR although the gradient w.r.t. the input reads only its shape: one cached dummy per shape"""
    key = (tuple(shape), like.device)
    t = _SHAPE_DUMMY.get(key)
    if t is None:
        t = torch.empty(shape, dtype=torch.float32, device=like.device)
        _SHAPE_DUMMY[key] = t
    return t


__all__ = [n for n in dir() if not n.startswith("__")]
