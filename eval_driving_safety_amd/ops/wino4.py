"""3x3 and 3x3x3 stride-1 convolutions by Winograd F(4x4, 3x3) on the float32 matrix cores (csrc/wino4.hip): 36 products per 16 outputs.
Part of the ``ops`` package.  Its own order of float operations (oracle: ``oracle_c.conv_wino4``); ~1e-5 of the output's magnitude away
from the direct kernels and torch."""
from ._base import *       # noqa: F401,F403
from .conv2d import _like


class ConvWino4Prep:
    """a layer's weights [Cout,Cin,3,3] or [Cout,Cin,3,3,3] transformed for adv_conv{2,3}d_wino4_f32 (G g G^T per channel pair and depth
    tap), for the forward and for the backward w.r.t. the input; each made on first use, once"""

    def __init__(self, weight):
        wt = _feat(weight.detach().contiguous(), "weight")
        if tuple(wt.shape[2:]) not in ((3, 3), (3, 3, 3)):
            raise ValueError("weight must be [Cout,Cin,3,3] or [Cout,Cin,3,3,3]")
        self.cout, self.cin, self.three_d, self._wt, self._u = int(wt.shape[0]), int(wt.shape[1]), wt.dim() == 5, wt, {}

    def u(self, transpose):
        t = self._u.get(bool(transpose))
        if t is None:
            lib = _lib.load()
            n = int((lib.adv_conv3d_wino4_prep_floats if self.three_d else lib.adv_conv2d_wino4_prep_floats)(self.cout, self.cin, int(transpose)))
            t = torch.empty((n,), dtype=torch.float32, device=self._wt.device)
            with _on(self._wt):
                _lib.call("adv_conv3d_wino4_prep_weights_f32" if self.three_d else "adv_conv2d_wino4_prep_weights_f32", _ptr(self._wt), _ptr(t), self.cout,
                          self.cin, int(transpose), _stream(self._wt))
            self._u[bool(transpose)] = t
        return t


def _wino4_call(x, u, three_d, cin, cout, bias, residual, relu, mask, tile, splits=1):
    xi = _feat(x, "x")
    if xi.dim() != (5 if three_d else 4) or xi.shape[1] != cin:
        raise ValueError("x must be [B,%d,%sH,W]" % (cin, "D," if three_d else ""))
    y = torch.empty((xi.shape[0], cout) + tuple(xi.shape[2:]), dtype=torch.float32, device=xi.device)
    if bias is not None:
        bias = _feat(bias, "bias")
        if tuple(bias.shape) != (cout,):
            raise ValueError("bias must be [cout]")
    bp = None if bias is None else _ptr(bias)
    with _on(xi):
        if three_d:
            b, _, d, h, w = xi.shape
            _lib.call("adv_conv3d_wino4_f32", _ptr(xi), _ptr(u), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout, d, h, w,
                      int(bool(relu)), int(tile), _stream(xi))
        else:
            b, _, h, w = xi.shape
            if splits == 0:      # this library's rule for the layer (1: one workgroup per tile)
                splits = int(_lib.load().adv_conv2d_wino4_ksplit_pick(b, cin, cout, h, w))
            if splits > 1:       # the contraction dealt to `splits` workgroups per tile; the parts meet in a scratch tensor (the caching allocator's)
                scratch = torch.empty((splits,) + tuple(y.shape), dtype=torch.float32, device=xi.device)
                _lib.call("adv_conv2d_wino4_ksplit_f32", _ptr(xi), _ptr(u), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), _ptr(scratch),
                          b, cin, cout, h, w, int(bool(relu)), int(tile), int(splits), _stream(xi))
                return y
            _lib.call("adv_conv2d_wino4_f32", _ptr(xi), _ptr(u), bp, _like(residual, y, "residual"), _like(mask, y, "mask"), _ptr(y), b, cin, cout, h, w,
                      int(bool(relu)), int(tile), _stream(xi))
    return y


def conv_wino4(x, prep, bias=None, residual=None, relu=False, mask=None, tile=-1, splits=1):
    """conv(x, 3x3 or 3x3x3, stride 1, padding 1) (+ bias) (+ residual) (ReLU) (zeroed where mask <= 0) by the F(4x4,3x3) kernel.
    ``splits`` (2D layers): > 1 = the contraction dealt to that many workgroups per tile (small maps; its own - reproducible - order of
    additions: oracle ``conv_wino4(..., chunk=...)``), 0 = the library's rule for the layer, 1 = one workgroup per tile"""
    return _wino4_call(x, prep.u(False), prep.three_d, prep.cin, prep.cout, bias, residual, relu, mask, tile, splits)


def conv_wino4_dgrad(grad, prep, residual=None, mask=None, tile=-1, splits=1):
    """the backward w.r.t. the input of the same layer: grad [B,Cout,...] -> [B,Cin,...] (+ residual, masked)"""
    return _wino4_call(grad, prep.u(True), prep.three_d, prep.cout, prep.cin, None, residual, False, mask, tile, splits)


def conv_wino4_ksplit_chunk(cin, tile, splits):
    """input channels per part of a K-split launch (adv_conv2d_wino4_ksplit_chunk): what the oracle needs to restate its order"""
    return int(_lib.load().adv_conv2d_wino4_ksplit_chunk(int(cin), int(tile), int(splits)))


__all__ = [n for n in dir() if not n.startswith("__")]
