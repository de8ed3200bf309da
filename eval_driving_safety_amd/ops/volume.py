"""after the 3D convolutions: depth regression, 5-D grid sampling, bilinear up-sampling, bird's-eye-view fold, focal loss (csrc/volume.hip, csrc/resize.hip).  Part of the ``ops`` package (split by kernel family from the former one-module ops.py; ``from eval_driving_safety_amd import ops``
still gives every name)."""
from ._base import *       # noqa: F401,F403  (torch, F, ctypes, _lib, routes, Space and the argument helpers)

# --------------------------------------------------------------------------------------------
# after the 3D convolutions (csrc/volume.hip): fused depth regression, grid_sample on 5-D volumes, sigmoid focal loss
def depth_regress(cost, depth_values, out_size, align_corners=False, with_stats=False):
    """cost [B,D,h,w], depth_values [Do] -> depth [B,H,W] = sum_k softmax_k(trilinear_upsample(cost, (Do,H,W)))[k] * depth_values[k],
    without ever writing the up-sampled volume.  with_stats: also the per-pixel softmax (max, sum) [B,2,H,W] the backward needs."""
    ci, zv = _feat(cost, "cost"), _feat(depth_values, "depth_values")
    if ci.dim() != 4 or zv.dim() != 1 or len(out_size) != 3 or zv.shape[0] != int(out_size[0]):
        raise ValueError("cost must be [B,D,h,w], out_size (Do,H,W) and depth_values [Do]")
    b, d, h, w = ci.shape
    do, ho, wo = (int(v) for v in out_size)
    depth = torch.empty((b, ho, wo), dtype=torch.float32, device=ci.device)
    stats = torch.empty((b, 2, ho, wo), dtype=torch.float32, device=ci.device) if with_stats else None
    with _on(ci):
        _lib.call("adv_depth_regress_f32", _ptr(ci), _ptr(zv), _ptr(depth), None if stats is None else _ptr(stats), b, d, h, w, do, ho, wo,
                  int(bool(align_corners)), _stream(ci))
    return (depth, stats) if with_stats else depth


def depth_regress_bwd(cost, depth_values, depth, stats, grad_depth, align_corners=False):
    ci, zv, dp, st, g = (_feat(cost, "cost"), _feat(depth_values, "depth_values"), _feat(depth, "depth"), _feat(stats, "stats"),
                         _feat(grad_depth, "grad_depth"))
    b, d, h, w = ci.shape
    _, ho, wo = dp.shape
    if tuple(st.shape) != (b, 2, ho, wo) or tuple(g.shape) != (b, ho, wo):
        raise ValueError("stats must be [B,2,H,W] and grad_depth [B,H,W]")
    work = torch.empty((b, d, ho, wo), dtype=torch.float32, device=ci.device)
    gc = torch.empty_like(ci)
    with _on(ci):
        _lib.call("adv_depth_regress_bwd_f32", _ptr(ci), _ptr(zv), _ptr(dp), _ptr(st), _ptr(g), _ptr(work), _ptr(gc), b, d, h, w, int(zv.shape[0]),
                  ho, wo, int(bool(align_corners)), _stream(ci))
    return gc


class DepthRegress(torch.autograd.Function):
    """depth = DepthRegress.apply(cost [B,D,h,w], depth_values [Do], (Do,H,W), align_corners); gradient w.r.t. cost only"""

    @staticmethod
    def forward(ctx, cost, depth_values, out_size, align_corners=False):
        cost = cost.contiguous()
        depth, stats = depth_regress(cost, depth_values, out_size, align_corners, with_stats=True)
        ctx.save_for_backward(cost, depth_values, depth, stats)
        ctx.align = bool(align_corners)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        cost, zv, depth, stats = ctx.saved_tensors
        return depth_regress_bwd(cost, zv, depth, stats, grad_depth.contiguous(), ctx.align), None, None, None


def grid_sample3d(vol, grid, align_corners=False):
    """torch.nn.functional.grid_sample(vol [B,C,D,H,W], grid [B,Z,Y,X,3], bilinear, zeros) - the same bits as torch on the CPU"""
    vi, gi = _feat(vol, "vol"), _feat(grid, "grid")
    if vi.dim() != 5 or gi.dim() != 5 or gi.shape[0] != vi.shape[0] or gi.shape[4] != 3:
        raise ValueError("vol must be [B,C,D,H,W] and grid [B,Z,Y,X,3]")
    b, c, d, h, w = vi.shape
    zo, yo, xo = gi.shape[1:4]
    out = torch.empty((b, c, zo, yo, xo), dtype=torch.float32, device=vi.device)
    with _on(vi):
        _lib.call("adv_grid_sample3d_f32", _ptr(vi), _ptr(gi), _ptr(out), b, c, d, h, w, zo, yo, xo, int(bool(align_corners)), _stream(vi))
    return out


class GridSamplePlan:
    """The backward's gather plan for one grid (build once per calibration): for every cell of the volume the sorted list of
    (output voxel, weight) that sample it."""

    def __init__(self, grid, vol_dims, align_corners=False):
        gi = _feat(grid, "grid")
        b, zo, yo, xo, _ = gi.shape
        d, h, w = (int(v) for v in vol_dims)
        nbytes = int(_lib.load().adv_grid_sample3d_plan_bytes(b, d, h, w, zo, yo, xo))
        if nbytes <= 0:
            raise ValueError("grid / volume too large for a 32-bit plan")
        self.buf = torch.empty((nbytes // 4,), dtype=torch.int32, device=gi.device)
        self.dims, self.out_dims, self.batch, self.align = (d, h, w), (zo, yo, xo), b, bool(align_corners)
        with _on(gi):
            _lib.call("adv_grid_sample3d_plan_f32", _ptr(gi), _ptr(self.buf), b, d, h, w, zo, yo, xo, int(self.align), _stream(gi))


def grid_sample3d_bwd(grad_out, plan, channels_last=True):
    """gradient w.r.t. the sampled volume; ``channels_last``: through a channels-last copy of grad_out (one run of C floats per list
    entry instead of C cache lines) - the same bits, less HBM traffic"""
    g = _feat(grad_out, "grad_out")
    b, c = g.shape[:2]
    if g.dim() != 5 or b != plan.batch or tuple(g.shape[2:]) != tuple(plan.out_dims):
        raise ValueError("grad_out must be [B,C,Z,Y,X] of the plan's grid")
    d, h, w = plan.dims
    zo, yo, xo = plan.out_dims
    gv = torch.empty((b, c, d, h, w), dtype=torch.float32, device=g.device)
    with _on(g):
        if channels_last:
            work = torch.empty((int(_lib.load().adv_grid_sample3d_bwd_workspace_floats(b, c, zo, yo, xo)),), dtype=torch.float32, device=g.device)
            _lib.call("adv_grid_sample3d_bwd_ws_f32", _ptr(g), _ptr(plan.buf), _ptr(gv), _ptr(work), b, c, d, h, w, zo, yo, xo, _stream(g))
        else:
            _lib.call("adv_grid_sample3d_bwd_f32", _ptr(g), _ptr(plan.buf), _ptr(gv), b, c, d, h, w, zo, yo, xo, _stream(g))
    return gv


class GridSample3d(torch.autograd.Function):
    """out = GridSample3d.apply(vol, grid, plan): gradient w.r.t. vol only (the grid is a function of the calibration), as a
    deterministic gather over ``plan = GridSamplePlan(grid, vol.shape[2:], align_corners)``"""

    @staticmethod
    def forward(ctx, vol, grid, plan):
        ctx.plan = plan
        return grid_sample3d(vol.contiguous(), grid, plan.align)

    @staticmethod
    def backward(ctx, grad_out):
        return grid_sample3d_bwd(grad_out.contiguous(), ctx.plan), None, None


def bilinear_up(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for x [B,C,h,w] (csrc/resize.hip)"""
    xi = _feat(x, "x")
    if xi.dim() != 4 or len(size) != 2 or min(size) < 1:
        raise ValueError("x must be [B,C,h,w] and size (ho, wo)")
    b, c, h, w = xi.shape
    out = torch.empty((b, c, int(size[0]), int(size[1])), dtype=torch.float32, device=xi.device)
    with _on(xi):
        _lib.call("adv_bilinear_up_f32", _ptr(xi), _ptr(out), b * c, h, w, int(size[0]), int(size[1]), _stream(xi))
    return out


def bilinear_up_bwd(grad_out, in_hw):
    """the adjoint of bilinear_up as a fixed-order gather: reproducible bit for bit (torch's backward scatters with atomicAdd)"""
    g = _feat(grad_out, "grad_out")
    if g.dim() != 4:
        raise ValueError("grad_out must be [B,C,ho,wo]")
    b, c, ho, wo = g.shape
    gin = torch.empty((b, c, int(in_hw[0]), int(in_hw[1])), dtype=torch.float32, device=g.device)
    with _on(g):
        _lib.call("adv_bilinear_up_bwd_f32", _ptr(g), _ptr(gin), b * c, int(in_hw[0]), int(in_hw[1]), ho, wo, _stream(g))
    return gin


class BilinearUp(torch.autograd.Function):
    """the FPN top-down path's up-sampling (attack/Stereo-RCNN/stereo_rcnn.py:92-108) with a deterministic backward"""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = tuple(x.shape[2:])
        return bilinear_up(x.contiguous(), size)

    @staticmethod
    def backward(ctx, grad):
        return bilinear_up_bwd(grad.contiguous(), ctx.in_hw), None


def bev_fold(v, pool):
    """[B,C,Z,Y,X] -> [B, C * (Y // pool), Z, X]: F.avg_pool3d(v, (1, pool, 1)).permute(0, 1, 3, 2, 4).reshape(...) in one pass"""
    vi = _feat(v, "v")
    if vi.dim() != 5 or pool < 1 or pool > vi.shape[3]:
        raise ValueError("v must be [B,C,Z,Y,X] with Y >= pool")
    b, c, z, y, x = vi.shape
    out = torch.empty((b, c * (y // pool), z, x), dtype=torch.float32, device=vi.device)
    with _on(vi):
        _lib.call("adv_bev_fold_f32", _ptr(vi), _ptr(out), b, c, z, y, x, int(pool), _stream(vi))
    return out


def bev_fold_bwd(grad_out, shape, pool, mask=None):
    """the fold's backward; ``mask`` (the forward's input, a ReLU output) zeroes the gradient where it is <= 0"""
    g = _feat(grad_out, "grad_out")
    b, c, z, y, x = shape
    if tuple(g.shape) != (b, c * (y // pool), z, x):
        raise ValueError("grad_out must be [B, C * (Y // pool), Z, X]")
    if mask is not None:
        mask = _feat(mask, "mask")
        if tuple(mask.shape) != tuple(shape):
            raise ValueError("mask must be laid out like the fold's input")
    gv = torch.empty(tuple(shape), dtype=torch.float32, device=g.device)
    with _on(g):
        _lib.call("adv_bev_fold_bwd_f32", _ptr(g), None if mask is None else _ptr(mask), _ptr(gv), b, c, z, y, x, int(pool), _stream(g))
    return gv


class BevFold(torch.autograd.Function):
    """the bird's-eye-view fold of the 3D geometric volume (height pooled by ``pool`` and folded into the channels), forward and backward
    one HBM-bound pass each (csrc/volume.hip) instead of torch's pooling kernel + permuting copy and their two backward passes"""

    @staticmethod
    def forward(ctx, v, pool, mask_input=False):
        """``mask_input``: v is a ReLU output this fold alone consumes and its producer left the mask to us (relu="consumer"): the gradient
        returned is already multiplied by (v > 0)"""
        ctx.shape, ctx.pool = tuple(v.shape), int(pool)
        v = v.contiguous()
        ctx.save_for_backward(v if mask_input else None)
        return bev_fold(v, pool)

    @staticmethod
    def backward(ctx, grad_out):
        (v,) = ctx.saved_tensors
        return bev_fold_bwd(grad_out.contiguous(), ctx.shape, ctx.pool, mask=v), None, None


def sigmoid_focal_loss(logits, targets, gamma=2.0, alpha=0.25, want_grad=False):
    """logits [N,K], targets int32 [N] (0 = background, c = class c, < 0 = ignored) -> per-element loss [N,K] (and d loss / d logit)"""
    li = _feat(logits, "logits")
    if li.dim() != 2 or not (isinstance(targets, torch.Tensor) and targets.is_cuda and targets.dtype == torch.int32 and targets.is_contiguous()
                             and tuple(targets.shape) == (li.shape[0],)):
        raise ValueError("logits must be [N,K] float32 and targets [N] int32, contiguous, on the GPU")
    loss = torch.empty_like(li)
    grad = torch.empty_like(li) if want_grad else None
    with _on(li):
        _lib.call("adv_sigmoid_focal_loss_f32", _ptr(li), _ptr(targets), _ptr(loss), None if grad is None else _ptr(grad), int(li.shape[0]),
                  int(li.shape[1]), float(gamma), float(alpha), _stream(li))
    return (loss, grad) if want_grad else loss


class SigmoidFocalLoss(torch.autograd.Function):
    """sum of the per-element focal losses (the reduction the FCOS-style heads use before dividing by the positive count)"""

    @staticmethod
    def forward(ctx, logits, targets, gamma=2.0, alpha=0.25):
        loss, grad = sigmoid_focal_loss(logits.contiguous(), targets, gamma, alpha, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.sum()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


__all__ = [n for n in dir() if not n.startswith("__")]
