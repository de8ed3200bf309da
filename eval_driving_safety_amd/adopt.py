"""``adopt(model)``: put an UPSTREAM detector - a torch ``nn.Module`` with its trained weights - on libadvengine's kernels.

The attacks differentiate w.r.t. the images only; the detector is in eval mode and its weights are constants
(attack/DSGN/pgd_attack.py:140,300-336; attack/Stereo-RCNN/pgd_attack.py:99,151-174).  That is what makes this legal:

  * an eval-mode BatchNorm after a convolution is a per-channel affine map of its output: folded into the convolution's weights
    and bias once, here (``fold_bn``), the BatchNorm module becomes ``nn.Identity``;
  * the convolution module is replaced by one that calls this package's kernels with the REAL weights - ``nn.Conv2d`` 1x1 / 3x3
    stride 1 (dilation 1 or 2) on csrc/conv2d.hip + csrc/wino2d.hip through ``ops.Conv2dAuto`` (the committed route table decides
    which), ``nn.Conv3d`` 3x3x3 stride 1 / 2 and ``nn.ConvTranspose3d`` 3x3x3 stride 2 on csrc/conv3d.hip (``ops.Conv3dK3``,
    ``ops.Conv3dK3S2``, ``ops.ConvTranspose3dK3S2``); bias (the folded BatchNorm shift) and - where the graph is a plain
    ``nn.Sequential`` conv -> [bn] -> relu - the ReLU run in the kernel's epilogue;
  * a stride-2 1x1 layer (ResNet's down-sampling projections) is a sub-sampling copy + the 1x1 GEMM kernel;
  * layers without a kernel here (7x7 stems, strided 3x3 layers, grouped convolutions ...) keep torch's operator but still lose their
    BatchNorm (folded) and get bias + ReLU as ONE fused pass (``ops.bias_act_``);
  * gradients flow to the INPUT only: the adopted modules return no weight gradients (``adopt`` switches ``requires_grad`` off on
    what it adopts, as adapters._freeze does for the whole model).

Which BatchNorm belongs to which convolution is a fact of the model's ``forward`` code, which a module walk cannot see.  Two
conventions are recognised, both universal in the detectors this package serves (PSMNet / DSGN ``convbn`` / ``convbn_3d`` helpers,
torchvision-style ResNets as Stereo R-CNN's ``resnet.py``):

  sequential   inside an ``nn.Sequential``: ``conv, bn``  /  ``conv, bn, relu``  /  ``conv, relu`` as consecutive children;
  paired names a module with children ``convK`` and ``bnK`` (K = 1, 2, 3 ...; also ``conv``/``bn``) of matching width, the
               Bottleneck / BasicBlock convention (``fold_named_pairs=True``).  ReLUs are NOT fused there (one ``self.relu``
               module is shared by several layers).

``verify=(args...)`` runs the model on those inputs before and after and raises if any output moved by more than ``tol`` of its
magnitude - the guard against a model whose forward does not follow the convention.  Returns a report of what was replaced.
"""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F


def fold_bn(weight, bias, bn, transposed=False):
    """(weight, bias) of ``bn(conv(x))`` in eval mode as one convolution: w' = w * s[cout], b' = (b - mean) * s + beta with
    s = gamma / sqrt(var + eps).  ``transposed``: the weight layout is [Cin, Cout, ...] (ConvTranspose)."""
    if bn.training:
        raise ValueError("BatchNorm in training mode cannot be folded (the attacks run the detector in eval mode)")
    if not bn.track_running_stats or bn.running_mean is None:
        raise ValueError("BatchNorm without running statistics cannot be folded")
    from .checkpoints import fold_bn_tensors
    gamma = bn.weight.detach() if bn.affine else torch.ones_like(bn.running_var)
    beta = bn.bias.detach() if bn.affine else torch.zeros_like(bn.running_var)
    w, b = fold_bn_tensors(weight.detach(), None if bias is None else bias.detach(), gamma, beta, bn.running_mean.detach(), bn.running_var.detach(),
                           bn.eps, transposed)
    return w.to(weight.dtype), b.to(weight.dtype)


class Call:
    """``verify=Call(args, kwargs)``: a forward call with keyword arguments (DSGN's ``calibs_Proj_R=``)"""

    def __init__(self, args, kwargs=None):
        self.args, self.kwargs = tuple(args), dict(kwargs or {})


def _run(model, verify):
    if isinstance(verify, Call):
        return model(*verify.args, **verify.kwargs)
    return model(*verify)


def _rng_snapshot():
    """python / numpy / torch (CPU and every CUDA device) generator states: an upstream forward may SAMPLE (Stereo R-CNN's proposal-target
    layer draws its rois), so the two verification runs must start from one state - and leave the caller's state as they found it"""
    import random
    st = {"py": random.getstate(), "torch": torch.random.get_rng_state()}
    try:
        import numpy as np
        st["np"] = np.random.get_state()
    except Exception:                             # noqa: BLE001 - numpy is optional here
        pass
    if torch.cuda.is_available():
        st["cuda"] = torch.cuda.get_rng_state_all()
    return st


def _rng_restore(st):
    import random
    random.setstate(st["py"])
    torch.random.set_rng_state(st["torch"])
    if "np" in st:
        import numpy as np
        np.random.set_state(st["np"])
    if "cuda" in st:
        torch.cuda.set_rng_state_all(st["cuda"])


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _numeric_padding(conv):
    """``padding="valid"`` / ``"same"`` (stride 1, an even (k - 1) * dilation per axis) as the numbers torch would use; else the string"""
    p = conv.padding
    if not isinstance(p, str):
        return tuple(p)
    if p == "valid":
        return tuple(0 for _ in conv.kernel_size)
    tot = [d * (k - 1) for k, d in zip(conv.kernel_size, conv.dilation)]
    if p == "same" and all(int(s) == 1 for s in conv.stride) and all(t % 2 == 0 for t in tot):
        return tuple(t // 2 for t in tot)
    return p


def _same(v):
    """all entries of an int tuple equal -> that int, else None"""
    v = tuple(v) if isinstance(v, (tuple, list)) else (v,)
    return int(v[0]) if all(int(q) == int(v[0]) for q in v) else None


class _Adopted(nn.Module):
    """common state of an adopted layer: folded weight + bias as buffers (they follow ``.to(device)``), lazily prepared kernel layouts"""

    kind = "?"

    def __init__(self, weight, bias, relu):
        super().__init__()
        self.register_buffer("weight", weight.detach().clone().contiguous())
        self.register_buffer("bias", None if bias is None else bias.detach().clone().contiguous())
        self.relu = bool(relu)
        self._prep, self._prep_dev = None, None

    def _ops(self, x):
        if not x.is_cuda:
            raise RuntimeError("%s: libadvengine has no CPU path - move the adopted model and its inputs to the ROCm device" % type(self).__name__)
        from . import ops
        return ops

    def _prepared(self, make):
        if self._prep is None or self._prep_dev != self.weight.device:
            self._prep, self._prep_dev = make(), self.weight.device
        return self._prep

    def _torch_epilogue(self, y, ops):
        """bias + ReLU behind a convolution torch computed, one fused pass; through ops.BiasAct so that the ReLU has its backward"""
        if self.bias is None and not self.relu:
            return y
        if not y.requires_grad:
            return ops.bias_act_(y, self.bias, None, self.relu)
        return ops.BiasAct.apply(y.contiguous(), self.bias, self.relu)


class AdoptedConv2d(_Adopted):
    """``[relu](conv2d(x, w', stride, padding, dilation) + b')`` with the folded weights; 1x1 and 3x3 stride-1 layers on libadvengine"""

    def __init__(self, conv, weight, bias, relu=False):
        super().__init__(weight, bias, relu)
        # (``padding`` may be the string "same" / "valid": such a layer keeps torch's operator, which takes the string as it is)
        self.stride, self.dilation, self.groups = _pair(conv.stride), _pair(conv.dilation), conv.groups
        self.padding = _numeric_padding(conv)
        self.padding_mode = conv.padding_mode
        k = tuple(weight.shape[2:])
        s, p, d = _same(self.stride), (None if isinstance(self.padding, str) else _same(self.padding)), _same(self.dilation)
        self.subsample = self.groups == 1 and self.padding_mode == "zeros" and s == 2 and k == (1, 1) and p == 0 and d == 1      # strided 1x1: sub-sample + GEMM
        self.native = self.subsample or (self.groups == 1 and self.padding_mode == "zeros" and s == 1 and k[0] == k[1] and
                                         ((k[0] == 1 and p == 0 and d == 1) or (k[0] == 3 and d in (1, 2) and p == d)))
        self.kind = "conv2d %dx%d s%s d%s %d->%d%s" % (k[0], k[1], s, d, weight.shape[1] * self.groups, weight.shape[0], "" if self.native else " (torch + fused epilogue)")

    def forward(self, x):
        ops = self._ops(x)
        stride = self.stride
        if self.subsample and x.dtype == torch.float32:
            x = x[:, :, ::2, ::2].contiguous()          # every other pixel of every other row: what a stride-2 1x1 layer reads
            stride = (1, 1)                             # (also for torch's operator below, should the kernel refuse the sub-sampled map)
        elif self.subsample:
            return self._torch_epilogue(F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups), ops)
        if self.native and ops.conv2d_supported(x, self.weight, 1, self.padding[0], self.dilation[0]):
            prep = self._prepared(lambda: ops.Conv2dPrep(self.weight, 1, self.padding[0], self.dilation[0]))
            return ops.Conv2dAuto.apply(x, prep, self.weight, self.bias, None, self.relu)
        if self.padding_mode != "zeros":
            raise RuntimeError("padding_mode %r is not adopted" % self.padding_mode)
        return self._torch_epilogue(F.conv2d(x, self.weight, None, stride, self.padding, self.dilation, self.groups), ops)


class AdoptedConv3d(_Adopted):
    """3x3x3 / padding 1 / stride 1 or 2 on the float32 matrix cores (csrc/conv3d.hip; the Winograd route where the table says so)"""

    def __init__(self, conv, weight, bias, relu=False):
        super().__init__(weight, bias, relu)
        self.stride, self.dilation, self.groups = tuple(conv.stride), tuple(conv.dilation), conv.groups
        self.padding = _numeric_padding(conv)
        cout, cin = weight.shape[:2]
        s = _same(self.stride)
        self.native = (tuple(weight.shape[2:]) == (3, 3, 3) and self.groups == 1 and not isinstance(self.padding, str) and _same(self.padding) == 1 and _same(self.dilation) == 1 and
                       conv.padding_mode == "zeros" and ((s == 1 and (cin % 4 == 0 or cin < 4) and (cout % 4 == 0 or cout < 4)) or
                                                         (s == 2 and cout % 4 == 0 and cin % 4 == 0)))
        self.kind = "conv3d 3x3x3 s%s %d->%d%s" % (s, cin, cout, "" if self.native else " (torch + fused epilogue)")

    def forward(self, x):
        ops = self._ops(x)
        cout = self.weight.shape[0]
        if not self.native or x.dtype != torch.float32:
            return self._torch_epilogue(F.conv3d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups), ops)
        if self.stride[0] == 1:
            def make():
                wino = ops.Conv3dWinoPrep(self.weight) if cout >= 4 else None
                return ops.conv3d_k3_prep(self.weight), ops.conv3d_k3_prep(self.weight, transpose=True), wino
            p, pt, wino = self._prepared(make)
            if cout < 4:                    # the narrow (score) layers: vector-ALU kernels without an epilogue
                y = ops.Conv3dK3.apply(x, p, pt, cout)
                return self._torch_epilogue(y, ops)
            return ops.Conv3dK3.apply(x, p, pt, cout, None, self.bias, self.relu, None, False, wino)
        p, pt = self._prepared(lambda: (ops.conv3d_k3_s2_prep(self.weight), ops.conv_transpose3d_k3_s2_prep(self.weight)))
        return ops.Conv3dK3S2.apply(x, p, pt, cout, self.bias, self.relu)


class AdoptedConvTranspose3d(_Adopted):
    """3x3x3 / stride 2 / padding 1 / output_padding 1 (the hourglass up-sampling layer) on csrc/conv3d.hip"""

    def __init__(self, conv, weight, bias, relu=False):
        super().__init__(weight, bias, relu)
        self.stride, self.padding, self.output_padding = tuple(conv.stride), tuple(conv.padding), tuple(conv.output_padding)
        self.dilation, self.groups = tuple(conv.dilation), conv.groups
        cin, cout = weight.shape[:2]
        self.native = (tuple(weight.shape[2:]) == (3, 3, 3) and self.groups == 1 and _same(self.stride) == 2 and _same(self.padding) == 1 and
                       _same(self.output_padding) == 1 and _same(self.dilation) == 1 and cin % 4 == 0 and cout % 4 == 0)
        self.kind = "conv_transpose3d 3x3x3 s2 %d->%d%s" % (cin, cout, "" if self.native else " (torch + fused epilogue)")

    def forward(self, x):
        ops = self._ops(x)
        if not self.native or x.dtype != torch.float32:
            return self._torch_epilogue(F.conv_transpose3d(x, self.weight, None, self.stride, self.padding, self.output_padding, self.groups, self.dilation), ops)
        p, pt = self._prepared(lambda: (ops.conv_transpose3d_k3_s2_prep(self.weight), ops.conv3d_k3_s2_prep(self.weight)))
        return ops.ConvTranspose3dK3S2.apply(x, p, pt, self.weight.shape[1], self.bias, self.relu)


_CONVS = {nn.Conv2d: AdoptedConv2d, nn.Conv3d: AdoptedConv3d, nn.ConvTranspose3d: AdoptedConvTranspose3d}
_BNS = {nn.Conv2d: nn.BatchNorm2d, nn.Conv3d: nn.BatchNorm3d, nn.ConvTranspose3d: nn.BatchNorm3d}


def _width(conv):
    return conv.out_channels


def _make(conv, bn, relu):
    transposed = isinstance(conv, nn.ConvTranspose3d)
    if bn is not None:
        w, b = fold_bn(conv.weight, conv.bias, bn, transposed)
    else:
        w, b = conv.weight.detach(), (None if conv.bias is None else conv.bias.detach())
    return _CONVS[type(conv)](conv, w, b, relu)


def adopt(model, fuse_relu=True, fold_named_pairs=True, verify=None, tol=1e-4, freeze=True, functional=True):
    """Replace, IN PLACE, the convolutions of ``model`` by libadvengine-backed modules carrying the model's own weights (see the module
    docstring).  ``functional``: also bind the two functional patterns of a DSGN forward - 5-D ``F.grid_sample`` and the
    ``F.interpolate(trilinear)`` -> softmax -> depth-weighted sum chain - to ``ops.GridSample3d`` / ``ops.DepthRegress`` by putting a proxy
    where the defining Python modules hold ``torch.nn.functional`` (adopt_functional.py; ``adopt_functional.unbind()`` undoes it).
    -> {"replaced": [(qualified name, what)], "folded_bn": n, "fused_relu": n, "kept": [(name, why)], "upsample_add": modules whose
    ``_upsample_add`` now has a deterministic backward, "functional": [(python module, global name)] rebound}."""
    if model.training:
        raise ValueError("adopt() needs the model in eval mode (BatchNorm statistics are folded)")
    before = rng = None
    if verify is not None:
        rng = _rng_snapshot()
        with torch.no_grad():
            before = _flat_outputs(_run(model, verify))
        _rng_restore(rng)
    report = {"replaced": [], "folded_bn": 0, "fused_relu": 0, "kept": []}
    for parent_name, parent in list(model.named_modules()):
        children = list(parent.named_children())
        taken = set()
        if isinstance(parent, nn.Sequential):
            i = 0
            while i < len(children):
                name, m = children[i]
                if type(m) in _CONVS:
                    bn = relu = None
                    j = i + 1
                    if j < len(children) and type(children[j][1]) is _BNS[type(m)] and children[j][1].num_features == _width(m):
                        bn, j = children[j], j + 1
                    if fuse_relu and j < len(children) and type(children[j][1]) is nn.ReLU:
                        relu, j = children[j], j + 1
                    setattr(parent, name, _make(m, None if bn is None else bn[1], relu is not None))
                    for extra in (bn, relu):
                        if extra is not None:
                            setattr(parent, extra[0], nn.Identity())
                            taken.add(extra[0])
                    report["folded_bn"] += bn is not None
                    report["fused_relu"] += relu is not None
                    report["replaced"].append((_q(parent_name, name), getattr(parent, name).kind))
                    taken.add(name)
                    i = j
                else:
                    i += 1
            continue
        by_name = dict(children)
        for name, m in children:
            if type(m) not in _CONVS or name in taken:
                continue
            bn_name = None
            if fold_named_pairs:
                mt = re.fullmatch(r"conv(\d*)", name)
                cand = ("bn" + mt.group(1)) if mt else None
                if cand in by_name and type(by_name[cand]) is _BNS[type(m)] and by_name[cand].num_features == _width(m):
                    bn_name = cand
            setattr(parent, name, _make(m, by_name[bn_name] if bn_name else None, False))
            if bn_name:
                setattr(parent, bn_name, nn.Identity())
                report["folded_bn"] += 1
            report["replaced"].append((_q(parent_name, name), getattr(parent, name).kind))
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            report["kept"].append((name, type(m).__name__ + ": no libadvengine counterpart for this module type here"))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            report["kept"].append((name, "BatchNorm not next to a convolution by either convention: left as it is"))
    # the FPN top-down path of the reference's detector class: ``_upsample_add(x, y)`` = F.interpolate(x, size=y's, mode='bilinear',
    # align_corners=False) + y (attack/Stereo-RCNN/stereo_rcnn.py:91-108).  torch's backward of that up-sampling scatters with atomicAdd -
    # the image gradient then differs in its last bits from run to run; ops.BilinearUp is the same operator with a fixed-order gather
    # backward.  Rebound on every module that defines the method (CPU tensors keep torch's operator).
    report["upsample_add"] = 0
    for m in model.modules():
        if callable(getattr(type(m), "_upsample_add", None)):
            m._upsample_add = _upsample_add
            report["upsample_add"] += 1
    report["functional"] = []
    if functional:
        from . import adopt_functional
        report["functional"] = adopt_functional.bind(model)
    if freeze:
        for p in model.parameters():
            p.requires_grad_(False)
    if before is not None:
        with torch.no_grad():
            after = _flat_outputs(_run(model, verify))
        _rng_restore(rng)                         # the caller's generators are where they were before adopt()
        if len(after) != len(before):
            raise RuntimeError("adopt(): the model returns %d tensors after adoption, %d before" % (len(after), len(before)))
        # Element by element an adopted detector stays within ``tol`` of its output's magnitude EXCEPT behind discrete decisions that a
        # last-bit difference can flip (a ReLU at zero, an NMS tie, a box just inside a threshold): those few elements move by whole
        # values in torch-CPU vs torch-GPU too (tests/test_upstream_binding.py measures it).  So the guard counts: at most ``outliers``
        # of an output's elements may be off by more than ``tol`` of its magnitude, and the output as a whole by 10 x tol in L2.
        outliers, moved = 1e-3, []
        for k, (a, b) in enumerate(zip(before, after)):
            if a.shape != b.shape:
                raise RuntimeError("adopt(): output %d has shape %s after adoption, %s before (a data-dependent shape: a discrete decision "
                                   "flipped, or the forward samples without the generators this check restores)" % (k, tuple(b.shape), tuple(a.shape)))
            if not a.numel():
                continue
            a64, d = a.double(), (a.double() - b.double()).abs()
            scale = float(a64.abs().max())
            share = float((d > tol * max(scale, 1e-30)).double().mean())
            l2 = float(d.norm() / a64.norm().clamp_min(1e-30))
            moved.append({"output": k, "max": float(d.max()), "magnitude": scale, "share_beyond_tol": share, "relative_l2": l2})
            if share > outliers or not l2 <= 10 * tol:
                raise RuntimeError("adopt(): output %d of the adopted model differs from the original's: %.3g of its elements by more than %.1e of its "
                                   "magnitude %.3g (allowed %.1e), relative L2 %.3g (allowed %.1e), worst element %.3g.  Either a BatchNorm was folded "
                                   "into a convolution it does not follow in this model's forward (pass fold_named_pairs=False, or adopt sub-modules), or "
                                   "the forward is not a function of its inputs alone (state, or sampling from a generator this check does not know)"
                                   % (k, share, tol, scale, outliers, l2, 10 * tol, float(d.max())))
        report["verified_outputs"] = len(after)
        report["verified"] = moved
    return report


def _upsample_add(x, y):
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        from . import ops
        return ops.BilinearUp.apply(x, tuple(y.shape[2:])) + y
    return F.interpolate(x, size=tuple(y.shape[2:]), mode="bilinear", align_corners=False) + y


def _q(parent, name):
    return (parent + "." if parent else "") + name


def _flat_outputs(out):
    """every float tensor in a (nested) tuple / list / dict of outputs, in a fixed order"""
    flat = []
    if isinstance(out, torch.Tensor):
        if out.is_floating_point():
            flat.append(out.detach().clone())
    elif isinstance(out, dict):
        for k in sorted(out, key=str):
            flat += _flat_outputs(out[k])
    elif isinstance(out, (tuple, list)):
        for v in out:
            flat += _flat_outputs(v)
    return flat
