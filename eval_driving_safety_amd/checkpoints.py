"""Upstream checkpoints onto this package's layer-list graphs.

The reference loads ``state_dict['state_dict']`` of ``finetune_53.tar`` into ``nn.DataParallel(StereoNet(cfg))``
(attack/DSGN/pgd_attack.py:142-145) and ``checkpoint['model']`` (+ ``checkpoint['uncert']``) of ``stereo_rcnn_12_6477.pth`` into
``resnet(classes, 101)`` (attack/Stereo-RCNN/pgd_attack.py:94-97).  ``adapters.DsgnShapedAdapter`` and ``surrogates.StereoRcnnR101``
hold the same layer lists with batch-norms FOLDED; the loaders here take an upstream key layout, fold each convolution's BatchNorm
(weight, bias, running_mean, running_var - eval mode) into it, check every shape and copy.  A key map is a dict

    our layer name -> (upstream convolution prefix, upstream BatchNorm prefix or None)

``STEREO_RCNN_KEYS`` follows the module names the reference's own substitute files use (attack/Stereo-RCNN/stereo_rcnn.py:69-85,
157-171; stereo_rpn.py:32-40) with torchvision's Bottleneck naming for ``RCNN_layer1..4``; ``DSGN_KEYS`` follows the PSMNet / DSGN
family's ``feature_extraction.* / dres0 / dres1 / hourglass`` naming.  The upstream repositories are not in the reference tree, so both
maps are [UPSTREAM-UNVERIFIED] where they go beyond names the reference itself spells out: a load reports exactly which upstream keys
were consumed, which were left over and which of our layers found nothing or a different shape, and ``strict=True`` (default) raises on
any of them - a silent partial load is not possible.  Pass your own map for a checkout whose names differ.
"""
import torch


def fold_bn_tensors(w, b, gamma, beta, mean, var, eps, transposed=False):
    """bn(conv(x)) in eval mode as one convolution: w' = w * s, b' = (b - mean) * s + beta, s = gamma / sqrt(var + eps) (float64 inside)"""
    s = gamma.double() / torch.sqrt(var.double() + eps)
    shape = [1] * w.dim()
    shape[1 if transposed else 0] = -1
    b0 = torch.zeros_like(mean, dtype=torch.float64) if b is None else b.double()
    return (w.double() * s.view(shape)).float().contiguous(), ((b0 - mean.double()) * s + beta.double()).float().contiguous()


def _strip(state):
    """``module.`` (nn.DataParallel, pgd_attack.py:137) off every key"""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}


class _Loader:
    def __init__(self, state, eps=1e-5):
        self.state, self.eps = _strip(state), eps
        self.used, self.problems = set(), []

    def take(self, key):
        v = self.state.get(key)
        if v is not None:
            self.used.add(key)
        return v

    def conv(self, ours, conv_prefix, bn_prefix, want_w, want_b, transposed=False):
        """-> (weight, bias) folded, shaped like ``want_w`` / ``want_b`` (shapes), or None after recording the problem"""
        w = self.take(conv_prefix + ".weight")
        if w is None:
            self.problems.append("%s: no %s.weight in the checkpoint" % (ours, conv_prefix))
            return None
        b = self.take(conv_prefix + ".bias")
        if bn_prefix is not None:
            parts = [self.take(bn_prefix + s) for s in (".weight", ".bias", ".running_mean", ".running_var")]
            self.take(bn_prefix + ".num_batches_tracked")
            if any(p is None for p in parts):
                self.problems.append("%s: incomplete BatchNorm %s.*" % (ours, bn_prefix))
                return None
            w, b = fold_bn_tensors(w, b, *parts, self.eps, transposed)
        if tuple(w.shape) != tuple(want_w):
            self.problems.append("%s: %s.weight is %s, the layer list has %s" % (ours, conv_prefix, tuple(w.shape), tuple(want_w)))
            return None
        if b is None:
            b = torch.zeros(want_b)
        if tuple(b.shape) != tuple(want_b):
            self.problems.append("%s: bias is %s, the layer list has %s" % (ours, tuple(b.shape), tuple(want_b)))
            return None
        return w.float(), b.float()

    def report(self, loaded, strict, what):
        left = sorted(k for k in self.state if k not in self.used and not k.endswith("num_batches_tracked"))
        rep = {"loaded_layers": loaded, "problems": list(self.problems), "unused_upstream_keys": left}
        if strict and (self.problems or left):
            raise ValueError("%s: the checkpoint does not match the layer list - %d problem(s), %d upstream key(s) left over:\n  %s" %
                             (what, len(self.problems), len(left), "\n  ".join(self.problems[:12] + left[:12])))
        return rep


# ------------------------------------------------------------------------------------------------ Stereo R-CNN (ResNet-101-FPN)
def stereo_rcnn_keys(blocks=(3, 4, 23, 3)):
    """our module path in surrogates.StereoRcnnR101 -> (upstream conv prefix, upstream bn prefix)"""
    m = {"stem": ("RCNN_layer0.0", "RCNN_layer0.1")}                       # Sequential(conv1, bn1, relu, maxpool)
    for li, n in enumerate(blocks, start=1):
        for b in range(n):
            up = "RCNN_layer%d.%d" % (li, b)
            for k in (1, 2, 3):
                m["layer%d.%d.conv%d" % (li, b, k)] = (up + ".conv%d" % k, up + ".bn%d" % k)
            if b == 0:
                m["layer%d.%d.down" % (li, b)] = (up + ".downsample.0", up + ".downsample.1")
    m["top"] = ("RCNN_toplayer", None)
    for i in range(3):
        m["lat.%d" % i] = ("RCNN_latlayer%d" % (i + 1), None)
        m["smooth.%d" % i] = ("RCNN_smooth%d" % (i + 1), None)
    m["rpn_conv"] = ("RCNN_rpn.RPN_Conv", None)
    m["rpn_reg"] = ("RCNN_rpn.RPN_bbox_pred_left_right", None)
    m["top7"] = ("RCNN_top.0", None)                                       # [UPSTREAM-UNVERIFIED] Sequential(conv 7x7, relu, conv 1x1, relu)
    m["top1"] = ("RCNN_top.2", None)
    for i in range(6):
        m["kpts_convs.%d" % i] = ("RCNN_kpts.%d" % (2 * i), None)         # [UPSTREAM-UNVERIFIED] conv, relu pairs
    m["kpts_class"] = ("kpts_class", None)
    return m


STEREO_RCNN_KEYS = stereo_rcnn_keys()
_SRCNN_PLAIN = {"cls_score": "RCNN_cls_score", "bbox_pred": "RCNN_bbox_pred", "dim_orien_pred": "RCNN_dim_orien_pred", "kpts_up": "RCNN_kpts.12"}


def load_stereo_rcnn(model, checkpoint, keys=None, strict=True):
    """``checkpoint['model']`` (attack/Stereo-RCNN/pgd_attack.py:95-96) -> surrogates.StereoRcnnR101 ``model``.  Returns the report and, if
    the checkpoint has it, ``uncert`` (:97).  After a load the surrogate's random-weight crutches are off: the image is not rescaled
    (``input_scale = 1``) and the RPN's regression output is used as it is (``bounded_rpn_deltas = False``).  The RPN's class layer is
    upstream a 2-way softmax over 2 x A channels ([bg x A, fg x A], stereo_rpn.py:36,79 + ``reshape(x, 2)``); the surrogate's A-channel
    sigmoid layer receives fg - bg, the same probability."""
    state = checkpoint["model"] if "model" in checkpoint else checkpoint
    keys = keys if keys is not None else stereo_rcnn_keys(tuple(len(getattr(model, "layer%d" % i)) for i in (1, 2, 3, 4)))
    ld = _Loader(state)
    mods = dict(model.named_modules())
    # STAGED: every tensor is checked and collected first; the model is written to only after the (strict) report has passed, so that a
    # checkpoint that does not fit leaves the model exactly as it was - random weights WITH their crutches - instead of half loaded
    staged = []            # (module, weight, bias)
    with torch.no_grad():
        for ours, (cp, bp) in keys.items():
            m = mods.get(ours)
            if m is None:
                ld.problems.append("%s: no such layer in the model" % ours)
                continue
            got = ld.conv(ours, cp, bp, m.weight.shape, m.bias.shape)
            if got is not None:
                staged.append((m, got[0], got[1]))
        w, b = ld.take("RCNN_rpn.RPN_cls_score.weight"), ld.take("RCNN_rpn.RPN_cls_score.bias")
        a = model.rpn_cls.weight.shape[0]
        if w is None or tuple(w.shape) != (2 * a,) + tuple(model.rpn_cls.weight.shape[1:]):
            ld.problems.append("rpn_cls: RCNN_rpn.RPN_cls_score.weight missing or not [2A, ...]")
        else:
            staged.append((model.rpn_cls, w[a:] - w[:a], (b[a:] - b[:a]) if b is not None else torch.zeros(a)))
        for ours, up in _SRCNN_PLAIN.items():
            m = mods[ours]
            w, b = ld.take(up + ".weight"), ld.take(up + ".bias")
            if w is None or tuple(w.shape) != tuple(m.weight.shape):
                ld.problems.append("%s: %s.weight missing or %s instead of %s" % (ours, up, None if w is None else tuple(w.shape), tuple(m.weight.shape)))
                continue
            staged.append((m, w, b if b is not None else torch.zeros_like(m.bias)))
        rep = ld.report(len(staged), strict, "load_stereo_rcnn")          # raises (strict) before anything has been written
        for m, w, b in staged:
            m.weight.copy_(w)
            m.bias.copy_(b)
            if hasattr(m, "_prep"):
                m._prep = None
    if not rep["problems"]:                  # a partial (non-strict) load keeps the random-weight crutches: half-random RPN deltas still need their bound
        model.input_scale, model.bounded_rpn_deltas = 1.0, False
    rep["uncert"] = checkpoint.get("uncert") if isinstance(checkpoint, dict) else None
    return rep


# ------------------------------------------------------------------------------------------------ DSGN
def dsgn_keys(blocks):
    """our layer name in adapters.DsgnShapedAdapter -> (upstream conv prefix, upstream bn prefix); PSMNet / DSGN naming
    [UPSTREAM-UNVERIFIED]: ``convbn`` = Sequential(conv, bn), BasicBlock.conv1 = Sequential(convbn, ReLU), conv2 = convbn"""
    fe = "feature_extraction."
    m = {"f0a": (fe + "firstconv.0.0", fe + "firstconv.0.1"), "f0b": (fe + "firstconv.2.0", fe + "firstconv.2.1"), "f0c": (fe + "firstconv.4.0", fe + "firstconv.4.1")}
    for pre, proj, li in blocks:
        i = pre.split(".")[1]
        up = fe + "layer%d.%s" % (li, i)
        m[pre + ".a"] = (up + ".conv1.0.0", up + ".conv1.0.1")
        m[pre + ".b"] = (up + ".conv2.0", up + ".conv2.1")
        if proj:
            m[pre + ".p"] = (up + ".downsample.0", up + ".downsample.1")
    for j, k in enumerate((64, 32, 16, 8), start=1):
        m["spp%d" % k] = (fe + "branch%d.1.0" % j, fe + "branch%d.1.1" % j)
    m["last_a"] = (fe + "lastconv.0.0", fe + "lastconv.0.1")
    m["last_b"] = (fe + "lastconv.2", None)
    m3 = {"dres0a": ("dres0.0.0", "dres0.0.1"), "dres0b": ("dres0.2.0", "dres0.2.1"), "dres1a": ("dres1.0.0", "dres1.0.1"), "dres1b": ("dres1.2.0", "dres1.2.1"),
          "hg1": ("hg_cv.conv1.0.0", "hg_cv.conv1.0.1"), "hg2": ("hg_cv.conv2.0", "hg_cv.conv2.1"), "hg3": ("hg_cv.conv3.0.0", "hg_cv.conv3.0.1"),
          "hg4": ("hg_cv.conv4.0.0", "hg_cv.conv4.0.1"), "hg5": ("hg_cv.conv5.0", "hg_cv.conv5.1"), "hg6": ("hg_cv.conv6.0", "hg_cv.conv6.1"),
          "cls_a": ("classif1.0.0", "classif1.0.1"), "cls_b": ("classif1.2", None),
          "gv1": ("rpn3d_conv.0.0", "rpn3d_conv.0.1"), "gh1": ("hg_rpn3d.conv1.0.0", "hg_rpn3d.conv1.0.1"), "gh2": ("hg_rpn3d.conv2.0", "hg_rpn3d.conv2.1"),
          "gh3": ("hg_rpn3d.conv3.0.0", "hg_rpn3d.conv3.0.1"), "gh4": ("hg_rpn3d.conv4.0.0", "hg_rpn3d.conv4.0.1"), "gh5": ("hg_rpn3d.conv5.0", "hg_rpn3d.conv5.1"),
          "gh6": ("hg_rpn3d.conv6.0", "hg_rpn3d.conv6.1")}
    m2 = {"bev_a": ("rpn3d_conv2.0.0", "rpn3d_conv2.0.1"), "bh1": ("rpn3d_conv3.conv1.0.0", "rpn3d_conv3.conv1.0.1"), "bh2": ("rpn3d_conv3.conv2.0", "rpn3d_conv3.conv2.1"),
          "bh3": ("rpn3d_conv3.conv3.0.0", "rpn3d_conv3.conv3.0.1"), "bh4": ("rpn3d_conv3.conv4.0.0", "rpn3d_conv3.conv4.0.1"),
          "head_cls": ("bbox_cls", None), "head_reg": ("bbox_reg", None), "head_ctr": ("bbox_centerness", None)}
    for i in range(4):
        m2["ct%d" % i] = ("rpn3d_cls_convs.%d.0" % (2 * i), "rpn3d_cls_convs.%d.1" % (2 * i))
        m2["rt%d" % i] = ("rpn3d_bbox_convs.%d.0" % (2 * i), "rpn3d_bbox_convs.%d.1" % (2 * i))
    mt = {"bh5": ("rpn3d_conv3.conv5.0", "rpn3d_conv3.conv5.1"), "bh6": ("rpn3d_conv3.conv6.0", "rpn3d_conv3.conv6.1")}
    m.update(m2)
    return {"conv2d": m, "conv3d": m3, "convT2d": mt}


def load_dsgn(adapter, checkpoint, keys=None, strict=True):
    """``state_dict['state_dict']`` (attack/DSGN/pgd_attack.py:143-144) -> adapters.DsgnShapedAdapter: every layer's weight and folded bias
    replaced, the kernels' prepared layouts rebuilt (they are made once per weight tensor)."""
    state = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    keys = keys if keys is not None else dsgn_keys(adapter.blocks)
    ld = _Loader(state)
    ops = adapter.ops
    loaded = 0
    with torch.no_grad():
        for ours, (cp, bp) in keys["conv2d"].items():
            w, b, s, p, d = adapter.w2[ours]
            got = ld.conv(ours, cp, bp, w.shape, b.shape)
            if got is not None:
                adapter.w2[ours] = (got[0].to(w.device), got[1].to(w.device), s, p, d)
                loaded += 1
        for ours, (cp, bp) in keys["convT2d"].items():
            w, b = adapter.wt2[ours]
            got = ld.conv(ours, cp, bp, w.shape, b.shape, transposed=True)
            if got is not None:
                adapter.wt2[ours] = (got[0].to(w.device), got[1].to(w.device))
                loaded += 1
        for ours, (cp, bp) in keys["conv3d"].items():
            e = adapter.w3[ours]
            got = ld.conv(ours, cp, bp, e["w"].shape, e["b"].shape, transposed=(e["kind"] == "t2"))
            if got is None:
                continue
            e["w"], e["b"] = got[0].to(e["w"].device), got[1].to(e["b"].device)
            if adapter.mfma_conv:                                           # as adapters.DsgnShapedAdapter.add3 prepared the random draw
                if e["kind"] == "s1":
                    e["p"], e["pt"] = ops.conv3d_k3_prep(e["w"]), ops.conv3d_k3_prep(e["w"], transpose=True)
                    e["wino"] = ops.Conv3dWinoPrep(e["w"]) if adapter.wino3d and e["cout"] >= 4 else None
                elif e["kind"] == "s2":
                    e["p"], e["pt"] = ops.conv3d_k3_s2_prep(e["w"]), ops.conv_transpose3d_k3_s2_prep(e["w"])
                else:
                    e["p"], e["pt"] = ops.conv_transpose3d_k3_s2_prep(e["w"]), ops.conv3d_k3_s2_prep(e["w"])
            loaded += 1
    adapter._p2 = {}                                                        # prepared 2D layouts belong to the old weights
    adapter.cls_bias = 0.0                                                  # the focal-loss prior of the random draw: a trained bbox_cls carries its own bias
    return ld.report(loaded, strict, "load_dsgn")
