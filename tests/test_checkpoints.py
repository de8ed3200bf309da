"""Upstream checkpoint layouts onto the layer-list graphs (checkpoints.py): ``checkpoint['model']`` / ``['uncert']`` of
attack/Stereo-RCNN/pgd_attack.py:94-97 into surrogates.StereoRcnnR101 and ``state_dict['state_dict']`` of attack/DSGN/pgd_attack.py:142-145
into adapters.DsgnShapedAdapter.  The upstream-named state dicts are produced by torch modules written HERE with the module names the
reference's substitute files spell out (stereo_rcnn.py:69-85,157-171; stereo_rpn.py:32-40) - BatchNorms with non-trivial statistics -
and the loaded graph must compute what that module computes.  CPU."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from eval_driving_safety_amd import checkpoints, surrogates


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, 4 * planes, 1, bias=False), nn.BatchNorm2d(4 * planes)
        self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride=stride, bias=False), nn.BatchNorm2d(4 * planes)) \
            if (stride != 1 or cin != 4 * planes) else None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + (x if self.downsample is None else self.downsample(x)))


class _Rpn(nn.Module):
    def __init__(self):
        super().__init__()
        self.RPN_Conv = nn.Conv2d(256, 512, 3, 1, 1)
        self.RPN_cls_score = nn.Conv2d(1024, 6, 1)
        self.RPN_bbox_pred_left_right = nn.Conv2d(1024, 18, 1)


class UpstreamNamed(nn.Module):
    """the module tree of the upstream network as the reference's stereo_rcnn.py / stereo_rpn.py name it, one block per stage"""

    def __init__(self, blocks=(1, 1, 1, 1)):
        super().__init__()
        self.RCNN_layer0 = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        cin = 64
        for i, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), blocks, (1, 2, 2, 2)), start=1):
            mods = []
            for b in range(n):
                mods.append(_Bottleneck(cin, planes, stride if b == 0 else 1))
                cin = 4 * planes
            setattr(self, "RCNN_layer%d" % i, nn.Sequential(*mods))
        self.RCNN_toplayer = nn.Conv2d(2048, 256, 1)
        self.RCNN_smooth1, self.RCNN_smooth2, self.RCNN_smooth3 = (nn.Conv2d(256, 256, 3, padding=1) for _ in range(3))
        self.RCNN_latlayer1, self.RCNN_latlayer2, self.RCNN_latlayer3 = nn.Conv2d(1024, 256, 1), nn.Conv2d(512, 256, 1), nn.Conv2d(256, 256, 1)
        self.RCNN_rpn = _Rpn()
        self.RCNN_top = nn.Sequential(nn.Conv2d(512, 2048, 7), nn.ReLU(), nn.Conv2d(2048, 2048, 1), nn.ReLU())
        self.RCNN_cls_score, self.RCNN_bbox_pred, self.RCNN_dim_orien_pred = nn.Linear(2048, 2), nn.Linear(2048, 12), nn.Linear(2048, 10)
        kp = []
        for _ in range(6):
            kp += [nn.Conv2d(256, 256, 3, padding=1), nn.ReLU()]
        self.RCNN_kpts = nn.Sequential(*kp, nn.ConvTranspose2d(256, 256, 2, 2), nn.ReLU())
        self.kpts_class = nn.Conv2d(256, 6, 1)
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear, nn.ConvTranspose2d)):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.5 / m.weight[0].numel()) ** 0.5)
                    if m.bias is not None:
                        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))

    def pyramid(self, im):                                               # stereo_rcnn.py:157-171
        c2 = self.RCNN_layer1(self.RCNN_layer0(im))
        c3 = self.RCNN_layer2(c2)
        c4 = self.RCNN_layer3(c3)
        c5 = self.RCNN_layer4(c4)
        up = lambda x, y: F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=False) + y      # noqa: E731
        p5 = self.RCNN_toplayer(c5)
        p4 = self.RCNN_smooth1(up(p5, self.RCNN_latlayer1(c4)))
        p3 = self.RCNN_smooth2(up(p4, self.RCNN_latlayer2(c3)))
        p2 = self.RCNN_smooth3(up(p3, self.RCNN_latlayer3(c2)))
        return [p2, p3, p4, p5, p5[:, :, ::2, ::2]]


def test_stereo_rcnn_checkpoint_loads_into_the_layer_list_graph():
    up = UpstreamNamed().eval()
    ckpt = {"model": up.state_dict(), "uncert": torch.tensor([0.1, -0.2, 0.3, 0.0, 0.5, -0.4])}
    net = surrogates.StereoRcnnR101(seed=3, blocks=(1, 1, 1, 1)).eval()
    assert net.input_scale == 1.0 / 64.0 and net.bounded_rpn_deltas
    rep = checkpoints.load_stereo_rcnn(net, ckpt)
    assert not rep["problems"] and not rep["unused_upstream_keys"] and torch.equal(rep["uncert"], ckpt["uncert"])
    assert rep["loaded_layers"] == 1 + 4 * 4 + 1 + 3 + 3 + 3 + 2 + 6 + 1 + 4
    assert net.input_scale == 1.0 and not net.bounded_rpn_deltas
    im = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(1)) * 50
    with torch.no_grad():
        want = up.pyramid(im)
        got = net.pyramid(im * net.input_scale)
        for a, b in zip(want, got):
            assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
        # stereo RPN: shared 3x3 on both eyes, concatenated; objectness fg - bg of the 2-way softmax; the 18 regression channels as they are
        both_up = torch.cat([F.relu(up.RCNN_rpn.RPN_Conv(want[2])), F.relu(up.RCNN_rpn.RPN_Conv(want[2] * 0.5))], 1)
        both = net.rpn_features(got[2], got[2] * 0.5)
        assert float((both - both_up).abs().max()) <= 2e-5 * float(both_up.abs().max())
        sc = up.RCNN_rpn.RPN_cls_score(both_up)
        prob_fg = torch.softmax(torch.stack([sc[:, :3], sc[:, 3:]], 0), 0)[1]
        assert float((torch.sigmoid(net.rpn_cls(both)) - prob_fg).abs().max()) <= 1e-5
        reg_up = up.RCNN_rpn.RPN_bbox_pred_left_right(both_up)
        assert float((net.rpn_deltas(both) - reg_up).abs().max()) <= 2e-5 * float(reg_up.abs().max())
        pooled = torch.randn(3, 512, 7, 7, generator=torch.Generator().manual_seed(2))
        tail_up = up.RCNN_top(pooled).flatten(1)
        assert float((net.head_to_tail(pooled) - tail_up).abs().max()) <= 2e-5 * float(tail_up.abs().max())
        assert float((net.cls_score(tail_up) - up.RCNN_cls_score(tail_up)).abs().max()) == 0
        f14 = torch.randn(2, 256, 14, 14, generator=torch.Generator().manual_seed(4))
        k_up = up.kpts_class(up.RCNN_kpts(f14)).sum(2)
        assert float((net.kpts_logits(f14) - k_up).abs().max()) <= 2e-5 * float(k_up.abs().max())


def test_stereo_rcnn_loader_refuses_a_partial_or_misshapen_checkpoint():
    up = UpstreamNamed().eval()
    sd = dict(up.state_dict())
    sd["RCNN_extra.weight"] = torch.zeros(1)
    del sd["RCNN_smooth2.weight"]
    sd["RCNN_layer2.0.conv2.weight"] = torch.zeros(128, 128, 1, 1)
    net = surrogates.StereoRcnnR101(seed=3, blocks=(1, 1, 1, 1)).eval()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    with pytest.raises(ValueError) as e:
        checkpoints.load_stereo_rcnn(net, {"model": sd})
    msg = str(e.value)
    assert "smooth.1: no RCNN_smooth2.weight" in msg and "layer2.0.conv2" in msg and "RCNN_extra.weight" in msg
    # refused BEFORE anything was written: the model is exactly what it was, random weights with their crutches on
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items()) and net.input_scale == 1.0 / 64.0 and net.bounded_rpn_deltas
    partial = surrogates.StereoRcnnR101(seed=3, blocks=(1, 1, 1, 1)).eval()
    rep = checkpoints.load_stereo_rcnn(partial, {"model": sd}, strict=False)
    assert len(rep["problems"]) == 2 and "RCNN_extra.weight" in rep["unused_upstream_keys"]
    assert partial.input_scale == 1.0 / 64.0 and partial.bounded_rpn_deltas          # a partial load keeps the crutches of the layers still random
    # the full-depth key map names every upstream tensor of a [3, 4, 23, 3] ResNet-101
    assert len(checkpoints.STEREO_RCNN_KEYS) == 1 + 3 * 33 + 4 + 1 + 6 + 2 + 2 + 6 + 1


def _dsgn_state(adapter, gen):
    """an upstream-named DSGN state dict for the adapter's layer list: conv weights + BatchNorm tensors under the PSMNet / DSGN names,
    wrapped in nn.DataParallel's ``module.`` prefix as finetune_53.tar is"""
    keys = checkpoints.dsgn_keys(adapter.blocks)
    sd, want = {}, {}

    def put(ours, cp, bp, wshape, nout, transposed=False):
        w = torch.randn(wshape, generator=gen) * 0.1
        sd["module." + cp + ".weight"] = w
        b = None
        if bp is None:
            b = torch.randn(nout, generator=gen) * 0.1
            sd["module." + cp + ".bias"] = b
            want[ours] = (w, b)
            return
        g, be = 1 + 0.2 * torch.randn(nout, generator=gen), 0.1 * torch.randn(nout, generator=gen)
        mu, var = 0.1 * torch.randn(nout, generator=gen), 0.5 + torch.rand(nout, generator=gen)
        for s, v in ((".weight", g), (".bias", be), (".running_mean", mu), (".running_var", var)):
            sd["module." + bp + s] = v
        sd["module." + bp + ".num_batches_tracked"] = torch.tensor(7)
        want[ours] = checkpoints.fold_bn_tensors(w, None, g, be, mu, var, 1e-5, transposed)

    for ours, (cp, bp) in keys["conv2d"].items():
        w, b = adapter.w2[ours][:2]
        put(ours, cp, bp, tuple(w.shape), b.shape[0])
    for ours, (cp, bp) in keys["convT2d"].items():
        w, b = adapter.wt2[ours]
        put(ours, cp, bp, tuple(w.shape), b.shape[0], transposed=True)
    for ours, (cp, bp) in keys["conv3d"].items():
        e = adapter.w3[ours]
        put(ours, cp, bp, tuple(e["w"].shape), e["b"].shape[0], transposed=(e["kind"] == "t2"))
    return sd, want


def test_dsgn_checkpoint_loads_into_the_layer_list_graph():
    from eval_driving_safety_amd import adapters
    ad = adapters.DsgnShapedAdapter(torch.device("cpu"), seed=0, torch_ops=True, image_hw=(64, 96))
    gen = torch.Generator().manual_seed(5)
    sd, want = _dsgn_state(ad, gen)
    rep = checkpoints.load_dsgn(ad, {"state_dict": sd})
    assert not rep["problems"] and not rep["unused_upstream_keys"] and rep["loaded_layers"] == len(want) == len(ad.w2) + len(ad.wt2) + len(ad.w3)
    for name, (w, b) in want.items():
        have = ad.w2[name][:2] if name in ad.w2 else (ad.wt2[name] if name in ad.wt2 else (ad.w3[name]["w"], ad.w3[name]["b"]))
        assert torch.equal(have[0], w) and torch.equal(have[1], b), name
    assert ad.cls_bias == 0.0
    # the loaded extractor computes bn(conv(x)) of the upstream tensors: first layer, by hand
    x = torch.randn(1, 3, 64, 96, generator=gen)
    k = "module.feature_extraction.firstconv.0."
    y = F.conv2d(x, sd[k + "0.weight"], None, 2, 1)
    y = F.relu(F.batch_norm(y, sd[k + "1.running_mean"], sd[k + "1.running_var"], sd[k + "1.weight"], sd[k + "1.bias"], False, 0.0, 1e-5))
    got = ad._c2(x, "f0a", True)
    assert float((got - y).abs().max()) <= 2e-6 * float(y.abs().max())
    with torch.no_grad():
        assert bool(torch.isfinite(ad.features(x)).all())
    sd.pop("module.dres1.2.1.running_var")
    with pytest.raises(ValueError, match="dres1b: incomplete BatchNorm dres1.2.1"):
        checkpoints.load_dsgn(adapters.DsgnShapedAdapter(torch.device("cpu"), seed=0, torch_ops=True, image_hw=(64, 96)), {"state_dict": sd})
