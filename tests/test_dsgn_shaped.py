"""adapters.PsvStereoAdapter(dsgn_head=True): the DSGN-shaped graph on libadvengine's kernels - fused depth
regression, plane-sweep -> 3D geometric volume resampling, 3D convolution, bird's-eye-view heads, focal loss - against the same
graph written with torch's own operators (F.interpolate + softmax, F.grid_sample, F.conv3d, an autograd focal loss)."""
import pytest
import torch
import torch.nn.functional as F


def _torch_reference_loss(net, x, extra, targets):
    b = x.shape[0] // 2
    imgL, imgR = x[:b], x[b:]
    fl, fr = net.features(imgL), net.features(imgR)
    cost = net.ops.PsvBuildLerp.apply(fl.contiguous(), fr.contiguous(), net.shifts(b))
    mfma, net.mfma_conv = net.mfma_conv, False
    try:
        score, feat = net._volume_net(cost, with_features=True)
        up = F.interpolate(score[:, None], size=net.up_size, mode="trilinear", align_corners=False)[:, 0]
        depth = (torch.softmax(up, 1) * net.depth_up.view(1, -1, 1, 1)).sum(1)
        prob = torch.softmax(score, dim=1)
        grid = net.gv_grid.expand(b, -1, -1, -1, -1)
        gv = F.grid_sample(feat * prob[:, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        gv = F.relu(F.conv3d(gv, net.g3, net.gb3, padding=1))
    finally:
        net.mfma_conv = mfma
    bb, c, zg, yg, xg = gv.shape
    bev = F.avg_pool3d(gv, (1, net.ypool, 1)).permute(0, 1, 3, 2, 4).reshape(bb, c * (yg // net.ypool), zg, xg)
    bev = F.relu(F.conv2d(F.relu(F.conv2d(bev, net.bev1, padding=1)), net.bev2, padding=1))
    cls = F.conv2d(bev, net.head_cls, padding=1) + net.cls_bias
    reg, ctr = F.conv2d(bev, net.head_reg, padding=1), F.conv2d(bev, net.head_ctr, padding=1)
    tcls = targets["cls"]
    logits = cls.permute(0, 2, 3, 1).reshape(-1)
    t = tcls.float()
    p = torch.sigmoid(logits)
    focal = -(t * 0.25 * (1 - p) ** 2 * F.logsigmoid(logits) + (1 - t) * 0.75 * p ** 2 * F.logsigmoid(-logits)).sum()
    npos = targets["npos"]
    det = focal / npos + F.smooth_l1_loss(reg.reshape(-1)[targets["posr_idx"]], targets["reg_pos"], reduction="sum") / npos + \
        F.binary_cross_entropy_with_logits(ctr.reshape(-1)[targets["pos_idx"]], targets["ctr_pos"], reduction="sum") / npos
    gt = extra.disp_true
    mask = (gt > float(net.depth[0])) & (gt <= float(net.depth[-1]) + 0.8)
    return F.smooth_l1_loss(depth[mask], gt[mask], reduction="mean") + det


@pytest.mark.gpu
@pytest.mark.parametrize("hourglass", [False, True])
def test_dsgn_shaped_graph_matches_torch_operators(hourglass):
    from eval_driving_safety_amd import adapters, data
    dev = torch.device("cuda", 0)
    hw = (96, 160)
    net = adapters.PsvStereoAdapter(dev, seed=3, hourglass=hourglass, dsgn_head=True, image_hw=hw, cu=80.0, cv=44.0, fu=180.0)
    gen = torch.Generator().manual_seed(11)
    left = torch.randn((2, 3) + hw, generator=gen)
    batch = data.StereoBatch(left, torch.roll(left, shifts=-6, dims=3) + 0.05 * torch.randn((2, 3) + hw, generator=gen), ["000000", "000001"], None)
    extra = net.synthetic_extra(batch, seed=2)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    loss, grad = net.loss_and_grad(x, extra)
    assert torch.isfinite(loss) and torch.isfinite(grad).all() and float(grad.abs().max()) > 0
    loss2, grad2 = net.loss_and_grad(x, extra)
    assert float(loss2) == float(loss) and torch.equal(grad, grad2), "the whole graph is reproducible bit for bit"
    xr = x.clone().requires_grad_(True)
    ref = _torch_reference_loss(net, xr, extra, net._tgt)
    ref.backward()
    assert abs(float(ref.detach()) - float(loss)) <= 2e-4 * max(1.0, abs(float(ref.detach())))
    scale = float(xr.grad.abs().max())
    assert float((xr.grad - grad).abs().max()) <= 5e-3 * scale        # float32 through ~15 layers, two summation orders
    # the attack direction is what the PGD step consumes: the signs agree wherever the gradient is not at rounding level
    big = xr.grad.abs() > 1e-2 * scale
    assert float((torch.sign(xr.grad[big]) == torch.sign(grad[big])).float().mean()) > 0.999


@pytest.mark.gpu
def test_dsgn_shaped_detections_are_deterministic_and_well_formed():
    from eval_driving_safety_amd import adapters
    dev = torch.device("cuda", 0)
    hw = (96, 160)
    net = adapters.PsvStereoAdapter(dev, seed=5, hourglass=False, dsgn_head=True, image_hw=hw, cu=80.0, cv=44.0, fu=180.0)
    gen = torch.Generator().manual_seed(2)
    left = torch.randn((2, 3) + hw, generator=gen)
    x = torch.cat([left, torch.roll(left, shifts=-5, dims=3)]).to(dev)
    a = net.detect(x, topk=16, nms_thresh=0.3, cu=80.0, cv=44.0)
    b = net.detect(x, topk=16, nms_thresh=0.3, cu=80.0, cv=44.0)
    assert a == b and len(a) == 2
    for dets in a:
        assert 1 <= len(dets) <= 16
        scores = [d[2] for d in dets]
        assert scores == sorted(scores, reverse=True)                      # NMS keeps score order
        for cls_id, bbox, score, centre, (h, w, l, ry) in dets:
            assert cls_id == 2 and 0 < score < 1 and bbox[0] < bbox[2] and bbox[1] < bbox[3] and centre[2] >= 1.0 and h > 0 and w > 0 and l > 0


@pytest.mark.gpu
def test_dsgn_layer_list_graph_matches_torch_operators():
    """adapters.DsgnShapedAdapter (SURVEY App. B's layer list: PSMNet-style extractor, dres0/dres1 + 3D hourglass, 64-channel 3DGV stack +
    3D hourglass, bird's-eye-view 2D hourglass, head towers) on libadvengine's kernels against the same graph, same weights, computed
    with torch's own operators; the FLOP count of a step is a property of the layer list, not of who computes it"""
    from eval_driving_safety_amd import adapters, data
    dev = torch.device("cuda", 0)
    hw = (96, 160)
    kw = dict(seed=3, image_hw=hw, cu=80.0, cv=44.0, fu=180.0)
    net, ref = adapters.DsgnShapedAdapter(dev, **kw), adapters.DsgnShapedAdapter(dev, torch_ops=True, **kw)
    gen = torch.Generator().manual_seed(11)
    left = torch.randn((1, 3) + hw, generator=gen)
    batch = data.StereoBatch(left, torch.roll(left, shifts=-6, dims=3) + 0.05 * torch.randn((1, 3) + hw, generator=gen), ["000000"], None)
    extra = net.synthetic_extra(batch, seed=2)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    # Everything 3D is this package's and deterministic; the 2D layers are MIOpen's, which answers a shape's FIRST call with a fallback
    # solver and may pick split-K solvers that accumulate with atomics (igemm ..._gkgs): ask it for deterministic ones, warm up, compare
    net.loss_and_grad(x, extra)
    loss, grad = net.loss_and_grad(x, extra)
    assert torch.isfinite(loss) and torch.isfinite(grad).all() and float(grad.abs().max()) > 0
    loss2, grad2 = net.loss_and_grad(x, extra)
    assert float(loss2) == float(loss)
    same = float((grad == grad2).float().mean())
    assert same == 1.0 or float((grad - grad2).abs().max()) <= 1e-6 * float(grad.abs().max()), \
        "the graph's gradient is reproducible (bit for bit where MIOpen's solvers are deterministic: %.6f of the elements equal)" % same
    ref.loss_and_grad(x.clone(), extra)
    loss_r, grad_r = ref.loss_and_grad(x.clone(), extra)
    assert abs(float(loss_r) - float(loss)) <= 5e-4 * max(1.0, abs(float(loss_r)))
    scale = float(grad_r.abs().max())
    assert float((grad_r - grad).abs().max()) <= 2e-2 * scale          # float32 through ~90 layers, two summation orders
    big = grad_r.abs() > 2e-2 * scale
    assert float((torch.sign(grad_r[big]) == torch.sign(grad[big])).float().mean()) > 0.995
    # ... and over ALL elements, which is what a PGD step consumes (every flipped sign moves that pixel by 2 alpha): reported, and bounded
    # loosely - elements whose gradient is at rounding level do flip between two float32 summation orders (tools/pipeline_agreement.py
    # measures the full-size 20-step figure; profiles/r04_pipeline_agreement.json)
    every = float((torch.sign(grad_r) == torch.sign(grad)).float().mean())
    print("sign agreement over all %d elements: %.6f" % (grad.numel(), every))
    assert every > 0.97
    f = net.flops_per_step(x, extra)
    assert f == ref.flops_per_step(x.clone(), extra) and f > 5e11      # the 3DGV stack alone is ~0.9 TFLOP per step at any image size


_TWO_PROCESS = """
import hashlib, sys, torch
from eval_driving_safety_amd import adapters, data, routes
dev = torch.device("cuda", 0)
net = adapters.DsgnShapedAdapter(dev, seed=0)
batch = next(iter(data.SyntheticStereo(1, "dsgn", batch=1, seed=0)))
extra = net.synthetic_extra(batch, seed=1)
x = torch.cat([batch.imgL, batch.imgR]).to(dev)
net.loss_and_grad(x, extra)
loss, grad = net.loss_and_grad(x, extra)
torch.cuda.synchronize()
print("DIGEST", hashlib.sha256(grad.cpu().numpy().tobytes()).hexdigest(), float(loss), routes.table_hash(), len(routes.misses()), routes.mode())
"""


@pytest.mark.gpu
def test_layer_list_gradient_is_identical_in_two_fresh_processes():
    """the full-size DSGN layer-list step in two separate processes: the same bytes.  Round 3 chose each layer's kernel by a stopwatch at
    first use, so two runs (or two ranks) could take different float summation orders; the routes now come from the committed table
    (routes_gfx950.json), which holds every layer shape of this graph - no lookup falls through to the fixed rule, nothing is timed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("ADV_ROUTES", None)
    outs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", _TWO_PROCESS], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:]
        outs.append([l for l in p.stdout.splitlines() if l.startswith("DIGEST")][0].split())
    assert outs[0] == outs[1], outs
    assert outs[0][5] == "table" and outs[0][4] == "0", "every layer shape of the BASELINE graph is in the committed table: %s" % outs[0]


def test_route_table_is_committed_and_well_formed():
    import json
    import os
    from eval_driving_safety_amd import routes
    path = os.path.join(os.path.dirname(routes.__file__), "routes_gfx950.json")
    doc = json.load(open(path))
    assert doc["arch"] == "gfx950" and len(doc["routes"]) >= 200 and set(doc["routes"].values()) <= set(routes.ROUTES)
    routes.configure("table")
    assert routes.mode() == "table" and len(routes.table_hash()) == 12 and routes.summary()["table_entries"] == len(doc["routes"])
    # a shape the table holds -> its entry; a shape it does not hold -> the fixed rule (Winograd where the layer has it, else the direct
    # kernel; torch only by table), never a timer - the candidates are not even called
    boom = lambda: (_ for _ in ()).throw(AssertionError("a route lookup must not run anything"))      # noqa: E731
    # (a 2D layer the table gives to torch - the one route the fixed rule never picks; the table holds only a handful of them)
    key = next(k for k, v in doc["routes"].items() if k[:2] in ("f|", "b|") and v == "")
    parts = key.split("|")
    tup = (parts[0], int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4]), tuple(int(v) for v in parts[5].split("x")), parts[6] == "1", parts[7] == "1")
    assert routes.key_str(tup) == key and routes.choose(tup, {"hip": boom, "": boom, "wino": boom}) == ""
    assert routes.choose(("f", 3, 8, 8, 1, (1, 8, 9, 9), False, True), {"hip": boom, "": boom, "wino": boom}) == "wino"
    assert routes.choose(("f", 1, 8, 8, 1, (1, 8, 9, 9), False, True), {"hip": boom, "": boom}) == "hip"
    assert routes.choose(("f3", 8, 8, (1, 8, 3, 9, 9), False, True), {"direct": boom, "wino": boom}) == "wino"
    assert len(routes.misses()) == 3
    # lookups are memoised per (key, candidates) - the answer, the record of decisions (used()) and the candidate set all still count
    routes._state["used"].clear()
    assert routes.choose(tup, {"hip": boom, "": boom, "wino": boom}) == "" and routes.used() == {key: ""}
    assert routes.choose(tup, {"hip": boom, "wino": boom}) == "wino"               # torch not on offer for this call: the fixed rule among the rest
    assert routes.choose(("f", 3, 8, 8, 1, [1, 8, 9, 9], False, True), {"hip": boom, "": boom, "wino": boom}) == "wino"      # unhashable key: not memoised, still answered
    routes.configure("fixed")
    assert routes.choose(tup, {"hip": boom, "": boom, "wino": boom}) == "wino" and routes.table_hash() == "fixed"      # (a new mode forgets the memo)
    # a 3x3 layer the table does not know, F(4x4,3x3) on offer: it is taken where its launch fills the chip with mostly real outputs - a
    # function of the key (no clock), fitted to the measured table (routes._wino4_pays)
    four = {"hip": boom, "": boom, "wino": boom, "wino4": boom}
    assert routes.choose(("f", 3, 128, 128, 1, (2, 128, 190, 300), False, True), four) == "wino4"          # 140 tiles x 2 images x 2 channel blocks
    assert routes.choose(("b", 3, 128, 128, 1, (2, 128, 190, 300), False, False), four) == "wino4"
    assert routes.choose(("f", 3, 256, 256, 1, (2, 256, 37, 120), False, True), four) == "wino4"           # 96 workgroups, 256 input channels: the K-split launch
    assert routes.choose(("f", 3, 64, 64, 1, (2, 64, 37, 120), False, True), four) == "wino"               # 24 workgroups and a short contraction: nothing to deal out
    assert routes.choose(("f", 3, 128, 8, 1, (2, 128, 190, 300), False, True), four) == "wino"             # 8 output channels: a 64-channel tile is padding
    assert routes.choose(("f3", 128, 128, (1, 128, 47, 5, 76), False, True), {"direct": boom, "wino": boom, "wino4": boom}) == "wino"     # 5-row planes: padding
    assert routes.choose(("f3", 32, 32, (1, 32, 47, 96, 312), False, True), {"direct": boom, "wino": boom, "wino4": boom}) == "wino4"
    assert routes.choose(("f", 1, 128, 128, 1, (2, 128, 190, 300), False, True), {"hip": boom, "": boom}) == "hip"
    routes.configure("table")
    assert routes.choose(tup, {"hip": boom, "": boom, "wino": boom}) == ""


def test_dilated_convolution_equals_plain_convolution_on_parity_sub_images():
    """DsgnShapedAdapter runs its dilation-2 blocks as dilation-1 blocks on the four (row, column) parity sub-images: the same values,
    exactly (the same products summed in the same order), and split / merge are inverse permutations"""
    import torch.nn.functional as F
    from eval_driving_safety_amd.adapters import DsgnShapedAdapter as A
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((2, 5, 8, 12), generator=gen)
    assert torch.equal(A._parity_merge(A._parity_split(x), 2), x)
    w = torch.randn((7, 5, 3, 3), generator=gen)
    ref = F.conv2d(x, w, None, 1, 2, 2)
    got = A._parity_merge(F.conv2d(A._parity_split(x), w, None, 1, 1, 1), 2)
    assert torch.equal(got, ref)
