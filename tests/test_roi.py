"""Stereo R-CNN RoI path natives: RoIAlign forward (bit-exact vs the oracle), backward
(deterministic gather: bit-exact vs the oracle's ordered float32 sum, and within rounding of a float64 sum), NMS (exact indices).
CPU part: the oracle's RoIAlign against torch autograd of an equivalent dense formulation."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O


def _rois(rs, n, b, img_h, img_w):
    x1 = rs.rand(n) * img_w * 0.8
    y1 = rs.rand(n) * img_h * 0.8
    bw = rs.rand(n) * img_w * 0.5 + 2
    bh = rs.rand(n) * img_h * 0.5 + 2
    rois = np.stack([rs.randint(0, b, n), x1, y1, np.minimum(x1 + bw, img_w + 10), np.minimum(y1 + bh, img_h + 10)], 1).astype(np.float32)
    rois[0, 1:] = [0, 0, 1, 1]                         # tiny roi: width/height clamp to 1
    if n > 1:
        rois[1, 1:] = [-20, -20, img_w + 40, img_h + 40]   # leaves the map on every side
    return rois


def test_oracle_roi_align_adjoint_and_constant():
    rs = np.random.RandomState(0)
    feat = rs.randn(2, 3, 12, 16).astype(np.float32)
    rois = _rois(rs, 5, 2, 12 * 16, 16 * 16)
    out = O.roi_align(feat, rois, (3, 4), 1 / 16.0)
    g = rs.randn(*out.shape).astype(np.float32)
    gf = O.roi_align_bwd(g, rois, feat.shape, 1 / 16.0)
    lhs = float((out.astype(np.float64) * g).sum())
    rhs = float((feat.astype(np.float64) * gf).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # the float32 sum in the gather kernel's order is the same gradient up to float32 rounding
    np.testing.assert_allclose(O.roi_align_bwd_ordered(g, rois, feat.shape, 1 / 16.0), gf, rtol=2e-5, atol=2e-5)
    ones = np.ones((1, 2, 10, 10), np.float32)
    inside = np.array([[0, 16, 16, 100, 120]], np.float32)       # fully inside -> every bin averages to 1
    assert np.allclose(O.roi_align(ones, inside, 7, 1 / 16.0), 1.0, atol=1e-6)


def test_oracle_skips_rois_with_a_negative_batch_index():
    rs = np.random.RandomState(4)
    feat = rs.randn(2, 3, 12, 16).astype(np.float32)
    rois = _rois(rs, 6, 2, 12 * 16, 16 * 16)
    skip = rois.copy()
    skip[[1, 4], 0] = -1
    out, ref = O.roi_align(feat, skip, 3, 1 / 16.0), O.roi_align(feat, rois, 3, 1 / 16.0)
    assert not out[[1, 4]].any() and out[[0, 2, 3, 5]].tobytes() == ref[[0, 2, 3, 5]].tobytes()
    g = rs.randn(*out.shape).astype(np.float32)
    keep = [0, 2, 3, 5]
    assert O.roi_align_bwd_ordered(g, skip, feat.shape, 1 / 16.0).tobytes() == O.roi_align_bwd_ordered(g[keep], rois[keep], feat.shape, 1 / 16.0).tobytes()


@pytest.mark.gpu
def test_hip_roi_align_skips_negative_batch_indices_and_pools_a_pyramid():
    """a roi with a negative batch index: the forward leaves its rows untouched, the backward adds nothing (bit-exact vs the oracle);
    ops.PyramidRoIAlign (every level handed the whole list, non-owned rois skipped, one output) == pooling each level's own rois apart"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(8)
    feat = rs.randn(2, 5, 20, 28).astype(np.float32)
    rois = _rois(rs, 9, 2, 20 * 8, 28 * 8)
    skip = rois.copy()
    skip[[2, 5, 6], 0] = -1
    tf, ts = torch.tensor(feat, device=dev), torch.tensor(skip, device=dev)
    out = torch.full((9, 5, 4, 4), 7.0, device=dev)
    ops.roi_align(tf, ts, 4, 1 / 8.0, out=out)
    got = out.cpu().numpy()
    want = O.roi_align(feat, skip, 4, 1 / 8.0)
    owned = [0, 1, 3, 4, 7, 8]
    assert (got[[2, 5, 6]] == 7.0).all() and got[owned].tobytes() == want[owned].tobytes()
    g = rs.randn(9, 5, 4, 4).astype(np.float32)
    gb = ops.roi_align_bwd(torch.tensor(g, device=dev), ts, feat.shape, 1 / 8.0).cpu().numpy()
    assert gb.tobytes() == O.roi_align_bwd_ordered(g, skip, feat.shape, 1 / 8.0).tobytes()
    # the pyramid: three levels of different size, owners by index
    feats = [rs.randn(1, 4, 24 >> l, 40 >> l).astype(np.float32) for l in range(3)]
    prois = _rois(rs, 11, 1, 24 * 4, 40 * 4)
    owner = rs.randint(0, 3, 11)
    scales = tuple(0.25 / (1 << l) for l in range(3))
    tfs = [torch.tensor(f, device=dev, requires_grad=True) for f in feats]
    pooled = ops.PyramidRoIAlign.apply(torch.tensor(prois, device=dev), torch.tensor(owner, device=dev), 3, scales, 0, *tfs)
    gp = rs.randn(11, 4, 3, 3).astype(np.float32)
    pooled.backward(torch.tensor(gp, device=dev))
    for l in range(3):
        idx = np.nonzero(owner == l)[0]
        if len(idx):
            assert pooled.detach().cpu().numpy()[idx].tobytes() == O.roi_align(feats[l], prois[idx], 3, scales[l]).tobytes(), l
        masked = prois.copy()
        masked[owner != l, 0] = -1
        assert tfs[l].grad.cpu().numpy().tobytes() == O.roi_align_bwd_ordered(gp, masked, feats[l].shape, scales[l]).tobytes(), l
    # a roi that NO level owns (an owner out of range: what .long() makes of a NaN level under a diverged RPN) and a NaN box: their rows are
    # zeros, on every run - not whatever the allocator handed out
    bad_owner = owner.copy()
    bad_owner[[1, 6]] = (7, -3)
    bad_rois = prois.copy()
    bad_rois[6, 1:] = np.nan
    for _ in range(2):
        junk = torch.full((64, 4, 3, 3), 123.0, device=dev)         # dirty the allocator's free list
        del junk
        again = ops.PyramidRoIAlign.apply(torch.tensor(bad_rois, device=dev), torch.tensor(bad_owner, device=dev), 3, scales, 0, *[t.detach() for t in tfs])
        a = again.cpu().numpy()
        assert not a[[1, 6]].any() and np.isfinite(a).all()
        keep = [i for i in range(11) if i not in (1, 6)]
        assert a[keep].tobytes() == pooled.detach().cpu().numpy()[keep].tobytes()


def _pyramid_fixture(golden):
    z = golden("pyramid_roi")
    return z, [z["feat%d" % l] for l in range(4)]


def test_pyramid_roi_pooling_equals_the_references_executed_method(golden, golden_index):
    """tests/golden/pyramid_roi.npz: the reference's ``PyramidRoI_Feat`` (attack/Stereo-RCNN/stereo_rcnn.py:110-141) executed with the
    oracle's RoIAlign as the per-level operator - surrogates.pyramid_roi_feat (level per roi, scale per level, rows back in roi order)
    around the same operator gives the same bytes, for the 7x7 and the 14x14 pooling"""
    from eval_driving_safety_amd import surrogates
    assert golden_index["pyramid_roi"]["levels_used"] == [2, 3, 4, 5]
    z, feats = _pyramid_fixture(golden)
    model = surrogates.StereoRcnnShaped.__new__(surrogates.StereoRcnnShaped)
    torch.nn.Module.__init__(model)
    model._roi_align = lambda f, r, p, s: torch.from_numpy(O.roi_align(f.numpy(), r.numpy(), p, float(s)))
    tf = [torch.tensor(f) for f in feats]
    for pooled, key in ((7, "pooled7"), (14, "pooled14")):
        got = model.pyramid_roi_feat(tf, torch.tensor(z["rois"]), torch.tensor(z["im_info"]), pooled)
        assert got.numpy().tobytes() == z[key].tobytes(), key


@pytest.mark.gpu
def test_hip_pyramid_roi_align_equals_the_references_executed_method(golden):
    """the same fixture through ops.PyramidRoIAlign (surrogates._pyramid_roi_feat_static: no compaction, skipped rois, one output) and through
    the compacting path on ops.RoIAlign: the reference's bytes"""
    from eval_driving_safety_amd import surrogates
    dev = torch.device("cuda", 0)
    z, feats = _pyramid_fixture(golden)
    model = surrogates.StereoRcnnShaped.__new__(surrogates.StereoRcnnShaped)
    torch.nn.Module.__init__(model)
    model._roi_align = None
    tf = [torch.tensor(f, device=dev) for f in feats]
    rois, info = torch.tensor(z["rois"], device=dev), torch.tensor(z["im_info"], device=dev)
    for pooled, key in ((7, "pooled7"), (14, "pooled14")):
        assert model._pyramid_roi_feat_static(tf, rois, float(z["im_info"][0, 0]), pooled).cpu().numpy().tobytes() == z[key].tobytes(), key
        assert model.pyramid_roi_feat(tf, rois, info, pooled).cpu().numpy().tobytes() == z[key].tobytes(), key


def test_oracle_nms_small_cases():
    boxes = np.float32([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10], [21, 21, 29, 29]])
    assert O.nms(boxes, 0.5).tolist() == [0, 2]
    assert O.nms(boxes, 0.95).tolist() == [0, 1, 2, 4]
    assert O.nms(boxes[:0], 0.5).tolist() == []


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(b=2, c=5, h=12, w=16, n=9, pooled=(3, 4), scale=1 / 16.0, sr=0),
    dict(b=1, c=16, h=38, w=125, n=24, pooled=7, scale=1 / 16.0, sr=0),      # FPN P4 of a 600x1987 image
    dict(b=1, c=8, h=150, w=497, n=12, pooled=14, scale=1 / 4.0, sr=0),       # FPN P2, keypoint head 14x14
    dict(b=2, c=4, h=20, w=20, n=7, pooled=2, scale=0.25, sr=2),
    dict(b=1, c=40, h=75, w=249, n=40, pooled=7, scale=1 / 8.0, sr=2),        # FPN P3, more channels than one workgroup's block
    dict(b=2, c=33, h=19, w=63, n=30, pooled=14, scale=1 / 32.0, sr=2),       # FPN P5: rois as large as the map
    dict(b=1, c=7, h=30, w=41, n=11, pooled=(5, 3), scale=1 / 16.0, sr=1),
    dict(b=1, c=6, h=30, w=41, n=11, pooled=4, scale=1 / 16.0, sr=3),         # a grid the register kernel is not built for
    dict(b=2, c=9, h=19, w=40, n=150, pooled=3, scale=1 / 16.0, sr=2),        # many rois on few tiles: long per-tile lists
    dict(b=1, c=3, h=10, w=33, n=65, pooled=2, scale=1 / 8.0, sr=0),          # lists longer than one batch
    dict(b=2, c=12, h=38, w=125, n=520, pooled=7, scale=1 / 16.0, sr=0),      # many rois, two images
    dict(b=1, c=70, h=24, w=40, n=90, pooled=14, scale=1 / 64.0, sr=2),       # 70 channels: a wave's block of 64 and a partial one (the channel-last copy pads to 72)
])
def test_hip_roi_align(cfg, route):
    from eval_driving_safety_amd import ops
    rs = np.random.RandomState(cfg["h"] + cfg["n"])
    feat = rs.randn(cfg["b"], cfg["c"], cfg["h"], cfg["w"]).astype(np.float32)
    rois = _rois(rs, cfg["n"], cfg["b"], cfg["h"] / cfg["scale"], cfg["w"] / cfg["scale"])
    dev = torch.device("cuda", 0)
    tf, tr = torch.tensor(feat, device=dev), torch.tensor(rois, device=dev)
    out = ops.roi_align(tf, tr, cfg["pooled"], cfg["scale"], cfg["sr"])
    want = O.roi_align(feat, rois, cfg["pooled"], cfg["scale"], cfg["sr"])
    assert out.cpu().numpy().tobytes() == want.tobytes(), "forward not bit-exact"
    with route(ADV_ROI_FWD_DIRECT="1"):                # four single gathers per sample: the same bits as the paired loads
        assert torch.equal(ops.roi_align(tf, tr, cfg["pooled"], cfg["scale"], cfg["sr"]), out)
    g = rs.randn(*want.shape).astype(np.float32)
    gf = ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"])
    wf = O.roi_align_bwd(g, rois, feat.shape, cfg["scale"], cfg["sr"])
    np.testing.assert_allclose(gf.cpu().numpy(), wf, rtol=2e-5, atol=2e-5)
    # deterministic gather: the float32 sum in the fixed order (roi, sample row, sample column, tap) - bit for bit,
    # and the same bits on a second run (the scatter-with-atomics formulation gives neither)
    # (more than 64 rois: G = ops.roi_align_bwd_segments(n) runs of consecutive roi indices summed separately, then added in run order)
    G = ops.roi_align_bwd_segments(cfg["n"])
    assert G == (1 if cfg["n"] <= 1024 else min(8, (cfg["n"] + 511) // 512)) and ops.roi_align_bwd_segments(1025) == 3
    assert gf.cpu().numpy().tobytes() == O.roi_align_bwd_ordered(g, rois, feat.shape, cfg["scale"], cfg["sr"], segments=G).tobytes(), "backward not bit-exact"
    again = ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"])
    assert torch.equal(gf, again)
    one = gf
    if G > 1:                                          # the single ordered sum of the same rois (test hook), for the register routes below
        with route(ADV_ROI_SEGMENTS="1"):
            one = ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"])
        assert one.cpu().numpy().tobytes() == O.roi_align_bwd_ordered(g, rois, feat.shape, cfg["scale"], cfg["sr"]).tobytes()
    with route(ADV_ROI_SEGMENTS="3"):                  # ... and three segments whatever the count: empty (tile, segment) pairs are skipped exactly
        three = ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"])
    assert three.cpu().numpy().tobytes() == O.roi_align_bwd_ordered(g, rois, feat.shape, cfg["scale"], cfg["sr"], segments=3).tobytes()
    # the shipped route keeps per-roi axis tables and gives a pixel to a wave (round 5); round 4's (accumulators in LDS, the lanes handed
    # work items) and round 3's register formulation, with the map-size-dependent and with eight channels per lane: the same bits
    with route(ADV_ROI_BWD_LDS="1"):
        assert torch.equal(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"]), gf)
    with route(ADV_ROI_BWD_SCALAR_ITEMS="1"):          # one channel per work item (round 3) instead of four: the same bits
        assert torch.equal(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"]), gf)
    with route(ADV_ROI_BWD_NO_STAGE="1"):              # grad_out gathered from global memory instead of the LDS-staged block
        assert torch.equal(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"]), gf)
    with route(ADV_ROI_BWD_REGS="1"):                  # (always one segment)
        assert torch.equal(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"]), one)
    with route(ADV_ROI_BWD_REGS="1", ADV_ROI_BWD_CB8="1"):
        assert torch.equal(ops.roi_align_bwd(torch.tensor(g, device=dev), tr, feat.shape, cfg["scale"], cfg["sr"]), one)
    # the channel-last copy of grad_out made by the caller (a pyramid's levels share it): the same bytes
    gcl = ops.roi_gout_channel_last(torch.tensor(g, device=dev))
    assert torch.equal(ops.roi_align_bwd_cl(gcl, tuple(g.shape[2:]), tr, feat.shape, cfg["scale"], cfg["sr"]), gf)
    # autograd wrapper
    tf2 = tf.clone().requires_grad_(True)
    o2 = ops.RoIAlign.apply(tf2, tr, cfg["pooled"], cfg["scale"], cfg["sr"])
    (o2 * torch.tensor(g, device=dev)).sum().backward()
    np.testing.assert_allclose(tf2.grad.cpu().numpy(), wf, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 5, 64, 65, 300, 2000, 4001, 6200])      # (6200 boxes: beyond the block scan's LDS image - the per-box scan)
def test_hip_nms_exact_indices(n, route):
    from eval_driving_safety_amd import ops
    rs = np.random.RandomState(n)
    ctr = rs.rand(n, 2) * 300
    wh = rs.rand(n, 2) * 80 + 4
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    if n >= 5:
        boxes[3] = boxes[0]                                  # exact duplicate: IoU 1
        boxes[4] = boxes[1] + np.float32(0.25)
    scores = np.sort(rs.rand(n).astype(np.float32))[::-1].copy()
    dev = torch.device("cuda", 0)
    for thresh in (0.3, 0.5):
        keep = ops.nms(torch.tensor(boxes, device=dev).reshape(n, 4), torch.tensor(scores, device=dev), thresh)
        assert keep.cpu().numpy().tolist() == O.nms(boxes.reshape(n, 4), thresh).tolist(), (n, thresh)
        with route(ADV_NMS_BOX_SCAN="1"):                    # round 1's scan, one box per step: the same indices
            assert torch.equal(ops.nms(torch.tensor(boxes, device=dev).reshape(n, 4), torch.tensor(scores, device=dev), thresh), keep)
