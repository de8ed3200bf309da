"""Box arithmetic of the proposal / target stage (ops.box_*): the oracle against the torch one-liners on the CPU; the kernels against the
same one-liners ON THE DEVICE (bit for bit - same expressions, the device's logf / expf) and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O


def _boxes(rs, n, w=1987.0, h=600.0):
    x1, y1 = rs.rand(n) * w * 0.9, rs.rand(n) * h * 0.9
    bw, bh = rs.rand(n) * 300 + 1, rs.rand(n) * 200 + 1
    return np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)


def test_oracle_boxes_follow_the_torch_one_liners():
    from eval_driving_safety_amd import surrogates as S
    rs = np.random.RandomState(0)
    a, g = _boxes(rs, 200), _boxes(rs, 7)
    assert O.box_iou(a, g).tobytes() == S._iou(torch.tensor(a), torch.tensor(g)).numpy().tobytes()
    arg = rs.randint(0, 7, 200)
    np.testing.assert_allclose(O.box_encode(a, g[arg]), S._encode(torch.tensor(a), torch.tensor(g[arg])).numpy(), rtol=1e-6, atol=1e-7)
    d = (rs.randn(200, 4) * 0.5).astype(np.float32)
    d[0, 2] = 9.0                                          # clamped at 4
    want = S._decode(torch.tensor(a), torch.tensor(d))
    want[:, 0::2].clamp_(0, 1987.0 - 1)
    want[:, 1::2].clamp_(0, 600.0 - 1)
    np.testing.assert_allclose(O.box_decode_clip(a, d, 1987.0, 600.0), want.numpy(), rtol=1e-6, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(1, 1), (300, 3), (4097, 30)])
def test_hip_boxes_are_the_one_liners_on_the_device(n, m):
    from eval_driving_safety_amd import ops, surrogates as S
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(n + m)
    a, gl = _boxes(rs, n), _boxes(rs, m)
    gr = gl.copy()
    gr[:, 0::2] -= (rs.rand(m, 1) * 30).astype(np.float32)
    if n > 2:
        a[1] = gl[0]                                       # IoU 1
        a[2] = a[1]
    if m > 2:
        gl[2] = gl[0]                                      # a tie between two ground-truth boxes: the first one wins
    ta, tl, tr = (torch.tensor(v, device=dev) for v in (a, gl, gr))
    iou, best, arg = ops.box_iou_rows(ta, tl)
    want = S._iou(ta, tl)
    wb, wa = want.max(1)
    assert torch.equal(iou, want) and torch.equal(best, wb) and torch.equal(arg, wa)
    assert iou.cpu().numpy().tobytes() == O.box_iou(a, gl).tobytes()
    none, b2, a2 = ops.box_iou_rows(ta, tl, want_matrix=False)
    assert none is None and torch.equal(b2, best) and torch.equal(a2, arg)
    # the six regression targets
    t6 = ops.box_encode6(ta, tl, tr, arg)
    el, er = S._encode(ta, tl[arg]), S._encode(ta, tr[arg])
    assert torch.equal(t6, torch.cat([el, er[:, 0:1], er[:, 2:3]], 1))
    np.testing.assert_allclose(t6[:, :4].cpu().numpy(), O.box_encode(a, gl[arg.cpu().numpy()]), rtol=2e-6, atol=1e-6)
    a_r = a.copy()
    a_r[:, 0::2] -= 7.0                                    # the right boxes of stereo rois: their own source
    t6r = ops.box_encode6(ta, tl, tr, arg, src_right=torch.tensor(a_r, device=dev))
    er = S._encode(torch.tensor(a_r, device=dev), tr[arg])
    assert torch.equal(t6r, torch.cat([el, er[:, 0:1], er[:, 2:3]], 1))
    # decoding + clipping + the minimum-size flags
    d = (rs.randn(n, 6) * 0.4).astype(np.float32)
    d[0, 2] = 11.0
    td = torch.tensor(d, device=dev)
    W, H, ms = 1987.0, 600.0, 8 * 1.6
    left, right, big = ops.box_decode_stereo(ta, td, W, H, ms)
    wl = S._decode(ta, td[:, :4])
    wr = S._decode(ta, torch.stack([td[:, 4], td[:, 1], td[:, 5], td[:, 3]], 1))
    for b in (wl, wr):
        b[:, 0::2].clamp_(0, W - 1)
        b[:, 1::2].clamp_(0, H - 1)
    wbig = ((wl[:, 2] - wl[:, 0] + 1 >= ms).long() * (wl[:, 3] - wl[:, 1] + 1 >= ms).long() * (wr[:, 2] - wr[:, 0] + 1 >= ms).long())
    assert torch.equal(left, wl) and torch.equal(right, wr) and torch.equal(big, wbig)
    np.testing.assert_allclose(left.cpu().numpy(), O.box_decode_clip(a, d[:, :4], W, H), rtol=2e-6, atol=2e-3)
    with pytest.raises(ValueError):
        ops.box_encode6(ta, tl, tr, arg.int())


@pytest.mark.gpu
@pytest.mark.parametrize("n,frac_big,n_gt", [(2000, 0.7, 3), (2000, 0.0, 2), (1500, 1.0, 0), (70, 0.3, 1), (1, 1.0, 0)])
def test_hip_proposal_partition_and_sampling_follow_the_torch_formulation(n, frac_big, n_gt):
    """the static forward's bookkeeping (stable partition by size, padded NMS list -> rois with the ground truth, sampled with replacement)
    as two launches against the tensor formulation they replace - gathers only: exact"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(n + n_gt)
    left, right = torch.tensor(_boxes(rs, n), device=dev), torch.tensor(_boxes(rs, n), device=dev)
    big = torch.tensor((rs.rand(n) < frac_big).astype(np.int64), device=dev)
    pl, pr, nvalid = ops.box_partition_stereo(left, right, big)
    nbig = big.sum()
    perm = torch.argsort(1 - big, stable=True)
    perm = torch.where(nbig > 0, perm, torch.arange(n, device=dev))
    assert torch.equal(pl, left[perm]) and torch.equal(pr, right[perm])
    assert int(nvalid) == (int(nbig) if int(nbig) > 0 else n)
    # a padded kept list: a prefix of valid indices, then entries beyond nvalid, then -1
    k = min(300, n)
    n_ok = min(k, int(rs.randint(0, k + 1)))
    valid = np.sort(rs.choice(int(nvalid), size=min(n_ok, int(nvalid)), replace=False))
    tail = np.full(k - len(valid), -1, np.int64)
    if int(nvalid) < n and len(tail):
        tail[0] = n - 1                                      # a kept small box: beyond nvalid, not a candidate
    keep = torch.tensor(np.concatenate([valid, tail]).astype(np.int64), device=dev)
    gl = torch.tensor(_boxes(rs, n_gt), device=dev) if n_gt else None
    gr = torch.tensor(_boxes(rs, n_gt), device=dev) if n_gt else None
    R = 512
    rl, rr, ol, orr = ops.box_sample_rois(keep, nvalid, pl, pr, gl, gr, R)
    nkeep = ((keep >= 0).long() * (keep < nvalid).long()).sum()
    kc = keep.clamp(min=0)
    cand_l, cand_r = pl[kc], pr[kc]
    if n_gt:
        cand_l, cand_r = torch.cat([gl, cand_l]), torch.cat([gr, cand_r])
    idx = torch.arange(R, device=dev) % (nkeep + n_gt).clamp(min=1)
    wl, wr = cand_l[idx], cand_r[idx]
    z = wl.new_zeros((R, 1))
    assert torch.equal(ol, wl) and torch.equal(orr, wr) and torch.equal(rl, torch.cat([z, wl], 1)) and torch.equal(rr, torch.cat([z, wr], 1))


@pytest.mark.gpu
@pytest.mark.parametrize("bounded", [True, False])
def test_hip_rpn_head_pack_is_the_permute_reshape_cat_formulation(bounded):
    """ops.RpnHeadPack against slices + 0.5 * tanh + permute(0, 2, 3, 1).reshape + cat over the levels: the same lists and, through autograd, the
    same gradient w.r.t. every level's head output - bit for bit"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(4)
    A = 3
    shapes = [(1, 7 * A, 19, 31), (2, 7 * A, 10, 16), (1, 7 * A, 1, 1)]
    heads = [(torch.randn(s, generator=gen) * 1.5).to(dev).requires_grad_(True) for s in shapes]
    refs = [h.detach().clone().requires_grad_(True) for h in heads]
    scores, deltas = ops.RpnHeadPack.apply(A, bounded, *heads)
    ws, wd = [], []
    for h in refs:
        s, d = h.split([A, 6 * A], 1)
        if bounded:
            d = 0.5 * torch.tanh(d)
        ws.append(s.permute(0, 2, 3, 1).reshape(-1))
        wd.append(d.permute(0, 2, 3, 1).reshape(-1, 6))
    ws, wd = torch.cat(ws), torch.cat(wd)
    assert torch.equal(scores, ws) and torch.equal(deltas, wd)
    gs, gd = torch.randn(ws.shape, generator=gen).to(dev), torch.randn(wd.shape, generator=gen).to(dev)
    ((scores * gs).sum() + (deltas * gd).sum()).backward()
    ((ws * gs).sum() + (wd * gd).sum()).backward()
    for h, r in zip(heads, refs):
        assert torch.equal(h.grad, r.grad)
    only_scores, _ = ops.RpnHeadPack.apply(A, bounded, *[h.detach().requires_grad_(True) for h in heads])
    only_scores.sum().backward()                              # (no gradient arrives for the deltas: zeros there)


@pytest.mark.gpu
def test_hip_objective_chain_is_the_scripts_loop_on_the_device():
    """ops.ObjectiveChain against the six-fold loop of attack/Stereo-RCNN/pgd_attack.py:165-171 (``loss = loss + term.mean() * exp(-u) + u``)
    run with torch operators on the device: the same loss and the same gradients w.r.t. the six terms, bit for bit"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(8)
    for trial in range(5):
        vals = (torch.rand((6,), generator=gen) * 3).to(dev)
        u = (torch.randn((6,), generator=gen) * 0.7).to(dev)
        a = [vals[k:k + 1].clone().requires_grad_(True) for k in range(6)]
        b = [vals[k:k + 1].clone().requires_grad_(True) for k in range(6)]
        loss = ops.ObjectiveChain.apply(torch.cat([t.reshape(1) for t in a]), u)
        want = 0.
        for k in range(6):
            want = want + b[k].mean() * torch.exp(-u[k]) + u[k]
        assert loss.shape == want.shape and torch.equal(loss, want)
        (loss * 1.7).backward()
        (want * 1.7).backward()
        for k in range(6):
            assert torch.equal(a[k].grad, b[k].grad)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,bce", [(5, 6, False), (512, 5, False), (300000, 6, False), (300000, 1, True), (7, 1, True)])
def test_hip_masked_loss_mean_gradient_is_torchs_and_the_value_agrees(n, k, bce):
    """ops.MaskedLossMean against the tensor formulation on the device: the gradient w.r.t. the prediction bit for bit (it does not depend on
    the order of the sum), the value within float32 summation error; an all-zero weight gives 0 / max(0, 1)"""
    import torch.nn.functional as F
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(n + k)
    shape = (n,) if bce else (n, k)
    pred = (torch.randn(shape, generator=gen) * 2).to(dev).requires_grad_(True)
    ref = pred.detach().clone().requires_grad_(True)
    w = (torch.rand((n,), generator=gen) < 0.3).float().to(dev)
    if bce:
        target = (torch.rand(shape, generator=gen) < 0.5).float().to(dev)
        got = ops.masked_bce_mean(pred, target, w)
        want = (F.binary_cross_entropy_with_logits(ref, target, reduction="none") * w).sum() / w.sum().clamp(min=1.0)
    else:
        target = (torch.randn(shape, generator=gen) * 2).to(dev)
        got = ops.masked_smooth_l1_mean(pred, target, w, float(k))
        want = (F.smooth_l1_loss(ref, target, reduction="none") * w[:, None]).sum() / (float(k) * w.sum()).clamp(min=1.0)
    assert got.shape == want.shape and abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    (got * 1.3).backward()
    (want * 1.3).backward()
    assert torch.equal(pred.grad, ref.grad)
    again = ops.MaskedLossMean.apply(pred.detach(), target, w, float(k), bce)
    assert float(again) == float(got)                       # deterministic
    zero = ops.MaskedLossMean.apply(pred.detach(), target, torch.zeros_like(w), float(k), bce)
    assert float(zero) == 0.0
