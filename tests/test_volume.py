"""csrc/volume.hip - what a plane-sweep detector does after its 3D convolutions (upstream DSGN code reached at
attack/DSGN/pgd_attack.py:308,324): fused depth regression (trilinear upsample + softmax + expectation), grid_sample on
5-D volumes (PSV -> 3D geometric volume) with a deterministic gather backward, sigmoid focal loss.
CPU part: the oracle against torch's own operators.  GPU part: the HIP kernels against the oracle / torch.
Tolerances: exp/log are library functions (device vs host) - 1e-5 relative of the result's scale; grid_sample forward and
backward involve no transcendental and are compared bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_np as O


def _cost(rs, b, d, h, w):
    return (rs.randn(b, d, h, w) * 2).astype(np.float32)


DR_CASES = [dict(b=1, d=6, h=5, w=7, out=(24, 20, 28), align=False), dict(b=2, d=4, h=3, w=5, out=(9, 7, 11), align=True),
            dict(b=1, d=5, h=4, w=6, out=(5, 4, 6), align=False), dict(b=1, d=8, h=6, w=9, out=(4, 3, 5), align=False),
            dict(b=1, d=1, h=1, w=2, out=(3, 2, 5), align=True)]


@pytest.mark.parametrize("cfg", DR_CASES)
def test_oracle_depth_regress_matches_torch(cfg):
    rs = np.random.RandomState(cfg["d"] * 7 + cfg["w"])
    cost = _cost(rs, cfg["b"], cfg["d"], cfg["h"], cfg["w"])
    zv = np.linspace(2.0, 40.4, cfg["out"][0]).astype(np.float32)
    up = O.trilinear_upsample(cost, cfg["out"], cfg["align"])
    tc = torch.tensor(cost, requires_grad=True)
    tup = F.interpolate(tc[:, None], size=cfg["out"], mode="trilinear", align_corners=cfg["align"])[:, 0]
    np.testing.assert_allclose(up, tup.detach().numpy(), rtol=1e-5, atol=1e-5)
    depth, stats = O.depth_regress(cost, zv, cfg["out"], cfg["align"])
    tdepth = (torch.softmax(tup, 1) * torch.tensor(zv).view(1, -1, 1, 1)).sum(1)
    np.testing.assert_allclose(depth, tdepth.detach().numpy(), rtol=1e-5, atol=1e-4)
    g = rs.randn(*depth.shape).astype(np.float32)
    tdepth.backward(torch.tensor(g))
    np.testing.assert_allclose(O.depth_regress_bwd(cost, zv, g, cfg["out"], cfg["align"]), tc.grad.numpy(), rtol=1e-4, atol=1e-4)


def _grid(rs, b, zo, yo, xo, spread=1.15):
    return (rs.rand(b, zo, yo, xo, 3) * 2 * spread - spread).astype(np.float32)      # some samples leave the volume


GS_CASES = [dict(b=2, c=70, dims=(3, 4, 5), out=(4, 3, 6), align=False), dict(b=1, c=3, dims=(4, 5, 6), out=(3, 4, 5), align=False), dict(b=2, c=2, dims=(3, 7, 5), out=(6, 2, 9), align=True),
            dict(b=1, c=5, dims=(1, 1, 1), out=(2, 2, 2), align=False), dict(b=1, c=1, dims=(6, 4, 9), out=(5, 5, 5), align=False)]


@pytest.mark.parametrize("cfg", GS_CASES)
def test_oracle_grid_sample3d_equals_torch_bit_for_bit(cfg):
    rs = np.random.RandomState(sum(cfg["dims"]) + cfg["c"])
    vol = rs.randn(cfg["b"], cfg["c"], *cfg["dims"]).astype(np.float32)
    grid = _grid(rs, cfg["b"], *cfg["out"])
    grid[0, 0, 0, 0] = [-1, -1, -1]
    grid[0, -1, -1, -1] = [1, 1, 1]
    want = F.grid_sample(torch.tensor(vol), torch.tensor(grid), mode="bilinear", padding_mode="zeros", align_corners=cfg["align"])
    got = O.grid_sample3d(vol, grid, cfg["align"])
    assert got.tobytes() == want.numpy().tobytes()
    tv = torch.tensor(vol, requires_grad=True)
    g = rs.randn(*got.shape).astype(np.float32)
    F.grid_sample(tv, torch.tensor(grid), mode="bilinear", padding_mode="zeros", align_corners=cfg["align"]).backward(torch.tensor(g))
    np.testing.assert_allclose(O.grid_sample3d_bwd(g, grid, cfg["dims"], cfg["align"]), tv.grad.numpy(), rtol=1e-5, atol=1e-5)


def _focal_torch(x, t, gamma, alpha):
    k = x.shape[1]
    cls = torch.arange(1, k + 1).view(1, -1)
    tt = t.view(-1, 1)
    p = torch.sigmoid(x)
    pos = (tt == cls).double()
    neg = ((tt != cls) & (tt >= 0)).double()
    return -(pos * alpha * (1 - p) ** gamma * torch.log(p) + neg * (1 - alpha) * p ** gamma * torch.log(1 - p))


def test_oracle_focal_loss_matches_autograd():
    rs = np.random.RandomState(3)
    x = (rs.randn(40, 3) * 3).astype(np.float64)
    t = rs.randint(-1, 4, 40)
    tx = torch.tensor(x, requires_grad=True)
    loss = _focal_torch(tx, torch.tensor(t), 2.0, 0.25)
    loss.sum().backward()
    l, g = O.sigmoid_focal_loss(x, t, 2.0, 0.25)
    np.testing.assert_allclose(l, loss.detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g, tx.grad.numpy(), rtol=1e-8, atol=1e-12)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", DR_CASES + [dict(b=1, d=12, h=12, w=39, out=(24, 48, 156), align=False)])
def test_hip_depth_regress_fwd_bwd(cfg):
    from eval_driving_safety_amd import ops
    rs = np.random.RandomState(cfg["d"] * 7 + cfg["w"])
    cost = _cost(rs, cfg["b"], cfg["d"], cfg["h"], cfg["w"])
    zv = np.linspace(2.0, 40.4, cfg["out"][0]).astype(np.float32)
    dev = torch.device("cuda", 0)
    tc, tz = torch.tensor(cost, device=dev), torch.tensor(zv, device=dev)
    depth, stats = ops.depth_regress(tc, tz, cfg["out"], cfg["align"], with_stats=True)
    want, wstats = O.depth_regress(cost, zv, cfg["out"], cfg["align"])
    np.testing.assert_allclose(depth.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * 40)
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), wstats[:, 0], rtol=1e-6, atol=1e-6)     # the softmax maximum: no exp involved
    g = rs.randn(*want.shape).astype(np.float32)
    gc = ops.depth_regress_bwd(tc, tz, depth, stats, torch.tensor(g, device=dev), cfg["align"])
    ref = O.depth_regress_bwd(cost, zv, g, cfg["out"], cfg["align"])
    np.testing.assert_allclose(gc.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    assert torch.equal(gc, ops.depth_regress_bwd(tc, tz, depth, stats, torch.tensor(g, device=dev), cfg["align"])), "not reproducible"
    # autograd wrapper against torch's unfused formulation on the GPU
    tcr = tc.clone().requires_grad_(True)
    up = F.interpolate(tcr[:, None], size=cfg["out"], mode="trilinear", align_corners=cfg["align"])[:, 0]
    ((torch.softmax(up, 1) * tz.view(1, -1, 1, 1)).sum(1) * torch.tensor(g, device=dev)).sum().backward()
    tcm = tc.clone().requires_grad_(True)
    (ops.DepthRegress.apply(tcm, tz, cfg["out"], cfg["align"]) * torch.tensor(g, device=dev)).sum().backward()
    scale = max(1.0, float(tcr.grad.abs().max()))
    assert float((tcm.grad - tcr.grad).abs().max()) <= 2e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", GS_CASES + [dict(b=2, c=9, dims=(12, 24, 39), out=(24, 5, 38), align=False)])
def test_hip_grid_sample3d_bit_exact_and_deterministic_backward(cfg):
    from eval_driving_safety_amd import ops
    rs = np.random.RandomState(sum(cfg["dims"]) + cfg["c"])
    vol = rs.randn(cfg["b"], cfg["c"], *cfg["dims"]).astype(np.float32)
    grid = _grid(rs, cfg["b"], *cfg["out"])
    grid[0, 0, 0, 0] = [-1, -1, -1]
    grid[0, -1, -1, -1] = [1, 1, 1]
    dev = torch.device("cuda", 0)
    tv, tg = torch.tensor(vol, device=dev), torch.tensor(grid, device=dev)
    out = ops.grid_sample3d(tv, tg, cfg["align"])
    want = F.grid_sample(torch.tensor(vol), torch.tensor(grid), mode="bilinear", padding_mode="zeros", align_corners=cfg["align"]).numpy()
    assert out.cpu().numpy().tobytes() == want.tobytes(), "forward differs from torch-CPU grid_sample"
    g = rs.randn(*want.shape).astype(np.float32)
    plan = ops.GridSamplePlan(tg, cfg["dims"], cfg["align"])
    gv = ops.grid_sample3d_bwd(torch.tensor(g, device=dev), plan)
    assert gv.cpu().numpy().tobytes() == O.grid_sample3d_bwd(g, grid, cfg["dims"], cfg["align"]).tobytes(), "backward differs from the ordered oracle"
    assert torch.equal(ops.grid_sample3d_bwd(torch.tensor(g, device=dev), plan, channels_last=False), gv), "NCDHW gather vs channels-last gather"
    plan2 = ops.GridSamplePlan(tg, cfg["dims"], cfg["align"])                       # the plan itself is reproducible
    assert torch.equal(plan.buf[:int(np.prod(cfg["dims"])) * cfg["b"] + 1], plan2.buf[:int(np.prod(cfg["dims"])) * cfg["b"] + 1])
    assert torch.equal(ops.grid_sample3d_bwd(torch.tensor(g, device=dev), plan2), gv)
    tvr = tv.clone().requires_grad_(True)
    (ops.GridSample3d.apply(tvr, tg, plan) * torch.tensor(g, device=dev)).sum().backward()
    tvt = torch.tensor(vol, requires_grad=True)
    F.grid_sample(tvt, torch.tensor(grid), mode="bilinear", padding_mode="zeros", align_corners=cfg["align"]).backward(torch.tensor(g))
    np.testing.assert_allclose(tvr.grad.cpu().numpy(), tvt.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_grid_sample3d_beyond_2_31_elements_equals_channel_chunks():
    """maximum sizes: a volume of 2.3e9 elements (element offsets beyond 32 bits) sampled in one launch, and its backward (both
    gather variants), equal the same work done on channel chunks whose offsets fit"""
    from eval_driving_safety_amd import ops
    if torch.cuda.mem_get_info(0)[0] < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free device memory")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    b, c, dims, out = 2, 800, (48, 96, 312), (96, 10, 152)
    assert b * c * dims[0] * dims[1] * dims[2] > 2 ** 31
    vol = torch.empty((b, c) + dims, device=dev)
    for i in range(b):
        vol[i].normal_(generator=gen)
    grid = torch.rand((b,) + out + (3,), device=dev, generator=gen) * 2.2 - 1.1
    got = ops.grid_sample3d(vol, grid, True)
    plan = ops.GridSamplePlan(grid, dims, True)
    g = torch.randn(got.shape, device=dev, generator=gen)
    gv = ops.grid_sample3d_bwd(g, plan)
    assert torch.equal(ops.grid_sample3d_bwd(g, plan, channels_last=False), gv)
    for lo in (0, 400):
        ch = slice(lo, lo + 400)
        assert torch.equal(got[:, ch], ops.grid_sample3d(vol[:, ch].contiguous(), grid, True)), lo
        assert torch.equal(gv[:, ch], ops.grid_sample3d_bwd(g[:, ch].contiguous(), plan)), lo


@pytest.mark.gpu
def test_hip_sigmoid_focal_loss():
    from eval_driving_safety_amd import ops
    rs = np.random.RandomState(5)
    x = (rs.randn(3000, 3) * 4).astype(np.float32)
    x[0] = [40, -40, 0]
    t = rs.randint(-1, 4, 3000).astype(np.int32)
    dev = torch.device("cuda", 0)
    tx, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    loss, grad = ops.sigmoid_focal_loss(tx, tt, 2.0, 0.25, want_grad=True)
    wl, wg = O.sigmoid_focal_loss(x, t, 2.0, 0.25)
    np.testing.assert_allclose(loss.cpu().numpy(), wl, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), wg, rtol=2e-5, atol=1e-7)
    txr = tx.clone().requires_grad_(True)
    ops.SigmoidFocalLoss.apply(txr, tt, 2.0, 0.25).backward()
    np.testing.assert_allclose(txr.grad.cpu().numpy(), wg, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 5, 4096, 100003])
def test_hip_relu_backward(n):
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(n)
    g = torch.randn((n,), device=dev, generator=gen)
    y = torch.relu(torch.randn((n,), device=dev, generator=gen))
    if n > 3:
        y[1], g[1] = 0.0, float("inf")           # threshold_backward semantics: a dead unit passes nothing, not inf * 0
    want = torch.where(y > 0, g, torch.zeros_like(g))
    assert torch.equal(ops.relu_backward(g, y), want)
    if n > 8:                                    # a view that is only 4-byte aligned takes the scalar path
        assert torch.equal(ops.relu_backward(g[1:], y[1:].contiguous()), want[1:])


BEV_CASES = [dict(shape=(1, 3, 5, 8, 7), pool=4), dict(shape=(2, 4, 6, 20, 12), pool=4), dict(shape=(1, 2, 3, 7, 9), pool=3), dict(shape=(1, 5, 4, 6, 33), pool=1)]


@pytest.mark.parametrize("cfg", BEV_CASES)
def test_oracle_bev_fold_equals_torch(cfg):
    """the bird's-eye-view fold is F.avg_pool3d(v, (1, P, 1)) -> permute(0, 1, 3, 2, 4) -> reshape: the same VALUES (a sum of P floats in
    ascending order, divided by P), and its backward is autograd's"""
    rs = np.random.RandomState(sum(cfg["shape"]))
    v = rs.randn(*cfg["shape"]).astype(np.float32)
    p = cfg["pool"]
    b, c, z, y, x = cfg["shape"]
    t = torch.tensor(v, requires_grad=True)
    ref = F.avg_pool3d(t, (1, p, 1)).permute(0, 1, 3, 2, 4).reshape(b, c * (y // p), z, x)
    got = O.bev_fold(v, p)
    assert got.shape == tuple(ref.shape) and np.allclose(got, ref.detach().numpy(), rtol=1e-6, atol=1e-7)
    g = rs.randn(*got.shape).astype(np.float32)
    ref.backward(torch.tensor(g))
    assert np.allclose(O.bev_fold_bwd(g, cfg["shape"], p), t.grad.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", BEV_CASES + [dict(shape=(1, 16, 24, 20, 76), pool=4)])
def test_hip_bev_fold_fwd_bwd_bit_exact(cfg, route):
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(sum(cfg["shape"]) + 1)
    v = rs.randn(*cfg["shape"]).astype(np.float32)
    p = cfg["pool"]
    tv = torch.tensor(v, device=dev, requires_grad=True)
    out = ops.BevFold.apply(tv, p)
    assert out.detach().cpu().numpy().tobytes() == O.bev_fold(v, p).tobytes()
    g = rs.randn(*out.shape).astype(np.float32)
    out.backward(torch.tensor(g, device=dev))
    assert tv.grad.cpu().numpy().tobytes() == O.bev_fold_bwd(g, cfg["shape"], p).tobytes()
    # the producer's ReLU mask inside the fold's backward (mask_input): grad where v > 0, zero elsewhere
    tv2 = torch.tensor(v, device=dev, requires_grad=True)
    ops.BevFold.apply(tv2, p, True).backward(torch.tensor(g, device=dev))
    assert tv2.grad.cpu().numpy().tobytes() == O.bev_fold_bwd(g, cfg["shape"], p, mask=v).tobytes()
    with route(ADV_BEV_SCALAR="1"):                    # widths that are multiples of four take 16-byte loads: the one-float version gives the same bytes
        tv3 = torch.tensor(v, device=dev, requires_grad=True)
        ops.BevFold.apply(tv3, p, True).backward(torch.tensor(g, device=dev))
        assert torch.equal(tv3.grad, tv2.grad)
    with pytest.raises(ValueError):
        ops.bev_fold(tv.detach(), v.shape[3] + 1)
