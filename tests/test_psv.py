"""K7 plane-sweep cost volume.  CPU: the oracle's forward against a torch slicing formulation and its
backward against torch autograd + the adjoint identity.  GPU: the HIP kernels against the oracle,
bit for bit (the backward sums planes in the same order)."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O


def _inputs(b, c, d, h, w, seed=0, max_shift=None):
    rs = np.random.RandomState(seed)
    left = rs.randn(b, c, h, w).astype(np.float32)
    right = rs.randn(b, c, h, w).astype(np.float32)
    max_shift = w if max_shift is None else max_shift
    shift = np.sort(rs.randint(0, max_shift + 1, size=(b, d)).astype(np.int32), axis=1)[:, ::-1].copy()
    return left, right, shift


def _torch_psv(left, right, shift):
    b, c, h, w = left.shape
    d = shift.shape[1]
    cost = left.new_zeros((b, 2 * c, d, h, w))
    for bi in range(b):
        for di in range(d):
            s = int(shift[bi, di])
            if s < w:
                cost[bi, :c, di, :, s:] = left[bi, :, :, s:]
                cost[bi, c:, di, :, s:] = right[bi, :, :, :w - s]
    return cost


@pytest.mark.parametrize("shape", [(1, 3, 5, 4, 16), (2, 2, 7, 3, 12), (1, 4, 6, 5, 10)])
def test_oracle_forward_backward_consistency(shape):
    b, c, d, h, w = shape
    left, right, shift = _inputs(b, c, d, h, w, seed=sum(shape))
    cost = O.psv_build(left, right, shift)
    tl, tr = torch.tensor(left, requires_grad=True), torch.tensor(right, requires_grad=True)
    tcost = _torch_psv(tl, tr, shift)
    assert np.array_equal(cost, tcost.detach().numpy())
    g = np.random.RandomState(1).randn(*cost.shape).astype(np.float32)
    tcost.backward(torch.tensor(g))
    gl, gr = O.psv_build_bwd(g, shift)
    np.testing.assert_allclose(gl, tl.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gr, tr.grad.numpy(), rtol=1e-5, atol=1e-5)
    lhs = float((cost.astype(np.float64) * g).sum())                      # <A x, g> == <x, A^T g>
    rhs = float((left.astype(np.float64) * gl).sum() + (right.astype(np.float64) * gr).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


GPU_SHAPES = [
    (1, 32, 48, 96, 312),     # DSGN: C=32, D=48 planes, 1/4 of 384x1248
    (2, 8, 12, 10, 312),
    (1, 4, 9, 7, 40),         # rows not a multiple of the 4-row tile
    (2, 3, 5, 6, 44),
    (1, 2, 4, 3, 13),         # W % 4 != 0 -> scalar kernels
    (1, 5, 3, 9, 1024),       # 4 rows x 256 float4 = 1024 lanes, the largest vector tile
    (1, 2, 3, 5, 1100),       # too wide for one tile -> scalar kernels
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", GPU_SHAPES)
def test_hip_psv_matches_oracle(shape):
    from eval_driving_safety_amd import ops
    b, c, d, h, w = shape
    left, right, shift = _inputs(b, c, d, h, w, seed=sum(shape), max_shift=min(w, 60))
    shift[0, 0] = min(w, 60)
    shift[-1, -1] = 0
    if d > 2:
        shift[0, 1] = w                                                 # a plane that is entirely zero
    if d > 3:
        shift[0, 2] = -3                                                # negative: treated as 0
        shift[-1, 0] = w + 17                                           # beyond the row: treated as W
    dev = torch.device("cuda", 0)
    tl, tr, ts = torch.tensor(left, device=dev), torch.tensor(right, device=dev), torch.tensor(shift, device=dev)
    cost = ops.psv_build(tl, tr, ts)
    want = O.psv_build(left, right, shift)
    assert cost.cpu().numpy().tobytes() == want.tobytes(), "forward"
    g = np.random.RandomState(2).randn(*want.shape).astype(np.float32)
    gl, gr = ops.psv_build_bwd(torch.tensor(g, device=dev), ts)
    wl, wr = O.psv_build_bwd(g, shift)
    assert gl.cpu().numpy().tobytes() == wl.tobytes(), "grad_left"
    assert gr.cpu().numpy().tobytes() == wr.tobytes(), "grad_right"


@pytest.mark.gpu
def test_hip_psv_autograd_function():
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    left, right, shift = _inputs(1, 4, 6, 8, 24, seed=3, max_shift=12)
    tl = torch.tensor(left, device=dev, requires_grad=True)
    tr = torch.tensor(right, device=dev, requires_grad=True)
    cost = ops.PsvBuild.apply(tl, tr, torch.tensor(shift, device=dev))
    w = torch.tensor(np.random.RandomState(4).randn(*cost.shape).astype(np.float32), device=dev)
    (cost * w).sum().backward()
    wl, wr = O.psv_build_bwd(w.cpu().numpy(), shift)
    assert tl.grad.cpu().numpy().tobytes() == wl.tobytes() and tr.grad.cpu().numpy().tobytes() == wr.tobytes()


# ------------------------------------------------------------------------------------------ interpolating form
def _float_shifts(b, d, w, seed):
    rs = np.random.RandomState(seed)
    sf = np.sort((rs.rand(b, d) * min(w, 60)).astype(np.float32), axis=1)[:, ::-1].copy()
    sf[0, 0] = np.float32(7.0)                                           # integral: degenerates to the integer form
    sf[-1, -1] = np.float32(0.0)
    if d > 2:
        sf[0, 1] = np.float32(w)                                         # a plane that is entirely zero
    if d > 3:
        sf[0, 2] = np.float32(-2.5)                                      # negative: treated as 0
        sf[-1, 0] = np.float32(w + 17.3)                                 # beyond the row: treated as W
    if d > 4:
        sf[0, 3] = np.float32(w - 0.25)                                  # only the last column survives
    return sf


@pytest.mark.parametrize("shape", [(1, 3, 6, 4, 16), (2, 2, 7, 3, 12)])
def test_oracle_lerp_is_the_adjoint_pair_and_contains_the_integer_form(shape):
    b, c, d, h, w = shape
    left, right, shift = _inputs(b, c, d, h, w, seed=sum(shape))
    # integral float shifts reproduce the integer volume (up to the sign of zero) and its adjoint
    assert np.array_equal(O.psv_build_lerp(left, right, shift.astype(np.float32)), O.psv_build(left, right, shift))
    sf = _float_shifts(b, d, w, 5)
    cost = O.psv_build_lerp(left, right, sf)
    # against a direct definition: value of the zero-extended right row at x - sf by linear interpolation
    for bi in range(b):
        for di in range(d):
            s = float(np.clip(sf[bi, di], 0, w))
            for x in range(w):
                pos = x - s
                if x < np.ceil(s):
                    assert not cost[bi, :, di, :, x].any()
                    continue
                i0 = int(np.floor(pos + 1e-9)) if abs(pos - round(pos)) < 1e-6 else int(np.floor(pos))
                frac = pos - i0
                v = (1 - frac) * right[bi, :, :, i0] + (frac * right[bi, :, :, i0 + 1] if i0 + 1 < w and frac > 0 else 0)
                np.testing.assert_allclose(cost[bi, c:, di, :, x], v, rtol=1e-5, atol=1e-5)
                assert np.array_equal(cost[bi, :c, di, :, x], left[bi, :, :, x])
    g = np.random.RandomState(1).randn(*cost.shape).astype(np.float32)
    gl, gr = O.psv_build_lerp_bwd(g, sf)
    lhs = float((cost.astype(np.float64) * g).sum())
    rhs = float((left.astype(np.float64) * gl).sum() + (right.astype(np.float64) * gr).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", GPU_SHAPES)
def test_hip_psv_lerp_matches_oracle(shape):
    from eval_driving_safety_amd import ops
    b, c, d, h, w = shape
    left, right, _ = _inputs(b, c, d, h, w, seed=sum(shape))
    sf = _float_shifts(b, d, w, sum(shape))
    dev = torch.device("cuda", 0)
    tl, tr, ts = torch.tensor(left, device=dev), torch.tensor(right, device=dev), torch.tensor(sf, device=dev)
    cost = ops.psv_build_lerp(tl, tr, ts)
    want = O.psv_build_lerp(left, right, sf)
    assert cost.cpu().numpy().tobytes() == want.tobytes(), "forward"
    g = np.random.RandomState(2).randn(*want.shape).astype(np.float32)
    gl, gr = ops.psv_build_lerp_bwd(torch.tensor(g, device=dev), ts)
    wl, wr = O.psv_build_lerp_bwd(g, sf)
    assert gl.cpu().numpy().tobytes() == wl.tobytes(), "grad_left"
    assert gr.cpu().numpy().tobytes() == wr.tobytes(), "grad_right"
    # autograd face
    tl2, tr2 = tl.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    (ops.PsvBuildLerp.apply(tl2, tr2, ts) * torch.tensor(g, device=dev)).sum().backward()
    assert tl2.grad.cpu().numpy().tobytes() == wl.tobytes() and tr2.grad.cpu().numpy().tobytes() == wr.tobytes()


@pytest.mark.gpu
def test_hip_psv_batch_beyond_2_31_elements_equals_per_item_launches():
    """maximum sizes: a batch of 24 DSGN-shaped cost volumes (2.2e9 elements, 8.8 GB) built and back-propagated in one launch
    each equals the items done one by one - integer-shift and interpolating variants"""
    from eval_driving_safety_amd import ops
    if torch.cuda.mem_get_info(0)[0] < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free device memory")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(21)
    b, c, d, h, w = 24, 32, 48, 96, 312
    assert b * 2 * c * d * h * w > 2 ** 31
    left = torch.randn((b, c, h, w), device=dev, generator=gen)
    right = torch.randn((b, c, h, w), device=dev, generator=gen)
    shift_i = torch.randint(0, 60, (b, d), device=dev, generator=gen, dtype=torch.int32)
    shift_f = torch.rand((b, d), device=dev, generator=gen) * 60.0
    for build, bwd, shift in ((ops.psv_build, ops.psv_build_bwd, shift_i), (ops.psv_build_lerp, ops.psv_build_lerp_bwd, shift_f)):
        cost = build(left, right, shift)
        gl, gr = bwd(cost, shift)                       # the volume itself as the incoming gradient
        for i in (0, 11, b - 1):
            s = slice(i, i + 1)
            one = build(left[s].contiguous(), right[s].contiguous(), shift[s].contiguous())
            assert torch.equal(one, cost[s]), i
            gl1, gr1 = bwd(one, shift[s].contiguous())
            assert torch.equal(gl1, gl[s]) and torch.equal(gr1, gr[s]), i
        del cost, gl, gr
