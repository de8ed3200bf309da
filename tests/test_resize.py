"""csrc/resize.hip - the bilinear up-sampling of the FPN top-down path (attack/Stereo-RCNN/stereo_rcnn.py:92-108 ``_upsample_add``:
F.upsample(x, size=(H, W), mode='bilinear') + y) and its adjoint as a fixed-order gather.
CPU part: the oracle against torch's operator and its autograd.  GPU part: the HIP kernels against the oracle, bit for bit (products and
sums only - no library function), against torch within float32 rounding, reproducibility, and - at the pyramid's full sizes - the
adjoint identity <up(x), g> = <x, up^T(g)>."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_np as O

# (leading dims, input size, output size): the pyramid's ragged ~2x steps, an exact 2x, the same size, a 1-pixel map, a large ratio
# (more candidates than the kernel keeps in registers), and a down-sampling (legal, never used by the path)
CASES = [((1, 2), (5, 8), (10, 15)), ((2, 3), (4, 6), (8, 12)), ((1, 1), (7, 9), (7, 9)), ((1, 2), (1, 1), (3, 5)),
         ((1, 2), (2, 3), (13, 31)), ((1, 1), (10, 13), (19, 25)), ((1, 2), (9, 12), (5, 7)), ((2, 2), (3, 1), (7, 2))]


def _ids(c):
    return "%dx%d-to-%dx%d" % (c[1] + c[2])


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_oracle_bilinear_up_matches_torch(case):
    lead, hw, size = case
    rs = np.random.RandomState(sum(hw) + 3 * sum(size))
    x = rs.randn(*lead, *hw).astype(np.float32)
    g = rs.randn(*lead, *size).astype(np.float32)
    t = torch.tensor(x, requires_grad=True)
    up = F.interpolate(t, size=size, mode="bilinear", align_corners=False)
    np.testing.assert_allclose(O.bilinear_up(x, size), up.detach().numpy(), rtol=1e-6, atol=1e-6)
    up.backward(torch.tensor(g))
    np.testing.assert_allclose(O.bilinear_up_bwd(g, hw), t.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_oracle_bilinear_up_is_the_adjoint_pair():
    rs = np.random.RandomState(5)
    x, g = rs.randn(1, 3, 6, 7).astype(np.float32), rs.randn(1, 3, 11, 15).astype(np.float32)
    lhs = float((O.bilinear_up(x, (11, 15)).astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * O.bilinear_up_bwd(g, (6, 7))).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


def _fixture_cases(golden, golden_index):
    z = golden("upsample_add")
    for i, c in enumerate(golden_index["upsample_add"]["cases"]):
        yield i, tuple(c["in"]), tuple(c["out"]), z


def test_oracle_against_the_references_upsample_add(golden, golden_index):
    """tests/golden/upsample_add.npz: the reference's ``_upsample_add`` (attack/Stereo-RCNN/stereo_rcnn.py:91-108) executed on seeded maps
    with its backward - the oracle's up-sampling + y and adjoint within float32 rounding of it, and adopt._upsample_add (what adopt()
    rebinds the method to) equal to it bit for bit on the CPU"""
    from eval_driving_safety_amd import adopt
    for i, hw, size, z in _fixture_cases(golden, golden_index):
        x, y, g = z["x%d" % i], z["y%d" % i], z["g%d" % i]
        assert x.shape[2:] == hw and y.shape[2:] == size
        np.testing.assert_allclose(O.bilinear_up(x, size) + y, z["out%d" % i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(O.bilinear_up_bwd(g, hw), z["grad_x%d" % i], rtol=0, atol=2e-5)
        t = torch.tensor(x, requires_grad=True)
        res = adopt._upsample_add(t, torch.tensor(y))
        res.backward(torch.tensor(g))
        assert res.detach().numpy().tobytes() == z["out%d" % i].tobytes() and t.grad.numpy().tobytes() == z["grad_x%d" % i].tobytes()


@pytest.mark.gpu
def test_hip_against_the_references_upsample_add(golden, golden_index):
    """the same fixture through csrc/resize.hip (ops.BilinearUp + y, as adopt() and the surrogates call it on the GPU)"""
    from eval_driving_safety_amd import adopt
    dev = torch.device("cuda", 0)
    for i, hw, size, z in _fixture_cases(golden, golden_index):
        t = torch.tensor(z["x%d" % i], device=dev, requires_grad=True)
        res = adopt._upsample_add(t, torch.tensor(z["y%d" % i], device=dev))
        res.backward(torch.tensor(z["g%d" % i], device=dev))
        np.testing.assert_allclose(res.detach().cpu().numpy(), z["out%d" % i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(t.grad.cpu().numpy(), z["grad_x%d" % i], rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_hip_bilinear_up_fwd_bwd_bit_exact(case):
    from eval_driving_safety_amd import ops
    lead, hw, size = case
    rs = np.random.RandomState(sum(hw) + 5 * sum(size))
    x = rs.randn(*lead, *hw).astype(np.float32)
    g = rs.randn(*lead, *size).astype(np.float32)
    dev = torch.device("cuda", 0)
    t = torch.tensor(x, device=dev, requires_grad=True)
    up = ops.BilinearUp.apply(t, size)
    assert up.detach().cpu().numpy().tobytes() == O.bilinear_up(x, size).tobytes()
    up.backward(torch.tensor(g, device=dev))
    assert t.grad.cpu().numpy().tobytes() == O.bilinear_up_bwd(g, hw).tobytes()
    t2 = torch.tensor(x, device=dev, requires_grad=True)
    ref = F.interpolate(t2, size=size, mode="bilinear", align_corners=False)
    ref.backward(torch.tensor(g, device=dev))
    assert float((up.detach() - ref.detach()).abs().max()) <= 1e-6 * max(float(ref.detach().abs().max()), 1.0)
    assert float((t.grad - t2.grad).abs().max()) <= 1e-5 * max(float(t2.grad.abs().max()), 1.0)


@pytest.mark.gpu
def test_hip_bilinear_up_rejects_bad_arguments():
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    with pytest.raises(ValueError):
        ops.bilinear_up(torch.zeros((2, 3, 4), device=dev), (8, 8))
    with pytest.raises(ValueError):
        ops.bilinear_up(torch.zeros((1, 2, 3, 4), device=dev), (0, 8))
    with pytest.raises(TypeError):
        ops.bilinear_up(torch.zeros((1, 2, 3, 4)), (6, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("hw,size", [((75, 249), (150, 497)), ((38, 125), (75, 249)), ((19, 63), (38, 125))])
def test_hip_bilinear_up_pyramid_sizes_adjoint_and_reproducible(hw, size):
    """the three up-samplings of the 600 x 1987 pyramid (both eyes, 256 channels): the kernels against torch, the adjoint identity in
    float64, and the backward twice - the same bytes (torch's own backward of this operator is the one that is not)"""
    from eval_driving_safety_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device="cpu").manual_seed(hw[0])
    x = torch.randn((2, 256) + hw, generator=gen).to(dev)
    g = torch.randn((2, 256) + size, generator=gen).to(dev)
    up = ops.bilinear_up(x, size)
    ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    assert float((up - ref).abs().max()) <= 2e-6 * float(ref.abs().max())          # (same taps; the four products are summed in another association)
    gin = ops.bilinear_up_bwd(g, hw)
    assert torch.equal(gin, ops.bilinear_up_bwd(g, hw))
    lhs = float((up.double() * g.double()).sum())
    rhs = float((x.double() * gin.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * float((up.double() * g.double()).abs().sum())
    t = x.clone().requires_grad_(True)
    F.interpolate(t, size=size, mode="bilinear", align_corners=False).backward(g)
    assert float((gin - t.grad).abs().max()) <= 1e-5 * float(t.grad.abs().max())
