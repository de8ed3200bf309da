"""Property tests (hypothesis) of the projection invariants on the oracle (CPU) and of HIP-vs-oracle parity
on random shapes / scalars (GPU).  SURVEY 4, item 5."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import oracle_np as O

shapes = st.tuples(st.integers(1, 3), st.just(3), st.integers(1, 9), st.integers(1, 13))
eps_s = st.floats(0.0, 0.5, allow_nan=False, width=32)
alpha_s = st.floats(0.0, 0.5, allow_nan=False, width=32)


def _draw(seed, shape):
    rs = np.random.RandomState(seed)
    clean = rs.rand(*shape).astype(np.float32)
    x = O.normalize((clean + (rs.rand(*shape).astype(np.float32) - 0.5) * 0.2).astype(np.float32))
    g = rs.randn(*shape).astype(np.float32)
    g[rs.rand(*shape) < 0.1] = 0
    return x, g, clean


@settings(max_examples=60, deadline=None)
@given(shapes, st.integers(0, 10 ** 6), alpha_s, eps_s)
def test_oracle_projection_invariants(shape, seed, alpha, eps):
    x, g, clean = _draw(seed, shape)
    y = O.denormalize(O.pgd_step_norm01(x, g, clean, alpha, eps))
    slack = 3e-7                                   # the normalise / denormalise round trip
    assert np.all(np.abs(y - clean) <= np.float32(eps) + slack)
    assert y.min() >= -slack and y.max() <= 1 + slack
    # zero gradient and alpha: the step is the projection of x itself
    y0 = O.pgd_step_norm01(x, np.zeros_like(g), clean, alpha, eps)
    y1 = O.pgd_step_norm01(x, g, clean, 0.0, eps)
    assert y0.tobytes() == y1.tobytes()


@settings(max_examples=40, deadline=None)
@given(shapes, st.integers(0, 10 ** 6), st.floats(0.0, 4.0, width=32), st.floats(0.0, 30.0, width=32))
def test_oracle_srcnn_projection_invariants(shape, seed, alpha, eps255):
    rs = np.random.RandomState(seed)
    clean = (rs.rand(*shape) * 255).astype(np.float32)
    for c in range(3):
        clean[:, c] -= np.float32(O.SRCNN_PIXEL_MEANS[c])
    x = clean + (rs.randn(*shape) * 5).astype(np.float32)
    g = rs.randn(*shape).astype(np.float32)
    y = O.pgd_step_meansub255(x, g, clean, alpha, eps255)
    assert np.all(np.abs(y - clean) <= np.float32(eps255) * (1 + 1e-6) + 1e-4)
    for c in range(3):
        assert y[:, c].min() >= O.SRCNN_LO[c] and y[:, c].max() <= O.SRCNN_HI[c]


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.tuples(st.integers(1, 4), st.just(3), st.integers(1, 40), st.integers(1, 70)), st.integers(0, 10 ** 6), alpha_s, eps_s,
       st.booleans())
def test_hip_matches_oracle_on_random_problems(shape, seed, alpha, eps, srcnn):
    import torch
    from eval_driving_safety_amd import ops
    x, g, clean = _draw(seed, shape)
    dev = torch.device("cuda", 0)
    if srcnn:
        x, clean = (x * 40).astype(np.float32), (clean * 200 - 100).astype(np.float32)
        want = O.pgd_step_meansub255(x, g, clean, alpha * 4, eps * 40)
        got = ops.pgd_step(torch.tensor(x, device=dev), torch.tensor(g, device=dev), torch.tensor(clean, device=dev), ops.Space.srcnn(),
                           alpha * 4, eps * 40)
    else:
        want = O.pgd_step_norm01(x, g, clean, alpha, eps)
        got = ops.pgd_step(torch.tensor(x, device=dev), torch.tensor(g, device=dev), torch.tensor(clean, device=dev), ops.Space.dsgn(),
                           alpha, eps)
    assert got.cpu().numpy().tobytes() == want.tobytes()
