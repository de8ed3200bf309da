"""A short, seeded run of tools/fuzz_gpu.py inside the GPU suite: random shapes through the HIP kernels (convolution under a random
alternative code path, grid_sample3d forward / backward, PGD steps with the 8-bit index in both pixel spaces) against the oracle,
bit for bit; round 3: the 2D convolutions (1x1 / 3x3, every tile shape, random epilogue, forward and backward).  The long runs are recorded in
profiles/r02_fuzz.log and profiles/r03_fuzz.log."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_the_oracle(seed):
    torch = pytest.importorskip("torch")
    pytest.importorskip("oracle.oracle_c", reason="make -C oracle first (build() does it)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(seed)
    kinds = [fuzz_gpu.conv_case] * 5 + [fuzz_gpu.conv2d_case] * 4 + [fuzz_gpu.wino3d_case] * 3 + [fuzz_gpu.grid_case] * 2 + [fuzz_gpu.pgd_case] * 3 + [fuzz_gpu.roi_case] * 3 + \
        [fuzz_gpu.depth_case] * 2
    for _ in range(60):
        kinds[int(rs.randint(len(kinds)))](rs, dev)
    fuzz_gpu.conv_case(rs, dev, big=True)
    torch.cuda.synchronize()
