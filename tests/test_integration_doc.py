"""INTEGRATION.md's ctypes stub is executable documentation: the first python block is run as written (only
the library path is made absolute) and its ``pgd_step_`` must reproduce the oracle."""
import os
import re

import numpy as np
import pytest

import synth
from oracle import oracle_np as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(.*?)```", text, flags=re.S).group(1)
    return block.replace('ctypes.CDLL("libadvengine.so")',
                         'ctypes.CDLL(%r)' % os.path.join(ROOT, "eval_driving_safety_amd", "libadvengine.so"))


def test_stub_binds_without_a_gpu():
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md", "exec"), ns)
    assert ns["DSGN"].kind == 0 and ns["SRCNN"].kind == 1
    assert [np.float32(v) for v in ns["DSGN"].scale] == [np.float32(v) for v in O.DSGN_STD]


@pytest.mark.gpu
def test_stub_runs_a_step_on_the_gpu():
    import torch
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md", "exec"), ns)
    x_np = np.concatenate([synth.dsgn_normalised(5, 40, 64), synth.dsgn_normalised(6, 40, 64)])
    g_np = synth.gradient(7, x_np.shape)
    clean_np = O.denormalize(x_np)
    x = torch.from_numpy(x_np.copy()).cuda()
    ns["pgd_step_"](x, torch.from_numpy(g_np).cuda(), torch.from_numpy(clean_np).cuda(), ns["DSGN"], 1 / 255, 0.03)
    torch.cuda.synchronize()
    assert x.cpu().numpy().tobytes() == O.pgd_step_norm01(x_np, g_np, clean_np, 1 / 255, 0.03).tobytes()


@pytest.mark.gpu
def test_clean_index_stub_runs_on_the_gpu():
    """the second python block (the clean image as one byte per element), executed as written after the first"""
    import torch
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert "adv_clean_index_build_f32" in blocks[1]
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md", "exec"), ns)
    H, W, h_img, w_img = 24, 40, 21, 37
    x_np = np.concatenate([synth.dsgn_padded(5, h_img, w_img, H, W), synth.dsgn_padded(6, h_img, w_img, H, W)])
    g_np = synth.gradient(7, x_np.shape)
    x = torch.from_numpy(x_np.copy()).cuda()
    ns.update(x=x, g=torch.from_numpy(g_np).cuda(), clean=torch.empty_like(x), n=2, H=H, W=W, h_img=h_img, w_img=w_img,
              alpha=1 / 255, eps=0.03)
    exec(compile(blocks[1], "INTEGRATION.md#2", "exec"), ns)
    torch.cuda.synchronize()
    assert ns["ok"].cpu().tolist() == [1, 1]
    assert x.cpu().numpy().tobytes() == O.pgd_step_norm01(x_np, g_np, O.denormalize(x_np), 1 / 255, 0.03).tobytes()
