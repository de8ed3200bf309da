#!/usr/bin/env python3
"""Where a stage of the strided 3x3x3 kernel (csrc/conv3d.hip conv3d_k3_s2_mfma) spends its time, from in-kernel s_memtime stamps
(a -DADV_S2_STAMPS build: tools/build_variant.sh s2stamps conv3d.hip -DADV_S2_STAMPS; run with ADVENGINE_LIB=tools/_build/libadv_s2stamps.so):
per wave the median cycles of a stage for issuing the next stage's LDS-DMA requests, for the matrix-instruction block, and for the wait +
barrier behind it (__syncthreads(): the requests must have landed)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops, _lib  # noqa: E402


def run(name, fn, stages):
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((8, 4, 64, 4), dtype=np.uint64)
    assert lib.adv_debug_s2_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    s = buf.astype(np.int64)[:, :, 1:stages - 1, :]
    ok = s[..., 0] > 0
    out = {"case": name, "stage_cycles_median": int(np.median((s[..., 3] - s[..., 0])[ok])), "issue_requests": int(np.median((s[..., 1] - s[..., 0])[ok])),
           "matrix_block": int(np.median((s[..., 2] - s[..., 1])[ok])), "wait_and_barrier": int(np.median((s[..., 3] - s[..., 2])[ok])),
           "wait_and_barrier_p90": int(np.percentile((s[..., 3] - s[..., 2])[ok], 90))}
    print(json.dumps(out), flush=True)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev).manual_seed(0)
    for name, cin, cout, d, h, w in (("hg1 32->64 [48,96,312]", 32, 64, 48, 96, 312), ("gh1 64->128 [192,20,304]", 64, 128, 192, 20, 304),
                                     ("hg3 64->64 [24,48,156]", 64, 64, 24, 48, 156), ("gh3 128->128 [96,10,152]", 128, 128, 96, 10, 152)):
        x = torch.randn((1, cin, d, h, w), device=dev, generator=g)
        wt = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=g) * 0.03
        prep = ops.conv3d_k3_prep(wt)
        bias = torch.randn((cout,), device=dev, generator=g)
        run(name, lambda: ops.conv3d_k3_s2(x, prep, cout, True, bias), cin // 2)


if __name__ == "__main__":
    main()
