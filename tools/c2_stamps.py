#!/usr/bin/env python3
"""Where a workgroup of the 1x1 kernel (csrc/conv2d.hip conv2d_1x1_mfma, the 64 x 64 tile) spends its life, from in-kernel s_memtime stamps
(a -DADV_C2_STAMPS build: tools/build_variant.sh c2stamps conv2d.hip -DADV_C2_STAMPS; run with ADVENGINE_LIB=tools/_build/libadv_c2stamps.so)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops, _lib  # noqa: E402


def run(name, fn, nstage):
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((16, 4, 64, 5), dtype=np.uint64)
    assert lib.adv_debug_c2_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    sall = buf.astype(np.int64)
    for part, s in (("first eight workgroups (full machine)", sall[:8]), ("last eight workgroups (the tail)", sall[8:])):
        report(name + " - " + part, s, nstage)


def report(name, s, nstage):
    life = s[:, :, 62, :]
    ok = life[..., 0] > 0
    med = lambda a: int(np.median(a[ok]))
    out = {"case": name, "workgroup_life_cycles_median": {"entry_to_setup": med(life[..., 1] - life[..., 0]), "first_fetches_commit_barrier": med(life[..., 2] - life[..., 1]),
                                                         "stage_loop": med(life[..., 3] - life[..., 2]), "epilogue_issue": med(s[:, :, 63, 0] - life[..., 3]),
                                                         "store_drain": med(life[..., 4] - s[:, :, 63, 0]), "whole": med(life[..., 4] - life[..., 0])}}
    st = s[:, :, 1:min(nstage, 62) - 2, :]
    oks = st[..., 0] > 0
    m2 = lambda a: int(np.median(a[oks]))
    out["stage_cycles_median"] = {"whole": m2(st[..., 4] - st[..., 0]), "fetch_issue": m2(st[..., 1] - st[..., 0]), "products": m2(st[..., 2] - st[..., 1]),
                                  "commit": m2(st[..., 3] - st[..., 2]), "barrier": m2(st[..., 4] - st[..., 3]), "barrier_p90": int(np.percentile((st[..., 4] - st[..., 3])[oks], 90))}
    w0 = s[0, 0, :min(nstage, 62), :]              # one wave's stages in order: whole, products, barrier
    out["workgroup0_wave0_stages"] = [[int(r[4] - r[0]), int(r[2] - r[1]), int(r[4] - r[3])] for r in w0 if r[0] > 0][:24]
    # start skew between the stamped workgroups (are they resident together?)
    print(json.dumps(out), flush=True)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev).manual_seed(0)
    for b, cin, cout, h, w in ((2, 256, 1024, 38, 125), (2, 1024, 256, 38, 125), (2, 64, 256, 150, 497), (2, 512, 128, 75, 249)):
        x = torch.randn((b, cin, h, w), device=dev, generator=g)
        prep = ops.Conv2dPrep(torch.randn((cout, cin, 1, 1), device=dev, generator=g) * 0.05, 1, 0)
        bias = torch.randn((cout,), device=dev, generator=g)
        run("1x1 %d->%d on [%d,%d,%d,%d]" % (cin, cout, b, cin, h, w), lambda: ops.conv2d(x, prep, bias, None, True), (cin + 15) // 16)


if __name__ == "__main__":
    main()
