#!/usr/bin/env python3
"""Tile-shape sweep of the strided (conv3d_k3_s2_mfma) and transposed (convt3d_k3_s2_mfma) hourglass kernels on the DSGN-shaped layer
shapes, through the -DADV_TEST_HOOKS build: ADV_CONV_S2_WD (waves over 1 / 2 / 4 output planes) x ADV_CONV_S2_PD (planes per wave), and
ADV_CONV_T_TD (1 x 4, 2 x 2, 4 x 1 input planes x rows).  One JSON line per layer: ms per shape, the shipped rule's pick ("auto").  Same bits."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import _lib, ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with _lib.using(_lib.HOOKS_LIB_PATH):
            return timed(fn)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for name, c, k, d, h, w in (("hg1", 32, 64, 48, 96, 312), ("gh1", 64, 128, 192, 20, 304), ("hg3", 64, 64, 24, 48, 156), ("gh3", 128, 128, 96, 10, 152)):
        x = torch.randn((1, c, d, h, w), device=dev, generator=g)
        wt = torch.randn((k, c, 3, 3, 3), device=dev, generator=g) * 0.03
        bias = torch.randn((k,), device=dev, generator=g)
        prep = ops.conv3d_k3_s2_prep(wt)
        fn = lambda: ops.conv3d_k3_s2(x, prep, k, relu=True, bias=bias)      # noqa: E731
        ms = {"auto": round(timed(fn), 4)}
        for wd in ("1", "2", "4"):
            for pd in ("1", "2"):
                if wd == "4" and pd == "2":
                    continue
                ms["wd%s_pd%s" % (wd, pd)] = round(with_env({"ADV_CONV_S2_WD": wd, "ADV_CONV_S2_PD": pd}, fn), 4)
        flops = 2.0 * k * c * 27 * ((d + 1) // 2) * ((h + 1) // 2) * ((w + 1) // 2)
        best = min((v, n) for n, v in ms.items() if n != "auto")
        print(json.dumps({"layer": "s2 %s %d->%d on [%d,%d,%d]" % (name, c, k, d, h, w), "ms": ms, "best": best[1], "auto_over_best": round(ms["auto"] / best[0], 3),
                          "auto_frac_of_157TF": round(flops / ms["auto"] / 1e9 / 157.3, 3)}), flush=True)
    for name, c, k, d, h, w in (("hg6", 64, 32, 24, 48, 156), ("gh6", 128, 64, 96, 10, 152), ("gh5", 128, 128, 48, 5, 76), ("hg1 bwd", 64, 32, 24, 48, 156), ("gh1 bwd", 128, 64, 96, 10, 152)):
        x = torch.randn((1, c, d, h, w), device=dev, generator=g)
        wt = torch.randn((c, k, 3, 3, 3), device=dev, generator=g) * 0.03
        bias = torch.randn((k,), device=dev, generator=g)
        cls = ops.conv_transpose3d_k3_s2_prep(wt)
        plain = name.endswith("bwd")
        fn = (lambda: ops.conv_transpose3d_k3_s2(x, cls, k)) if plain else (lambda: ops.conv_transpose3d_k3_s2(x, cls, k, relu=True, bias=bias))      # noqa: E731
        ms = {"auto": round(timed(fn), 4)}
        for td in ("1", "2", "4"):
            ms["td%s" % td] = round(with_env({"ADV_CONV_T_TD": td}, fn), 4)
        flops = 2.0 * k * c * 27 * d * h * w
        best = min((v, n) for n, v in ms.items() if n != "auto")
        print(json.dumps({"layer": "t2 %s %d->%d on [%d,%d,%d]" % (name, c, k, d, h, w), "ms": ms, "best": best[1], "auto_over_best": round(ms["auto"] / best[0], 3),
                          "auto_frac_of_157TF": round(flops / ms["auto"] / 1e9 / 157.3, 3)}), flush=True)


if __name__ == "__main__":
    main()
