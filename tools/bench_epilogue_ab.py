#!/usr/bin/env python3
"""A/B of epilogue changes in the 2D kernels (tools/build_variant.sh <name> conv2d.hip / wino2d.hip): the ResNet-101 graph's 1x1 and 3x3 layer
shapes with bias + ReLU (forward) and with skip connection + mask (backward w.r.t. the input), one process per library.
usage: python tools/bench_epilogue_ab.py [lib.so ...]   ("" = the shipped library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tools"))
from eval_driving_safety_amd import ops
from bench_wino_tiles import timed
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for k, b, cin, cout, h, w in ((1, 2, 256, 1024, 38, 125), (1, 2, 1024, 256, 38, 125), (1, 2, 64, 256, 150, 497), (1, 2, 256, 64, 150, 497), (1, 2, 512, 128, 75, 249),
                              (1, 2, 128, 512, 75, 249), (1, 2, 2048, 512, 19, 63), (1, 512, 256, 1024, 14, 14), (3, 2, 256, 256, 38, 125), (3, 2, 512, 512, 19, 63), (3, 2, 128, 128, 75, 249)):
    x = torch.randn((b, cin, h, w), device=dev, generator=g)
    prep = ops.Conv2dPrep(torch.randn((cout, cin, k, k), device=dev, generator=g) * 0.05, 1, k // 2)
    bias, res = torch.randn((cout,), device=dev, generator=g), torch.randn((b, cout, h, w), device=dev, generator=g)
    gy, xin = torch.randn((b, cout, h, w), device=dev, generator=g), torch.randn((b, cin, h, w), device=dev, generator=g)
    name = "%%dx%%d %%d->%%d @[%%d,%%d,%%d]" %% (k, k, cin, cout, b, h, w)
    wino = k == 3
    out[name + " fwd bias relu"] = round(min(timed(lambda: ops.conv2d(x, prep, bias, None, True, wino=wino)) for _ in range(2)), 4)
    out[name + " fwd bias skip relu"] = round(min(timed(lambda: ops.conv2d(x, prep, bias, res, True, wino=wino)) for _ in range(2)), 4)
    out[name + " dgrad skip mask"] = round(min(timed(lambda: ops.conv2d_dgrad(gy, prep, residual=xin, mask=xin, wino=wino)) for _ in range(2)), 4)
    del x, prep, bias, res, gy, xin
print(json.dumps(out))
''' % (ROOT, ROOT)


def main():
    for lib in sys.argv[1:] or [""]:
        env = dict(os.environ)
        if lib:
            env["ADVENGINE_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        ok = r.returncode == 0 and r.stdout.strip()
        print(json.dumps({"lib": os.path.basename(lib) or "shipped", "ms": json.loads(r.stdout.strip().splitlines()[-1]) if ok else "failed: " + r.stderr[-300:]}), flush=True)


if __name__ == "__main__":
    main()
