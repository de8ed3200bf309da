#!/usr/bin/env python3
"""F(4x4,3x3) kernel variants (tools/build_variant.sh ... wino4.hip -D...): the same layers on each library given on the command line
(one process per library: ADVENGINE_LIB is read at import).  usage: python tools/bench_wino4_variants.py [lib.so ...]"""
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
for b, cin, cout, h, w in ((2, 256, 256, 150, 497), (2, 128, 128, 96, 312), (2, 64, 64, 192, 624), (512, 256, 256, 14, 14)):
    x = torch.randn((b, cin, h, w), device=dev, generator=g)
    prep = ops.ConvWino4Prep(torch.randn((cout, cin, 3, 3), device=dev, generator=g) * 0.05)
    out["%%d->%%d @[%%d,%%d,%%d]" %% (cin, cout, b, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, None, None, True)) for _ in range(2)), 4)
    if os.environ.get("WINO4_EPILOGUES"):      # the same layer with the epilogue's loads: bias; bias + skip connection + mask
        bias, res = torch.randn((cout,), device=dev, generator=g), torch.randn((b, cout, h, w), device=dev, generator=g)
        out["%%d->%%d @[%%d,%%d,%%d] +bias" %% (cin, cout, b, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, bias, None, True)) for _ in range(2)), 4)
        out["%%d->%%d @[%%d,%%d,%%d] +bias+skip+mask" %% (cin, cout, b, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, bias, res, True, res)) for _ in range(2)), 4)
        del bias, res
for c, d, h, w in ((128, 96, 10, 152), (32, 48, 96, 312), (64, 48, 96, 312)):
    x = torch.randn((1, c, d, h, w), device=dev, generator=g)
    prep = ops.ConvWino4Prep(torch.randn((c, c, 3, 3, 3), device=dev, generator=g) * 0.05)
    out["3D %%d @[%%d,%%d,%%d]" %% (c, d, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, None, None, True)) for _ in range(2)), 4)
    if os.environ.get("WINO4_EPILOGUES"):
        bias, res = torch.randn((c,), device=dev, generator=g), torch.randn((1, c, d, h, w), device=dev, generator=g)
        out["3D %%d @[%%d,%%d,%%d] +bias" %% (c, d, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, bias, None, True)) for _ in range(2)), 4)
        out["3D %%d @[%%d,%%d,%%d] +bias+skip+mask" %% (c, d, h, w)] = round(min(timed(lambda: ops.conv_wino4(x, prep, bias, res, True, res)) for _ in range(2)), 4)
        del bias, res
print(json.dumps(out))
''' % (ROOT, ROOT)


def main():
    libs = sys.argv[1:] or [""]
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["ADVENGINE_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
        print(json.dumps({"lib": os.path.basename(lib) or "shipped", "ms": json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else "failed"}), flush=True)


if __name__ == "__main__":
    main()
