#!/usr/bin/env python3
"""The R101 graph's 1x1 layer shapes (bias + ReLU, forward) on whichever library ADVENGINE_LIB names - one JSON line.  Used for the
residency A/B of the 1x1 kernel (hooks build: ADVENGINE_LIB=eval_driving_safety_amd/libadvengine_hooks.so ADV_C2_OCC=4..8 caps the
workgroups per compute unit by LDS padding) and for its compile-time phase ablations (tools/build_variant.sh c2X conv2d.hip
-DADV_C2_NOFETCH / _NOBARRIER / _NOCOMMIT / _NOSTORE: wrong results, timing only) - profiles/r06_c2_stamps.jsonl."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from eval_driving_safety_amd import ops
from bench_wino_tiles import timed
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for b, cin, cout, h, w in ((2, 256, 1024, 38, 125), (2, 1024, 256, 38, 125), (2, 64, 256, 150, 497), (2, 256, 64, 150, 497), (2, 512, 128, 75, 249), (2, 128, 512, 75, 249), (2, 2048, 512, 19, 63), (2, 512, 2048, 19, 63), (512, 256, 1024, 14, 14)):
    x = torch.randn((b, cin, h, w), device=dev, generator=g)
    prep = ops.Conv2dPrep(torch.randn((cout, cin, 1, 1), device=dev, generator=g) * 0.05, 1, 0)
    bias = torch.randn((cout,), device=dev, generator=g)
    out["%d->%d@[%d,%d,%d]" % (cin, cout, b, h, w)] = round(min(timed(lambda: ops.conv2d(x, prep, bias, None, True)) for _ in range(2)), 4)
print(json.dumps({"occ": os.environ.get("ADV_C2_OCC", "8 (default)"), "ms": out}))
