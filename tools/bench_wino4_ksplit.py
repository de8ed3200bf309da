"""K-split of the F(4x4,3x3) kernel on the small-map 3x3 layers (ResNet-101 stages 4 / 5 at 600 x 1987, two eyes): one workgroup per tile
leaves most compute units idle there.  Per layer: F(2x2,3x3) (the route table's choice so far), F(4x4,3x3) whole, and 2 .. 8 parts per tile.
One JSON line per layer -> profiles/rNN_wino4_ksplit.jsonl"""
import json
import sys

import torch

sys.path.insert(0, ".")
from eval_driving_safety_amd import ops  # noqa: E402

LAYERS = [(2, 256, 256, 38, 125), (2, 512, 512, 19, 63), (2, 256, 256, 75, 249), (2, 128, 128, 75, 249), (1, 256, 256, 38, 125), (2, 256, 256, 19, 63), (1, 64, 64, 96, 312)]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    dev = torch.device("cuda", 0)
    for (b, cin, cout, h, w) in LAYERS:
        x = torch.randn((b, cin, h, w), device=dev)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
        bias = torch.randn((cout,), device=dev)
        p2 = ops.Conv2dPrep(wt, 1, 1)
        p4 = ops.ConvWino4Prep(wt)
        row = {"layer": [b, cin, cout, h, w], "gflop_direct": round(2e-9 * b * cin * cout * 9 * h * w, 2)}
        row["wino_f2_ms"] = round(timeit(lambda: ops.conv2d(x, p2, bias, None, True, wino=True)), 4)
        row["wino4_ms"] = round(timeit(lambda: ops.conv_wino4(x, p4, bias, None, True)), 4)
        for s in (2, 3, 4, 6, 8):
            row["wino4_parts%d_ms" % s] = round(timeit(lambda: ops.conv_wino4(x, p4, bias, None, True, splits=s)), 4)
        row["rule_parts"] = int(ops._lib.load().adv_conv2d_wino4_ksplit_pick(b, cin, cout, h, w))
        row["wino4_rule_ms"] = round(timeit(lambda: ops.conv_wino4(x, p4, bias, None, True, splits=0)), 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
