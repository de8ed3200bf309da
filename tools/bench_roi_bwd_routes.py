#!/usr/bin/env python3
"""RoIAlign backward: round 5's route (per-roi axis tables + a wave per pixel) against round 4's (accumulators in LDS, ADV_ROI_BWD_LDS=1 in
the -DADV_TEST_HOOKS build) - on the detector's own 512 proposals (tools/data/r101_rois.npy, the case of profiles/r04_roi_bwd_phases.json)
and on synthetic roi sets from spread to identical.  One JSON line per (roi set, pooled size): both times, and that the bits agree."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route  # noqa: E402
from bench_roi_bwd_phases import timed  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    sets = [("R101 proposals (tools/data/r101_rois.npy)", torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "r101_rois.npy"))).to(dev))]
    for label, n, spread, wpx, hpx in (("uniform 8x8 px", 512, 1.0, 32, 32), ("uniform 4x13 px", 512, 1.0, 15, 51), ("clustered 4x13 px", 512, 0.08, 15, 51),
                                       ("clustered 8x8 px", 512, 0.08, 32, 32), ("uniform 30x30 px", 512, 1.0, 120, 120), ("identical 4x13 px", 512, 0.0, 15, 51)):
        x1 = 900 + (rs.rand(n) - 0.5) * 1900 * spread
        y1 = 300 + (rs.rand(n) - 0.5) * 500 * spread
        sets.append((label, torch.tensor(np.stack([np.zeros(n), x1, y1, x1 + wpx, y1 + hpx], 1).astype(np.float32), device=dev)))
    for label, rois in sets:
        for pooled in (7, 14):
            g = torch.randn((rois.shape[0], 256, pooled, pooled), device=dev)
            call = lambda: ops.roi_align_bwd(g, rois, (1, 256, 150, 497), 0.25, 0)      # noqa: E731
            new = call()
            ms = timed(call)
            with hooks_route(ADV_ROI_BWD_LDS="1"):
                old = call()
                old_ms = timed(call)
            print(json.dumps({"rois": label, "n": int(rois.shape[0]), "map": [150, 497], "channels": 256, "pooled": pooled, "ms": round(ms, 4),
                              "r04_lds_route_ms": round(old_ms, 4), "same_bits": bool(torch.equal(new, old))}), flush=True)


if __name__ == "__main__":
    main()
