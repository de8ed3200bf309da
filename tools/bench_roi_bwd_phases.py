#!/usr/bin/env python3
"""Where the RoIAlign backward's time goes on the proposals the ResNet-101-FPN-shaped detector produces (tools/data/r101_rois.npy: the 512
rois of surrogates.StereoRcnnR101(seed 0) on the synthetic 600x1987 pair - 301 distinct, narrow, 127 of the 304 P2 tiles touched, the
longest tile list 80): the shipped kernel, and the -DADV_TEST_HOOKS build with parts of it switched off (ADV_ROI_DBG bit 1: no work
items, bit 2: no classification, bit 4: no sample loops; ADV_ROI_BWD_SCALAR_ITEMS: round 3's one-channel items).  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import _lib, ops  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rois = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "r101_rois.npy"))).to(dev)
    out = {}
    for pooled in (7, 14):
        g = torch.randn((rois.shape[0], 256, pooled, pooled), device=dev)
        call = lambda: ops.roi_align_bwd(g, rois, (1, 256, 150, 497), 0.25, 0)      # noqa: E731
        row = {"shipped_ms": timed(call)}
        for name, env in (("no_items", {"ADV_ROI_DBG": "1"}), ("no_classification", {"ADV_ROI_DBG": "2"}), ("no_sample_loops", {"ADV_ROI_DBG": "4"}),
                          ("no_items_no_classification", {"ADV_ROI_DBG": "3"}), ("scalar_items", {"ADV_ROI_BWD_SCALAR_ITEMS": "1"}),
                          ("no_stage", {"ADV_ROI_BWD_NO_STAGE": "1"}), ("acc_chan_32", {"ADV_ROI_ACC_CHAN": "32"}), ("acc_chan_8", {"ADV_ROI_ACC_CHAN": "8"}),
                          ("register_gather", {"ADV_ROI_BWD_REGS": "1"}), ("register_gather_8ch", {"ADV_ROI_BWD_REGS": "1", "ADV_ROI_BWD_CB8": "1"}),
                          ("register_gather_no_samples", {"ADV_ROI_BWD_REGS": "1", "ADV_ROI_DBG": "4"})):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                with _lib.using(_lib.HOOKS_LIB_PATH):
                    row[name + "_ms"] = timed(call)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        out["pooled_%d" % pooled] = {k: round(v, 4) for k, v in row.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
