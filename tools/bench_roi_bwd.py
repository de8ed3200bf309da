#!/usr/bin/env python3
"""RoIAlign backward (deterministic gather) on the rois the ResNet-101-FPN Stereo R-CNN-shaped detector actually produces at 600x1987:
per pyramid level and pooled size - how many rois, their size at the level's scale, time per call.  One JSON line per (level, pooled)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import data, ops, surrogates  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import hooks_route  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    model = surrogates.StereoRcnnR101(seed=0, rois_per_image=512).to(dev).eval()
    batch = next(iter(data.SyntheticStereo(1, "srcnn", batch=1, seed=0)))
    extra = surrogates.synthetic_srcnn_extra(batch, dev)
    with torch.no_grad():
        out = model(batch.imgL.to(dev), batch.imgR.to(dev), extra.im_info, extra.gt_boxes_left, extra.gt_boxes_right, extra.gt_boxes_merge,
                    extra.gt_dim_orien, extra.gt_kpts, extra.num_boxes)
    rois = out[0].reshape(-1, 5).contiguous()
    h = rois[:, 4] - rois[:, 2] + 1
    w = rois[:, 3] - rois[:, 1] + 1
    level = torch.round(torch.log(torch.sqrt(h * w) / 224.0) + 4).clamp(2, 5)
    for i, l in enumerate((2, 3, 4, 5)):
        idx = torch.nonzero(level == l).view(-1)
        if idx.numel() == 0:
            continue
        stride = 4 * 2 ** i
        fh, fw = (600 + stride - 1) // stride, (1987 + stride - 1) // stride
        r = rois[idx].contiguous()
        for pooled in (7, 14):
            g = torch.randn((r.shape[0], 256, pooled, pooled), device=dev)
            def timed():
                for _ in range(2):
                    ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 5
            ms = timed()
            with hooks_route(ADV_ROI_BWD_LDS="1"):      # round 4's route: accumulators in LDS, a tile's rois one after the other
                same = torch.equal(ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0), ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0))
                lds_ms = timed()
                older = ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0)
            same = same and torch.equal(older, ops.roi_align_bwd(g, r, (1, 256, fh, fw), 1.0 / stride, 0))
            print(json.dumps({"level": "P%d" % l, "map": [fh, fw], "pooled": pooled, "rois": int(r.shape[0]),
                              "roi_w_at_level_px_min_mean_max": [round(float(v), 2) for v in ((w[idx] / stride).min(), (w[idx] / stride).mean(), (w[idx] / stride).max())],
                              "roi_h_at_level_px_min_mean_max": [round(float(v), 2) for v in ((h[idx] / stride).min(), (h[idx] / stride).mean(), (h[idx] / stride).max())],
                              "ms_per_call": round(ms, 4), "r04_lds_route_ms": round(lds_ms, 4), "same_bits": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
