#!/usr/bin/env python3
"""Which torch operators make up the element-wise share of the ResNet-101-FPN-shaped (default) or DSGN-shaped (--dsgn) detector step: one eager forward + backward under
torch.profiler (shapes recorded), the device kernels that are not libadvengine's / MIOpen's / rocBLAS's grouped by (operator, input shapes)
- launches and device time per step.  One JSON object."""
import json
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import adapters, data, surrogates  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if "--dsgn" in sys.argv:                 # the DSGN-shaped layer-list graph instead
        net = adapters.DsgnShapedAdapter(dev, seed=0)
        batch = next(iter(data.SyntheticStereo(1, "dsgn", batch=1, seed=0)))
        batch.extra = net.synthetic_extra(batch, seed=1)
    else:
        surrogates.FoldedConv.impl = "auto"
        model = surrogates.StereoRcnnR101(seed=0, rois_per_image=512).to(dev).eval()
        net = adapters.StereoRcnnAdapter(model, torch.zeros(6, device=dev))
        batch = next(iter(data.SyntheticStereo(1, "srcnn", batch=1, seed=0)))
        batch.extra = surrogates.synthetic_srcnn_extra(batch, dev)
    x = torch.cat([batch.imgL, batch.imgR]).to(dev)
    for _ in range(3):
        net.loss_and_grad(x, batch.extra)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        net.loss_and_grad(x, batch.extra)
        torch.cuda.synchronize()
    rows = defaultdict(lambda: [0, 0.0])
    total = 0.0
    for ev in prof.key_averages(group_by_input_shape=True):
        dt = float(getattr(ev, "self_device_time_total", 0.0) or getattr(ev, "self_cuda_time_total", 0.0))
        if dt <= 0:
            continue
        total += dt
        name = ev.key
        if any(s in name for s in ("Conv2d", "conv2d", "Conv3d", "conv3d", "Wino", "wino", "RoIAlign", "advengine", "miopen", "mm", "linear", "convolution", "mfma")):
            continue
        key = "%s %s" % (name, str(ev.input_shapes)[:150])
        rows[key][0] += ev.count
        rows[key][1] += dt
    top = sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("ADV_TRACE_TOP", "60"))]
    print(json.dumps({"device_us_total": total, "small_op_us": sum(v[1] for v in rows.values()), "small_op_calls": sum(v[0] for v in rows.values()),
                      "top": [{"op": k, "calls": v[0], "device_us": round(v[1], 1)} for k, v in top]}, indent=1))


if __name__ == "__main__":
    main()
