#!/bin/bash
# Kernel trace of `--model upstream` on the stand-in Stereo R-CNN checkout (tests/fake_upstream/srcnn_checkout: upstream module names, imports
# ROIAlign / nms from model.roi_layers): which kernels the reference's own call sites reach once the shims are installed and the network is
# adopted.  usage (GPU box): tools/profile_upstream_standin.sh <tag>  -> gpurun_out/prof_<tag>/kernel_stats.csv
set -u
TAG=${1:-r04upstream}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
WORK=$OUT/work
mkdir -p $WORK
export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests/fake_upstream/srcnn_checkout
cd $WORK
python3 - <<PY
import os, sys, torch
from model.stereo_rcnn.resnet import resnet
net = resnet(("__background__", "Car"), 101, pretrained=False)
net.create_architecture()
os.makedirs("models_stereo", exist_ok=True)
torch.save({"model": net.state_dict(), "uncert": torch.tensor([0.1, -0.2, 0.3, 0.0, 0.5, -0.4])}, "models_stereo/stereo_rcnn_12_6477.pth")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m eval_driving_safety_amd.cli.srcnn_pgd_attack --iter 2 --eps 0.03 --debug --debugnum 2 > $OUT/run.log 2>&1
cp $(find $OUT/trace -name '*_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $WORK
grep -E "adopted|roi_layers|attacked" $OUT/run.log
grep -E "conv_wino|conv2d_1x1_mfma|conv2d_3x3_mfma|roi_align|nms_|pgd_step|bias_act" $OUT/kernel_stats.csv | cut -c1-80,200-260
