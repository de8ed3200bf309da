#!/bin/bash
# Kernel trace of `dsgn_pgd_attack --model upstream` on the stand-in DSGN checkout (tests/fake_upstream/dsgn_checkout: upstream module names; its layers
# import the compiled extension `dsgn._C`; F.grid_sample and the trilinear-softmax depth regression inline in forward): which kernels the reference's
# own call sites (attack/DSGN/pgd_attack.py:220,308,324) reach once upstream_shims.install("dsgn") and adopt() have run.
# usage (GPU box): tools/profile_upstream_dsgn_standin.sh <tag>  -> gpurun_out/prof_<tag>/kernel_stats.csv
set -u
TAG=${1:-r05upstream_dsgn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
WORK=$OUT/work
mkdir -p $WORK
export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests:$R/tests/golden:$R/tests/fake_upstream/dsgn_checkout
cd $WORK
python3 - <<PY
import os, sys, torch
import _upstream
_upstream.bind_dsgn_extension("reference")          # only to construct the module and save "the checkpoint"; the CLI run below installs the shim
sys.path.insert(0, os.path.join("$R", "tests"))
import test_cli_upstream as T
T.make_kitti_folder("data/kitti/training", ["000003", "000011"])
T.make_dsgn_checkpoint("outputs/temp/DSGN_car_pretrained/finetune_53.tar")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m eval_driving_safety_amd.cli.dsgn_pgd_attack \
  --data_path data/kitti/training --split_file data/kitti/training/val.txt --loadmodel outputs/temp/DSGN_car_pretrained/finetune_53.tar \
  -btest 1 -d 0 --debug --debugnum 1 --iter 3 --eps 0.03 > $OUT/run.log 2>&1
cp $(find $OUT/trace -name '*_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $WORK
grep -E "adopted|upstream_shims|attacked|Error|error" $OUT/run.log
grep -E "psv_|grid_sample3d|depth_regress|focal|conv_wino|conv3d|conv2d|nms|pgd_step|plan_" $OUT/kernel_stats.csv | cut -c1-90,200-260
