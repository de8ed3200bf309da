#!/bin/bash
# Where the F(4x4,3x3) kernel's LDS bank conflicts come from: the LDS counters of one layer with parts of the kernel switched off
# (ADV_WINO4_DBG ablation bits of the -DADV_TEST_HOOKS build: 1 transform, 8 commits of the input tile, 32 operand reads, 64 weight DMA ...).
# usage: tools/gpu_pmc_wino4_lds.sh <tag> "<case substring>" <abl> [<abl> ...]   -> gpurun_out/prof_<tag>/abl_<n>/
set -u
TAG=$1; CASE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export ADVENGINE_LIB=$R/eval_driving_safety_amd/libadvengine_hooks.so
for A in "$@"; do
  export ADV_WINO4_DBG=$A
  [ "$A" = "0" ] && unset ADV_WINO4_DBG
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $OUT/abl_$A -- python3 $R/tools/pmc_layers.py --manifest $OUT/manifest_$A.json --only "$CASE" > $OUT/abl_$A.log 2>&1
  python3 - "$OUT/abl_$A" "$A" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for p in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "conv_wino4<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
d = {k: acc[k] / max(n[k], 1) for k in acc}
print("abl", sys.argv[2], {k: round(v) for k, v in d.items()}, "conflict/active %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done
