#!/bin/bash
# Kernel statistics of the STEADY-STATE steps of an end-to-end leg: two rocprofv3 --kernel-trace --stats runs of the same command - the
# warm-up alone (ADV_STOP_AFTER_WARMUP=1: kernel loads, MIOpen's one-time solver search with its naive reference kernels) and the whole
# leg - whose per-kernel difference tools/summarize_step_profile.py reports.   usage: tools/gpu_profile_step.sh <tag> --full|--r101 [args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export ADV_STOP_AFTER_WARMUP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/warm -- python3 $R/tools/bench_end_to_end.py "$@" > $OUT/warm.log 2>&1
export ADV_STOP_AFTER_WARMUP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 $R/tools/bench_end_to_end.py "$@" > $OUT/full.log 2>&1
cp $(find $OUT/warm -name '*_kernel_stats.csv' | head -1) $OUT/warmup_kernel_stats.csv
cp $(find $OUT/full -name '*_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/warm $OUT/full
python3 $R/tools/summarize_step_profile.py $OUT/kernel_stats.csv $OUT/summary.json $OUT/warmup_kernel_stats.csv
