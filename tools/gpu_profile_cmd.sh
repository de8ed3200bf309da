#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of one python command of this repo.
# usage: tools/gpu_profile_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,run.log}
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$R/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1
STATS=$(find $OUT/trace -name '*_kernel_stats.csv' | head -1)
[ -n "$STATS" ] && cp $STATS $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -25 $OUT/kernel_stats.csv | cut -c1-180
