#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + two separate PMC passes of bench.py.
# usage: tools/gpu_profile.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $OUT/write.log 2>&1
grep -h '^{' $OUT/trace.log $OUT/fetch.log $OUT/write.log > $OUT/bench_lines.jsonl
ls -R $OUT | head -40
