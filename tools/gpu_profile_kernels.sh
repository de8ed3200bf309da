#!/bin/bash
# rocprofv3 of the secondary kernels (tools/bench_kernels.py): kernel trace + separate FETCH/WRITE PMC passes.
set -u
TAG=${1:-r01k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_kernels.py > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/bench_kernels.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/bench_kernels.py > $OUT/write.log 2>&1
ls -R $OUT | head -30
