#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + two separate PMC passes of bench.py.
#   trace        : the HEADLINE leg alone (+ the all-float32 leg, other kernels) - the average of pgd_step_vec4_idx in its --stats summary is the
#                  duration behind roofline.frac of the line the same command prints (bench_lines.jsonl; tests/test_profiles.py holds the two together)
#   trace_other  : the Stereo R-CNN-shape leg and the delivered-iterates leg (the same kernel beside the copy engine's blits: its own file)
# usage: tools/gpu_profile.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
HEAD="--no-cpu-baseline --no-end-to-end --no-delivered --no-srcnn"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 $HEAD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_other -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-float-path > $OUT/trace_other.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 $HEAD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 $HEAD > $OUT/write.log 2>&1
grep -h '^{' $OUT/trace.log > $OUT/bench_line_of_the_trace.json
grep -h '^{' $OUT/trace.log $OUT/fetch.log $OUT/write.log > $OUT/bench_lines.jsonl
ls -R $OUT | head -40
