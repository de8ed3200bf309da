#!/bin/bash
# third part of the round's measurement set, after the FPN up-sampling moved to csrc/resize.hip: default bench line, R101 step profile, fuzz
# with the resize cases: -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r04_final4.json 2> gpurun_out/bench_r04_final4.err
bash tools/gpu_profile_step.sh r04r101 --r101 --pairs 1 --reps 1 > gpurun_out/final_profile3.log 2>&1
for s in 10 11 12; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r04_fuzz_c.log 2>&1
tail -3 gpurun_out/r04_fuzz_c.log; tail -c 600 gpurun_out/bench_r04_final4.json
