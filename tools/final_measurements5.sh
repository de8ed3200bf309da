#!/bin/bash
# the round's closing measurement set (final build: resize kernels, deep staging, static Stereo R-CNN forward, chained head masks, 285-entry
# route table): full GPU suite, default bench line, headline profile + HBM counters, steady-state step profiles, fuzz: -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/final_gpu_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r04_final6.json 2> gpurun_out/bench_r04_final6.err
bash tools/gpu_profile.sh r04 > gpurun_out/final_profile5.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_r04 r04x >> gpurun_out/final_profile5.log 2>&1
bash tools/gpu_profile_step.sh r04dsgn --full --pairs 1 --reps 1 >> gpurun_out/final_profile5.log 2>&1
bash tools/gpu_profile_step.sh r04r101 --r101 --pairs 1 --reps 1 >> gpurun_out/final_profile5.log 2>&1
for s in 16 17; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r04_fuzz_e.log 2>&1
cat gpurun_out/final_gpu_tests.log; tail -2 gpurun_out/r04_fuzz_e.log; tail -c 300 gpurun_out/bench_r04_final6.json
