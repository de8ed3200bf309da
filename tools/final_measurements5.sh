#!/bin/bash
# round 5's last measurement set: default bench line, kernel statistics + HBM counters of bench.py, steady-state step profiles of both
# detector graphs (B = 1, B = 4, R101), both upstream stand-in traces, fuzz  -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_r05_final.err
bash tools/gpu_profile.sh r05 > gpurun_out/final_profile5.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_r05 r05 >> gpurun_out/final_profile5.log 2>&1
bash tools/gpu_profile_step.sh r05dsgn --full --pairs 1 --reps 1 >> gpurun_out/final_profile5.log 2>&1
bash tools/gpu_profile_step.sh r05dsgnb4 --full --pairs 4 --reps 1 >> gpurun_out/final_profile5.log 2>&1
bash tools/gpu_profile_step.sh r05r101 --r101 --pairs 1 --reps 1 >> gpurun_out/final_profile5.log 2>&1
for s in 21 22; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r05_fuzz.log 2>&1
tail -c 600 gpurun_out/bench_r05_final.json; tail -2 gpurun_out/r05_fuzz.log
