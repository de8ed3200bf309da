#!/bin/bash
# second half of the round's measurement set, after the last kernel changes (any-width transposed kernel, fold mask): -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/final_gpu_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r04_final3.json 2> gpurun_out/bench_r04_final3.err
bash tools/gpu_profile_step.sh r04dsgn --full --pairs 1 --reps 1 > gpurun_out/final_profile2.log 2>&1
bash tools/gpu_profile_step.sh r04dsgnb4 --full --pairs 4 --reps 1 >> gpurun_out/final_profile2.log 2>&1
bash tools/gpu_profile_layers.sh r04layers >> gpurun_out/final_profile2.log 2>&1
python tools/bench_conv3d_layers.py > gpurun_out/r04_conv3d_layers.jsonl 2>/dev/null
for s in 7 8 9; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r04_fuzz_b.log 2>&1
cat gpurun_out/final_gpu_tests.log; tail -3 gpurun_out/r04_fuzz_b.log
