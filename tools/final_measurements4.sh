#!/bin/bash
# the round's last measurement set (after csrc/resize.hip and the Winograd kernel's deep staging): full GPU suite, default bench line, steady-state
# step profiles of both detector graphs, per-layer 3D table, Winograd tile table, fuzz: -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/final_gpu_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r04_final5.json 2> gpurun_out/bench_r04_final5.err
bash tools/gpu_profile_step.sh r04dsgn --full --pairs 1 --reps 1 > gpurun_out/final_profile4.log 2>&1
bash tools/gpu_profile_step.sh r04dsgnb4 --full --pairs 4 --reps 1 >> gpurun_out/final_profile4.log 2>&1
bash tools/gpu_profile_step.sh r04r101 --r101 --pairs 1 --reps 1 >> gpurun_out/final_profile4.log 2>&1
python tools/bench_conv3d_layers.py > gpurun_out/r04_conv3d_layers.jsonl 2>/dev/null
python tools/bench_wino_tiles.py > gpurun_out/r04_wino_tiles.jsonl 2>/dev/null
for s in 13 14 15; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r04_fuzz_d.log 2>&1
cat gpurun_out/final_gpu_tests.log; tail -3 gpurun_out/r04_fuzz_d.log
