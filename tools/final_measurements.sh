#!/bin/bash
# The round's measurement set in one go (GPU box): full GPU suite, default bench line, headline profile + HBM counters, steady-state step
# profiles of both detector graphs, per-kernel counters, per-layer tables, RoIAlign backward, pipeline agreement, fuzz.  -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/final_gpu_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err
bash tools/gpu_profile.sh r04 > gpurun_out/final_profile.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_r04 r04x >> gpurun_out/final_profile.log 2>&1
bash tools/gpu_profile_step.sh r04dsgn --full --pairs 1 --reps 1 >> gpurun_out/final_profile.log 2>&1
bash tools/gpu_profile_step.sh r04dsgnb4 --full --pairs 4 --reps 1 >> gpurun_out/final_profile.log 2>&1
bash tools/gpu_profile_step.sh r04r101 --r101 --pairs 1 --reps 1 >> gpurun_out/final_profile.log 2>&1
bash tools/gpu_profile_layers.sh r04layers >> gpurun_out/final_profile.log 2>&1
python tools/bench_conv3d_layers.py > gpurun_out/r04_conv3d_layers.jsonl 2>/dev/null
python tools/bench_roi_bwd.py > gpurun_out/r04_roi_bwd.jsonl 2>/dev/null
python tools/bench_roi_bwd_phases.py > gpurun_out/r04_roi_bwd_phases.json 2>/dev/null
python tools/bench_s2_t2_tiles.py > gpurun_out/r04_s2_t2_tiles.jsonl 2>/dev/null
python tools/pipeline_agreement.py --out gpurun_out/pipeline_agreement.json > /dev/null 2>&1
for s in 1 2 3 4 5 6; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/r04_fuzz.log 2>&1
cat gpurun_out/final_gpu_tests.log; tail -3 gpurun_out/r04_fuzz.log
