#!/bin/bash
# The measurement set a round ends with (everything lands under gpurun_out/, the summaries are then copied to profiles/ by hand, named per round):
#   default bench line; kernel statistics + HBM counters of the headline leg (tools/gpu_profile.sh); steady-state step profiles of both
#   detector-shaped graphs (B = 1, B = 4, R101); per-kernel counters on named layer shapes (tools/gpu_profile_layers.sh); operator traces of the
#   small launches; both upstream stand-in traces; fuzz; the microarchitecture probes.
# usage (GPU box): bash tools/final_measurements.sh r06
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err
bash tools/gpu_profile.sh $TAG > gpurun_out/final_profile_$TAG.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof_$TAG $TAG >> gpurun_out/final_profile_$TAG.log 2>&1
bash tools/gpu_profile_step.sh ${TAG}dsgn --full --pairs 1 --reps 1 >> gpurun_out/final_profile_$TAG.log 2>&1
bash tools/gpu_profile_step.sh ${TAG}dsgnb4 --full --pairs 4 --reps 1 >> gpurun_out/final_profile_$TAG.log 2>&1
bash tools/gpu_profile_step.sh ${TAG}r101 --r101 --pairs 1 --reps 1 >> gpurun_out/final_profile_$TAG.log 2>&1
bash tools/gpu_profile_layers.sh ${TAG}layers >> gpurun_out/final_profile_$TAG.log 2>&1
python3 tools/trace_r101_small_ops.py > gpurun_out/${TAG}_r101_small_ops.json 2>/dev/null
python3 tools/trace_r101_small_ops.py --dsgn > gpurun_out/${TAG}_dsgn_small_ops.json 2>/dev/null
[ -x tools/profile_upstream_standin.sh ] && bash tools/profile_upstream_standin.sh ${TAG}upstream >> gpurun_out/final_profile_$TAG.log 2>&1
[ -x tools/profile_upstream_dsgn_standin.sh ] && bash tools/profile_upstream_dsgn_standin.sh ${TAG}upstreamdsgn >> gpurun_out/final_profile_$TAG.log 2>&1
for s in 41 42; do timeout 600 python tools/fuzz_gpu.py --cases 400 --seed $s --big 1 2>&1 | tail -1; done > gpurun_out/${TAG}_fuzz.log 2>&1
python3 tools/bench_wino4.py > gpurun_out/${TAG}_wino4_layers.jsonl 2>/dev/null
python3 tools/bench_wino4_phases.py > gpurun_out/${TAG}_wino4_phases.jsonl 2>/dev/null
[ -x tools/_build/mfma_valu_probe ] && tools/_build/mfma_valu_probe > gpurun_out/${TAG}_mfma_valu_probe.json 2>&1
tail -c 600 gpurun_out/bench_${TAG}_final.json; tail -2 gpurun_out/${TAG}_fuzz.log
