#!/bin/bash
# round 6, F(4x4,3x3) kernel work: parity of the shipped build, the layer times of the libraries given, the phase ablation of the hooks build
# usage (on the GPU box): bash tools/gpu_r06_wino4.sh <tag> [lib.so ...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r06_$TAG
mkdir -p $O
[ -x tools/_build/dma_probe ] && timeout 60 tools/_build/dma_probe > $O/dma_probe.jsonl 2>&1
timeout 900 python -m pytest tests/test_wino4.py -m gpu -x -q > $O/pytest_wino4.log 2>&1
tail -3 $O/pytest_wino4.log
timeout 900 python tools/bench_wino4_variants.py "$@" > $O/variants.jsonl 2> $O/variants.err
cat $O/variants.jsonl
timeout 600 python tools/bench_wino4_phases.py > $O/phases.jsonl 2> $O/phases.err
cat $O/phases.jsonl | cut -c1-200
cat $O/dma_probe.jsonl
