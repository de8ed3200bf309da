#!/bin/bash
# Counters for the convolution / RoI kernels (tools/pmc_layers.py): separate rocprofv3 --pmc passes (never with trace domains), the program
# directly after "--", + one kernel trace for the durations.   usage: tools/gpu_profile_layers.sh <tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r04layers}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P="python3 $R/tools/pmc_layers.py --manifest $OUT/manifest.json ${ADV_PMC_ONLY:+--only $ADV_PMC_ONLY}"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $P > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- $P > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $P > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $P > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $P > $OUT/write.log 2>&1
tail -1 $OUT/trace.log $OUT/pmc1.log $OUT/pmc2.log $OUT/fetch.log $OUT/write.log
python3 $R/tools/summarize_pmc_layers.py $OUT $OUT/summary.json | tail -40
rm -rf $OUT/trace/*/*.db 2>/dev/null
