#!/bin/bash
# PMC passes for the MFMA convolution (tools/bench_kernels.py --conv): matrix-pipe busy cycles, wave-cycle breakdown
# (parked / issue-stalled / issuing), instruction counts, LDS conflicts, clock.  Separate passes, no trace domains with --pmc.
set -u
TAG=${1:-r01conv}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $R/tools/bench_kernels.py --conv > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $R/tools/bench_kernels.py --conv > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 $R/tools/bench_kernels.py --conv > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_kernels.py --conv > $OUT/trace.log 2>&1
tail -2 $OUT/pmc1.log; ls $OUT/*/*/ | head
