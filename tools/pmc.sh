#!/bin/bash
# PMC passes over the conv micro-benchmark (run via gpurun).  usage: bash tools_pmc.sh <tag> "<MB_ONLY filter>"
TAG=${1:-pmc}; ONLY=${2:-L1 conv1_1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export MB_ONLY="$ONLY"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_a -o a -- python $R/profiles/microbench_conv.py > $R/gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_b -o b -- python $R/profiles/microbench_conv.py > $R/gpurun_out/${TAG}_b.log 2>&1
tail -3 $R/gpurun_out/${TAG}_a.log
