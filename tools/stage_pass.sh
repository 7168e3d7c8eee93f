#!/bin/bash
# per-stage counters of one single-stream SNUNet step: the three single-stream --pmc passes + the launch table + profiles/stage_traffic.py
# usage (GPU box): bash tools/stage_pass.sh <tag>   -> gpurun_out/<tag>_snunet_stage_traffic.json, <tag>_snunet_launch_map.json
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-solo"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_fetch -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/stage_${TAG}.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_write -o write -- python $R/bench.py $ARGS >> $R/gpurun_out/stage_${TAG}.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_mfma -o mfma -- python $R/bench.py $ARGS >> $R/gpurun_out/stage_${TAG}.log 2>&1
BENCH_LAUNCH_MAP=$R/gpurun_out/${TAG}_snunet_launch_map.json python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/stage_${TAG}.log
cd $R
python profiles/stage_traffic.py gpurun_out/prof_${TAG}_solo gpurun_out/${TAG}_snunet_launch_map.json gpurun_out/${TAG}_snunet_stage_traffic.json
rm -rf gpurun_out/prof_${TAG}_solo_fetch gpurun_out/prof_${TAG}_solo_write gpurun_out/prof_${TAG}_solo_mfma
