#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box via gpurun); raw outputs land in gpurun_out/prof_<tag>_*, the committed
# summary + per-kernel traffic table in profiles/ come from profiles/summarize.py.  Counters are collected in their own passes
# (--pmc never together with a trace domain other than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).
# usage: bash tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r05}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-solo $*"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_stats -o stats -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_fetch -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_write -o write -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_${TAG}_mfma -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_mfma.log 2>&1
# the same kernels alone on the machine: one stream (roofline.solo / frac_solo of the bench line)
# (PROFILE_SOLO=0 skips them: single-stream models)
if [ "${PROFILE_SOLO:-1}" = "1" ]; then
export KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_solo_stats -o stats -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_solo.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_fetch -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_solo_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_write -o write -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_solo_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_${TAG}_solo_mfma -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_solo_mfma.log 2>&1
unset KSMI_OVERLAP_WGRAD KSMI_OVERLAP_LANES
fi
cd $R
du -sh gpurun_out/prof_${TAG}_* | head
