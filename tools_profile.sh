#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box via gpurun); summaries land in gpurun_out/prof_*
# usage: bash tools_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_stats -o stats -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_fetch -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_write -o write -- python $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write -type f | head -30
du -sh gpurun_out/prof_${TAG}_*
