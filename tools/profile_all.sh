#!/bin/bash
# full evidence set of a round: bench lines (contract form) + rocprofv3 summaries for the four model families
# usage (GPU box): bash tools/profile_all.sh <tag>
TAG=${1:-r01g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_${TAG}_snunet.json 2> $R/gpurun_out/bench_${TAG}.err
for m in changeformer floodvit unet; do python $R/bench.py --model $m --no-cpu-baseline > $R/gpurun_out/bench_${TAG}_$m.json 2>> $R/gpurun_out/bench_${TAG}.err; done
bash $R/tools_profile.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
cd /tmp && export TMPDIR=/tmp
for m in changeformer floodvit unet; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_${m}_stats -o stats -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_$m.log 2>&1
done
cd $R
for m in "" _changeformer _floodvit _unet; do python profiles/summarize.py gpurun_out/prof_${TAG}$m gpurun_out/${TAG}${m}_summary.md >> gpurun_out/prof_${TAG}.log 2>&1; done
