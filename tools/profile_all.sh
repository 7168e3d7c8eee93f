#!/bin/bash
# full evidence set of a round: bench lines (contract form) + rocprofv3 summaries for the four model families
# usage (GPU box): bash tools/profile_all.sh <tag>
TAG=${1:-r01g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_${TAG}_snunet.json 2> $R/gpurun_out/bench_${TAG}.err
for m in changeformer floodvit unet mae; do python $R/bench.py --model $m --no-cpu-baseline > $R/gpurun_out/bench_${TAG}_$m.json 2>> $R/gpurun_out/bench_${TAG}.err; done
bash $R/tools/profile.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
cd /tmp && export TMPDIR=/tmp
for m in changeformer floodvit unet mae; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_${m}_stats -o stats -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_$m.log 2>&1
done
cd $R
for m in "" _changeformer _floodvit _unet _mae; do python profiles/summarize.py gpurun_out/prof_${TAG}$m gpurun_out/${TAG}${m}_summary.md >> gpurun_out/prof_${TAG}.log 2>&1; done
python $R/profiles/stream_probe.py 2>&1 | grep MiB > $R/gpurun_out/${TAG}_stream_probe.txt
python $R/profiles/gemm_probe.py 2>&1 | grep rows > $R/gpurun_out/${TAG}_gemm_probe.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_torchrun_stdout.txt 2> $R/gpurun_out/${TAG}_torchrun_stderr.txt
