#!/bin/bash
# SNUNet half of tools/profile_all.sh (rocprofv3 passes + summaries + per-stage table) and the bench lines of every family
# usage (GPU box): bash tools/profile_snunet.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
bash $R/tools/profile.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
python profiles/summarize.py gpurun_out/prof_${TAG} gpurun_out/${TAG}_snunet_summary.md "SNUNet-ECAM bs=32 bf16 train step on THREE HIP streams (kernel durations overlap: their sum exceeds the step; single-stream durations and counters: ${TAG}_snunet_solo_summary.md)" "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo" gpurun_out/${TAG}_snunet_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
python profiles/summarize.py gpurun_out/prof_${TAG}_solo gpurun_out/${TAG}_snunet_solo_summary.md "SNUNet-ECAM bs=32 bf16 train step on ONE stream (KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0)" "KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo" gpurun_out/${TAG}_snunet_solo_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 BENCH_LAUNCH_MAP=$R/gpurun_out/${TAG}_snunet_launch_map.json python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/prof_${TAG}.log
python profiles/stage_traffic.py gpurun_out/prof_${TAG}_solo gpurun_out/${TAG}_snunet_launch_map.json gpurun_out/${TAG}_snunet_stage_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
cp gpurun_out/${TAG}_snunet*_traffic.json profiles/ 2>/dev/null
python $R/bench.py > $R/gpurun_out/${TAG}_bench_snunet.json 2> $R/gpurun_out/${TAG}_bench.err
for m in changeformer floodvit unet mae siam-conc siam-diff bit-cd; do python $R/bench.py --model $m --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_$m.json 2>> $R/gpurun_out/${TAG}_bench.err; done
python $R/bench.py --model changeformer --channels 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_changeformer_slc.json 2>> $R/gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/prof_${TAG}_*fetch gpurun_out/prof_${TAG}_*write gpurun_out/prof_${TAG}_solo_stats gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_*mfma
du -sh gpurun_out; ls gpurun_out | grep ${TAG}_ | head -40; for f in gpurun_out/${TAG}_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
