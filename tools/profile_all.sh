#!/bin/bash
# full evidence set of a round: bench lines (contract form) + rocprofv3 summaries for the model families
# usage (GPU box): bash tools/profile_all.sh <tag>      -> gpurun_out/<tag>_* ; copy what is to be judged into profiles/
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
bash $R/tools/profile.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
cd /tmp && export TMPDIR=/tmp
for m in changeformer floodvit unet mae; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_${m}_stats -o stats -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-solo > $R/gpurun_out/prof_${TAG}_$m.log 2>&1
done
# HBM traffic counters of the other families (own passes: --pmc only with --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass)
for m in changeformer floodvit unet mae; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_${m}_fetch -o fetch -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-solo >> $R/gpurun_out/prof_${TAG}_$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_${TAG}_${m}_write -o write -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-solo >> $R/gpurun_out/prof_${TAG}_$m.log 2>&1
done
# MFMA-busy counters (own pass)
for m in changeformer floodvit unet mae; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_${TAG}_${m}_mfma -o mfma -- python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-solo >> $R/gpurun_out/prof_${TAG}_$m.log 2>&1
done
cd $R
python profiles/summarize.py gpurun_out/prof_${TAG} gpurun_out/${TAG}_snunet_summary.md "SNUNet-ECAM bs=32 bf16 train step on THREE HIP streams (kernel durations overlap: their sum exceeds the step; single-stream durations and counters: ${TAG}_snunet_solo_summary.md)" "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo" gpurun_out/${TAG}_snunet_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
python profiles/summarize.py gpurun_out/prof_${TAG}_solo gpurun_out/${TAG}_snunet_solo_summary.md "SNUNet-ECAM bs=32 bf16 train step on ONE stream (KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0)" "KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo" gpurun_out/${TAG}_snunet_solo_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
# per-stage counted traffic / MFMA-busy of one single-stream step: per-dispatch PMC rows aligned with the launch table of the same plan
KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 BENCH_LAUNCH_MAP=$R/gpurun_out/${TAG}_snunet_launch_map.json python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/prof_${TAG}.log
python profiles/stage_traffic.py gpurun_out/prof_${TAG}_solo gpurun_out/${TAG}_snunet_launch_map.json gpurun_out/${TAG}_snunet_stage_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
for m in changeformer floodvit unet mae; do
  python profiles/summarize.py gpurun_out/prof_${TAG}_$m gpurun_out/${TAG}_${m}_summary.md "$m train step (bench.py --model $m)" "python bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-solo" gpurun_out/${TAG}_${m}_traffic.json >> gpurun_out/prof_${TAG}.log 2>&1
done
# the bench lines (contract form) last: roofline.traffic is read from the per-kernel tables the passes above just produced
cp gpurun_out/${TAG}_*_traffic.json profiles/ 2>/dev/null
python $R/bench.py > $R/gpurun_out/${TAG}_bench_snunet.json 2> $R/gpurun_out/${TAG}_bench.err
for m in changeformer floodvit unet mae siam-conc siam-diff bit-cd; do python $R/bench.py --model $m --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_$m.json 2>> $R/gpurun_out/${TAG}_bench.err; done
python $R/bench.py --model changeformer --channels 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_changeformer_slc.json 2>> $R/gpurun_out/${TAG}_bench.err
python $R/profiles/stream_probe.py 2>&1 | grep MiB > $R/gpurun_out/${TAG}_stream_probe.txt
python $R/profiles/gemm_probe.py 2>&1 | grep rows > $R/gpurun_out/${TAG}_gemm_probe.txt
python $R/tools/gemm_durations.py gpurun_out/prof_${TAG}_floodvit_stats/stats_results.db > $R/gpurun_out/${TAG}_floodvit_gemm_durations.txt 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-solo > $R/gpurun_out/${TAG}_torchrun_stdout.txt 2> $R/gpurun_out/${TAG}_torchrun_stderr.txt
# the raw rocpd databases are tens of MiB each and gpurun merges at most 64 MiB back: keep the summaries, drop the databases
rm -rf gpurun_out/prof_${TAG}_*_fetch gpurun_out/prof_${TAG}_*_write gpurun_out/prof_${TAG}_solo_stats gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_mfma gpurun_out/prof_${TAG}_*_stats gpurun_out/prof_${TAG}_*_mfma
du -sh gpurun_out; ls gpurun_out | grep ${TAG}_ | head -40
