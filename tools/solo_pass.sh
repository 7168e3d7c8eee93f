R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_solo_stats -o stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo > $R/gpurun_out/prof_r02_solo.log 2>&1 < /dev/null
cd $R
python profiles/summarize.py gpurun_out/prof_r02_solo gpurun_out/r02_snunet_solo_summary.md "SNUNet-ECAM bs=32 bf16 train step on ONE stream (KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0)" "KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo"
rm -rf gpurun_out/prof_r02_solo_stats
python bench.py > gpurun_out/r02_bench_snunet.json 2>/dev/null < /dev/null
