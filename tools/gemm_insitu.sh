#!/bin/bash
# FloodViT (or MODEL=mae ...) step under rocprofv3 with the gemm2 instance pinned (KSMI_GEMM2_MT / KSMI_GEMM2_NS) -> per-instance in-situ durations.
# usage (GPU box): bash tools/gemm_insitu.sh "auto 2:2 3:2 4:2 4:3 5:3"   -> gpurun_out/r03/gemm_insitu.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
OUT=$R/gpurun_out/r03/gemm_insitu.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for c in $1; do
  unset KSMI_GEMM2_MT KSMI_GEMM2_NS
  if [ "$c" != auto ]; then export KSMI_GEMM2_MT=${c%%:*} KSMI_GEMM2_NS=${c##*:}; fi
  rm -rf /tmp/prof_gi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_gi -o stats -- python $R/bench.py --model ${MODEL:-floodvit} --steps 5 --warmup 2 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | cut -c1-140 >> $OUT
  echo "== $c" >> $OUT
  python $R/tools/gemm_durations.py /tmp/prof_gi/stats_results.db | grep -v tn >> $OUT
done
cat $OUT
