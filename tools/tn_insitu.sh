#!/bin/bash
# FloodViT (or MODEL=mae ...) step under rocprofv3 with the weight-gradient tiling pinned (KSMI_TN_BT : KSMI_TN_SPLIT : KSMI_TN_NS) -> in-situ durations.
# usage (GPU box): bash tools/tn_insitu.sh "auto 64:1:4 128:2:3"   -> gpurun_out/r03/tn_insitu.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
OUT=$R/gpurun_out/r03/tn_insitu.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for c in $1; do
  unset KSMI_TN_BT KSMI_TN_SPLIT KSMI_TN_NS
  if [ "$c" != auto ]; then IFS=: read bt sp ns <<< "$c"; export KSMI_TN_BT=$bt KSMI_TN_SPLIT=$sp KSMI_TN_NS=$ns; fi
  rm -rf /tmp/prof_ti
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ti -o stats -- python $R/bench.py --model ${MODEL:-floodvit} --steps 5 --warmup 2 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | cut -c1-140 >> $OUT
  echo "== $c" >> $OUT
  python $R/tools/gemm_durations.py /tmp/prof_ti/stats_results.db | grep "tn" >> $OUT
done
cat $OUT
