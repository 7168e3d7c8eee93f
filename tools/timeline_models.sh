cd /tmp && export TMPDIR=/tmp
R=/root/repo
for m in floodvit changeformer; do
  extra=""; [ $m = changeformer ] && extra="--channels 4"
  rm -rf /tmp/tr_$m
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$m -o t -- python $R/bench.py --model $m $extra --steps 6 --warmup 3 --no-cpu-baseline --no-solo > /tmp/tr_$m.log 2>&1
  f=$(find /tmp/tr_$m -name "*kernel_trace.csv" | head -1)
  echo "== $m $f"
  mk=adam; [ $m = changeformer ] && mk=sgd; python $R/tools/timeline.py $f $mk:4 2>&1 | head -40
done
