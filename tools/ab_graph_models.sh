#!/bin/bash
# same-box: eager step against the captured HIP graph, per model family (bench.py --graph)
run() { v=$(python bench.py $1 --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step'))" 2>&1 | tail -1); echo "[$1] $v"; }
for m in ${MODELS:-floodvit mae changeformer unet bit-cd siam-conc}; do
  run "--model $m"
  run "--model $m --graph"
done
