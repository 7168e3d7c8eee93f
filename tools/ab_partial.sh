#!/bin/bash
# same-box A/B: 16-channel / thin-head convolutions on the persistent kernels (default) against the first-generation route
run() { v=$(env $1 python bench.py --model $2 --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); echo "[$1] [$2] $v"; }
for m in ${MODELS:-unet siam-conc siam-diff bit-cd}; do
  run "${OFF:-KSMI_IGEMM3_PARTIAL=0}" $m
  run "A=1" $m
  run "${OFF:-KSMI_IGEMM3_PARTIAL=0}" $m
  run "A=1" $m
done
