#!/bin/bash
# same-box A/B: Adam writes the bf16 operand copy itself (default) against the separate cast pass
run() { v=$(env $1 python bench.py --model $2 --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); echo "[$1] [$2] $v"; }
for m in ${MODELS:-floodvit mae}; do
  run "KSMI_ADAM_MIRROR=0" $m
  run "A=1" $m
  run "KSMI_ADAM_MIRROR=0" $m
  run "A=1" $m
done
