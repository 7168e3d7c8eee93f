#!/bin/bash
# same-box A/B of environment switches on the headline step: bash tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" ...  (one bench line per setting, repeated twice)
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep [$cfg] $v"
  done
done
