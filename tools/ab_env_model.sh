#!/bin/bash
# same-box A/B of environment switches on one bench model: bash tools/ab_env_model.sh "<bench args>" "<VAR=val ...>" "<VAR=val ...>" ...
m="$1"; shift
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep [$m] [$cfg] $v"
  done
done
