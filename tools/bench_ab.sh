#!/bin/bash
# eager vs HIP-graph replay of the fused train step, one line per run (usage on the GPU box: bash tools/bench_ab.sh)
for m in snunet changeformer floodvit unet; do
  for g in "" "--graph"; do
    python bench.py --model $m --no-cpu-baseline $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', '${g:-eager}', d['value'], 'tiles/s', d['ms_per_step'], 'ms')"
  done
done
