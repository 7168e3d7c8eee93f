#!/bin/bash
# same-box: eager three-stream step against the captured graph under the runtime's graph knobs
run() { v=$(env $1 python bench.py $2 --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1); echo "[$1] [$2] $v"; }
for m in "" "--model floodvit"; do
  run "A=1" "$m"
  run "A=1" "$m --graph"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "$m --graph"
  run "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "$m --graph"
  run "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "$m --graph"
  run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "$m --graph"
done
