#!/bin/bash
# parity + solo timing of the Up weight-gradient launches (GPU box): bash tools/ab_upred.sh -> gpurun_out/upred.txt
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "up_" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_snunet.py -x -q 2>&1 | tail -3
for rep in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])"; done
BENCH_LAUNCH_MAP=gpurun_out/map_upred.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import json
m = json.load(open("gpurun_out/map_upred.json"))
print(f"solo step {sum(e['ms'] for e in m):.3f} ms")
for e in m:
    if e["kind"].startswith("up_gemm") or "2x2s2" in e["kind"]: print(f"   {e['kind']:22s} {e['tag'][:50]:50s} {e['ms']*1e3:7.1f} us  {e['kernels']}")
PY
} > gpurun_out/upred.txt 2>&1
cat gpurun_out/upred.txt
