#!/bin/bash
# same-box A/B of the level-1 Up weight gradient on the GEMM path (GPU box) -> gpurun_out/up128.txt
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_snunet.py -x -q 2>&1 | tail -3
bash tools/ab_env.sh "KSMI_UP_WGRAD128=0" "KSMI_UP_WGRAD128=1"
for n in 0 1; do
KSMI_UP_WGRAD128=$n BENCH_LAUNCH_MAP=gpurun_out/map_up128_$n.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - $n <<'PY'
import json, sys
m = json.load(open(f"gpurun_out/map_up128_{sys.argv[1]}.json"))
print(f"UP_WGRAD128={sys.argv[1]} solo step {sum(e['ms'] for e in m):.3f} ms")
for e in m:
    if "Up2_" in e["tag"]: print(f"   {e['kind']:22s} {e['tag'][:50]:50s} {e['ms']*1e3:7.1f} us  {e['kernels']}")
PY
done
} > gpurun_out/up128.txt 2>&1
cat gpurun_out/up128.txt
