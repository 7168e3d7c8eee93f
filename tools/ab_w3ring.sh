#!/bin/bash
# wgrad3 fragment prefetch depth (KSMI_WGRAD3_DEEP) and LDS ring (KSMI_WGRAD3_NST): parity tests, then same-box A/B of the headline step and of the 3x3 weight-gradient launches
# (GPU box): bash tools/ab_w3ring.sh   -> gpurun_out/w3ring.txt, gpurun_out/map_deep{2,0}.json
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad3 or conv3x3" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_snunet.py -x -q 2>&1 | tail -3
bash tools/ab_env.sh "KSMI_WGRAD3_DEEP=0" "KSMI_WGRAD3_DEEP=1" "KSMI_WGRAD3_DEEP=1 KSMI_WGRAD3_NST=4"
for n in 0 1; do
  KSMI_WGRAD3_DEEP=$n BENCH_LAUNCH_MAP=gpurun_out/map_deep$n.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python - $n <<'PY'
import json, sys
m = json.load(open(f"gpurun_out/map_deep{sys.argv[1]}.json"))
r = [e for e in m if e["kind"] == "igemm_wgrad<3x3s1>"]
print(f"DEEP={sys.argv[1]}: 3x3 wgrad n={len(r)} total {sum(e['ms'] for e in r)*1e3:.1f} us; solo step {sum(e['ms'] for e in m):.3f} ms")
for e in r: print(f"   {e['tag'][:60]:60s} {e['ms']*1e3:7.1f} us  {e['kernels']}")
PY
done
} > gpurun_out/w3ring.txt 2>&1
tail -60 gpurun_out/w3ring.txt
