#!/bin/bash
# solo times of the 3x3 weight-gradient launches under forced workgroup counts (GPU box) -> gpurun_out/w3wgs.txt
mkdir -p gpurun_out
{
for n in 0 256 384 512 768 1024; do
KSMI_WGRAD3_WGS=$n BENCH_LAUNCH_MAP=gpurun_out/map_w3wgs_$n.json python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - $n <<'PY'
import json, sys
m = json.load(open(f"gpurun_out/map_w3wgs_{sys.argv[1]}.json"))
r = [e for e in m if e["kind"] == "igemm_wgrad<3x3s1>"]
print(f"WGS={sys.argv[1]}: total {sum(e['ms'] for e in r)*1e3:.1f} us")
seen = {}
for e in r:
    k = e['tag'].split(' ', 1)[1]
    seen.setdefault(k, []).append(e['ms'] * 1e3)
for k, v in seen.items(): print(f"   {k:32s} {sum(v)/len(v):7.1f} us x{len(v)}")
PY
done
} > gpurun_out/w3wgs.txt 2>&1
cat gpurun_out/w3wgs.txt
