set -x
timeout 600 python -m pytest "tests/test_gpu_graph.py::test_side_lane_equals_single_stream" -k changeformer -x -q 2>&1 | tail -5
timeout 600 python tools/race_probe.py changeformer 2 delay 2>&1 | tail -8
for rep in 1 2; do for cfg in KSMI_CF_SIDE_TOKENS=0 KSMI_CF_SIDE_TOKENS=1; do
 v=$(env $cfg python bench.py --model changeformer --channels 4 --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "rep $rep [$cfg] $v"
done; done
timeout 900 python -m pytest tests/test_gpu_changeformer.py tests/test_gpu_cformer.py -x -q 2>&1 | tail -3
