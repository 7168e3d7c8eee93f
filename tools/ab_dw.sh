timeout 600 python -m pytest tests/test_gpu_cformer.py -x -q -k dwconv 2>&1 | tail -5
for cfg in KSMI_DW_ROW=0 KSMI_DW_ROW=1 "KSMI_DW_ROW=1 KSMI_DW_SEG=28"; do
 env $cfg BENCH_LAUNCH_MAP=gpurun_out/cfmap_$(echo $cfg | tr -c 'A-Za-z0-9\n' '_').json python bench.py --model changeformer --channels 4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
