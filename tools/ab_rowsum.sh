set -x
timeout 900 python -m pytest tests/test_gpu_changeformer.py tests/test_gpu_floodvit.py tests/test_gpu_mae.py tests/test_gpu_bitcd.py tests/test_gpu_bitcd_tokens.py tests/test_gpu_unet.py tests/test_gpu_fcsiam.py tests/test_gpu_graph.py tests/test_gpu_bench_size.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do for m in "changeformer --channels 4" "floodvit"; do for cfg in KSMI_ROWSUM_BATCH=1 KSMI_ROWSUM_BATCH=16 KSMI_ROWSUM_BATCH=64; do
 v=$(env $cfg python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "rep $rep [$m] [$cfg] $v"
done; done; done
