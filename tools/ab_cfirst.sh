timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_input_pipeline.py -x -q -k "first or raw or pipeline or preprocess" 2>&1 | tail -3
BENCH_LAUNCH_MAP=gpurun_out/snmap_cfirst.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('snunet', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('snunet', d['value'], d['ms_per_step'])"
