mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
KSMI_WGRAD_NT=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench2.json 2>gpurun_out/ab_bench.err
BENCH_DETAIL=wgrad python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail.txt 2>&1
KSMI_WGRAD_NT=2 BENCH_DETAIL=wgrad python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail2.txt 2>&1
