mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_snunet.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail.txt 2>&1
KSMI_WGRAD_WGS=2048 KSMI_WGRAD_SPLITS=512 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench2.json 2>gpurun_out/ab_bench.err
KSMI_WGRAD_WGS=2048 KSMI_WGRAD_SPLITS=512 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail2.txt 2>&1
