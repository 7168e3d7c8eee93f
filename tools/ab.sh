mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_snunet.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab_tests.txt
for m in 0; do KSMI_WDBG=$m python profiles/wgrad_ab.py 2>&1 | grep wdbg; done > gpurun_out/wgrad_ab.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model changeformer > gpurun_out/ab_bench_cf.json 2>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model unet > gpurun_out/ab_bench_unet.json 2>gpurun_out/ab_bench.err
