mkdir -p gpurun_out
python profiles/epilogue_ab.py 2>&1 | grep "us$" > gpurun_out/ab_t.txt
python profiles/phase_times.py 2>&1 | grep "blocks" >> gpurun_out/ab_t.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_snunet.py tests/test_gpu_unet.py tests/test_gpu_changeformer.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
