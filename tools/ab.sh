mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
python bench.py --model changeformer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench_cf.json 2>>gpurun_out/ab_bench.err
python bench.py --model unet --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench_unet.json 2>>gpurun_out/ab_bench.err
python bench.py --model floodvit --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench_fv.json 2>>gpurun_out/ab_bench.err
