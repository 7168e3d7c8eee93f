mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench.json 2>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model changeformer > gpurun_out/ab_bench_cf.json 2>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model unet > gpurun_out/ab_bench_unet.json 2>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model floodvit > gpurun_out/ab_bench_fv.json 2>gpurun_out/ab_bench.err
BENCH_DETAIL=igemm python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail.txt 2>&1
