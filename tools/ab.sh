mkdir -p gpurun_out
python bench.py --model mae --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_mae.json 2>gpurun_out/ab_bench.err
python bench.py --model mae --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_mae64.json 2>>gpurun_out/ab_bench.err
python bench.py --model mae --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail_mae.txt 2>&1
