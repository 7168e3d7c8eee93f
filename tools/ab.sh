mkdir -p gpurun_out
for s in 256 512 1024; do
KSMI_WGRAD_SPLITS=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_bench_s$s.json 2>gpurun_out/ab_bench.err
done
KSMI_WGRAD_SPLITS=512 BENCH_DETAIL=wgrad python bench.py --steps 5 --warmup 2 --no-cpu-baseline --time-all > gpurun_out/ab_detail.txt 2>&1
