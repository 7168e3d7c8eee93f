mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab_tests.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model changeformer > gpurun_out/ab_cf.json 2>>gpurun_out/ab_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model floodvit > gpurun_out/ab_fv.json 2>>gpurun_out/ab_bench.err
KSMI_NO_HIPBLASLT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model changeformer > gpurun_out/ab_cf_no.json 2>>gpurun_out/ab_bench.err
