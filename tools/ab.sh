mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_mae.py tests/test_gpu_floodvit.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/ab_tests.txt
