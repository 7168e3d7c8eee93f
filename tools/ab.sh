mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_tokens.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/ab_tests.txt
