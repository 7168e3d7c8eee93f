#!/bin/bash
# second side stream for the weight gradients (KSMI_SIDE2=1): bit-equality with the single-stream step, then same-box A/B (GPU box)
mkdir -p gpurun_out
{
KSMI_SIDE2=1 timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_snunet.py -q -x 2>&1 | tail -3
bash tools/ab_env.sh "KSMI_SIDE2=0" "KSMI_SIDE2=1"
} > gpurun_out/side2.txt 2>&1
cat gpurun_out/side2.txt
