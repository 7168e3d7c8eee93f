mkdir -p gpurun_out
python profiles/stream_probe.py 2>&1 | grep MiB > gpurun_out/stream_probe.txt
