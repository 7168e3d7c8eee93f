mkdir -p gpurun_out
KSMI_DP_FORCE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_sn_dp.json 2> gpurun_out/ab_sn_dp.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_sn_tr.json 2> gpurun_out/ab_sn_tr.err
