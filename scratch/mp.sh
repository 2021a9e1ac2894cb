mkdir -p gpurun_out
export DYNMM_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 > gpurun_out/bench_mp2.log 2>&1
tail -3 gpurun_out/bench_mp2.log | cut -c1-600
