cd $GRAFT_REPO_ROOT
DSI_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --events 200000 > gpurun_out/job21_full.log 2>&1
grep -v "NCCL WARN\|^$" gpurun_out/job21_full.log | head -60
