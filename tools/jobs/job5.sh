cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_windows.py -q -x 2>&1 | tail -30
echo "=== bench stereo"; timeout 900 python bench.py > gpurun_out/b_stereo.json 2> gpurun_out/b_stereo.err; tail -c 3000 gpurun_out/b_stereo.json; tail -5 gpurun_out/b_stereo.err
echo "=== bench windows"; timeout 900 python bench.py --workload windows > gpurun_out/b_windows.json 2> gpurun_out/b_windows.err; tail -c 2500 gpurun_out/b_windows.json; tail -5 gpurun_out/b_windows.err
echo "=== bench cameras4"; timeout 900 python bench.py --workload cameras4 --no-cpu > gpurun_out/b_cam4.json 2> gpurun_out/b_cam4.err; tail -c 2000 gpurun_out/b_cam4.json; tail -5 gpurun_out/b_cam4.err
echo "=== torchrun 1 rank engine collective"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu 2>&1 | tail -c 800
