cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "depth_map_of_fusion" 2>&1 | grep "passed\|failed" | tail -2
export TMPDIR=/tmp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr24 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --workload windows --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/tr24.log 2>&1)
python tools/rocpd_summary.py gpurun_out/tr24/*.db | head -8; rm -rf gpurun_out/tr24
