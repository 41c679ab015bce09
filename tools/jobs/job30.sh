cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
export TMPDIR=/tmp
prof() { # tag, args
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr30 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed $2 > $GRAFT_REPO_ROOT/gpurun_out/tr30.log 2>&1)
echo "$1:"; python tools/rocpd_summary.py gpurun_out/tr30/*.db | grep "plane_coef\|vote\|sort"; rm -rf gpurun_out/tr30
}
prof stereo "--steps 20 --warmup 2"
prof windows "--workload windows --steps 30 --warmup 3"
