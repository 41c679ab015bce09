cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for t in 4 5 6 3; do
(cd /tmp && DSI_COEF_TILE=$t timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr28 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --dims 1024 1024 256 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/tr28.log 2>&1)
echo "tile $t 1024:"; python tools/rocpd_summary.py gpurun_out/tr28/*.db | grep "plane_coef\|vote"; rm -rf gpurun_out/tr28
(cd /tmp && DSI_COEF_TILE=$t timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr28 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --steps 10 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/tr28.log 2>&1)
echo "tile $t stereo:"; python tools/rocpd_summary.py gpurun_out/tr28/*.db | grep "plane_coef\|vote"; rm -rf gpurun_out/tr28
done
