cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for x in 0 12000 25000 50000; do
(cd /tmp && DSI_SORT_EXTRA_LDS=$x timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr30 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --steps 10 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/tr30.log 2>&1)
echo "extra lds $x:"; python tools/rocpd_summary.py gpurun_out/tr30/*.db | grep "sort"; rm -rf gpurun_out/tr30
done
