cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_process.py -q -x -k "nary or four_cameras or plane_sharded or fusion" 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
export TMPDIR=/tmp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr27 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --workload cameras4 > $GRAFT_REPO_ROOT/gpurun_out/tr27.log 2>&1)
tail -1 gpurun_out/tr27.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cameras4 %.0f Mev/s step %.3f ms kern %.4f ms' % (d['value'], d['ms_per_step'], r['kernel_avg_ms']))"
python tools/rocpd_summary.py gpurun_out/tr27/*.db | head -12; rm -rf gpurun_out/tr27
