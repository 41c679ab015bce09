# the resolver's partition without global atomics (table [stretch][rank] + k_tie_colscan): per-kernel times, then the tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/rt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/rt -o t -- python tools/resolver_probe.py > gpurun_out/rt.log 2>&1
grep "resolver wall" gpurun_out/rt.log | cut -c1-22
python tools/rocpd_summary.py gpurun_out/rt/*.db 2>&1 | grep "k_tie_"
rm -rf gpurun_out/rt
timeout 1500 python -m pytest tests/test_gpu_exact_ties.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python tools/fuzz_round6.py 8000 40 2>&1 | tail -2
