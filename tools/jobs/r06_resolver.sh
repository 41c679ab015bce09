cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_exact_ties.py -q -m gpu -x > gpurun_out/res_tests.txt 2>&1
tail -3 gpurun_out/res_tests.txt
mkdir -p gpurun_out/rt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/rt -o t -- python tools/resolver_probe.py > gpurun_out/rt.log 2>&1
grep "resolver wall" gpurun_out/rt.log | cut -c1-30
python tools/rocpd_summary.py gpurun_out/rt/*.db 2>&1 | grep -i "tie\|kernel" | head -20
rm -rf gpurun_out/rt
python tools/resolver_probe.py 2>&1 | grep "resolver wall" | cut -c1-30
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_process.py -q -m gpu -x -k "configs4_end_to_end or exact or configs3" > gpurun_out/res_tests2.txt 2>&1
tail -3 gpurun_out/res_tests2.txt
