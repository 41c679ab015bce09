cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/rt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/rt -o t -- python tools/resolver_probe.py > gpurun_out/rt.log 2>&1
python tools/rocpd_summary.py gpurun_out/rt/*.db 2>&1 | grep -i "tie\|kernel" | head -20
rm -rf gpurun_out/rt
