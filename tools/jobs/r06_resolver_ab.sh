# same-box A/B of resolver variants (experiments flavour): the event pass walking the voxels plane by plane (default) or in
# the cand list's order (DSI_TIE_BY_PLANE=0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DSI_ENGINE_EXPERIMENTS=1
run() {
  mkdir -p gpurun_out/rt
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/rt -o t -- python tools/resolver_probe.py > gpurun_out/rt.log 2>&1
  echo "== $*"; grep -o "elapsed_ms.: [0-9.]*" gpurun_out/rt.log | tail -2 | tr '\n' ' '; echo
  python tools/rocpd_summary.py gpurun_out/rt/*.db 2>&1 | grep "k_tie_hits\|k_tie_desc"
  rm -rf gpurun_out/rt
}
run DSI_TIE_BY_PLANE=0
run DSI_TIE_BY_PLANE=1
run DSI_TIE_BY_PLANE=0
run DSI_TIE_BY_PLANE=1
unset DSI_ENGINE_EXPERIMENTS
timeout 900 python -m pytest tests/test_gpu_exact_ties.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python tools/fuzz_fused_and_resolver.py 61000 150 2>&1 | tail -2
