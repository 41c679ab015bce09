# the exact tie resolver at configs[1]'s size (tools/resolver_probe.py): kernel trace of every resolver kernel, the dispatches of
# the last call one by one, and one SQ counter pass (what binds the event pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/rt gpurun_out/rp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/rt -o t -- python tools/resolver_probe.py > gpurun_out/rt.log 2>&1
python tools/rocpd_summary.py gpurun_out/rt/*.db > gpurun_out/r06_tie_resolver_kernel_trace_stats.txt 2>&1
(echo "# one call of dsi_mapper_resolve_near_ties at configs[1] (tools/resolver_probe.py, third call), dispatch by dispatch"
 python tools/trace_timeline.py gpurun_out/rt/*.db k_tie k_packet_geometry --last 15
 grep "resolver wall" gpurun_out/rt.log) > gpurun_out/r06_tie_resolver_timeline.txt
cat gpurun_out/r06_tie_resolver_timeline.txt | cut -c1-150
rm -rf gpurun_out/rt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d gpurun_out/rp -o p -- python tools/resolver_probe.py > gpurun_out/rp.log 2>&1
python tools/rocpd_summary.py gpurun_out/rp/*.db 2>&1 | grep "k_tie_\|^kernel\|^##" > gpurun_out/r06_tie_resolver_pmc_counters.txt
grep "k_tie_hits_binned" gpurun_out/r06_tie_resolver_pmc_counters.txt
rm -rf gpurun_out/rp
