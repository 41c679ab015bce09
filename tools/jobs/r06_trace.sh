cd $GRAFT_REPO_ROOT
echo "== defer on"; python tools/fused_trace.py 2>&1 | grep -v "^xcd\|^wg " | tail -16
echo "== defer off"; DSI_FUSED_DEFER=0 python tools/fused_trace.py 2>&1 | grep -v "^xcd\|^wg " | tail -16
