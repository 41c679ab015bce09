cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "NCCL WARN\|^$" | tail -25
