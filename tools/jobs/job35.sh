cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu --no-host-fed --dims 1024 1024 256 --steps 5 --warmup 1 2>&1 | tail -15
