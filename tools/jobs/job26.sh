cd $GRAFT_REPO_ROOT
for seed in 77 1 2 3; do
  timeout 600 python tools/fuzz_shapes.py $seed 2>&1 | tail -4
  DSI_PERSISTENT=1 timeout 600 python tools/fuzz_shapes.py $((seed+100)) 2>&1 | tail -2
done
