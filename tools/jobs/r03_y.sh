#!/bin/bash
cd "$(dirname "$0")/../.."
for lg in 1 2 3 4 5; do
  echo "== cut table in LDS, pass lg $lg"; timeout 300 python tools/fused_trace.py 500000 -1 -1 $lg 2>&1 | sed -n 2,9p | grep -E "span|first wave|mean wave|last wave|phase total"
done
for lg in 2 3 5; do
  echo "== cut words from global memory, pass lg $lg"; FUSED_NO_CUT_TABLE=1 timeout 300 python tools/fused_trace.py 500000 -1 -1 $lg 2>&1 | sed -n 2,9p | grep -E "span|first wave|mean wave|last wave|phase total"
done
