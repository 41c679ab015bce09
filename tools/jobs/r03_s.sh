#!/bin/bash
cd "$(dirname "$0")/../.."
for lg in 1 2 3; do echo "== fused pass_lg $lg"; timeout 300 python tools/fused_trace.py 500000 -1 -1 $lg 2>&1 | grep -E "kernel span|finish times|first wave|last wave|phase total|barrier"; done
