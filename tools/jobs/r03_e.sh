#!/bin/bash
cd "$(dirname "$0")/../.."
for fc in 24000 32000 40000; do echo "=== fixed $fc"; timeout 300 python tools/fused_trace.py 500000 -1 $fc 2>&1 | tail -22 | grep -E "kernel span|finish times|phase total"; done
for lg in 3 4; do echo "=== fixed 32000 pass_lg $lg"; timeout 300 python tools/fused_trace.py 500000 -1 32000 $lg 2>&1 | tail -22 | grep -E "kernel span|finish times|phase total|last wave|first wave|^wg"; done
