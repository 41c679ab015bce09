#!/bin/bash
cd "$(dirname "$0")/../.."
FUSED_TRACE_WARM=40 timeout 300 python tools/fused_trace.py 2>&1 | tail -48
