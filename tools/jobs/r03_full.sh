#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
