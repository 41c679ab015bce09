#!/bin/bash
# Round-2 profiles (on the GPU box): the default bench (configs[1]) with the full PMC set, then the
# configs[2] window stream and the configs[4] shape with the SQ / LDS / traffic passes.
#   tools/profile_r02.sh   -> gpurun_out/profiles_r02*/
set -u
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r02
bash tools/profile_shape.sh r02_windows "--workload windows"
bash tools/profile_shape.sh r02_1024 "--dims 1024 1024 256 --events 10000000"
bash tools/profile_shape.sh r02_cameras4 "--workload cameras4"
