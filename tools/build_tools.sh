#!/bin/bash
# tools/build_tools.sh -- cross-compiles the GPU-side measurement helpers into build/ (git-ignored, shipped by gpurun):
#   build/wg_timing      diagnostic build of k_compose_wg with phase / clock stamps (tools/wg_timing.hip)
#   build/traffic_calib  record-copy kernel that calibrates FETCH_SIZE / WRITE_SIZE (tools/traffic_calib.hip)
#   build/inst_bench     per-instruction issue costs (tools/inst_bench.hip)
set -e
cd "$(dirname "$0")/.."
mkdir -p build
H=/opt/rocm/bin/hipcc
$H --offload-arch=gfx950 -O3 -std=c++17 -o build/traffic_calib tools/traffic_calib.hip &
$H --offload-arch=gfx950 -O2 -std=c++17 -o build/inst_bench tools/inst_bench.hip &
$H --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value $WG_TIMING_FLAGS -o build/wg_timing tools/wg_timing.hip cofhe_amd/csrc/wire.hip cofhe_amd/csrc/wide.hip &
wait
ls -la build/wg_timing build/traffic_calib build/inst_bench
