# usage (through gpurun): bash tools/gpu_wide_timing.sh [pairs]  -> gpurun_out/wt/wide_timing.txt: per-phase time of the wide composition
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wt
timeout -k 10 300 python3 tools/wg_timing.py gen gpurun_out/wt > /dev/null
timeout -k 10 120 build/wide_timing gpurun_out/wt/delta.bin gpurun_out/wt/a.bin gpurun_out/wt/b.bin ${1:-64} | tee gpurun_out/wt/wide_timing.txt
