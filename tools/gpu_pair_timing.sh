set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wg_timing
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -o $OUT/pair_timing tools/pair_timing.hip
timeout -k 10 300 python tools/wg_timing.py gen $OUT
timeout -k 10 120 $OUT/pair_timing $OUT/delta.bin $OUT/a.bin $OUT/b.bin | tee $OUT/pair_report.txt
rm -f $OUT/pair_timing $OUT/a.bin $OUT/b.bin
