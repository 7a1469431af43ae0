set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wg_timing
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value $WG_TIMING_FLAGS -o $OUT/wg_timing tools/wg_timing.hip cofhe_amd/csrc/wire.hip
timeout -k 10 300 python tools/wg_timing.py gen $OUT
timeout -k 10 120 $OUT/wg_timing $OUT/delta.bin $OUT/a.bin $OUT/b.bin > $OUT/wg.csv
python tools/wg_timing.py report $OUT/wg.csv | tee $OUT/report.txt
rm -f $OUT/wg_timing $OUT/a.bin $OUT/b.bin
