# through gpurun: per-workgroup phase CSVs (full 128x128 grid and a quarter of it) for every build/wg_timing_* binary given in W
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build/wgt_inputs
[ -f build/wgt_inputs/a.bin ] || timeout -k 10 300 python tools/wg_timing.py gen build/wgt_inputs
for w in $W; do
  n=$(basename $w)
  OUT=gpurun_out/wg_$n; mkdir -p $OUT
  head -c $((8192*672)) build/wgt_inputs/a.bin > $OUT/a4.bin
  head -c $((8192*672)) build/wgt_inputs/b.bin > $OUT/b4.bin
  timeout -k 10 120 $w build/wgt_inputs/delta.bin build/wgt_inputs/a.bin build/wgt_inputs/b.bin 0.3 > $OUT/full.csv 2> $OUT/full.txt
  timeout -k 10 120 $w build/wgt_inputs/delta.bin $OUT/a4.bin $OUT/b4.bin 0.3 > $OUT/quarter.csv 2> $OUT/quarter.txt
  rm -f $OUT/*.bin
  python tools/wg_timing.py report $OUT/full.csv | head -1
  python tools/wg_timing.py report $OUT/quarter.csv | head -1
done
