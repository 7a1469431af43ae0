# through gpurun: phase breakdown of the compose kernel (tools/wg_timing.hip builds under build/wg_timing_*), at the full
# 128x128 launch (4 workgroups per CU) and at a quarter of it (1 per CU)
set -e
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wg_timing2
mkdir -p $OUT
if [ -f build/wgt_inputs/a.bin ]; then cp build/wgt_inputs/*.bin $OUT/; else timeout -k 10 300 python tools/wg_timing.py gen $OUT; fi
head -c $((8192*672)) $OUT/a.bin > $OUT/a4.bin
head -c $((8192*672)) $OUT/b.bin > $OUT/b4.bin
for w in build/wg_timing_*; do
  n=$(basename $w)
  timeout -k 10 120 $w $OUT/delta.bin $OUT/a.bin $OUT/b.bin 0.3 > $OUT/$n.full.csv 2> $OUT/$n.full.txt
  timeout -k 10 120 $w $OUT/delta.bin $OUT/a4.bin $OUT/b4.bin 0.3 > $OUT/$n.quarter.csv 2> $OUT/$n.quarter.txt
  echo "=== $n full"; cat $OUT/$n.full.txt; python tools/wg_timing.py report $OUT/$n.full.csv | head -3
  echo "=== $n quarter"; cat $OUT/$n.quarter.txt; python tools/wg_timing.py report $OUT/$n.quarter.csv | head -3
done
rm -f $OUT/*.bin $OUT/*.csv
