# usage: bash tools/gpu_traffic.sh <tag>  (through gpurun): HBM-side counters of k_compose_wg, calibrated on a
# record-copy kernel with the same access pattern (tools/traffic_calib.hip); separate --pmc passes
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $OUT/traffic_calib tools/traffic_calib.hip
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $C -d $OUT/calib_$C -o c -- $OUT/traffic_calib 32768 10 > $OUT/calib_$C.json 2> $OUT/calib_$C.err || (tail -5 $OUT/calib_$C.err; exit 1)
  timeout -k 10 600 rocprofv3 --output-format csv --pmc $C -d $OUT/bench_$C -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-family2 > $OUT/bench_$C.json 2> $OUT/bench_$C.err || (tail -5 $OUT/bench_$C.err; exit 1)
done
# VALU wave-instructions per launch and the clock held (GRBM_GUI_ACTIVE is summed over the 8 XCDs): own pass
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/bench_VALU -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-family2 > $OUT/bench_VALU.json 2> $OUT/bench_VALU.err || (tail -5 $OUT/bench_VALU.err; exit 1)
# per-kernel durations of the same command (no counters)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-family2 > $OUT/bench_stats.json 2> $OUT/bench_stats.err || (tail -5 $OUT/bench_stats.err; exit 1)
cd $GRAFT_REPO_ROOT
python3 $GRAFT_REPO_ROOT/tools/traffic_report.py $OUT | tee $OUT/traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -f $OUT/traffic_calib
find $OUT -size +8M -delete
