# usage: bash tools/gpu_counters.sh <tag> [nomatmul]   (through gpurun) -> gpurun_out/counters_<tag>/{traffic,valu,valu_matmul}.json,
# kernel_stats.csv, clock.json, issue weights: every number bench.py's roofline objects quote.  Copy the directory's JSON / CSV
# files to profiles/rNN_<tag>/ to commit them.  Counters in their own rocprofv3 runs (--pmc only), the program directly after `--`.
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/counters_$TAG
mkdir -p $OUT
INST_BENCH=${INST_BENCH:-profiles/r02_microbench/inst_bench_run3.txt}
# static instruction mix x measured issue costs (no GPU needed, done first so a later failure keeps it)
python3 tools/issue_weights.py part0 k_compose_wg $INST_BENCH > $OUT/issue_weights_compose.json
python3 tools/issue_weights.py part2 k_tree_level $INST_BENCH > $OUT/issue_weights_matmul.json
bash tools/codeobj_report.sh > $OUT/codeobj_report.txt 2>&1 || true
# the two helper binaries are cross-compiled in the build container (tools/build_tools.sh -> build/, which travels with the
# snapshot); built here only when missing (minutes of GPU-box time)
if [ -x build/traffic_calib ]; then cp build/traffic_calib $OUT/traffic_calib; else hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $OUT/traffic_calib tools/traffic_calib.hip; fi
# in-kernel clock: diagnostic build with s_memtime / s_memrealtime stamps (no stamp executes in the product kernel)
if [ -x build/wg_timing ]; then cp build/wg_timing $OUT/wg_timing; else hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -o $OUT/wg_timing tools/wg_timing.hip cofhe_amd/csrc/wire.hip cofhe_amd/csrc/wide.hip; fi
timeout -k 10 300 python3 tools/wg_timing.py gen $OUT
timeout -k 10 300 $OUT/wg_timing $OUT/delta.bin $OUT/a.bin $OUT/b.bin 3 > $OUT/wg.csv 2> $OUT/wg_timing.txt || (tail -5 $OUT/wg_timing.txt; exit 1)
grep '^CLOCK_JSON' $OUT/wg_timing.txt | sed 's/^CLOCK_JSON //' > $OUT/clock.json
python3 tools/wg_timing.py report $OUT/wg.csv > $OUT/wg_report.txt || true
echo "clock: $(cat $OUT/clock.json)"
export TMPDIR=/tmp
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-family2"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $C -d $OUT/calib_$C -o c -- $OUT/traffic_calib 32768 10 > $OUT/calib_$C.json 2> $OUT/calib_$C.err || (tail -5 $OUT/calib_$C.err; exit 1)
  timeout -k 10 600 rocprofv3 --output-format csv --pmc $C -d $OUT/bench_$C -o b -- $B > $OUT/bench_$C.json 2> $OUT/bench_$C.err || (tail -5 $OUT/bench_$C.err; exit 1)
done
echo "tcc passes done"
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/bench_SQA -o b -- $B > $OUT/bench_SQA.json 2> $OUT/bench_SQA.err || (tail -5 $OUT/bench_SQA.err; exit 1)
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $OUT/bench_SQB -o b -- $B > $OUT/bench_SQB.json 2> $OUT/bench_SQB.err || (tail -5 $OUT/bench_SQB.err; exit 1)
echo "sq passes done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-family2 > $OUT/bench_stats.json 2> $OUT/bench_stats.err || (tail -5 $OUT/bench_stats.err; exit 1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
if [ "$2" != "nomatmul" ]; then
  M="python3 $GRAFT_REPO_ROOT/bench.py --workload scal_matmul --rows 256 --cols 256 --steps 1 --warmup 1 --no-cpu-baseline"
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/matmul_stats -o s -- $M > $OUT/matmul_stats.json 2> $OUT/matmul_stats.err || (tail -5 $OUT/matmul_stats.err; exit 1)
  cp $(find $OUT/matmul_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_matmul.csv
  echo "matmul stats done"
  timeout -k 10 900 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/matmul_SQA -o b -- $M > $OUT/matmul_SQA.json 2> $OUT/matmul_SQA.err || (tail -5 $OUT/matmul_SQA.err; exit 1)
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 900 rocprofv3 --output-format csv --pmc $C -d $OUT/matmul_$C -o b -- $M > $OUT/matmul_$C.json 2> $OUT/matmul_$C.err || (tail -5 $OUT/matmul_$C.err; exit 1)
  done
  echo "matmul passes done"
fi
cd $GRAFT_REPO_ROOT
python3 tools/counters_report.py $OUT > $OUT/report.json
rm -f $OUT/traffic_calib $OUT/wg_timing $OUT/a.bin $OUT/b.bin
find $OUT -size +8M -delete
cat $OUT/valu.json
