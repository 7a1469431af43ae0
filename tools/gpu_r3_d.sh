# through gpurun: matrix-product parity subset + C3 bench + per-workgroup phase CSV (full and quarter grid)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build/wgt_inputs
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "matmul or scal or c4 or window or segment" > gpurun_out/r3_tests_d.log 2>&1 || (tail -40 gpurun_out/r3_tests_d.log; exit 1)
tail -2 gpurun_out/r3_tests_d.log
timeout -k 10 600 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mm_d.json 2> gpurun_out/bench_mm_d.err || (tail -20 gpurun_out/bench_mm_d.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_mm_d.json')); print(d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'], d['device_status'])"
timeout -k 10 300 python tools/wg_timing.py gen build/wgt_inputs
OUT=gpurun_out/wg_d; mkdir -p $OUT
head -c $((8192*672)) build/wgt_inputs/a.bin > $OUT/a4.bin
head -c $((8192*672)) build/wgt_inputs/b.bin > $OUT/b4.bin
timeout -k 10 120 build/wg_timing_head build/wgt_inputs/delta.bin build/wgt_inputs/a.bin build/wgt_inputs/b.bin 0.3 > $OUT/full.csv 2> $OUT/full.txt
timeout -k 10 120 build/wg_timing_head build/wgt_inputs/delta.bin $OUT/a4.bin $OUT/b4.bin 0.3 > $OUT/quarter.csv 2> $OUT/quarter.txt
rm -f $OUT/*.bin
