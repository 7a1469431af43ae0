# round 4, final record on the shipped hash: the whole GPU suite, the default bench line, the matrix-product bench line,
# per-operation timings, the C++ harness shapes, the wide-layout timings -> gpurun_out/r04_final/ (copy to profiles/r04_final/)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --capture=sys > $O/gpu_tests.log 2>&1 || (tail -40 $O/gpu_tests.log; exit 1)
tail -2 $O/gpu_tests.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || (tail -20 $O/bench.err; exit 1)
cut -c1-400 $O/bench.json
timeout -k 10 600 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 2 --warmup 1 > $O/bench_scal_matmul_256.json 2> $O/bench_mm.err || (tail -20 $O/bench_mm.err; exit 1)
cut -c1-300 $O/bench_scal_matmul_256.json
timeout -k 10 900 python tools/bench_ops.py > $O/ops.jsonl 2> $O/ops.err || (tail -5 $O/ops.err; exit 1)
cut -c1-160 $O/ops.jsonl
timeout -k 10 600 python tools/gpu_wide_time.py > $O/wide_time.txt 2>&1 || (tail -5 $O/wide_time.txt; exit 1)
bash tools/gpu_wide_timing.sh 64 > /dev/null && cp gpurun_out/wt/wide_timing.txt $O/wide_phases.txt
bash tools/gpu_local_bench.sh > $O/local_bench.log 2>&1 || (tail -5 $O/local_bench.log; exit 1)
cp -r gpurun_out/local_bench $O/
echo final record done
