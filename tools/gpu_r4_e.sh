# round 4, call E: the whole GPU suite, the default bench line, the matrix-product bench line, per-operation timings, local_bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --capture=sys > gpurun_out/r4_tests_e.log 2>&1 || (tail -40 gpurun_out/r4_tests_e.log; exit 1)
tail -2 gpurun_out/r4_tests_e.log
timeout -k 10 600 python bench.py > gpurun_out/r4_bench_e.json 2> gpurun_out/r4_bench_e.err || (tail -20 gpurun_out/r4_bench_e.err; exit 1)
cut -c1-700 gpurun_out/r4_bench_e.json
timeout -k 10 600 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 2 --warmup 1 > gpurun_out/r4_bench_mm_e.json 2> gpurun_out/r4_bench_mm_e.err || (tail -20 gpurun_out/r4_bench_mm_e.err; exit 1)
cut -c1-900 gpurun_out/r4_bench_mm_e.json
timeout -k 10 900 python tools/bench_ops.py > gpurun_out/r4_ops_e.jsonl 2> gpurun_out/r4_ops_e.err || (tail -5 gpurun_out/r4_ops_e.err; exit 1)
cut -c1-160 gpurun_out/r4_ops_e.jsonl
