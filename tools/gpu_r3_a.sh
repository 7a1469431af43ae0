set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests.log 2>&1 || (tail -40 gpurun_out/r3_tests.log; exit 1)
tail -3 gpurun_out/r3_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err || (tail -20 gpurun_out/bench.err; exit 1)
cat gpurun_out/bench.json
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --force-comm --no-cpu-baseline --no-family2 > gpurun_out/bench_comm.json 2> gpurun_out/bench_comm.err || (tail -20 gpurun_out/bench_comm.err; exit 1)
cat gpurun_out/bench_comm.json
timeout -k 10 900 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 2 --warmup 1 > gpurun_out/bench_matmul.json 2> gpurun_out/bench_matmul.err || (tail -20 gpurun_out/bench_matmul.err; exit 1)
cat gpurun_out/bench_matmul.json
