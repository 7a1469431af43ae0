# through gpurun: matrix-product parity subset + C3 bench + ops bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "matmul or scal or c4 or window or segment" > gpurun_out/r3_tests_e.log 2>&1 || (tail -40 gpurun_out/r3_tests_e.log; exit 1)
tail -2 gpurun_out/r3_tests_e.log
timeout -k 10 600 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mm_e.json 2> gpurun_out/bench_mm_e.err || (tail -20 gpurun_out/bench_mm_e.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_mm_e.json')); print(d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'], d['device_status'])"
timeout -k 10 900 python tools/bench_ops.py > gpurun_out/ops_e.jsonl 2> gpurun_out/ops_e.err || (tail -20 gpurun_out/ops_e.err; exit 1)
cat gpurun_out/ops_e.jsonl | cut -c1-400
