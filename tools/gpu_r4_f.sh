# round 4, call F: matrix-product bench line and timings after the level buffers were made cacheable
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload scal_matmul --rows 256 --cols 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_mm_f.json 2> gpurun_out/r4_bench_mm_f.err || (tail -20 gpurun_out/r4_bench_mm_f.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/r4_bench_mm_f.json')); print(d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'])"
timeout -k 10 500 python tools/gpu_tree_time.py 2>&1 | tee gpurun_out/r4_tree_time_f.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --capture=sys -k "scal_matmul or 256_sampled" 2>&1 | tail -2
