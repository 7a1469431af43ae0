# round 4, call B: matrix-product tests (tree and chains), tree timing, then the compose variants
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --capture=sys -k "scal_matmul or golden or c4_row or 256_sampled" > gpurun_out/r4_tests_b.log 2>&1 || (tail -40 gpurun_out/r4_tests_b.log; exit 1)
tail -2 gpurun_out/r4_tests_b.log
timeout -k 10 500 python tools/gpu_tree_time.py 2>&1 | tee gpurun_out/r4_tree_time.txt
SKIP_PARITY=1 LIBS="build/libcofhe_hip_cur.so build/libcofhe_hip_g64.so build/libcofhe_hip_cap7.so build/libcofhe_hip_cap9.so" bash tools/gpu_variants.sh 2>&1 | tee gpurun_out/r4_variants_c.txt
