# round 4, call D: the wide layout -- parity tests, latency
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --capture=sys -k "wide_layout or ladder_forms or decrypt or threshold" > gpurun_out/r4_tests_d.log 2>&1 || (tail -40 gpurun_out/r4_tests_d.log; exit 1)
tail -2 gpurun_out/r4_tests_d.log
timeout -k 10 600 python tools/gpu_wide_time.py 2>&1 | tee gpurun_out/r4_wide_time.txt
