# through gpurun: full GPU suite, then interleaved bench of the variants in LIBS
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_g.log 2>&1 || (tail -60 gpurun_out/r3_tests_g.log; exit 1)
tail -3 gpurun_out/r3_tests_g.log
SKIP_PARITY=1 bash tools/gpu_variants.sh
