# round 4, call A: the whole GPU suite on the in-tree library, then interleaved bench rounds of the variants in build/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --capture=sys > gpurun_out/r4_tests_a.log 2>&1 || (tail -40 gpurun_out/r4_tests_a.log; exit 1)
tail -2 gpurun_out/r4_tests_a.log
SKIP_PARITY=1 bash tools/gpu_variants.sh 2>&1 | tee gpurun_out/r4_variants_b.txt
