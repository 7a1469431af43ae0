# round 4, call C: sizes x variants, then decrypt-family tests and per-operation timings of the in-tree library
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SIZES="128x128 256x256 1024x1024" LIBS="build/libcofhe_hip_cur.so build/libcofhe_hip_e1full.so build/libcofhe_hip_div2.so" bash tools/gpu_sizes_variants.sh 2>&1 | tee gpurun_out/r4_sizes_variants.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --capture=sys -k "decrypt or threshold or shared_exponent or soak or encrypt" 2>&1 | tail -3
OPS_SKIP=big,k256 timeout -k 10 600 python tools/bench_ops.py > gpurun_out/r4_ops_c.jsonl 2> gpurun_out/r4_ops_c.err || (tail -5 gpurun_out/r4_ops_c.err; exit 1)
grep -E "decrypt|scal_ciphertext_tensors 1-D|encrypt" gpurun_out/r4_ops_c.jsonl | cut -c1-200
