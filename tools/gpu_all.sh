set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || (tail -20 gpurun_out/bench.err; exit 1)
cat gpurun_out/bench.json
cd cofhe_amd/host && timeout -k 10 300 ./local_bench ciphertext_matadd 64 64 | tee $GRAFT_REPO_ROOT/gpurun_out/local_bench.txt
