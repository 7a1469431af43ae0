# through gpurun: the reference harness's own shapes (benchmarks/local.cpp defaults) through the C++ drop-in API
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/local_bench
B=cofhe_amd/host/local_bench
[ -x $B ] || python -c "import __graft_entry__ as g; g.build()"
for run in "ciphertext_matadd 64 64" "encrypt_decrypt 64 64" "scal_matmul 8 64 64" "ciphertext_matadd 128 128" "threshold 16 16 2 3"; do
  f=gpurun_out/local_bench/$(echo $run | tr ' ' '_').txt
  timeout -k 10 300 $B $run > $f 2>&1 || (cat $f; exit 1)
  echo "== $run"; cat $f | cut -c1-220
done
