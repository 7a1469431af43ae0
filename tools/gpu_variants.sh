set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for f in build/libcofhe_hip_*.so; do
  echo "== $f"
  COFHE_HIP_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
