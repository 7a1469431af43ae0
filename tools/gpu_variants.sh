set -e
cd $GRAFT_REPO_ROOT
for f in cofhe_amd/libcofhe_hip.so build/libcofhe_hip_*.so; do
  echo "== $f"
  COFHE_HIP_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
