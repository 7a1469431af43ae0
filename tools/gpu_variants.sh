set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# interleaved rounds of every variant in one call (one device): median and min of the per-launch time
for round in 1 2 3; do
for f in cofhe_amd/libcofhe_hip.so build/libcofhe_hip_*.so; do
  [ -f "$f" ] || continue
  echo -n "== round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --steps 20 --warmup 3 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
done
