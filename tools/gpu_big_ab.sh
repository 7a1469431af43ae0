set -e
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for f in build/libcofhe_hip_ship.so build/libcofhe_hip_wps3.so; do
  echo -n "== 1024^2 round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --rows 1024 --cols 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['add_ciphertext_records']['ms_per_add'])"
  echo -n "== 128^2  round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --steps 20 --warmup 3 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['add_ciphertext_records']['ms_per_add'])"
done
done
