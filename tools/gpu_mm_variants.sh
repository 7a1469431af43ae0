# through gpurun: scal_matmul (ROWS x 256 . 256 x 256, harness exponents) for the builds in LIBS, two interleaved rounds
set -e
cd $GRAFT_REPO_ROOT
R=${ROWS:-64}
for round in 1 2; do
for f in $LIBS; do
  echo -n "== matmul $R round $round $f  "
  timeout -k 10 600 python bench.py --lib $GRAFT_REPO_ROOT/$f --workload scal_matmul --rows $R --cols 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'], d['device_status'])"
done
done
