# through gpurun: 1024x1024 matadd (steady state: 64 residency rounds) for the builds in LIBS, two interleaved rounds
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
for f in $LIBS; do
  echo -n "== 1024x1024 round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --rows 1024 --cols 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
done
