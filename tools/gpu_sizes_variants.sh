# through gpurun: launch time of the compose kernel at several tensor sizes for every variant library (per-workgroup latency
# at 1 workgroup per CU vs the full 4 per CU): ROWSxCOLS list in SIZES
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS=${LIBS:-"$(ls build/libcofhe_hip_*.so 2>/dev/null)"}
SIZES=${SIZES:-64x64 90x91 128x128}
for sz in $SIZES; do
r=${sz%x*}; c=${sz#*x}
for round in 1 2; do
for f in $LIBS; do
  echo -n "== $sz round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --rows $r --cols $c --steps 20 --warmup 3 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['device_status'])"
done
done
done
