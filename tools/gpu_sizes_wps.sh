# through gpurun: launch time of the compose kernel and of the folded ciphertext addition at tensor sizes around the residency
# limits of 2, 3 and 4 workgroups per CU, for every variant library in LIBS (tools/build_variant.sh NAME -DCOFHE_WPS=k)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SIZES=${SIZES:-32x32 64x64 90x91 100x100 110x111 128x128}
for sz in $SIZES; do
r=${sz%x*}; c=${sz#*x}
for f in $LIBS; do
  echo -n "== $sz $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --rows $r --cols $c --steps 20 --warmup 3 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('launch_ms', d['roofline']['launch_ms'], 'folded_ms', d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
done
