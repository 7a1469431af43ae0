# through gpurun: every build/libcofhe_hip_*.so (tools/build_variant.sh) -- first a parity subset of the GPU suite against the
# oracle (COFHE_TEST_LIB), then interleaved rounds of the bench in one call (one device): value, ms per step, ms per launch
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS=${LIBS:-"cofhe_amd/libcofhe_hip.so $(ls build/libcofhe_hip_*.so 2>/dev/null)"}
SUBSET='golden or lopsided or add_16x16 or 16x16_config or compose_with_powers or fixed_base_golden or decrypt_golden or pow_fixed_base or every_window'
if [ "$SKIP_PARITY" != "1" ]; then
for f in $LIBS; do
  [ -f "$f" ] || continue
  echo -n "== parity $f  "
  COFHE_TEST_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "$SUBSET" > gpurun_out/parity_$(basename $f).log 2>&1 || true
  tail -1 gpurun_out/parity_$(basename $f).log
done
fi
for round in 1 2 3; do
for f in $LIBS; do
  [ -f "$f" ] || continue
  echo -n "== round $round $f  "
  timeout -k 10 300 python bench.py --lib $GRAFT_REPO_ROOT/$f --steps 20 --warmup 3 --no-cpu-baseline --no-family2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
done
