set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_b.log 2>&1 || (tail -40 gpurun_out/r3_tests_b.log; exit 1)
tail -3 gpurun_out/r3_tests_b.log
for r in 1 2 3; do
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b$r.json 2> gpurun_out/bench_b.err || (tail -20 gpurun_out/bench_b.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_b$r.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['input_family_ii']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
W=build/wg_timing_r3c bash tools/gpu_wg_spread.sh 2>&1 | tail -12
grep -A14 "phase means" gpurun_out/wg_spread/run1.txt
