# through gpurun: GPU suite + bench + per-workgroup stamps of the head build (round 3, session 2)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build/wgt_inputs
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_c.log 2>&1 || (tail -40 gpurun_out/r3_tests_c.log; exit 1)
tail -3 gpurun_out/r3_tests_c.log
for r in 1 2; do
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c$r.json 2> gpurun_out/bench_c.err || (tail -20 gpurun_out/bench_c.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_c$r.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['input_family_ii']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
timeout -k 10 300 python tools/wg_timing.py gen build/wgt_inputs
W=build/wg_timing_head bash tools/gpu_wg_spread.sh 2>&1 | tail -12
grep -A14 "phase means" gpurun_out/wg_spread/run1.txt
