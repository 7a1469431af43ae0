# through gpurun: full GPU suite + bench (x3) of the current build
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_f.log 2>&1 || (tail -40 gpurun_out/r3_tests_f.log; exit 1)
tail -3 gpurun_out/r3_tests_f.log
for r in 1 2 3; do
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_f$r.json 2> gpurun_out/bench_f.err || (tail -20 gpurun_out/bench_f.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_f$r.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['input_family_ii']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
