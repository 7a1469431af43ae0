# through gpurun: full GPU suite, bench x2, ops bench of the in-tree build
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_h.log 2>&1 || (tail -60 gpurun_out/r3_tests_h.log; exit 1)
tail -3 gpurun_out/r3_tests_h.log
for r in 1 2; do
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_h$r.json 2> gpurun_out/bench_h.err || (tail -20 gpurun_out/bench_h.err; exit 1)
python -c "import json; d=json.load(open('gpurun_out/bench_h$r.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['input_family_ii']['launch_ms'], d['add_ciphertext_records']['ms_per_add'], d['device_status'])"
done
timeout -k 10 900 python tools/bench_ops.py > gpurun_out/ops_i.jsonl 2> gpurun_out/ops_i.err || (tail -20 gpurun_out/ops_i.err; exit 1)
cut -c1-130 gpurun_out/ops_i.jsonl
