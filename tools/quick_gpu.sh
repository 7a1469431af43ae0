set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ct-ops/s', d['value'], 'ms/step', d['ms_per_step'], 'launch_ms', d['roofline']['launch_ms'])"
