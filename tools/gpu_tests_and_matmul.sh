set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 600 python tools/gpu_matmul_time.py
