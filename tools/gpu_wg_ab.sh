# usage (through gpurun): bash tools/gpu_wg_ab.sh A B -> phase timing and rare-route flags of build/wg_timing_A vs build/wg_timing_B
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wgab
timeout -k 10 300 python3 tools/wg_timing.py gen gpurun_out/wgab > /dev/null
for v in "$@"; do
  echo "== $v"
  timeout -k 10 300 build/$v gpurun_out/wgab/delta.bin gpurun_out/wgab/a.bin gpurun_out/wgab/b.bin 1 > gpurun_out/wgab/$v.csv 2> gpurun_out/wgab/$v.txt
  grep -A14 "phase means" gpurun_out/wgab/$v.txt
  python3 - gpurun_out/wgab/$v.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
from collections import Counter
c = Counter(int(r["flags"]) for r in rows)
print("flags histogram (1 common factor, 2 general route, 4 long-division step, 8 add-back):", dict(sorted(c.items())))
print("rounds1 mean %.2f rounds2 mean %.2f" % (sum(float(r["rounds1"]) for r in rows) / len(rows), sum(float(r["rounds2"]) for r in rows) / len(rows)))
PY
done
