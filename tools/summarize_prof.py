"""Condenses rocprofv3 output directories (kernel stats + PMC passes) into a short text."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, "**", pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

print("== per-dispatch resources (kernel trace) ==")
seen = set()
for f, r in rows("*kernel_trace.csv"):
    n = r.get("Kernel_Name", "")
    if n in seen:
        continue
    seen.add(n)
    print({k: r[k] for k in r if k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")})

print("== PMC counters: per-kernel mean over dispatches ==")
acc = defaultdict(lambda: defaultdict(list))
for f, r in rows("*counter_collection.csv"):
    acc[r.get("Kernel_Name", "?")][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
for kname, cs in acc.items():
    if "compose" not in kname and "pow" not in kname and "matmul" not in kname:
        continue
    print(kname)
    for cn, vals in sorted(cs.items()):
        print("   %-24s mean %.6g  (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
