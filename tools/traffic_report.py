"""HBM-side traffic of k_compose_wg from rocprofv3 FETCH_SIZE / WRITE_SIZE, corrected with the factors
measured on tools/traffic_calib.hip (same access pattern, known byte count)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def mean_counter(sub, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


N = 32768
REC = 672
known = {"FETCH_SIZE": 2 * N * REC, "WRITE_SIZE": N * REC}
res = {"records_per_launch": N, "counter_unit": "KiB as reported by rocprofv3 (x1024 -> bytes)"}
total = 0.0
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    cal, n1 = mean_counter("calib_" + cn, "k_copy2", cn)
    ker, n2 = mean_counter("bench_" + cn, "k_compose_wg", cn)
    if cal is None or ker is None:
        res[cn] = None
        continue
    cal_b, ker_b = cal * 1024.0, ker * 1024.0
    factor = known[cn] / cal_b
    res[cn] = {"calibration_counted_bytes": cal_b, "calibration_known_bytes": known[cn], "factor": round(factor, 4),
               "k_compose_wg_counted_bytes": ker_b, "k_compose_wg_corrected_bytes": round(ker_b * factor),
               "dispatches": [n1, n2]}
    total += ker_b * factor
res["traffic_bytes_per_launch"] = round(total)
res["algorithmic_record_bytes_per_launch"] = 3 * N * REC
print(json.dumps(res, indent=1))
