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
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_code_hash  # noqa: E402
res["kernel_code_hash"] = kernel_code_hash()
res["traffic_bytes_per_launch"] = round(total)
res["algorithmic_record_bytes_per_launch"] = 3 * N * REC
print(json.dumps(res, indent=1))

# VALU wave-instructions per launch (roofline_valu of bench.py) -> valu.json beside traffic.json
valu, nv = mean_counter("bench_VALU", "k_compose_wg", "SQ_INSTS_VALU")
if valu is not None:
    salu, _ = mean_counter("bench_VALU", "k_compose_wg", "SQ_INSTS_SALU")
    waves, _ = mean_counter("bench_VALU", "k_compose_wg", "SQ_WAVES")
    gui, _ = mean_counter("bench_VALU", "k_compose_wg", "GRBM_GUI_ACTIVE")
    v = {"records_per_launch": N, "kernel_code_hash": res["kernel_code_hash"], "valu_wave_insts_per_launch": round(valu),
         "salu_insts_per_launch": round(salu) if salu else None, "waves_per_launch": round(waves) if waves else None,
         "valu_per_wave": round(valu / waves, 1) if waves else None,
         "grbm_gui_active_sum_over_8_xcds": round(gui) if gui else None, "clock_ghz": 2.4, "dispatches": nv,
         "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 1"}
    with open(os.path.join(out, "valu.json"), "w") as fh:
        json.dump(v, fh, indent=1)
