#!/usr/bin/env python3
"""tools/gpu_counters.sh <dir> -> traffic.json, valu.json (k_compose_wg at the bench's C2 launch) and, when the matmul passes
are there, valu_matmul.json (k_scal_matmul_wnaf at 256^3): everything bench.py's roofline objects quote, recomputable by hand
from these files and kernel_stats.csv.

traffic:  FETCH_SIZE / WRITE_SIZE of separate --pmc passes, corrected with factors calibrated on tools/traffic_calib.hip (a
          record-copy kernel with the product's access pattern and a known byte count): MI355X_MICROARCH.md's rule for access
          widths other than 16 B per lane.
valu:     SQ_INSTS_VALU (wave-instructions), SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_* (QUAD-cycles per the guide),
          SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE (summed over the 8 XCDs); the clock measured INSIDE the kernel by the diagnostic
          build (tools/wg_timing.hip: delta s_memtime / delta s_memrealtime, median over workgroups after >= 2 s of load);
          the mix-weighted issue cost (tools/issue_weights.py)."""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_code_hash  # noqa: E402


def counters(sub, kernel):
    """{counter: (mean over dispatches, dispatches)} of one pass"""
    acc = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def counters_per_product(sub, kernel, marker="k_pow_table"):
    """{counter: (sum over the dispatches of `kernel` / number of products, dispatches)}: the tree form of the matrix product
    launches k_tree_level once per level and row chunk, so a product's figure is a SUM; products are counted by the
    dispatches of `marker` (one per product) in the same pass"""
    acc, marks = {}, {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if marker in r.get("Kernel_Name", ""):
                    marks[r["Counter_Name"]] = marks.get(r["Counter_Name"], 0) + 1
    return {k: (sum(v) / max(1, marks.get(k, 1)), len(v)) for k, v in acc.items()}


def kernel_total_ns_per_product(kernel, sub, marker="k_pow_table"):
    tot, calls, prods = None, 0, 1
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Name"]:
                    tot, calls = float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]), int(r["Calls"])
                if marker in r["Name"]:
                    prods = int(r["Calls"])
    return (tot / prods if tot is not None else None), calls


def kernel_avg_ns(kernel, sub="stats"):
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Name"]:
                    return float(r["AverageNs"]), int(r["Calls"])
    return None, 0


def load_json(name):
    try:
        with open(os.path.join(out, name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


N = 32768
REC = 672
code = kernel_code_hash()

# ---------------------------------------------------------------- traffic.json
known = {"FETCH_SIZE": 2 * N * REC, "WRITE_SIZE": N * REC}
res = {"records_per_launch": N, "counter_unit": "KiB as reported by rocprofv3 (x1024 -> bytes)", "kernel_code_hash": code}
total = 0.0
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = counters("calib_" + cn, "k_copy2").get(cn)
    ker = counters("bench_" + cn, "k_compose_wg").get(cn)
    if not cal or not ker:
        res[cn] = None
        continue
    cal_b, ker_b = cal[0] * 1024.0, ker[0] * 1024.0
    factor = known[cn] / cal_b
    res[cn] = {"calibration_counted_bytes": cal_b, "calibration_known_bytes": known[cn], "factor": round(factor, 4),
               "k_compose_wg_counted_bytes": ker_b, "k_compose_wg_corrected_bytes": round(ker_b * factor), "dispatches": [cal[1], ker[1]]}
    total += ker_b * factor
    mmb = 0.0
    for kn in ("k_tree_level", "k_scal_matmul_wnaf", "k_pow_table"):          # per PRODUCT: every heavy kernel of the call
        mm = counters_per_product("matmul_" + cn, kn).get(cn)
        if mm:
            res[cn][kn + "_corrected_bytes_per_product"] = round(mm[0] * 1024.0 * factor)
            mmb += mm[0] * 1024.0 * factor
    if mmb:
        res[cn]["matmul_corrected_bytes_per_product"] = round(mmb)
if res.get("FETCH_SIZE") and res.get("WRITE_SIZE"):
    res["traffic_bytes_per_launch"] = round(total)
    res["algorithmic_record_bytes_per_launch"] = 3 * N * REC
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)


# ---------------------------------------------------------------- valu.json
def valu_object(passes, kernel, stats_sub, weights_file, clock, extra, per_product=False):
    c = {}
    for p in passes:
        c.update(counters_per_product(p, kernel) if per_product else counters(p, kernel))
    if "SQ_INSTS_VALU" not in c:
        return None
    g = lambda k: c[k][0] if k in c else None
    avg_ns, calls = kernel_total_ns_per_product(kernel, stats_sub) if per_product else kernel_avg_ns(kernel, stats_sub)
    gui = g("GRBM_GUI_ACTIVE")
    v = dict(extra)
    v.update({"kernel": kernel, "kernel_code_hash": code, "valu_wave_insts_per_launch": round(g("SQ_INSTS_VALU")),
              "salu_insts_per_launch": round(g("SQ_INSTS_SALU")) if g("SQ_INSTS_SALU") else None,
              "waves_per_launch": round(g("SQ_WAVES")) if g("SQ_WAVES") else None,
              "dispatches": c["SQ_INSTS_VALU"][1]})
    if g("SQ_WAVES"):
        v["valu_per_wave"] = round(g("SQ_INSTS_VALU") / g("SQ_WAVES"), 1)
    for name, key in (("SQ_ACTIVE_INST_VALU", "sq_active_inst_valu_quadcycles"), ("SQ_WAVE_CYCLES", "sq_wave_cycles_quadcycles"),
                      ("SQ_BUSY_CYCLES", "sq_busy_cycles"), ("SQ_WAIT_ANY", "sq_wait_any_quadcycles"),
                      ("SQ_WAIT_INST_ANY", "sq_wait_inst_any_quadcycles"), ("SQ_ACTIVE_INST_ANY", "sq_active_inst_any_quadcycles"),
                      ("SQ_ACTIVE_INST_SCA", "sq_active_inst_sca_quadcycles"), ("SQ_ACTIVE_INST_LDS", "sq_active_inst_lds_quadcycles"),
                      ("SQ_WAIT_INST_LDS", "sq_wait_inst_lds_quadcycles"), ("SQ_INSTS_LDS", "lds_insts_per_launch")):
        if g(name) is not None:
            v[key] = round(g(name))
    if gui:
        v["grbm_gui_active_sum_over_8_xcds"] = round(gui)
        cycles = gui / 8.0                                   # shader cycles the launch took (per XCD)
        if g("SQ_ACTIVE_INST_VALU"):
            # quad-cycles x 4 = SIMD-cycles spent issuing VALU, over the 1024 SIMDs' cycles of the launch
            v["busy_frac_counters"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * cycles), 4)
            v["busy_frac_formula"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
        if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
            v["valu_share_of_wave_cycles"] = round(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 4)
    if avg_ns:
        v["rocprofv3_average_ns"] = round(avg_ns, 1)
        v["rocprofv3_calls"] = calls
        if gui:
            v["clock_ghz_grbm"] = round(gui / 8.0 / avg_ns, 4)     # reads high on dispatches shorter than ~0.3 ms (the guide)
    if clock:
        v["clock_ghz_in_kernel"] = round(clock["clock_ghz_in_kernel"], 4)
        v["clock_source"] = ("tools/wg_timing.hip (COFHE_WG_TIMING build of k_compose_wg): delta s_memtime / delta s_memrealtime x 100 MHz, "
                             "median over %d workgroups of the last launch after %.0f s of back-to-back launches (min %.3f, max %.3f)"
                             % (clock["workgroups"], clock["load_seconds"], clock["min"], clock["max"]))
    w = load_json(weights_file)
    if w:
        v["issue_cycles_per_valu_inst"] = w["issue_cycles_per_valu_inst"]
        v["issue_weights_source"] = ("tools/issue_weights.py: static VALU mix of %s (%d instructions) x the W=4 issue costs of %s; per-class table in %s"
                                     % (kernel, w["static_valu_instructions"], w["bench_table"], weights_file))
    v["source"] = "tools/gpu_counters.sh: rocprofv3 --pmc passes of `python3 bench.py ...` (program directly after --), own passes for the TCC counters"
    return v


clock = load_json("clock.json")
v = valu_object(["bench_SQA", "bench_SQB"], "k_compose_wg", "stats", "issue_weights_compose.json", clock, {"records_per_launch": N})
if v:
    with open(os.path.join(out, "valu.json"), "w") as fh:
        json.dump(v, fh, indent=1)
vm = valu_object(["matmul_SQA", "matmul_SQB"], "k_tree_level", "matmul_stats", "issue_weights_matmul.json", clock,
                 {"records_per_launch": 256 * 256 * 2, "shape": [256, 256, 256],
                  "per": "ONE 256^3 product: the figures are sums over the k_tree_level launches of a product (levels x row chunks)"}, per_product=True)
if vm:
    vm["clock_note"] = "the in-kernel clock was measured on k_compose_wg (same composition code, same occupancy)"
    if res.get("FETCH_SIZE") and res.get("WRITE_SIZE") and "matmul_corrected_bytes_per_product" in res["FETCH_SIZE"] and \
            "matmul_corrected_bytes_per_product" in res["WRITE_SIZE"]:
        vm["traffic_bytes_per_launch"] = res["FETCH_SIZE"]["matmul_corrected_bytes_per_product"] + res["WRITE_SIZE"]["matmul_corrected_bytes_per_product"]
        vm["traffic_note"] = "HBM-side bytes of one product: k_tree_level + k_scal_matmul_wnaf (Horner) + k_pow_table"
    with open(os.path.join(out, "valu_matmul.json"), "w") as fh:
        json.dump(vm, fh, indent=1)
print(json.dumps({"traffic": res.get("traffic_bytes_per_launch"), "valu": v, "valu_matmul": vm}, indent=1))
