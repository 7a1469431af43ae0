#!/usr/bin/env python3
"""Instruction-mix-weighted VALU issue cost of a kernel: the STATIC mix of its gfx950 code (llvm-objdump of the code object
inside cofhe_amd/csrc/obj/<part>.o) weighted with the measured issue costs of tools/inst_bench.hip (shader cycles per
wave-instruction per SIMD at 4 resident waves per SIMD -- the occupancy the kernels run at), read from a committed run
of that tool.  Output: JSON with the weighted mean, the per-class counts and which instructions fell back to a class
default.  Used by tools/counters_report.py for roofline_valu.peak = SIMDs x clock / weighted cycles."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def bench_table(path):
    """W=4 column of an inst_bench run: {label: cycles}"""
    tab = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            tab[m.group(1).strip()] = float(m.group(4))
    return tab


def cost_of(op, tab):
    """(cycles, source label) for one VALU opcode (suffixes stripped)"""
    def t(label, default):
        return (tab.get(label, default), label if label in tab else "default:" + label)
    full_rate = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32",
                 "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_u32", "v_min_u32", "v_accvgpr_read_b32",
                 "v_accvgpr_write_b32", "v_fmac_f32", "v_mac_f32", "v_max_f32", "v_min_f32")
    if op in full_rate:
        return t("[blk] v_add_u32" if op != "v_mov_b32" else "v_mov_b32", 1.98)
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return t("v_mad_u64_u32", 3.98)
    if op.startswith(("v_add_co", "v_sub_co", "v_subrev_co")):
        return t("v_add_co_u32", 3.98)
    if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co")):
        return t("v_addc_co_u32 (chain)", 3.98)
    if op.startswith("v_cmp"):
        return t("v_cmp_lt_u32", 3.99)
    if op.startswith("v_cndmask"):
        return t("[blk] v_cndmask s[20:21]", 3.29)      # the e64 / SGPR-mask cost; the e32-after-SALU anomaly is not priced in
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log")):
        return t("v_rcp_f32", 5.58)
    if op.startswith("v_cvt_f32_u32"):
        return t("v_cvt_f32_u32", 3.20)
    if op.startswith("v_cvt_u32_f32"):
        return t("v_cvt_u32_f32", 2.89)
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi"):
        return t("v_mul_lo_u32", 3.27)
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64")):
        return t("v_fma_f64", 3.27)
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return t("[blk] v_readfirstlane", 3.11)
    if op.startswith("v_lshl_add_u64"):
        return t("v_lshl_add_u64", 3.27)
    if op.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")):
        return t("v_lshrrev_b64", 3.27)
    return t("v_lshlrev_b32", 3.20)                      # shifts, bfe, perm, alignbit, add3, and_or, ffbh, dpp moves: one class


def static_mix(obj, kernel):
    tmp = tempfile.mkdtemp()
    o = os.path.join(tmp, "x.o")
    subprocess.check_call(["cp", obj, o])
    subprocess.check_call([BIN + "/llvm-objdump", "--offloading", o], cwd=tmp, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    co = os.path.join(tmp, co)
    syms = subprocess.check_output([BIN + "/llvm-readelf", "-sW", co], text=True)
    sym = [ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln and kernel in ln][0]
    dis = subprocess.check_output([BIN + "/llvm-objdump", "-d", "--disassemble-symbols=" + sym, co], text=True)
    mix = {}
    for ln in dis.splitlines():
        parts = ln.split()
        if len(parts) > 1 and re.match(r"^(v_|s_|ds_|global_|scratch_|buffer_)", parts[0]):
            op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0])
            dpp = parts[0].endswith("_dpp")
            key = op + ("|dpp" if dpp else "")
            mix[key] = mix.get(key, 0) + 1
    subprocess.call(["rm", "-rf", tmp])
    return mix


def main():
    part, kernel, bench_file = sys.argv[1], sys.argv[2], sys.argv[3]
    tab = bench_table(bench_file)
    mix = static_mix(os.path.join(ROOT, "cofhe_amd", "csrc", "obj", part + ".o"), kernel)
    total = cyc = 0.0
    by_label, fallbacks = {}, {}
    for key, n in mix.items():
        op = key.split("|")[0]
        if not op.startswith("v_"):
            continue
        c, label = cost_of(op, tab)
        if key.endswith("|dpp"):
            c, label = tab.get("v_mov_b32_dpp row_shr", 3.27), "v_mov_b32_dpp row_shr"
        total += n
        cyc += n * c
        e = by_label.setdefault(label, [0, c])
        e[0] += n
        if label == "v_lshlrev_b32" and op != "v_lshlrev_b32":
            fallbacks[op] = fallbacks.get(op, 0) + n
    out = {"kernel": kernel, "object": "cofhe_amd/csrc/obj/%s.o" % part, "bench_table": os.path.relpath(bench_file, ROOT),
           "occupancy_column": "W=4 (waves resident per SIMD)", "static_valu_instructions": int(total),
           "issue_cycles_per_valu_inst": round(cyc / total, 4),
           "classes": {k: {"count": v[0], "cycles": v[1]} for k, v in sorted(by_label.items(), key=lambda kv: -kv[1][0])},
           "priced_as_generic_3.2_cycle_ops": dict(sorted(fallbacks.items(), key=lambda kv: -kv[1])[:20]),
           "non_valu": {k: v for k, v in sorted(mix.items(), key=lambda kv: -kv[1]) if not k.startswith("v_")}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
