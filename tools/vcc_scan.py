#!/usr/bin/env python3
"""Scans a gfx950 .s for VOP2 (e32) v_cndmask_b32 that read a VCC last written by the SCALAR unit:
measured on MI355X (tools/inst_bench.hip) at ~12 cycles each instead of ~3 (the e64 encoding, or a VCC
written by a VALU compare, does not pay this)."""
import re, sys, collections
fn = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else None
cur = None
last = {}          # kernel -> 'S'/'V'/None
stats = collections.defaultdict(lambda: collections.Counter())
for line in open(fn):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); last[cur] = None; continue
    if cur is None: continue
    t = line.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    if re.match(r'^\.?L\w+:', t): continue
    op = t.split()[0]
    args = t[len(op):]
    if op.startswith('v_cndmask_b32_e32'):
        stats[cur]['cnd_e32_total'] += 1
        stats[cur]['cnd_e32_after_' + str(last[cur])] += 1
    elif op.startswith('v_cndmask_b32_e64') and re.search(r'\bvcc\b', args):
        stats[cur]['cnd_e64_vcc'] += 1
    # writers of vcc
    dst = args.split(',')[0].strip() if args else ''
    writes_vcc = False
    if op.startswith('s_') and dst in ('vcc', 'vcc_lo', 'vcc_hi'): writes_vcc = True; w = 'S'
    elif op.startswith('v_cmp') and (op.endswith('_e32') or dst == 'vcc'): writes_vcc = True; w = 'V'
    elif op.startswith(('v_add_co', 'v_sub_co', 'v_addc_co', 'v_subb_co', 'v_subrev_co', 'v_subbrev_co', 'v_mad_u64_u32', 'v_mad_i64_i32', 'v_div_scale')):
        if op.endswith('_e32') or re.search(r',\s*vcc\s*,', args) : writes_vcc = True; w = 'V'
    if writes_vcc: last[cur] = w
for k, c in stats.items():
    if kern and kern not in k: continue
    print(k, dict(c))
