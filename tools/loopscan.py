#!/usr/bin/env python3
"""tools/loopscan.py <object.o> <kernel> -- loops of a gfx950 kernel (backward branches in the disassembly of the code object
inside a hipcc object file) with their instruction mix: where the spill loads / stores, moves and selects sit relative to
the barriers of the served remainder sequence.  Static view; tools/codeobj_report.sh has the whole-kernel totals."""
import os
import re
import subprocess
import sys
import tempfile

obj, kern = sys.argv[1], sys.argv[2]
BIN = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
subprocess.check_call(["cp", obj, tmp + "/x.o"])
subprocess.check_call([BIN + "/llvm-objdump", "--offloading", "x.o"], cwd=tmp, stdout=subprocess.DEVNULL)
co = tmp + "/" + [f for f in os.listdir(tmp) if "gfx950" in f][0]
syms = subprocess.check_output([BIN + "/llvm-readelf", "-sW", co], text=True)
row = [l.split() for l in syms.splitlines() if " FUNC " in l and kern in l][0]
start, size = int(row[1], 16), int(row[2])
dis = subprocess.check_output([BIN + "/llvm-objdump", "-d", "--start-address=%d" % start, "--stop-address=%d" % (start + size), co], text=True).splitlines()
ins, addr_of = [], {}
for l in dis:
    m = re.search(r"//\s*([0-9A-Fa-f]{12}):", l)
    p = l.split()
    if m and len(p) > 1 and re.match(r"^(v_|s_|ds_|scratch_|global_|buffer_)", p[0]):
        addr_of[int(m.group(1), 16)] = len(ins)
        ins.append((int(m.group(1), 16), l.split("//")[0].strip()))
bars = [i for i, (_, t) in enumerate(ins) if t.startswith("s_barrier")]
print("instructions", len(ins), "barriers at", bars)
loops = []
for i, (a, t) in enumerate(ins):
    m = re.match(r"(s_cbranch\w*|s_branch)\s+(\d+)", t)
    if m:
        off = int(m.group(2))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt in addr_of and addr_of[tgt] <= i:
            loops.append((addr_of[tgt], i))


def stats(a, b):
    seg = [t for _, t in ins[a:b + 1]]
    c = lambda pre: sum(1 for x in seg if x.startswith(pre))
    return dict(n=len(seg), valu=c("v_"), salu=c("s_"), lds=c("ds_"), scr_ld=c("scratch_load"), scr_st=c("scratch_store"), mov=c("v_mov_b32"),
                cnd=c("v_cndmask"), nop=c("s_nop"), bar=c("s_barrier"), mad=c("v_mad_u64"))


for a, b in sorted(loops):
    st = stats(a, b)
    if st["bar"] > 0 or st["n"] > 120:
        print("loop [%d, %d]" % (a, b), st)
if len(sys.argv) > 3:
    a, b = int(sys.argv[3]), int(sys.argv[4])
    for _, t in ins[a:b + 1]:
        print("   ", t)
