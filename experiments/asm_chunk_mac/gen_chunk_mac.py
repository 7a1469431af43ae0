#!/usr/bin/env python3
"""prints the inline-asm statements of chunk_mac (cofhe_amd/csrc/mp.hpp, COFHE_ASM_MAC): python3 tools/gen_chunk_mac.py"""
# generates the inline-asm body of chunk_mac (5x5 limbs): rows (0,1,2) interleaved, then rows (3,4), then the row-end carries
def T(r): return "v[%d:%d]" % (116 + 4 * r, 117 + 4 * r)          # product + carry of row slot r
def Tlo(r): return "v%d" % (116 + 4 * r)
def Thi(r): return "v%d" % (117 + 4 * r)
def CP(r): return "v[%d:%d]" % (118 + 4 * r, 119 + 4 * r)          # (carry, 0) of row slot r
def CPlo(r): return "v%d" % (118 + 4 * r)
def CPhi(r): return "v%d" % (119 + 4 * r)
S = ["vcc", "s[92:93]", "s[94:95]"]
DUM = "s[96:97]"
def M(slot, row, j):
    src2 = "0" if j == 0 else CP(slot)
    return "v_mad_u64_u32 %s, %s, %%[x%d], %%[y%d], %s" % (T(slot), DUM, row, j, src2)
def A(slot, row, j):
    return "v_add_co_u32_e64 %%[w%d], %s, %%[w%d], %s" % (row + j, S[slot], row + j, Tlo(slot))
def C(slot, row, j):
    dst = "%%[e%d]" % row if j == 4 else CPlo(slot)
    return "v_addc_co_u32_e64 %s, %s, 0, %s, %s" % (dst, S[slot], Thi(slot), S[slot])
def triple(rows):
    out = ["v_mov_b32 %s, 0" % CPhi(s) for s in range(len(rows))]
    for j in range(5):
        for s, r in enumerate(rows):
            out += [M(s, r, j), A(s, r, j)]
        for s, r in enumerate(rows):
            out.append(C(s, r, j))
    return out
def pair(rows):
    a, b = rows
    out = ["v_mov_b32 %s, 0" % CPhi(s) for s in range(2)]
    out += [M(0, a, 0), A(0, a, 0), M(1, b, 0), A(1, b, 0), C(0, a, 0)]
    for j in range(1, 5):
        out += [M(0, a, j), C(1, b, j - 1), A(0, a, j), M(1, b, j), A(1, b, j), C(0, a, j)]
    out += ["s_nop 0", C(1, b, 4)]
    return out
def emit(name, lines, ops_out, ops_in, clob):
    s = "    asm(" + "\n        ".join('"%s\\n\\t"' % l for l in lines[:-1]) + '\n        "%s"\n' % lines[-1]
    s += "        : " + ", ".join(ops_out) + "\n        : " + ", ".join(ops_in) + "\n        : " + ", ".join('"%s"' % c for c in clob) + ");\n"
    return s
vregs = lambda n: ["v%d" % (116 + i) for i in range(4 * n)]
sregs = ["vcc", "s92", "s93", "s94", "s95", "s96", "s97"]
code = ""
code += emit("t", triple([0, 1, 2]),
             ['[w%d] "+v"(w[%d])' % (k, k) for k in range(0, 7)] + ['[e%d] "=&v"(e%d)' % (r, r) for r in (0, 1, 2)],
             ['[x%d] "v"(x[%d])' % (r, r) for r in (0, 1, 2)] + ['[y%d] "v"(y[%d])' % (j, j) for j in range(5)], vregs(3) + sregs)
code += emit("p", pair([3, 4]),
             ['[w%d] "+v"(w[%d])' % (k, k) for k in range(3, 9)] + ['[e%d] "=&v"(e%d)' % (r, r) for r in (3, 4)],
             ['[x%d] "v"(x[%d])' % (r, r) for r in (3, 4)] + ['[y%d] "v"(y[%d])' % (j, j) for j in range(5)], vregs(2) + sregs)
fin = ["v_add_co_u32_e64 %[w5], vcc, %[w5], %[e0]"]
for k, r in ((6, 1), (7, 2), (8, 3), (9, 4)):
    fin += ["s_nop 1", "v_addc_co_u32_e64 %%[w%d], vcc, %%[w%d], %%[e%d], vcc" % (k, k, r)]
fin += ["s_nop 1", "v_addc_co_u32_e64 %[w10], vcc, 0, %[w10], vcc", "s_nop 1"]
code += emit("f", fin, ['[w%d] "+v"(w[%d])' % (k, k) for k in range(5, 11)], ['[e%d] "v"(e%d)' % (r, r) for r in range(5)], ["vcc"])
print(code)
