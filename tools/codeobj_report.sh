#!/bin/bash
# tools/codeobj_report.sh -> per-kernel register / scratch / code size and the instruction mix of k_compose_wg: the
# static view of the gfx950 code objects of the last build (cofhe_amd/csrc/obj/*.o, one per part of cofhe_hip.hip)
set -e
cd "$(dirname "$0")/.."
BIN=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
for n in part0 part1 part2 wire; do
  cp cofhe_amd/csrc/obj/$n.o $TMP/$n.o
  (cd $TMP && $BIN/llvm-objdump --offloading $n.o > /dev/null && mv $n.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 $n.co)
done
echo "== kernels (metadata notes of the code objects)"
for f in $TMP/*.co; do
  $BIN/llvm-readelf --notes "$f" 2>/dev/null | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    print('%-34s vgpr %-4s sgpr %-4s scratch %-5s B  vgpr_spills %-4s lds %-6s B' % (re.sub(r'^_ZN7cofhe_k\d+', '', g('name'))[:34], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('vgpr_spill_count'), g('group_segment_fixed_size')))
"
done
echo "== machine code bytes per kernel"
for f in $TMP/*.co; do $BIN/llvm-readelf -sW "$f" | awk '$4=="FUNC" {printf "%-80s %8d\n", $8, $3}'; done | sort -u
echo "== instruction mix of k_compose_wg (static, all paths)"
SYM=$($BIN/llvm-readelf -sW $TMP/part0.co | awk '$4=="FUNC" && /k_compose_wg/ {print $8}' | head -1)
$BIN/llvm-objdump -d --disassemble-symbols=$SYM $TMP/part0.co | awk 'NF>1 && $1 ~ /^(v_|s_|ds_|global_|scratch_|buffer_)/ {print $1}' | sed -E 's/_e(32|64)$//; s/_dpp$//; s/_sdwa$//' | sort | uniq -c | sort -rn > $TMP/mix.txt
awk '{t+=$1} END {print "total instructions:", t}' $TMP/mix.txt
awk '$2 ~ /^v_/ {v+=$1} $2 ~ /^s_/ {s+=$1} $2 ~ /^ds_/ {d+=$1} $2 ~ /^(global|scratch|buffer)_/ {m+=$1} END {print "VALU", v, " SALU", s, " LDS", d, " VMEM", m}' $TMP/mix.txt
head -40 $TMP/mix.txt
rm -rf $TMP
