# through gpurun: per-workgroup start / end stamps of two runs of the timing build -> which CUs / XCCs finish late, and whether
# the same ones do in both runs
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/wg_spread
mkdir -p $OUT
cp build/wgt_inputs/*.bin $OUT/
W=${W:-build/wg_timing_f64cap8}
# run 2: the same records rotated by 5000 (the data of a workgroup lands on another CU): does the lateness follow the CU or the data?
python3 -c "
import numpy as np,sys
for n in ('a','b'):
    x=np.fromfile('$OUT/%s.bin'%n,dtype=np.uint32).reshape(-1,168); np.roll(x,5000,axis=0).tofile('$OUT/%s_rot.bin'%n)
"
timeout -k 10 120 $W $OUT/delta.bin $OUT/a.bin $OUT/b.bin 0.3 > $OUT/run1.csv 2> $OUT/run1.txt
timeout -k 10 120 $W $OUT/delta.bin $OUT/a_rot.bin $OUT/b_rot.bin 0.3 > $OUT/run2.csv 2> $OUT/run2.txt
rm -f $OUT/*.bin
python3 - <<'PY'
import csv, statistics as st, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/wg_spread'
runs=[list(csv.DictReader(open(out+'/run%d.csv'%r))) for r in (1,2)]
def cu_key(r):
    h=int(r['hw_id']); return (int(r['xcc_id']),(h>>13)&7,(h>>12)&1,(h>>8)&15)
res=[]
for rows in runs:
    cu={}
    for r in rows:
        cu.setdefault(cu_key(r),[]).append((float(r['start_us']),float(r['end_us'])))
    res.append(cu)
    ends=sorted(max(e for _,e in v) for v in cu.values())
    durs=[e-s for v in cu.values() for s,e in v]
    print('CUs',len(cu),'last-end per CU: min %.0f median %.0f p90 %.0f max %.0f | wg duration mean %.0f sd %.0f'%(ends[0],ends[len(ends)//2],ends[int(len(ends)*.9)],ends[-1],st.mean(durs),st.pstdev(durs)))
    # within-CU: sorted end times of the 4 WGs
    q=[sorted(e for _,e in v) for v in cu.values() if len(v)==4]
    print('  mean of k-th finisher per CU:',[round(st.mean(x[k] for x in q),1) for k in range(4)])
    byx={}
    for k,v in cu.items(): byx.setdefault(k[0],[]).append(max(e for _,e in v))
    print('  per XCC mean last-end:',{x:round(st.mean(v)) for x,v in sorted(byx.items())})
common=set(res[0])&set(res[1])
a=[max(e for _,e in res[0][k]) for k in common]; b=[max(e for _,e in res[1][k]) for k in common]
ma,mb=st.mean(a),st.mean(b)
cov=sum((x-ma)*(y-mb) for x,y in zip(a,b))/len(a)
print('correlation of per-CU last-end between the two runs: %.3f over %d CUs'%(cov/(st.pstdev(a)*st.pstdev(b)),len(common)))
slow=sorted(common,key=lambda k:-(max(e for _,e in res[0][k])))[:8]
print('slowest CUs run1:',[(k,round(max(e for _,e in res[0][k])),round(max(e for _,e in res[1][k]))) for k in slow])
PY
