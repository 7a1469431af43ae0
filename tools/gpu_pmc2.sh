set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_CVT -d $OUT/a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err || (tail -20 $OUT/a.err; exit 1)
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/b.err || (tail -20 $OUT/b.err; exit 1)
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 -d $OUT/c -o c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/c.err || (tail -20 $OUT/c.err; exit 1)
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -A 40 "k_compose" $OUT/summary.txt | grep -v "^==" | head -40
find $OUT -size +8M -delete
