# usage: bash tools/gpu_profile.sh <tag>   (run through gpurun; writes gpurun_out/prof_<tag>/)
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# pass 1: kernel trace + stats
timeout -k 10 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err || (tail -20 $OUT/trace.err; exit 1)
# pass 2,3,4: PMC counters, own runs
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc1.err || (tail -20 $OUT/pmc1.err; exit 1)
timeout -k 10 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc2.err || (tail -20 $OUT/pmc2.err; exit 1)
timeout -k 10 600 rocprofv3 --output-format csv --pmc WRITE_SIZE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $OUT/pmc3 -o pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc3.err || (tail -20 $OUT/pmc3.err; exit 1)
cd $OUT
find . -name "*.csv" | head -30
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
# keep only small files
find $OUT -size +8M -delete
