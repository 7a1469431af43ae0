# usage: bash tools/gpu_profile_ops.sh <tag>   (run through gpurun): per-operation timings plus the
# rocprofv3 kernel statistics of the same run (every kernel of the path, incl. the C3 matrix product)
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ops_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 1000 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py > $OUT/bench_ops.jsonl 2> $OUT/trace.err || (tail -20 $OUT/trace.err; exit 1)
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -v "at::native\|rocclr" $OUT/summary.txt | head -40
cat $OUT/bench_ops.jsonl
find $OUT -size +8M -delete
