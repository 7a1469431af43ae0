# through gpurun: tools/bench_ops.py (without the 1024x1024 and k = 256 legs) against the variant libraries in LIBS
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in $LIBS; do
  echo "== $f"
  OPS_SKIP=big,k256 timeout -k 10 600 python -c "
import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ['bench_ops.py']
import torch; torch.cuda.init()
import cofhe_amd
cofhe_amd.load_library(os.path.abspath('$f'))
__file__ = os.path.abspath('tools/bench_ops.py')
exec(compile(open('tools/bench_ops.py').read(), 'tools/bench_ops.py', 'exec'))
" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except ValueError: continue
    print('   %-70s %-16s %9.3f ms' % (d['op'][:70], d['shape'], d['ms']))
"
done
