#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- builds build/libcofhe_hip_NAME.so (tuning variants of the
# product library; bench.py loads one with --lib).  SRC=<dir> builds from another source tree (e.g. an export of an
# earlier commit: git archive HEAD cofhe_amd/csrc include | tar -x -C /tmp/base_src; SRC=/tmp/base_src).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
S=${SRC:-.}
mkdir -p build/obj_$name
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value $*"
pids=()
for part in 0 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DCOFHE_PART=$part -c $S/cofhe_amd/csrc/cofhe_hip.hip -o build/obj_$name/part$part.o & pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c $S/cofhe_amd/csrc/wire.hip -o build/obj_$name/wire.o & pids+=($!)
/opt/rocm/bin/hipcc $FLAGS -c $S/cofhe_amd/csrc/shard.hip -o build/obj_$name/shard.o & pids+=($!)
/opt/rocm/bin/hipcc $FLAGS -c $S/cofhe_amd/csrc/wide.hip -o build/obj_$name/wide.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libcofhe_hip_$name.so build/obj_$name/part0.o build/obj_$name/part1.o build/obj_$name/part2.o build/obj_$name/wire.o build/obj_$name/shard.o build/obj_$name/wide.o -ldl
echo "built build/libcofhe_hip_$name.so"
