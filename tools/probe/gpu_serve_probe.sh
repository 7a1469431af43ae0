# through gpurun: euclid_serve on the GPU, Python client, current code and the code without the hint fix
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/probe; mkdir -p $OUT/csrc_nofix
cp cofhe_amd/csrc/*.hpp $OUT/csrc_nofix/
sed -i 's|^    tx = ty = tx > ty ? tx : ty;|    // (hint fix removed for the probe)|' $OUT/csrc_nofix/mp.hpp
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -Icofhe_amd/csrc -o $OUT/serve_fix tools/probe/serve_probe.hip
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -I$OUT/csrc_nofix -o $OUT/serve_nofix tools/probe/serve_probe.hip
echo "== with the fix"; python tools/probe/serve_probe.py $OUT/serve_fix ${1:?ops file: lines "<stop_bits> <x hex> <y hex>"}
echo "== without the fix"; python tools/probe/serve_probe.py $OUT/serve_nofix $1
