#!/bin/bash
# round 6, call H: a variant list through tools/variants.py (64-layer passes of the headline scene), then parity of the default build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; rm -rf $O; mkdir -p $O
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/build.err
for rep in 1 2; do
RT_BATCH=64 RT_VARIANTS="$(cat $1)" timeout 900 python tools/variants.py run ${2:-bistro} 64 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $O/variants.txt
done
cat $O/variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "shade_forms or sparse_lights or frame_vs_reference or batching or partition or sharding" > $O/parity.log 2>&1; echo "pytest exit $?"; tail -3 $O/parity.log
