#!/bin/bash
# round 3: upload timeline with the device-side collapse, GPU tests of the builders, bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03i
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
cat > /tmp/upload_probe.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import bench
from ray_amd import hip, api
blob, info = bench.get_scene_blob("bistro", bench.WORKLOADS["bistro"], 0, 1, lambda: None)
ctx = hip.Context(0)
ctx.upload_static(api.pmj_table()); ctx.resize(64, 64)
for k in range(2):
    t0 = time.perf_counter(); ctx.upload_scene_blob(blob); print("upload_scene_blob wall %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
PY
RAYHIP_TRACE_UPLOAD=1 timeout 300 python /tmp/upload_probe.py 2>&1 | grep -E "rayhip_scene_upload|wall" > $OUT/upload_timeline.txt; cat $OUT/upload_timeline.txt
timeout 900 python -m pytest tests/test_gpu_bvh_build.py tests/test_gpu_parity.py tests/test_gpu_instance_update.py -m gpu -q -x > $OUT/gputest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 exit $?"
python3 -c "
import json; d=json.load(open('$OUT/bench20.json')); print('bench20', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), {k: round(v) for k,v in d['stage_us_per_step'].items()})" || tail -5 $OUT/bench20.err
