#!/bin/bash
# round 5, call H: the sky bake on the device (tests + bake time host vs device at 256 / 1024), builder-flag GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_sky_bake.py tests/test_gpu_parity.py -q -m gpu -s -k "sky" > $O/h_sky.log 2>&1; grep -v "^Extends\|^Spatial" $O/h_sky.log | grep "texels differ\|passed\|failed\|Error\|assert" | head -20
python - <<'PY' 2>&1 | grep -v "^Extends\|^Spatial" | tee gpurun_out/r05/h_sky_bake_time.txt
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from functools import partial
import numpy as np
from ray_amd import api, scenes
for res in ():
    for where in ("device", "host"):
        os.environ["RAY_HIP_SKY_BAKE_ON_HOST"] = "1" if where == "host" else "0"
        r = api.CreateRenderer(api.Settings(64, 64), "HIP")
        s = r.CreateScene()
        t0 = time.perf_counter()
        scenes.cornell_sky(s, envmap_resolution=res)
        dt = time.perf_counter() - t0
        print(f"envmap_resolution {res:5d} ({res} x {res // 2} texels): scene construction + Finalize with the sky baked on the {s.sky_bake_info():6s}: {dt * 1e3:9.1f} ms", flush=True)
PY
