import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, util
from ray_amd import api, hip, scenes
w, h = 64, 48
r = api.CreateRenderer(api.Settings(w, h), "HIP")
s = r.CreateScene(); scenes.cornell_sky(s, envmap_resolution=64)
print("baked on", s.sky_bake_info())
b_dev = np.frombuffer(api.export_scene_blob(s), dtype=np.uint8)
s2 = api.CreateSceneHIP(); scenes.cornell_sky(s2, envmap_resolution=64)
print("baked on", s2.sky_bake_info())
b_host = np.frombuffer(api.export_scene_blob(s2), dtype=np.uint8)
print("blob sizes", b_dev.size, b_host.size, "equal:", b_dev.size == b_host.size and bool(np.array_equal(b_dev, b_host)))
if b_dev.size == b_host.size:
    d = np.nonzero(b_dev != b_host)[0]
    print("differing bytes", d.size, d[:20])
region = api.RegionContext((0, 0, w, h))
for _ in range(2):
    r.RenderScene(s, region)
img = r.get_raw_pixels_ref()
print("RendererHIP frame: nan", int(np.isnan(img).sum()), "max", float(np.nanmax(img)))
ref, _ = O.render_ref(lambda sc: scenes.cornell_sky(sc, envmap_resolution=64), w, h, 2)
rimg = ref.get_raw_pixels_ref()
print("RendererRef frame: nan", int(np.isnan(rimg).sum()), "max", float(np.nanmax(rimg)))
print(util.frame_metrics(img, rimg))
