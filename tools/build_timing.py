import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ["RT_BUILD_TIMING"] = "1"
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
for db in (True, True, False):
    t = time.perf_counter(); sc = api.Scene(s, device_build=db); print("device_build", db, "create s", time.perf_counter() - t, flush=True); sc.close()
