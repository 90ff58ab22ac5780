"""rt_intersect_closest / rt_intersect_any with HOST buffers (what a caller without device memory uses): rays cross PCIe both ways."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s)
for n in (1, 1000, 1 << 20, 1 << 24):
    rays = scenes.bounce_rays(s, n, 3)
    segs = scenes.shadow_segments(s, n, 4)
    sc.traverse(rays); sc.visible(segs)
    reps = 200 if n <= 1000 else 3
    t = time.perf_counter()
    for _ in range(reps): sc.traverse(rays)
    dt = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps): sc.visible(segs)
    dv = (time.perf_counter() - t) / reps
    print(f"n={n:9d}: closest {dt*1e3:9.3f} ms ({n/dt/1e6:8.1f} Mrays/s)   any {dv*1e3:9.3f} ms ({n/dv/1e6:8.1f} Mrays/s)", flush=True)
