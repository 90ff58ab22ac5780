"""Launch floor: per-kernel HIP-event times of single-sample frames at small resolutions on the bench scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
for W, H in ((8, 8), (64, 64), (256, 256), (960, 540), (1920, 1080)):
    s = scenes.interior("bathroom", 1.0, W, H)
    sc = api.Scene(s)
    rt = api.RayTracer(sc)
    rt.render(api.PATH, 4, 1, 0); rt.sync(); rt.clear()
    rt.profile_enable(True); rt.profile(reset=True)
    frames = 16
    t = time.perf_counter()
    for i in range(frames):
        rt.render(api.PATH, 1, 1, 100 + i); rt.sync()
    dt = time.perf_counter() - t
    p = rt.profile(reset=True)
    print(f"{W}x{H}: {dt*1e3/frames:.3f} ms/frame; extend {p['extend'][0]/frames:.3f} ms / {p['extend'][1]//frames} launches, connect {p['connect'][0]/frames:.3f}, shade {p['shade'][0]/frames:.3f}, control {p['control'][0]/frames:.3f}", flush=True)
    rt.close(); sc.close()
