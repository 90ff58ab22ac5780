"""One sample per frame with the film looked at after every frame (what an interactive viewer does): nothing can be merged,
every frame is one 2 M-path launch sequence.  Usage: python tools/bench_interactive.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
rt.render(api.PATH, 4, 1, 0); rt.sync(); rt.clear(); rt.counters(reset=True)
t = time.perf_counter()
for i in range(frames):
    rt.render(api.PATH, 1, 1, 100 + i); rt.sync()
dt = time.perf_counter() - t
c = rt.counters()
print(f"{os.environ.get('RT_LIB', 'default')}: {dt*1e3/frames:.3f} ms/frame (1 spp), {(c['closest_rays']+c['any_rays'])/dt/1e6:.1f} Mrays/s", flush=True)
t = time.perf_counter()
for i in range(frames):
    rt.render(api.PATH, 1, 1, 200 + i); img = rt.tonemap()
dt = time.perf_counter() - t
print(f"  with tonemap + readback per frame: {dt*1e3/frames:.3f} ms/frame", flush=True)
