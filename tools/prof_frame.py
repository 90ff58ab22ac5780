"""Per-kernel HIP-event times of single-sample frames on the bench scene.  Usage: python tools/prof_frame.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
rt.render(api.PATH, 4, 1, 0); rt.sync(); rt.clear()
rt.profile_enable(True); rt.profile(reset=True)
t = time.perf_counter()
for i in range(frames):
    rt.render(api.PATH, 1, 1, 100 + i); rt.sync()
dt = time.perf_counter() - t
print(f"{dt*1e3/frames:.3f} ms/frame with per-kernel events")
for k, (ms, n) in rt.profile(reset=True).items():
    print(f"  {k:12s} {ms/frames:8.3f} ms/frame in {n//frames if frames else 0} launches")
sc.counters_enable(True)
rt2 = api.RayTracer(sc); rt2.render(api.PATH, 1, 1, 500); c = rt2.counters()
print("max steps closest", c["max_steps"], "any", c["any_max_steps"], "lane occupancy", c["lane_busy"] / c["lane_slots"], c["any_lane_busy"] / c["any_lane_slots"])
