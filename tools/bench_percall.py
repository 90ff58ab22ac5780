"""Throughput when the caller adds samples the way upstream's main loop does: one render() call per spp (Main.cpp:79-137).
Usage: python tools/bench_percall.py [spp_per_call ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
rt.render(api.PATH, 8, 1, 0); rt.sync(); rt.clear()
for per in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    total = 32
    rt.counters(reset=True)
    t = time.perf_counter()
    for i in range(total // per):
        rt.render(api.PATH, per, 1, 100 + i * per)
    rt.sync()
    dt = time.perf_counter() - t
    c = rt.counters(reset=True)
    print(f"{per} spp per call x {total // per} calls: {dt*1e3/total:.3f} ms/spp, {(c['closest_rays']+c['any_rays'])/dt/1e6:.1f} Mrays/s", flush=True)
