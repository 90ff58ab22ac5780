import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
a = api.RayTracer(sc); b = api.RayTracer(sc); c = api.RayTracer(sc); d = api.RayTracer(sc)
for r in (a, b, c, d): r.render(api.PATH, 4, 1, 0); r.sync()
def run(rts, spp_each):
    for r in rts: r.sync()
    t = time.perf_counter()
    for i, r in enumerate(rts): r.render(api.PATH, spp_each, 1, 100 + i * spp_each)
    for r in rts: r.sync()
    return time.perf_counter() - t
for name, rts, spp in (("1 renderer x16", [a], 16), ("2 renderers x8", [a, b], 8), ("4 renderers x4", [a, b, c, d], 4), ("2 renderers x16", [a, b], 16), ("1 renderer x32", [a], 32)):
    ts = [run(rts, spp) for _ in range(3)]
    tot = spp * len(rts)
    print(f"{name}: {min(ts)*1e3/tot:.2f} ms/spp", flush=True)
