"""One warm render + one measured render of a mode on the bench scene (for rocprofv3 --kernel-trace --stats).
Usage: python3 tools/run_mode.py MODE SPP [VPLS]   MODE = path|light|mlt|radiosity"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
mode = {"path": api.PATH, "light": api.LIGHT, "mlt": api.MLT, "radiosity": api.RADIOSITY}[sys.argv[1]]
spp = int(sys.argv[2]); vpls = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
if vpls: rt.traceVPLs(vpls, 1)
rt.render(mode, spp, 1, 0); rt.sync(); rt.clear(); rt.counters(reset=True)
t = time.perf_counter(); rt.render(mode, spp, 1, 1000); rt.sync(); dt = time.perf_counter() - t
c = rt.counters()
print(f"{sys.argv[1]} {spp} spp: {dt*1e3/spp:.3f} ms/spp, {(c['closest_rays']+c['any_rays'])/dt/1e6:.1f} Mrays/s", flush=True)
