"""Path tracing throughput at a given resolution / spp per call.  Usage: python tools/bench_res.py W H SPP [SPP...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
W, H = int(sys.argv[1]), int(sys.argv[2])
s = scenes.interior("bathroom", 1.0, W, H)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
for spp in map(int, sys.argv[3:]):
    rt.render(api.PATH, spp, 1, 0); rt.sync(); rt.clear(); rt.counters(reset=True)
    t = time.perf_counter(); rt.render(api.PATH, spp, 1, 1000); rt.sync(); dt = time.perf_counter() - t
    c = rt.counters(reset=True)
    print(f"{os.environ.get('RT_LIB','default')} {W}x{H} {spp:4d} spp: {dt*1e3/spp:8.3f} ms/spp {(c['closest_rays']+c['any_rays'])/dt/1e6:8.1f} Mrays/s", flush=True)
