"""Per-kernel times (HIP events) of light tracing on the bench scene.  Usage: python tools/prof_light.py [spp [streams]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s, device_build=True)
rt = api.RayTracer(sc)
rt.render(api.LIGHT, spp, 1, 0); rt.sync(); rt.clear(); rt.counters(reset=True)
t = time.perf_counter(); rt.render(api.LIGHT, spp, 1, 100); rt.sync(); dt = time.perf_counter() - t
c = rt.counters(reset=True)
print(f"unprofiled: {dt*1e3/spp:.3f} ms/spp, {(c['closest_rays']+c['any_rays'])/dt/1e6:.1f} Mrays/s, closest {c['closest_rays']/spp/1e6:.2f} M/spp any {c['any_rays']/spp/1e6:.2f} M/spp")
if len(sys.argv) > 2: rt.set_streams(int(sys.argv[2]))   # 1: every kernel alone on the GPU (unshared durations)
rt.render(api.LIGHT, spp, 1, 150); rt.sync()
rt.profile_enable(True); rt.profile(reset=True)
t = time.perf_counter(); rt.render(api.LIGHT, spp, 1, 200); rt.sync(); dt = time.perf_counter() - t
print(f"profiled: {dt*1e3/spp:.3f} ms/spp")
for k, v in rt.profile(reset=True).items(): print(k, v)
sc.counters_enable(True)
rt2 = api.RayTracer(sc); rt2.render(api.LIGHT, 4, 1, 300); c = rt2.counters()
print({k: v for k, v in c.items()})
print("closest: nodes/ray", c["nodes_visited"] / c["closest_rays"], "tris/ray", c["tris_tested"] / c["closest_rays"], "lane occ", c["lane_busy"] / max(c["lane_slots"], 1), "max steps", c.get("max_steps"))
print("any: nodes/ray", c["any_nodes_visited"] / c["any_rays"], "tris/ray", c["any_tris_tested"] / c["any_rays"], "lane occ", c["any_lane_busy"] / max(c["any_lane_slots"], 1), "max steps", c.get("any_max_steps"))
