"""Which rays make a launch's tail?  Counted passes of 1 spp on the bench scene: longest walk (wave steps), exact-walk rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
for name, kw in (("bathroom", {}), ("bathroom, conductor instead of the rough dielectric", {"conductor": True}), ("living-room", {})):
    s = scenes.interior(name.split(",")[0], 1.0, 1920, 1080, **kw)
    sc = api.Scene(s, device_build=True)
    sc.counters_enable(True)
    rt = api.RayTracer(sc)
    rt.render(api.PATH, 2, 1, 0)
    c = rt.counters()
    print(name, {k: c[k] for k in ("closest_rays", "any_rays", "exact_path_rays", "max_steps", "any_max_steps")},
          "steps/ray", (c["nodes_visited"] + c["tris_tested"]) / c["closest_rays"], flush=True)
    rt.close(); sc.close()
