"""Where the lanes of the traversal kernels go: of all lane-slots (64 x wave steps), how many held a walk (busy) and how many
did a node or a triangle that step (served).   python tools/lane_stats.py [spp]
With RT_LIB pointing at a library built with -DRT_DBG_WAVESTEPS the node / triangle counts are 64 x the wave steps of each kind."""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = scenes.bathroom(1.0, 1920, 1080)
sc = api.Scene(s); sc.counters_enable(True)
rt = api.RayTracer(sc)
rt.render(api.PATH, spp, 1, 0); rt.sync()
c = rt.counters()
for tag, rays, nodes, tris, slots, busy in (("closest", "closest_rays", "nodes_visited", "tris_tested", "lane_slots", "lane_busy"),
                                            ("any-hit", "any_rays", "any_nodes_visited", "any_tris_tested", "any_lane_slots", "any_lane_busy")):
    served = c[nodes] + c[tris]
    print(f"{tag}: rays {c[rays]}  nodes/ray {c[nodes]/c[rays]:.2f} tris/ray {c[tris]/c[rays]:.2f}  lane-slots/ray {c[slots]/c[rays]:.1f}  busy {c[busy]/c[slots]:.3f}  served {served/c[slots]:.3f}"
          f"  served/busy {served/c[busy]:.3f}  refused {c.get('gate_refused' if tag == 'closest' else 'any_gate_refused')}")
