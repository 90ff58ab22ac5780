"""Where a traversal wave's time goes: needs a library built with -DRT_DBG_PHASE (tools/build_variants.sh phase "-DRT_DBG_PHASE";
RT_LIB=build_ab/phase.so python tools/dbg_phase.py).  Such a build reports shader cycles through the counters of a counting pass."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s); sc.counters_enable(True)
rt = api.RayTracer(sc); rt.render(api.PATH, 4, 1, 0); rt.sync(); c = rt.counters()
print({k: c[k] for k in c if 'lane' in k or 'nodes' in k or 'tris' in k})
for tag, al, rf, rt_, nx in (("extend", "lane_slots", "lane_busy", "nodes_visited", "tris_tested"), ("connect", "any_lane_slots", "any_lane_busy", "any_nodes_visited", "any_tris_tested")):
    print(tag, "refill share %.3f  of which settle+retire %.3f  next (work atomic) %.3f  load+begin %.3f" % (c[rf] / c[al], c[rt_] / c[al], c[nx] / c[al], (c[rf] - c[rt_] - c[nx]) / c[al]))
