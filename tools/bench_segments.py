"""Any-hit microbench on two kinds of segments from surface points of the bench scene: to points on the ceiling panels (what NEE
shoots) and to the camera origin (what light tracing's connectToCamera shoots).  Usage: python tools/bench_segments.py [n]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s)
rng = np.random.default_rng(5)
p, nn = scenes.surface_points(s, n, rng)
p = (p + nn * 1e-3).astype(np.float32)   # just off the surface, on the side the normal points to
panels = np.array([(-2.0, -1.2), (2.0, -1.2), (-2.0, 1.2), (2.0, 1.2)], np.float32)
li = rng.integers(0, 4, n)
to_light = np.stack([panels[li, 0] + rng.uniform(-0.5, 0.5, n), np.full(n, 2.98), panels[li, 1] + rng.uniform(-0.3, 0.3, n)], 1).astype(np.float32)
cam = np.tile(np.array(s.cam_params["frm"], np.float32), (n, 1))
dev = torch.device("cuda:0")
for name, q in (("to the light panels", to_light), ("to the camera", cam)):
    segs = np.concatenate([p, q], 1).astype(np.float32)
    d_s = torch.from_numpy(segs).to(dev); d_v = torch.empty(n, dtype=torch.uint8, device=dev)
    sc.counters_enable(True); sc.visible_dev(d_s.data_ptr(), n, d_v.data_ptr(), 0); c = sc.counters(); sc.counters_enable(False)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        sc.visible_dev(d_s.data_ptr(), n, d_v.data_ptr(), 0)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    vis = float(d_v.float().mean())
    print(f"{name:20s}: {n/min(ts)/1e6:8.1f} Mrays/s  nodes/ray {c['any_nodes_visited']/n:.1f} tris/ray {c['any_tris_tested']/n:.1f} lane occ {c['any_lane_busy']/max(c['any_lane_slots'],1):.3f} "
          f"max steps {c['any_max_steps']} visible {vis:.2f} exact-path {c['exact_path_rays']}", flush=True)
