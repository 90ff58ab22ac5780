"""Traversal-only microbench: Mrays/s of the closest/any kernels on resident ray batches (bathroom-class scene).
Usage: python tools/bench_intersect.py [--detail 1.0] [--n 4194304] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--detail", type=float, default=1.0)
ap.add_argument("--n", type=int, default=1 << 22)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--scene", default="bathroom")
a = ap.parse_args()

s = scenes.interior(a.scene, a.detail, 1920, 1080)
t0 = time.time()
sc = api.Scene(s)
print("scene", s.name, sc.info(), "create %.2fs" % (time.time() - t0), flush=True)
o = sc.generate_rays(np.stack(np.meshgrid(np.arange(1920) + 0.5, np.arange(1080) + 0.5), -1).reshape(-1, 2).astype(np.float32))
sets = {"primary(scanline)": o, "bounce": scenes.bounce_rays(s, a.n, 2)}
dev = torch.device("cuda:0")
for name, rays in sets.items():
    d_r = torch.from_numpy(rays).to(dev)
    d_h = torch.empty(len(rays) * 5, dtype=torch.float32, device=dev)
    for mode in (0,):
        sc.counters_enable(True)
        sc.traverse_dev(d_r.data_ptr(), len(rays), d_h.data_ptr(), mode)
        c = sc.counters()
        sc.counters_enable(False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            torch.cuda.synchronize(); t = time.perf_counter()
            sc.traverse_dev(d_r.data_ptr(), len(rays), d_h.data_ptr(), mode)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        best = min(ts)
        nodes, tris = c["nodes_visited"] / len(rays), c["tris_tested"] / len(rays)
        bpr = 24 + 20 + 32 + nodes * 80 + tris * 64   # ray, hit, gate box; 80-B quantised 8-wide nodes, 64-B triangle records
        print(f"{name:18s} closest mode{mode}: {len(rays)/best/1e6:9.1f} Mrays/s  nodes/ray {nodes:.1f} tris/ray {tris:.1f} "
              f"alg bytes/ray {bpr:.0f} -> {len(rays)*bpr/best/1e9:.0f} GB/s  exact-path rays {c['exact_path_rays']} lane occ {c['lane_busy']/max(c['lane_slots'],1):.3f} max steps/ray {c['max_steps']} (mean {(c['nodes_visited']+c['tris_tested'])/len(rays):.0f})", flush=True)
segs = scenes.shadow_segments(s, a.n, 3)
d_s = torch.from_numpy(segs).to(dev)
d_v = torch.empty(len(segs), dtype=torch.uint8, device=dev)
sc.counters_enable(True); sc.visible_dev(d_s.data_ptr(), len(segs), d_v.data_ptr(), 0); c = sc.counters(); sc.counters_enable(False)
ts = []
for _ in range(a.reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    sc.visible_dev(d_s.data_ptr(), len(segs), d_v.data_ptr(), 0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print(f"shadow any-hit: {len(segs)/min(ts)/1e6:9.1f} Mrays/s nodes/ray {c['any_nodes_visited']/len(segs):.1f} tris/ray {c['any_tris_tested']/len(segs):.1f} lane occ {c['any_lane_busy']/max(c['any_lane_slots'],1):.3f} max steps/ray {c['any_max_steps']}", flush=True)
