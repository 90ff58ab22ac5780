"""Experiment: how much do the traversal kernels gain when incoherent rays arrive sorted by (origin cell Morton code, direction octant)?
Rays are sorted on the host (numpy); results are order-independent."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes

def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x
def keys(o, d, lo, hi, bits, dir_first=False):
    c = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    m = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    octant = ((d[:, 0] < 0).astype(np.uint64)) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2)
    return (octant << np.uint64(3 * bits)) | m if dir_first else (m << np.uint64(3)) | octant

s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s)
n = 1 << 23
lo = s.verts[:, :, :3].reshape(-1, 3).min(0); hi = s.verts[:, :, :3].reshape(-1, 3).max(0)
dev = torch.device("cuda:0")
def run(name, arr, any_hit):
    d_r = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    d_o = torch.empty(len(arr) * (1 if any_hit else 20), dtype=torch.uint8, device=dev)
    f = sc.visible_dev if any_hit else sc.traverse_dev
    sc.counters_enable(True); f(d_r.data_ptr(), len(arr), d_o.data_ptr(), 0); c = sc.counters(); sc.counters_enable(False)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); f(d_r.data_ptr(), len(arr), d_o.data_ptr(), 0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    p = "any_" if any_hit else ""
    print(f"{name:44s} {len(arr)/min(ts)/1e6:8.1f} Mrays/s  lane occ {c[p+'lane_busy']/max(c[p+'lane_slots'],1):.3f} nodes/ray {c[p+'nodes_visited' if any_hit else 'nodes_visited']/len(arr):.1f}", flush=True)
rays = scenes.bounce_rays(s, n, 2)
run("bounce unsorted", rays, False)
for bits in (3, 5, 7):
    k = keys(rays[:, :3], rays[:, 3:], lo, hi, bits)
    run(f"bounce sorted cell 2^{bits} + octant", rays[np.argsort(k, kind='stable')], False)
k = keys(rays[:, :3], rays[:, 3:], lo, hi, 5, dir_first=True)
run("bounce sorted octant-major, cell 2^5", rays[np.argsort(k, kind='stable')], False)
# bucket-only order: 4096 buckets (cell 2^3 + octant), random order inside a bucket = what a counting-sort scatter gives
k = keys(rays[:, :3], rays[:, 3:], lo, hi, 3)
run("bounce binned 4096 buckets (stable)", rays[np.argsort(k, kind='stable')], False)
segs = scenes.shadow_segments(s, n, 3)
run("shadow unsorted", segs, True)
d = segs[:, 3:] - segs[:, :3]
for bits in (3, 5):
    k = keys(segs[:, :3], d, lo, hi, bits)
    run(f"shadow sorted cell 2^{bits} + octant", segs[np.argsort(k, kind='stable')], True)
# NEE-like segments: endpoints on the four ceiling panels
rng = np.random.default_rng(5)
lights = np.array([(-2.0, -1.2), (2.0, -1.2), (-2.0, 1.2), (2.0, 1.2)])
li = rng.integers(0, 4, n)
p2 = np.stack([lights[li, 0] + rng.uniform(-0.5, 0.5, n), np.full(n, 2.98), lights[li, 1] + rng.uniform(-0.3, 0.3, n)], 1).astype(np.float32)
nee = np.concatenate([segs[:, :3], p2], 1).astype(np.float32)
run("NEE-like unsorted", nee, True)
k = keys(nee[:, :3], nee[:, 3:] - nee[:, :3], lo, hi, 4) | (li.astype(np.uint64) << np.uint64(20))
run("NEE-like sorted light, cell 2^4, octant", nee[np.argsort(k, kind='stable')], True)
k = keys(nee[:, :3], nee[:, 3:] - nee[:, :3], lo, hi, 5)
run("NEE-like sorted cell 2^5 + octant", nee[np.argsort(k, kind='stable')], True)
