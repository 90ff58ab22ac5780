"""How much of a traversal launch is its end: launch time over batch size for incoherent rays, T(n) = floor + n / rate.
Usage: python tools/probe_tail.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080)
sc = api.Scene(s)
dev = torch.device("cuda:0")
nmax = 1 << 24
rays = scenes.bounce_rays(s, nmax, 2); segs = scenes.shadow_segments(s, nmax, 3)
d_r = torch.from_numpy(rays).to(dev); d_s = torch.from_numpy(segs).to(dev)
d_h = torch.empty(nmax * 5, dtype=torch.float32, device=dev); d_v = torch.empty(nmax, dtype=torch.uint8, device=dev)
for name, fn, src, dst in (("closest", sc.traverse_dev, d_r, d_h), ("any-hit", sc.visible_dev, d_s, d_v)):
    pts = []
    for n in (1 << 16, 1 << 18, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
        ts = []
        for rep in range(7):
            torch.cuda.synchronize(); t = time.perf_counter()
            fn(src.data_ptr(), n, dst.data_ptr(), 0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        pts.append((n, sorted(ts)[len(ts) // 2]))
    (n1, t1), (n2, t2) = pts[-2], pts[-1]
    rate = (n2 - n1) / (t2 - t1)
    print(name, "marginal rate %.0f Mrays/s;" % (rate / 1e6), " ".join(f"n=2^{int(np.log2(n))}: {t*1e3:.3f} ms (floor {(t - n / rate)*1e3:.3f})" for n, t in pts), flush=True)
