"""Which path vertices make non-finite pixels?  The GPU finds (pixel, sample) pairs by bisection over sample ranges, the CPU oracle
replays those samples with a per-vertex log (orc_trace_path).
usage: python tools/find_nonfinite.py [variant] [width] [height] [seed] [total spp] [glazed|dielectric ...]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa: F401
import orc
from raytracer_amd import api, scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "bathroom"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
total = int(sys.argv[5]) if len(sys.argv) > 5 else 96
flags = {k: True for k in sys.argv[6:]}
s = scenes.interior(variant, 1.0, W, H, **flags)
sc = api.Scene(s); rt = api.RayTracer(sc)

def bad_pixels(first, count):
    rt.clear(); rt.render(api.PATH, count, seed, first)
    film, _ = rt.film()
    ys, xs = np.nonzero(~np.isfinite(film).all(axis=2))
    return set(zip(xs.tolist(), ys.tolist()))

found = []
def search(first, count):
    bad = bad_pixels(first, count)
    if not bad:
        return
    if count == 1:
        found.extend((x, y, first) for x, y in bad)
        return
    h = count // 2
    search(first, h); search(first + h, count - h)
chunk = 4096 if total >= 4096 else total
for f in range(0, total, chunk):
    search(f, min(chunk, total - f))
    if len(found) >= 8:
        break
print("non-finite samples:", len(found), found[:8])
o = orc.OracleScene(s)
L = orc.lib()
for x, y, k in found[:8]:
    log = np.zeros((8, 20), np.float32); L3 = np.zeros(3, np.float32)
    n = L.orc_trace_path(o.h, x, y, seed, k, log.ctypes.data, 8, L3.ctypes.data)
    print(f"pixel ({x},{y}) sample {k}: oracle L = {L3}")
    for r in log[:n]:
        print(f"   mat {int(r[0])} T {r[1:4]} direct {r[4:7]} f {r[7:10]} pdf {r[10]:.6g} wi {r[11:14]} wo {r[14:17]} sN {r[17:20]}")
