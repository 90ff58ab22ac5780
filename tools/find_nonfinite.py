"""Which path vertices make the non-finite pixels of the default bench scene?  GPU finds (pixel, sample) pairs, the CPU oracle
replays those samples with a per-vertex log (orc_trace_path)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa: F401
import orc
from raytracer_amd import api, scenes
W, H = 1920, 1080
s = scenes.interior("bathroom", 1.0, W, H)
sc = api.Scene(s); rt = api.RayTracer(sc)
found = []
for k in range(96):
    rt.clear(); rt.render(api.PATH, 1, 1, k)
    film, _ = rt.film()
    ys, xs = np.nonzero(~np.isfinite(film).all(axis=2))
    found += [(int(x), int(y), k) for x, y in zip(xs, ys)]
print("non-finite samples:", len(found))
o = orc.OracleScene(s)
L = orc.lib()
L.orc_trace_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
for x, y, k in found[:12]:
    log = np.zeros((8, 20), np.float32); L3 = np.zeros(3, np.float32)
    n = L.orc_trace_path(o.h, x, y, 1, k, log.ctypes.data, 8, L3.ctypes.data)
    print(f"pixel ({x},{y}) sample {k}: L = {L3}")
    for r in log[:n]:
        print(f"   mat {int(r[0])} T {r[1:4]} direct {r[4:7]} f {r[7:10]} pdf {r[10]:.6g} wi {r[11:14]} wo {r[14:17]} sN {r[17:20]}")
