"""How often does the conservative wide walk end on a hit that upstream's box test of the hit's gate box refuses (rt_settle),
and are those rays still answered exactly?  Rays are aimed at triangle vertices and edge points -- the places where a hit sits
on the surface of its leaf's box -- and the wide walk (mode 0) is compared with the reference-order walk (mode 1).
   python tools/probe_refused.py [scene] [n]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (its HIP runtime must be loaded first)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, scenes  # noqa: E402


def aimed_rays(s, n, seed):
    rng = np.random.default_rng(seed)
    v = s.verts[:, :, :3].astype(np.float32)
    tri = rng.integers(0, len(v), n)
    w = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    kind = rng.integers(0, 3, n)
    w[kind == 0] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (kind == 0).sum())]          # a vertex
    e = kind == 1                                                                                   # a point on an edge
    w[e, rng.integers(0, 3, e.sum())] = 0
    w[e] /= w[e].sum(1, keepdims=True)
    target = (v[tri] * w[:, :, None]).sum(1).astype(np.float32)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 3)).astype(np.float32)
    d = (target - o).astype(np.float32)
    return np.concatenate([o, d], 1), np.concatenate([o, target], 1)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
    s = {"soup": lambda: scenes.random_soup(3000, seed=5), "cornell": lambda: scenes.cornell(256, 256),
         "lattice": lambda: scenes.axis_aligned_grid(6), "bathroom": lambda: scenes.interior("bathroom", 0.06, 320, 180)}[name]()
    sc = api.Scene(s)
    sc.counters_enable(True)
    rays, segs = aimed_rays(s, n, 7)
    sc.counters()
    a = sc.traverse(rays)
    c = sc.counters()
    b = sc.traverse(rays, mode=1)
    same = all((a[k].view(np.uint32) == b[k].view(np.uint32)).all() for k in ("ID", "t", "alpha", "beta", "gamma"))
    va, vb = sc.visible(segs), sc.visible(segs, mode=1)
    c2 = sc.counters()
    print(name, s.ntris, "triangles;", n, "aimed rays: gate refusals", c["gate_refused"], "closest,", c2["any_gate_refused"], "any-hit; hits",
          int((a["t"] < 3e38).sum()), "occluded", int((va == 0).sum()), "; wide walk == reference-order walk:", same, bool((va == vb).all()))


if __name__ == "__main__":
    main()
