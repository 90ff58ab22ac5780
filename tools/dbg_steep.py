"""Debug: steep rays (one direction component 1e-17) on the soup scaled by 1e12: which rays differ between the wide walk, the
reference-order walk (mode 1) and the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import torch  # noqa
from raytracer_amd import api, scenes
import orc
from util import aimed_rays
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1e12
base = scenes.random_soup(4000, seed=12)
v = base.verts.copy()
v[:, :, :3] = (v[:, :, :3] * np.float32(scale)).astype(np.float32)
v[:200, :, 2] = 0
s = scenes.SceneData("scaled", v, base.tri_mat.copy(), base.materials, base.textures)
scenes.set_camera(s, (0, 0, 12 * scale), (0, 0, 0), (0, 1, 0), 45.0, 64, 64)
o, sc = orc.OracleScene(s), api.Scene(s)
rays, segs = aimed_rays(s, 150_000, seed=3)
steep = rays[:5000].copy()
steep[np.arange(5000), 3 + np.arange(5000) % 3] = np.where(np.arange(5000) % 2 == 0, 1e-17, -1e-17).astype(np.float32)
want = o.traverse(steep, threads=8)
g0 = sc.traverse(steep, 0); g1 = sc.traverse(steep, 1)
bad = np.nonzero((g0["t"] < 3e38) != (want["t"] < 3e38))[0]
print("differ wide vs oracle:", len(bad), " mode1 vs oracle:", int(((g1["t"] < 3e38) != (want["t"] < 3e38)).sum()))
np.set_printoptions(precision=9, linewidth=200)
for i in bad[:6]:
    print(i, "ray", steep[i], "\n   wide", g0[i], "\n   ref-order", g1[i], "\n   oracle", want[i])
