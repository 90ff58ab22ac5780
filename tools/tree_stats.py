"""Slot statistics of the 4-wide tree of a scene: used slots per node, triangles per leaf slot.   python tools/tree_stats.py [bathroom|living-room] [detail]"""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "bathroom"
s = scenes.interior(name, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, 1920, 1080)
sc = api.Scene(s)
w = sc.wide_nodes()
ref = w["child"]
used = ref != 0xFFFFFFFF
leaf = used & ((ref & 0x80000000) != 0)
cnt = ((ref >> 27) & 15) + 1
print(name, s.ntris, "triangles", len(w), "nodes; used slots per node:", np.bincount(used.sum(1), minlength=5).tolist(),
      "; leaf slots by triangle count:", np.bincount(cnt[leaf], minlength=8).tolist(), "; leaf slots per node:", np.bincount(leaf.sum(1), minlength=5).tolist())
full = used.sum(1) == 4
two = (leaf & (cnt == 2))
print("nodes with a free slot and a 2-triangle leaf:", int(((~full) & two.any(1)).sum()), "; 2-triangle leaf slots in full nodes:", int(two[full].sum()), "of", int(two.sum()))
# depth of the wide tree: nodes are numbered level by level, parents before children
depth = np.zeros(len(w), np.int32)
for k in range(4):
    pass
inner = used & ~leaf
order_ok = True
for i in range(len(w)):
    for k in range(4):
        if inner[i, k]:
            depth[ref[i, k]] = depth[i] + 1
ld = np.repeat(depth[:, None], 4, 1)[leaf]
area = 2 * ((w["box"][:, 3] - w["box"][:, 0]) * (w["box"][:, 4] - w["box"][:, 1]) + (w["box"][:, 4] - w["box"][:, 1]) * (w["box"][:, 5] - w["box"][:, 2]) + (w["box"][:, 3] - w["box"][:, 0]) * (w["box"][:, 5] - w["box"][:, 2]))
la = area[leaf]
print("wide tree depth: max", int(depth.max()), "mean over leaf slots", float(ld.mean()), "area-weighted", float((ld * la).sum() / la.sum()), "; log4(nodes)", np.log(len(w)) / np.log(4))
print("leaf-slot depth histogram:", np.bincount(ld).tolist())
root_area = area[0][used[0]].max()
# SAH-style expected visits for a random ray: sum over inner child slots of area / root area (+1 for the root)
print("expected node visits (sum of inner-slot areas / scene area):", 1 + float(area[inner].sum()) / float(root_area), " expected leaf-slot visits:", float(la.sum()) / float(root_area))
# where triangle tests go for random rays: a leaf slot is entered in proportion to the area of its (tight) box, and then tests all its
# triangles; a triangle is hit in proportion to twice its own area.  waste = tests per hit, by triangle size.
order = sc.leaf_order()
v = s.verts[order][:, :, :3].astype(np.float64)
tri_area = 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
tests = np.zeros(len(v)); slot_of = np.zeros(len(v), np.int64)
for i, k in zip(*np.nonzero(leaf)):
    r = int(ref[i, k]); first, c = r & 0x07FFFFFF, ((r >> 27) & 15) + 1
    tv = v[first:first + c].reshape(-1, 3)
    ext = tv.max(0) - tv.min(0)
    a = 2 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2])
    tests[first:first + c] = a
print("random rays: triangle tests per true hit =", tests.sum() / (2 * tri_area).sum())
o2 = np.argsort(-tests)
cum = np.cumsum(tests[o2]) / tests.sum()
for frac in (0.0001, 0.001, 0.01, 0.1):
    n = max(1, int(len(v) * frac))
    print(f"  the {n} triangles ({frac:.2%}) with the largest slot boxes take {cum[n - 1]:.1%} of all tests; their own hit share {(2 * tri_area[o2[:n]]).sum() / (2 * tri_area).sum():.1%}")
