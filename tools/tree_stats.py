"""Slot statistics of the 8-wide walk tree of a scene.   python tools/tree_stats.py [bathroom|living-room] [detail]"""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "bathroom"
s = scenes.interior(name, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, 1920, 1080)
sc = api.Scene(s)
q, gate, trisw = sc.walk_nodes()
meta = q["meta"].astype(np.int64)
used = meta != 0
inner = used & ((meta & 0x1F) >= 24)
leaf = used & ~inner
cnt = np.zeros_like(meta); cnt[leaf] = np.array([0, 1, 0, 2, 0, 0, 0, 3])[meta[leaf] >> 5]
print(name, s.ntris, "triangles,", len(q), "nodes, info", {k: v for k, v in sc.info().items() if k.startswith("wide")})
print("used slots per node:", np.bincount(used.sum(1), minlength=9).tolist(), "mean", used.sum(1).mean())
print("inner children per node:", np.bincount(inner.sum(1), minlength=9).tolist())
print("leaf slots per node:", np.bincount(leaf.sum(1), minlength=9).tolist())
print("triangles per leaf slot:", np.bincount(cnt[leaf], minlength=4).tolist(), "; triangles per node (own leaf slots):", np.bincount(cnt.sum(1), minlength=25).tolist())
only_leaf = inner.sum(1) == 0
print("nodes without inner children:", int(only_leaf.sum()), "their used-slot histogram:", np.bincount(used[only_leaf].sum(1), minlength=9).tolist(),
      "their triangle-count histogram:", np.bincount(cnt[only_leaf].sum(1), minlength=25).tolist())
# expected visits for random rays ~ sum of box areas of inner slots / root area
step = np.ldexp(1.0, q["exp"].astype(np.int64))
ext = (q["hi"].astype(np.float64) - q["lo"].astype(np.float64)) * step[:, :, None]          # (n, 3, 8)
area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 0] * ext[:, 2])
root = area[0][used[0]].sum()
fill = used.sum(1)
print("area-weighted visits: inner slots", area[inner].sum() / root, "leaf slots", area[leaf].sum() / root)
# visits weighted by the area of the node's own box (union of its slots ~ parent's inner slot): where do visited nodes sit by fill?
node_area = np.zeros(len(q)); node_area[0] = root
for i in range(len(q)):
    k = 0
    for sl in range(8):
        if inner[i, sl]:
            node_area[q["child_base"][i] + k] = area[i, sl]; k += 1
w = node_area / node_area.sum()
print("share of (area-weighted) node visits by used slots:", [round(float(w[fill == f].sum()), 3) for f in range(9)], "mean fill of a visited node", float((w * fill).sum()))
# what the 8-bit grid costs: box area (~ visit probability for random rays) of the slots as the walk sees them against their exact boxes
glo = q["org"][:, :, None].astype(np.float64) + q["lo"].astype(np.float64) * step[:, :, None]
ghi = q["org"][:, :, None].astype(np.float64) + q["hi"].astype(np.float64) * step[:, :, None]
tv = trisw["v"].astype(np.float64)                       # (n, 3 verts, 3)
tlo, thi = tv.min(1), tv.max(1)
ex_area_leaf, gr_area_leaf = 0.0, 0.0
nodes_i, slots_i = np.nonzero(leaf)
off = (meta[nodes_i, slots_i] & 0x1F); c = cnt[nodes_i, slots_i]
pos = q["tri_base"][nodes_i].astype(np.int64) + off
lo_e = tlo[pos].copy(); hi_e = thi[pos].copy()
for k in (1, 2):
    m = c > k
    lo_e[m] = np.minimum(lo_e[m], tlo[pos[m] + k]); hi_e[m] = np.maximum(hi_e[m], thi[pos[m] + k])
e = hi_e - lo_e
ex = 2 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 0] * e[:, 2])
g = (ghi - glo)[nodes_i, :, slots_i]
gr = 2 * (g[:, 0] * g[:, 1] + g[:, 1] * g[:, 2] + g[:, 0] * g[:, 2])
print("leaf slots: grid-box area / exact-box area, summed:", gr.sum() / ex.sum(), " median per slot:", float(np.median(gr / np.maximum(ex, 1e-30))),
      " (triangle-count weighted tests: grid", float((gr * c).sum() / root), "exact", float((ex * c).sum() / root), ")")
