"""What would a cost-optimal 8-wide collapse be worth?  Surface-area cost of the walk tree as built (walk_tree.hip: open the
largest slot first) against the optimum of the dynamic program of Ylitie, Karras & Laine 2017 (section 3.1) over the same
reference-topology binary tree, leaves cut down to single triangles, a leaf slot holding at most 3.
  cost = c_node * sum over 8-wide nodes of A(node box) / A(root) + c_tri * sum over leaf slots of triangles * A(slot box) / A(root)
i.e. the expected node steps and triangle tests of a long random ray that nothing stops.
Usage: python tools/exp_collapse.py [--detail 1.0] [--scene bathroom] [--c-node 2.0] [--c-tri 1.0]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--detail", type=float, default=1.0)
ap.add_argument("--scene", default="bathroom")
ap.add_argument("--c-node", type=float, default=2.0)
ap.add_argument("--c-tri", type=float, default=1.0)
a = ap.parse_args()
s = scenes.interior(a.scene, a.detail, 1920, 1080)
sc = api.Scene(s, device_build=True)
info = sc.info()
ref = sc.ref_nodes()
nodes8, gate, trisw = sc.walk_nodes()
nt = info["n_tris"]
print({k: info[k] for k in ("n_tris", "bvh2_nodes", "wide_nodes", "wide_levels", "wide_max_stack")}, flush=True)


def area(lo, hi):
    d = np.maximum(hi.astype(np.float64) - lo.astype(np.float64), 0.0)
    return 2.0 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])


A0 = float(area(ref["lo"][0], ref["hi"][0]))

# ---- the tree as built: every used slot of every node, boxes off the 8-bit grid
scale = np.ldexp(1.0, nodes8["exp"].astype(np.int32)).astype(np.float64)                    # (n, 3)
lo = nodes8["org"].astype(np.float64)[:, :, None] + nodes8["lo"].astype(np.float64) * scale[:, :, None]   # (n, 3, 8)
hi = nodes8["org"].astype(np.float64)[:, :, None] + nodes8["hi"].astype(np.float64) * scale[:, :, None]
d = np.maximum(hi - lo, 0.0)
sa = 2.0 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]) / A0                     # (n, 8)
meta = nodes8["meta"]
inner = (meta & 0x20) != 0
inner &= (meta & 0x1F) >= 24
used = meta != 0
leaf = used & ~inner
cnt = np.select([(meta >> 5) == 1, (meta >> 5) == 3, (meta >> 5) == 7], [1, 2, 3], 0)
built_nodes = 1.0 + float((sa * inner).sum())
built_tris = float((sa * leaf * cnt).sum())
print(f"as built : {len(nodes8)} nodes, expected node steps {built_nodes:.2f}, triangle tests {built_tris:.2f}, "
      f"cost {a.c_node * built_nodes + a.c_tri * built_tris:.2f}   (slot boxes on the 8-bit grid)", flush=True)

# ---- unit tree: reference nodes + halving subtrees under leaves of 2 or more triangles
tl = np.empty((nt, 3)); th = np.empty((nt, 3))
v = trisw["v"].astype(np.float64)
tl[trisw["rank"]] = v.min(axis=1); th[trisw["rank"]] = v.max(axis=1)


def range_bounds(first, count):
    """own bounds of triangle ranges [first, first + count) (DFS ranks)"""
    n = len(first)
    idx = np.empty(2 * n, np.int64)
    idx[0::2] = first; idx[1::2] = first + count
    tlp = np.vstack([tl, np.full((1, 3), np.inf)]); thp = np.vstack([th, np.full((1, 3), -np.inf)])
    order = np.argsort(first, kind="stable")
    # reduceat needs non-decreasing starts for the pairs trick; ranges are disjoint or nested per call level, so sort by start
    f, c = first[order], count[order]
    idx[0::2] = f; idx[1::2] = f + c
    lo_ = np.minimum.reduceat(tlp, idx, axis=0)[0::2]
    hi_ = np.maximum.reduceat(thp, idx, axis=0)[0::2]
    out_lo = np.empty((n, 3)); out_hi = np.empty((n, 3))
    out_lo[order] = lo_; out_hi[order] = hi_
    return out_lo, out_hi


nref = len(ref)
is_leaf_ref = (ref["left"] < 0) & (ref["right"] < 0)
L = [ref["left"].astype(np.int64).copy()]; R = [ref["right"].astype(np.int64).copy()]
first = [np.where(is_leaf_ref, ref["first"], ref["sub_first"]).astype(np.int64)]
count = [np.where(is_leaf_ref, ref["count"], ref["sub_count"]).astype(np.int64)]
ar = [area(ref["lo"], ref["hi"]) / A0]
# leaves are bounded by their own triangles (as the walk tree bounds its leaf slots)
lf = np.nonzero(is_leaf_ref)[0]
blo, bhi = range_bounds(first[0][lf], count[0][lf])
ar[0][lf] = area(blo, bhi) / A0
total = nref
parents = lf[count[0][lf] >= 2]      # units still to be halved
pf, pc = first[0][parents], count[0][parents]
L0, R0 = L[0], R[0]
L0[is_leaf_ref] = -1; R0[is_leaf_ref] = -1
chunks = []
while len(parents):
    h = (pc + 1) // 2
    cf = np.concatenate([pf, pf + h]); cc = np.concatenate([h, pc - h])
    ids = total + np.arange(len(cf), dtype=np.int64)
    blo, bhi = range_bounds(cf, cc)
    chunks.append((parents, ids[:len(pf)], ids[len(pf):]))
    L.append(np.full(len(cf), -1, np.int64)); R.append(np.full(len(cf), -1, np.int64))
    first.append(cf); count.append(cc); ar.append(area(blo, bhi) / A0)
    total += len(cf)
    again = cc >= 2
    parents, pf, pc = ids[again], cf[again], cc[again]
Lc = np.concatenate(L); Rc = np.concatenate(R); cn = np.concatenate(count); An = np.concatenate(ar)
for par, li, ri in chunks:
    Lc[par] = li; Rc[par] = ri
N = total
print(f"unit tree: {N} units ({nref} reference nodes)", flush=True)

# depth levels
depth = np.zeros(N, np.int32)
frontier = np.array([0], np.int64)
levels = []
while len(frontier):
    levels.append(frontier)
    inn = frontier[Lc[frontier] >= 0]
    nxt = np.concatenate([Lc[inn], Rc[inn]])
    depth[nxt] = len(levels)
    frontier = nxt
print(f"{len(levels)} levels", flush=True)

INF = np.inf
C = np.full((N, 8), INF); Nn = np.zeros((N, 8)); Nt = np.zeros((N, 8))   # column i: a forest of at most i slots; column 0 unused
for lv in reversed(levels):
    single = lv[Lc[lv] < 0]
    if len(single):
        C[single, 1:] = (An[single] * cn[single] * a.c_tri)[:, None]
        Nt[single, 1:] = (An[single] * cn[single])[:, None]
    inn = lv[Lc[lv] >= 0]
    if not len(inn):
        continue
    l, r = Lc[inn], Rc[inn]
    m = len(inn)
    D = np.full((m, 9), INF); Dn = np.zeros((m, 9)); Dt = np.zeros((m, 9))
    for j in range(2, 9):
        for k in range(1, j):
            if k > 7 or j - k > 7:
                continue
            c = C[l, k] + C[r, j - k]
            better = c < D[:, j]
            D[better, j] = c[better]
            Dn[better, j] = (Nn[l, k] + Nn[r, j - k])[better]
            Dt[better, j] = (Nt[l, k] + Nt[r, j - k])[better]
    as_node = An[inn] * a.c_node + D[:, 8]
    as_leaf = np.where(cn[inn] <= 3, An[inn] * cn[inn] * a.c_tri, INF)
    leaf_wins = as_leaf <= as_node
    C[inn, 1] = np.where(leaf_wins, as_leaf, as_node)
    Nn[inn, 1] = np.where(leaf_wins, 0.0, An[inn] + Dn[:, 8])
    Nt[inn, 1] = np.where(leaf_wins, An[inn] * cn[inn], Dt[:, 8])
    for i in range(2, 8):
        keep = C[inn, i - 1] <= D[:, i]
        C[inn, i] = np.where(keep, C[inn, i - 1], D[:, i])
        Nn[inn, i] = np.where(keep, Nn[inn, i - 1], Dn[:, i])
        Nt[inn, i] = np.where(keep, Nt[inn, i - 1], Dt[:, i])
# the root is a node whatever it costs
root_l, root_r = Lc[0], Rc[0]
best = (INF, 0, 0)
for k in range(1, 8):
    c = C[root_l, k] + C[root_r, 8 - k]
    if c < best[0]:
        best = (c, Nn[root_l, k] + Nn[root_r, 8 - k], Nt[root_l, k] + Nt[root_r, 8 - k])
opt_nodes, opt_tris = 1.0 + best[1], best[2]
print(f"optimum  : expected node steps {opt_nodes:.2f}, triangle tests {opt_tris:.2f}, cost {a.c_node * opt_nodes + a.c_tri * opt_tris:.2f}   (exact boxes)")
