"""Device build of the reference-topology tree (csrc/bvh_build.hip) against the host build: same leaf
order, same tree statistics, same wide tree size, and bit-identical traversal results."""
import time

import numpy as np
import pytest

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def both(s):
    return api.Scene(s, device_build=False), api.Scene(s, device_build=True)


def same_walk_nodes(h, d):
    """What the kernels walk, from a host-built and a device-built scene: the reference-topology nodes node for node (the two
    builders number them differently: compared through the triangle range each node stands for), and the arrays derived from
    them on the device (csrc/walk_tree.hip: 8-wide nodes, walk-ordered records, gate boxes) byte for byte.  Box planes compare as
    integers, except that +0.0 and -0.0 count as the same plane (a min / max over vertices that hold both zeros returns
    whichever it saw last on the host and the smaller key on the device; no test reads the sign of a zero)."""
    def same_planes(x, y):
        x, y = x.view(np.uint32), y.view(np.uint32)
        z = ((x & 0x7FFFFFFF) == 0) & ((y & 0x7FFFFFFF) == 0)
        return np.array_equal(x[~z], y[~z])
    ra, rb = h.ref_nodes(), d.ref_nodes()
    assert len(ra) == len(rb)
    ka, kb = np.lexsort((-ra["sub_count"], ra["sub_first"])), np.lexsort((-rb["sub_count"], rb["sub_first"]))   # DFS pre-order
    ra, rb = ra[ka], rb[kb]
    for f in ("sub_first", "sub_count", "first", "count"):
        assert np.array_equal(ra[f], rb[f]), f
    assert np.array_equal(ra["left"] < 0, rb["left"] < 0) and np.array_equal(ra["right"] < 0, rb["right"] < 0)
    boxed = ra["sub_count"] >= 2                                  # a single-triangle leaf keeps upstream's never-initialised box
    assert same_planes(ra["lo"][boxed], rb["lo"][boxed]) and same_planes(ra["hi"][boxed], rb["hi"][boxed])
    (na, ga, ta), (nb, gb, tb) = h.walk_nodes(), d.walk_nodes()
    assert len(na) == len(nb)
    for f in na.dtype.names:
        assert same_planes(na["org"], nb["org"]) if f == "org" else np.array_equal(na[f], nb[f]), f
    assert np.array_equal(ta.view(np.uint8), tb.view(np.uint8))
    assert same_planes(ga["lo"], gb["lo"]) and same_planes(ga["hi"], gb["hi"])
    return len(na)


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(64, 64, with_specular=True),
    lambda: scenes.interior("bathroom", 0.15, 160, 90),
    lambda: scenes.interior("living-room", 0.1, 160, 90),
    lambda: scenes.random_soup(5000, seed=4),
    lambda: scenes.axis_aligned_grid(6),          # coincident centroids / empty bins / ties
], ids=["cornell-spec", "bathroom", "living-room", "soup", "grid"])
def test_device_build_equals_host_build(make):
    s = make()
    h, d = both(s)
    ih, idv = h.info(), d.info()
    for k in ("n_tris", "n_lights", "bvh2_nodes", "bvh2_depth", "bvh2_leaves", "bvh2_max_leaf", "wide_nodes", "wide_max_stack", "wide_levels"):
        assert ih[k] == idv[k], k
    assert np.array_equal(h.leaf_order(), d.leaf_order())
    assert same_walk_nodes(h, d) == ih["wide_nodes"]
    # the byte per triangle a closest-hit walk files its hit by: BSDF class of the triangle's material, 0x80 for an emitter
    cls_of = {scenes.DIFFUSE: 0, scenes.ORENNAYAR: 0, scenes.MIRROR: 1, scenes.GLASS: 2, scenes.CONDUCTOR: 3, scenes.DIELECTRIC: 4, scenes.PLASTIC: 5}
    per_mat = np.array([cls_of[m.type] | (0x80 if max(m.emission) > 0 else 0) for m in s.materials], np.uint8)
    want_cls = per_mat[s.tri_mat[d.leaf_order()]]
    assert np.array_equal(d.tri_classes(), want_cls) and np.array_equal(h.tri_classes(), want_cls)
    rays = np.concatenate([scenes.bounce_rays(s, 100000, 5), scenes.primary_rays(s, 50000, 6) if s.cam_P is not None else scenes.bounce_rays(s, 50000, 7)])
    for mode in (0, 1):
        a, b = h.traverse(rays, mode), d.traverse(rays, mode)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    segs = scenes.shadow_segments(s, 100000, 8)
    assert np.array_equal(h.visible(segs), d.visible(segs))
    h.close(); d.close()


def test_long_leaves_are_cut_the_same_way():
    """More than RT_LEAF_MAX (3) triangles whose centroids coincide cannot be split (Geometry.h:418's axis stays -1) and end up in one leaf;
    the wide tree cuts such a range into a small subtree of slots.  Host and device must cut alike."""
    rng = np.random.default_rng(6)
    s = scenes.random_soup(600, seed=8)
    v = s.verts.copy()
    for base, cnt in ((0, 40), (100, 17), (200, 300)):           # fans around one centroid: rotate a triangle about its centre
        c = np.array([0.3 * base / 100, 1.0, -0.5], np.float32)
        for k in range(cnt):
            a = rng.uniform(0, 2 * np.pi); r = rng.uniform(0.1, 0.5)
            d3 = np.array([[np.cos(a + j * 2 * np.pi / 3) * r, np.sin(a + j * 2 * np.pi / 3) * r, 0] for j in range(3)], np.float32)
            v[base + k, :, :3] = c + d3 - d3.mean(0)
    s.verts = v
    h, d = both(s)
    assert h.info()["bvh2_max_leaf"] == d.info()["bvh2_max_leaf"]
    assert np.array_equal(h.leaf_order(), d.leaf_order())
    same_walk_nodes(h, d)
    rays = scenes.bounce_rays(s, 100000, 5)
    assert np.array_equal(h.traverse(rays).view(np.uint8), d.traverse(rays).view(np.uint8))
    h.close(); d.close()


@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_tiny_scenes(n):
    s = scenes.random_soup(n, seed=2)
    h, d = both(s)
    assert h.info()["bvh2_nodes"] == d.info()["bvh2_nodes"] and np.array_equal(h.leaf_order(), d.leaf_order())
    same_walk_nodes(h, d)
    rays = np.random.default_rng(1).normal(size=(4096, 6)).astype(np.float32)
    assert np.array_equal(h.traverse(rays).view(np.uint8), d.traverse(rays).view(np.uint8))
    h.close(); d.close()


def test_full_size_bathroom_build_and_time():
    s = scenes.interior("bathroom", 1.0, 1920, 1080)
    t = time.perf_counter(); h = api.Scene(s, device_build=False); th = time.perf_counter() - t
    t = time.perf_counter(); d = api.Scene(s, device_build=True); td = time.perf_counter() - t
    t = time.perf_counter(); d2 = api.Scene(s, device_build=True); td2 = time.perf_counter() - t; d2.close()   # second time: no first-launch cost
    print(f"scene create: host build {th:.2f} s, device build {td:.2f} s, again {td2:.3f} s (n_tris {s.ntris})")
    ih, idv = h.info(), d.info()
    for k in ("n_tris", "n_lights", "bvh2_nodes", "bvh2_depth", "bvh2_leaves", "bvh2_max_leaf", "wide_nodes", "wide_max_stack", "wide_levels"):
        assert ih[k] == idv[k], k
    assert np.array_equal(h.leaf_order(), d.leaf_order())
    assert same_walk_nodes(h, d) == ih["wide_nodes"]
    assert td2 < 1.0                                       # all-device set-up: 0.05 s on an idle box (0.19 s with the host-side records and collapse of round 1); generous, the box may be loaded
    assert idv["wide_max_stack"] <= 32 and idv["wide_nodes"] < idv["n_tris"] // 4
    rays = scenes.bounce_rays(s, 1 << 20, 3)
    assert np.array_equal(h.traverse(rays).view(np.uint8), d.traverse(rays).view(np.uint8))
    h.close(); d.close()


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(64, 64),
    lambda: scenes.random_soup(5000, seed=9),
    lambda: scenes.axis_aligned_grid(7),
    lambda: scenes.interior("bathroom", 0.1, 320, 180),
    lambda: scenes.random_soup(1, seed=2),
], ids=["cornell", "soup", "lattice", "bathroom", "one-triangle"])
def test_walk_nodes_bound_what_they_stand_for(make):
    """The 80-B 8-wide nodes the kernels read (rt_scene_debug_walk_nodes) against the reference tree they are derived from
    (rt_scene_debug_ref_nodes) and the triangles: every slot's grid box contains the vertices of every triangle below it (a leaf
    slot's within one grid step); inner children are numbered consecutively from child_base in slot order, leaf triangles
    consecutively from tri_base; every triangle sits in exactly one leaf slot, and its record in the walk-ordered array is its
    DFS-order record with the rank in place of orig_id; every triangle's gate box is the box of the last reference node upstream
    really tests on the way down to it (its leaf's, or the parent's for a leaf of one triangle: Geometry.h:401, :441-471 --
    recomputed here from the reference nodes); the stack bound holds for the tree's shape."""
    s = make()
    sc = api.Scene(s)
    ref = sc.ref_nodes()
    q, gate, trisw = sc.walk_nodes()
    info = sc.info()
    order = sc.leaf_order()
    v = s.verts[order][:, :, :3].astype(np.float64)            # triangles in DFS leaf order
    # gate boxes, from the reference tree
    is_leaf = (ref["left"] < 0) & (ref["right"] < 0)
    seen = np.zeros(len(order), bool)
    def put(first, cnt, lo, hi):
        assert (gate["lo"][first:first + cnt] == lo).all() and (gate["hi"][first:first + cnt] == hi).all()
        assert not seen[first:first + cnt].any()
        seen[first:first + cnt] = True
    if is_leaf[0]:
        big = np.float32(3.402823466e38)
        if ref["count"][0] >= 2: put(0, int(ref["count"][0]), ref["lo"][0], ref["hi"][0])
        else: put(0, int(ref["count"][0]), -big, big)
    for i in np.nonzero(~is_leaf)[0]:
        for c in (int(ref["left"][i]), int(ref["right"][i])):
            if c >= 0 and is_leaf[c]:
                own = ref["count"][c] >= 2
                put(int(ref["first"][c]), int(ref["count"][c]), ref["lo"][c] if own else ref["lo"][i], ref["hi"][c] if own else ref["hi"][i])
                tv = v[ref["first"][c]:ref["first"][c] + ref["count"][c]].reshape(-1, 3)     # and that box does hold its triangles
                g = gate[int(ref["first"][c])]
                assert (g["lo"] <= tv.min(0)).all() and (g["hi"] >= tv.max(0)).all()
    assert seen.all(), "a triangle without a gate box"
    # walk-ordered records
    assert sorted(trisw["rank"].tolist()) == list(range(len(order)))
    assert np.array_equal(trisw["v"].astype(np.float64), v[trisw["rank"]])
    # the 8-wide tree, top down: (node, lo, hi) bounds of everything below come back up
    n = len(q)
    assert n == info["wide_nodes"]
    org = q["org"].astype(np.float64)
    step = np.ldexp(1.0, q["exp"].astype(np.int64))
    placed = np.zeros(len(order), bool)
    entered = np.zeros(n, bool)
    lo_below = np.full((n, 3), np.inf); hi_below = np.full((n, 3), -np.inf)
    need = np.zeros(n, np.int64)
    for i in range(n - 1, -1, -1):                             # children have larger numbers than their parents (level order)
        rank_in, off, n_inner, n_leaf, worst = 0, 0, 0, 0, 0
        for sl in range(8):
            m = int(q["meta"][i, sl])
            glo = org[i] + q["lo"][i, :, sl] * step[i]; ghi = org[i] + q["hi"][i, :, sl] * step[i]
            if m == 0:
                assert not (q["imask"][i] >> sl) & 1
                continue
            if (m & 0x1F) >= 24:                               # inner child
                assert m == (0x20 | (24 + sl)) and (q["imask"][i] >> sl) & 1
                c = int(q["child_base"][i]) + rank_in
                rank_in += 1; n_inner += 1
                assert i < c < n and not entered[c]
                entered[c] = True
                assert (glo <= lo_below[c]).all() and (ghi >= hi_below[c]).all(), (i, sl)
                lo_below[i] = np.minimum(lo_below[i], lo_below[c]); hi_below[i] = np.maximum(hi_below[i], hi_below[c])
                worst = max(worst, need[c])
            else:
                assert not (q["imask"][i] >> sl) & 1
                cnt = {1: 1, 3: 2, 7: 3}[m >> 5]
                assert (m & 0x1F) == off and off + cnt <= 24
                pos = int(q["tri_base"][i]) + off
                off += cnt; n_leaf += 1
                assert not placed[pos:pos + cnt].any()
                placed[pos:pos + cnt] = True
                tv = trisw["v"][pos:pos + cnt].astype(np.float64).reshape(-1, 3)
                assert (glo <= tv.min(0)).all() and (ghi >= tv.max(0)).all(), (i, sl)
                assert (tv.min(0) - glo <= step[i]).all() and (ghi - tv.max(0) <= step[i]).all(), (i, sl)
                lo_below[i] = np.minimum(lo_below[i], tv.min(0)); hi_below[i] = np.maximum(hi_below[i], tv.max(0))
        need[i] = (1 if n_inner >= 2 else 0) + (1 if n_leaf else 0) + worst
    assert placed.all() and entered[1:].all() and not entered[0]
    assert info["wide_max_stack"] == need[0] + 2
    sc.close()
