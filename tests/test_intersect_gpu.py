"""GPU: Scene::traverse / Scene::visible through the C ABI against the oracle and the reference fixtures.
Bar: bit-exact hit/miss, triangle ID, t, alpha, beta, gamma -- no tolerance and no excluded rays."""
import numpy as np
import pytest

import orc
from raytracer_amd import api, scenes
from raytracer_amd.abi import HIT_DTYPE
from util import GOLDEN_SCENES, aimed_rays, assert_hits_bitexact, axis_rays, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GOLDEN_SCENES)
@pytest.mark.parametrize("mode", [0, 1])
def test_reference_fixtures(name, mode):
    s, g, want = load_golden(name)
    sc = api.Scene(s)
    assert (sc.leaf_order() == g["leaf_order"]).all()
    assert_hits_bitexact(sc.traverse(g["rays"], mode), want, f"{name} mode {mode}")
    assert (sc.visible(g["segs"], mode) == g["visible"]).all()
    sc.close()


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(256, 256, with_specular=True),
    lambda: scenes.interior("bathroom", 0.25, 640, 360),
    lambda: scenes.interior("living-room", 0.2, 640, 360),
    lambda: scenes.random_soup(20000, seed=3),
    lambda: scenes.axis_aligned_grid(10),
], ids=["cornell-spec", "bathroom", "living-room", "soup", "lattice"])
def test_against_oracle(make):
    s = make()
    o, sc = orc.OracleScene(s), api.Scene(s)
    n = 200_000
    rays = np.concatenate([scenes.primary_rays(s, n, 1), scenes.bounce_rays(s, n, 2), axis_rays(s, 20_000)])
    want = o.traverse(rays, threads=8)
    assert_hits_bitexact(sc.traverse(rays, 0), want, s.name + " wide")
    assert_hits_bitexact(sc.traverse(rays, 1), want, s.name + " reference-order")
    segs = scenes.shadow_segments(s, n, 3)
    vis = o.visible(segs, threads=8)
    assert (sc.visible(segs, 0) == vis).all()
    assert (sc.visible(segs, 1) == vis).all()
    sc.close()


def test_cornell_fixed_ray_set_of_the_survey():
    """SURVEY.md 8(d), BASELINE config 2: Cornell box at 1024x1024, 2^20 primary rays + 2^20 cosine-bounce rays + 2^20 NEE
    segments from seed 1, every one bit-exact.  Primary pin: the committed digest of the compiled reference's answers
    (tests/golden/cornell_survey_digest.json, made by tests/golden/make_golden.py digest); the per-ray comparison against the
    oracle (or the reference build, when it travelled with the tree) says where a mismatch would be."""
    import json
    from util import GOLDEN, hits_digest
    s = scenes.cornell(1024, 1024)
    ref = orc.RefScene(s) if orc.have_ref() else orc.OracleScene(s)
    sc = api.Scene(s)
    n = 1 << 20
    rays = np.concatenate([scenes.primary_rays(s, n, 1), scenes.bounce_rays(s, n, 1)])
    pin = json.load(open(f"{GOLDEN}/cornell_survey_digest.json"))
    got0 = sc.traverse(rays, 0)
    assert hits_digest(got0, sc.visible(scenes.shadow_segments(s, n, 1), 0)) == pin["sha256"]
    assert int((got0["t"] < 3e38).sum()) == pin["hits"]
    want = ref.traverse(rays)
    assert_hits_bitexact(got0, want, "cornell 2^21 rays, wide")
    assert_hits_bitexact(sc.traverse(rays, 1), want, "cornell 2^21 rays, reference order")
    segs = scenes.shadow_segments(s, n, 1)
    vis = ref.visible(segs)
    assert np.array_equal(sc.visible(segs, 0), vis) and np.array_equal(sc.visible(segs, 1), vis)
    assert 0 < int(vis.sum()) < n
    sc.close()


def test_primary_rays_from_device_camera_match_oracle():
    s = scenes.cornell(512, 512)
    o, sc = orc.OracleScene(s), api.Scene(s)
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 512, (50_000, 2)).astype(np.float32)
    a, b = sc.generate_rays(xy), o.generate_rays(xy)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()  # Camera::generateRay, bit for bit
    sc.close()


def test_edge_cases():
    s = scenes.cornell(64, 64)
    sc = api.Scene(s)
    assert len(sc.traverse(np.zeros((0, 6), np.float32))) == 0          # empty batch
    assert len(sc.visible(np.zeros((0, 6), np.float32))) == 0
    r = sc.traverse(np.array([[0, 1, 0, 0, 0, 0]], np.float32))          # zero direction: every denom == 0 -> miss
    assert r["t"][0] > 3e38 and r["ID"][0] == 0xFFFFFFFF
    ragged = scenes.primary_rays(s, 1000 + 37, 5)                         # not a multiple of the wave/block size
    o = orc.OracleScene(s)
    assert_hits_bitexact(sc.traverse(ragged), o.traverse(ragged), "ragged")
    sc.close()
    # empty scene and single-triangle scene (root is an always-passing leaf upstream, Geometry.h:401)
    e = scenes.cornell(64, 64)
    e.verts, e.tri_mat = e.verts[:0], e.tri_mat[:0]
    se = api.Scene(e)
    assert (se.traverse(ragged)["t"] > 3e38).all() and (se.visible(scenes.shadow_segments(s, 100)) == 1).all()
    se.close()
    for k in (1, 2, 3):
        one = scenes.cornell(64, 64)
        one.verts, one.tri_mat = one.verts[:k].copy(), one.tri_mat[:k].copy()
        so, oo = api.Scene(one), orc.OracleScene(one)
        rays = np.concatenate([scenes.primary_rays(s, 5000, 6), scenes.bounce_rays(s, 5000, 7)])
        assert_hits_bitexact(so.traverse(rays), oo.traverse(rays), f"{k}-triangle scene")
        segs = scenes.shadow_segments(s, 5000, 8)
        assert (so.visible(segs) == oo.visible(segs)).all()
        so.close()


def test_large_host_batches_are_pipelined_and_identical():
    """Host-buffer batches above 2 Mi items are cut into chunks that overlap copy-in, trace and copy-out on three streams: the
    results must be those of the same rays passed in small calls (ragged last chunk included)."""
    s = scenes.interior("bathroom", 0.3, 64, 36)
    sc = api.Scene(s)
    n = (5 << 20) + 12345
    rays = scenes.bounce_rays(s, n, 21)
    segs = scenes.shadow_segments(s, n, 22)
    big_h, big_v = sc.traverse(rays), sc.visible(segs)
    step = 1 << 20
    for a in range(0, n, step):
        assert np.array_equal(big_h[a:a + step].view(np.uint8), sc.traverse(rays[a:a + step]).view(np.uint8)), a
        assert np.array_equal(big_v[a:a + step], sc.visible(segs[a:a + step])), a
    sc.close()


def test_hostile_rays_match_the_compiled_reference():
    """Rays no renderer should produce but a caller can: NaN / inf components, exact zeros, denormals, huge magnitudes, segments
    of zero and sub-epsilon length.  Whatever upstream's arithmetic makes of them (comparisons with NaN reject nothing, Min/Max
    are ternaries) is the answer; the GPU must give it bit for bit -- against the compiled reference when its build travelled
    with the tree, else against the oracle."""
    s = scenes.interior("bathroom", 0.15, 64, 36)
    sc = api.Scene(s)
    ref = orc.RefScene(s) if orc.have_ref() else orc.OracleScene(s)
    rng = np.random.default_rng(12)
    base = scenes.bounce_rays(s, 20000, 4).copy()
    specials = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-45, 1e-38, 3e38, -3e38, 1.0, -1.0], np.float32)
    rays = base.copy()
    for k in range(len(rays)):                      # one to three components replaced by a special value
        for c in rng.choice(6, size=rng.integers(1, 4), replace=False):
            rays[k, c] = specials[rng.integers(len(specials))]
    axis = base[:3000].copy()                       # axis-parallel directions with exact zeros, origins on vertex coordinates
    axis[:, 3:] = 0.0
    axis[np.arange(3000), 3 + rng.integers(0, 3, 3000)] = rng.choice(np.array([1.0, -1.0, 1e-30, 2.5], np.float32), 3000)
    v = s.verts[:, :, :3].reshape(-1, 3)
    axis[:, :3] = v[rng.integers(0, len(v), 3000)]
    rays = np.concatenate([rays, axis]).astype(np.float32)
    got, want = sc.traverse(rays), ref.traverse(rays)
    assert_hits_bitexact(got, want, "hostile rays", nan_payload_free=True)
    segs = scenes.shadow_segments(s, 20000, 9).copy()
    segs[:2000, 3:] = segs[:2000, :3]                                          # zero length: dir = 0/0
    segs[2000:4000, 3:] = segs[2000:4000, :3] + np.float32(1e-4) * rng.normal(size=(2000, 3)).astype(np.float32)   # shorter than 2 eps
    for k in range(4000, 12000):
        for c in rng.choice(6, size=rng.integers(1, 3), replace=False):
            segs[k, c] = specials[rng.integers(len(specials))]
    assert np.array_equal(sc.visible(segs), ref.visible(segs))
    sc.close()


def test_hostile_scene_matches_the_compiled_reference():
    """Geometry no exporter should write but the ABI accepts: zero-area triangles (two equal vertices, collinear vertices),
    exact duplicates (equal centroids: upstream's split fails and leaves a many-triangle leaf), coordinates of 1e15 next to unit
    ones, triangles thinner than a float ulp of their position."""
    base = scenes.random_soup(6000, seed=5)
    v = base.verts.copy()
    rng = np.random.default_rng(3)
    idx = rng.permutation(len(v))
    v[idx[:400], 1, :3] = v[idx[:400], 0, :3]                                         # two equal vertices
    v[idx[400:800], 2, :3] = (v[idx[400:800], 0, :3] + v[idx[400:800], 1, :3]) * np.float32(0.5)   # collinear
    v[idx[800:1400]] = v[idx[1400:1460]].repeat(10, axis=0)                             # ten copies each of 60 triangles
    v[idx[1460:1500], :, :3] *= np.float32(1e15)                                       # far outliers
    v[idx[1500:1600], 1, :3] = v[idx[1500:1600], 0, :3] + np.float32(1e-9)             # below the ulp of the position
    s = scenes.SceneData("hostile", v, base.tri_mat.copy(), base.materials, base.textures)
    scenes.set_camera(s, (0, 0, 12), (0, 0, 0), (0, 1, 0), 45.0, 64, 64)
    ref = orc.RefScene(s) if orc.have_ref() else orc.OracleScene(s)
    for db in (False, True):
        sc = api.Scene(s, device_build=db)
        if not db:
            host_order = sc.leaf_order()
        else:
            assert np.array_equal(sc.leaf_order(), host_order)
        ro = ref.leaf_order()
        assert np.array_equal(sc.leaf_order(), ro[0] if isinstance(ro, tuple) else ro)
        rays = np.concatenate([scenes.primary_rays(s, 20000, 1), scenes.bounce_rays(base, 40000, 2)])
        assert_hits_bitexact(sc.traverse(rays), ref.traverse(rays), f"hostile scene (device_build={db})", nan_payload_free=True)
        segs = scenes.shadow_segments(base, 40000, 3)
        assert np.array_equal(sc.visible(segs), ref.visible(segs))
        sc.close()


@pytest.mark.parametrize("make,label", [(scenes.bathroom, "bathroom"), (scenes.living_room, "living-room")])
def test_full_size_wide_walk_equals_reference_order_walk(make, label):
    """At the bench's full size (BASELINE configs 3-4) the oracle is too slow; the size-independent property is that the
    ordered, culled 4-wide walk returns exactly what the reference-order walk (pinned above) returns."""
    s = make(1.0, 1920, 1080)
    sc = api.Scene(s, device_build=(label == "living-room"))
    info = sc.info()
    assert info["n_tris"] > 500_000
    n = 1 << 20
    rays = np.concatenate([scenes.primary_rays(s, n, 1), scenes.bounce_rays(s, n, 2)])
    a, b = sc.traverse(rays, 0), sc.traverse(rays, 1)
    assert_hits_bitexact(a, b, label + " full size")
    segs = scenes.shadow_segments(s, n, 3)
    assert (sc.visible(segs, 0) == sc.visible(segs, 1)).all()
    # spot-check a slice against the CPU oracle as well
    o = orc.OracleScene(s)
    idx = np.random.default_rng(1).integers(0, len(rays), 20_000)
    assert_hits_bitexact(a[idx], o.traverse(rays[idx], threads=16), label + " oracle slice")
    sc.close()


def test_launches_on_two_streams_share_the_scene_scratch_safely():
    """rt_intersect_*_dev on two caller streams at once: the scene's work counter and stack-spill arrays are one
    set, so the library chains such launches on the device.  Both results must be the single-call results, bit for bit."""
    import torch
    s = scenes.interior("bathroom", 0.3, 64, 36)
    sc = api.Scene(s)
    n = 3_000_000                                       # > 2 Mi items, the size ADVICE names
    ra, rb = scenes.bounce_rays(s, n, 21), scenes.bounce_rays(s, n, 22)
    sg = scenes.shadow_segments(s, n, 23)
    want_a, want_b, want_v = sc.traverse(ra), sc.traverse(rb), sc.visible(sg)
    d_a, d_b, d_s = (torch.from_numpy(x).cuda() for x in (ra, rb, sg))
    o_a = torch.zeros(n * 5, dtype=torch.int32, device="cuda"); o_b = torch.zeros_like(o_a)
    o_v = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(2):
        sc.traverse_dev(d_a.data_ptr(), n, o_a.data_ptr(), 0, s1.cuda_stream)
        sc.traverse_dev(d_b.data_ptr(), n, o_b.data_ptr(), 0, s2.cuda_stream)
        sc.visible_dev(d_s.data_ptr(), n, o_v.data_ptr(), 0, s3.cuda_stream)
    torch.cuda.synchronize()
    got_a, got_b = o_a.cpu().numpy().view(HIT_DTYPE), o_b.cpu().numpy().view(HIT_DTYPE)
    assert_hits_bitexact(got_a, want_a, "stream 1")
    assert_hits_bitexact(got_b, want_b, "stream 2")
    assert np.array_equal(o_v.cpu().numpy(), want_v)
    # two host threads through the host-buffer entry points (each with its own chunk pipeline and run stream)
    import threading
    res = {}
    th = [threading.Thread(target=lambda k=k, r=r: res.__setitem__(k, sc.traverse(r))) for k, r in (("a", ra), ("b", rb))]
    for t in th: t.start()
    for t in th: t.join()
    assert_hits_bitexact(res["a"], want_a, "thread a")
    assert_hits_bitexact(res["b"], want_b, "thread b")
    sc.close()


def test_cull_slack_holds_for_far_origins_grazing_and_near_axis_rays():
    """The ordered walk may skip a box only through a slack of 2^-17 of the largest coordinate in play (rt_device.h,
    RT_PAD_SCALE).  Rays chosen to strain it: origins 10^2..10^5 scene sizes away aimed at surface points
    (the slack scales with |o|, the hit point's rounding too), directions within 10^-3..10^-7 of an axis (|1/dir| up to 10^7: the
    pad in t is at its largest), rays grazing the tessellated fixtures within 10^-4 rad of their tangent planes, and shadow
    segments of the same kinds.  The reference-order walk (no culling at all, upstream's own sequence) must agree bit for bit."""
    s = scenes.interior("bathroom", 0.5, 64, 36)
    sc = api.Scene(s)
    rng = np.random.default_rng(77)
    n = 400_000
    pos, nn = scenes.surface_points(s, 3 * n, rng)
    # far origins
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = 10.0 ** rng.uniform(2, 6, (n, 1))
    far = np.concatenate([pos[:n] - d * dist, d * 10.0 ** rng.uniform(-3, 3, (n, 1))], 1)
    # near-axis directions through surface points
    ax = rng.integers(0, 3, n)
    da = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-7, -3, (n, 1))
    da[np.arange(n), ax] = rng.choice([-1.0, 1.0], n)
    near_axis = np.concatenate([pos[n:2 * n] - da * rng.uniform(0.1, 5, (n, 1)), da], 1)
    # grazing: direction in the tangent plane, tilted by 1e-6..1e-4 towards the surface
    t = np.cross(nn[2 * n:], rng.normal(size=(n, 3))); t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
    g = t - nn[2 * n:] * 10.0 ** rng.uniform(-6, -4, (n, 1))
    graze = np.concatenate([pos[2 * n:] - g * rng.uniform(0.01, 3, (n, 1)), g], 1)
    rays = np.concatenate([far, near_axis, graze]).astype(np.float32)
    wide, ref = sc.traverse(rays, 0), sc.traverse(rays, 1)
    assert_hits_bitexact(wide, ref, "strained rays, wide vs reference order", nan_payload_free=True)
    assert 0.3 < (ref["t"] < 3e38).mean() < 1.0
    segs = np.concatenate([rays[:, :3], rays[:, :3] + rays[:, 3:] * rng.uniform(0.5, 2.0, (len(rays), 1)).astype(np.float32) *
                           np.where(ref["t"] < 3e38, ref["t"], 1.0)[:, None]], 1).astype(np.float32)   # ends around the first hit
    v0, v1 = sc.visible(segs, 0), sc.visible(segs, 1)
    assert np.array_equal(v0, v1) and 0.05 < v0.mean() < 0.95
    if orc.have_ref():      # and against the compiled reference itself on a sample
        r = orc.RefScene(s)
        k = rng.choice(len(rays), 60_000, replace=False)
        assert_hits_bitexact(wide[k], r.traverse(rays[k]), "strained rays vs compiled reference", nan_payload_free=True)
        assert np.array_equal(v0[k], r.visible(segs[k]))
        r.close()
    sc.close()


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(256, 256),
    lambda: scenes.random_soup(3000, seed=5),
    lambda: scenes.axis_aligned_grid(6),
    lambda: scenes.interior("bathroom", 0.06, 320, 180),
], ids=["cornell", "soup", "lattice", "bathroom"])
def test_answers_refused_by_the_gate_box_are_recomputed_exactly(make):
    """The wide walk culls with quantised, padded boxes only; upstream's candidate set ("every ancestor box passes rayAABB",
    Geometry.h:441-490) is restored by testing the reported hit's gate box with upstream's expression and walking the ray again
    with the exact walk when that test fails.  Rays aimed at vertices and edges put hits on the surface of their leaf's box:
    the counters must show that refusals happen on these rays (the path under test runs), and every answer must equal the
    oracle's and the reference-order walk's bit for bit."""
    s = make()
    o, sc = orc.OracleScene(s), api.Scene(s)
    sc.counters_enable(True)
    n = 1 << 20
    rays, segs = aimed_rays(s, n)
    sc.counters()
    got = sc.traverse(rays, 0)
    refused = sc.counters()["gate_refused"]
    vis = sc.visible(segs, 0)
    any_refused = sc.counters()["any_gate_refused"]
    assert_hits_bitexact(got, sc.traverse(rays, 1), s.name + " wide vs reference-order, aimed rays")
    assert (vis == sc.visible(segs, 1)).all()
    k = 200_000
    assert_hits_bitexact(got[:k], o.traverse(rays[:k], threads=8), s.name + " wide vs oracle, aimed rays")
    assert (vis[:k] == o.visible(segs[:k], threads=8)).all()
    assert refused > 0, "no closest-hit answer was refused: the aimed rays no longer reach rt_settle's fallback"
    print(s.name, "refused", refused, "of", n, "closest;", any_refused, "any-hit")
    sc.close()


@pytest.mark.parametrize("scale,shift", [(1e-12, 0.0), (1e12, 0.0), (1.0, 3e5), (1e-3, 40.0), (1e15, 0.0)],
                         ids=["tiny", "huge", "far-from-origin", "small-and-offset", "beyond-the-scaled-step"])
def test_quantised_nodes_on_scaled_and_shifted_scenes(scale, shift):
    """The walk's nodes put child boxes on a 256-step grid whose step is a power of two taken from the node's extent: scenes a
    factor 1e12 smaller or larger than the fixtures, or far from the origin (where a coordinate's ulp approaches the size of a
    triangle), must give the oracle's answers bit for bit, and a flat scene (zero extent on an axis: step 0) too.  The node step
    multiplies 1/dir by the grid step * 2^24 (plane bytes read as f16 denormals): rays with a direction component of 1e-17 keep
    that product finite at every scale but 1e15, where |1/dir| * the largest coordinate passes 1e31; and beside components of the
    scene's size they make upstream's own arithmetic overflow on the large scenes (see below).  Either way they must take the
    exact walk (rt_ray_fits_scene) -- the counters say that they do."""
    base = scenes.random_soup(4000, seed=12)
    v = base.verts.copy()
    v[:, :, :3] = (v[:, :, :3] * np.float32(scale) + np.float32(shift)).astype(np.float32)
    v[:200, :, 2] = np.float32(shift)                      # a flat patch: leaves and nodes without extent along z
    s = scenes.SceneData("scaled", v, base.tri_mat.copy(), base.materials, base.textures)
    c = np.float32(shift)
    scenes.set_camera(s, (c, c, c + 12 * scale), (c, c, c), (0, 1, 0), 45.0, 64, 64)
    o, sc = orc.OracleScene(s), api.Scene(s)
    rays, segs = aimed_rays(s, 150_000, seed=3)
    steep = rays[:5000].copy()                             # one direction component tiny but not zero: |1/dir| = 1e17, a tame ray
    steep[np.arange(5000), 3 + np.arange(5000) % 3] = np.where(np.arange(5000) % 2 == 0, 1e-17, -1e-17).astype(np.float32)
    rays = np.concatenate([rays, scenes.primary_rays(s, 20_000, 1), axis_rays(s, 10_000), steep])
    sc.counters_enable(True); sc.counters()
    # nan_payload_free: at 1e12 upstream accepts eleven of the steep rays' plane hits at t ~ 1e28 with NaN barycentrics (their
    # points overflow; a comparison with NaN rejects nothing) -- the GPU must report the same hits (rt_ray_fits_scene sends such
    # rays to the exact walk), and 1 - (NaN + NaN) carries another payload on x86 than on the GPU
    assert_hits_bitexact(sc.traverse(rays, 0), o.traverse(rays, threads=8), f"scaled x{scale} +{shift}", nan_payload_free=True)
    exact = sc.counters()["exact_path_rays"]
    sc.counters_enable(False)
    print(f"scale {scale}: {exact} of {len(rays)} rays took an exact walk")
    if scale >= 1e12:
        assert exact >= 5000, "rays whose scaled grid step or whose hit points can overflow must take the exact walk"
    assert (sc.visible(segs, 0) == o.visible(segs, threads=8)).all()
    q, _, _ = sc.walk_nodes()
    assert np.isfinite(q["org"]).all() and (q["exp"] >= -126).all()
    sc.close()
