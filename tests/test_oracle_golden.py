"""CPU: the oracle (oracle/rt_oracle.c) against fixtures produced by the compiled reference
(tests/golden/make_golden.py) and against the reference's own unit-test vectors (RTTest/RTTest.cpp)."""
import os

import numpy as np
import pytest

import orc
from util import GOLDEN, GOLDEN_SCENES, assert_hits_bitexact, bits, load_golden


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_oracle_matches_reference_fixture(name):
    s, g, want = load_golden(name)
    o = orc.OracleScene(s)
    assert (o.leaf_order() == g["leaf_order"]).all()
    info = o.info()
    assert [info["depth"], info["nodes"], info["leaves"]] == g["stats"].tolist()
    assert_hits_bitexact(o.traverse(g["rays"]), want, name)
    assert (o.visible(g["segs"]) == g["visible"]).all()
    # UseBVH=false brute force (Scene.h:123) agrees with the tree on hit/miss wherever no box-rounding is in play:
    # it may only ever find *more* hits than the tree (the tree can drop a hit on a box edge), never fewer.
    b = o.traverse(g["rays"][:512], brute=True)
    assert ((b["t"] < 3e38) | ~(want["t"][:512] < 3e38)).all()


def test_rttest_vectors():
    g = np.load(f"{GOLDEN}/scalars.npz")
    L = orc.lib()
    tuv = np.zeros(3, np.float32)
    tri, ray = g["tri"], g["ray"]
    hit = L.orc_tri_intersect(tri.ctypes.data, ray.ctypes.data, tuv.ctypes.data)
    # RTTest.cpp:39 TestRayInterset, production intersector: hit, t=1, u=0.5 (weight of v0), v=0.25
    assert hit == 1 and tuv.tolist() == [1.0, 0.5, 0.25]
    assert g["tri0"].tolist() == [1.0, 1.0, 0.5, 0.25]
    # RTTest.cpp:28 InfinteArea
    assert np.isinf(L.orc_aabb_default_area()) and np.isinf(g["aabb_default_area"][0])
    # RTTest.cpp:15 TestRayIntersect (plane n=(0,1,1) d=0, ray from (0,10,0) down): t = 10
    assert g["plane"].tolist() == [1.0, 10.0]


def test_sampling_frame_matrix_tables():
    g = np.load(f"{GOLDEN}/scalars.npz")
    L = orc.lib()
    out = np.zeros(4, np.float32)
    for w in range(3):
        for i in range(256):
            L.orc_sampling(w, float(g["r12"][i, 0]), float(g["r12"][i, 1]), out.ctypes.data)
            assert (bits(out) == bits(g["sampling"][w, i])).all(), (w, i)
    fr = np.zeros(15, np.float32)
    for i in range(256):
        n, v = np.ascontiguousarray(g["frame_in"][i, 0]), np.ascontiguousarray(g["frame_in"][i, 1])
        L.orc_frame(n.ctypes.data, v.ctypes.data, fr.ctypes.data)
        assert (bits(fr) == bits(g["frame_out"][i])).all()
    m = np.zeros(16, np.float32)
    L.orc_matrix(0, g["persp_in"].ctypes.data, m.ctypes.data)
    assert (bits(m) == bits(g["persp"])).all()
    L.orc_matrix(1, g["lookat_in"].ctypes.data, m.ctypes.data)
    assert (bits(m) == bits(g["lookat"])).all()
    L.orc_matrix(2, np.ascontiguousarray(g["lookat"]).ctypes.data, m.ctypes.data)
    assert (bits(m) == bits(g["lookat_inv"])).all()
    L.orc_matrix(2, np.ascontiguousarray(g["persp"]).ctypes.data, m.ctypes.data)
    assert (bits(m) == bits(g["persp_inv"])).all()


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference_build():
    from raytracer_amd import scenes
    from util import axis_rays
    for s in (scenes.interior("living-room", 0.08, 160, 90), scenes.random_soup(2000, seed=9)):
        o, r = orc.OracleScene(s), orc.RefScene(s)
        assert (o.leaf_order() == r.leaf_order()[0]).all()
        rays = np.concatenate([scenes.primary_rays(s, 3000, 4), scenes.bounce_rays(s, 3000, 5), axis_rays(s, 500)])
        assert_hits_bitexact(o.traverse(rays), r.traverse(rays), s.name)
        segs = scenes.shadow_segments(s, 3000, 6)
        assert (o.visible(segs) == r.visible(segs)).all()
        for i in (0, s.ntris // 3, s.ntris - 1):
            assert (bits(o.tri_derived(i)) == bits(r.tri_derived(i))).all()
        r.close()


def pss_stream_positions(trace):
    """Where in PSS::random's stream (an MTRandom(1), Sampling.h:28, :47) each proposal's two nextPSSP() values come from, if they
    are fresh pushes (PSS::next, Sampling.h:60-67) and every stored coordinate costs two draws per timer tick it is behind
    (mutate, Sampling.h:70-82; smallStepPerturb, :85-94)."""
    pos, timer, timers, out = 0, 0, [], []
    for large in trace[:, 0] == 1:
        if large:
            timer += 1; timers = []                      # largeStep: currentTimer++, proposedState.reset() (Sampling.h:118-122)
        else:
            for k in range(len(timers)):                 # smallStep -> smallStepPerturb(currentTimer)
                pos += 2 * (timer - timers[k]); timers[k] = timer
        out.append(pos); pos += 2; timers += [0, 0]      # two fresh coordinates, pssTimer 0
    return np.array(out)


def test_pss_sampler_never_hands_out_a_mutated_coordinate():
    """What the PSSMLT design on both sides rests on (DESIGN section 4 "PSSMLT", oracle/rt_oracle.c orc_mlt_render): upstream's
    PSSSampler, driven by PSSMLTRender's own call sequence (Renderer.h:1013-1061), returns from nextPSSP() a value it has just
    pushed from its generator -- never one of the coordinates a small step mutated -- so every proposal's film position is a
    fresh uniform draw whatever the step kind.  Pinned by the compiled reference (Sampling.h is one of the three headers
    oracle/_ref builds): tests/golden/scalars_r3.npz, made by `make_golden.py r3` from oracle/ref_harness.cpp ref_pss_trace."""
    g = np.load(os.path.join(GOLDEN, "scalars_r3.npz"))
    tr, mt = g["pss_trace"], g["mt_seed1"]
    idx_before, size_before, size_after = tr[:, 3], tr[:, 4], tr[:, 5]
    assert (idx_before == size_before).all()             # currentIndex is past every stored coordinate: PSS::next pushes
    assert (size_after == size_before + 2).all()
    large = tr[:, 0] == 1
    assert (size_before[large] == 0).all() and (size_before[~large][1:] > 0).all() and 0.2 < large.mean() < 0.4
    assert size_before.max() >= 20                       # runs of small steps: the mutated coordinates pile up unread
    pos = pss_stream_positions(tr)
    # MTRandom(1) = std::mt19937(1) through uniform_real_distribution<float>(0, 1): one 32-bit output / 2^32 in float, kept below 1.
    # The fixture pins the first 2^15 draws from the compiled reference; numpy's MT19937 continues the same stream.
    x = np.random.RandomState(1).randint(0, 2 ** 32, size=int(pos[-1]) + 2, dtype=np.uint64).astype(np.float32) / np.float32(4294967296.0)
    x = np.where(x >= 1, np.nextafter(np.float32(1), np.float32(0)), x).astype(np.float32)
    assert np.array_equal(bits(x[:len(mt)]), bits(mt))
    mt = x
    assert np.array_equal(bits(tr[:, 1]), bits(mt[pos])) and np.array_equal(bits(tr[:, 2]), bits(mt[pos + 1]))   # raw generator output
    assert np.array_equal(bits(g["pss_init"]), bits(mt[:2]))                                                     # currentState likewise
    # the step decisions and accept draws are PSSSampler::random's own stream, interleaved with pathTrace's draws
    k = int(g["pss_path_draws"])
    dec = 2 * k + np.arange(len(tr)) * (k + 2)
    assert np.array_equal(large, mt[dec] < g["pss_large_prob"]) and np.array_equal(bits(tr[:, 7]), bits(mt[dec + k + 1]))
    # the restatement consumes primary-sample-space coordinates the same way: two fresh draws per proposal from a generator
    # that nothing else reads (stream 2, keyed by the proposal's index), decisions from another (stream 4)
    a = orc.mlt_trace(seed=5, sample=3, n=64)
    for i in range(64):
        assert np.array_equal(bits(a["xy"][i]), bits(orc.rng_floats(5, 3, 101 + i, 2, 2)))
    chain = orc.rng_floats(5, 3, 0, 4, 128)
    assert np.array_equal(a["large"], chain[0::2] < 0.3) and np.array_equal(bits(a["accept_draw"]), bits(chain[1::2]))
    if orc.have_ref() and hasattr(orc.ref(), "ref_pss_trace"):   # and live, when the reference is here
        import ctypes as C
        live = np.zeros_like(tr); init = np.zeros(2, np.float32)
        orc.ref().ref_pss_trace(len(tr), k, C.c_float(0.3), live.ctypes.data, init.ctypes.data)
        assert np.array_equal(bits(live), bits(tr))


def test_oracle_pssmlt_is_deterministic_and_covers_the_film():
    """PSSMLT restatement (Renderer.h:964-1063): parity unpinned (Renderer.h needs the Windows window header, so upstream's
    own PSSMLT cannot be built here); this pins the restatement's stated behaviour: same seed -> same film, disjoint
    sample indices -> different chains, proposals land uniformly over the film (every proposal's (x, y) is a fresh draw)."""
    from raytracer_amd import scenes
    o = orc.OracleScene(scenes.cornell(32, 32))
    a = o.mlt_render(1, seed=3)
    b = o.mlt_render(1, seed=3)
    c = o.mlt_render(1, seed=3, first_sample=1)
    assert np.isfinite(a).all() and np.array_equal(a, b) and not np.array_equal(a, c)
    assert (a.sum(-1) > 0).mean() > 0.6          # 100 000 proposals over 1024 pixels: only black pixels stay empty
    both = o.mlt_render(2, seed=3)
    assert np.allclose(both, a + c, rtol=1e-3, atol=1e-2)   # same splats, different fp32 summation order


def test_oracle_film_filters():
    """ImageFilter / Film::splat restatement (Imaging.h:138-247).  Parity unpinned: upstream only ever constructs BoxFilter
    (Renderer.h:61) and holds no fixture for the other two, so this checks the restated formulas' own known values and
    the invariants of Film::splat (weights normalised per sample: the film's sum does not depend on the filter)."""
    import math
    from raytracer_amd import scenes
    L = orc.lib()
    f32 = np.float32
    g0 = f32(math.exp(-0.0) - float(f32(math.exp(-1.0))))           # gauss(0) = exp(Nalpha * 0) - secondPart
    g1 = f32(math.exp(-1.0) - float(f32(math.exp(-1.0))))           # gauss(+-1): what float rounding of secondPart leaves
    assert L.orc_filter_eval(1, 0.5, -0.5) == f32(g0 * g0)          # offsets are truncated to int: (0, 0)
    assert L.orc_filter_eval(1, -1.5, 0.5) == f32(g1 * g0) and abs(g1) < 1e-7
    assert L.orc_filter_eval(0, 0.3, -0.9) == 1.0
    B = C = f32(1) / f32(3)
    a2, b2, c2, d2 = (-B) - (6 * C), (6 * B) + (30 * C), -(12 * B) - (48 * C), (8 * B) - (24 * C)
    mn_half = f32((float(a2) * 0.125 + float(b2) * 0.25 + float(f32(c2 * f32(0.5))) + float(d2)) / 6)
    assert L.orc_filter_eval(2, 0.5, 0.5) == f32(mn_half * mn_half)  # 0 < |x| <= 1 takes the *second* polynomial upstream
    c1 = 6 - (2 * B)
    assert L.orc_filter_eval(2, 0.0, 0.0) == f32(f32(float(c1) / 6) * f32(float(c1) / 6))
    assert L.orc_filter_eval(2, 1.5, 0.5) == 0.0 and L.orc_filter_eval(2, 2.5, 0.5) == 0.0
    s = scenes.cornell(48, 48)
    o = orc.OracleScene(s)
    box, _ = o.render(0, 2, seed=4, threads=4)
    for kind in (1, 2):
        o.set_filter(kind)
        f, _ = o.render(0, 2, seed=4, threads=4)
        assert abs(f.sum(dtype=np.float64) / box.sum(dtype=np.float64) - 1) < 1e-4
        # pixel-centre splats: both filters put (to 1e-7) a quarter of each sample on the 2x2 block at and after the pixel
        want = np.zeros_like(box, dtype=np.float64)
        H, W = box.shape[:2]
        for dy in (0, 1):
            for dx in (0, 1):
                ys, xs = np.arange(H) + dy, np.arange(W) + dx
                ny = np.minimum(2, H - np.arange(H))[:, None]; nx = np.minimum(2, W - np.arange(W))[None, :]
                share = box / (ny * nx)[..., None]
                want[dy:, dx:] += share[:H - dy, :W - dx]
        assert np.allclose(f, want, rtol=2e-4, atol=1e-5), kind
    o.set_filter(0)
    again, _ = o.render(0, 2, seed=4, threads=4)
    assert np.array_equal(again, box)


# ------------------------------------------------------------------------------------------------ round-2 pins
def test_vec2_index_truncates_like_the_reference():
    """Vec2::operator[] returns int (Sampling.h:204): the compiled reference turns every coordinate in [0, 1) into 0.  This is
    the reference-side evidence under the EnvironmentMap restatement (tabulated draw -> theta = 0 -> pdf = +inf)."""
    g = np.load(f"{GOLDEN}/scalars_r2.npz")
    xy, want = g["vec2_xy"], g["vec2_index"]
    inside = ((xy >= 0) & (xy < 1)).all(axis=1)
    assert inside.sum() >= 256 and (want[inside] == 0).all()
    assert want[-3].tolist() == [1, 1] and want[-2].tolist() == [2, 0] and want[-1].tolist() == [-1, 3]   # C truncation toward zero
    assert (want == np.trunc(xy).astype(np.int32)).all()
    if orc.have_ref() and hasattr(orc.ref(), "ref_vec2_index"):
        import ctypes as C
        L = orc.ref()
        live = np.array([[L.ref_vec2_index(C.c_float(x), C.c_float(y), i) for i in (0, 1)] for x, y in xy], np.int32)
        assert (live == want).all()


def test_triangle_centre_and_area_match_the_reference():
    g = np.load(f"{GOLDEN}/scalars_r2.npz")
    L = orc.lib()
    import ctypes as C
    L.orc_tri_centre.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros(5, np.float32)
    for tri, want in zip(g["tris"], g["centre_area"]):
        L.orc_tri_centre(np.ascontiguousarray(tri).ctypes.data, out.ctypes.data)
        assert (bits(out[:3]) == bits(want[:3])).all() and bits(out[4:5]) == bits(want[4:5])
    assert g["centre_area"][5, 4] == 0.0 and g["centre_area"][6, 4] == 0.0     # the two degenerate ones: dropped by loadInstance (SceneLoader.h:229)


def test_environment_map_tables_and_their_collapse():
    """EnvironmentMap::build / Sample1D / SampleContinuous / sample / PDF (Lights.h:207-424) restated in the oracle: the tables
    are per-row pdfs (rows sum to 1, the row pdfs sum to 1), SampleContinuous lands in the texel its running sums select,
    and after Vec2::operator[]'s int every draw ends at wi = (0, 1, 0) with pdf = +inf, PDF(wi) = +inf for every direction."""
    from raytracer_amd import scenes
    s = scenes.open_box(32, 32, "env")
    o = orc.OracleScene(s)
    nv, nu = s.textures[s.bg_env_tex].shape[:2]
    cond, marg, func_int = o.env_tables(nu, nv)
    assert func_int > 0 and np.allclose(cond.sum(axis=1), 1.0, atol=1e-5) and abs(marg.sum() - 1.0) < 1e-5
    lum = (s.textures[s.bg_env_tex].astype(np.float64) * [0.2126, 0.7152, 0.0722]).sum(-1)
    func = lum * np.sin((np.arange(nv) + 0.5) / nv * np.pi)[:, None]
    assert np.allclose(marg, func.sum(1) / func.sum(), rtol=1e-4) and np.allclose(cond, func / func.sum(1, keepdims=True), rtol=1e-4)
    rng = np.random.default_rng(4)
    u = np.concatenate([rng.uniform(0, 1, (4000, 3)), [[0.0, 0.0, 0], [0.99999994, 0.99999994, 0], [0.5, 0.99999994, 0]]]).astype(np.float32)
    sc = o.env_probe(1, u)
    d0, d1, pdf = sc[:, 0], sc[:, 1], sc[:, 2]
    # the continuous coordinates invert the running sums: the row d1 falls in has cumulative pdf bracketing u1
    cm = np.concatenate([[0], np.cumsum(marg.astype(np.float64))])
    row = np.minimum((d1[:4000] * nv).astype(int), nv - 1)
    assert ((cm[row] <= u[:4000, 1] + 1e-5) & (u[:4000, 1] <= cm[row + 1] + 1e-5)).mean() > 0.999
    assert (pdf[:4000] > 0).all()
    smp = o.env_probe(0, u)
    assert (smp[:4000, :3] == [0.0, 1.0, 0.0]).all() and np.isposinf(smp[:4000, 3]).all()
    w = rng.normal(size=(500, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True)
    assert np.isposinf(o.env_probe(2, w)[:, 0]).all()
    o.close()
