"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libref_geom.so, i.e. the
reference's own Geometry.h/Core.h/Sampling.h built by oracle/Makefile).  Run in a container that holds
/root/reference:   python tests/golden/make_golden.py
The fixtures are data only: seeded inputs (scene triangles, rays, segments) and the reference's outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def axis_rays(s, n, rng):
    """Rays with exact-zero direction components and origins on vertex coordinates (NaN/inf slab cases)."""
    v = s.verts[:, :, :3].reshape(-1, 3)
    o = v[rng.integers(0, len(v), n)].astype(np.float32) + np.float32(0.0)
    o[:, 1] += rng.choice([0.0, 0.5, 2.0], n).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    ax = rng.integers(0, 3, n)
    d[np.arange(n), ax] = rng.choice([-1.0, 1.0], n)
    two = rng.uniform(0, 1, n) < 0.4
    d[two, (ax[two] + 1) % 3] = rng.uniform(-1, 1, two.sum()).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)


def main():
    cases = {
        "cornell": scenes.cornell(256, 256),
        "cornell_spec": scenes.cornell(256, 256, with_specular=True),
        "soup": scenes.random_soup(3000, seed=5),
        "lattice": scenes.axis_aligned_grid(6),
        "bathroom_small": scenes.interior("bathroom", 0.06, 320, 180),
    }
    for name, s in cases.items():
        r = orc.RefScene(s)
        rng = np.random.default_rng(11)
        n = 4096
        rays = np.concatenate([scenes.primary_rays(s, n, 1), scenes.bounce_rays(s, n, 2), axis_rays(s, 1024, rng)])
        segs = scenes.shadow_segments(s, n, 3)
        hits = r.traverse(rays)
        vis = r.visible(segs)
        order, st = r.leaf_order()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), verts=s.verts, tri_mat=s.tri_mat, rays=rays, segs=segs,
                            hit_id=hits["ID"], hit_t=hits["t"], hit_alpha=hits["alpha"], hit_beta=hits["beta"], hit_gamma=hits["gamma"],
                            visible=vis, leaf_order=order, stats=np.array([st["depth"], st["nodes"], st["leaves"]], np.int32))
        print(name, s.ntris, "hits", int((hits["t"] < 3e38).sum()), "/", len(rays), "visible", float(vis.mean()), st)
        r.close()
    # scalar known-answer vectors of the reference's own unit tests (RTTest/RTTest.cpp:15-72), evaluated by the reference
    L = orc.ref()
    tri = np.array([0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0], np.float32)
    ray = np.array([0.25, 0.25, 1.0, 0, 0, -1], np.float32)
    out = {}
    for which in (0, 2, 3):
        tuv = np.zeros(3, np.float32)
        hit = L.ref_tri_intersect(tri.ctypes.data, ray.ctypes.data, which, tuv.ctypes.data)
        out[f"tri{which}"] = np.concatenate([[hit], tuv]).astype(np.float32)
    t = np.zeros(1, np.float32)
    pn = np.array([0, 1, 1], np.float32)
    pr = np.array([0, 10, 0, 0, -1, 0], np.float32)
    out["plane"] = np.array([L.ref_plane_intersect(pn.ctypes.data, 0.0, pr.ctypes.data, t.ctypes.data), t[0]], np.float32)
    out["aabb_default_area"] = np.array([L.ref_aabb_default_area()], np.float32)
    # sampling / frame / matrix tables
    rng = np.random.default_rng(3)
    r12 = rng.uniform(0, 1, (256, 2)).astype(np.float32)
    samp = np.zeros((3, 256, 4), np.float32)
    for w in range(3):
        for i in range(256):
            L.ref_sampling(w, float(r12[i, 0]), float(r12[i, 1]), samp[w, i].ctypes.data)
    nv = rng.normal(size=(256, 2, 3)).astype(np.float32)
    fr = np.zeros((256, 15), np.float32)
    for i in range(256):
        L.ref_frame(nv[i, 0].ctypes.data, nv[i, 1].ctypes.data, fr[i].ctypes.data)
    mats = {}
    pin = np.array([0.001, 10000.0, 16.0 / 9.0, 45.0], np.float32)
    m = np.zeros(16, np.float32); L.ref_matrix(0, pin.ctypes.data, m.ctypes.data); mats["persp"] = m.copy()
    lin = np.array([3.3, 1.65, 2.5, -1.2, 0.9, -0.9, 0, 1, 0], np.float32)
    L.ref_matrix(1, lin.ctypes.data, m.ctypes.data); mats["lookat"] = m.copy()
    inv = np.zeros(16, np.float32); L.ref_matrix(2, mats["lookat"].ctypes.data, inv.ctypes.data); mats["lookat_inv"] = inv.copy()
    L.ref_matrix(2, mats["persp"].ctypes.data, inv.ctypes.data); mats["persp_inv"] = inv.copy()
    mt = np.zeros(16, np.float32); L.ref_mtrandom(1, 16, mt.ctypes.data)
    np.savez_compressed(os.path.join(HERE, "scalars.npz"), tri=tri, ray=ray, r12=r12, sampling=samp, frame_in=nv, frame_out=fr,
                        persp_in=pin, lookat_in=lin, mtrandom_seed1=mt, **out, **mats)
    print({k: v.tolist() for k, v in out.items()})


def pins_r2():
    """Round-2 pins from the compiled reference (Sampling.h Vec2, Core.h Matrix helpers as loadInstance uses them,
    Triangle::centre): tests/golden/scalars_r2.npz.   python tests/golden/make_golden.py r2"""
    import ctypes as C
    L = orc.ref()
    rng = np.random.default_rng(21)
    xy = rng.uniform(0, 1, (256, 2)).astype(np.float32)
    xy = np.concatenate([xy, np.array([[0.0, 0.99999994], [1.0, 1.5], [2.75, -0.5], [-1.25, 3.0]], np.float32)])
    vec2 = np.array([[L.ref_vec2_index(C.c_float(x), C.c_float(y), i) for i in (0, 1)] for x, y in xy], np.int32)
    ang = 0.7
    worlds = np.stack([
        np.array([[2 * np.cos(ang), 0, 2 * np.sin(ang), 0.5], [0, 1.5, 0, -1.0], [-np.sin(ang), 0, np.cos(ang), 3.0], [0, 0, 0, 1]]),
        np.array([[0.3, -1.2, 0.4, 7.0], [0.9, 0.1, -0.6, -2.5], [0.2, 0.8, 1.7, 0.25], [0, 0, 0, 1]]),
        np.eye(4)]).astype(np.float32)
    pos = rng.uniform(-2, 2, (len(worlds), 48, 3)).astype(np.float32)
    nrm = rng.normal(size=(len(worlds), 48, 3)).astype(np.float32)
    out_p, out_n, out_v = np.zeros_like(pos), np.zeros_like(nrm), np.zeros_like(nrm)
    out_t, out_m = np.zeros((len(worlds), 16), np.float32), np.zeros((len(worlds), 16), np.float32)
    for w, W in enumerate(worlds):
        Wf = np.ascontiguousarray(W.reshape(-1))
        for i in range(pos.shape[1]):
            L.ref_matrix_apply(0, Wf.ctypes.data, pos[w, i].ctypes.data, out_p[w, i].ctypes.data)
            L.ref_matrix_apply(3, Wf.ctypes.data, nrm[w, i].ctypes.data, out_n[w, i].ctypes.data)
            L.ref_matrix_apply(1, Wf.ctypes.data, nrm[w, i].ctypes.data, out_v[w, i].ctypes.data)
        L.ref_matrix_apply(2, Wf.ctypes.data, None, out_t[w].ctypes.data)
        other = np.ascontiguousarray(worlds[(w + 1) % len(worlds)].reshape(-1))
        L.ref_matrix_apply(4, Wf.ctypes.data, other.ctypes.data, out_m[w].ctypes.data)
    tris = rng.uniform(-3, 3, (32, 3, 8)).astype(np.float32)
    tris[5, 1, :3] = tris[5, 0, :3]      # zero area
    tris[6, 2, :3] = tris[6, 1, :3]
    centre = np.zeros((32, 5), np.float32)
    for i in range(32):
        L.ref_tri_centre(tris[i].ctypes.data, centre[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "scalars_r2.npz"), vec2_xy=xy, vec2_index=vec2, worlds=worlds, pos=pos, nrm=nrm,
                        mul_point=out_p, normal_xf=out_n, mul_vec=out_v, transpose=out_t, mul=out_m, tris=tris, centre_area=centre)
    print("vec2 index values over [0,1)^2:", np.unique(vec2[:256]), "edge rows:", vec2[256:].tolist())


def pins_r3():
    """Round-3 pin from the compiled reference: PSS / PSSSampler (Sampling.h:43-139) driven by PSSMLTRender's call sequence
    (oracle/ref_harness.cpp ref_pss_trace) -> tests/golden/scalars_r3.npz.   python tests/golden/make_golden.py r3"""
    import ctypes as C
    L = orc.ref()
    n, path_draws = 600, 7
    trace = np.zeros((n, 8), np.float32); init = np.zeros(2, np.float32)
    L.ref_pss_trace(n, path_draws, C.c_float(0.3), trace.ctypes.data, init.ctypes.data)
    mt = np.zeros(1 << 15, np.float32)      # MTRandom(1): the stream both PSS::random and PSSSampler::random start from (Sampling.h:28)
    L.ref_mtrandom(1, len(mt), mt.ctypes.data)
    np.savez_compressed(os.path.join(HERE, "scalars_r3.npz"), pss_trace=trace, pss_init=init, pss_path_draws=np.int32(path_draws),
                        pss_large_prob=np.float32(0.3), mt_seed1=mt)
    print("proposals", n, "large steps", int(trace[:, 0].sum()), "largest pssCoords", int(trace[:, 5].max()),
          "fresh every time:", bool((trace[:, 3] == trace[:, 4]).all()))


def survey_digest():
    """SURVEY 8(d)'s fixed Cornell ray set (3 x 2^20, seed 1) through the compiled reference: a digest of its answers, so the
    GPU test's primary pin is a committed fixture and not the untracked _ref binary.   python tests/golden/make_golden.py digest"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import hits_digest
    s = scenes.cornell(1024, 1024)
    r = orc.RefScene(s)
    n = 1 << 20
    rays = np.concatenate([scenes.primary_rays(s, n, 1), scenes.bounce_rays(s, n, 1)])
    segs = scenes.shadow_segments(s, n, 1)
    hits, vis = r.traverse(rays), r.visible(segs)
    out = {"scene": "scenes.cornell(1024, 1024)", "rays": "primary_rays(s, 2^20, 1) ++ bounce_rays(s, 2^20, 1)", "segments": "shadow_segments(s, 2^20, 1)",
           "sha256": hits_digest(hits, vis), "hits": int((hits["t"] < 3e38).sum()), "visible": int(vis.sum()),
           "source": "oracle/_ref/libref_geom.so (the reference's Geometry.h compiled by oracle/Makefile)"}
    json.dump(out, open(os.path.join(HERE, "cornell_survey_digest.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "r2":
        pins_r2()
    elif len(sys.argv) > 1 and sys.argv[1] == "r3":
        pins_r3()
    elif len(sys.argv) > 1 and sys.argv[1] == "digest":
        survey_digest()
    else:
        main()
