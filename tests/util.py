"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

from raytracer_amd import scenes
from raytracer_amd.abi import HIT_DTYPE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_SCENES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("scalars"))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    s = scenes.SceneData(name, g["verts"], g["tri_mat"], [scenes.Material()], [scenes.solid((0.7, 0.7, 0.7))])
    s.tri_mat = np.zeros_like(g["tri_mat"])  # geometry-only fixtures: one material
    hits = np.zeros(len(g["hit_t"]), HIT_DTYPE)
    for k in ("ID", "t", "alpha", "beta", "gamma"):
        hits[k] = g["hit_" + k.lower() if k != "ID" else "hit_id"]
    return s, g, hits


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_hits_bitexact(got, want, what="", nan_payload_free=False):
    """hit/miss, triangle ID, t, alpha, beta, gamma identical bit for bit (no tolerance, no exclusions).
    nan_payload_free: a NaN matches a NaN whatever its sign / payload bits (x86 makes -qNaN from inf - inf, the GPU +qNaN)."""
    def same(a, b):
        eq = bits(a) == bits(b)
        return eq | (np.isnan(a) & np.isnan(b)) if nan_payload_free else eq
    hg, hw = got["t"] < 3e38, want["t"] < 3e38
    assert (hg == hw).all(), f"{what}: hit/miss differs on {int((hg != hw).sum())} rays"
    assert same(got["t"], want["t"]).all(), f"{what}: t bits differ on {int((~same(got['t'], want['t'])).sum())} rays"
    m = hw
    assert (got["ID"][m] == want["ID"][m]).all(), f"{what}: triangle ID differs on {int((got['ID'][m] != want['ID'][m]).sum())} rays"
    for k in ("alpha", "beta", "gamma"):
        assert same(got[k][m], want[k][m]).all(), f"{what}: {k} bits differ on {int((~same(got[k][m], want[k][m])).sum())} rays"


def rel_mse(img, ref, eps=1e-2):
    """relative MSE as used for the north_star image gate: mean((a-b)^2 / (b^2 + eps))."""
    img, ref = np.asarray(img, np.float64), np.asarray(ref, np.float64)
    return float(np.mean((img - ref) ** 2 / (ref ** 2 + eps)))


def axis_rays(s, n, seed=11):
    rng = np.random.default_rng(seed)
    v = s.verts[:, :, :3].reshape(-1, 3)
    o = v[rng.integers(0, len(v), n)].astype(np.float32)
    o[:, 1] += rng.choice([0.0, 0.5, 2.0], n).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    ax = rng.integers(0, 3, n)
    d[np.arange(n), ax] = rng.choice([-1.0, 1.0], n)
    two = rng.uniform(0, 1, n) < 0.4
    d[two, (ax[two] + 1) % 3] = rng.uniform(-1, 1, two.sum()).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)


def hits_digest(hits, vis=None):
    """sha256 over hit/miss, triangle ID, t, alpha, beta, gamma (hits only; a miss has no defined ID) and a visibility array."""
    import hashlib
    h = hashlib.sha256()
    m = hits["t"] < 3e38
    h.update(np.packbits(m).tobytes())
    h.update(np.ascontiguousarray(hits["ID"][m]).tobytes())
    for k in ("t", "alpha", "beta", "gamma"):
        h.update(np.ascontiguousarray(hits[k][m]).tobytes())
    if vis is not None:
        h.update(np.packbits(np.asarray(vis, np.uint8)).tobytes())
    return h.hexdigest()


def aimed_rays(s, n, seed=7):
    """Rays aimed at triangle vertices, edge points and interior points from origins around the scene, and the segments that
    run through those points to twice the distance: hits that sit on the surface of their leaf's box, where the conservative
    wide walk's answer is most often refused by the gate test (rt_settle) and recomputed by the exact walk."""
    rng = np.random.default_rng(seed)
    v = s.verts[:, :, :3].astype(np.float32)
    tri = rng.integers(0, len(v), n)
    w = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    kind = rng.integers(0, 3, n)
    w[kind == 0] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, int((kind == 0).sum()))]
    e = np.nonzero(kind == 1)[0]
    w[e, rng.integers(0, 3, len(e))] = 0
    w[e] /= w[e].sum(1, keepdims=True)
    target = (v[tri] * w[:, :, None]).sum(1).astype(np.float32)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 3)).astype(np.float32)
    d = (target - o).astype(np.float32)
    return np.concatenate([o, d], 1), np.concatenate([o, o + 2 * d], 1).astype(np.float32)
