"""Procedural scenes for the RTBase hot path.

The reference's scene assets (cornell-box, bathroom, living-room, ...) are
git-ignored upstream and absent (/root/reference/.gitignore:6-15), so every
workload here is generated from a fixed seed.  A scene is the
flat data `loadScene`/`loadInstance` (SceneLoader.h:104-289) would have handed to
`Scene::init`: world-space triangles (3 x {p, normal, uv}), a material index per
triangle, a material table, textures, a background and a camera.

Everything is float32 and deterministic: the same arrays are fed to the HIP
library, to the CPU oracle and (geometry only) to the compiled reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# material type tags shared with include/rtamd.h (RT_MAT_*)
DIFFUSE, MIRROR, CONDUCTOR, GLASS, DIELECTRIC, ORENNAYAR, PLASTIC = range(7)
FLAG_LAYERED = 1


@dataclass
class Material:
    type: int = DIFFUSE
    albedo_tex: int = 0
    flags: int = 0
    emission: tuple = (0.0, 0.0, 0.0)
    int_ior: float = 1.33   # defaults of SceneLoader.h:126-127
    ext_ior: float = 1.0
    roughness: float = 1.0
    eta: tuple = (0.0, 0.0, 0.0)
    k: tuple = (0.0, 0.0, 0.0)
    sigma: float = 1.0


@dataclass
class SceneData:
    name: str
    verts: np.ndarray            # float32 [n, 3, 8]  (p.xyz, normal.xyz, u, v)
    tri_mat: np.ndarray          # uint32 [n]
    materials: list
    textures: list               # float32 [h, w, 3] each
    bg_kind: int = 0             # 0 BackgroundColour, 1 EnvironmentMap
    bg_colour: tuple = (0.0, 0.0, 0.0)
    bg_env_tex: int = -1
    cam_P: np.ndarray = field(default=None)   # float32 [16] row-major
    cam_V: np.ndarray = field(default=None)   # camera-to-world
    width: int = 0
    height: int = 0
    cam_params: dict = field(default=None)    # from / to / up / fov / flip_x as given to set_camera (scene.json properties)

    @property
    def ntris(self) -> int:
        return int(self.verts.shape[0])


# ----------------------------------------------------------------------------- camera (SceneLoader.h:241-259)
def perspective(n, f, aspect, fov):
    """Matrix::perspective (Core.h:460), float32 arithmetic."""
    f32 = np.float32
    t = f32(1.0) / f32(math.tan(float(f32(fov) * f32(0.5) * f32(3.141592654) / f32(180.0))))
    m = np.zeros(16, dtype=np.float32)
    m[0] = t / f32(aspect)
    m[5] = t
    m[10] = -f32(f) / (f32(f) - f32(n))
    m[11] = -(f32(f) * f32(n)) / (f32(f) - f32(n))
    m[14] = f32(-1.0)
    return m


def look_at(frm, to, up):
    """Matrix::lookAt (Core.h:439)."""
    frm, to, up = (np.asarray(v, dtype=np.float32) for v in (frm, to, up))

    def nrm(v):
        return (v / np.sqrt(np.dot(v, v))).astype(np.float32)

    d = nrm(frm - to)
    left = nrm(np.cross(up, d).astype(np.float32))
    new_up = np.cross(d, left).astype(np.float32)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = left, new_up, d
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(frm, left), -np.dot(frm, new_up), -np.dot(frm, d)
    return m.reshape(16)


def set_camera(scene: SceneData, frm, to, up, fov, width, height, flip_x=False):
    P = perspective(0.001, 10000.0, float(width) / float(height), fov)
    if flip_x:
        P[0] = -P[0]
    V = look_at(frm, to, up).reshape(4, 4).astype(np.float64)
    V = np.linalg.inv(V).astype(np.float32).reshape(16)  # V = lookAt().invert() (SceneLoader.h:251-252)
    scene.cam_P, scene.cam_V, scene.width, scene.height = P, V, int(width), int(height)
    scene.cam_params = dict(frm=tuple(map(float, frm)), to=tuple(map(float, to)), up=tuple(map(float, up)), fov=float(fov), flip_x=bool(flip_x))
    return scene


# ----------------------------------------------------------------------------- mesh helpers
class _Builder:
    def __init__(self):
        self.chunks, self.mats = [], []

    def add(self, pos, nrm, uv, mat):
        """pos,nrm: [n,3,3]; uv: [n,3,2]"""
        pos = np.asarray(pos, dtype=np.float32)
        nrm = np.asarray(nrm, dtype=np.float32)
        uv = np.asarray(uv, dtype=np.float32)
        v = np.concatenate([pos, nrm, uv], axis=2).astype(np.float32)
        # loadInstance drops zero-area triangles (SceneLoader.h:229)
        e1 = pos[:, 2] - pos[:, 1]
        e2 = pos[:, 0] - pos[:, 2]
        area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
        keep = area > 1e-12
        v = v[keep]
        self.chunks.append(v)
        self.mats.append(np.full(v.shape[0], mat, dtype=np.uint32))

    def finish(self):
        if not self.chunks:
            return np.zeros((0, 3, 8), np.float32), np.zeros((0,), np.uint32)
        return np.ascontiguousarray(np.concatenate(self.chunks, 0)), np.ascontiguousarray(np.concatenate(self.mats, 0))


def _grid_tris(P, N, UV):
    """P,N: [r,c,3], UV: [r,c,2] vertex grids -> two triangles per cell."""
    a, b, c, d = (slice(None, -1), slice(None, -1)), (slice(None, -1), slice(1, None)), \
                 (slice(1, None), slice(1, None)), (slice(1, None), slice(None, -1))

    def tri(x, i, j, k):
        return np.stack([x[i], x[j], x[k]], axis=2).reshape(-1, 3, x.shape[-1])

    pos = np.concatenate([tri(P, a, b, c), tri(P, a, c, d)], 0)
    nrm = np.concatenate([tri(N, a, b, c), tri(N, a, c, d)], 0)
    uv = np.concatenate([tri(UV, a, b, c), tri(UV, a, c, d)], 0)
    return pos, nrm, uv


def quad_grid(origin, eu, ev, nu, nv, normal, rng=None, jitter=0.0, uv_scale=1.0):
    """Planar patch origin + s*eu + t*ev, (nu x nv) cells, constant vertex normal."""
    origin, eu, ev, normal = (np.asarray(x, dtype=np.float32) for x in (origin, eu, ev, normal))
    s = np.linspace(0, 1, nu + 1, dtype=np.float32)
    t = np.linspace(0, 1, nv + 1, dtype=np.float32)
    S, T = np.meshgrid(s, t, indexing="xy")
    P = origin + S[..., None] * eu + T[..., None] * ev
    if rng is not None and jitter > 0:
        P = P + rng.uniform(-jitter, jitter, size=P.shape).astype(np.float32)
    N = np.broadcast_to(normal, P.shape).astype(np.float32)
    UV = np.stack([S * uv_scale, T * uv_scale], -1).astype(np.float32)
    return _grid_tris(P.astype(np.float32), N, UV)


def box(lo, hi, rot_y=0.0, centre=None):
    """Axis-aligned box [lo,hi] optionally rotated about +y through `centre`: 12 triangles, outward normals."""
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [  # (origin, eu, ev, normal) with eu x ev == normal direction
        ((x0, y0, z1), (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, 1)),
        ((x1, y0, z0), (x0 - x1, 0, 0), (0, y1 - y0, 0), (0, 0, -1)),
        ((x1, y0, z1), (0, 0, z0 - z1), (0, y1 - y0, 0), (1, 0, 0)),
        ((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0), (-1, 0, 0)),
        ((x0, y1, z1), (x1 - x0, 0, 0), (0, 0, z0 - z1), (0, 1, 0)),
        ((x0, y0, z0), (x1 - x0, 0, 0), (0, 0, z1 - z0), (0, -1, 0)),
    ]
    ps, ns, us = [], [], []
    for o, eu, ev, n in faces:
        p, nn, uv = quad_grid(o, eu, ev, 1, 1, n)
        ps.append(p), ns.append(nn), us.append(uv)
    P, N, UV = np.concatenate(ps), np.concatenate(ns), np.concatenate(us)
    if rot_y != 0.0:
        c = np.asarray(centre if centre is not None else (lo + hi) * 0.5, np.float32)
        ca, sa = np.float32(math.cos(rot_y)), np.float32(math.sin(rot_y))
        R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], np.float32)
        P = ((P - c) @ R.T + c).astype(np.float32)
        N = (N @ R.T).astype(np.float32)
    return P, N, UV


def displaced_sphere(centre, radius, res, rng, amp=0.0, freq=3.0, squash=(1.0, 1.0, 1.0)):
    """UV sphere (res x 2res cells) with a smooth seeded radial displacement; analytic-ish vertex normals
    (normalised radial direction, which is what a GEM export of a smooth object carries)."""
    nth, nph = int(res), int(2 * res)
    th = np.linspace(0.0, math.pi, nth + 1, dtype=np.float64)
    ph = np.linspace(0.0, 2.0 * math.pi, nph + 1, dtype=np.float64)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], -1)
    r = np.ones_like(TH)
    if amp > 0:
        k = rng.uniform(0.5, 1.5, size=(4, 3)) * freq
        phs = rng.uniform(0, 2 * math.pi, size=4)
        for i in range(4):
            r += (amp / 4.0) * np.sin(d @ k[i] + phs[i])
    # close the seam exactly so the mesh is watertight
    r[:, -1] = r[:, 0]
    d[:, -1] = d[:, 0]
    P = (np.asarray(centre, np.float64) + d * (r * radius)[..., None] * np.asarray(squash, np.float64)).astype(np.float32)
    N = d.astype(np.float32)
    UV = np.stack([PH / (2 * math.pi), TH / math.pi], -1).astype(np.float32)
    return _grid_tris(P, N, UV)


def heightfield(origin, size_x, size_z, nx, nz, rng, amp):
    """Displaced floor patch (carpet / water surface)."""
    xs = np.linspace(0, 1, nx + 1)
    zs = np.linspace(0, 1, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="xy")
    k = rng.uniform(4, 18, size=(5, 2))
    phs = rng.uniform(0, 2 * math.pi, size=5)
    H = np.zeros_like(X)
    for i in range(5):
        H += np.sin(X * k[i, 0] + Z * k[i, 1] + phs[i])
    H *= amp / 5.0
    P = np.stack([origin[0] + X * size_x, origin[1] + H, origin[2] + Z * size_z], -1).astype(np.float32)
    gx = np.gradient(H, axis=1) * nx / size_x
    gz = np.gradient(H, axis=0) * nz / size_z
    N = np.stack([-gx, np.ones_like(H), -gz], -1)
    N = (N / np.linalg.norm(N, axis=-1, keepdims=True)).astype(np.float32)
    UV = np.stack([X, Z], -1).astype(np.float32)
    # winding such that e1 x e2 points up (+y): swap to (a, d, c)/(a, c, b) ordering
    pos, nrm, uv = _grid_tris(P, N, UV)
    return pos[:, ::-1].copy(), nrm[:, ::-1].copy(), uv[:, ::-1].copy()


def checker_texture(n=256, cells=8, a=(0.85, 0.85, 0.85), b=(0.15, 0.2, 0.35)):
    i = (np.arange(n) * cells // n)
    m = ((i[:, None] + i[None, :]) & 1).astype(bool)
    t = np.where(m[..., None], np.asarray(a, np.float32), np.asarray(b, np.float32)).astype(np.float32)
    return np.ascontiguousarray(t)


def noise_texture(n, rng, base=(0.7, 0.65, 0.6), amp=0.2):
    t = np.asarray(base, np.float32) + rng.uniform(-amp, amp, size=(n, n, 1)).astype(np.float32)
    return np.ascontiguousarray(np.clip(t, 0.02, 0.98).astype(np.float32))


def solid(c):
    return np.asarray(c, np.float32).reshape(1, 1, 3).copy()


# ----------------------------------------------------------------------------- scenes
def cornell(width=1024, height=1024, with_specular=False):
    """Cornell box: 5 walls + ceiling light + short and tall box (34 triangles); 1024^2 like the
    reference's screenshots (RTBase/pathTrace.png).  `with_specular` swaps the tall box to mirror and
    adds a glass sphere so every pure-specular branch of pathTrace is exercised."""
    b = _Builder()
    tex = [solid((0.73, 0.73, 0.73)), solid((0.65, 0.05, 0.05)), solid((0.12, 0.45, 0.15)), solid((1, 1, 1))]
    mats = [Material(DIFFUSE, 0), Material(DIFFUSE, 1), Material(DIFFUSE, 2),
            Material(DIFFUSE, 3, emission=(17.0, 12.0, 4.0))]
    W, Hh = 1.0, 2.0
    b.add(*quad_grid((-W, 0, W), (2 * W, 0, 0), (0, 0, -2 * W), 1, 1, (0, 1, 0)), 0)       # floor
    b.add(*quad_grid((-W, Hh, -W), (2 * W, 0, 0), (0, 0, 2 * W), 1, 1, (0, -1, 0)), 0)     # ceiling
    b.add(*quad_grid((-W, 0, -W), (2 * W, 0, 0), (0, Hh, 0), 1, 1, (0, 0, 1)), 0)          # back
    b.add(*quad_grid((-W, 0, W), (0, 0, -2 * W), (0, Hh, 0), 1, 1, (1, 0, 0)), 1)          # left (red)
    b.add(*quad_grid((W, 0, -W), (0, 0, 2 * W), (0, Hh, 0), 1, 1, (-1, 0, 0)), 2)          # right (green)
    b.add(*quad_grid((-0.25, Hh - 0.01, -0.25), (0.5, 0, 0), (0, 0, 0.5), 1, 1, (0, -1, 0)), 3)  # light
    b.add(*box((0.15, 0.0, 0.1), (0.75, 0.6, 0.7), rot_y=-0.3), 0)                          # short box
    if with_specular:
        tex.append(solid((0.95, 0.95, 0.95)))
        mats.append(Material(MIRROR, 4))
        mats.append(Material(GLASS, 4, int_ior=1.5, ext_ior=1.0))
        b.add(*box((-0.75, 0.0, -0.75), (-0.15, 1.2, -0.15), rot_y=0.3), 4)
        b.add(*displaced_sphere((0.45, 0.95, 0.4), 0.3, 12, np.random.default_rng(7)), 5)
    else:
        b.add(*box((-0.75, 0.0, -0.75), (-0.15, 1.2, -0.15), rot_y=0.3), 0)                 # tall box
    v, m = b.finish()
    s = SceneData("cornell" + ("-spec" if with_specular else ""), v, m, mats, tex)
    return set_camera(s, (0, 1.0, 4.4), (0, 1.0, 0), (0, 1, 0), 40.0, width, height)


def open_box(width=96, height=96, background="env"):
    """Cornell box without ceiling or light under a sky: `background` "env" = lat-long EnvironmentMap (Lights.h:138),
    "colour" = BackgroundColour (Lights.h:87).  Exercises the non-area branch of computeDirect (Renderer.h:186-199) and
    the un-weighted background term of pathTrace (Renderer.h:254)."""
    b = _Builder()
    rng = np.random.default_rng(5)
    tex = [solid((0.73, 0.73, 0.73)), solid((0.65, 0.05, 0.05)), solid((0.12, 0.45, 0.15))]
    mats = [Material(DIFFUSE, 0), Material(DIFFUSE, 1), Material(DIFFUSE, 2)]
    W, Hh = 1.0, 2.0
    b.add(*quad_grid((-W, 0, W), (2 * W, 0, 0), (0, 0, -2 * W), 1, 1, (0, 1, 0)), 0)
    b.add(*quad_grid((-W, 0, -W), (2 * W, 0, 0), (0, Hh, 0), 1, 1, (0, 0, 1)), 0)
    b.add(*quad_grid((-W, 0, W), (0, 0, -2 * W), (0, Hh, 0), 1, 1, (1, 0, 0)), 1)
    b.add(*quad_grid((W, 0, -W), (0, 0, 2 * W), (0, Hh, 0), 1, 1, (-1, 0, 0)), 2)
    b.add(*box((-0.3, 0.0, -0.3), (0.3, 0.7, 0.3), rot_y=0.4), 0)
    v, m = b.finish()
    s = SceneData("open-box-" + background, v, m, mats, tex)
    if background == "env":
        yy, xx = np.meshgrid(np.linspace(0, 1, 32, dtype=np.float32), np.linspace(0, 1, 64, dtype=np.float32), indexing="ij")
        sky = np.stack([0.4 + 0.6 * (1 - yy), 0.5 + 0.5 * (1 - yy), 0.9 + 0 * yy], -1).astype(np.float32)
        sky += (8.0 * np.exp(-((xx - 0.3) ** 2 + (yy - 0.25) ** 2) / 0.004))[..., None].astype(np.float32)   # a sun
        sky += rng.uniform(0, 0.05, sky.shape).astype(np.float32)
        s.textures.append(np.ascontiguousarray(sky))
        s.bg_kind, s.bg_env_tex = 1, len(s.textures) - 1
    else:
        s.bg_kind, s.bg_colour = 0, (0.9, 1.0, 1.3)
    return set_camera(s, (0, 1.0, 4.4), (0, 0.8, 0), (0, 1, 0), 40.0, width, height)


def interior(variant="bathroom", detail=1.0, width=1920, height=1080, seed=1, conductor=False, dielectric=False, glazed=False):
    """Closed-room stand-in for the bathroom / living-room scenes: ~1.0 M triangles at detail=1
    (triangle count scales ~detail^2), tessellated displaced fixtures, thin slats (blinds), glass bulbs,
    a mirror, a tiled floor with a bilinear checker texture, diffuse / Oren-Nayar / glass / mirror materials (plastic
    and the rough dielectric only in the opt-in variants below) and four ceiling panels.  Vertices carry a <=1e-3 seeded jitter so there are no exact coplanar duplicates.

    The default scene holds only BSDFs whose upstream estimators stay finite (diffuse, Oren-Nayar, glass, mirror), so that
    its film is finite and the converged-image gate can run on it.  Upstream's non-finite cases are opt-in variants of the
    same geometry, reported separately: `conductor=True` makes the small sphere of the bathroom a rough conductor,
    `dielectric=True` a rough ("frosted") dielectric, whose half-vector draw (Materials.h:571) yields inf / NaN throughputs;
    `glazed=True` is SURVEY 8(d)'s full mix with PlasticBSDF on the fixtures, the blinds and the shelves.  A camera ray
    (whose direction upstream leaves un-normalised, |dir| ~ 1e-3, Scene.h:60; a mirror passes that length on) that meets
    plastic at grazing incidence has wo.z < 3e-8 in the shading frame, Fresnel rounds to exactly 1, kd = 0, the Phong term
    underflows and PlasticBSDF::sample returns f = 0 with pdf = 0 (Materials.h:861-930): 0 / 0.  On the smooth-shaded
    curved fixtures that is one NaN sample in 3 million at 1080p, on the flat slats one in 10^9 -- which of them depends
    on the last bit of the Fresnel term, so GPU and CPU do not even meet it on the same samples."""
    rng = np.random.default_rng(seed + (0 if variant == "bathroom" else 1000))
    b = _Builder()
    tex = [solid((0.8, 0.8, 0.78)),                       # 0 wall paint
           checker_texture(256, 8),                      # 1 floor tiles
           noise_texture(128, rng),                      # 2 speckled wall
           solid((1, 1, 1)),                             # 3 white
           solid((0.75, 0.25, 0.2)),                     # 4 terracotta
           solid((0.2, 0.35, 0.7)),                      # 5 blue
           solid((0.9, 0.9, 0.9)),                       # 6 porcelain
           solid((0.55, 0.4, 0.25))]                     # 7 wood
    mats = [Material(DIFFUSE, 0),                                       # 0 walls
            Material(DIFFUSE, 1),                                       # 1 floor
            Material(DIFFUSE, 2),                                       # 2 feature wall
            Material(DIFFUSE, 3, emission=(22.0, 20.0, 17.0)),          # 3 ceiling panels
            Material(PLASTIC, 6, int_ior=1.5, ext_ior=1.0, roughness=0.05) if glazed   # 4 porcelain fixtures (curved, smooth normals)
            else Material(DIFFUSE, 6),
            Material(GLASS, 3, int_ior=1.5, ext_ior=1.0),               # 5 glass
            Material(MIRROR, 3),                                        # 6 mirror
            Material(DIFFUSE, 4),                                       # 7
            Material(ORENNAYAR, 5, sigma=0.4),                          # 8
            Material(PLASTIC, 7, int_ior=1.4, ext_ior=1.0, roughness=0.2) if glazed else Material(DIFFUSE, 7),   # 9 slats / shelves
            Material(DIELECTRIC, 3, int_ior=1.33, ext_ior=1.0, roughness=0.1) if dielectric  # 10 frosted (one-sided), opt-in
            else (Material(PLASTIC, 5, int_ior=1.33, ext_ior=1.0, roughness=0.1) if glazed else Material(DIFFUSE, 5))]   # 10 blue ball
    if conductor:
        mats.append(Material(CONDUCTOR, 3, roughness=0.05, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14)))  # 11
    X, Y, Z = 4.0, 3.0, 3.0
    o = 0.02  # walls overshoot the corners so jitter cannot open cracks
    j = 1e-3
    wd = max(2, int(round(16 * detail)))
    b.add(*quad_grid((-X - o, 0, Z + o), (2 * X + 2 * o, 0, 0), (0, 0, -2 * Z - 2 * o), 3 * wd, 2 * wd, (0, 1, 0), rng, j, uv_scale=6.0), 1)
    b.add(*quad_grid((-X - o, Y, -Z - o), (2 * X + 2 * o, 0, 0), (0, 0, 2 * Z + 2 * o), wd, wd, (0, -1, 0), rng, j), 0)
    b.add(*quad_grid((-X - o, -o, -Z), (2 * X + 2 * o, 0, 0), (0, Y + 2 * o, 0), wd, wd, (0, 0, 1), rng, j, uv_scale=3.0), 2)
    b.add(*quad_grid((X + o, -o, Z), (-2 * X - 2 * o, 0, 0), (0, Y + 2 * o, 0), wd, wd, (0, 0, -1), rng, j), 0)
    b.add(*quad_grid((-X, -o, Z + o), (0, 0, -2 * Z - 2 * o), (0, Y + 2 * o, 0), wd, wd, (1, 0, 0), rng, j), 0)
    b.add(*quad_grid((X, -o, -Z - o), (0, 0, 2 * Z + 2 * o), (0, Y + 2 * o, 0), wd, wd, (-1, 0, 0), rng, j), 0)
    # ceiling light panels
    for cx, cz in ((-2.0, -1.2), (2.0, -1.2), (-2.0, 1.2), (2.0, 1.2)):
        b.add(*quad_grid((cx - 0.5, Y - 0.02, cz - 0.3), (1.0, 0, 0), (0, 0, 0.6), 1, 1, (0, -1, 0)), 3)
    # mirror on the feature wall
    b.add(*quad_grid((-1.5, 0.9, -Z + 0.03), (3.0, 0, 0), (0, 1.6, 0), 1, 1, (0, 0, 1)), 6)
    # blinds: thin slats in front of the -x wall
    nsl = max(4, int(round(60 * detail)))
    for i in range(nsl):
        y0 = 0.6 + 1.9 * i / nsl
        P, N, UV = box((-X + 0.08, y0, -1.6), (-X + 0.16, y0 + 0.012, 1.6))
        ang = np.float32(0.5)
        c, s_ = np.float32(math.cos(ang)), np.float32(math.sin(ang))
        R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)  # tilt about z
        ctr = np.array([-X + 0.12, y0 + 0.006, 0.0], np.float32)
        b.add(((P - ctr) @ R.T + ctr).astype(np.float32), (N @ R.T).astype(np.float32), UV, 9)
    # big tessellated fixtures
    res = max(6, int(round(250 * detail)))
    if variant == "bathroom":
        b.add(*displaced_sphere((-1.8, 0.55, -1.4), 0.9, res, rng, amp=0.12, freq=4.0, squash=(1.7, 0.6, 0.9)), 4)   # tub
        b.add(*displaced_sphere((2.4, 0.9, -2.2), 0.45, res, rng, amp=0.08, freq=5.0, squash=(1.0, 0.7, 1.0)), 4)    # basin
        b.add(*displaced_sphere((0.6, 0.5, 0.9), 0.5, res, rng, amp=0.2, freq=6.0), 7)                                 # stool
        b.add(*displaced_sphere((2.6, 0.45, 1.6), 0.45, res, rng, amp=0.05, freq=3.0), 5)                              # glass vase
        b.add(*displaced_sphere((-2.9, 0.4, 1.9), 0.4, max(6, res // 2), rng, amp=0.3, freq=7.0), 8)
        b.add(*displaced_sphere((-0.4, 0.35, -0.6), 0.35, max(6, res // 2), rng, amp=0.1, freq=5.0), 10 if not conductor else 11)
    else:
        nh = max(8, int(round(512 * detail)))
        b.add(*heightfield((-2.5, 0.03, -1.5), 5.0, 3.0, nh, nh, rng, 0.02), 8)                                        # carpet
        b.add(*displaced_sphere((-2.2, 0.6, -1.9), 0.7, res, rng, amp=0.1, freq=4.0, squash=(1.9, 0.8, 0.9)), 7)      # sofa
        b.add(*displaced_sphere((2.3, 0.55, -1.8), 0.6, res, rng, amp=0.15, freq=5.0, squash=(1.0, 0.9, 1.0)), 4)
        b.add(*displaced_sphere((0.2, 0.75, 0.2), 0.3, max(6, res // 2), rng, amp=0.05, freq=3.0), 5)
        for i in range(max(2, int(round(14 * detail)))):                                                                # shelves
            b.add(*box((X - 0.5, 0.3 + 0.18 * i, -2.0), (X - 0.05, 0.32 + 0.18 * i, 2.0)), 9)
    # glass bulbs under the panels
    for cx, cz in ((-2.0, -1.2), (2.0, 1.2)):
        b.add(*displaced_sphere((cx, Y - 0.35, cz), 0.12, max(6, int(round(24 * detail))), rng), 5)
    v, m = b.finish()
    s = SceneData(f"{variant}-class", v, m, mats, tex)
    return set_camera(s, (3.3, 1.65, 2.5), (-1.2, 0.9, -0.9), (0, 1, 0), 55.0, width, height)


def bathroom(detail=1.0, width=1920, height=1080, seed=1, **variant):
    return interior("bathroom", detail, width, height, seed, **variant)


def living_room(detail=1.0, width=1920, height=1080, seed=1):
    return interior("living-room", detail, width, height, seed)


def random_soup(n, seed=1, extent=4.0, size=0.6):
    """Unstructured triangle soup for traversal stress tests (deep, overlapping BVH)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n, 1, 3))
    p = (c + rng.uniform(-size, size, size=(n, 3, 3))).astype(np.float32)
    e1, e2 = p[:, 2] - p[:, 1], p[:, 0] - p[:, 2]
    nn = np.cross(e1, e2)
    nn = (nn / np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    b = _Builder()
    b.add(p, np.repeat(nn[:, None, :], 3, 1), np.zeros((n, 3, 2), np.float32), 0)
    v, m = b.finish()
    s = SceneData("soup", v, m, [Material(DIFFUSE, 0)], [solid((0.7, 0.7, 0.7))])
    return set_camera(s, (0, 0, 3 * extent), (0, 0, 0), (0, 1, 0), 45.0, 256, 256)


def axis_aligned_grid(n=8):
    """Axis-aligned unit quads on a lattice: exercises flat boxes, coplanar ties and rays in box planes."""
    b = _Builder()
    for i in range(n):
        for k in range(n):
            b.add(*quad_grid((i, 0, k), (1, 0, 0), (0, 0, 1), 1, 1, (0, 1, 0)), 0)
            b.add(*quad_grid((i, k, 0), (1, 0, 0), (0, 1, 0), 1, 1, (0, 0, 1)), 0)
    v, m = b.finish()
    s = SceneData("lattice", v, m, [Material(DIFFUSE, 0)], [solid((0.7, 0.7, 0.7))])
    return set_camera(s, (n / 2, n, 2 * n), (n / 2, 0, 0), (0, 1, 0), 45.0, 128, 128)


# ----------------------------------------------------------------------------- fixed ray sets
def primary_rays(scene: SceneData, n, seed=1):
    """Camera rays through uniformly random film positions; restates Camera::generateRay (Scene.h:47-64)
    in float64->float32 only to *produce inputs* -- both sides then consume the identical float32 rays."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, scene.width, n)
    y = rng.uniform(0, scene.height, n)
    P = scene.cam_P.reshape(4, 4).astype(np.float64)
    V = scene.cam_V.reshape(4, 4).astype(np.float64)
    iP = np.linalg.inv(P)
    xc = 2 * x / (scene.width - 1) - 1
    yc = 1 - 2 * y / (scene.height - 1)
    clip = np.stack([xc, yc, np.zeros(n), np.ones(n)], 0)
    pc = iP @ clip
    pc = pc / pc[3]
    pw = V @ pc
    pw = pw / pw[3]
    org = V @ np.array([0, 0, 0, 1.0])
    d = (pw[:3].T - org[:3]).astype(np.float32)
    o = np.broadcast_to(org[:3].astype(np.float32), d.shape)
    return np.ascontiguousarray(np.concatenate([o, d], 1).astype(np.float32))


def surface_points(scene: SceneData, n, rng):
    tri = rng.integers(0, scene.ntris, n)
    r1, r2 = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    a = 1 - np.sqrt(r1)
    bb = r2 * np.sqrt(r1)
    c = 1 - a - bb
    p = scene.verts[tri, :, 0:3].astype(np.float64)
    pos = p[:, 0] * a[:, None] + p[:, 1] * bb[:, None] + p[:, 2] * c[:, None]
    e1, e2 = p[:, 2] - p[:, 1], p[:, 0] - p[:, 2]
    nn = np.cross(e1, e2)
    nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-30)
    return pos, nn


def bounce_rays(scene: SceneData, n, seed=2):
    """Cosine-distributed secondary rays leaving random surface points (unit directions, EPSILON offset)."""
    rng = np.random.default_rng(seed)
    pos, nn = surface_points(scene, n, rng)
    flip = rng.uniform(0, 1, n) < 0.5
    nn[flip] *= -1
    r1, r2 = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    th, ph = np.arccos(np.sqrt(r1)), 2 * np.pi * r2
    loc = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1)
    t = np.where(np.abs(nn[:, :1]) > 0.9, np.array([[0, 1.0, 0]]), np.array([[1.0, 0, 0]]))
    u = np.cross(t, nn)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(nn, u)
    d = u * loc[:, :1] + v * loc[:, 1:2] + nn * loc[:, 2:3]
    o = pos + d * 1e-3
    return np.ascontiguousarray(np.concatenate([o, d], 1).astype(np.float32))


def shadow_segments(scene: SceneData, n, seed=3):
    """NEE-style segments between random surface points (p1 -> p2)."""
    rng = np.random.default_rng(seed)
    p1, _ = surface_points(scene, n, rng)
    p2, _ = surface_points(scene, n, rng)
    return np.ascontiguousarray(np.concatenate([p1, p2], 1).astype(np.float32))
