"""scene.json + .gem loader of the C++ host (raytracer_amd/host/gem.hpp) against scene directories written by
raytracer_amd/gem.py.  No upstream assets exist here, so parity is on the format as GEMLoader.h / SceneLoader.h read it:
the loader's output (dumped without touching a device) must equal what loadInstance would build, array for array."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import abi, api, gem, scenes
from test_host_cpp import _cli, read_hdr
from util import rel_mse


def dump(scene_dir, tmp_path, name="d.rtsc"):
    out = str(tmp_path / name)
    r = subprocess.run([_cli(), "-scene", str(scene_dir), "-dump", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return abi.load_rtsc(out), r.stdout


def same_material(got, want: scenes.Material):
    f32 = np.float32
    assert (got.type, got.albedo_tex, got.flags) == (want.type, want.albedo_tex, 0)
    assert tuple(got.emission) == tuple(f32(x) for x in want.emission)
    assert (got.int_ior, got.ext_ior, got.roughness, got.sigma) == (f32(want.int_ior), f32(want.ext_ior), f32(want.roughness), f32(want.sigma))
    assert tuple(got.eta) == tuple(f32(x) for x in want.eta) and tuple(got.k) == tuple(f32(x) for x in want.k)


@pytest.mark.parametrize("make,indexed", [
    (lambda: scenes.cornell(40, 30, with_specular=True), True),
    (lambda: scenes.interior("bathroom", 0.06, 64, 36, conductor=True, dielectric=True, glazed=True), True),
    (lambda: scenes.open_box(32, 32, "env"), False),
], ids=["cornell-spec", "bathroom", "envmap"])
def test_loader_reproduces_exported_scene(tmp_path, make, indexed):
    s = make()
    want = gem.export_scene(s, str(tmp_path / "scene"), indexed=indexed)
    got, out = dump(tmp_path / "scene", tmp_path)
    assert f"number of triangle: {want.ntris}" in out
    assert got.verts.shape == want.verts.shape and np.array_equal(got.verts.view(np.uint32), want.verts.view(np.uint32))   # bit for bit
    assert np.array_equal(got.tri_mat, want.tri_mat)
    assert len(got.materials) == len(want.materials)
    for g, w in zip(got.materials, want.materials):
        same_material(g, w)
    assert len(got.textures) == len(want.textures)
    for g, w in zip(got.textures, want.textures):
        assert g.shape == w.shape and np.array_equal(g, w)
    assert (got.bg_kind, got.bg_env_tex, got.width, got.height) == (want.bg_kind, want.bg_env_tex, want.width, want.height)
    assert np.allclose(got.cam_P, want.cam_P, rtol=1e-6, atol=0) and np.allclose(got.cam_V, want.cam_V, rtol=1e-5, atol=1e-6)


def test_instance_transform_multi_mesh_and_loader_rules(tmp_path):
    d = tmp_path / "scene"; d.mkdir()
    rng = np.random.default_rng(3)
    def mesh(ntri):
        v = np.zeros((ntri * 3, 11), np.float32)
        v[:, 0:3] = rng.uniform(-1, 1, (ntri * 3, 3)); v[:, 3:6] = rng.normal(size=(ntri * 3, 3)); v[:, 6:9] = (1, 0, 0); v[:, 9:11] = rng.uniform(0, 1, (ntri * 3, 2))
        return v
    a, b = mesh(4), mesh(3)
    a[3:6, 0:3] = a[3, 0:3]                                     # a zero-area triangle: dropped (SceneLoader.h:229)
    # two un-indexed meshes in one file: mesh 2's indices are offset by mesh 1's index count (SceneLoader.h:220), here == its vertex count
    gem.write_gem(str(d / "two.gem"), [({}, a, np.arange(12, dtype=np.uint32)), ({"k": "v"}, b, np.arange(9, dtype=np.uint32))])
    gem.write_gem(str(d / "one.gem"), [({}, b, np.arange(9, dtype=np.uint32))])
    gem.write_png(str(d / "red.png"), np.full((2, 2, 3), (255, 0, 51), np.uint8))
    ang = 0.7
    W = np.array([[2 * np.cos(ang), 0, 2 * np.sin(ang), 0.5], [0, 1.5, 0, -1.0], [-np.sin(ang), 0, np.cos(ang), 3.0], [0, 0, 0, 1]], np.float32)
    doc = {"width": 64, "height": 48, "fov": 35.5, "from": "0 1 5", "to": "0 0.5 0", "up": "0 1 0", "flipX": 1, "instances": [
        {"filename": "two.gem", "world": [float(x) for x in W.reshape(-1)], "bsdf": "plastic", "reflectance": "red.png", "roughness": 0.25, "emission": "1 2 3",
         "coatingThickness": 0.5, "coatingSigmaA": "1 1 1"},
        {"filename": "one.gem", "world": [1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], "bsdf": "velvet", "reflectance": "red.png"},       # unknown bsdf: dropped
        {"filename": "one.gem", "world": [1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], "bsdf": "dielectric", "reflectance": "missing.png", "roughness": 0.0005, "intIOR": 1.7},
    ]}
    (d / "scene.json").write_text(json.dumps(doc))
    got, out = dump(d, tmp_path)
    assert "Error in loading" in out
    L = api.lib()
    inv = np.zeros(16, np.float32)
    assert L.rt_matrix_invert(W.reshape(-1).ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p)) == 0
    N = inv.reshape(4, 4).T                                       # vecTransform = invert().transpose() (SceneLoader.h:199-200)
    as8 = lambda g: np.concatenate([g[:, 0:6], g[:, 9:11]], 1)
    exp = gem.transform_vertices(as8(np.concatenate([a, b], 0)), W, N).reshape(-1, 3, 8)
    keep = np.ones(7, bool); keep[1] = False
    I4 = np.eye(4, dtype=np.float32)
    exp = np.concatenate([exp[keep], gem.transform_vertices(as8(b), I4, I4).reshape(-1, 3, 8)], 0)
    assert got.verts.shape == exp.shape and np.array_equal(got.verts.view(np.uint32), exp.view(np.uint32))
    assert list(got.tri_mat) == [0] * 6 + [1] * 3
    m0, m1 = got.materials
    assert (m0.type, m0.albedo_tex, m0.flags, tuple(m0.emission), m0.roughness) == (scenes.PLASTIC, 0, 0, (1.0, 2.0, 3.0), 0.25)   # the coating leaves no trace
    assert (m0.int_ior, m0.ext_ior) == (np.float32(1.33), 1.0)
    assert (m1.type, m1.albedo_tex, m1.int_ior) == (scenes.GLASS, 1, np.float32(1.7))        # dielectric with roughness < 0.001 is glass (SceneLoader.h:152)
    assert np.array_equal(got.textures[0], np.full((2, 2, 3), (1.0, 0.0, 0.2), np.float32))
    assert got.textures[1].shape == (1, 1, 3) and (got.textures[1] == 1.0).all()              # unreadable texture -> loadDefault white
    assert got.cam_P[0] < 0 and (got.width, got.height) == (64, 48)                           # flipX negates P[0][0] (SceneLoader.h:253-257)


def test_rgbe_and_png_decoders(tmp_path):
    rng = np.random.default_rng(1)
    d = tmp_path / "scene"; d.mkdir()
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8); img[5:20, 10:40] = 7     # runs for the RLE / deflate back-references
    gem.write_png(str(d / "a.png"), img)
    hdr = (rng.uniform(0, 1, (9, 40, 3)) ** 4 * 50).astype(np.float32); hdr[2:5, 3:30] = 0.25
    want_flat = gem.write_hdr(str(d / "flat.hdr"), hdr, rle=False)
    want_rle = gem.write_hdr(str(d / "rle.hdr"), hdr, rle=True)
    v = np.zeros((3, 11), np.float32); v[:, 0:3] = np.eye(3); v[:, 3:6] = (0, 0, 1)
    gem.write_gem(str(d / "t.gem"), [({}, v, np.arange(3, dtype=np.uint32))])
    I = [1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    doc = {"width": 8, "height": 8, "fov": 45, "from": "0 0 3", "to": "0 0 0", "up": "0 1 0", "envmap": "rle.hdr", "instances": [
        {"filename": "t.gem", "world": I, "bsdf": "diffuse", "reflectance": "a.png"}, {"filename": "t.gem", "world": I, "bsdf": "mirror", "reflectance": "flat.hdr"},
        {"filename": "t.gem", "world": I, "bsdf": "diffuse", "reflectance": "a.png"}]}
    (d / "scene.json").write_text(json.dumps(doc))
    got, _ = dump(d, tmp_path)
    assert len(got.textures) == 3 and got.materials[2].albedo_tex == 0 and (got.bg_kind, got.bg_env_tex) == (1, 2)   # the texture manager de-duplicates by filename
    assert np.array_equal(got.textures[0], img.astype(np.float32) / np.float32(255.0))
    assert np.array_equal(got.textures[1], want_flat) and np.array_equal(got.textures[2], want_rle) and np.array_equal(want_flat, want_rle)


@pytest.mark.gpu
def test_render_from_gem_directory(tmp_path):
    s = scenes.cornell(64, 48, with_specular=True)
    want_scene = gem.export_scene(s, str(tmp_path / "scene"))
    out = tmp_path / "o.hdr"
    r = subprocess.run([_cli(), "-scene", str(tmp_path / "scene"), "-SPP", "16", "-sppPerCall", "16", "-seed", "4", "-outputFilename", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = read_hdr(str(out))
    sc = api.Scene(want_scene); rt = api.RayTracer(sc)
    rt.render(api.PATH, 16, 4, 0)
    want = rt.film()[0] / 16
    rt.close(); sc.close()
    # same triangles, materials and samples; the camera matrices differ in the last bits (fp32 MESA inverse vs numpy), so a few paths flip
    assert rel_mse(got, want) < 2e-3 and abs(got.mean() / want.mean() - 1) < 5e-3


def test_corrupt_counts_fail_cleanly(tmp_path):
    """Counts inside a .gem file (meshes, properties, vertices, indices) are checked against the file's size before anything is
    sized by them: a damaged file is an error message, not a multi-gigabyte allocation."""
    import struct
    s = scenes.cornell(32, 32)
    gem.export_scene(s, str(tmp_path / "scene"))
    mesh = sorted(p for p in os.listdir(tmp_path / "scene") if p.endswith(".gem"))[0]
    path = tmp_path / "scene" / mesh
    good = path.read_bytes()
    for pos, val in ((8, 0x7FFFFFFF), (12, 0xFFFFFFFF)):    # number of meshes, number of material properties of mesh 0
        path.write_bytes(good[:pos] + struct.pack("<I", val) + good[pos + 4:])
        r = subprocess.run([_cli(), "-scene", str(tmp_path / "scene"), "-dump", str(tmp_path / "x.rtsc")], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "truncated GE model file" in (r.stdout + r.stderr), (pos, r.stdout, r.stderr)
    path.write_bytes(good[:len(good) // 2])
    r = subprocess.run([_cli(), "-scene", str(tmp_path / "scene"), "-dump", str(tmp_path / "x.rtsc")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "truncated" in (r.stdout + r.stderr)


def test_hand_assembled_gem_bytes_against_reference_transforms(tmp_path):
    """SURVEY 8(f1) against bytes this repository's writer (raytracer_amd/gem.py) never touched: the .gem file is packed here,
    field by field, from the layout GEMModelLoader::load / loadMesh read (GEMLoader.h:218-365: magic 4058972161, isAnimated,
    mesh count; per mesh: property count + (length, bytes) string pairs, vertex count + 44-byte GEMStaticVertex records
    {position, normal, tangent, u, v}, index count + u32 indices), scene.json is literal text with the parser's corner cases
    (exponent notation, true / false, numbers that pass through std::to_string's six decimals, GEMLoader.h:459-480), and the
    expected world-space vertices are the COMPILED REFERENCE's Matrix::mulPoint / invert().transpose().mulVec().normalize()
    outputs (tests/golden/scalars_r2.npz, SceneLoader.h:196-219), compared bit for bit."""
    import struct
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scalars_r2.npz"))
    W = g["worlds"][1]
    pos, nrm, want_p, want_n = g["pos"][1], g["nrm"][1], g["mul_point"][1], g["normal_xf"][1]
    rng = np.random.default_rng(17)
    uv = rng.uniform(0, 1, (48, 2)).astype(np.float32)

    def lstr(b):
        return struct.pack("<i", len(b)) + b

    def mesh(lo, hi, indices, props):
        out = struct.pack("<I", len(props))
        for k, v in props:
            out += lstr(k) + lstr(v)
        out += struct.pack("<I", hi - lo)
        for i in range(lo, hi):   # GEMStaticVertex: position, normal, tangent, u, v = 11 floats
            out += struct.pack("<11f", *pos[i], *nrm[i], 1.0, 0.0, 0.0, *uv[i])
        out += struct.pack("<I", len(indices)) + struct.pack(f"<{len(indices)}I", *indices)
        return out

    idx0 = [int(i) for i in rng.permutation(24)]                       # 8 triangles over mesh 0's 24 vertices, shuffled
    idx1 = [int(i) for i in rng.permutation(24)]                       # mesh 1: offset by mesh 0's INDEX count (SceneLoader.h:220)
    blob = struct.pack("<III", 4058972161, 0, 2) + mesh(0, 24, idx0, [(b"name", b"first"), (b"note", b"")]) + mesh(24, 48, idx1, [])
    d = tmp_path / "scene"; d.mkdir()
    (d / "hand.gem").write_bytes(blob)
    world = ", ".join("%.9e" % float(x) for x in W.reshape(-1))
    (d / "scene.json").write_text('{\n  "width": 6.4e1, "height": 48,\n  "fov": 3.55E+1, "from": "0 1 5", "to": "0 0.5 0", "up": "0 1 0", "flipX": true,\n'
                                  '  "instances": [\n    {"world": [' + world + '], "reflectance": "none.png", "filename": "hand.gem", "intIOR": 1.3333333333,\n'
                                  '     "roughness": 2.5e-1, "bsdf": "plastic", "emission": "0 0 0", "unused": false}\n  ]\n}\n')
    got, out = dump(d, tmp_path)
    assert "number of triangle: 16" in out
    exp = np.zeros((16, 3, 8), np.float32)
    for t in range(16):
        for k in range(3):
            i = idx0[t * 3 + k] if t < 8 else 24 + idx1[(t - 8) * 3 + k]
            exp[t, k] = np.concatenate([want_p[i], want_n[i], uv[i]])
    assert got.verts.shape == exp.shape
    assert np.array_equal(got.verts.view(np.uint32), exp.view(np.uint32))                      # positions, normals, uv: bit for bit
    m, = got.materials
    assert m.type == scenes.PLASTIC and m.roughness == np.float32(0.25)
    assert m.int_ior == np.float32(float("%.6f" % 1.3333333333)) != np.float32(1.3333333333)    # "1.333333": std::to_string keeps six decimals
    assert (got.width, got.height) == (64, 48) and got.cam_P[0] < 0                            # 6.4e1 -> 64; flipX: true -> "1"
