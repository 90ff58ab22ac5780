"""C++ host mirror (raytracer_amd/host/rtamd.hpp) and the headless driver built on it (Main.cpp's arguments)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from raytracer_amd import abi, api, scenes
from util import rel_mse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracer_amd", "host", "rtamd_cli")


def _cli():
    if not os.path.exists(CLI):  # built by `make host` / __graft_entry__.build(); the GPU box uses the prebuilt binary
        subprocess.run(["make", "-C", ROOT, "host"], check=True)
    return CLI


def read_hdr(path):
    """Flat (non-RLE) Radiance RGBE, as Film::save writes through stbi_write_hdr for small widths (Imaging.h:278)."""
    b = open(path, "rb").read()
    head, _, rest = b.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE")
    line, _, data = rest.partition(b"\n")
    t = line.split()
    h, w = int(t[1]), int(t[3])
    px = np.frombuffer(data, np.uint8, w * h * 4).reshape(h, w, 4).astype(np.float32)
    scale = np.where(px[..., 3:] > 0, np.exp2(px[..., 3:] - 136.0), 0.0)
    return px[..., :3] * scale


def test_cli_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([_cli(), "-scene", "cornell-box", "-SPP", "1", "-width", "16", "-height", "16"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    r = subprocess.run([_cli(), "-scene", "/nonexistent/scene"], capture_output=True, text=True)
    assert r.returncode == 1 and "loadScene" in r.stderr


def test_rtsc_export_layout(tmp_path):
    s = scenes.cornell(32, 24)
    p = tmp_path / "c.rtsc"
    abi.save_rtsc(s, str(p))
    b = p.read_bytes()
    magic, ver, nt, nm, ntex = struct.unpack_from("<5I", b, 0)
    w, h = struct.unpack_from("<2i", b, 20)
    assert (magic, ver, nt, nm, ntex, w, h) == (abi.RTSC_MAGIC, 1, s.ntris, len(s.materials), len(s.textures), 32, 24)
    tex_bytes = sum(12 + t.size * 4 for t in s.textures)
    assert len(b) == 20 + 8 + 128 + 20 + 64 * nm + tex_bytes + 4 * nt + 96 * nt
    assert np.array_equal(np.frombuffer(b[-96 * nt:], np.float32).reshape(nt, 3, 8), s.verts)


def test_rtcamera_key_presses(tmp_path):
    """RTCamera (SceneLoader.h:8-88) as Main.cpp drives it with W/S/A/D/E/Q: the C++ mirror's view matrix after a key sequence
    against the same steps restated here (float32, Rodrigues rotation about `up`, movespeed = 5 % of the scene diagonal)."""
    f = np.float32
    frm, to, up = np.array([0, 1.0, 4.4], f), np.array([0, 1.0, 0], f), np.array([0, 1, 0], f)   # the built-in box (rtamd.hpp cornellBox)
    box = subprocess.run([_cli(), "-scene", "cornell-box", "-width", "32", "-height", "32", "-dump", str(tmp_path / "a.rtsc")], capture_output=True, text=True, timeout=60)
    assert box.returncode == 0, box.stderr
    s0 = abi.load_rtsc(str(tmp_path / "a.rtsc"))
    p = s0.verts[:, :, :3].reshape(-1, 3)
    move = f(np.sqrt(((p.max(0) - p.min(0)).astype(f) ** 2).sum(dtype=f)) * f(0.05))
    nrm = lambda v: (v * (f(1) / np.sqrt(np.dot(v, v)))).astype(f)
    keys = "WWADDEQSAAW"
    for k in keys:
        d = nrm(to - frm)
        if k == "W": frm = frm + d * move; to = frm + d * move
        elif k == "S": frm = frm - d * move; to = frm + d * move
        elif k in "AD":
            rad = f((5.0 if k == "A" else -5.0) * f(np.pi / 180.0))
            c, sn = f(np.cos(rad)), f(np.sin(rad))
            to = frm + (d * c + np.cross(up, d).astype(f) * sn + up * (np.dot(up, d) * (f(1) - c))).astype(f)
        elif k == "E": frm = frm + up * move; to = to + up * move
        elif k == "Q": frm = frm - up * move; to = to - up * move
        frm, to = frm.astype(f), to.astype(f)
    want = np.linalg.inv(scenes.look_at(frm, to, up).reshape(4, 4).astype(np.float64)).astype(f)
    r = subprocess.run([_cli(), "-scene", "cornell-box", "-width", "32", "-height", "32", "-keys", keys, "-dump", str(tmp_path / "b.rtsc")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = np.asarray(abi.load_rtsc(str(tmp_path / "b.rtsc")).cam_V, f).reshape(4, 4)
    assert np.allclose(got, want, atol=2e-5), (got, want)
    assert not np.allclose(got, np.asarray(s0.cam_V, f).reshape(4, 4), atol=1e-3)
    bad = subprocess.run([_cli(), "-scene", "cornell-box", "-keys", "X", "-dump", str(tmp_path / "c.rtsc")], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "-keys" in bad.stderr


@pytest.mark.gpu
def test_cli_fly_camera_moves_the_render(tmp_path):
    """Main.cpp:87-116 headless: a key press moves the camera and clears the film, so the saved image is the last view's."""
    outs = []
    for keys in ("", "WWA", "WWAD"):
        out = tmp_path / f"k{len(keys)}.hdr"
        r = subprocess.run([_cli(), "-scene", "cornell-box", "-width", "48", "-height", "48", "-SPP", "6", "-sppPerCall", "2", "-outputFilename", str(out)]
                           + (["-keys", keys] if keys else []), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(read_hdr(str(out)))
    assert np.isfinite(outs[1]).all() and outs[1].mean() > 0
    assert np.abs(outs[0] - outs[1]).mean() > 0.02 * outs[0].mean()      # another view
    assert np.abs(outs[1] - outs[2]).mean() > 0.02 * outs[0].mean()


@pytest.mark.gpu
def test_cli_aov_films(tmp_path):
    """-aov: RayTracer::getAOVFilms (albedoFilm / normalFilm of getTileFilms, Renderer.h:577-603) through the C++ host equals
    the AOV modes rendered through the Python host, and the beauty render beside it is not disturbed."""
    s = scenes.cornell(64, 48)
    abi.save_rtsc(s, str(tmp_path / "scene.rtsc"))
    outs = []
    for extra in ([], ["-aov", str(tmp_path / "a")]):
        out = tmp_path / f"o{len(extra)}.hdr"
        r = subprocess.run([_cli(), "-scene", str(tmp_path), "-SPP", "4", "-sppPerCall", "4", "-seed", "9", "-outputFilename", str(out)] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    sc = api.Scene(s); rt = api.RayTracer(sc)
    for mode, name in ((api.AOV_ALBEDO, "albedo"), (api.AOV_NORMAL, "normal")):
        want = rt.renderAOV(mode, 1, 9, 0)
        got = np.fromfile(str(tmp_path / f"a.{name}.f32"), np.float32).reshape(want.shape)
        assert np.array_equal(got, want) and np.abs(want).max() > 0
    rt.close(); sc.close()


@pytest.mark.gpu
def test_cli_matches_python_host(tmp_path):
    s = scenes.cornell(64, 48)
    abi.save_rtsc(s, str(tmp_path / "scene.rtsc"))
    out = tmp_path / "o.hdr"
    r = subprocess.run([_cli(), "-scene", str(tmp_path), "-SPP", "12", "-sppPerCall", "5", "-seed", "9", "-outputFilename", str(out), "-png", "1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    assert "triangle:" in lines and len([x for x in lines if x.replace(".", "").replace("e-", "").isdigit()]) >= 3   # one time per render() call
    got = read_hdr(str(out))
    sc = api.Scene(s); rt = api.RayTracer(sc)
    rt.render(api.PATH, 12, 9, 0)
    want = rt.film()[0] / 12
    rt.close(); sc.close()
    assert got.shape == want.shape
    assert rel_mse(got, want) < 1e-4 and np.abs(got - want).max() <= want.max() / 100   # RGBE keeps 8 mantissa bits
    png = (tmp_path / "o.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", png[16:24])
    i = png.index(b"IDAT")
    n = struct.unpack(">I", png[i - 4:i])[0]
    raw = zlib.decompress(png[i + 4:i + 4 + n])
    assert (w, h) == (64, 48) and len(raw) == (64 * 3 + 1) * 48
    img = np.frombuffer(raw, np.uint8).reshape(48, 64 * 3 + 1)[:, 1:].reshape(48, 64, 3)
    tm = np.clip(np.power(np.maximum(want, 0), 1 / 2.2) * 255, 0, 255)   # Film::tonemap (Imaging.h:251)
    assert np.abs(img.astype(np.float32) - tm).mean() < 1.5


@pytest.mark.gpu
def test_cli_builtin_cornell_and_modes(tmp_path):
    for typ, extra in ((0, []), (1, []), (4, ["-vpls", "20"]), (2, [])):
        out = tmp_path / f"t{typ}.hdr"
        r = subprocess.run([_cli(), "-scene", "cornell-box", "-width", "48", "-height", "48", "-SPP", "2", "-type", str(typ),
                            "-outputFilename", str(out)] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        img = read_hdr(str(out))
        assert img.shape == (48, 48, 3) and np.isfinite(img).all() and img.mean() > 0


@pytest.mark.gpu
def test_reference_unit_test_through_cpp_host():
    """RTTest.cpp's triangle case via rtamd::Scene::traverse / visible (tests/host/rttest.cpp)."""
    exe = os.path.join(ROOT, "tests", "host", "rttest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "host"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
