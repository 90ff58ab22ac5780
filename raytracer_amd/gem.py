"""Writer for upstream's scene directories (scene.json + .gem meshes + PNG / Radiance textures): the on-disk format
`loadScene` reads (SceneLoader.h:236-289, GEMLoader.h:218-365, :679-738).  No such assets ship with the reference, so
this is how the C++ host's loader (raytracer_amd/host/gem.hpp) gets exercised, and how the procedural stand-ins here
can be handed to an unmodified RTBase build.

`export_scene` returns the SceneData the loader must reproduce from the files it wrote: one material per instance in
instance order, textures in order of first use and quantised to what the image file holds, normals re-normalised the way
loadInstance does (SceneLoader.h:214-215), zero-area triangles dropped (:229)."""
from __future__ import annotations

import json
import os
import struct
import zlib

import numpy as np

from . import scenes
from .scenes import Material, SceneData

GEM_MAGIC = 4058972161
BSDF_NAMES = {scenes.DIFFUSE: "diffuse", scenes.MIRROR: "mirror", scenes.CONDUCTOR: "conductor", scenes.GLASS: "glass",
              scenes.DIELECTRIC: "dielectric", scenes.ORENNAYAR: "orennayar", scenes.PLASTIC: "plastic"}


def write_png(path, rgb8):
    """8-bit RGB PNG, zlib-compressed scanlines with filter type 0/1/2 mixed (exercises the loader's unfiltering)."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = rgb8.shape
    rows = []
    for y in range(h):
        cur = rgb8[y].reshape(-1).astype(np.int16)
        ft = y % 3 if y else 0
        if ft == 1:
            prev = np.concatenate([np.zeros(3, np.int16), cur[:-3]])
            cur = cur - prev
        elif ft == 2:
            cur = cur - rgb8[y - 1].reshape(-1).astype(np.int16)
        rows.append(bytes([ft]) + (cur & 255).astype(np.uint8).tobytes())
    def chunk(tag, data):
        c = tag + data
        return struct.pack(">I", len(data)) + c + struct.pack(">I", zlib.crc32(c) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(b"".join(rows), 6)) + chunk(b"IEND", b""))


def rgbe_encode(rgb):
    rgb = np.asarray(rgb, np.float32)
    m = rgb.max(-1)
    e = np.zeros(m.shape, np.int32)
    mant, ex = np.frexp(np.where(m > 1e-32, m, 1.0))
    scale = np.where(m > 1e-32, mant * 256.0 / np.where(m > 1e-32, m, 1.0), 0.0)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, ex + 128, 0).astype(np.uint8)
    return out


def rgbe_decode(rgbe):
    """stbi__hdr_convert: value = mantissa * 2^(e - 136), no rounding offset."""
    e = rgbe[..., 3:].astype(np.int32)
    f = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), 0).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * f).astype(np.float32)


def write_hdr(path, rgb, rle=False):
    rgbe = rgbe_encode(rgb)
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        if not rle or w < 8:
            f.write(rgbe.tobytes())
        else:
            for y in range(h):
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for c in range(4):
                    ch = rgbe[y, :, c]
                    x = 0
                    while x < w:
                        run = 1
                        while x + run < w and run < 127 and ch[x + run] == ch[x]:
                            run += 1
                        if run >= 3:
                            f.write(bytes([128 + run, int(ch[x])])); x += run
                        else:
                            n = min(128, w - x)
                            f.write(bytes([n]) + ch[x:x + n].tobytes()); x += n
    return rgbe_decode(rgbe)


def write_gem(path, meshes):
    """meshes: list of (properties: dict, vertices float32 [n, 11] {pos, normal, tangent, u, v}, indices uint32 [m])."""
    def s(t):
        t = t.encode()
        return struct.pack("<i", len(t)) + t
    with open(path, "wb") as f:
        f.write(struct.pack("<III", GEM_MAGIC, 0, len(meshes)))
        for props, v, idx in meshes:
            f.write(struct.pack("<I", len(props)))
            for k, val in props.items():
                f.write(s(k) + s(str(val)))
            v = np.ascontiguousarray(v, np.float32); idx = np.ascontiguousarray(idx, np.uint32)
            assert v.ndim == 2 and v.shape[1] == 11
            f.write(struct.pack("<I", v.shape[0]) + v.tobytes() + struct.pack("<I", idx.shape[0]) + idx.tobytes())


def _normalize_f32(n):
    n = np.asarray(n, np.float32)
    l = np.float32(1.0) / np.sqrt((n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1]) + n[..., 2] * n[..., 2], dtype=np.float32)
    return (n * l[..., None]).astype(np.float32)


def transform_vertices(v, W, N):
    """loadInstance's per-vertex transform in the same fp32 operation order (SceneLoader.h:213-215; Matrix::mulPoint / mulVec,
    Core.h:295-309): v [.., 8] {p, normal, u, v}; W world matrix, N = transpose(invert(W)), both [4, 4] float32."""
    v = np.asarray(v, np.float32); W = np.asarray(W, np.float32); N = np.asarray(N, np.float32)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    P = np.stack([((x * W[r, 0] + y * W[r, 1]) + z * W[r, 2]) + W[r, 3] for r in range(3)], -1).astype(np.float32)
    nx, ny, nz = v[..., 3], v[..., 4], v[..., 5]
    T = np.stack([((nx * N[r, 0] + ny * N[r, 1]) + nz * N[r, 2]) for r in range(3)], -1).astype(np.float32)
    return np.concatenate([P, _normalize_f32(T), v[..., 6:8]], -1).astype(np.float32)


def export_scene(s: SceneData, out_dir: str, indexed=True) -> SceneData:
    """One instance (identity world matrix, one single-mesh .gem) per material that has triangles."""
    os.makedirs(out_dir, exist_ok=True)
    cp = s.cam_params
    assert cp is not None, "scene has no from/to/up/fov (use scenes.set_camera)"
    tex_files, new_tex, tex_of = {}, [], {}

    def texture(i):
        if i not in tex_files:
            t = np.asarray(s.textures[i], np.float32)
            if float(t.max()) > 1.0:   # HDR values cannot live in an 8-bit PNG
                name = f"tex{i}.hdr"
                val = write_hdr(os.path.join(out_dir, name), t, rle=t.shape[1] >= 8)
            else:
                name = f"tex{i}.png"
                q = np.clip(np.rint(t * 255.0), 0, 255).astype(np.uint8)
                write_png(os.path.join(out_dir, name), q)
                val = (q.astype(np.float32) / np.float32(255.0)).astype(np.float32)   # Imaging.h:59
            tex_files[i] = name
            tex_of[i] = len(new_tex)
            new_tex.append(val)
        return tex_files[i]

    instances, verts_out, mat_out, new_mats = [], [], [], []
    for mi, m in enumerate(s.materials):
        sel = np.nonzero(s.tri_mat == mi)[0]
        if sel.size == 0:
            continue
        v = s.verts[sel]                                   # [n, 3, 8]
        flat = v.reshape(-1, 8)
        if indexed:
            uniq, inv = np.unique(flat.view(np.dtype((np.void, 32))), return_inverse=True)
            vtx = uniq.view(np.float32).reshape(-1, 8)
            idx = inv.astype(np.uint32).reshape(-1)
        else:
            vtx, idx = flat, np.arange(flat.shape[0], dtype=np.uint32)
        gv = np.zeros((vtx.shape[0], 11), np.float32)
        gv[:, 0:6] = vtx[:, 0:6]; gv[:, 6] = 1.0; gv[:, 9:11] = vtx[:, 6:8]
        fname = f"mesh{mi}.gem"
        write_gem(os.path.join(out_dir, fname), [({"name": fname}, gv, idx)])
        inst = {"filename": fname, "world": [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0],
                "bsdf": BSDF_NAMES[m.type], "reflectance": texture(m.albedo_tex)}
        if m.type in (scenes.GLASS, scenes.PLASTIC, scenes.DIELECTRIC):
            inst["intIOR"] = float(m.int_ior); inst["extIOR"] = float(m.ext_ior)
        if m.type in (scenes.PLASTIC, scenes.DIELECTRIC, scenes.CONDUCTOR):
            inst["roughness"] = float(m.roughness)
        if m.type == scenes.CONDUCTOR:
            inst["eta"] = " ".join(repr(float(x)) for x in m.eta); inst["k"] = " ".join(repr(float(x)) for x in m.k)
        if m.type == scenes.ORENNAYAR:
            inst["alpha"] = float(m.sigma)
        if any(e > 0 for e in m.emission):
            inst["emission"] = " ".join(repr(float(x)) for x in m.emission)
        instances.append(inst)
        # what loadInstance makes of it
        e = transform_vertices(v, np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32))   # identity still turns -0.0 into +0.0
        p = e[:, :, 0:3]
        e1, e2 = p[:, 2] - p[:, 1], p[:, 0] - p[:, 2]
        c = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], -1).astype(np.float32)
        area = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2], dtype=np.float32) * np.float32(0.5)
        e = e[area > 0]
        verts_out.append(e)
        mat_out.append(np.full(e.shape[0], len(new_mats), np.uint32))
        f32 = lambda x: float(np.float32(x))
        nm = Material(m.type, tex_of[m.albedo_tex], emission=tuple(f32(x) for x in m.emission))
        # loader defaults for what scene.json does not carry (SceneLoader.h:126-168)
        nm.int_ior, nm.ext_ior = (f32(m.int_ior), f32(m.ext_ior)) if "intIOR" in inst else (f32(1.33), 1.0)
        nm.roughness = f32(m.roughness) if "roughness" in inst else 1.0
        nm.sigma = f32(m.sigma) if "alpha" in inst else 1.0
        if m.type == scenes.CONDUCTOR:
            nm.eta, nm.k = tuple(f32(x) for x in m.eta), tuple(f32(x) for x in m.k)
        if m.type == scenes.DIELECTRIC and nm.roughness < 0.001:
            nm.type = scenes.GLASS
        new_mats.append(nm)
    doc = {"width": int(s.width), "height": int(s.height), "fov": float(cp["fov"]),
           "from": " ".join(repr(x) for x in cp["frm"]), "to": " ".join(repr(x) for x in cp["to"]), "up": " ".join(repr(x) for x in cp["up"]),
           "instances": instances}
    if cp.get("flip_x"):
        doc["flipX"] = 1
    bg_env = -1
    if s.bg_kind == 1:
        doc["envmap"] = texture(s.bg_env_tex)
        bg_env = tex_of[s.bg_env_tex]
    with open(os.path.join(out_dir, "scene.json"), "w") as f:
        json.dump(doc, f, indent=1)
    out = SceneData(s.name + "-gem", np.ascontiguousarray(np.concatenate(verts_out, 0)), np.concatenate(mat_out, 0), new_mats, new_tex,
                    1 if s.bg_kind == 1 else 0, (0.0, 0.0, 0.0), bg_env)
    scenes.set_camera(out, cp["frm"], cp["to"], cp["up"], cp["fov"], s.width, s.height, cp.get("flip_x", False))
    return out
