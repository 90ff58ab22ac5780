"""ctypes mirror of include/rtamd.h (descriptor structs + scene marshalling).

Only data-layout code lives here; the library loader and the RTBase-style
host objects are in raytracer_amd/api.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .scenes import SceneData


class Material(C.Structure):  # rt_material (64 B)
    _fields_ = [("type", C.c_uint32), ("albedo_tex", C.c_int32), ("flags", C.c_uint32),
                ("emission", C.c_float * 3), ("int_ior", C.c_float), ("ext_ior", C.c_float),
                ("roughness", C.c_float), ("eta", C.c_float * 3), ("k", C.c_float * 3), ("sigma", C.c_float)]


class Texture(C.Structure):  # rt_texture
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p), ("alpha", C.c_void_p)]


class Background(C.Structure):  # rt_background
    _fields_ = [("kind", C.c_uint32), ("colour", C.c_float * 3), ("env_tex", C.c_int32)]


class Camera(C.Structure):  # rt_camera
    _fields_ = [("P", C.c_float * 16), ("V", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32)]


class Hit(C.Structure):  # rt_hit == IntersectionData (Geometry.h:256)
    _fields_ = [("ID", C.c_uint32), ("t", C.c_float), ("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float)]


HIT_DTYPE = np.dtype([("ID", np.uint32), ("t", np.float32), ("alpha", np.float32), ("beta", np.float32), ("gamma", np.float32)])
assert HIT_DTYPE.itemsize == 20 and C.sizeof(Hit) == 20 and C.sizeof(Material) == 64


class SceneDesc(C.Structure):  # rt_scene_desc
    _fields_ = [("verts", C.c_void_p), ("tri_material", C.c_void_p), ("n_tris", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32),
                ("background", Background)]


class Marshalled:
    """Owns every buffer a scene descriptor points into."""

    def __init__(self, s: SceneData):
        self.verts = np.ascontiguousarray(s.verts, dtype=np.float32)
        self.tri_mat = np.ascontiguousarray(s.tri_mat, dtype=np.uint32)
        assert self.verts.ndim == 3 and self.verts.shape[1:] == (3, 8)
        self.n_tris = self.verts.shape[0]
        self.mats = (Material * max(1, len(s.materials)))()
        for i, m in enumerate(s.materials):
            d = self.mats[i]
            d.type, d.albedo_tex, d.flags = m.type, m.albedo_tex, m.flags
            d.emission[:] = m.emission
            d.int_ior, d.ext_ior, d.roughness, d.sigma = m.int_ior, m.ext_ior, m.roughness, m.sigma
            d.eta[:] = m.eta
            d.k[:] = m.k
        self.n_mats = len(s.materials)
        self.tex_data = [np.ascontiguousarray(t, dtype=np.float32) for t in s.textures]
        self.tex = (Texture * max(1, len(self.tex_data)))()
        for i, t in enumerate(self.tex_data):
            assert t.ndim == 3 and t.shape[2] == 3
            self.tex[i].height, self.tex[i].width = t.shape[0], t.shape[1]
            self.tex[i].rgb = t.ctypes.data
            self.tex[i].alpha = None
        self.n_tex = len(self.tex_data)
        self.bg = Background()
        self.bg.kind = s.bg_kind
        self.bg.colour[:] = s.bg_colour
        self.bg.env_tex = s.bg_env_tex
        self.cam = Camera()
        if s.cam_P is not None:
            self.cam.P[:] = [float(x) for x in s.cam_P]
            self.cam.V[:] = [float(x) for x in s.cam_V]
            self.cam.width, self.cam.height = s.width, s.height

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        d.verts = self.verts.ctypes.data
        d.tri_material = self.tri_mat.ctypes.data
        d.n_tris = self.n_tris
        d.materials = C.cast(self.mats, C.POINTER(Material))
        d.n_materials = self.n_mats
        d.textures = C.cast(self.tex, C.POINTER(Texture))
        d.n_textures = self.n_tex
        d.background = self.bg
        return d


RTSC_MAGIC = 0x43535452  # "RTSC"


def save_rtsc(s: SceneData, path: str) -> None:
    """Flat little-endian scene file read by the C++ host driver (raytracer_amd/host/rtamd.hpp loadScene):
    header {magic, version=1, n_tris, n_materials, n_textures, width, height, P[16], V[16], bg_kind, bg_colour[3], bg_env_tex},
    rt_material[n_materials], per texture {w, h, has_alpha, rgb[w*h*3]}, tri_material[n_tris], verts[n_tris*24]."""
    m = Marshalled(s)
    with open(path, "wb") as f:
        f.write(np.array([RTSC_MAGIC, 1, m.n_tris, m.n_mats, m.n_tex], np.uint32).tobytes())
        f.write(np.array([s.width, s.height], np.int32).tobytes())
        f.write(np.asarray(s.cam_P, np.float32).tobytes())
        f.write(np.asarray(s.cam_V, np.float32).tobytes())
        f.write(np.array([s.bg_kind], np.uint32).tobytes())
        f.write(np.asarray(s.bg_colour, np.float32).tobytes())
        f.write(np.array([s.bg_env_tex], np.int32).tobytes())
        f.write(bytes(m.mats)[: 64 * m.n_mats])
        for t in m.tex_data:
            f.write(np.array([t.shape[1], t.shape[0], 0], np.int32).tobytes())
            f.write(t.tobytes())
        f.write(m.tri_mat.tobytes())
        f.write(m.verts.tobytes())


def load_rtsc(path: str) -> SceneData:
    """Reader for save_rtsc / the C++ host's saveRTSC (materials come back as abi.Material structs)."""
    b = open(path, "rb").read()
    magic, ver, nt, nm, ntex = np.frombuffer(b, np.uint32, 5, 0)
    assert magic == RTSC_MAGIC and ver == 1
    w, h = np.frombuffer(b, np.int32, 2, 20)
    o = 28
    P = np.frombuffer(b, np.float32, 16, o).copy(); o += 64
    V = np.frombuffer(b, np.float32, 16, o).copy(); o += 64
    bg_kind = int(np.frombuffer(b, np.uint32, 1, o)[0]); o += 4
    bg_colour = tuple(float(x) for x in np.frombuffer(b, np.float32, 3, o)); o += 12
    bg_env = int(np.frombuffer(b, np.int32, 1, o)[0]); o += 4
    mats = [Material.from_buffer_copy(b, o + 64 * i) for i in range(nm)]; o += 64 * int(nm)
    tex = []
    for _ in range(ntex):
        tw, th, _a = np.frombuffer(b, np.int32, 3, o); o += 12
        tex.append(np.frombuffer(b, np.float32, int(tw) * int(th) * 3, o).reshape(int(th), int(tw), 3).copy()); o += int(tw) * int(th) * 12
    tm = np.frombuffer(b, np.uint32, int(nt), o).copy(); o += 4 * int(nt)
    verts = np.frombuffer(b, np.float32, int(nt) * 24, o).reshape(int(nt), 3, 8).copy(); o += 96 * int(nt)
    assert o == len(b)
    return SceneData("rtsc", verts, tm, mats, tex, bg_kind, bg_colour, bg_env, P, V, int(w), int(h))
