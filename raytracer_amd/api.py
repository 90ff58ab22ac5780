"""RTBase-style host objects over the C ABI (include/rtamd.h).

`Scene` and `RayTracer` keep the reference's method names for the hot path --
Scene.build/traverse/visible (Scene.h:92, :118, :169), RayTracer.init/render/
clear/getSPP/saveHDR (Renderer.h:56, :374, :107, :652, :657) -- but operate on
batches.  There is no CPU implementation behind them: if librtamd.so (the HIP
library) is missing or no GPU is visible, they raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .abi import Camera, HIT_DTYPE, Marshalled, SceneDesc
from .scenes import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB", os.path.join(_HERE, "librtamd.so"))  # RT_LIB: A/B builds of the same HIP library

PATH, LIGHT, MLT, RADIOSITY, PATH_ST, AOV_ALBEDO, AOV_NORMAL, DIRECT = 0, 1, 2, 4, 5, 10, 11, 12
FILTER_BOX, FILTER_GAUSSIAN, FILTER_MITCHELL = 0, 1, 2  # ImageFilter of the film (Imaging.h:138-214)


class RtError(RuntimeError):
    pass


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_lights", C.c_uint32), ("bvh2_nodes", C.c_uint32), ("bvh2_depth", C.c_uint32),
                ("bvh2_leaves", C.c_uint32), ("bvh2_max_leaf", C.c_uint32), ("wide_nodes", C.c_uint32), ("wide_max_stack", C.c_uint32),
                ("device_bytes", C.c_uint64), ("build_seconds", C.c_float), ("wide_levels", C.c_uint32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Counters(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("exact_path_rays", C.c_uint64), ("any_nodes_visited", C.c_uint64), ("any_tris_tested", C.c_uint64),
                ("lane_slots", C.c_uint64), ("lane_busy", C.c_uint64), ("any_lane_slots", C.c_uint64), ("any_lane_busy", C.c_uint64),
                ("max_steps", C.c_uint64), ("any_max_steps", C.c_uint64), ("gate_refused", C.c_uint64), ("any_gate_refused", C.c_uint64)]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# DNode8 / DTri of csrc/rt_common.h as the walk reads them
WALK_NODE_DTYPE = np.dtype([("org", np.float32, 3), ("exp", np.int8, 3), ("imask", np.uint8), ("child_base", np.uint32), ("tri_base", np.uint32),
                            ("meta", np.uint8, 8), ("lo", np.uint8, (3, 8)), ("hi", np.uint8, (3, 8))])
WALK_TRI_DTYPE = np.dtype([("v", np.float32, (3, 3)), ("n", np.float32, 3), ("d", np.float32), ("inv_area", np.float32), ("rank", np.uint32), ("mat_gsign", np.uint32)])
assert WALK_NODE_DTYPE.itemsize == 80 and WALK_TRI_DTYPE.itemsize == 64

_lib = None

# every symbol include/rtamd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "rt_last_error", "rt_device_count", "rt_device_set", "rt_scene_create", "rt_scene_destroy", "rt_scene_set_camera",
    "rt_scene_get_info", "rt_scene_leaf_order", "rt_host_build_info", "rt_intersect_closest", "rt_intersect_closest_dev",
    "rt_intersect_any", "rt_intersect_any_dev", "rt_generate_rays", "rt_counters_enable", "rt_counters_read",
    "rt_renderer_create", "rt_renderer_destroy", "rt_render", "rt_film_clear", "rt_film_spp", "rt_film_read", "rt_film_bind",
    "rt_film_set_spp", "rt_vpl_trace", "rt_vpl_count", "rt_render_counters", "rt_renderer_sync", "rt_film_save_hdr",
    "rt_film_tonemap", "rt_profile_enable", "rt_profile_read", "rt_debug_bsdf_sample", "rt_debug_bsdf_eval", "rt_debug_rng", "rt_matrix_invert", "rt_scene_create_ex",
    "rt_renderer_set_filter", "rt_render_aov", "rt_intersect_closest_noalloc", "rt_renderer_set_streams", "rt_debug_env", "rt_scene_debug_ref_nodes",
    "rt_scene_debug_walk_nodes", "rt_scene_debug_tri_classes",
]


def _sig(L, name, argtypes=None, restype=None, set_restype=False):
    # a symbol missing from a stale build surfaces as AttributeError at call time (and in tests/test_abi.py)
    try:
        f = getattr(L, name)
    except AttributeError:
        return
    if argtypes is not None:
        f.argtypes = argtypes
    if set_restype:
        f.restype = restype


def lib():
    """Loads librtamd.so; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} not found: build it with `make product` (hipcc --offload-arch=gfx950). "
                          "raytracer_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        _sig(L, "rt_last_error", None, C.c_char_p, True)
        vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
        _sig(L, "rt_scene_create", [C.POINTER(SceneDesc), C.POINTER(vp)])
        _sig(L, "rt_scene_create_ex", [C.POINTER(SceneDesc), C.c_uint32, C.POINTER(vp)])
        _sig(L, "rt_scene_destroy", [vp])
        _sig(L, "rt_scene_destroy", None, None, True)
        _sig(L, "rt_scene_set_camera", [vp, C.POINTER(Camera)])
        _sig(L, "rt_scene_get_info", [vp, C.POINTER(SceneInfo)])
        _sig(L, "rt_scene_leaf_order", [vp, vp])
        _sig(L, "rt_scene_debug_ref_nodes", [vp, vp, sz])
        _sig(L, "rt_scene_debug_walk_nodes", [vp, vp, sz, vp, sz, vp])
        _sig(L, "rt_scene_debug_tri_classes", [vp, vp, sz])
        _sig(L, "rt_host_build_info", [C.POINTER(SceneDesc), vp, C.POINTER(SceneInfo)])
        _sig(L, "rt_intersect_closest", [vp, vp, sz, vp, i32])
        _sig(L, "rt_intersect_closest_dev", [vp, vp, sz, vp, i32, vp])
        _sig(L, "rt_intersect_closest_noalloc", [vp, vp, sz, vp, i32, vp])
        _sig(L, "rt_intersect_any", [vp, vp, sz, vp, i32])
        _sig(L, "rt_intersect_any_dev", [vp, vp, sz, vp, i32, vp])
        _sig(L, "rt_generate_rays", [vp, vp, sz, vp])
        _sig(L, "rt_counters_enable", [vp, i32])
        _sig(L, "rt_counters_read", [vp, C.POINTER(Counters), i32])
        _sig(L, "rt_renderer_create", [vp, C.POINTER(vp)])
        _sig(L, "rt_renderer_destroy", [vp])
        _sig(L, "rt_renderer_destroy", None, None, True)
        _sig(L, "rt_render", [vp, i32, i32, u64, u64])
        _sig(L, "rt_film_clear", [vp])
        _sig(L, "rt_film_spp", [vp])
        _sig(L, "rt_film_read", [vp, vp, C.POINTER(i32)])
        _sig(L, "rt_film_bind", [vp, vp])
        _sig(L, "rt_film_set_spp", [vp, i32])
        _sig(L, "rt_vpl_trace", [vp, i32, u64])
        _sig(L, "rt_renderer_set_filter", [vp, i32])
        _sig(L, "rt_renderer_set_streams", [vp, i32])
        _sig(L, "rt_render_aov", [vp, i32, i32, u64, u64, vp])
        _sig(L, "rt_vpl_count", [vp])
        _sig(L, "rt_render_counters", [vp, C.POINTER(Counters), i32])
        _sig(L, "rt_renderer_sync", [vp])
        _sig(L, "rt_profile_enable", [vp, i32])
        _sig(L, "rt_profile_read", [vp, vp, vp, i32])
        _sig(L, "rt_film_save_hdr", [vp, C.c_char_p])
        _sig(L, "rt_film_tonemap", [vp, vp])
        _sig(L, "rt_debug_bsdf_sample", [vp, C.c_uint32, vp, vp, vp, vp, sz, vp])
        _sig(L, "rt_debug_bsdf_eval", [vp, C.c_uint32, vp, vp, vp, vp, sz, vp])
        _sig(L, "rt_debug_rng", [u64, u64, C.c_uint32, C.c_uint32, i32, vp])
        _sig(L, "rt_debug_env", [vp, i32, vp, sz, vp])
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RtError(f"{what} failed ({rc}): {lib().rt_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def host_build_info(s: SceneData):
    """Leaf order + tree statistics from the host builder only (no GPU needed)."""
    m = Marshalled(s)
    d = m.desc()
    order = np.zeros(s.ntris, np.uint32)
    info = SceneInfo()
    _check(lib().rt_host_build_info(C.byref(d), order.ctypes.data, C.byref(info)), "rt_host_build_info")
    return order, info.asdict()


def rng_floats(seed, sample, pixel, stream, n):
    out = np.zeros(n, np.float32)
    _check(lib().rt_debug_rng(seed, sample, pixel, stream, n, out.ctypes.data), "rt_debug_rng")
    return out


class Scene:
    """Scene::init + Scene::build on the GPU (Scene.h:150, :92)."""

    def __init__(self, s: SceneData, device_build=None):   # None: the library's default (device), False: host build
        self.L = lib()
        self.data = s
        self.m = Marshalled(s)
        d = self.m.desc()
        h = C.c_void_p()
        _check(self.L.rt_scene_create_ex(C.byref(d), C.c_uint32(0 if device_build is None else (1 if device_build else 2)), C.byref(h)), "rt_scene_create")
        self.h = h
        self.width, self.height, self.ntris = s.width, s.height, s.ntris
        if s.cam_P is not None:
            self.set_camera(self.m.cam)

    def set_camera(self, cam: Camera):
        _check(self.L.rt_scene_set_camera(self.h, C.byref(cam)), "rt_scene_set_camera")
        self.width, self.height = cam.width, cam.height

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = SceneInfo()
        _check(self.L.rt_scene_get_info(self.h, C.byref(i)), "rt_scene_get_info")
        return i.asdict()

    def leaf_order(self):
        o = np.zeros(self.ntris, np.uint32)
        _check(self.L.rt_scene_leaf_order(self.h, o.ctypes.data), "rt_scene_leaf_order")
        return o

    def ref_nodes(self):
        """The reference-topology binary tree from device memory: structured array of 48-B nodes (csrc/rt_common.h DNode2)."""
        n = self.info()["bvh2_nodes"]
        out = np.zeros(n, np.dtype([("lo", np.float32, 3), ("left", np.int32), ("hi", np.float32, 3), ("right", np.int32), ("first", np.int32),
                                    ("count", np.int32), ("sub_first", np.int32), ("sub_count", np.int32)]))
        _check(self.L.rt_scene_debug_ref_nodes(self.h, out.ctypes.data, n), "rt_scene_debug_ref_nodes")
        return out

    def walk_nodes(self):
        """(nodes, gate, trisw): the 80-B 8-wide nodes the traversal kernels read (csrc/rt_common.h DNode8), the per-triangle gate
        boxes (DFS leaf order) and the triangle records in the order the 8-wide tree names them (DFS rank in `rank`)."""
        n = self.info()["wide_nodes"]
        nodes = np.zeros(n, WALK_NODE_DTYPE)
        gate = np.zeros(self.ntris, np.dtype([("lo", np.float32, 3), ("pad0", np.float32), ("hi", np.float32, 3), ("pad1", np.float32)]))
        trisw = np.zeros(self.ntris, WALK_TRI_DTYPE)
        _check(self.L.rt_scene_debug_walk_nodes(self.h, nodes.ctypes.data, n, gate.ctypes.data, self.ntris, trisw.ctypes.data), "rt_scene_debug_walk_nodes")
        return nodes, gate, trisw

    def tri_classes(self):
        """One byte per triangle (DFS leaf order): BSDF class | 0x80 for an emitter, as ExtendJob::retire reads it."""
        out = np.zeros(self.ntris, np.uint8)
        _check(self.L.rt_scene_debug_tri_classes(self.h, out.ctypes.data, self.ntris), "rt_scene_debug_tri_classes")
        return out

    # Scene::traverse (Scene.h:118)
    def traverse(self, rays, mode=0):
        rays = _f32(rays).reshape(-1, 6)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        _check(self.L.rt_intersect_closest(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data, mode), "rt_intersect_closest")
        return out

    # Scene::visible (Scene.h:169)
    def visible(self, segs, mode=0):
        segs = _f32(segs).reshape(-1, 6)
        out = np.zeros(segs.shape[0], np.uint8)
        _check(self.L.rt_intersect_any(self.h, segs.ctypes.data, segs.shape[0], out.ctypes.data, mode), "rt_intersect_any")
        return out

    def traverse_dev(self, d_rays_ptr, n, d_out_ptr, mode=0, stream=None):
        _check(self.L.rt_intersect_closest_dev(self.h, d_rays_ptr, n, d_out_ptr, mode, stream), "rt_intersect_closest_dev")

    def traverse_noalloc(self, d_rays_ptr, n, d_out_ptr, mode=0, stream=None):
        _check(self.L.rt_intersect_closest_noalloc(self.h, d_rays_ptr, n, d_out_ptr, mode, stream), "rt_intersect_closest_noalloc")

    def visible_dev(self, d_segs_ptr, n, d_out_ptr, mode=0, stream=None):
        _check(self.L.rt_intersect_any_dev(self.h, d_segs_ptr, n, d_out_ptr, mode, stream), "rt_intersect_any_dev")

    # Camera::generateRay (Scene.h:47)
    def generate_rays(self, xy):
        xy = _f32(xy).reshape(-1, 2)
        out = np.zeros((xy.shape[0], 6), np.float32)
        _check(self.L.rt_generate_rays(self.h, xy.ctypes.data, xy.shape[0], out.ctypes.data), "rt_generate_rays")
        return out

    def counters_enable(self, on=True):
        _check(self.L.rt_counters_enable(self.h, 1 if on else 0), "rt_counters_enable")

    def counters(self, reset=True):
        c = Counters()
        _check(self.L.rt_counters_read(self.h, C.byref(c), 1 if reset else 0), "rt_counters_read")
        return c.asdict()

    def bsdf_sample(self, material, wo, n, uv, key_ctr):
        wo, n, uv = _f32(wo).reshape(-1, 3), _f32(n).reshape(-1, 3), _f32(uv).reshape(-1, 2)
        kc = np.ascontiguousarray(key_ctr, np.uint32).reshape(-1, 2)
        out = np.zeros((wo.shape[0], 7), np.float32)
        _check(self.L.rt_debug_bsdf_sample(self.h, material, wo.ctypes.data, n.ctypes.data, uv.ctypes.data, kc.ctypes.data,
                                           wo.shape[0], out.ctypes.data), "rt_debug_bsdf_sample")
        return out

    def env_probe(self, which, in3):
        """EnvironmentMap sample (0) / SampleContinuous (1) / PDF (2) on the device: in [n, 3] -> out [n, 4]."""
        a = _f32(in3).reshape(-1, 3)
        out = np.zeros((a.shape[0], 4), np.float32)
        _check(self.L.rt_debug_env(self.h, which, a.ctypes.data, a.shape[0], out.ctypes.data), "rt_debug_env")
        return out

    def bsdf_eval(self, material, wo, n, uv, wi):
        wo, n, uv, wi = _f32(wo).reshape(-1, 3), _f32(n).reshape(-1, 3), _f32(uv).reshape(-1, 2), _f32(wi).reshape(-1, 3)
        out = np.zeros((wo.shape[0], 4), np.float32)
        _check(self.L.rt_debug_bsdf_eval(self.h, material, wo.ctypes.data, n.ctypes.data, uv.ctypes.data, wi.ctypes.data,
                                         wo.shape[0], out.ctypes.data), "rt_debug_bsdf_eval")
        return out


class RayTracer:
    """RayTracer (Renderer.h:29): init / render / clear / getSPP / saveHDR, film on the GPU."""

    def __init__(self, scene: Scene):
        self.L = lib()
        self.scene = scene
        h = C.c_void_p()
        _check(self.L.rt_renderer_create(scene.h, C.byref(h)), "rt_renderer_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, mode=PATH, spp=1, seed=1, first_sample=0):
        _check(self.L.rt_render(self.h, mode, spp, seed, first_sample), "rt_render")

    def sync(self):
        _check(self.L.rt_renderer_sync(self.h), "rt_renderer_sync")

    def clear(self):
        _check(self.L.rt_film_clear(self.h), "rt_film_clear")

    def getSPP(self):
        return self.L.rt_film_spp(self.h)

    def film(self):
        """Un-normalised accumulation film [H, W, 3] and SPP (Film::film, Imaging.h:219)."""
        out = np.zeros((self.scene.height, self.scene.width, 3), np.float32)
        spp = C.c_int()
        _check(self.L.rt_film_read(self.h, out.ctypes.data, C.byref(spp)), "rt_film_read")
        return out, spp.value

    def bind_film(self, device_ptr):
        _check(self.L.rt_film_bind(self.h, device_ptr), "rt_film_bind")

    def set_spp(self, spp):
        _check(self.L.rt_film_set_spp(self.h, spp), "rt_film_set_spp")

    def setFilter(self, kind):
        """film->init(w, h, new BoxFilter / GaussianFilter / MitchellNetravaliFilter) (Renderer.h:61): FILTER_BOX / _GAUSSIAN / _MITCHELL."""
        _check(self.L.rt_renderer_set_filter(self.h, kind), "rt_renderer_set_filter")

    def set_streams(self, n):
        _check(self.L.rt_renderer_set_streams(self.h, n), "rt_renderer_set_streams")

    def renderAOV(self, mode, spp=1, seed=1, first_sample=0):
        """albedoFilm / normalFilm of getTileFilms (Renderer.h:577-603): an AOV pass into a film of its own (un-normalised)."""
        out = np.zeros((self.scene.height, self.scene.width, 3), np.float32)
        _check(self.L.rt_render_aov(self.h, mode, spp, seed, first_sample, out.ctypes.data), "rt_render_aov")
        return out

    def traceVPLs(self, n, seed=1):
        _check(self.L.rt_vpl_trace(self.h, n, seed), "rt_vpl_trace")
        return self.L.rt_vpl_count(self.h)

    def counters(self, reset=True):
        c = Counters()
        _check(self.L.rt_render_counters(self.h, C.byref(c), 1 if reset else 0), "rt_render_counters")
        return c.asdict()

    KERNELS = ("generate", "extend", "shade", "connect", "gather", "accumulate", "control")

    def profile_enable(self, on=True):
        _check(self.L.rt_profile_enable(self.h, 1 if on else 0), "rt_profile_enable")

    def profile(self, reset=True):
        """{kernel: (total ms, launches)} from HIP events on the renderer's stream."""
        ms = np.zeros(7, np.float64)
        n = np.zeros(7, np.uint32)
        _check(self.L.rt_profile_read(self.h, ms.ctypes.data, n.ctypes.data, 1 if reset else 0), "rt_profile_read")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.KERNELS)}

    def saveHDR(self, filename):
        _check(self.L.rt_film_save_hdr(self.h, filename.encode()), "rt_film_save_hdr")

    def tonemap(self):
        out = np.zeros((self.scene.height, self.scene.width, 3), np.uint8)
        _check(self.L.rt_film_tonemap(self.h, out.ctypes.data), "rt_film_tonemap")
        return out
