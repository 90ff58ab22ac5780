"""TEST INFRASTRUCTURE — ctypes loaders for the CPU oracle (oracle/liboracle.so)
and, when it has been built in a container that holds /root/reference, for the
compiled reference geometry (oracle/_ref/libref_geom.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from raytracer_amd.abi import Background, Camera, HIT_DTYPE, Marshalled, Material, Texture
from raytracer_amd.scenes import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.environ.get("ORC_SO") or os.path.join(_HERE, "liboracle.so")   # ORC_SO: another build of rt_oracle.c (tools/bench_cpu_flags.py)
REF_SO = os.path.join(_HERE, "_ref", "libref_geom.so")


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Material), C.c_uint32,
                                       C.POINTER(Texture), C.c_uint32, C.POINTER(Background)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.orc_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_leaf_order.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tri_derived.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        for f in (L.orc_traverse, L.orc_traverse_brute, L.orc_visible):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        for f in (L.orc_traverse_mt, L.orc_visible_mt):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_generate_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_project.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_project.restype = C.c_int
        L.orc_mlt_render.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_mlt_render.restype = None
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_vpl_trace.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.orc_vpl_trace.restype = C.c_int
        L.orc_vpl_count.argtypes = [C.c_void_p]
        L.orc_vpl_count.restype = C.c_int
        L.orc_vpl_get.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_filter.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_filter.restype = None
        L.orc_filter_eval.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_filter_eval.restype = C.c_float
        L.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_bsdf_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rng_floats.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_sampling.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_matrix.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_tri_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_tri_shading.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_tri_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tri_intersect.restype = C.c_int
        L.orc_aabb_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_aabb_ray.restype = C.c_int
        L.orc_aabb_default_area.restype = C.c_float
        L.orc_texture_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_env_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_env_probe.restype = C.c_int
        L.orc_env_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_env_tables.restype = C.c_float
        L.orc_trace_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_trace_path.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """CPU oracle scene: Scene::init + Scene::build + camera."""

    def __init__(self, s: SceneData):
        self.L = lib()
        self.m = Marshalled(s)
        d = self.m.desc()
        self.h = self.L.orc_scene_create(d.verts, d.tri_material, d.n_tris, d.materials, d.n_materials,
                                         d.textures, d.n_textures, C.byref(self.m.bg))
        if s.cam_P is not None:
            self.L.orc_scene_set_camera(self.h, C.byref(self.m.cam))
        self.width, self.height, self.ntris = s.width, s.height, s.ntris

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_filter(self, kind):
        """film->init(w, h, new BoxFilter / GaussianFilter / MitchellNetravaliFilter) (Renderer.h:61): 0 / 1 / 2."""
        self.L.orc_set_filter(self.h, kind)

    def info(self):
        st = np.zeros(5, np.int32)
        self.L.orc_scene_info(self.h, st.ctypes.data)
        return dict(depth=int(st[0]), nodes=int(st[1]), leaves=int(st[2]), lights=int(st[3]), max_leaf=int(st[4]))

    def leaf_order(self):
        o = np.zeros(self.ntris, np.uint32)
        self.L.orc_leaf_order(self.h, o.ctypes.data)
        return o

    def tri_derived(self, i):
        o = np.zeros(11, np.float32)
        self.L.orc_tri_derived(self.h, i, o.ctypes.data)
        return o

    def traverse(self, rays, threads=1, brute=False):
        rays = _f32(rays)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        if brute:
            self.L.orc_traverse_brute(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data)
        elif threads > 1:
            self.L.orc_traverse_mt(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data, threads)
        else:
            self.L.orc_traverse(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data)
        return out

    def visible(self, segs, threads=1):
        segs = _f32(segs)
        out = np.zeros(segs.shape[0], np.uint8)
        if threads > 1:
            self.L.orc_visible_mt(self.h, segs.ctypes.data, segs.shape[0], out.ctypes.data, threads)
        else:
            self.L.orc_visible(self.h, segs.ctypes.data, segs.shape[0], out.ctypes.data)
        return out

    def generate_rays(self, xy):
        xy = _f32(xy)
        out = np.zeros((xy.shape[0], 6), np.float32)
        self.L.orc_generate_rays(self.h, xy.ctypes.data, xy.shape[0], out.ctypes.data)
        return out

    def project(self, p):
        p = _f32(p)
        xy = np.zeros(2, np.float32)
        ok = self.L.orc_project(self.h, p.ctypes.data, xy.ctypes.data)
        return bool(ok), xy

    def render(self, mode=0, spp=1, seed=1, first_sample=0, threads=8, film=None):
        if film is None:
            film = np.zeros((self.height, self.width, 3), np.float32)
        st = Stats()
        self.L.orc_render(self.h, mode, spp, seed, first_sample, film.ctypes.data, threads, C.byref(st))
        stats = dict(closest_rays=st.closest_rays, any_rays=st.any_rays, nodes_visited=st.nodes_visited, tris_tested=st.tris_tested)
        return film, stats

    def mlt_render(self, spp=1, seed=1, first_sample=0, film=None):
        """renderSTMLT (Renderer.h:353): 100 000 proposals per spp, un-normalised film sum."""
        if film is None:
            film = np.zeros((self.height, self.width, 3), np.float32)
        self.L.orc_mlt_render(self.h, spp, seed, first_sample, film.ctypes.data)
        return film

    def vpl_trace(self, n, seed=1):
        return self.L.orc_vpl_trace(self.h, n, seed)

    def vpls(self):
        n = self.L.orc_vpl_count(self.h)
        out = np.zeros((n, 10), np.float32)
        if n:
            self.L.orc_vpl_get(self.h, out.ctypes.data)
        return out

    def bsdf_sample(self, mat, wo, n, uv, key, ctr):
        wo, n, uv = _f32(wo), _f32(n), _f32(uv)
        out = np.zeros(7, np.float32)
        self.L.orc_bsdf_sample(self.h, mat, wo.ctypes.data, n.ctypes.data, uv.ctypes.data, key, ctr, out.ctypes.data)
        return out

    def bsdf_eval(self, mat, wo, n, uv, wi):
        wo, n, uv, wi = _f32(wo), _f32(n), _f32(uv), _f32(wi)
        out = np.zeros(4, np.float32)
        self.L.orc_bsdf_eval(self.h, mat, wo.ctypes.data, n.ctypes.data, uv.ctypes.data, wi.ctypes.data, out.ctypes.data)
        return out

    def env_probe(self, which, in3):
        """EnvironmentMap sample (0) / SampleContinuous (1) / PDF (2): in [n, 3] -> out [n, 4]."""
        a = _f32(in3).reshape(-1, 3)
        out = np.zeros((a.shape[0], 4), np.float32)
        for i in range(a.shape[0]):
            assert self.L.orc_env_probe(self.h, which, a[i].ctypes.data, out[i].ctypes.data) == 1, "scene has no environment map"
        return out

    def env_tables(self, nu, nv):
        """(conditional pdfs [nv, nu], marginal pdfs [nv], funcInt) of EnvironmentMap::build (Lights.h:207-253)."""
        cond, marg = np.zeros((nv, nu), np.float32), np.zeros(nv, np.float32)
        f = self.L.orc_env_tables(self.h, cond.ctypes.data, marg.ctypes.data)
        return cond, marg, float(f)

    def texture_sample(self, tex, u, v):
        out = np.zeros(3, np.float32)
        self.L.orc_texture_sample(self.h, tex, u, v, out.ctypes.data)
        return out


def mlt_trace(seed, sample, n):
    """What orc_mlt_render's sampler hands the chain for the first n proposals (no tracing)."""
    out = np.zeros((n, 4), np.float32)
    L = lib()
    L.orc_mlt_trace.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    L.orc_mlt_trace.restype = None
    L.orc_mlt_trace(seed, sample, n, out.ctypes.data)
    return {"large": out[:, 0] < np.float32(0.3), "xy": out[:, 1:3].copy(), "accept_draw": out[:, 3].copy()}


def rng_floats(seed, sample, pixel, stream, n):
    out = np.zeros(n, np.float32)
    lib().orc_rng_floats(seed, sample, pixel, stream, n, out.ctypes.data)
    return out


# ----------------------------------------------------------------------------- compiled reference (geometry only)
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_build.restype = C.c_void_p
        L.ref_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ref_free.argtypes = [C.c_void_p]
        for f in (L.ref_traverse, L.ref_traverse_brute, L.ref_visible):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ref_leaf_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_tri_derived.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_tri_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_tri_intersect.restype = C.c_int
        L.ref_plane_intersect.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.ref_plane_intersect.restype = C.c_int
        L.ref_aabb_default_area.restype = C.c_float
        L.ref_aabb_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_aabb_ray.restype = C.c_int
        L.ref_tri_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.ref_tri_shading.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.ref_sampling.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.ref_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_matrix.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.ref_mul_point_div.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_mtrandom.argtypes = [C.c_uint, C.c_int, C.c_void_p]
        if hasattr(L, "ref_vec2_index"):   # round-2 probes (an older _ref build lacks them)
            L.ref_vec2_index.argtypes = [C.c_float, C.c_float, C.c_int]
            L.ref_vec2_index.restype = C.c_int
            L.ref_matrix_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            L.ref_tri_centre.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "ref_pss_trace"):    # round-3 probe
            L.ref_pss_trace.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        _ref = L
    return _ref


class RefScene:
    """The reference's own Triangle::init + BVHNode::build/traverse/traverseVisible."""

    def __init__(self, s: SceneData):
        self.L = ref()
        self.verts = np.ascontiguousarray(s.verts, np.float32)
        self.mats = np.ascontiguousarray(s.tri_mat, np.uint32)
        self.ntris = s.ntris
        self.h = self.L.ref_build(self.verts.ctypes.data, self.mats.ctypes.data, self.ntris)

    def close(self):
        if self.h:
            self.L.ref_free(self.h)
            self.h = None

    def traverse(self, rays, brute=False):
        rays = _f32(rays)
        out = np.zeros(rays.shape[0], HIT_DTYPE)
        (self.L.ref_traverse_brute if brute else self.L.ref_traverse)(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data)
        return out

    def visible(self, segs):
        segs = _f32(segs)
        out = np.zeros(segs.shape[0], np.uint8)
        self.L.ref_visible(self.h, segs.ctypes.data, segs.shape[0], out.ctypes.data)
        return out

    def leaf_order(self):
        o = np.zeros(self.ntris, np.uint32)
        st = np.zeros(3, np.int32)
        self.L.ref_leaf_order(self.h, o.ctypes.data, st.ctypes.data)
        return o, dict(depth=int(st[0]), nodes=int(st[1]), leaves=int(st[2]))

    def tri_derived(self, i):
        o = np.zeros(11, np.float32)
        self.L.ref_tri_derived(self.h, i, o.ctypes.data)
        return o
