"""Device build of the reference-topology tree (csrc/bvh_build.hip) against the host build: same leaf
order, same tree statistics, same wide tree size, and bit-identical traversal results."""
import time

import numpy as np
import pytest

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def both(s):
    return api.Scene(s, device_build=False), api.Scene(s, device_build=True)


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(64, 64, with_specular=True),
    lambda: scenes.interior("bathroom", 0.15, 160, 90),
    lambda: scenes.interior("living-room", 0.1, 160, 90),
    lambda: scenes.random_soup(5000, seed=4),
    lambda: scenes.axis_aligned_grid(6),          # coincident centroids / empty bins / ties
], ids=["cornell-spec", "bathroom", "living-room", "soup", "grid"])
def test_device_build_equals_host_build(make):
    s = make()
    h, d = both(s)
    ih, idv = h.info(), d.info()
    for k in ("n_tris", "n_lights", "bvh2_nodes", "bvh2_depth", "bvh2_leaves", "bvh2_max_leaf", "wide_nodes", "wide_max_stack"):
        assert ih[k] == idv[k], k
    assert np.array_equal(h.leaf_order(), d.leaf_order())
    rays = np.concatenate([scenes.bounce_rays(s, 100000, 5), scenes.primary_rays(s, 50000, 6) if s.cam_P is not None else scenes.bounce_rays(s, 50000, 7)])
    for mode in (0, 1):
        a, b = h.traverse(rays, mode), d.traverse(rays, mode)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    segs = scenes.shadow_segments(s, 100000, 8)
    assert np.array_equal(h.visible(segs), d.visible(segs))
    h.close(); d.close()


@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_tiny_scenes(n):
    s = scenes.random_soup(n, seed=2)
    h, d = both(s)
    assert h.info()["bvh2_nodes"] == d.info()["bvh2_nodes"] and np.array_equal(h.leaf_order(), d.leaf_order())
    rays = np.random.default_rng(1).normal(size=(4096, 6)).astype(np.float32)
    assert np.array_equal(h.traverse(rays).view(np.uint8), d.traverse(rays).view(np.uint8))
    h.close(); d.close()


def test_full_size_bathroom_build_and_time():
    s = scenes.interior("bathroom", 1.0, 1920, 1080)
    t = time.perf_counter(); h = api.Scene(s, device_build=False); th = time.perf_counter() - t
    t = time.perf_counter(); d = api.Scene(s, device_build=True); td = time.perf_counter() - t
    print(f"scene create: host build {th:.2f} s, device build {td:.2f} s (n_tris {s.ntris})")
    assert h.info()["bvh2_nodes"] == d.info()["bvh2_nodes"] and h.info()["wide_nodes"] == d.info()["wide_nodes"]
    assert np.array_equal(h.leaf_order(), d.leaf_order())
    rays = scenes.bounce_rays(s, 1 << 20, 3)
    assert np.array_equal(h.traverse(rays).view(np.uint8), d.traverse(rays).view(np.uint8))
    h.close(); d.close()
