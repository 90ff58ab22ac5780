"""GPU: rt_intersect_closest_noalloc (static device scratch, no hipMalloc in the call) against rt_intersect_closest_dev,
through the C ABI, on device buffers: bit-identical hits on two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd.abi import HIT_DTYPE

pytestmark = pytest.mark.gpu


def _dev_hits(torch, sc, fn, rays, stream=None):
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.full((len(rays) * 5,), 0x7F, dtype=torch.int32, device="cuda")   # poison: every entry must be written
    torch.cuda.synchronize()
    fn(d_rays.data_ptr(), len(rays), d_out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(HIT_DTYPE).copy()


def test_noalloc_entry_point_matches_the_allocating_one():
    import torch
    s = scenes.interior("bathroom", 0.3, 64, 36)     # deep tree: the stacks overflow LDS into the scratch
    sc = api.Scene(s)
    info = sc.info()
    assert info["wide_max_stack"] > 12, info     # more entries than a lane keeps in LDS (RT_STACK_LDS)
    big = np.concatenate([scenes.primary_rays(s, 300_000, 1), scenes.bounce_rays(s, 300_000, 2)])
    small = scenes.bounce_rays(s, 70_001, 5)
    for rays in (big, small):                        # two calls in a row, the second with a smaller input
        want = _dev_hits(torch, sc, sc.traverse_dev, rays)
        got = _dev_hits(torch, sc, sc.traverse_noalloc, rays)
        assert (want["t"] < 3e38).mean() > 0.9
        for k in ("ID", "t", "alpha", "beta", "gamma"):
            assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    # on a caller's stream too, and a tiny batch
    st = torch.cuda.Stream()
    tiny = big[:7]
    got = _dev_hits(torch, sc, sc.traverse_noalloc, tiny, st.cuda_stream)
    want = _dev_hits(torch, sc, sc.traverse_dev, tiny)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    # calls that arrive on different streams share the GPU's one scratch array: the entry point chains them on the device
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d_big, d_small = torch.from_numpy(big).cuda(), torch.from_numpy(small).cuda()
    o_big = torch.full((len(big) * 5,), 0x7F, dtype=torch.int32, device="cuda"); o_small = torch.full((len(small) * 5,), 0x7F, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        sc.traverse_noalloc(d_big.data_ptr(), len(big), o_big.data_ptr(), 0, s1.cuda_stream)
        sc.traverse_noalloc(d_small.data_ptr(), len(small), o_small.data_ptr(), 0, s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(o_big.cpu().numpy().view(HIT_DTYPE).view(np.uint8), _dev_hits(torch, sc, sc.traverse_dev, big).view(np.uint8))
    assert np.array_equal(o_small.cpu().numpy().view(HIT_DTYPE).view(np.uint8), _dev_hits(torch, sc, sc.traverse_dev, small).view(np.uint8))
    # its rays are counted like rt_intersect_closest_dev's when counting is on
    sc.counters_enable(True); sc.counters()
    _dev_hits(torch, sc, sc.traverse_noalloc, small)
    assert sc.counters()["closest_rays"] == len(small)
    sc.counters_enable(False)
    # under HIP graph capture (what the entry point is for): captured once, replayed twice with new inputs in the same buffers
    g = torch.cuda.CUDAGraph()
    d_in = torch.from_numpy(small).cuda(); d_out = torch.zeros(len(small) * 5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        sc.traverse_noalloc(d_in.data_ptr(), len(small), d_out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    for rays in (small, scenes.bounce_rays(s, len(small), 9)):
        d_in.copy_(torch.from_numpy(rays).cuda()); d_out.zero_(); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(HIT_DTYPE).view(np.uint8), _dev_hits(torch, sc, sc.traverse_dev, rays).view(np.uint8))
    # limits: a clear error, nothing launched
    with pytest.raises(api.RtError, match="2\\^24"):
        sc.traverse_noalloc(0x1000, (1 << 24) + 1, 0x1000)
    sc.traverse_noalloc(None, 0, None)               # n == 0 is a no-op
    sc.close()
