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
    # limits: a clear error, nothing launched
    with pytest.raises(api.RtError, match="2\\^24"):
        sc.traverse_noalloc(0x1000, (1 << 24) + 1, 0x1000)
    sc.traverse_noalloc(None, 0, None)               # n == 0 is a no-op
    sc.close()
