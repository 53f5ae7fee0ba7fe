"""rt_trace_device_static(): the closest hit with visit counts for callers that cannot allocate.

Its per-workgroup count records live in a static device array of the library instead of a buffer
the library allocates; everything else is rt_trace_device().  Checked here against that entry
point on the same rays: hit records bit for bit, counts exactly -- on two calls in a row, the second
with a smaller batch (the array keeps the first call's records), for the per-lane walk and the wave
sweep, and the size limit's error."""
import numpy as np
import pytest
import torch

from helpers import asset, mirror, random_rays

pytestmark = pytest.mark.gpu


def _pair(rt, sc, d_rays, n):
    want = torch.zeros(max(n, 1) * 10, dtype=torch.int32, device="cuda")
    got = torch.full((max(n, 1) * 10,), -1, dtype=torch.int32, device="cuda")
    d_counts = torch.full((3,), 123456789, dtype=torch.int64, device="cuda")   # must be overwritten
    torch.cuda.synchronize()
    counts = sc.trace_device(d_rays.data_ptr(), n, want.data_ptr(), count=True)
    sc.trace_device_static(d_rays.data_ptr(), n, got.data_ptr(), d_counts.data_ptr())
    sc.ctx.synchronize()
    assert tuple(int(v) for v in d_counts.cpu()) == tuple(counts)
    if n:
        assert torch.equal(got[:n * 10], want[:n * 10])
    return counts


def test_static_scratch_trace_equals_the_allocating_one(rt, ob, ctx):
    for osc, lo, hi in ((ob.Scene.cornell_box(asset("cornell_box.obj")), -550.0, 0.0),
                        (ob.Scene.cornell_box(asset("cornell_box_plane.obj")), -550.0, 0.0),
                        (ob.Scene.world1(), -6.0, 6.0)):
        sc = mirror(rt, ctx, osc)
        rays = np.concatenate([osc.record_rays(128, 128, 4, 6, cap=300000), random_rays(ob, 11, 100000, lo, hi)])
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 6)).cuda()
        for mode in (rt.TRAVERSAL_PER_LANE, rt.TRAVERSAL_SWEEP if sc.num_entries <= 4096 else rt.TRAVERSAL_AUTO):
            sc.set_traversal(mode)
            big = _pair(rt, sc, d_rays, len(rays))
            small = _pair(rt, sc, d_rays, 1000)                # fewer workgroups than the call before
            one = _pair(rt, sc, d_rays, 1)
            assert sum(big) > sum(small) > sum(one) > 0
            assert _pair(rt, sc, d_rays, 0) == (0, 0, 0)
        # against the oracle too: the counts are the reference recursion's visits
        sc.set_traversal(rt.TRAVERSAL_PER_LANE)
        _, st = osc.trace(rays[:50000])
        assert _pair(rt, sc, d_rays, 50000) == (st["box_tests"], st["tri_tests"], st["sph_tests"])


def test_static_scratch_trace_limit_and_errors(rt, ob, ctx):
    sc = mirror(rt, ctx, ob.Scene.world1())
    n = rt.TRACE_STATIC_MAX_RAYS
    rays = random_rays(ob, 5, 4096, -6.0, 6.0)
    d_rays = torch.from_numpy(np.tile(rays.view(np.float32).reshape(-1, 6), (n // 4096 + 1, 1))).cuda()
    assert _pair(rt, sc, d_rays, n)[2] > 0                      # the largest batch it takes
    d_hits = torch.zeros(10, dtype=torch.int32, device="cuda")
    d_counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    with pytest.raises(rt.RtError, match="RT_TRACE_STATIC_MAX_RAYS"):
        sc.trace_device_static(d_rays.data_ptr(), n + 1, d_hits.data_ptr(), d_counts.data_ptr())
    with pytest.raises(rt.RtError, match="NULL"):
        sc.trace_device_static(d_rays.data_ptr(), 16, d_hits.data_ptr(), None)
