"""The benchmark workloads at their full sizes (1024^2 x 1024 spp Cornell box, 1080p x 256 spp sphere
scene, 2048^2 x 512 spp glass mesh, 512^2 x 64 spp full Cornell box), through properties that do
not need a CPU render of that size:

* the same frame twice is the same frame bit for bit, with the same ray count;
* the frame cut into eight ranks' shares (interleaved row bands x sample ranges, the partition the
  multi-GPU job uses) and summed has exactly the frame's paths and rays and the frame's radiance up
  to f32 summation grouping;
* the frame's mean radiance and rays per path agree with the oracle's render of the same scene at
  the same resolution and a few samples per pixel (both are unbiased estimates of the same image;
  the oracle's mean over >= 1 M pixels x 4 samples carries well under 1 % of noise)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT, asset

sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from raytracer_amd.partition import partition   # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle_scene(ob, kind):
    if kind == "cornell_plane":
        return ob.Scene.cornell_box(asset("cornell_box_plane.obj"))
    if kind == "cornell":
        return ob.Scene.cornell_box(asset("cornell_box.obj"))
    if kind == "world1":
        return ob.Scene.world1()
    return ob.Scene.world2(asset("thing.obj"), 1)


@pytest.mark.parametrize("name", ["cornell_plane_1024_1k", "world1_1080p_256", "thing_glass_2048_512",
                                  "cornell_full_512_64"])
def test_full_size_frame(rt, ob, ctx, name):
    kind, w, h, spp, depth = bench.WORKLOADS[name]
    world = bench.make_world(ctx, kind)
    cam = world.camera(w, h)
    accum = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")

    def render(**kw):
        accum.zero_()
        torch.cuda.synchronize()
        st = world.scene.render_device(cam, world.params(w, h, spp, depth, **kw), accum.data_ptr())
        return accum.clone(), st

    frame, st = render()
    again, st2 = render()
    assert st["paths"] == w * h * spp and st["rays_traced"] == st2["rays_traced"]
    assert torch.equal(frame.view(torch.int32), again.view(torch.int32))
    assert torch.isfinite(frame).all()

    # eight ranks' shares into one buffer (what the RCCL reduce adds up)
    accum.zero_()
    torch.cuda.synchronize()
    rays = paths = 0
    for rank in range(8):
        bands, samples = partition(w, h, spp, 8, rank)
        s = world.scene.render_device(cam, world.params(w, h, spp, depth, samples=samples, bands=bands), accum.data_ptr())
        rays += s["rays_traced"]
        paths += s["paths"]
    assert paths == st["paths"] and rays == st["rays_traced"]
    torch.testing.assert_close(accum, frame, rtol=1e-4, atol=1e-4 * spp)

    # against the oracle at a few samples per pixel
    o_spp = 4
    oimg, _, ost = _oracle_scene(ob, kind).render(w, h, o_spp, depth, nthreads=16, want_image=False)
    mean_gpu = float(frame.double().mean()) / spp
    mean_cpu = float(np.asarray(oimg, dtype=np.float64).mean()) / o_spp
    print(f"{name}: mean radiance gpu {mean_gpu:.5f} oracle {mean_cpu:.5f}; rays per path "
          f"gpu {st['rays_traced'] / st['paths']:.4f} oracle {ost['rays'] / ost['paths']:.4f}")
    assert abs(mean_gpu - mean_cpu) <= 0.02 * mean_cpu
    assert abs(st["rays_traced"] / st["paths"] - ost["rays"] / ost["paths"]) < 0.02
    world.close()
