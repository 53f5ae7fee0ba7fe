"""Partition logic and the N>1 reduce path on CPU (gloo, world_size 2 and 4)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import partition as P


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("shape", [(1024, 1024, 1024), (100, 77, 12), (64, 200, 3), (1920, 1080, 256)])
def test_partition_covers_every_pixel_sample_once(world, shape):
    w, h, spp = shape
    cover = np.zeros((h, w), dtype=np.int64)
    for r in range(world):
        (rows, groups, group), (s0, s1) = P.partition(w, h, spp, world, r)
        assert 0 <= s0 <= s1 <= spp and rows % 8 == 0 and group < groups
        for x0, y0, x1, y1 in P.band_rects(w, h, groups, group):
            assert 0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h
            assert y0 % 8 == 0                      # cuts keep the 8x8 tiles whole
            cover[y0:y1, x0:x1] += s1 - s0
    assert (cover == spp).all()


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8, 16])
def test_library_partition_is_the_python_one(world):
    """rt_partition (what rt_render_multi gives every GPU, and what a one-process-per-GPU job asks
    for) against partition.py, including a job that starts at a later sample and one with fewer
    samples than ranks."""
    import raytracer_amd as rt
    for w, h, s0, s1 in [(1024, 1024, 0, 1024), (100, 77, 5, 17), (64, 200, 0, 3), (1920, 1080, 100, 356)]:
        job = rt.Scene.params(w, h, s1, 6, 0.5, samples=(s0, s1))
        for r in range(world):
            (rows, groups, group), (a, b) = P.partition(w, h, s1 - s0, world, r)
            p = rt.partition_params(job, world, r)
            assert (p.band_rows, p.band_count, p.band_index) == (rows, groups, group)
            assert (p.sample_begin, p.sample_end) == (s0 + a, s0 + b)
            assert (p.width, p.height, p.x0, p.y0, p.x1, p.y1, p.seed) == (w, h, 0, 0, w, h, job.seed)
        assert (job.band_count, job.sample_begin, job.sample_end) == (0, s0, s1)      # the job itself is untouched
    with pytest.raises(rt.RtError):
        rt.partition_params(rt.Scene.params(8, 8, 4, 6, 0.5), 2, 2)
    with pytest.raises(rt.RtError):
        rt.partition_params(rt.Scene.params(64, 64, 4, 6, 0.5, bands=(16, 2, 0)), 2, 0)


def test_split_factors():
    assert P.split_factors(1, 1024) == (1, 1)
    assert P.split_factors(2, 1024) == (1, 2)
    assert P.split_factors(4, 1024) == (2, 2)
    assert P.split_factors(8, 1024) == (2, 4)
    assert P.split_factors(8, 2) == (8, 1)
    assert [P.sample_range(10, 4, k) for k in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        P.partition(8, 8, 1, 2, 2)


def _fake_render(width, height):
    """Stands in for the GPU call: a deterministic value per (pixel, sample, channel)."""
    def render_part(bands, samples, accum):
        rows, groups, group = bands
        img = accum.view(height, width, 3)
        for x0, y0, x1, y1 in P.band_rects(width, height, groups, group):
            ys, xs = np.mgrid[y0:y1, x0:x1]
            for s in range(*samples):
                v = ((xs * 73 + ys * 151 + s * 31) % 97).astype(np.float32) / 7.0
                for c in range(3):
                    img[y0:y1, x0:x1, c] += torch.from_numpy(v * (c + 1))
    return render_part


def _worker(rank, world, port, w, h, spp, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from raytracer_amd import distributed as D
    r, ws, _ = D.init_from_env("gloo")
    assert (r, ws) == (rank, world)
    accum = torch.zeros(h * w * 3, dtype=torch.float32)
    D.render_partitioned(_fake_render(w, h), w, h, spp, rank, world, accum)
    again = torch.zeros(h * w * 3, dtype=torch.float32)
    D.render_partitioned(_fake_render(w, h), w, h, spp, rank, world, again)
    if rank == 0:
        assert torch.equal(accum, again)            # the reduce is deterministic run to run
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_render_and_reduce_gloo(tmp_path, world):
    w, h, spp = 40, 136, 6
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    want = torch.zeros(h * w * 3, dtype=torch.float32)
    _fake_render(w, h)((P.BAND_ROWS, 1, 0), (0, spp), want)
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-6)
