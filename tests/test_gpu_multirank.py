"""The N>1 paths on the GPU.

One process per GPU: bench.py under torch.distributed.run with 2 and 4 ranks sharing the one GPU of
the test box (gloo backend: RCCL refuses two ranks on one device).  Every rank renders its
partition (row bands x sample range) with the HIP path; the reduced frame must equal the 1-rank
frame: same ray count exactly, same radiance up to f32 summation grouping.

One process for all GPUs: rt_context_create_multi + rt_render_multi.  With one GPU the call goes
through RCCL (a communicator of one rank) and must give rt_render_device's frame bit for bit; with
2, 3, 4 and 8 workers sharing the box's GPU (the library then adds the frames itself, RCCL being
unable to) the partition, the per-GPU scene copies and the host threads are the real ones."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _run(n, backend="gloo", extra=()):
    args = ["--steps", "1", "--warmup", "0", "--workload", "thing_glass_2048_512", "--size", "160x136",
            "--spp", "12", "--no-cpu", "--gpus", str(n)] + list(extra)
    env = dict(os.environ, RT_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if (n == 1 and not extra) or "--share-gpus" in extra:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
               "--master-addr", "127.0.0.1", "--master-port", str(29600 + n), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_two_and_four_ranks_match_one_rank():
    one = _run(1)
    assert one["n_gpus"] == 1 and one["config"]["paths_per_step"] == 160 * 136 * 12
    for n in (2, 4):
        many = _run(n)
        assert many["n_gpus"] == n and many["scaling"] == "strong"
        assert many["config"]["rays_per_step"] == one["config"]["rays_per_step"]
        assert many["config"]["accum_sum"] == pytest.approx(one["config"]["accum_sum"], rel=1e-5)
        assert many["roofline"]["algorithmic"]["GB/s"] is not None and "cpu_baseline" not in many
        assert "processes x 1 GPU" in many["config"]["parallelism"]


def test_one_process_drives_every_gpu():
    """bench.py --gpus N started plainly (no torchrun): one multi-GPU context, rt_render_multi."""
    one = _run(1)
    assert "1 process x 1 GPU" in one["config"]["parallelism"]
    for n in (2, 4):
        many = _run(n, extra=["--share-gpus"])
        assert many["n_gpus"] == n and f"1 process x {n} GPU" in many["config"]["parallelism"]
        assert many["config"]["rays_per_step"] == one["config"]["rays_per_step"]
        assert many["config"]["paths_per_step"] == one["config"]["paths_per_step"]
        assert many["config"]["accum_sum"] == pytest.approx(one["config"]["accum_sum"], rel=1e-5)


def test_wide_walk_on_every_gpu_of_a_context():
    """The opt-in wide walk through the in-process multi-GPU path: the tree is collapsed once and every
    worker's copy of the scene gets the same arrays; same rays as one GPU, and both forms of starting N ranks agree."""
    one = _run(1, extra=["--traversal", "wide"])
    assert one["config"]["kernel_form"] == "wide_refill"
    plain = _run(1)
    assert one["config"]["rays_per_step"] == plain["config"]["rays_per_step"]     # the same hits on every ray, ties included
    # (the last one: every worker builds the tree itself on its GPU, from the primitives alone)
    for extra in (["--share-gpus", "--traversal", "wide"], ["--traversal", "wide"],
                  ["--share-gpus", "--traversal", "wide", "--wide-build", "device"]):
        many = _run(3, extra=extra)
        assert many["n_gpus"] == 3 and many["config"]["kernel_form"] == "wide_refill"
        assert many["config"]["rays_per_step"] == one["config"]["rays_per_step"]    # (another tree when built on the GPU: the same hits)
        assert many["config"]["accum_sum"] == pytest.approx(one["config"]["accum_sum"], rel=1e-5)


def test_render_multi_against_render_device():
    import numpy as np
    import torch
    import raytracer_amd as rt
    w, h, spp, depth = 200, 136, 12, 6
    single = rt.Context(0)
    world1 = rt.World.world2(single, variant=1)
    cam = world1.camera(w, h)
    job = world1.params(w, h, spp, depth)
    want = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st1 = world1.scene.render_device(cam, job, want.data_ptr())
    # one GPU through the multi-GPU entry point: RCCL with one rank, the same frame bit for bit
    ctx = rt.Context(devices=[0])
    assert ctx.num_devices == 1
    wm = rt.World.world2(ctx, variant=1)
    got = torch.zeros_like(want)
    torch.cuda.synchronize()
    st = wm.scene.render_multi_device(cam, job, got.data_ptr())
    assert st["n_devices"] == 1 and st["rays_traced"] == st1["rays_traced"] and st["paths"] == w * h * spp
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert st["rccl_ranks"] == 1 and st["reduce_mode"] == rt.REDUCE_RCCL      # the reduce really went through RCCL
    assert st["device_ms_render"][0] > 0 and st["slowest_device"] == 0 and st["ms_render_min"] == st["ms_render"]
    # a failed call leaves the caller's buffer alone, so it can be repeated (every GPU renders into a frame of
    # its own; the sum is added at the end): here the second call simply adds the frame once more
    st = wm.scene.render_multi_device(cam, job, got.data_ptr())
    assert torch.equal(got.view(torch.int32), (want + want).view(torch.int32))
    got.zero_()
    st = wm.scene.render_multi_device(cam, job, got.data_ptr())
    host, sth = wm.scene.render_multi(cam, job)
    assert np.array_equal(host.reshape(-1).view(np.uint32), want.cpu().numpy().view(np.uint32))
    with pytest.raises(rt.RtError, match="rt_context_create_multi"):
        world1.scene.render_multi_device(cam, job, got.data_ptr())
    with pytest.raises(rt.RtError, match="band_count"):
        wm.scene.render_multi_device(cam, world1.params(w, h, spp, depth, bands=(64, 2, 0)), got.data_ptr())
    wm.close()
    ctx.close()
    # several workers on the one GPU: every worker renders rt_partition(n, r) on its own scene copy
    for n in (2, 3, 4, 8):
        ctx = rt.Context(devices=[0] * n)
        assert ctx.num_devices == n
        wm = rt.World.world2(ctx, variant=1)
        got.zero_()
        torch.cuda.synchronize()
        st = wm.scene.render_multi_device(cam, job, got.data_ptr())
        assert st["n_devices"] == n and st["rays_traced"] == st1["rays_traced"] and st["paths"] == w * h * spp
        torch.testing.assert_close(got, want, rtol=2e-5, atol=1e-6)
        again = torch.zeros_like(want)
        torch.cuda.synchronize()
        wm.scene.render_multi_device(cam, job, again.data_ptr())
        assert torch.equal(again.view(torch.int32), got.view(torch.int32))           # deterministic for fixed n
        # the scene's other traversal forms follow to every copy
        wm.scene.set_traversal(rt.TRAVERSAL_SWEEP if wm.scene.num_entries <= 4096 else rt.TRAVERSAL_PER_LANE)
        again.zero_()
        torch.cuda.synchronize()
        wm.scene.render_multi_device(cam, job, again.data_ptr())
        assert torch.equal(again.view(torch.int32), got.view(torch.int32))
        assert st["rccl_ranks"] == 0 and st["reduce_mode"] == rt.REDUCE_ORDERED     # one GPU named n times: added in rank order
        assert all(ms > 0 for ms in st["device_ms_render"][:min(n, 16)]) and st["slowest_device"] < n
        assert st["ms_render_min"] <= st["ms_render"]
        # the strict-order reduce (every peer's frame copied to the root, added there rank by rank): the code
        # path distinct GPUs take when it is selected, here with copies inside the one GPU -- the same bits
        ctx.set_reduce(rt.REDUCE_ORDERED)
        again.zero_()
        torch.cuda.synchronize()
        st_o = wm.scene.render_multi_device(cam, job, again.data_ptr())
        assert st_o["reduce_mode"] == rt.REDUCE_ORDERED and st_o["rccl_ranks"] == 0
        assert torch.equal(again.view(torch.int32), got.view(torch.int32))
        wm.close()
        ctx.close()
    with pytest.raises(rt.RtError, match="rt_context_create_multi"):
        single.set_reduce(rt.REDUCE_ORDERED)
    world1.close()
    single.close()


def test_headless_main_on_two_workers(tmp_path):
    """examples/rt_render_main.cpp --devices 0,0: the C++ caller through rt_context_create_multi and
    rt_render_multi (via the host layer's render()); same PPM as one GPU up to 8-bit rounding."""
    import numpy as np
    from helpers import asset
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    frames = []
    for extra in ([], ["--devices", "0,0"], ["--gpus", "1"]):
        out = tmp_path / f"f{len(frames)}.ppm"
        p = subprocess.run([exe, "--scene", "world2_glass", "--obj", asset("thing.obj"), "--size", "96x80", "--spp", "24",
                            "--out", str(out)] + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stdout + p.stderr
        assert ("on 2 GPU(s)" in p.stdout) == (extra == ["--devices", "0,0"])
        frames.append(np.frombuffer(out.read_bytes()[len(b"P6\n96 80\n255\n"):], dtype=np.uint8).astype(int))
    assert np.abs(frames[0] - frames[1]).max() <= 1 and np.array_equal(frames[0], frames[2])


def test_one_rank_through_rccl():
    """The barrier, the reduce and the all-reduces of the N>1 path on the nccl (= RCCL) backend,
    with the single rank this box can hold, launched exactly as a node run is launched."""
    one = _run(1)
    rccl = _run(1, backend="nccl", extra=["--force-dist"])
    assert rccl["n_gpus"] == 1
    assert rccl["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    assert rccl["config"]["accum_sum"] == one["config"]["accum_sum"]


def test_render_module_two_ranks_write_the_one_rank_frame(tmp_path):
    """python -m raytracer_amd.render: one rank, and two ranks (gloo, sharing the box's GPU) that split
    the frame by row bands and sample range and reduce to rank 0.  The PPMs must agree up to the
    last bit of 8-bit rounding (the two halves of a pixel's samples are summed separately)."""
    import numpy as np
    env = dict(os.environ, RT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    args = ["--scene", "world2_glass", "--size", "96x80", "--spp", "24"]
    frames = []
    for n in (1, 2):
        out = tmp_path / f"f{n}.ppm"
        if n == 1:
            cmd = [sys.executable, "-m", "raytracer_amd.render"]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                   "--master-addr", "127.0.0.1", "--master-port", "29655", "-m", "raytracer_amd.render"]
        p = subprocess.run(cmd + args + ["--out", str(out)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert "Finished in" in p.stdout
        raw = out.read_bytes()
        header = b"P6\n96 80\n255\n"
        assert raw.startswith(header)
        frames.append(np.frombuffer(raw[len(header):], dtype=np.uint8).astype(int))
    assert len(frames[0]) == 96 * 80 * 3 and frames[0].max() > 100
    assert np.abs(frames[0] - frames[1]).max() <= 1
    # one rank through the wide walk, its tree built on the GPU: the same picture
    out = tmp_path / "wide.ppm"
    p = subprocess.run([sys.executable, "-m", "raytracer_amd.render"] + args + ["--wide", "device", "--out", str(out)],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    wide = np.frombuffer(out.read_bytes()[len(header):], dtype=np.uint8).astype(int)
    assert np.array_equal(wide, frames[0])
