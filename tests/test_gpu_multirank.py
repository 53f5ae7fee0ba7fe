"""The N>1 path on the GPU: bench.py under torch.distributed.run with 2 and 4 ranks sharing the
one GPU of the test box (gloo backend: RCCL refuses two ranks on one device).  Every rank renders
its partition (row bands x sample range) with the HIP path; the reduced frame must equal the
1-rank frame: same ray count exactly, same radiance up to f32 summation grouping."""
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
    if n == 1 and not extra:
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
        assert many["roofline"]["achieved"] is not None and "cpu_baseline" not in many


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
