#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the wavefront path tracer on the Cornell box
(cornell_box_plane.obj, 1024x1024, 1024 spp, max depth 6), whole job over N GPUs.

  python bench.py --gpus N --steps K --warmup W
      one process: a multi-GPU context over GPUs 0 .. N-1 (rt_context_create_multi), the scene built
      once and uploaded to every GPU, rt_render_multi_device() per step (every GPU renders its share
      on a host thread of its own, one RCCL sum-reduce to GPU 0).  N = 1 takes the same path.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
      one process per GPU: every rank renders rt_partition(N, rank) with rt_render_device(), then
      torch.distributed.reduce (RCCL) combines the buffers on rank 0.

A step is one full render of the frame.  Scene build, BVH upload and OBJ parsing are outside the
timed region.  1 ray = 1 closest-hit query.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracer_amd as rt                                     # noqa: E402
from raytracer_amd import distributed as D                     # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, GB/s
N_SIMDS = 1024                 # 256 CUs x 4
N_CUS = 256
VMEM_CYCLES = 16.0             # CU cycles per wave-wide 16-byte vector memory instruction, at best (tools/ta_rate.hip)
WORKLOADS = {
    # name: (constructor, width, height, spp, max depth)
    "cornell_plane_1024_1k": ("cornell_plane", 1024, 1024, 1024, 6),
    "world1_1080p_256": ("world1", 1920, 1080, 256, 6),
    "thing_glass_2048_512": ("world2_glass", 2048, 2048, 512, 6),
    "cornell_full_512_64": ("cornell", 512, 512, 64, 8),
    "soup_10m_4k_256": ("soup", 3840, 2160, 256, 16),
}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_workloads.json")
KERNEL_SOURCES = ("rt_device.hip", "rt_device_walk.hip", "rt_internal.hpp", "rt_api.cpp", "rt_build.cpp", "rt_wide.cpp", "rt_wide_device.hip", "Makefile")


def make_world(ctx, kind):
    if kind == "cornell_plane":
        return rt.World.cornell_box(ctx, os.path.join(rt.ASSET_DIR, "cornell_box_plane.obj"))
    if kind == "cornell":
        return rt.World.cornell_box(ctx, os.path.join(rt.ASSET_DIR, "cornell_box.obj"))
    if kind == "world1":
        return rt.World.world1(ctx)
    if kind == "world2_glass":
        return rt.World.world2(ctx, variant=1)
    if kind == "soup":
        return rt.World.soup(ctx, int(os.environ.get("RT_SOUP_TRIS", "10000000")), grid=64, seed=12345)
    raise ValueError(kind)


# stream bytes of the other two stages of the per-batch kernel: shade reads the queue index, the ray
# record, the hit and the primitive's shading data (4 + 48 + 8 + 20 B) and writes the next ray
# (48 B); raygen writes a ray record (48 B)
SHADE_BYTES_PER_RAY = 4.0 + 48.0 + 8.0 + 20.0 + 48.0
RAYGEN_BYTES_PER_PATH = 48.0


def trace_bytes_per_ray(n_box, n_tri, n_sph, box_bytes=32.0):
    """Algorithmic bytes the closest-hit kernel moves per ray: 32 B per box test (24 B box + 8 B
    links), 36 B per triangle test (three positions), 16 B per sphere test (centre + radius),
    plus the ray record read (origin, direction, path slot, rng key: 32 B), the hit record written
    (8 B) and the queue index written (4 B).  The opt-in wide walk counts node visits where the
    others count box tests: a node of eight 8-bit boxes is 80 B."""
    return box_bytes * n_box + 36.0 * n_tri + 16.0 * n_sph + 32.0 + 8.0 + 4.0


def kernel_source_hash():
    """Content hash of the sources the kernels are built from (what a PMC record was measured on)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_record(workload, key):
    """Per-ray hardware counts of this workload's per-batch kernel from the committed rocprofv3 PMC
    passes (rocprofv3 cannot run inside this process): HBM bytes per ray ((2*FETCH_SIZE + WRITE_SIZE)
    KiB, the gfx950 correction for 16-byte-per-lane streams), vector instructions per ray.  They are
    properties of the kernel and the workload, so multiplied by the rate (and divided by the clock)
    measured in THIS run they give this run's HBM and issue rates.  Records are keyed
    <workload>/<tree>/<kernel form>.  None when there is no record."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)["records"].get(f"{workload}/{key}")
    except (OSError, KeyError, ValueError):
        return None
    if rec is not None:
        rec = dict(rec, stale=rec.get("source_hash") != kernel_source_hash())
    return rec


def roofline(workload, key, form, rays_per_s, n_gpus, algo_gbs, bytes_per_ray, tests, extra, clock_ghz=None):
    """The bound of the dominant kernel as a fraction <= 1.  Three candidates, all from measured
    counts: HBM (PMC bytes per ray x measured rays/s against the 8 TB/s peak), vector-instruction
    issue (PMC instructions per ray x measured rays/s, priced at the SIMD cycles per instruction this
    part sustains with 8 waves per SIMD, tools/valu_rate.hip: 2.1 plain, 3.07 packed f32) and the
    CUs' address units (PMC busy cycles per ray x measured rays/s: the share of time they are busy).
    The bound is the largest fraction.  The algorithmic rate of the scope table's per-ray formula is reported
    beside it: it exceeds the HBM peak when the flattened tree is served from the caches."""
    out = {"kernel": extra.pop("kernel"), "algorithmic": {"bytes_per_ray": round(bytes_per_ray, 1), "GB/s": round(algo_gbs, 1) if algo_gbs else None,
                                                         "of_hbm_peak": round(algo_gbs / (HBM_PEAK_GBS * n_gpus), 4) if algo_gbs else None,
                                                         "tests_per_ray": tests}}
    rec = pmc_record(workload, key)
    if rec is None:
        out.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS * n_gpus, "unit": "GB/s", "frac": None, "traffic": None,
                    "note": f"no PMC record for {workload}/{key} in profiles/pmc_workloads.json: only the algorithmic rate is known"})
        out.update(extra)
        return out
    per_gpu = rays_per_s / n_gpus
    hbm_gbs = rec["hbm_bytes_per_ray"] * per_gpu / 1e9
    hbm_frac = hbm_gbs / HBM_PEAK_GBS
    cyc = rec["simd_cycles_per_valu_inst"]
    # the clock of THIS run (one wave of every launch reads the cycle counter and the 100 MHz counter at its
    # start and end: rt_stats.shader_clock_ghz); the profiled run's lower clock only where that is missing
    run_clock = clock_ghz if clock_ghz else rec["clock_ghz"]
    issue_rate = rec["valu_insts_per_ray"] * per_gpu / (N_SIMDS * run_clock * 1e9)    # per SIMD cycle
    issue_peak = 1.0 / cyc
    issue_frac = issue_rate / issue_peak
    out["hbm"] = {"achieved_GB/s": round(hbm_gbs, 1), "peak_GB/s": HBM_PEAK_GBS, "frac": round(hbm_frac, 4),
                  "bytes_per_ray": round(rec["hbm_bytes_per_ray"], 1)}
    out["valu_issue"] = {"achieved_inst_per_simd_cycle": round(issue_rate, 4), "peak_inst_per_simd_cycle": round(issue_peak, 4),
                         "frac": round(issue_frac, 4), "insts_per_ray": round(rec["valu_insts_per_ray"], 1),
                         "active_lanes_per_inst": rec.get("active_lanes"), "clock_ghz": round(run_clock, 3),
                         "clock_source": "measured in this run (rt_stats.shader_clock_ghz)" if clock_ghz else "the profiled run's",
                         "clock_ghz_profiled_run": rec["clock_ghz"]}
    # third candidate: the CUs' address units (TA), through which every vector memory instruction goes.
    # TA_TA_BUSY over the profiled run gives busy cycles per ray; at this run's rate that is the share
    # of time the units are busy.  (Priced instead at the 16 CU cycles a wave-wide access takes at
    # best, tools/ta_rate.hip, the instruction count gives the lower figure reported beside it.)
    vmem_frac = None
    if rec.get("ta_busy") is not None and rec.get("mrays_profiled"):
        vmem_frac = rec["ta_busy"] * per_gpu / (rec["mrays_profiled"] * 1e6)
        vmem_rate = rec["vmem_insts_per_ray"] * per_gpu / (N_CUS * run_clock * 1e9)      # per CU cycle
        out["address_unit"] = {"busy_share": round(vmem_frac, 4), "busy_share_profiled": rec["ta_busy"],
                               "vmem_insts_per_ray": round(rec["vmem_insts_per_ray"], 2),
                               "vmem_insts_per_cu_cycle": round(vmem_rate, 4),
                               "frac_at_16_cycles_per_inst": round(vmem_rate * VMEM_CYCLES, 4)}
    best = max(hbm_frac, issue_frac, vmem_frac or 0.0)
    if best == hbm_frac:
        out.update({"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(min(hbm_frac, 1.0), 4)})
    elif best == issue_frac:
        out.update({"bound": "valu_issue", "achieved": round(issue_rate, 4), "peak": round(issue_peak, 4),
                    "unit": "vector instructions per SIMD cycle", "frac": round(min(issue_frac, 1.0), 4)})
    else:
        out.update({"bound": "address_unit", "achieved": round(vmem_frac, 4), "peak": 1.0,
                    "unit": "share of time the CUs' address units are busy", "frac": round(min(vmem_frac, 1.0), 4)})
    out["traffic"] = int(rec["hbm_bytes_per_ray"] * extra.get("rays_per_launch", 0)) or None
    out["traffic_unit"] = "HBM bytes per launch of the dominant kernel: PMC bytes per ray x this run's rays per launch"
    out["pmc"] = {"file": "profiles/pmc_workloads.json", "profiled_form": rec.get("form"), "this_run_form": form,
                  "git_sha": rec.get("git_sha"), "source_hash": rec.get("source_hash"), "stale": rec["stale"],
                  "per_gpu": n_gpus > 1}
    if rec.get("stages"):                      # the staged launches' counters: which stage the instructions belong to
        out["stages"] = rec["stages"]
    if rec["stale"] or (rec.get("form") and rec.get("form") != form):
        out["note"] = "the PMC record was taken on other kernel sources or another kernel form than this run's: re-run tools/profile_workloads.sh"
    out.update(extra)
    return out


def host_threads():
    """Threads for the CPU leg: every hardware thread this process may run on (its affinity mask),
    capped by OMP_NUM_THREADS where the machine's operator sets one (the GPU boxes grant a job a
    16-thread share of a much larger host and export it that way)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(n, omp) if omp > 0 else n)


def cpu_baseline(kind, width, height, depth, budget_s, gpu_render):
    """The oracle's tile-per-task renderer on the host cores, on a bounded sample of the workload
    (same scene and resolution, fewer samples per pixel)."""
    from oracle import binding as ob
    if kind == "cornell_plane":
        osc = ob.Scene.cornell_box(os.path.join(rt.ASSET_DIR, "cornell_box_plane.obj"))
    elif kind == "cornell":
        osc = ob.Scene.cornell_box(os.path.join(rt.ASSET_DIR, "cornell_box.obj"))
    elif kind == "world1":
        osc = ob.Scene.world1()
    elif kind == "world2_glass":
        osc = ob.Scene.world2(os.path.join(rt.ASSET_DIR, "thing.obj"), 1)
    else:
        return None                            # (the soup has no oracle constructor; its tests mirror the description)
    threads = host_threads()
    # calibrate on 1 spp, then pick the spp that fills the budget
    _, _, st = osc.render(width, height, 1, depth, nthreads=threads, want_image=False)
    spp = int(max(1, min(64, budget_s / max(st["seconds"], 1e-3))))
    acc, _, st = osc.render(width, height, spp, depth, nthreads=threads, want_image=False)
    # image parity at matched spp: RMSE of the per-pixel mean radiance, GPU vs this oracle image,
    # next to the oracle's own two-seed RMSE (the Monte-Carlo noise floor at this spp)
    # (this second render runs on 10 threads, the pool size the reference hard-codes at main.rs:23,
    # and gives the baseline's second data point)
    acc2, _, st10 = osc.render(width, height, spp, depth, seed_offset=7919, nthreads=10, want_image=False)
    gpu = gpu_render(spp)
    clip = 4.0
    rm = lambda a, b: float(np.sqrt(np.mean((np.minimum(a, clip * spp) / spp - np.minimum(b, clip * spp) / spp) ** 2)))
    tm = lambda a: np.minimum(np.sqrt(np.maximum(a / spp, 0.0)), 1.0)          # main.rs:369-373 before the 8-bit pack
    rt_ = lambda a, b: float(np.sqrt(np.mean((tm(a) - tm(b)) ** 2)))
    image = {"spp": spp, "rmse_gpu_vs_oracle": round(rm(gpu, acc), 5), "rmse_oracle_vs_oracle": round(rm(acc2, acc), 5),
             "tonemapped_rmse_gpu_vs_oracle": round(rt_(gpu, acc), 5),
             "tonemapped_rmse_oracle_vs_oracle": round(rt_(acc2, acc), 5),
             "clip": clip, "note": "linear per-pixel mean radiance (values clipped at 4 to tame fireflies) and "
                                   "tonemapped [0,1] values; bound used by the tests: gpu <= 1.10 x oracle-vs-oracle"}
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    return {
        "value": round(st["rays"] / st["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": threads,
        "host": {"cpus_visible": os.cpu_count(), "cpus_in_affinity_mask": affinity,
                 "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")},
        "value_10_threads": round(st10["rays"] / st10["seconds"] / 1e6, 3),
        "kind": "port",
        "sample": f"{kind} {width}x{height}, {spp} spp of the workload's samples, depth {depth}, "
                  f"64x64 tiles, {st['rays']} rays in {st['seconds']:.1f} s",
        "box_tests_per_ray": round(st["box_tests"] / st["rays"], 2),
        "tri_tests_per_ray": round(st["tri_tests"] / st["rays"], 2),
        "image_parity": image,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_plane_1024_1k", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debugging)")
    ap.add_argument("--size", default="", help="override WxH (debugging)")
    ap.add_argument("--emulate", default="", help="r/N: render rank r's share of an N-rank job on this one GPU (debugging)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="without torchrun: let the --gpus workers share the GPUs that exist (worker i on GPU i %% count; tests)")
    ap.add_argument("--backend", default=os.environ.get("RT_DIST_BACKEND", ""),
                    help="torch.distributed backend under torchrun (default nccl = RCCL; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--build", choices=["auto", "reference", "sah"], default="auto",
                    help="mesh trees: both (the first render keeps the faster; the library's default), the reference's "
                         "median split alone (its visiting order), or the surface-area split alone")
    ap.add_argument("--no-reference-leg", action="store_true",
                    help="skip the second timed run on the reference's trees in the reference's visiting order")
    ap.add_argument("--staged", action="store_true",
                    help="one launch per stage and depth (times the closest-hit kernel alone) instead of one kernel per batch")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even for one rank")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--traversal", choices=["auto", "per_lane", "sweep", "ordered", "octant", "wide"], default="auto")
    ap.add_argument("--wide-build", choices=["collapse", "device"], default="collapse",
                    help="--traversal wide: the tree collapsed on the host from the binary trees, or built on the GPU from the primitives")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: libraries that write to file descriptor 1 on
    # their own (RCCL prints a version banner when its first communicator is made) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # Two ways to N GPUs.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank of
    # N and drives one GPU.  Started plainly it drives all N GPUs itself through one context.
    per_process = "WORLD_SIZE" in os.environ or args.force_dist
    if per_process:
        rank, world, local = D.init_from_env(args.backend or None, force=args.force_dist)
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if local >= torch.cuda.device_count():  # several ranks on one GPU (tests only)
            local = local % max(torch.cuda.device_count(), 1)
        devices = None
    else:
        rank, world, local = 0, args.gpus, 0
        have = torch.cuda.device_count()
        if args.gpus < 1 or (args.gpus > have and not args.share_gpus):
            raise SystemExit(f"--gpus {args.gpus} but {have} GPU(s) visible")
        devices = [i % max(have, 1) for i in range(args.gpus)]
        local = devices[0]
    grouped = per_process and (world > 1 or args.force_dist)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kind, width, height, spp, depth = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    if args.size:
        width, height = (int(v) for v in args.size.lower().split("x"))

    ctx = rt.Context(local) if per_process else rt.Context(devices=devices)
    accum = torch.zeros(height * width * 3, dtype=torch.float32, device=dev)
    modes = {"auto": rt.TRAVERSAL_AUTO, "per_lane": rt.TRAVERSAL_PER_LANE, "sweep": rt.TRAVERSAL_SWEEP,
             "ordered": rt.TRAVERSAL_ORDERED, "octant": rt.TRAVERSAL_OCTANT, "wide": rt.TRAVERSAL_WIDE}

    def build_world(build):
        rt.set_default_build({"auto": rt.BUILD_AUTO, "reference": rt.BUILD_REFERENCE, "sah": rt.BUILD_SAH}[build])
        t0 = time.time()
        if grouped and world > 1 and kind == "soup":
            # one process per GPU: rank 0 flattens the scene and leaves it in the binary scene cache, the
            # other ranks load it instead of building the same trees again on the same host cores
            import tempfile
            cache = os.environ.get("RT_SCENE_CACHE") or os.path.join(tempfile.gettempdir(), "rt_scene_cache")
            os.makedirs(cache, exist_ok=True)
            rt.set_default_cache_dir(cache)
            if rank != 0:
                dist.barrier()
            w = make_world(ctx, kind)
            if rank == 0:
                dist.barrier()
            rt.set_default_cache_dir(None)
        else:
            w = make_world(ctx, kind)
        return w, time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(wld, traversal, steps, warmup, count_visits=True):
        """W warm-up steps, then exactly K timed steps between barriers; the max over ranks of the elapsed
        time, the sums of rays and paths, and what rank 0's (or the one process's) library calls reported."""
        wld.scene.set_traversal(modes[traversal])
        cam = wld.camera(width, height)

        def job(flags, samples=None, n_spp=None):
            return wld.params(width, height, n_spp or spp, depth, samples=samples, flags=flags, max_paths=args.max_paths)

        def share(p):
            """this process's part of job p"""
            if args.emulate:
                er, en = (int(v) for v in args.emulate.split("/"))
                return rt.partition_params(p, en, er)
            return rt.partition_params(p, world, rank) if per_process else p

        def render(p):
            if per_process or args.emulate:
                return wld.scene.render_device(cam, share(p), accum.data_ptr())
            return wld.scene.render_multi_device(cam, p, accum.data_ptr())      # every GPU's share + the reduce

        def step(flags):
            accum.zero_()
            torch.cuda.current_stream().synchronize()          # the renderer runs on its own streams
            st = render(job(flags))                            # returns after the streams have drained
            if per_process:
                D.reduce_to_root(accum)
            return st

        timed_flags = rt.RENDER_PROFILE | (rt.RENDER_STAGED if args.staged else 0)
        for _ in range(warmup):                                # (the first render of a scene also picks its kernel form)
            step(timed_flags)
        r = {"job": job, "render": render, "cam": cam}
        barrier()
        t0 = time.perf_counter()
        acc = {k: 0.0 for k in ("rays_traced", "paths", "ms_trace", "ms_shade", "ms_reduce", "trace_launches", "ms_kernel_span",
                                "ms_render", "ms_render_min")}
        clocks = []
        st = None
        for _ in range(steps):
            st = step(timed_flags)
            for k in acc:
                acc[k] += st[k]
            if st["shader_clock_ghz"] > 0:
                clocks.append(st["shader_clock_ghz"])
        barrier()
        elapsed = time.perf_counter() - t0
        stats = torch.tensor([elapsed, acc["rays_traced"], acc["paths"]], dtype=torch.float64, device=dev)
        mx, sm = stats.clone(), stats.clone()
        if grouped:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        if count_visits:
            # per-ray visit counts for the roofline: a counting pass over a few samples (not timed), on the
            # tree and in the form the timed steps ran
            accum_keep = accum.clone()
            accum.zero_()
            torch.cuda.synchronize()
            cst = render(job(rt.RENDER_COUNT_VISITS, samples=(0, min(spp, 4))))
            if grouped:
                c = torch.tensor([float(cst[k]) for k in ("box_tests", "tri_tests", "sph_tests", "rays_traced")], dtype=torch.float64, device=dev)
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                cst = dict(zip(("box_tests", "tri_tests", "sph_tests", "rays_traced"), c.tolist()))
            for k, name in (("box_tests", "n_box"), ("tri_tests", "n_tri"), ("sph_tests", "n_sph")):
                r[name] = cst[k] / max(cst["rays_traced"], 1)
            accum.copy_(accum_keep)
        r.update(acc)
        r.update({"elapsed_max": float(mx[0]), "total_rays": float(sm[1]), "total_paths": float(sm[2]), "last": st,
                  "clock_ghz": sum(clocks) / len(clocks) if clocks else None, "steps": steps,
                  "mean": accum.view(height, width, 3).mean().item() / spp, "accum_sum": float(accum.double().sum().item())})
        return r

    if grouped:
        D.reduce_to_root(accum)                                # builds the communicator outside the timed region
        torch.cuda.synchronize()
    wld, build_s = build_world(args.build)
    if args.wide_build == "device":
        wld.scene.set_wide_build(rt.WIDE_BUILD_DEVICE)
    m = measure(wld, args.traversal, args.steps, args.warmup)
    # The same job once more on the reference's trees in the reference's visiting order (PER_LANE / SWEEP,
    # whichever the library picks among those two): the strict form beside the chosen one.
    ref = None
    chosen_is_strict = (m["last"]["tree"] == 0 and rt.FORM_NAMES[m["last"]["kernel_form"]] in
                        ("sweep", "walk", "walk_lds", "walk_refill", "walk_lds_refill"))
    if not args.no_reference_leg and not chosen_is_strict and not args.staged and not args.emulate:
        ref_world, _ = build_world("reference")
        rm = measure(ref_world, "auto", args.steps, max(args.warmup, 1), count_visits=False)
        ref = {"value": round(rm["total_rays"] / rm["elapsed_max"] / 1e6, 2), "unit": "Mrays/s",
               "ms_per_step": round(rm["elapsed_max"] / args.steps * 1e3, 3),
               "build": "reference", "kernel_form": rt.FORM_NAMES[rm["last"]["kernel_form"]],
               "rays_per_step": int(rm["total_rays"] / args.steps), "accum_sum": rm["accum_sum"],
               "note": "the reference's trees walked in the reference's visiting order: hits exact by construction"}
        ref_world.close()
    elif chosen_is_strict:
        ref = {"note": "the timed run itself is a reference-order form on the reference's trees"}

    if rank == 0:
        st = m["last"]
        n_box, n_tri, n_sph = m["n_box"], m["n_tri"], m["n_sph"]
        rays, paths, launches, ms_trace = m["rays_traced"], m["paths"], m["trace_launches"], m["ms_trace"]
        elapsed_max, total_rays, total_paths = m["elapsed_max"], m["total_rays"], m["total_paths"]
        value = total_rays / elapsed_max / 1e6
        form_name = rt.FORM_NAMES[st["kernel_form"]] if st["kernel_form"] < len(rt.FORM_NAMES) else str(st["kernel_form"])
        tree_name = rt.TREE_NAMES[st["tree"]]
        is_wide = form_name.startswith("wide")
        bytes_per_ray = trace_bytes_per_ray(n_box, n_tri, n_sph, 80.0 if is_wide else 32.0)
        # the dominant kernel on one GPU (rank 0's, or the slowest GPU's of a one-process job):
        # algorithmic bytes of its launches.  Staged: the closest-hit kernel alone.  Default: the
        # per-batch kernel, which also holds raygen and shade, so their stream bytes count too
        # (shaded rays >= rays - paths: every ray either ends its path or is shaded).
        n_here = 1 if per_process else world               # GPUs whose rays and launches `rays`, `launches` hold
        rays_gpu, paths_gpu, launches_gpu = rays / n_here, paths / n_here, launches / n_here
        algo_bytes = bytes_per_ray * rays_gpu
        if not args.staged:
            algo_bytes += SHADE_BYTES_PER_RAY * max(rays_gpu - paths_gpu, 0) + RAYGEN_BYTES_PER_PATH * paths_gpu
        # The pipelines' kernels overlap, so the algorithmic rate is taken over the wall time of the timed
        # steps (what the chip delivered), not over the kernels' summed durations.
        algo_gbs = algo_bytes / elapsed_max / 1e9
        span = m["ms_kernel_span"]
        extra = {
            "kernel": "k_trace (closest hit)" if args.staged else
                      "k_path (raygen + closest hit and shade per depth, one launch per batch)",
            "rays_per_launch": int(rays_gpu / max(launches_gpu, 1)),
            "algorithmic_bytes_per_launch": int(algo_bytes / max(launches_gpu, 1)),
            "avg_launch_ms": round(ms_trace / max(launches_gpu, 1), 4), "launches": int(launches_gpu),
            "launch_timing": "HIP events around every launch on the stream it runs on; launches of the pipelines overlap "
                             "(launches_in_flight), so summed durations exceed the step",
            "launches_in_flight": round(ms_trace / span, 2) if span > 0 else None,
            "kernel_span_ms_per_step": round(span / args.steps, 3),
            "summed_launch_ms_per_step": round(ms_trace / args.steps, 2),
            "shade_ms_per_step": round(m["ms_shade"] / args.steps, 2),
        }
        headline = not (args.spp or args.size or args.emulate or args.staged)
        out = {
            "metric": "Mrays/sec (whole node), Cornell box 1024^2 1k spp",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {kind} {width}x{height}, {spp} spp, max depth {depth}",
                       "build": args.build, "traversal": args.traversal, "tree": tree_name, "kernel_form": form_name,
                       **({"wide_build": args.wide_build} if args.traversal == "wide" else {}),
                       "parallelism": (f"{world} processes x 1 GPU, torch.distributed reduce" if per_process else
                                       f"1 process x {world} GPU(s), rt_render_multi (RCCL reduce in the library)"),
                       "partition": "rt_partition: interleaved 64-row bands x sample ranges",
                       "rays_per_step": int(total_rays / args.steps), "paths_per_step": int(total_paths / args.steps),
                       "mean_radiance": round(m["mean"], 5), "scene_build_s": round(build_s, 3),
                       "reduce_ms_per_step": round(m["ms_reduce"] / args.steps, 3),
                       "accum_sum": m["accum_sum"]},
            "reference_order": ref,
            "roofline": roofline(args.workload if headline or args.spp else None, f"{tree_name}/{form_name}",
                                 form_name, total_rays / elapsed_max,
                                 world, algo_gbs, bytes_per_ray,
                                 {"box": round(n_box, 2), "tri": round(n_tri, 2), "sph": round(n_sph, 2)}, extra,
                                 clock_ghz=m["clock_ghz"]),
        }
        if not per_process:
            # what the library reports about its GPUs and its reduce (the last timed step)
            n_show = min(world, 16)
            out["multi_gpu"] = {
                "rccl_ranks": st["rccl_ranks"], "rccl_version": rt.lib.rt_rccl_version(), "reduce": "rccl" if st["reduce_mode"] == rt.REDUCE_RCCL else "ordered (copied to the root, added in rank order)",
                "reduce_ms": round(st["ms_reduce"], 3),
                "device_ms_render": [round(v, 3) for v in st["device_ms_render"][:n_show]],
                "ms_render_min": round(st["ms_render_min"], 3), "ms_render_max": round(st["ms_render"], 3),
                "slowest_device": st["slowest_device"],
                "device_kernel_form": [rt.FORM_NAMES[f] for f in st["device_form"][:n_show]],
                "device_tree": [rt.TREE_NAMES[t] for t in st["device_tree"][:n_show]]}
        if world == 1 and not args.no_cpu:
            def gpu_render(n_spp):
                accum.zero_()
                torch.cuda.synchronize()
                wld.scene.render_device(m["cam"], m["job"](0, n_spp=n_spp), accum.data_ptr())
                return accum.view(height, width, 3).cpu().numpy()
            base = cpu_baseline(kind, width, height, depth, args.cpu_seconds, gpu_render)
            if base:
                out["cpu_baseline"] = base
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
