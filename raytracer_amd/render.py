"""Render a frame on every GPU of the job and write it as a PPM on rank 0:

    python -m raytracer_amd.render --scene cornell --size 1024x1024 --spp 1024 --out frame.ppm
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m raytracer_amd.render ...

One process per GPU.  Every rank builds the scene on its own GPU, renders its share of the frame
(interleaved 64-row bands x a sample range: partition.py) into a zeroed full-frame buffer, and one
sum-reduce (RCCL over xGMI with the default backend) combines the buffers on rank 0, which
tonemaps and packs the frame as the reference does (main.rs:369-373) and writes it.  With one
process this is the reference's render() on one GPU.
"""
import argparse
import os
import time

import torch

import raytracer_amd as rt
from raytracer_amd import distributed as D
from raytracer_amd.partition import partition

SCENES = ("cornell", "cornell_plane", "world1", "world2", "world2_glass")


def make_world(ctx, scene, obj=None):
    if scene == "cornell":
        return rt.World.cornell_box(ctx, obj or os.path.join(rt.ASSET_DIR, "cornell_box.obj"))
    if scene == "cornell_plane":
        return rt.World.cornell_box(ctx, obj or os.path.join(rt.ASSET_DIR, "cornell_box_plane.obj"))
    if scene == "world1":
        return rt.World.world1(ctx)
    if scene in ("world2", "world2_glass"):
        return rt.World.world2(ctx, obj, variant=1 if scene == "world2_glass" else 0)
    raise ValueError(scene)


def write_ppm(path, packed, width, height):
    rgb = torch.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], dim=-1).to(torch.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (width, height))
        f.write(rgb.cpu().numpy().tobytes())


def render_frame(world, width, height, spp, depth, rank, world_size, device, seed=rt.DEFAULT_SEED):
    """This rank's share of the frame, reduced to rank 0.  Returns (accum[h*w*3] on `device`, stats);
    the buffer is complete on rank 0 only."""
    accum = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
    torch.cuda.current_stream().synchronize()              # the renderer runs on its own stream
    bands, samples = partition(width, height, spp, world_size, rank)
    st = world.scene.render_device(world.camera(width, height),
                                   world.params(width, height, spp, depth, samples=samples, bands=bands, seed=seed),
                                   accum.data_ptr())
    D.reduce_to_root(accum)
    return accum, st


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", choices=SCENES, default="cornell")
    ap.add_argument("--obj", default=None, help="OBJ file of the Cornell / world2 scenes (default: the shipped asset)")
    ap.add_argument("--size", default="400x400")           # main.rs:24
    ap.add_argument("--spp", type=int, default=400)        # main.rs:20
    ap.add_argument("--depth", type=int, default=6)        # main.rs:21
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=rt.DEFAULT_SEED)
    ap.add_argument("--sah", action="store_true", help="opt-in SAH mesh trees")
    ap.add_argument("--wide", choices=["off", "collapse", "device"], default="off",
                    help="the opt-in wide walk: eight-child tree collapsed on the host from the binary trees, or built on the GPU")
    ap.add_argument("--backend", default=os.environ.get("RT_DIST_BACKEND", ""))
    ap.add_argument("--out", default="frame.ppm")
    args = ap.parse_args(argv)
    width, height = (int(v) for v in args.size.lower().split("x"))

    rank, world_size, local = D.init_from_env(args.backend or None)
    if local >= torch.cuda.device_count():                 # several ranks on one GPU (tests only)
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    t0 = time.perf_counter()
    ctx = rt.Context(local)
    if args.sah:
        rt.set_default_build(rt.BUILD_SAH)
    world = make_world(ctx, args.scene, args.obj)
    if args.wide != "off":
        if args.wide == "device":
            world.scene.set_wide_build(rt.WIDE_BUILD_DEVICE)
        world.scene.set_traversal(rt.TRAVERSAL_WIDE)
    accum, st = render_frame(world, width, height, args.spp, args.depth, rank, world_size, device, args.seed)
    if rank == 0:
        packed = torch.zeros(height * width, dtype=torch.int32, device=device)
        torch.cuda.current_stream().synchronize()          # the fill runs on torch's stream, the tonemap on the context's
        ctx.tonemap_device(accum.data_ptr(), args.spp, packed.data_ptr(), height * width)
        write_ppm(args.out, packed, width, height)
        print(f"Finished in {time.perf_counter() - t0:.2f}s: {args.scene} {width}x{height}, {args.spp} spp on "
              f"{world_size} GPU(s); rank 0 traced {st['rays_traced']} rays in {st['ms_render']:.1f} ms -> {args.out}")
    world.close()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
