"""One process per GPU: render this rank's share, then one sum-reduce of the accumulation
buffers to rank 0 (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).
The reduce is the only collective of the path."""
import os

import torch
import torch.distributed as dist

from .partition import partition


def init_from_env(backend=None, force=False):
    """(rank, world_size, local_rank); initialises the default process group when WORLD_SIZE > 1
    (or for a single rank too when `force` is set: the collectives then run through the backend
    as they would on a node, which is how the RCCL path is exercised on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def reduce_to_root(accum):
    """Sum the per-rank accumulation buffers into rank 0's (in place).  No-op without a process group.
    Returns when the sum is complete in the buffer: under RCCL dist.reduce() only orders torch's
    current stream behind the collective, and the renderer's kernels (tonemap included) run on the
    context's own stream, which knows nothing of torch's."""
    if dist.is_available() and dist.is_initialized():
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if accum.is_cuda:
            torch.cuda.current_stream(accum.device).synchronize()
    return accum


def render_partitioned(render_part, width, height, spp, rank, world_size, accum):
    """Runs `render_part(bands, (s0, s1), accum)` for this rank's partition, then reduces.
    `render_part` adds the linear sample sums of that band group / sample range into `accum`
    (the GPU path passes a closure over Scene.render_device).  Returns accum (complete on rank 0
    only)."""
    bands, samples = partition(width, height, spp, world_size, rank)
    render_part(bands, samples, accum)
    return reduce_to_root(accum)
