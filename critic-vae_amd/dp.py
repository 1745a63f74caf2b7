"""Data parallelism for the training step: one process per GPU, one flat all-reduce per step.

The reference is single-device (vae_parameters.py:2); the north star shards the minibatch over
the GPUs of one node.  Semantics are standard DDP: rank r takes rows
[r*B, (r+1)*B) of the global batch, BatchNorm statistics and the MS-SSIM means are per rank,
gradients are summed by ONE all-reduce over the flat fp32 gradient buffer (10.3 MB; RCCL over
xGMI when the backend is "nccl") and the 1/N is folded into the fused Adam's grad_scale.
"""
import os

# The pool's hosts only support dmabuf IPC; with the legacy mode RCCL's hipIpcGetMemHandle fails ("invalid argument")
# as soon as two ranks exchange buffers.  The HSA runtime reads the variable when the process first touches the GPU,
# so it is set at import (an exported value wins) and again in init() for callers that import late.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    world, rank, local = env_world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # CVAE_DIST_BACKEND=gloo: rehearse the multi-rank path without RCCL (e.g. 2 ranks, 1 GPU)
            backend = os.environ.get("CVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(device_index(local))
            kw["device_id"] = torch.device("cuda", device_index(local))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def device_index(local_rank):
    """GPU of this rank: LOCAL_RANK, unless CVAE_DEVICE pins every rank to one device (rehearsals)."""
    forced = os.environ.get("CVAE_DEVICE")
    return int(forced) if forced is not None else local_rank


def shard_rows(global_batch, world, rank):
    """Row range of the global batch owned by `rank` (global_batch must divide evenly)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, per


def allreduce_mean_(flat_grad, world):
    """In-place mean over ranks of the flat gradient (sum all-reduce, then 1/N)."""
    if world > 1:
        dist.all_reduce(flat_grad)
        flat_grad.mul_(1.0 / world)
    return flat_grad


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
