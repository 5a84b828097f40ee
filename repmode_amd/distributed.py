"""Data-parallel training across the GPUs of one node: one process per GPU, gradients all-reduced by
RCCL over xGMI (``torch.distributed`` backend "nccl" IS RCCL on ROCm), overlapped with backward.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(fnet_model.py:40-44), which is replaced, not translated.  Samples are independent units of the
MoDE path (each has its own merged filter and its own conv, RepMode.py:183-189, 204-208); the only
couplings are training-mode BatchNorm statistics -- kept per rank, as DataParallel's replicas
would -- and the gradient sum.  So the path shards with exactly one collective per step: an
all-reduce (SUM / world) of the 193 gradient tensors (123,877,633 elements at mult_chan=32), issued
per bucket as soon as the bucket's gradients exist.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce is bound by one link, so
buckets are sized to keep several collectives in flight under the remaining backward work instead
of one large tail transfer: 48 MB fp32 buckets (~10 per step).  88 % of the gradient bytes belong
to the deep levels (enc4 / bottleneck / dec4) whose backward finishes in the first third of the
backward pass -- the level 0-1 encoder backward (most of the FLOPs) hides their transfer.
"""
import os

import torch
import torch.distributed as dist

BUCKET_MB = 48


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def wrap_ddp(net, device=None):
    """DistributedDataParallel with the settings the MoDE path wants:
      * every parameter gets a gradient every step (all experts and the whole gate matrix take part;
        unused task columns receive exact zeros) -> ``find_unused_parameters=False``;
      * BatchNorm running statistics stay per rank -> ``broadcast_buffers=False``;
      * gradients live inside the communication buckets -> ``gradient_as_bucket_view=True``.
    """
    from torch.nn.parallel import DistributedDataParallel as DDP
    kwargs = dict(broadcast_buffers=False, find_unused_parameters=False, gradient_as_bucket_view=True,
                  bucket_cap_mb=BUCKET_MB)
    if device is not None and device.type == 'cuda':
        return DDP(net, device_ids=[device.index], output_device=device.index, **kwargs)
    return DDP(net, **kwargs)


def shard_batch(global_batch, rank, world):
    """Contiguous per-rank slice [lo, hi) of a global batch (weak scaling keeps hi-lo fixed)."""
    per = global_batch // world
    return rank * per, (rank + 1) * per


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (used for step timing)."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
