#!/usr/bin/env python3
"""Cost of the data-parallel machinery itself on ONE GPU (a single-rank RCCL group: bucket copies, autograd hooks and a
trivial all-reduce per bucket, no peer traffic) -- the part of multi-GPU scaling loss that is not communication:
the stock DistributedDataParallel wrapper against repmode_amd.distributed.GradReducer.
    python tools/ddp_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repmode_amd.model import Model

def run(distributed):
    torch.manual_seed(0)
    m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16, distributed=distributed)
    x = torch.randn(8, 1, 32, 64, 64, device='cuda'); t = torch.randn(8, 1, 32, 64, 64, device='cuda')
    task = torch.arange(8) % 12
    for _ in range(15): m.do_train_iter(x, t, task)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): m.do_train_iter(x, t, task)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e3

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
a = run(False); b = run('ddp'); e = run('reducer-always'); g = run('reducer'); c = run(False); d = run('ddp'); f = run('reducer-always'); h = run('reducer')
print('plain %.2f / %.2f ms per step, DDP(1 rank) %.2f / %.2f, GradReducer(1 rank) %.2f / %.2f, GradReducer without the '
      'collectives %.2f / %.2f' % (a, c, b, d, e, f, g, h))
torch.distributed.destroy_process_group()
