"""world_size-2 gloo test of the data-parallel path on CPU.

The HIP kernels need a GPU, so the network plugged into the distributed wrapper here is the CPU
oracle (same module tree / parameter set); what is under test is the product's distributed logic
(repmode_amd/distributed.py): sharding, DDP settings, gradient averaging == single-process training
on the concatenated batch with per-shard BatchNorm statistics (what DataParallel would compute)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import Opts


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from repmode_amd import distributed as dist_
    from oracle import repmode_oracle as orc
    r, w, _ = dist_.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=2)
    ddp = dist_.wrap_ddp(net, None)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 16, 16, 16, generator=g)
    t = torch.randn(4, 1, 16, 16, 16, generator=g)
    tasks = torch.tensor([1, 4, 4, 9])
    lo, hi = dist_.shard_batch(4, rank, world)
    ddp.train()
    loss = torch.nn.functional.mse_loss(ddp(x[lo:hi], tasks[lo:hi]), t[lo:hi])
    loss.backward()
    mx = dist_.max_over_ranks(float(rank), torch.device('cpu'))
    assert mx == world - 1
    torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_ddp_equals_single_process(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    g0 = torch.load(tmp_path / 'g0.pt')
    g1 = torch.load(tmp_path / 'g1.pt')
    # single process: mean over shards of per-shard losses (per-shard BN statistics)
    from oracle import repmode_oracle as orc
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 16, 16, 16, generator=g)
    t = torch.randn(4, 1, 16, 16, 16, generator=g)
    tasks = torch.tensor([1, 4, 4, 9])
    net.train()
    loss = 0.5 * (torch.nn.functional.mse_loss(net(x[:2], tasks[:2]), t[:2]) +
                  torch.nn.functional.mse_loss(net(x[2:], tasks[2:]), t[2:]))
    loss.backward()
    for k, p in net.named_parameters():
        assert torch.equal(g0[k], g1[k]), k                      # all-reduced: identical on both ranks
        ref = p.grad
        assert (g0[k] - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max())) + 1e-7, k


def test_shard_batch():
    from repmode_amd.distributed import shard_batch
    assert [shard_batch(192, r, 8) for r in (0, 7)] == [(0, 24), (168, 192)]
