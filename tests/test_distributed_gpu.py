"""Two data-parallel ranks driving the REAL HIP path (both on cuda:0, gloo transport -- RCCL refuses two
ranks on one device): checks that DistributedDataParallel composes with the custom autograd Functions
(every parameter gets a gradient every step, gradients are identical on both ranks after the all-reduce,
and equal the single-process average over shards with per-shard BatchNorm statistics)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import Opts

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(5)
    # 16x64x64: the deepest level keeps 32 values per BatchNorm channel per rank (well conditioned)
    x = torch.randn(4, 1, 16, 64, 64, generator=g)
    t = torch.randn(4, 1, 16, 64, 64, generator=g)
    return x, t, [1, 4, 4, 9]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from repmode_amd import distributed as dist_
    from repmode_amd.model import Model
    dist_.init_from_env(backend='gloo')
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=2, dtype=torch.float32, distributed=True)
    x, t, tasks = _data()
    lo, hi = dist_.shard_batch(4, rank, world)
    m.ddp.train()
    out = m.ddp(x[lo:hi].cuda(), tasks[lo:hi])
    torch.nn.functional.mse_loss(out, t[lo:hi].cuda()).backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.net.named_parameters()}
    assert all(p.grad is not None for p in m.net.parameters())
    # one full optimiser step through the harness as well
    m.do_train_iter(x[lo:hi], t[lo:hi], torch.tensor(tasks[lo:hi]))
    torch.save(grads, os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_ddp_two_ranks_on_one_gpu(tmp_path):
    mp.start_processes(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    g0, g1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    from repmode_amd.model import Model
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=2, dtype=torch.float32)
    x, t, tasks = _data()
    m.net.train()
    loss = 0.5 * (torch.nn.functional.mse_loss(m.net(x[:2].cuda(), tasks[:2]), t[:2].cuda()) +
                  torch.nn.functional.mse_loss(m.net(x[2:].cuda(), tasks[2:]), t[2:].cuda()))
    loss.backward()
    gmax = max(float(p.grad.abs().max()) for p in m.net.parameters())
    for k, p in m.net.named_parameters():
        assert torch.equal(g0[k], g1[k]), k
        ref = p.grad.cpu()
        # f32 atomics (split-K, BatchNorm partial sums) make the summation order run-dependent; the deep
        # batch-norm chain amplifies it -> 2e-2 like the whole-net golden test.  Parameters whose gradient is tiny
        # next to the network's largest are judged on that scale (their own maximum is mostly that noise).
        assert (g0[k] - ref).abs().max() <= 2e-2 * max(float(ref.abs().max()), 1e-2 * gmax), k
