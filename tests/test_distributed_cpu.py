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


def _worker(rank, world, port, out_dir, how='ddp'):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from repmode_amd import distributed as dist_
    from oracle import repmode_oracle as orc
    r, w, _ = dist_.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=2)
    compress = how.endswith('-bf16')
    if how.startswith('ddp'):
        ddp = dist_.wrap_ddp(net, None, grad_compress='auto' if how == 'ddp-switch-bf16' else 'bf16' if compress else None)
    else:
        if rank == 1:                       # the reducer must bring rank 1 onto rank 0's parameters itself
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.25)
        ddp, red = net, dist_.GradReducer(net, bucket_mb=0.01, comm_dtype=torch.bfloat16 if compress else None)
        assert len(red.buckets) > 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 16, 16, 16, generator=g)
    t = torch.randn(4, 1, 16, 16, 16, generator=g)
    tasks = torch.tensor([1, 4, 4, 9])
    lo, hi = dist_.shard_batch(4, rank, world)
    ddp.train()
    if how == 'ddp-switch-bf16':
        # what Model._apply_grad_dtype_rule does when the rule picks bfloat16 after the first steps: the ONE wrapper has run a
        # float32 step, then its communication hook's flag flips (no second wrapper, no broadcast of parameters or buffers)
        assert ddp.grad_dtype_switch is not None and ddp.grad_dtype_switch.dtype is None
        bn_before = [b.clone() for b in net.buffers()]
        torch.nn.functional.mse_loss(ddp(x[lo:hi], tasks[lo:hi]), t[lo:hi]).backward()
        net.zero_grad(set_to_none=True)
        ddp.grad_dtype_switch.dtype = 'bf16'
        with torch.no_grad():                 # (the float32 step moved the running statistics: put them back for the comparison)
            for b, b0 in zip(net.buffers(), bn_before):
                b.copy_(b0)
    loss = torch.nn.functional.mse_loss(ddp(x[lo:hi], tasks[lo:hi]), t[lo:hi])
    loss.backward()
    if not how.startswith('ddp'):
        red.finish()
        assert red.last_copied == len(list(net.parameters()))        # (the CPU oracle does not write into the buckets)
        lo_, hi_ = red.buckets[0].flat.data_ptr(), red.buckets[0].flat.data_ptr() + red.buckets[0].flat.numel() * 4
        assert lo_ <= red.buckets[0].entries[0].param.grad.data_ptr() < hi_
    mx = dist_.max_over_ranks(float(rank), torch.device('cpu'))
    assert mx == world - 1
    torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('how', ['reducer', 'ddp', 'reducer-bf16', 'ddp-bf16', 'ddp-switch-bf16'])
def test_two_rank_ddp_equals_single_process(tmp_path, how):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), how), nprocs=world, join=True,
                       start_method='spawn')
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
        # (bf16 buckets: each rank's gradient is rounded to 8 bits of mantissa before the sum, and the sum once more)
        tol = 1e-5 if not how.endswith('-bf16') else 2e-2
        assert (g0[k] - ref).abs().max() <= tol * max(1.0, float(ref.abs().max())) + 1e-7, k


def _worker8(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from repmode_amd import distributed as dist_
    from oracle import repmode_oracle as orc
    dist_.init_from_env(backend='gloo')
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=2)
    ddp = dist_.wrap_ddp(net, None)
    x, t, tasks = _batch8()
    lo, hi = dist_.shard_batch(x.shape[0], rank, world)
    ddp.train()
    torch.nn.functional.mse_loss(ddp(x[lo:hi], tasks[lo:hi]), t[lo:hi]).backward()
    # the buckets' dtype rule: every rank measures its own backward, all take the MAX -> the same decision everywhere
    bwd = dist_.max_over_ranks(5.0 + rank, torch.device('cpu'))
    nbytes = 4 * 123877633
    pick = dist_.pick_grad_dtype(nbytes, world, bwd, 'nccl')
    torch.save({'grads': {k: p.grad.clone() for k, p in net.named_parameters()}, 'bwd': bwd, 'pick': pick,
                'running_mean': net.state_dict()['encoder_block1.conv_more.conv1.subsequent_layer.0.running_mean'].clone()},
               os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.destroy_process_group()


def _batch8():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(16, 1, 16, 16, 16, generator=g)
    t = torch.randn(16, 1, 16, 16, 16, generator=g)
    tasks = torch.tensor([0, 0, 1, 5, 11, 3, 7, 7, 2, 9, 4, 4, 6, 10, 8, 1])      # a different task mix on every rank
    return x, t, tasks


@pytest.mark.timeout(900)
def test_eight_rank_ddp_equals_single_process(tmp_path):
    """The target topology (BASELINE configs[3]: eight ranks), on CPU over gloo at mult_chan 2: every rank with its own task
    mix, gradients == single-process training on the concatenated batch with per-shard BatchNorm statistics, BatchNorm running
    statistics stay per rank, and every rank takes the same decision of the buckets' dtype rule from different local timings."""
    world = 8
    mp.start_processes(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / ('g%d.pt' % r)) for r in range(world)]
    from oracle import repmode_oracle as orc
    from repmode_amd.distributed import shard_batch
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=2)
    x, t, tasks = _batch8()
    net.train()
    loss = 0
    for r in range(world):
        lo, hi = shard_batch(16, r, world)
        loss = loss + torch.nn.functional.mse_loss(net(x[lo:hi], tasks[lo:hi]), t[lo:hi]) / world
    loss.backward()
    for k, p in net.named_parameters():
        for r in range(1, world):
            assert torch.equal(got[0]['grads'][k], got[r]['grads'][k]), k
        assert (got[0]['grads'][k] - p.grad).abs().max() <= 1e-5 * max(1.0, float(p.grad.abs().max())) + 1e-7, k
    assert all(g['bwd'] == 12.0 for g in got) and len({g['pick'] for g in got}) == 1
    # per-rank BatchNorm statistics: the ranks saw different shards, nothing broadcast them
    assert not torch.equal(got[0]['running_mean'], got[5]['running_mean'])


def test_shard_batch():
    from repmode_amd.distributed import shard_batch
    assert [shard_batch(192, r, 8) for r in (0, 7)] == [(0, 24), (168, 192)]


class _ToyScale(torch.autograd.Function):
    """y = x * w (elementwise, w a parameter): writes w's gradient into ops._grad_out like the MoDE kernels' host code."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, dy):
        from repmode_amd import ops
        x, w = ctx.saved_tensors
        gw = ops._grad_out(w)
        torch.mul(dy, x, out=gw)
        return dy * w, gw


def test_grad_reducer_adopts_bucket_slices_single_process():
    """Without a process group: gradients written into ``grad_buffer`` slices become ``param.grad`` with no copy, the
    others are gathered; an existing .grad is never aliased; a parameter without gradient is reported."""
    from repmode_amd import ops
    from repmode_amd.distributed import GradReducer

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.randn(5, 7))
            self.lin = torch.nn.Linear(7, 3)
            self.b = torch.nn.Parameter(torch.randn(5, 3))

        def forward(self, x):
            return _ToyScale.apply(self.lin(_ToyScale.apply(x, self.a)), self.b)

    torch.manual_seed(1)
    net, ref = Toy(), Toy()
    ref.load_state_dict(net.state_dict())
    x = torch.randn(5, 7)
    red = GradReducer(net, bucket_mb=1e-4)
    assert len(red.buckets) >= 2
    ops.set_grad_sink(red)
    try:
        for step in range(2):
            net.zero_grad(set_to_none=True)
            net(x).square().sum().backward()
            red.finish()
            assert red.last_copied == 2                          # lin.weight, lin.bias
            for e in (red.by_param[net.a], red.by_param[net.b]):
                assert e.param.grad.data_ptr() == e.ptr
        ref(x).square().sum().backward()
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-6), k
        # accumulation into an existing gradient: the kernel must get a private buffer, the sum lands in the bucket
        net(x).square().sum().backward()
        red.finish()
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.allclose(p.grad, 2 * q.grad, rtol=1e-5, atol=1e-6), k
        # a parameter left out of the graph
        net.zero_grad(set_to_none=True)
        net.lin(x).sum().backward()
        with pytest.raises(RuntimeError, match='received no gradient'):
            red.finish()
        net.zero_grad(set_to_none=True)
        net(x).square().sum().backward()
        red.finish()                                             # and the reducer is usable again
    finally:
        ops.set_grad_sink(None)
        red.remove()


def test_gradient_dtype_rule():
    """distributed.pick_grad_dtype: bfloat16 buckets when the per-link ring estimate of the float32
    all-reduce exceeds 0.6 of the measured backward pass, on RCCL only."""
    from repmode_amd import distributed as dist_
    nbytes = 4 * 123877633                                   # the network's gradients
    ring8 = dist_.ring_allreduce_ms(nbytes, 8)
    assert abs(ring8 - 2 * 7 / 8 * nbytes / 153e9 * 1e3) < 1e-9 and 5.6 < ring8 < 5.8      # SURVEY: ~5.7 ms
    assert dist_.ring_allreduce_ms(nbytes, 1) == 0.0
    assert dist_.pick_grad_dtype(nbytes, 8, 15.0, 'nccl') is None          # configs[3]: 24 patches per rank, ~15 ms backward
    assert dist_.pick_grad_dtype(nbytes, 8, 6.5, 'nccl') == 'bf16'         # 8 patches per rank: the ring is as long as the backward
    assert dist_.pick_grad_dtype(nbytes, 2, 6.5, 'nccl') is None           # two ranks: 3.2 ms hides
    assert dist_.pick_grad_dtype(nbytes, 8, 6.5, 'gloo') is None
    assert dist_.pick_grad_dtype(nbytes, 1, 6.5, 'nccl') is None
