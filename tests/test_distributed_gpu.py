"""Two data-parallel ranks driving the REAL HIP path (both on cuda:0, gloo transport -- RCCL refuses two
ranks on one device): checks that the gradient reducer (repmode_amd.distributed.GradReducer: the MoDE gradient
kernels write into the communication buckets) and, for comparison, DistributedDataParallel compose with the custom
autograd Functions (every parameter gets a gradient every step, gradients are identical on both ranks after the
all-reduce, and equal the single-process average over shards with per-shard BatchNorm statistics)."""
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


def _worker(rank, world, port, out_dir, how, backend='gloo'):
    # gloo: both ranks on device 0; nccl (= RCCL): one device per rank
    local = rank if backend == 'nccl' else 0
    # (deterministic mode: the comparison with the single-process reference below is then not limited by the order of float atomics)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', REPMODE_DETERMINISTIC='1')
    from repmode_amd import distributed as dist_
    from repmode_amd.model import Model
    dist_.init_from_env(backend=backend)
    torch.cuda.set_device(local)
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-4, gpu_ids=local, mult_chan=2, dtype=torch.float32, distributed=how)
    x, t, tasks = _data()
    lo, hi = dist_.shard_batch(4, rank, world)
    module = m.ddp if how == 'ddp' else m.net
    module.train()
    out = module(x[lo:hi].to(m.device), tasks[lo:hi])
    torch.nn.functional.mse_loss(out, t[lo:hi].to(m.device)).backward()
    if how != 'ddp':
        m.reducer.finish()
        # the 19 MoDE blocks' five expert gradients each were written into the buckets by the kernels
        assert m.reducer.last_copied == len(list(m.net.parameters())) - 19 * 5, m.reducer.last_copied
        for k, p in m.net.named_parameters():
            assert p.grad.data_ptr() == m.reducer.by_param[p].ptr, k
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.net.named_parameters()}
    assert all(p.grad is not None for p in m.net.parameters())
    # one full optimiser step through the harness as well
    m.do_train_iter(x[lo:hi], t[lo:hi], torch.tensor(tasks[lo:hi]))
    torch.save(grads, os.path.join(out_dir, 'g%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('how', ['reducer', 'ddp'])
def test_ddp_two_ranks_on_one_gpu(tmp_path, how):
    mp.start_processes(_worker, args=(2, _free_port(), str(tmp_path), how), nprocs=2, join=True, start_method='spawn')
    g0, g1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    from repmode_amd.model import Model
    from repmode_amd import ops
    try:
        ops.set_deterministic(True)
        torch.manual_seed(0)
        m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=2, dtype=torch.float32)
        x, t, tasks = _data()
        m.net.train()
        loss = 0.5 * (torch.nn.functional.mse_loss(m.net(x[:2].cuda(), tasks[:2]), t[:2].cuda()) +
                      torch.nn.functional.mse_loss(m.net(x[2:].cuda(), tasks[2:]), t[2:].cuda()))
        loss.backward()
    finally:
        ops.set_deterministic(False)
    gmax = max(float(p.grad.abs().max()) for p in m.net.parameters())
    for k, p in m.net.named_parameters():
        assert torch.equal(g0[k], g1[k]), k
        ref = p.grad.cpu()
        # ranks and reference both run with a fixed summation order (deterministic mode); what is left is the average itself
        # ((a + b) / 2 on the ranks, 0.5 a + 0.5 b accumulated by autograd here): 1e-5 of the tensor's (or 1e-2 of the network's)
        # largest entry.  Round 2 compared under float atomics with 2e-2 and failed on a marginal 2-element tensor.
        assert (g0[k] - ref).abs().max() <= 1e-5 * max(float(ref.abs().max()), 1e-2 * gmax), k


@pytest.mark.timeout(900)
@pytest.mark.parametrize('how', ['reducer', 'ddp'])
def test_two_ranks_over_rccl(tmp_path, how):
    """The same two-rank check over RCCL (backend "nccl"), one GPU per rank: needs two devices -- skipped on the 1-GPU
    boxes this round's tests run on, there for the driver's multi-GPU node."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (RCCL refuses two ranks on one device)')
    mp.start_processes(_worker, args=(2, _free_port(), str(tmp_path), how, 'nccl'), nprocs=2, join=True, start_method='spawn')
    g0, g1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k               # identical averages on both ranks


def test_reducer_buckets_hold_the_kernels_gradients_single_process():
    """No process group: the reducer only provides the gradient buffers.  Three distinct tasks -> the deep levels
    run the per-expert formulation, the others the merged one.  The kernels then write the same numbers into bucket slices
    that they otherwise write into fresh tensors (and the 1x1 experts' gemm3 without its pre-zeroed split): in deterministic
    mode every gradient of the reducer run equals the plain run's BITWISE (round 2: 2-norm with a 10 % floor under float
    atomics, which could hide a partially wrong gradient -- ADVICE round 2)."""
    from repmode_amd.model import Model
    from repmode_amd import ops
    x, t, _ = _data()
    tasks = torch.tensor([1, 4, 9, 4])
    res = []
    try:
        ops.set_deterministic(True)
        for distributed in ('reducer', False):
            torch.manual_seed(0)
            m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=4, dtype=torch.float32, distributed=distributed)
            m.do_train_iter(x, t, tasks)
            if distributed:
                r = m.reducer
                assert ops._grad_out(r.buckets[0].entries[0].param) is not None
                n_par = len(list(m.net.parameters()))
                # all five expert gradients of every MoDE block are written into the buckets by the kernels -- also on the
                # per-expert levels (enc4, bottleneck, dec4), whose 1x1 experts' gradients come out of repmode_gemm3
                assert r.last_copied == n_par - 19 * 5, r.last_copied
                for p in m.net.parameters():
                    assert p.grad.data_ptr() == r.by_param[p].ptr
            res.append({k: p.grad.detach().cpu().clone() for k, p in m.net.named_parameters()})
            m.do_train_iter(x, t, tasks)                       # (a second step re-uses the buckets)
            if distributed:
                m.reducer.remove()
                ops.set_grad_sink(None)
    finally:
        ops.set_deterministic(False)
        ops.set_grad_sink(None)
    for k in res[1]:
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.timeout(900)
@pytest.mark.parametrize('rule', ['default', 'bf16 branch'])
def test_bench_two_ranks_share_one_gpu(rule):
    """bench.py's N > 1 branch (BASELINE configs[3]'s code path: one process per rank through torch.distributed.run, the
    DistributedDataParallel step, the max-over-ranks clock, the no_sync() leg behind ``config.no_comm_value``, ``config.comm``)
    on a one-GPU box: both ranks on cuda:0 over gloo (REPMODE_BENCH_SHARE_GPU=1 -- a check of the code path, never a
    measurement).  The line must parse and describe a two-rank job.  'bf16 branch': the buckets' dtype rule is made to pick
    bfloat16 (``--grad-compress auto``, threshold 0, gloo admitted), so that the wrapper's communication hook switches mid-run --
    in the steps before the timed region -- and the remaining steps travel as bfloat16: the branch an 8-GPU job that opts into
    the rule takes at 8 patches per rank.  'default': float32 buckets, pinned."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, REPMODE_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if rule == 'bf16 branch':
        env.update(REPMODE_COMPRESS_IF_RING_OVER='0', REPMODE_GRAD_RULE_BACKENDS='nccl,gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1',
           '--batch', '2', '--no-cpu-baseline']
    if rule == 'bf16 branch':
        cmd += ['--no-fwd', '--grad-compress', 'auto']      # ('default' keeps rank 0's forward-only leg: the other rank waits for it at the final barrier)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2'
    for k in ('value', 'ms_per_step'):
        assert d[k] > 0 and d[k] == d[k]
    cfg = d['config']
    assert cfg['no_comm_value'] > 0 and cfg['no_comm_ms_per_step'] > 0
    comm = cfg['comm']
    assert comm['backend'] == 'gloo' and comm['world_size'] == 2 and comm['ranks_seen'] == 2
    assert comm['scheme'] == 'DistributedDataParallel' and comm['bucket_mb'] == 48
    assert comm['n_buckets'] == len(comm['bucket_mbytes']) >= 4 and abs(sum(comm['bucket_mbytes']) - 4 * 123877633 / 2 ** 20) < 2
    assert 0 < comm['efficiency'] and cfg['ms_per_step_unprofiled'] > 0
    if rule == 'bf16 branch':
        assert comm['dtype_rule'].startswith('auto: ring estimate')  # (the rule ran: in the 3 setup steps + 1 warmup step)
        assert cfg['setup_steps_before_warmup'] == 3
        assert comm['dtype'] == 'bf16' and comm['dtype_rule'].endswith('-> bf16')
    else:
        # the default: float32 buckets, pinned (what the reference averages in) -- no rule, no setup steps
        assert comm['dtype'] == 'f32' and comm['dtype_rule'] == 'pinned' and cfg['setup_steps_before_warmup'] == 0
    assert 0 < cfg['final_loss'] < 10
    assert ('fwd' in d) == (rule == 'default')
    assert abs(comm['grad_bytes_fp32'] - 4 * 123877633) < 8
