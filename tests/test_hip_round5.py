"""Round-5 kernels against the CPU oracle (through the C ABI and the operator library): a per-expert MoDE block of the deep
levels as ONE launch per direction (csrc/deep_mode.hip; RepMode.py:171-192 by linearity + :204-208 and their autograd), the
box-mean pair that feeds its data gradient.  Needs a real MI355X: every test is marked ``gpu``.  Tolerances as
tests/test_hip_parity.py states them (bf16 operands rounded on both sides, float accumulation: 1e-4 of the tensor's max)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import record, rel_err
from oracle import repmode_oracle as orc
from test_hip_parity import DEV, TOL_BF16_ACC, _ops, _rand_experts

pytestmark = pytest.mark.gpu


DM_CASES = [
    # (n, (d, h, w), ci, co)
    (8, (2, 4, 4), 64, 128),      # two whole samples per tile; the reduction split over workgroups (zeroed outputs)
    (5, (1, 2, 2), 32, 32),       # smaller than the 2 x 4 x 4 brick, an odd sample count (a half-empty last tile)
    (24, (2, 4, 4), 32, 64),      # batch 24
    (3, (4, 8, 8), 48, 96),       # one 8 x 8 plane per tile; channel counts that are no tile multiples (3 chunks, 8 waves)
    (2, (4, 8, 8), 128, 256),     # a real level-3 layer's width: 8 chunks = one per wave
    (3, (2, 4, 8), 16, 40),       # the plane tile on a volume smaller than it, a single chunk
    (2, (3, 6, 7), 24, 40),       # ragged in every direction, a half-filled last chunk
    (8, (4, 8, 8), 256, 256),     # enc4.conv2 / dec4.conv2 at the benchmarked batch: plain stores
    (8, (2, 4, 4), 256, 512),     # bottle.conv1 at the benchmarked batch: 4 slices of the reduction
    (1, (4, 8, 8), 512, 256),     # dec4.conv1's width, four chunks per wave
]


def _box(x, k):
    """zero-padded k^3 box mean of [N, C, D, H, W] (RepMode.py:161-163, 176-180)."""
    c = x.shape[1]
    return F.conv3d(x, x.new_full((c, 1, k, k, k), 1.0 / k ** 3), padding=k // 2, groups=c)


def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize('case', DM_CASES)
def test_deep_mode_fwd_vs_oracle(case):
    """P_e = conv(x, K_e) for the five experts and y = sum_e g_e * P_e from ONE launch against F.conv3d on the same
    bf16-rounded operands (the avg experts' box means as the box kernel hands them over, themselves checked against the
    oracle's depthwise form)."""
    ops = _ops()
    n, shape, ci, co = case
    d, h, w = shape
    assert ops.deep_mode_plan(0, n, d, h, w, ci, co) >= 1
    gen = torch.Generator().manual_seed(n * 1000 + ci + co + w)
    k5, k3, k1, a3, a5 = _rand_experts(co, ci, gen)[:5]
    x = _bf(torch.randn(n, ci, *shape, generator=gen))
    gn = torch.softmax(torch.randn(n, 5, co, generator=gen), 1)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    wf, _ = ops.expert_frags(k5.to(DEV), k3.to(DEV), torch.bfloat16, want_wd=False)
    xs = ops.box_expand(x_cl)
    xs_c = xs.cpu().permute(0, 1, 5, 2, 3, 4)                       # [3, N, C, D, H, W]
    assert rel_err(xs_c[0], x) == 0.0
    assert rel_err(xs_c[1], _box(x, 3)) < 1e-5 and rel_err(xs_c[2], _box(x, 5)) < 1e-5
    k1d, a3d, a5d = (t.reshape(co, ci).contiguous().to(DEV) for t in (k1, a3, a5))
    p, y = ops.deep_mode_fwd(x_cl, wf, xs, k1d, a3d, a5d, gn.to(DEV))
    refs = [F.conv3d(x, _bf(k5), padding=2), F.conv3d(x, _bf(k3), padding=1), F.conv3d(x, _bf(k1)),
            F.conv3d(_bf(xs_c[1]), _bf(a3)), F.conv3d(_bf(xs_c[2]), _bf(a5))]
    p_c = p.cpu().permute(0, 1, 5, 2, 3, 4)
    errs = [rel_err(p_c[e], refs[e]) for e in range(5)]
    y_ref = sum(gn[:, e, :, None, None, None] * refs[e] for e in range(5))
    ey = rel_err(y.cpu().permute(0, 4, 1, 2, 3), y_ref)
    record('test_deep_mode_fwd_vs_oracle', case=str(case), p_err=max(errs), y_err=ey)
    assert max(errs) < TOL_BF16_ACC, errs
    assert ey < TOL_BF16_ACC


@pytest.mark.parametrize('case', DM_CASES)
def test_deep_mode_dgrad_vs_oracle(case):
    """dx = conv(G_0, flip K5^T) + conv(G_1, flip pad(K3)^T) + G_2 K1 + box3(G_3)/27 A3 + box5(G_4)/125 A5 from ONE launch
    against the oracle's transposed convolutions on the same bf16-rounded operands; float output 1e-4, bf16 output one
    rounding (4e-3)."""
    ops = _ops()
    n, shape, ci, co = case
    d, h, w = shape
    plan = ops.deep_mode_plan(1, n, d, h, w, ci, co)
    assert plan >= 1
    gen = torch.Generator().manual_seed(n * 999 + ci + co + h)
    k5, k3, k1, a3, a5 = _rand_experts(co, ci, gen)[:5]
    lo = _bf(torch.randn(2, n, co, *shape, generator=gen))
    s = torch.randn(3, n, co, *shape, generator=gen)
    _, wd = ops.expert_frags(k5.to(DEV), k3.to(DEV), torch.bfloat16, want_wd=True)
    lo_cl = lo.permute(0, 1, 3, 4, 5, 2).contiguous().to(DEV, torch.bfloat16)
    s_cl = s.permute(0, 1, 3, 4, 5, 2).contiguous().to(DEV)
    k1d, a3d, a5d = (t.reshape(co, ci).contiguous().to(DEV) for t in (k1, a3, a5))
    dx_ref = (F.conv_transpose3d(lo[0], _bf(k5), padding=2) + F.conv_transpose3d(lo[1], _bf(k3), padding=1) +
              F.conv_transpose3d(_bf(s[0]), _bf(k1)) + F.conv_transpose3d(_bf(s[1]), _bf(a3)) + F.conv_transpose3d(_bf(s[2]), _bf(a5)))
    dxf = ops.deep_mode_dgrad(lo_cl, wd, s_cl[0], s_cl[1], s_cl[2], k1d, a3d, a5d, ci, out_dtype=torch.float32)
    ef = rel_err(dxf.cpu().permute(0, 4, 1, 2, 3), dx_ref)
    record('test_deep_mode_dgrad_vs_oracle', case=str(case), plan=plan, err_f32=ef)
    assert dxf.dtype == torch.float32 and ef < TOL_BF16_ACC
    if plan == 1:
        dxb = ops.deep_mode_dgrad(lo_cl, wd, s_cl[0], s_cl[1], s_cl[2], k1d, a3d, a5d, ci, out_dtype=torch.bfloat16)
        assert dxb.dtype == torch.bfloat16
        assert rel_err(dxb.float().cpu().permute(0, 4, 1, 2, 3), dx_ref) < 4e-3
        # the element-typed output is the float one rounded once (the waves' sums meet in another order: a few flips)
        flips = (dxb.float() != dxf.bfloat16().float()).float().mean().item()
        assert flips < 1e-2, flips


def test_deep_mode_refuses_what_it_does_not_take():
    """Shapes outside the two tiles, float32 and the deterministic mode answer 0 (the operator library then keeps round 4's
    launches); calling the kernel anyway is an error, not a wrong answer."""
    ops = _ops()
    from repmode_amd import _lib
    assert ops.deep_mode_plan(0, 2, 6, 10, 8, 16, 32) == 0          # several bricks per sample
    assert ops.deep_mode_plan(0, 2, 8, 8, 8, 16, 32) == 0           # deeper than four planes
    assert ops.deep_mode_plan(1, 2, 4, 8, 16, 16, 32) == 0          # wider than the plane tile
    assert ops.deep_mode_plan(0, 2, 4, 8, 8, 12, 32) == 0           # reduction channels % 8
    assert ops.deep_mode_plan(0, 2, 4, 8, 8, 16, 32, torch.float32) == 0
    assert ops.deep_mode_plan(0, 2, 4, 8, 8, 16, 32) == 1
    ops.set_deterministic(True)
    try:
        assert ops.deep_mode_plan(0, 2, 4, 8, 8, 16, 32) == 0
    finally:
        ops.set_deterministic(False)
    x = torch.zeros(2, 6, 10, 8, 16, device=DEV, dtype=torch.bfloat16)
    z = torch.zeros(16, device=DEV)
    with pytest.raises(_lib.RepModeHipError):
        _lib.call('repmode_deep_mode_fwd', x.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                  z.data_ptr(), z.data_ptr(), 2, 6, 10, 8, 16, 32, None)


@pytest.mark.parametrize('shape,c', [((4, 8, 8), 64), ((2, 4, 4), 256), ((3, 6, 7), 24), ((1, 2, 2), 8)])
def test_box_pair_vs_oracle(shape, c):
    """box3(a) / 27 and box5(b) / 125 of two different tensors from one launch (autograd of RepMode.py:176-180: the box mean is
    self-adjoint)."""
    ops = _ops()
    gen = torch.Generator().manual_seed(c + shape[2])
    a = torch.randn(3, c, *shape, generator=gen)
    b = torch.randn(3, c, *shape, generator=gen)
    o3, o5 = ops.box_pair(a.permute(0, 2, 3, 4, 1).contiguous().to(DEV), b.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    assert rel_err(o3.cpu().permute(0, 4, 1, 2, 3), _box(a, 3)) < 1e-5
    assert rel_err(o5.cpu().permute(0, 4, 1, 2, 3), _box(b, 5)) < 1e-5


@pytest.mark.parametrize('ci,co,shape,n', [(64, 64, (2, 4, 4), 8), (48, 96, (4, 8, 8), 3), (256, 128, (4, 8, 8), 8), (128, 256, (4, 8, 8), 8),
                                           (256, 512, (2, 4, 4), 8)])
def test_deep_mode_in_the_operator(ci, co, shape, n):
    """The per-expert block through the operator library as one launch per direction (default) and as round 4's five: output,
    data gradient and every parameter gradient agree (bf16: a summation-order difference can flip a rounding), and the
    one-launch form matches the oracle's block on the bf16-rounded input."""
    ops = _ops()
    gen = torch.Generator().manual_seed(ci + co + n)
    ps = _rand_experts(co, ci, gen)
    tasks = [(5 * i + 2) % 12 for i in range(n)]
    x = torch.randn(n, *shape, ci, generator=gen).bfloat16()
    r = torch.randn(n, *shape, co, generator=gen)
    res = []
    before = ops.get_deep_mode()
    try:
        for mask in (15, 7, 0, 1, 2):      # (15: + the box-mean operands out of the gate mix's backward launch)
            ops.set_deep_mode(mask)
            dev = [p.to(DEV).requires_grad_(True) for p in ps]
            xd = x.to(DEV).requires_grad_(True)
            plan = ops.TaskPlan(tasks, 12, DEV, training=True)
            y = ops.mode_conv3d(xd, *dev, plan, mode='unmerged')
            (y.float() * r.to(DEV)).sum().backward()
            res.append([y.detach().float().cpu(), xd.grad.float().cpu()] + [p.grad.cpu() for p in dev])
    finally:
        ops.set_deep_mode(before)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert rel_err(a, b) < 1e-2
    xr = x.float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) for p in ps]
    y_ref = orc.mode_conv_pre_bn(xr, *pr, torch.tensor(tasks), training=True)
    (y_ref * r.permute(0, 4, 1, 2, 3)).sum().backward()
    errs = {'y': rel_err(res[0][0].permute(0, 4, 1, 2, 3), y_ref.detach()), 'dx': rel_err(res[0][1].permute(0, 4, 1, 2, 3), xr.grad)}
    for i, (got, p) in enumerate(zip(res[0][2:], pr)):
        errs['p%d' % i] = rel_err(got, p.grad)
    record('test_deep_mode_in_the_operator', case='%d->%d %s n=%d' % (ci, co, shape, n), **errs)
    assert max(errs.values()) < 2e-2, errs


@pytest.mark.timeout(600)
def test_hip_graph_step_with_the_builds_own_optimizer():
    """Model(hip_graph=True) replays the ROUND-5 step: the build's own optimizer pass (repmode_amd.optim.Adam, capturable: the
    step count on the device) is captured with forward + backward, and the per-expert blocks' operands kept across steps are the
    ones the captured pass writes (no layout launch inside the graph).  A bf16 network whose deep levels take the per-expert
    formulation, replayed against the same model stepping launch by launch: the same losses (the two differ by the order of
    float atomics only), the same step count, an interchangeable optimizer state."""
    from conftest import Opts
    from repmode_amd.model import Model
    from repmode_amd.optim import Adam
    ops = _ops()
    gen = torch.Generator().manual_seed(3)
    n, shape = 4, (16, 32, 32)
    tasks = torch.tensor([1, 5, 7, 1])
    xs = [torch.randn(n, 1, *shape, generator=gen) for _ in range(7)]
    ts = [torch.randn(n, 1, *shape, generator=gen) for _ in range(7)]
    models = []
    for graph in (True, False):
        torch.manual_seed(0)
        models.append(Model(Opts(), lr=1e-3, gpu_ids=0, mult_chan=8, dtype=torch.bfloat16, hip_graph=graph))
    g, e = models
    assert isinstance(g.optimizer, Adam) and g.optimizer.device_step and isinstance(e.optimizer, Adam) and not e.optimizer.device_step
    e.net.load_state_dict(g.net.state_dict())
    lg, le = [], []
    for i in range(7):
        g.do_train_iter(xs[i], ts[i], tasks)
        lg.append(float(g.last_loss))
        e.do_train_iter(xs[i], ts[i], tasks)
        le.append(float(e.last_loss))
        if i == 1:
            assert ops.torch_ops().frag_store_size() > 0          # warm-up steps left the per-expert blocks' operands in the store
    st = next(iter(g._graphs.values()))
    assert st['graph'] is not None and st['calls'] == Model.GRAPH_WARMUP + 1
    record('hip_graph_own_adam', graph=lg, eager=le)
    for a, b in zip(lg, le):
        assert abs(a - b) < 2e-2 * abs(b), (lg, le)
    assert le[-1] < le[0]                                         # (it trains)
    # the step count: 7 on the device, the host mirror after sync_steps(), and in a state_dict a stock optimizer can load
    assert int(g.optimizer._step_dev.item()) == 7
    sd = g.optimizer.state_dict()
    assert all(int(s['step']) == 7 for s in sd['state'].values())
    stock = torch.optim.Adam(g.net.parameters(), lr=1e-3)
    stock.load_state_dict(sd)
    # parameters after 7 steps, graph vs launch by launch.  Loose: Adam moves every element by ~lr per step whatever the
    # gradient's size, so a bf16 gradient element near zero whose sign the order of float atomics decides moves its parameter the
    # other way (measured 0.08 of the parameters' norm at lr 1e-3 on 0.05-scale parameters; the losses above agree to 1e-3)
    pg = torch.cat([p.detach().reshape(-1) for p in g.net.parameters()])
    pe = torch.cat([p.detach().reshape(-1) for p in e.net.parameters()])
    assert float((pg - pe).norm() / pe.norm()) < 0.2


def test_stale_expert_operands_are_caught_on_the_device():
    """The per-expert blocks' bf16 operands are kept across steps and trusted on autograd's version counters + an optimizer
    hook; a write that moves neither (``p.data.mul_``, a collective's broadcast, a foreign kernel) used to leave stale filters
    in use (advisor, round 4).  Every forward pass now compares sampled parameter bytes with the stored operands ON THE DEVICE and
    lays the block out again where they differ: the forward after such a write equals the forward after an explicit
    ``clear_frag_store()``, and with the check switched off it demonstrably does not."""
    from conftest import Opts
    from repmode_amd.model import Model
    ops = _ops()
    t = ops.torch_ops()
    gen = torch.Generator().manual_seed(5)
    n, shape = 4, (16, 32, 32)
    tasks = torch.tensor([1, 5, 7, 1])
    x, tgt = torch.randn(n, 1, *shape, generator=gen), torch.randn(n, 1, *shape, generator=gen)
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-3, gpu_ids=0, mult_chan=8, dtype=torch.bfloat16)
    for _ in range(2):
        m.do_train_iter(x, tgt, tasks)
    assert t.frag_store_size() > 0
    k5 = m.net.bottle_block.conv1.expert_conv5x5_conv
    xd = x.to(DEV)

    def fwd():
        with torch.no_grad():
            return m.net(xd, tasks).float().cpu()
    # (deterministic mode: two forwards of the same state agree bitwise, so "caught" can be asserted exactly -- the bf16 network
    # amplifies the order of float atomics to ~1e-2 at the output otherwise)
    ops.set_deterministic(True)
    try:
        y0 = fwd()
        assert torch.equal(y0, fwd())
        for verify in (False, True):
            t.set_frag_verify(verify)
            ver = k5._version
            k5.data.mul_(-2.0)                         # `.data` has a version counter of its own: the parameter's does not move
            assert k5._version == ver
            y1 = fwd()
            t.clear_frag_store()
            y2 = fwd()                                 # operands laid out afresh from the parameters
            assert rel_err(y2, y0) > 1e-2, rel_err(y2, y0)         # the write matters
            if verify:
                assert torch.equal(y1, y2)             # caught: the same as a fresh layout
            else:
                assert rel_err(y1, y2) > 1e-2          # not caught without the check: what the advisor described
            y0 = y2
    finally:
        t.set_frag_verify(True)
        ops.set_deterministic(False)


@pytest.mark.parametrize('ci,co,shape,n', [(64, 128, (4, 8, 8), 8), (128, 64, (2, 4, 4), 8), (32, 40, (3, 6, 7), 5)])
def test_batchnorm_statistics_from_the_one_launch_forward(ci, co, shape, n):
    """A per-expert MoDE block + BatchNorm3d + ReLU (RepMode.py:194-214) with the batch statistics taken from the one-launch
    forward's epilogue (default) against the separate statistics pass: output, running statistics and every gradient agree, and
    the block matches the oracle's block on the bf16-rounded input."""
    from repmode_amd.nn_modules.RepMode import MoDEConv
    ops = _ops()
    gen = torch.Generator().manual_seed(ci + co)
    tasks = torch.tensor([(5 * i + 2) % 12 for i in range(n)])
    x = torch.randn(n, ci, *shape, generator=gen).bfloat16()
    r = torch.randn(n, co, *shape, generator=gen)
    torch.manual_seed(1)
    blk = MoDEConv(5, 12, ci, co, dtype=torch.bfloat16).to(DEV).train()
    state = {k: v.clone() for k, v in blk.state_dict().items()}
    res = []
    before = ops.get_deep_mode()
    try:
        for mask in (15, 3):
            ops.set_deep_mode(mask)
            blk.load_state_dict(state)
            blk.zero_grad(set_to_none=True)
            xd = x.to(DEV).requires_grad_(True)
            y = blk(xd, tasks)
            (y.float() * r.to(DEV)).sum().backward()
            sl = blk.subsequent_layer[0]
            res.append([y.detach().float().cpu(), xd.grad.float().cpu(), sl.running_mean.cpu().clone(), sl.running_var.cpu().clone()] +
                       [p.grad.float().cpu() for p in blk.parameters()])
    finally:
        ops.set_deep_mode(before)
    errs = [rel_err(a, b) for a, b in zip(res[0], res[1])]
    record('bn_stats_from_deep_mode', case='%d->%d %s' % (ci, co, shape), errs=errs)
    assert errs[2] < 1e-5 and errs[3] < 1e-4, errs           # running mean / variance: float sums in another order
    assert max(errs) < 1e-2, errs                            # (bf16 output: a rounding can flip)
    ref = orc.MoDEConv(5, 12, ci, co)
    ref.load_state_dict(state)
    ref.train()
    xr = x.float().requires_grad_(True)
    yr = ref(xr, tasks)
    (yr * r).sum().backward()
    # (the float32 oracle on the bf16-rounded input: the HIP block also rounds its experts and its 1x1 operands to bf16, and a
    # ReLU / normalisation behind that -- 5e-2 of the tensor's max; the unnormalised block is held to 2e-2 in
    # test_deep_mode_in_the_operator)
    ey, edx = rel_err(res[0][0], yr.detach()), rel_err(res[0][1], xr.grad)
    erm = rel_err(res[0][2], ref.subsequent_layer[0].running_mean)
    record('bn_stats_from_deep_mode_vs_oracle', case='%d->%d %s' % (ci, co, shape), y=ey, dx=edx, running_mean=erm)
    # dx behind BatchNorm + ReLU: an output within 3e-3 of the oracle's still flips the ReLU mask of the
    # elements nearest zero, and every flipped element moves its data-gradient neighbourhood by a whole term (0.08-0.19 of max
    # against the oracle's OWN mask: recorded above as `dx`).  So the oracle runs a second time with the HIP path's mask in
    # place of its ReLU -- the masks must agree on more than 99 % of the elements -- and dx is BOUNDED against that: 2e-2 of
    # max, where a 10 % error would not pass.
    mask = (res[0][0] > 0).float()
    disagree = float(((yr.detach() > 0).float() != mask).float().mean())

    class _Mask(torch.nn.Module):
        def forward(self, z):
            return z * mask

    ref2 = orc.MoDEConv(5, 12, ci, co)
    ref2.load_state_dict(state)
    ref2.train()
    assert isinstance(ref2.subsequent_layer[1], torch.nn.ReLU)
    ref2.subsequent_layer[1] = _Mask()
    xr2 = x.float().requires_grad_(True)
    (ref2(xr2, tasks) * r).sum().backward()
    edx_masked = rel_err(res[0][1], xr2.grad)
    record('bn_stats_from_deep_mode_dx_same_mask', case='%d->%d %s' % (ci, co, shape), dx=edx_masked, mask_disagreement=disagree)
    assert disagree < 1e-2, disagree
    assert edx_masked < 2e-2, edx_masked
    assert rel_err(1.1 * res[0][1], xr2.grad) > 2e-2          # (an injected 10 % error falls outside the bound)
    assert ey < 2e-2, ey
    assert erm < 5e-3, erm
