"""Round-6 kernels against the CPU oracle (through the C ABI and the operator library).  Needs a real MI355X: every test is
marked ``gpu``.  Tolerances as tests/test_hip_parity.py states them (bf16 operands rounded on both sides, float accumulation:
1e-4 of the tensor's max)."""
import ctypes

import pytest
import torch

from conftest import record, rel_err
from oracle import repmode_oracle as orc
from test_hip_parity import DEV, TOL_BF16_ACC, _ops

pytestmark = pytest.mark.gpu


WGRAD_COL_CASES = [
    # (N, D, H, W, Cin, Cout, tasks): the column-walking filter gradient (csrc/conv5_wgrad_col.hip) -- the 8 x 32 tile
    (2, 6, 10, 40, 32, 32, [5, 9]),                         # ragged in y and x, one sample per slot
    (3, 5, 9, 35, 16, 48, [5, 9, 5]),                       # a slot with two samples, one 16-channel tile of ci
    (2, 4, 16, 64, 64, 32, [1, 1]),                         # one slot: ranges that share every unit (float atomics)
    (6, 2, 8, 32, 40, 72, [3, 3, 7, 3, 0, 7]),              # D = 2: no step has all five planes; half-filled channel tiles
    (1, 1, 4, 33, 8, 8, [2]),                               # D = 1, a half-filled 16-channel tile on both sides
    (9, 8, 16, 32, 64, 64, [4, 4, 4, 8, 8, 1, 1, 1, 1]),    # slots of 3 / 2 / 4 samples
    (2, 3, 8, 32, 16, 16, [0, 1]),                          # D = 3
    # volumes 16 .. 31 voxels wide: the 16 x 16 tile (H > 8) and the 8 x 16 tile
    (8, 8, 16, 16, 64, 128, [0, 1, 2, 3, 4, 5, 6, 7]),      # level 2 of the network at the benchmarked batch: plain stores only
    (8, 8, 16, 16, 128, 128, [0, 1, 2, 3, 4, 5, 6, 7]),     # two whole units per workgroup
    (8, 8, 16, 16, 128, 128, [0, 0, 0, 1, 2, 2, 3, 0]),     # whole units per workgroup with 4 / 1 / 2 / 1 samples per slot
    (2, 6, 10, 20, 32, 32, [5, 9]),
    (3, 5, 9, 17, 16, 48, [5, 9, 5]),
    (4, 4, 8, 16, 32, 64, [2, 2, 3, 3]),                    # H = 8: the 8 x 16 tile
    (2, 7, 5, 24, 24, 40, [6, 6]),
]


def _wgrad_reference(case, gen):
    n, d, h, w, cin, cout, tasks = case
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    uniq = sorted(set(tasks), key=tasks.index)
    return x, dy, uniq


@pytest.mark.parametrize('case', WGRAD_COL_CASES)
def test_conv5_wgrad_column_form_vs_oracle(case):
    """The filter gradient's column-walking form (mode 2: wherever eligible) against the oracle -- autograd of the per-sample
    convolution, RepMode.py:207 -- and against conv5_wgrad.hip's regular grid (mode 0; same products, another split of the voxel
    sums over workgroups: float summation order)."""
    ops = _ops()
    n, d, h, w, cin, cout, tasks = case
    gen = torch.Generator().manual_seed(sum(case[:6]) + 11)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(plan.nslots, cout, cin, 5, 5, 5, requires_grad=True)
    slots = torch.tensor([plan.slot_task_host.index(t) for t in tasks])
    (orc.conv_per_sample(x, wt[slots]) * dy).sum().backward()
    dw_ref = wt.grad.reshape(plan.nslots, cout, cin, 125).permute(0, 3, 1, 2)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    got = []
    default = ops.get_wgrad_col()
    try:
        for mode in (2, 0):
            ops.set_wgrad_col(mode)
            got.append(ops.conv5_wgrad(x_cl, dy_cl, plan, cout).cpu())
    finally:
        ops.set_wgrad_col(default)
    e = rel_err(got[0], dw_ref)
    record('wgrad_col', case=list(case[:6]), err=e, vs_regular=rel_err(got[0], got[1]))
    assert e < TOL_BF16_ACC
    assert rel_err(got[0], got[1]) < 1e-4


WGRAD_SPLIT_CASES = [
    # (N, D, H, W, Cin, Cout, tasks, q): the taps of a unit split over q workgroups (plain stores only, whatever the unit count)
    (8, 16, 32, 32, 64, 64, [0, 1, 2, 3, 4, 5, 6, 7], 2),       # level 1 at the benchmarked batch: 128 units x 2 parts = 256 workgroups
    (8, 16, 32, 32, 32, 64, [0, 1, 2, 3, 4, 5, 6, 7], 2),       # its first layer: 64 units x 2 parts on half the chip
    (3, 5, 9, 35, 16, 48, [5, 9, 5], 2),                        # few units: a workgroup owns parts of several units, one after the other
    (6, 2, 8, 32, 40, 72, [3, 3, 7, 3, 0, 7], 2),
    (4, 4, 12, 16, 32, 64, [2, 2, 3, 3], 2),                    # the 16 x 16 tile
]


@pytest.mark.parametrize('case', WGRAD_SPLIT_CASES)
def test_conv5_wgrad_column_form_tap_split_vs_oracle(case):
    """The column form with a unit's 125 taps split over q workgroups (each walks the unit's whole voxel range with 1 / q of the
    accumulators): every element written once, by plain stores, into a buffer that starts out full of NaN -- against the oracle."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, cin, cout, tasks, q = case
    gen = torch.Generator().manual_seed(sum(case[:6]) + q)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(plan.nslots, cout, cin, 5, 5, 5, requires_grad=True)
    slots = torch.tensor([plan.slot_task_host.index(t) for t in tasks])
    (orc.conv_per_sample(x, wt[slots]) * dy).sum().backward()
    dw_ref = wt.grad.reshape(plan.nslots, cout, cin, 125).permute(0, 3, 1, 2)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    mode, split = ops.get_wgrad_col(), ops.get_wgrad_col_split()
    try:
        ops.set_wgrad_col(2)
        ops.set_wgrad_col_split(q)
        direct = ctypes.c_int(-1)
        _lib.call('repmode_conv5_wgrad_plan', plan.nslots, n, d, h, w, cin, cout, _lib.BF16, 0, ctypes.byref(direct))
        assert direct.value == 1
        dw = torch.full((plan.nslots, 125, cout, cin), float('nan'), device=DEV)
        _lib.call('repmode_conv5_wgrad_ex', x_cl.data_ptr(), dy_cl.data_ptr(), plan.sample_slot.data_ptr(), plan.nslots,
                  dw.data_ptr(), n, d, h, w, cin, cout, _lib.BF16, 8, torch.cuda.current_stream().cuda_stream)
    finally:
        ops.set_wgrad_col(mode)
        ops.set_wgrad_col_split(split)
    assert torch.isfinite(dw).all()
    e = rel_err(dw.cpu(), dw_ref)
    record('wgrad_col_split', case=list(case[:6]) + [q], err=e)
    assert e < TOL_BF16_ACC


@pytest.mark.parametrize('tasks', [list(range(8)), [3, 3, 3, 5, 9, 9, 3, 5], list(range(12)) * 2])
def test_conv5_wgrad_column_form_plain_stores_into_uninitialised_memory(tasks):
    """With enough units to fill the chip a workgroup takes whole units and the planner promises plain stores only
    (repmode_conv5_wgrad_plan), whatever the slots' sample counts: the launch must then overwrite EVERY element of a buffer full
    of NaNs (mode bit 3: no memset)."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, cin, cout = len(tasks), 8, 16, 16, 128, 128        # (3 slots x 8 x 8 units: three quarters of the chip)
    gen = torch.Generator().manual_seed(77)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    default = ops.get_wgrad_col()
    try:
        ops.set_wgrad_col(1)                 # (the default policy takes this shape)
        direct = ctypes.c_int(-1)
        _lib.call('repmode_conv5_wgrad_plan', plan.nslots, n, d, h, w, cin, cout, _lib.BF16, 0, ctypes.byref(direct))
        assert direct.value == 1
        dw = torch.full((plan.nslots, 125, cout, cin), float('nan'), device=DEV)
        _lib.call('repmode_conv5_wgrad_ex', x_cl.data_ptr(), dy_cl.data_ptr(), plan.sample_slot.data_ptr(), plan.nslots,
                  dw.data_ptr(), n, d, h, w, cin, cout, _lib.BF16, 8, torch.cuda.current_stream().cuda_stream)
        ops.set_wgrad_col(0)
        ref = ops.conv5_wgrad(x_cl, dy_cl, plan, cout)
    finally:
        ops.set_wgrad_col(default)
    assert torch.isfinite(dw).all()
    assert rel_err(dw.cpu(), ref.cpu()) < 1e-4


def test_conv5_wgrad_column_form_channel_ranges_of_a_skip_connection():
    """repmode_conv5_wgrad_part: the two tensors of a skip connection fill two input-channel ranges of ONE cleared dw -- against
    the filter gradient of the concatenated tensor."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, ca, cb, cout = 3, 4, 16, 32, 32, 16, 32
    tasks = [4, 7, 4]
    gen = torch.Generator().manual_seed(5)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    xa = torch.randn(n, d, h, w, ca, generator=gen).to(DEV, torch.bfloat16)
    xb = torch.randn(n, d, h, w, cb, generator=gen).to(DEV, torch.bfloat16)
    dy = torch.randn(n, d, h, w, cout, generator=gen).to(DEV, torch.bfloat16)
    default = ops.get_wgrad_col()
    try:
        ops.set_wgrad_col(0)
        ref = ops.conv5_wgrad(torch.cat((xa, xb), -1).contiguous(), dy, plan, cout)
        ops.set_wgrad_col(2)
        dw = torch.zeros((plan.nslots, 125, cout, ca + cb), device=DEV)
        stream = torch.cuda.current_stream().cuda_stream
        for part, off in ((xa, 0), (xb, ca)):
            _lib.call('repmode_conv5_wgrad_part', part.data_ptr(), dy.data_ptr(), plan.sample_slot.data_ptr(), plan.nslots,
                      dw.data_ptr(), n, d, h, w, part.shape[-1], ca + cb, off, cout, _lib.BF16, 8, stream)
    finally:
        ops.set_wgrad_col(default)
    assert rel_err(dw.cpu(), ref.cpu()) < 1e-4


BN_FUSED_CASES = [
    # (shape [N, D, H, W, C], in dtype, out dtype): tensors the one-launch pass takes (C % 8 == 0, rows per thread within the
    # budget: 16 for bf16 / bf16, 4 otherwise)
    ((8, 16, 32, 32, 64), torch.bfloat16, torch.bfloat16),     # a level-1 tensor at the benchmarked batch: 16 rows per thread, 256 workgroups
    ((8, 8, 16, 16, 128), torch.bfloat16, torch.bfloat16),     # level 2
    ((8, 2, 4, 4, 512), torch.bfloat16, torch.bfloat16),       # the bottleneck: 64 workgroups of one row per thread
    ((3, 5, 7, 9, 40), torch.bfloat16, torch.bfloat16),        # a row count that is no multiple of anything
    ((2, 4, 8, 8, 32), torch.float32, torch.float32),
    ((2, 8, 16, 16, 64), torch.float32, torch.bfloat16),
    ((1, 1, 1, 3, 8), torch.bfloat16, torch.float32),          # fewer rows than a workgroup has row slots
]


@pytest.mark.parametrize('shape,in_dtype,out_dtype', BN_FUSED_CASES)
def test_batchnorm_pass_as_one_launch(shape, in_dtype, out_dtype):
    """BatchNorm3d + ReLU (RepMode.py:146-149, 212; :80-84; :97-101), training mode, forward and backward as ONE launch each
    (grid-wide barrier, the tensor held in registers) against torch's own BatchNorm3d on the CPU and against the two-launch
    passes -- three times in a row on one stream: the barrier's counters must come back to zero."""
    ops = _ops()
    import copy
    from test_hip_parity import nrm_err
    c = shape[-1]
    gen = torch.Generator().manual_seed(c + shape[1])
    x = (torch.randn(*shape, generator=gen) * 1.5 + 0.3).to(in_dtype).float()
    r = torch.randn(*shape, generator=gen).to(out_dtype).float()
    bn = torch.nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=gen)
        bn.bias.uniform_(-0.5, 0.5, generator=gen)
    bn.train()
    xr = x.clone().requires_grad_(True)
    ref = copy.deepcopy(bn)
    yr = torch.relu(ref(xr.permute(0, 4, 1, 2, 3))).permute(0, 2, 3, 4, 1)
    (yr * r).sum().backward()
    res = {}
    default = ops.get_bn_fused()
    try:
        for fused in (1, 0):
            ops.set_bn_fused(fused)
            for rep in range(3 if fused else 1):
                bd = copy.deepcopy(bn).to(DEV)
                xd = x.to(DEV, in_dtype).requires_grad_(True)
                y = ops.bn_relu(xd, bd, True, out_dtype)
                (y.float() * r.to(DEV)).sum().backward()
                torch.cuda.synchronize()
                out = (y.detach().float().cpu(), xd.grad.float().cpu(), bd.weight.grad.cpu(), bd.bias.grad.cpu(),
                       bd.running_mean.cpu(), bd.running_var.cpu())
                if rep:       # (a stuck or half-reset barrier would show as a hang or as stale statistics)
                    assert rel_err(out[4], res[fused][4]) < 1e-5 and rel_err(out[5], res[fused][5]) < 1e-5
                res[fused] = out
    finally:
        ops.set_bn_fused(default)
    y, dx, dg, db, rm, rv = res[1]
    tol = 1e-4 if out_dtype == torch.float32 else 1e-2
    assert rel_err(y, yr.detach()) < tol
    err = rel_err if in_dtype == torch.float32 else nrm_err
    gtol = 1e-3 if in_dtype == torch.float32 and out_dtype == torch.float32 else 3e-2
    assert err(dx, xr.grad) < gtol
    assert err(dg, ref.weight.grad) < gtol and err(db, ref.bias.grad) < gtol
    # the two-launch passes: the same sums in the same slices, another order of the float atomics
    # (bf16: a last-bit difference of a mean moves a few outputs by one bf16 step and flips a few ReLU masks)
    low = out_dtype == torch.bfloat16 or in_dtype == torch.bfloat16
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        e = (nrm_err if low and 1 <= i <= 3 else rel_err)(a, b)
        assert e < (2e-2 if low else 1e-5), (i, e)
    assert rel_err(rm, res[0][4]) < 1e-5 and rel_err(rv, res[0][5]) < 1e-5


def test_captured_operands_outlive_the_store():
    """a captured train step reads and writes the per-expert blocks' stored operands BY ADDRESS.  When the store
    drops its entries (another optimizer steps, another Model is built, a launch fails) the buffers must stay alive for the
    replays: the Model holds them beside its graph.  Here the store is emptied after the capture and the freed memory is handed
    to tensors full of NaN; the replays must go on producing the losses of a model that steps launch by launch."""
    from conftest import Opts
    from repmode_amd.model import Model
    ops = _ops()
    gen = torch.Generator().manual_seed(4)
    n, shape = 4, (16, 32, 32)
    tasks = torch.tensor([2, 6, 9, 2])
    xs = [torch.randn(n, 1, *shape, generator=gen) for _ in range(8)]
    ts = [torch.randn(n, 1, *shape, generator=gen) for _ in range(8)]
    torch.manual_seed(0)
    g = Model(Opts(), lr=1e-3, gpu_ids=0, mult_chan=8, dtype=torch.bfloat16, hip_graph=True)
    torch.manual_seed(0)
    e = Model(Opts(), lr=1e-3, gpu_ids=0, mult_chan=8, dtype=torch.bfloat16, hip_graph=False)
    e.net.load_state_dict(g.net.state_dict())
    lg, le, junk = [], [], []
    for i in range(8):
        g.do_train_iter(xs[i], ts[i], tasks)
        lg.append(float(g.last_loss))
        if i == Model.GRAPH_WARMUP:                                   # the step that captured
            st = next(iter(g._graphs.values()))
            assert st['graph'] is not None and len(st['operands']) > 0
            held = sum(t.numel() * t.element_size() for t in st['operands'])
            ops.torch_ops().clear_frag_store()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            # whatever the allocator would have re-used: poison it
            junk = [torch.full((max(1, held // 8),), float('nan'), device=DEV) for _ in range(4)]
        e.do_train_iter(xs[i], ts[i], tasks)
        le.append(float(e.last_loss))
    del junk
    record('captured_operands', graph=lg, eager=le)
    assert all(l == l for l in lg), lg
    for a, b in zip(lg, le):
        assert abs(a - b) < 2e-2 * abs(b), (lg, le)


WGRAD_DUAL_CASES = [
    # (N, (D, H, W), Cin, Cout): both conv experts' filter gradients of a per-expert block from one column-walk launch
    (8, (4, 8, 8), 64, 64),        # level 3's volume at the benchmarked batch: 2 samples per step, 4 groups
    (8, (2, 4, 4), 64, 128),       # level 4's: 4 samples per step, the 4-slot ring
    (5, (2, 4, 4), 32, 32),        # an odd sample count: a half-empty last group
    (3, (3, 6, 7), 24, 40),        # ragged in every direction, half-filled 16-channel tiles
    (1, (1, 2, 2), 16, 16),        # one sample, one plane
    (2, (4, 8, 12), 16, 32),       # two columns per plane
    (7, (5, 3, 4), 32, 16),        # a low volume with more than two planes: the 8 x 8 tile half empty
    (2, (4, 8, 8), 128, 256),      # a real layer's width: every workgroup several units
]


@pytest.mark.parametrize('case', WGRAD_DUAL_CASES)
def test_conv5_wgrad_column_form_dual_experts_vs_oracle(case):
    """repmode_conv5_wgrad_dual with the experts' layouts through the column walk (mode 2): dk5[co][ci][125] from the 5x5x5
    expert's gate-scaled output gradient and dk3[co][ci][27] from the 3x3x3 expert's -- autograd of RepMode.py:171-175, 204-208
    by linearity -- against torch's own filter gradients of the two convolutions on the same bf16-rounded operands, and against
    conv5_wgrad.hip's dual launch (mode 0)."""
    import torch.nn.functional as F
    from repmode_amd import _lib
    ops = _ops()
    n, shape, cin, cout = case
    d, h, w = shape
    gen = torch.Generator().manual_seed(n * 100 + cin + cout + w)
    x = torch.randn(n, cin, *shape, generator=gen).bfloat16().float()
    dya = torch.randn(n, cout, *shape, generator=gen).bfloat16().float()
    dyb = torch.randn(n, cout, *shape, generator=gen).bfloat16().float()
    k5 = torch.zeros(cout, cin, 5, 5, 5, requires_grad=True)
    k3 = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    ((F.conv3d(x, k5, padding=2) * dya).sum() + (F.conv3d(x, k3, padding=1) * dyb).sum()).backward()
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    x_cl, a_cl, b_cl = cl(x), cl(dya), cl(dyb)
    got = []
    default = ops.get_wgrad_col()
    try:
        for mode in (2, 0):
            ops.set_wgrad_col(mode)
            d5 = torch.full((cout, cin, 5, 5, 5), float('nan'), device=DEV)
            d3 = torch.full((cout, cin, 3, 3, 3), float('nan'), device=DEV)
            _lib.call('repmode_conv5_wgrad_dual', x_cl.data_ptr(), a_cl.data_ptr(), b_cl.data_ptr(), d5.data_ptr(), d3.data_ptr(),
                      n, d, h, w, cin, cout, 2, 3, torch.cuda.current_stream().cuda_stream)
            got.append((d5.cpu(), d3.cpu()))
    finally:
        ops.set_wgrad_col(default)
    e5, e3 = rel_err(got[0][0], k5.grad), rel_err(got[0][1], k3.grad)
    record('wgrad_col_dual', case=[n, list(shape), cin, cout], dk5=e5, dk3=e3)
    assert torch.isfinite(got[0][0]).all() and torch.isfinite(got[0][1]).all()
    assert e5 < TOL_BF16_ACC and e3 < TOL_BF16_ACC
    assert rel_err(got[0][0], got[1][0]) < 1e-4 and rel_err(got[0][1], got[1][1]) < 1e-4


@pytest.mark.parametrize('case', [
    # (N, D, H, W, Cin, Cout, tasks): the one-channel ends' filter gradient on even widths -- the software-pipelined path
    (2, 4, 8, 32, 32, 1, [3, 3]),
    (3, 6, 9, 40, 1, 32, [1, 4, 1]),          # first layer, ragged in z / y, a width that is no multiple of the tile
    (3, 5, 12, 64, 24, 1, [7, 7, 2]),         # last layer, 24 channels (one channel tile, 8 dead rows)
    (2, 3, 8, 34, 1, 16, [0, 5]),
    (4, 2, 16, 64, 1, 32, [2, 2, 2, 9]),      # a slot of three samples
])
def test_conv5_wgrad_thin_pipelined_vs_oracle(case):
    """conv5_wgrad_thin (the 125 taps take the place of the missing channel dimension): the software-pipelined tile loop with
    the single-channel halo fetched as dwords, against autograd of the per-sample convolution (RepMode.py:207)."""
    ops = _ops()
    n, d, h, w, cin, cout, tasks = case
    gen = torch.Generator().manual_seed(sum(case[:6]) + 1)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(plan.nslots, cout, cin, 5, 5, 5, requires_grad=True)
    slots = torch.tensor([plan.slot_task_host.index(t) for t in tasks])
    (orc.conv_per_sample(x, wt[slots]) * dy).sum().backward()
    dw_ref = wt.grad.reshape(plan.nslots, cout, cin, 125).permute(0, 3, 1, 2)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    got = ops.conv5_wgrad(x_cl, dy_cl, plan, cout).cpu()
    e = rel_err(got, dw_ref)
    record('wgrad_thin', case=list(case[:6]), err=e)
    assert e < TOL_BF16_ACC


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_stride2_operands_of_several_stages_from_one_launch(dtype):
    """repmode_k2_frags_multi: the fragment-major operands (both roles) of several stride-2 stages' 2x2x2 filters (RepMode.py:81
    Conv3d, :98 ConvTranspose3d) from ONE launch are, element for element, what one repmode_k2_frags2 launch per filter lays
    out -- ragged channel counts included."""
    from repmode_amd import _lib
    ops = _ops()
    gen = torch.Generator().manual_seed(12)
    shapes = [(32, 32, False), (64, 64, False), (40, 24, False), (256, 512, True), (24, 40, True), (8, 8, True)]   # (rows = Co, red = Ci, up)
    ws = [(torch.randn(ci, co, 2, 2, 2, generator=gen) if up else torch.randn(co, ci, 2, 2, 2, generator=gen)).to(DEV)
          for co, ci, up in shapes]
    want = [ops.k2_weight_frags(w, co, ci, up, dtype, both=True) for w, (co, ci, up) in zip(ws, shapes)]
    outs = [(torch.full_like(a, float('nan')), torch.full_like(b, float('nan'))) for a, b in want]
    n = len(ws)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    _lib.call('repmode_k2_frags_multi', n, P(*[w.data_ptr() for w in ws]), I(*[s[0] for s in shapes]), I(*[s[1] for s in shapes]),
              I(*[int(s[2]) for s in shapes]), _lib.BF16 if dtype == torch.bfloat16 else _lib.F32,
              P(*[o[0].data_ptr() for o in outs]), P(*[o[1].data_ptr() for o in outs]), torch.cuda.current_stream().cuda_stream)
    for (a, b), (x, y) in zip(want, outs):
        assert torch.equal(a.float().cpu(), x.float().cpu()) and torch.equal(b.float().cpu(), y.float().cpu())


def test_capturable_adam_keeps_its_device_buffers_across_load_state_dict():
    """a captured step holds the optimizer's device step count and constants BY ADDRESS; loading a state_dict
    into the optimizer must keep those buffers and overwrite the count in place -- and the next update must be the one
    torch.optim.Adam makes from the same state."""
    from repmode_amd.optim import Adam
    _ops()
    gen = torch.Generator().manual_seed(8)
    shapes = [(64, 32, 5, 5, 5), (64,), (17, 3)]
    ps = [torch.randn(*s, generator=gen).to(DEV).requires_grad_(True) for s in shapes]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    ours, stock = Adam(ps, lr=1e-2, capturable=True), torch.optim.Adam(qs, lr=1e-2)
    for step in range(3):
        gs = [torch.randn(*s, generator=gen).to(DEV) for s in shapes]
        for p, q, g in zip(ps, qs, gs):
            p.grad, q.grad = g.clone(), g.clone()
        ours.step()
        stock.step()
    ptr_step, ptr_hyper = ours._step_dev.data_ptr(), ours._hyper_dev.data_ptr()
    assert int(ours._step_dev.item()) == 3
    for _ in range(2):                                   # the stock optimizer goes on alone: its state is at step 5
        for q in qs:
            q.grad = torch.randn(q.shape, generator=gen).to(DEV)
        stock.step()
    with torch.no_grad():
        for p, q in zip(ps, qs):
            p.copy_(q)
    import copy
    ours.load_state_dict(copy.deepcopy(stock.state_dict()))      # (load_state_dict keeps tensors that need no cast: no shared moments)
    assert ours._step_dev.data_ptr() == ptr_step and ours._hyper_dev.data_ptr() == ptr_hyper
    assert int(ours._step_dev.item()) == 5
    gs = [torch.randn(*s, generator=gen).to(DEV) for s in shapes]
    for p, q, g in zip(ps, qs, gs):
        p.grad, q.grad = g.clone(), g.clone()
    ours.step()
    stock.step()
    assert int(ours._step_dev.item()) == 6
    for p, q in zip(ps, qs):
        assert rel_err(p.detach().cpu(), q.detach().cpu()) < 1e-6



def test_conv5_wgrad_forms_agree_on_random_shapes():
    """The filter gradient's forms against each other on 48 seeded random shapes (ragged volumes 16 .. 70 voxels wide, 1 .. 12
    planes, channel counts off the 32-channel tiles, 1 .. 8 slots with uneven sample counts): stream-K with the cross-tile
    window pipeline (mode 3), the wave-specialised grid (2) and the column walk (col mode 2, where eligible) against the regular
    grid (0) -- same products, another split of the voxel sums (each form is checked against the oracle on fixed cases; this
    one looks for shapes a form mishandles).  tools/wgrad_fuzz.py is the same loop for any number of cases."""
    import random
    ops = _ops()
    rng = random.Random(20261001)
    ws, col = ops.get_wgrad_ws(), ops.get_wgrad_col()
    worst = 0.0
    try:
        for it in range(48):
            n = rng.choice([1, 2, 3, 5, 8, 11])
            d, h = rng.choice([1, 2, 3, 4, 7, 12]), rng.choice([3, 4, 8, 9, 16, 21])
            w = rng.choice([16, 17, 24, 32, 33, 40, 64, 70])
            cin, cout = rng.choice([8, 16, 24, 32, 40, 64, 96]), rng.choice([8, 16, 32, 48, 64, 72])
            ntask = rng.choice([1, 2, 3, n])
            tasks = [rng.randrange(12) % ntask for _ in range(n)]
            plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
            g = torch.Generator(device=DEV).manual_seed(it)
            x = torch.randn(n, d, h, w, cin, device=DEV, generator=g).bfloat16()
            dy = torch.randn(n, d, h, w, cout, device=DEV, generator=g).bfloat16()
            ops.set_wgrad_col(0)
            ops.set_wgrad_ws(0)
            ref = ops.conv5_wgrad(x, dy, plan, cout).float()
            scale = ref.abs().max().item() + 1e-30
            for ws_mode, col_mode in ((3, 0), (2, 0), (3, 2)):
                ops.set_wgrad_ws(ws_mode)
                ops.set_wgrad_col(col_mode)
                got = ops.conv5_wgrad(x, dy, plan, cout).float()
                e = (got - ref).abs().max().item() / scale
                worst = max(worst, e)
                assert e < 1e-4, (it, (n, d, h, w, cin, cout, tasks), ws_mode, col_mode, e)
    finally:
        ops.set_wgrad_ws(ws)
        ops.set_wgrad_col(col)
    record('wgrad_forms_random', cases=48, worst=worst)


@pytest.mark.parametrize('case', [
    # (N, D, H, W, Cin, Cout, tasks): stream-K ranges of one, two and three tile steps on the 8 x 32 tile (the cross-tile window
    # pipeline's prologue / last-tile paths: a workgroup whose first tile is also its last), and a unit boundary in mid-range
    (1, 1, 8, 32, 32, 32, [2]),              # one sample, one plane: only dz = 2 has a step -- ONE step in the launch
    (1, 2, 8, 32, 32, 32, [2]),              # dz = 1, 2, 3: four steps
    (1, 1, 16, 64, 8, 8, [7]),               # four tiles of one plane, half-filled channel tiles
    (2, 1, 8, 32, 40, 72, [3, 5]),           # two slots, 2 x 3 channel tiles: every workgroup crosses unit boundaries
    (3, 3, 9, 33, 32, 32, [1, 1, 1]),        # ragged tiles whose second x tile holds one voxel column
])
def test_conv5_wgrad_stream_k_short_ranges_vs_oracle(case):
    """The stream-K filter gradient with the window pipeline running across tiles (csrc/conv5_wgrad.hip, SK_PIPE) where a
    workgroup's range is a step or two: the prologue's tile is the last one, the in-tile barrier "for the next tile" has no
    next tile -- against the oracle (autograd of the per-sample convolution, RepMode.py:207) and the regular grid."""
    ops = _ops()
    n, d, h, w, cin, cout, tasks = case
    gen = torch.Generator().manual_seed(sum(case[:6]) + 31)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    dy = torch.randn(n, cout, d, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(plan.nslots, cout, cin, 5, 5, 5, requires_grad=True)
    slots = torch.tensor([plan.slot_task_host.index(t) for t in tasks])
    (orc.conv_per_sample(x, wt[slots]) * dy).sum().backward()
    dw_ref = wt.grad.reshape(plan.nslots, cout, cin, 125).permute(0, 3, 1, 2)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    got = []
    ws, col = ops.get_wgrad_ws(), ops.get_wgrad_col()
    try:
        ops.set_wgrad_col(0)
        for mode in (3, 0):
            ops.set_wgrad_ws(mode)
            got.append(ops.conv5_wgrad(x_cl, dy_cl, plan, cout).cpu())
    finally:
        ops.set_wgrad_ws(ws)
        ops.set_wgrad_col(col)
    e = rel_err(got[0], dw_ref)
    record('wgrad_sk_short', case=list(case[:6]), err=e)
    assert e < TOL_BF16_ACC
    assert rel_err(got[0], got[1]) < 1e-4


def test_conv5_forms_agree_on_random_shapes():
    """The forward / data-gradient convolution's forms against each other on 40 seeded random shapes (ragged volumes 8 .. 70
    voxels wide, channel counts off the 16- / 32-channel tiles, uneven slots): the wave-specialised pipeline with its default
    switches (conv_pipe 57: 32- and 16-voxel bricks, row-stationary taps on 32-channel layers, forced onto grids smaller than the
    chip) against the two-workgroup kernel (conv_pipe 0) -- same products; the row-stationary order sums the 125 taps in
    another order (float rounding, then one bf16 rounding of the output: 2e-2 of the tensor's max at bf16's 8 bits is loose;
    measured ~4e-3)."""
    import random
    ops = _ops()
    rng = random.Random(20261002)
    pipe = ops.get_conv_pipe()
    worst = 0.0
    try:
        for it in range(40):
            n = rng.choice([1, 2, 3, 5, 8])
            d, h = rng.choice([1, 2, 4, 5, 9]), rng.choice([3, 4, 8, 9, 16])
            w = rng.choice([8, 15, 16, 17, 24, 32, 33, 40, 64, 70])
            cin, cout = rng.choice([8, 16, 24, 32, 40, 64]), rng.choice([8, 16, 32, 48, 64])
            ntask = rng.choice([1, 2, n])
            tasks = [rng.randrange(12) % ntask for _ in range(n)]
            plan = ops.TaskPlan(torch.tensor(tasks), 12, DEV, training=True)
            g = torch.Generator(device=DEV).manual_seed(1000 + it)
            x = torch.randn(n, d, h, w, cin, device=DEV, generator=g).bfloat16()
            experts = [torch.randn(cout, cin, k, k, k, device=DEV, generator=g) * 0.1 for k in (5, 3, 1, 1, 1)]
            gate = torch.softmax(torch.randn(plan.nslots, 5, cout, device=DEV, generator=g), dim=1)
            wf, _ = ops.gatrep_merge(*experts, gate, torch.bfloat16)
            ops.set_conv_pipe(0)
            ref = ops.conv5(x, wf, plan.sample_slot, cout).float()
            ops.set_conv_pipe(57 | 4)
            got = ops.conv5(x, wf, plan.sample_slot, cout).float()
            e = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
            worst = max(worst, e)
            assert e < 2e-2, (it, (n, d, h, w, cin, cout, tasks), e)
    finally:
        ops.set_conv_pipe(pipe)
    record('conv5_forms_random', cases=40, worst=worst)


def test_bn_relu_one_launch_agrees_with_two_launches_on_random_shapes():
    """BatchNorm3d + ReLU, training mode, forward and backward: the one-launch passes (grid-wide barrier; set_bn_fused(1))
    against the two-launch passes (0) on 30 seeded random shapes -- tensors from a few rows to the largest the one-launch form
    takes, 8 .. 512 channels, bf16 and float: outputs, running statistics and all three gradients within float summation
    order (1e-5 of the tensor's max; bf16 outputs: one rounding step)."""
    import copy
    import random
    ops = _ops()
    rng = random.Random(20261003)
    fused = ops.get_bn_fused()
    worst = 0.0
    try:
        for it in range(30):
            c = rng.choice([8, 16, 32, 64, 128, 256, 512])
            n, d, h, w = rng.choice([1, 2, 8]), rng.choice([1, 2, 4, 8, 16]), rng.choice([2, 4, 8, 16, 32]), rng.choice([4, 8, 16, 32])
            dtype = rng.choice([torch.bfloat16, torch.float32])
            g = torch.Generator(device=DEV).manual_seed(2000 + it)
            x = (torch.randn(n, d, h, w, c, device=DEV, generator=g) * 1.5 + 0.3).to(dtype)
            dy = torch.randn(n, d, h, w, c, device=DEV, generator=g).to(dtype)
            res = []
            for mode in (1, 0):
                ops.set_bn_fused(mode)
                bn = torch.nn.BatchNorm3d(c).to(DEV)
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.2, 0.2, c))
                bn.train()
                xi = x.clone().requires_grad_(True)
                y = ops.bn_relu(xi, bn)
                y.backward(dy)
                res.append([y.float(), xi.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()])
            tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
            for a, b in zip(*res):
                e = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
                worst = max(worst, e if dtype == torch.float32 else 0.0)
                assert e < tol, (it, (n, d, h, w, c, dtype), e)
    finally:
        ops.set_bn_fused(fused)
    record('bn_fused_random', cases=30, worst_f32=worst)


def test_per_expert_block_agrees_with_round4_launches_on_random_shapes():
    """A per-expert MoDE block through the operator library -- one launch per direction (deep_mode 15), the dual filter-gradient
    launch in its wave-specialised form with the lean loader on two-plane tiles -- against round 4's launches (deep_mode 0) on 14
    seeded random shapes: ragged volumes up to 6 x 10 x 12, 8 .. 128 channels, 1 .. 6 samples of mixed tasks.  Output, data
    gradient and all parameter gradients within 1e-2 (bf16: a summation-order difference can flip a rounding), as
    test_deep_mode_in_the_operator's fixed cases (which also pin the form to the oracle)."""
    import random
    from test_hip_round5 import _rand_experts
    ops = _ops()
    rng = random.Random(20261004)
    before = ops.get_deep_mode()
    worst = 0.0
    try:
        for it in range(14):
            ci, co = 8 * rng.randint(1, 16), 8 * rng.randint(1, 16)
            shape = (rng.choice([1, 2, 3, 4, 6]), rng.choice([2, 4, 5, 8, 10]), rng.choice([4, 7, 8, 12]))
            n = rng.randint(1, 6)
            gen = torch.Generator().manual_seed(3000 + it)
            ps = _rand_experts(co, ci, gen)
            tasks = [rng.randrange(12) for _ in range(n)]
            x = torch.randn(n, *shape, ci, generator=gen).bfloat16()
            r = torch.randn(n, *shape, co, generator=gen)
            res = []
            for mask in (15, 0):
                ops.set_deep_mode(mask)
                dev = [p.to(DEV).requires_grad_(True) for p in ps]
                xd = x.to(DEV).requires_grad_(True)
                plan = ops.TaskPlan(tasks, 12, DEV, training=True)
                y = ops.mode_conv3d(xd, *dev, plan, mode='unmerged')
                (y.float() * r.to(DEV)).sum().backward()
                res.append([y.detach().float().cpu(), xd.grad.float().cpu()] + [p.grad.cpu() for p in dev])
            for a, b in zip(*res):
                e = rel_err(a, b)
                worst = max(worst, e)
                assert e < 1e-2, (it, (ci, co, shape, n, tasks), e)
    finally:
        ops.set_deep_mode(before)
    record('per_expert_block_random', cases=14, worst=worst)
