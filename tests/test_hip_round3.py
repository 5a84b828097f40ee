"""Round-3 kernels against the CPU oracle (through the C ABI): the deep levels' uniform-grid convolution pair
(csrc/conv5_deep.hip), the row-stationary tap loop and 16-byte stores of conv5_igemm, the one-channel layers' own kernels.
Needs a real MI355X: every test is marked ``gpu``.  Tolerances as tests/test_hip_parity.py states them."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import repmode_oracle as orc
from test_hip_parity import DEV, TOL_BF16_ACC, _kc, _layout_wf, _ops, _rand_experts

pytestmark = pytest.mark.gpu


DEEP_CASES = [
    # (n, (d, h, w), ci, co)
    (8, (2, 4, 4), 64, 128),      # level-4 tile: four samples per workgroup, two groups
    (5, (1, 2, 2), 32, 32),       # smaller than the level-4 brick, a ragged last group
    (24, (2, 4, 4), 32, 64),      # batch 24: six groups (forward) / twelve (data gradient)
    (3, (4, 8, 8), 48, 96),       # level-3 tile, channel counts that are no tile multiples
    (2, (4, 8, 8), 128, 256),     # level-3 tile at a real layer's width (8 input-channel chunks, split 8 ways)
    (3, (2, 4, 8), 16, 40),       # level-3 tile on a volume smaller than its brick
    (2, (6, 10, 8), 16, 32),      # several bricks per sample (no skipped taps: the zero halo does it)
]


@pytest.mark.parametrize('case', DEEP_CASES)
def test_conv5_deep_vs_oracle(case):
    """Forward form (P5 = conv(x, K5), P3 = conv(x, pad K3)) and data-gradient form (conv(G5, flip K5^T) + conv(G3, flip
    pad(K3)^T)) of repmode_conv5_deep against the oracle's F.conv3d on bf16-rounded operands (float accumulation on both
    sides: 1e-4)."""
    ops = _ops()
    n, shape, ci, co = case
    gen = torch.Generator().manual_seed(n * 1000 + ci + co + shape[2])
    k5, k3 = _rand_experts(co, ci, gen)[:2]
    k5r, k3r = k5.bfloat16().float(), k3.bfloat16().float()
    x = torch.randn(n, ci, *shape, generator=gen).bfloat16().float()
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    assert ops.conv5_deep_supported(x_cl)
    wf, wd = ops.expert_frags(k5.to(DEV), k3.to(DEV), torch.bfloat16, want_wd=True)
    # ---- forward form, output cleared by the call and handed over pre-zeroed
    p5_ref = F.conv3d(x, k5r, padding=2)
    p3_ref = F.conv3d(x, k3r, padding=1)
    for zeroed in (False, True):
        out = torch.zeros(2 * n, *shape, co, device=DEV) if zeroed else torch.full((2 * n, *shape, co), 7.0, device=DEV)
        y = ops.conv5_deep(x_cl, wf, co, two_in=False, out=out, zeroed=zeroed)
        y = y.permute(0, 4, 1, 2, 3).cpu()
        assert rel_err(y[:n], p5_ref) < TOL_BF16_ACC, '5x5x5 expert'
        assert rel_err(y[n:], p3_ref) < TOL_BF16_ACC, '3x3x3 expert'
    # ---- data-gradient form: dx = d/dx [<conv(x, K5), G5> + <conv(x, pad K3), G3>]
    g = torch.randn(2 * n, co, *shape, generator=gen).bfloat16().float()
    dx_ref = F.conv_transpose3d(g[:n], k5r, padding=2) + F.conv_transpose3d(g[n:], k3r, padding=1)
    g_cl = g.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    assert ops.conv5_deep_supported(g_cl)
    dx = ops.conv5_deep(g_cl, wd, ci, two_in=True)
    assert rel_err(dx.permute(0, 4, 1, 2, 3).cpu(), dx_ref) < TOL_BF16_ACC, 'data gradient'


@pytest.mark.parametrize('ci,co,shape,n', [(64, 64, (2, 4, 4), 8), (48, 96, (4, 8, 8), 3), (256, 128, (4, 8, 8), 8)])
def test_deep_conv_in_the_operator(ci, co, shape, n):
    """The per-expert block through the operator library with the deep kernel (default) and with the general kernel's
    dual-expert launch: output, data gradient and every parameter gradient agree (bf16: an atomics-order difference can
    flip a rounding), and the deep form matches the oracle's block."""
    ops = _ops()
    gen = torch.Generator().manual_seed(ci + co + n)
    ps = _rand_experts(co, ci, gen)
    tasks = [(5 * i + 2) % 12 for i in range(n)]
    x = torch.randn(n, *shape, ci, generator=gen).bfloat16()
    r = torch.randn(n, *shape, co, generator=gen)
    res = []
    try:
        ops.set_deep_fwd_min(0)                # (the forward form on every shape, not only where it is the faster one)
        for deep in (True, False):
            ops.set_deep_conv(deep)
            dev = [p.to(DEV).requires_grad_(True) for p in ps]
            xd = x.to(DEV).requires_grad_(True)
            plan = ops.TaskPlan(tasks, 12, DEV, training=True)
            y = ops.mode_conv3d(xd, *dev, plan, mode='unmerged')
            (y.float() * r.to(DEV)).sum().backward()
            res.append([y.detach().float().cpu(), xd.grad.float().cpu()] + [p.grad.cpu() for p in dev])
    finally:
        ops.set_deep_conv(True)
        ops.set_deep_fwd_min(6000)
    for a, b in zip(res[0], res[1]):
        assert rel_err(a, b) < 1e-2
    # the oracle's block on the bf16-rounded operands
    xr = x.float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) for p in ps]
    y_ref = orc.mode_conv_pre_bn(xr, *pr, torch.tensor(tasks), training=True)
    (y_ref * r.permute(0, 4, 1, 2, 3)).sum().backward()
    assert rel_err(res[0][0].permute(0, 4, 1, 2, 3), y_ref.detach()) < 2e-2
    assert rel_err(res[0][1].permute(0, 4, 1, 2, 3), xr.grad) < 2e-2
    for got, p in zip(res[0][2:], pr):
        assert rel_err(got, p.grad) < 2e-2


ROWSTAT_CASES = [
    # (N, D, H, W, Cin, Cout): the 4 x 4 x 32 tile (row-stationary tap loop), ragged in every direction
    (1, 6, 10, 40, 16, 48),
    (2, 5, 9, 35, 24, 32),
    (1, 4, 4, 64, 32, 64),
    (2, 3, 7, 33, 40, 16),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', ROWSTAT_CASES)
def test_conv5_wide_volumes_vs_oracle(case, dtype):
    """conv5_igemm on volumes with x extent >= 32 (row-stationary tap loop; bf16 output through 16-byte permlane-swapped
    stores when the channel count allows) against the oracle, float and element-typed output."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, cin, cout = case
    gen = torch.Generator().manual_seed(sum(case))
    nslots = 2
    slots = torch.tensor([i % nslots for i in range(n)], dtype=torch.int32)
    x = torch.randn(n, cin, d, h, w, generator=gen).to(dtype).float()
    wt = (torch.randn(nslots, cout, cin, 5, 5, 5, generator=gen) / np.sqrt(cin * 125)).to(dtype).float()
    y_ref = orc.conv_per_sample(x, wt[slots.long()])
    code = ops.dtype_code(dtype)
    wf = _layout_wf(wt, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), _kc(dtype)).to(DEV, dtype)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, dtype)
    y = ops.conv5(x_cl, wf, slots.to(DEV), cout, out_f32=True)
    assert rel_err(y.permute(0, 4, 1, 2, 3).cpu(), y_ref) < TOL_BF16_ACC
    if dtype == torch.bfloat16:
        yb = ops.conv5(x_cl, wf, slots.to(DEV), cout, out_f32=False)
        assert yb.dtype == torch.bfloat16
        assert rel_err(yb.float().permute(0, 4, 1, 2, 3).cpu(), y_ref) < 6e-3
        # the 16-byte stores put every channel where the 8-byte ones would: the float output rounded once, up to the
        # summation order of the two kernels' operand roles
        assert rel_err(yb.float().cpu(), y.cpu()) < 5e-3


@pytest.mark.parametrize('ci,co,shape,n', [(64, 128, (8, 16, 16), 4), (24, 40, (5, 7, 19), 3), (128, 96, (4, 4, 16), 8)])
def test_conv5_merged_experiment_vs_oracle(ci, co, shape, n):
    """The A/B experiment's kernel (GatRep inside the conv, repmode_conv5_merged) computes the block's convolution: against
    the oracle's merged filter + per-sample F.conv3d on bf16-rounded inputs.  The experts enter the kernel rounded to bf16
    one by one (the shipped path rounds the MERGED filter once), hence the block tolerance 2e-2."""
    ops = _ops()
    gen = torch.Generator().manual_seed(ci + co + n)
    k5, k3, k1, a3, a5, gw, gb = _rand_experts(co, ci, gen)
    tasks = [(3 * i + 1) % 12 for i in range(n)]
    plan = ops.TaskPlan(tasks, 12, DEV)
    x = torch.randn(n, ci, *shape, generator=gen).bfloat16().float()
    g_ref = orc.gate_probs(gw, gb, torch.tensor(plan.slot_task_host), co)
    w_ref = orc.merge_filters(orc.expert_bank(k5, k3, k1, a3, a5), g_ref)
    y_ref = orc.conv_per_sample(x, w_ref[plan.sample_slot.cpu().long()])
    d = [t.to(DEV) for t in (k5, k3, k1, a3, a5, gw, gb)]
    g = ops.gate_softmax(d[5], d[6], plan, co)
    w2, _ = ops.expert_frags(d[0], d[1], torch.bfloat16, want_wd=False)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    y = ops.conv5_merged(x_cl, w2, d[2], d[3], d[4], g, plan.sample_slot, co)
    assert rel_err(y.permute(0, 4, 1, 2, 3).cpu(), y_ref) < 2e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_patch_gather_and_blend_match_the_indexing_expressions(dtype):
    """repmode_patch_gather / repmode_patch_blend against the reference's own expressions (fnet_model.py:196-217: slice +
    cat; ``pred_sum[patch] += out * gauss``, ``weight_sum[patch] += gauss`` patch by patch) on overlapping patches, LIFO
    batches as predict() forms them, 40 patches in one call (two launches): bit-identical."""
    ops = _ops()
    from repmode_amd.model import get_gaussian, patch_grid
    gen = torch.Generator().manual_seed(11)
    img, patch = (20, 50, 70), (8, 16, 32)
    vol = torch.randn(1, 1, *img, generator=gen).to(DEV)
    gauss = torch.from_numpy(get_gaussian(patch)).to(DEV)
    patches = patch_grid(img, patch)
    assert len(patches) > 40
    ps_k, ws_k = torch.zeros_like(vol), torch.zeros_like(vol)
    ps_r, ws_r = torch.zeros_like(vol), torch.zeros_like(vol)
    for bs in (3, 40, 5):
        batch = [patches.pop() for _ in range(bs)]
        starts = [s for s, _ in batch]
        crops = ops.patch_gather(vol, starts, patch)
        ref = torch.cat([vol[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] for s, e in batch], dim=0)
        assert torch.equal(crops, ref)
        out = torch.randn(bs, 1, *patch, generator=gen).to(DEV, dtype)
        ops.patch_blend(out, gauss, starts, ps_k, ws_k)
        for i, (s, e) in enumerate(batch):
            ps_r[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] += out[i:i + 1] * gauss
            ws_r[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] += gauss
        assert torch.equal(ps_k, ps_r) and torch.equal(ws_k, ws_r)
    # a patch that leaves the volume is refused
    from repmode_amd import _lib
    with pytest.raises(_lib.RepModeHipError):
        ops.patch_gather(vol, [(15, 0, 0)], patch)


PIPE_CASES = [
    # (N, D, H, W, Cin, Cout, Cin1, Cout1): volumes 32 or more voxels wide; ragged bricks, several channel chunks, one and two
    # channel sub-tiles per wave, a two-tensor input (skip connection) and a two-tensor output (its data gradient)
    (2, 6, 10, 40, 32, 32, 0, 0),
    (3, 5, 9, 35, 48, 64, 0, 0),
    (2, 8, 8, 64, 64, 96, 0, 0),
    (2, 4, 12, 33, 64, 32, 32, 0),
    (2, 7, 5, 32, 32, 64, 0, 32),
    (9, 4, 4, 32, 16, 24, 0, 0),
    (70, 4, 4, 32, 16, 32, 0, 0),       # more than 64 samples: slots beyond the lane-read register
    (2, 4, 8, 32, 24, 32, 0, 0),        # a half-empty last channel chunk (24 = 16 + 8)
    (2, 4, 8, 32, 32, 48, 0, 0),        # two channel sub-tiles per wave, the second one half used
]


@pytest.mark.parametrize('mode', [5, 7, 13, 15, 29, 61])
@pytest.mark.parametrize('case', PIPE_CASES)
def test_conv5_pipelined_form_vs_oracle_and_the_two_workgroup_form(case, mode):
    """conv5_pipe_kernel (one workgroup per CU, double-buffered halo image, persistent item ranges; mode 5: two channel sub-tiles
    per wave where the layer has them, 7: one) and conv5_ws_kernel (13, 15: the same with loader waves) against the oracle and -- same products in the same order -- BIT-identical to
    the two-workgroup kernel.  Small grids: several items per workgroup only where the grid exceeds the CU count, so the
    batch-9 case and the train-step tests cover the persistent walk."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, cin, cout, cin1, cout1 = case
    gen = torch.Generator().manual_seed(sum(case) + mode)
    nslots = 3
    slots = torch.tensor([i % nslots for i in range(n)], dtype=torch.int32)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    wt = (torch.randn(nslots, cout, cin, 5, 5, 5, generator=gen) / np.sqrt(cin * 125)).bfloat16().float()
    y_ref = orc.conv_per_sample(x, wt[slots.long()])
    code = ops.dtype_code(torch.bfloat16)
    wf = _layout_wf(wt, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), _kc(torch.bfloat16)).to(DEV, torch.bfloat16)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    outs = []
    default = ops.get_conv_pipe()
    try:
        for m in (mode, 0):
            ops.set_conv_pipe(m)
            if cin1 or cout1:
                xa = x_cl[..., :cin1].contiguous() if cin1 else x_cl
                xb = x_cl[..., cin1:].contiguous() if cin1 else None
                ya, yb = ops.conv5_pair(xa, xb, wf, slots.to(DEV), cout, cout1)
                y = torch.cat([ya, yb], dim=-1) if yb is not None else ya
            else:
                y = ops.conv5(x_cl, wf, slots.to(DEV), cout)
            outs.append(y.float().cpu())
    finally:
        ops.set_conv_pipe(default)
    assert rel_err(outs[0].permute(0, 4, 1, 2, 3), y_ref) < 6e-3          # (bf16 output)
    if mode & 32:       # row-stationary tap order (32-channel layers): another summation order of the 125 taps
        assert rel_err(outs[0], outs[1]) < 8e-3
    else:
        assert torch.equal(outs[0], outs[1])


PIPE16_CASES = [
    # (N, D, H, W, Cin, Cout, Cin1, Cout1): volumes 16..31 voxels wide (round 4: the 4 x 4 x 16 brick of conv5_ws_kernel -- level 2
    # of the network at the 32x64x64 patch); ragged bricks in every direction, one and several channel chunks, a skip
    # connection's two-tensor input and its data gradient's two-tensor output, more items than workgroups
    (2, 8, 16, 16, 64, 128, 0, 0),      # the network's level-2 shape (enc3.conv1)
    (3, 5, 9, 20, 48, 64, 0, 0),
    (2, 6, 10, 24, 32, 32, 0, 0),
    (2, 4, 12, 17, 64, 32, 32, 0),
    (2, 7, 5, 16, 32, 64, 0, 32),
    (2, 4, 8, 31, 24, 48, 0, 0),        # a half-empty last channel chunk; the second channel sub-tile half used
    (40, 4, 8, 16, 16, 64, 0, 0),       # 320 (one sub-tile) / 160 items: persistent item ranges
]


@pytest.mark.parametrize('mode', [77, 79, 93])
@pytest.mark.parametrize('case', PIPE16_CASES)
def test_conv5_wave_specialised_16_voxel_bricks_vs_oracle_and_the_two_workgroup_form(case, mode):
    """conv5_ws_kernel's 16-voxel-brick form (bit 6 of the switch; 77: by price one or two channel sub-tiles per wave, 79: one,
    93: items along z first) against the oracle and BIT-identical to the two-workgroup kernel's unsplit bf16-output launch
    (same products in the same order: chunk by chunk, tap by tap)."""
    ops = _ops()
    from repmode_amd import _lib
    n, d, h, w, cin, cout, cin1, cout1 = case
    gen = torch.Generator().manual_seed(sum(case) + mode)
    nslots = 3
    slots = torch.tensor([i % nslots for i in range(n)], dtype=torch.int32)
    x = torch.randn(n, cin, d, h, w, generator=gen).bfloat16().float()
    wt = (torch.randn(nslots, cout, cin, 5, 5, 5, generator=gen) / np.sqrt(cin * 125)).bfloat16().float()
    y_ref = orc.conv_per_sample(x, wt[slots.long()])
    code = ops.dtype_code(torch.bfloat16)
    wf = _layout_wf(wt, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), _kc(torch.bfloat16)).to(DEV, torch.bfloat16)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)
    outs = []
    default = ops.get_conv_pipe()
    try:
        for m in (mode, 0):
            ops.set_conv_pipe(m)
            assert _lib.load().repmode_conv5_elem_out(n, d, h, w, cin, cout, code) == (1 if m else 0)
            if cin1 or cout1:
                xa = x_cl[..., :cin1].contiguous() if cin1 else x_cl
                xb = x_cl[..., cin1:].contiguous() if cin1 else None
                ya, yb = ops.conv5_pair(xa, xb, wf, slots.to(DEV), cout, cout1)
                y = torch.cat([ya, yb], dim=-1) if yb is not None else ya
            else:
                y = ops.conv5(x_cl, wf, slots.to(DEV), cout)
            outs.append(y.float().cpu())
    finally:
        ops.set_conv_pipe(default)
    assert rel_err(outs[0].permute(0, 4, 1, 2, 3), y_ref) < 6e-3          # (bf16 output)
    assert torch.equal(outs[0], outs[1])


WGRAD_WS_CASES = [
    # (N, D, H, W, Cin, Cout): volumes 32 or more voxels wide (the 1 x 8 x 32 tile), ragged in every direction
    (2, 6, 10, 40, 32, 32),
    (3, 5, 9, 35, 16, 48),
    (2, 4, 16, 64, 64, 32),
]


@pytest.mark.parametrize('case', WGRAD_WS_CASES)
def test_conv5_wgrad_wave_specialised_vs_oracle(case):
    """The filter gradient's wave-specialised form (loader waves fetch and transpose tile k + 1 while MFMA waves multiply tile
    k; forced with mode 2 -- by default only long tile loops take it) against the oracle, and against the two-workgroup form
    (same products; the voxel range is split differently over workgroups: float atomics order)."""
    ops = _ops()
    n, d, h, w, cin, cout = case
    gen = torch.Generator().manual_seed(sum(case) + 3)
    tasks = [5, 9, 5][:n]
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
    default = ops.get_wgrad_ws()
    try:
        for mode in (2, 0):
            ops.set_wgrad_ws(mode)
            got.append(ops.conv5_wgrad(x_cl, dy_cl, plan, cout).cpu())
    finally:
        ops.set_wgrad_ws(default)
    assert rel_err(got[0], dw_ref) < TOL_BF16_ACC
    assert rel_err(got[0], got[1]) < 1e-4        # (float atomics in another order: ~1e-6 measured)


WGRAD_SK_CASES = [
    # (N, D, H, W, Cin, Cout, tasks): the stream-K form (round 4) -- ragged volumes, several (co, ci) tiles, slots with one,
    # two and three samples, volumes so thin that the outer dz planes have no input plane at all (D = 2), one workgroup and many
    (2, 6, 10, 40, 32, 32, [5, 9]),
    (3, 5, 9, 35, 16, 48, [5, 9, 5]),
    (2, 4, 16, 64, 64, 32, [1, 1]),
    (6, 2, 8, 32, 40, 72, [3, 3, 7, 3, 0, 7]),
    (1, 1, 4, 33, 8, 8, [2]),
    (9, 8, 16, 32, 64, 64, [4, 4, 4, 8, 8, 1, 1, 1, 1]),
    # volumes 16..31 voxels wide (the 1 x 8 x 16 tile: level 2 of the network)
    (8, 8, 16, 16, 64, 128, [0, 1, 2, 3, 4, 5, 6, 7]),
    (2, 6, 10, 20, 32, 32, [5, 9]),
    (3, 5, 9, 17, 16, 48, [5, 9, 5]),
]


@pytest.mark.parametrize('case', WGRAD_SK_CASES)
def test_conv5_wgrad_stream_k_vs_oracle(case):
    """The filter gradient's stream-K form (mode 3, the default where eligible: persistent wave-specialised workgroups that each
    take an equal range of the launch's tile-step sequence, flushing the accumulators at unit boundaries) against the oracle and
    against the regular grid (mode 0): same products, another split of the voxel sums over workgroups (float atomics)."""
    ops = _ops()
    n, d, h, w, cin, cout, tasks = case
    gen = torch.Generator().manual_seed(sum(case[:6]) + 5)
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
    default = ops.get_wgrad_ws()
    try:
        for mode in (3, 0):
            ops.set_wgrad_ws(mode)
            got.append(ops.conv5_wgrad(x_cl, dy_cl, plan, cout).cpu())
    finally:
        ops.set_wgrad_ws(default)
    assert rel_err(got[0], dw_ref) < TOL_BF16_ACC
    assert rel_err(got[0], got[1]) < 1e-4
