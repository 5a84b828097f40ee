"""The benchmarked dtype end to end (VERDICT round 2, weak #1 / missing #3, #5): bf16 train steps against the reference's
captured scalars and the oracle's gradients, bf16 sliding-window inference against the oracle's blend, and the plug-in
called exactly the way the reference harness calls it (fnet_model.py:52, 98-113: importlib, device-tensor task,
torch.cuda.amp.autocast, GradScaler).  Needs a real MI355X."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, Opts, load_golden, record, rel_err
from oracle import repmode_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel2(a, b):
    """2-norm relative error."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cos_ratio(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30)), float(a.norm() / b.norm().clamp_min(1e-30))


@pytest.mark.timeout(1200)
def test_bf16_train_iter_losses_and_gradients():
    """bf16 ``Model.do_train_iter`` at mult_chan 32 on the G4b inputs (three 16x64x64 patches, tasks 3, 7, 3, Adam 1e-3, the
    reference's seed-0 initial state): the two losses within 2 % of what the REAL fnet_model.Model.do_train_iter logged in
    float32 (measured: 2e-5 and 1.5e-3), and the first step's parameter gradients against the oracle's float32 autograd from
    the same state.

    What can be asked of those gradients (profiles/r03_bf16_grad_diag.txt, tools/bf16_grad_diag.py: the same comparison tensor
    by tensor, next to the float32 HIP path, which agrees with the oracle to < 1e-2 everywhere): a pre-activation stored in
    bf16 flips the ReLU mask of the ~0.3 % of units nearest zero, each flip is a full-size error in one term of every sum it
    enters, and the error compounds through the 19 blocks in backward order -- conv_out 2.8e-2, decoder_block1.conv2 0.11,
    ... saturating at a cosine of ~0.74 against the float32 gradient from the bottleneck on, norm ratio 1.00 +- 0.05
    throughout.  So: the loss end of the network tightly, the whole gradient by direction and norm."""
    from repmode_amd.model import Model
    g = load_golden('g4b_model_train_iter.npz')
    tasks = torch.from_numpy(g['tasks'])
    # the oracle's gradients of step 1 (float32, CPU), from the seeded initial state
    torch.manual_seed(0)
    ref = orc.Net(Opts(), mult_chan=32)
    ref.train()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    x0, t0 = torch.from_numpy(g['xs'][0]), torch.from_numpy(g['targets'][0])
    loss_ref = torch.nn.functional.mse_loss(ref(x0, tasks), t0)
    loss_ref.backward()
    assert abs(float(loss_ref.detach()) - g['losses'][0]) < 1e-3 * abs(g['losses'][0])      # (the oracle is the fixture's arithmetic)
    torch.manual_seed(0)
    m = Model(Opts(), nn_module='RepMode', lr=float(g['lr']), gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
    refp = dict(ref.named_parameters())
    for s in range(len(g['losses'])):
        m.do_train_iter(torch.from_numpy(g['xs'][s]), torch.from_numpy(g['targets'][s]), tasks)
        loss = float(m.last_loss)
        record('bf16_train_iter', step=s, loss=loss, loss_ref=float(g['losses'][s]))
        assert abs(loss - g['losses'][s]) < 2e-2 * abs(g['losses'][s]), (s, loss, g['losses'][s])
        if s == 0:
            got = {k: p.grad.detach().float().cpu() for k, p in m.net.named_parameters()}
    near = {k: _rel2(got[k], refp[k].grad) for k in got if k.startswith('conv_out.')}
    mid = {k: _rel2(got[k], refp[k].grad) for k in got if k.startswith('decoder_block1.conv_less.conv2.')}
    cos, ratio = _cos_ratio(torch.cat([got[k].reshape(-1) for k in got]), torch.cat([refp[k].grad.reshape(-1) for k in got]))
    record('bf16_train_iter_grads', conv_out=max(near.values()), dec1_conv2=max(mid.values()), cos=cos, ratio=ratio)
    assert max(near.values()) < 5e-2, near                  # the loss end: one bf16 block (measured 2.8e-2)
    assert max(mid.values()) < 0.2, mid                     # one BatchNorm + ReLU further (measured 0.11)
    assert cos > 0.7 and abs(ratio - 1.0) < 0.05, (cos, ratio)        # the whole 123.9 M-element gradient (measured 0.764, 1.006)


def _lib_elem_out():
    """The kernel library's own answer to 'does this convolution write a bf16 tensor' -- the emulating oracle's policy."""
    from repmode_amd import _lib
    lib = _lib.load()
    return lambda n, d, h, w, cin, cout: lib.repmode_conv5_elem_out(n, d, h, w, cin, cout, _lib.BF16) != 0


# What a network-level bf16 comparison can resolve (measured, round 4): the bf16 network is
# discontinuous at float-noise scale -- a 1e-6 relative perturbation of the parameters of the EMULATING oracle itself moves a
# few values across bf16 rounding boundaries and ReLU thresholds, and the differences grow ~1.6x per block: 6e-4 after the
# first block, 4e-2 at the output, 0.53 on the whole parameter gradient (0.013 on conv_out's, 0.08 one block earlier, 0.5-0.75
# from the bottleneck back to the first layer).  Two correct bf16 implementations with different float summation orders
# therefore agree no better than that, and the HIP path against the emulation measures exactly that: output 3.5e-2, whole
# gradient 0.47, worst tensor 0.71.  So the test calibrates each tensor's bound on the emulation's own sensitivity
# (`noise`: the emulation against a copy whose parameters were perturbed by 1e-6): the HIP path must be indistinguishable
# from "another bf16 implementation", and where the noise floor is low (the loss end) a 10 % systematic error is far outside.
# Deep layers are pinned one block at a time on identical operands (test_hip_parity.py::
# test_full_size_bf16_blocks_at_bench_config: every block's output, data gradient and seven parameter gradients within 2e-2).
EMU_CASES = {
    # G4b's batch shape (two distinct tasks: every block merged) and a four-sample batch with three distinct tasks (levels 3-4
    # take the per-expert formulation: bf16 experts, gemm3's rounded operands, gate-scaled output gradients)
    'merged': dict(tasks=[3, 7, 3], shape=(16, 64, 64)),
    'per_expert': dict(tasks=[1, 5, 7, 1], shape=(16, 64, 64)),
}


def _emulated_run(state, x, tgt, tasks, perturb_seed=0):
    emu = orc.Net(Opts(), mult_chan=32, emulate=torch.bfloat16, elem_out=_lib_elem_out())
    emu.load_state_dict(state)
    if perturb_seed:
        gen = torch.Generator().manual_seed(perturb_seed)
        with torch.no_grad():
            for p in emu.parameters():
                p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=gen))
    emu.train()
    ye = emu(x, tasks)
    loss_e = torch.nn.functional.mse_loss(ye, tgt)
    loss_e.backward()
    return ye.detach(), float(loss_e.detach()), {k: p.grad.detach() for k, p in emu.named_parameters()}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('case', sorted(EMU_CASES))
def test_bf16_gradients_against_the_bf16_emulating_oracle(case):
    """Network-level bf16 parity against an independent bf16 implementation (missing 4 / weak 1): one bf16
    forward + backward of the mult_chan-32 network through the HIP path against ``orc.Net(emulate=torch.bfloat16)`` -- the CPU
    oracle rounding to bf16 exactly where the kernels do (block inputs, merged filters / per-expert operands, element-typed conv
    outputs, BatchNorm outputs, bf16 gradient tensors), float32 in between -- from the same seeded state.  Bounds are
    calibrated per tensor on the emulation's own float-noise sensitivity (see the comment above): loss 2e-3; output and whole
    gradient within 1.5x of the noise floor; every parameter gradient within 3x of its own noise floor (+ 3e-2); the loss-end
    tensors (noise floor ~1e-2) within 5e-2 outright, where a systematic 10 % error is asserted to be caught."""
    from repmode_amd.nn_modules.RepMode import Net
    cfg = EMU_CASES[case]
    tasks = torch.tensor(cfg['tasks'])
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(len(cfg['tasks']), 1, *cfg['shape'], generator=gen)
    tgt = torch.randn(len(cfg['tasks']), 1, *cfg['shape'], generator=gen)
    torch.manual_seed(0)
    net = Net(Opts(), mult_chan=32, dtype=torch.bfloat16)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    y = net(x.to(DEV), tasks)
    loss = torch.nn.functional.mse_loss(y.float(), tgt.to(DEV))
    loss.backward()
    got = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ye, loss_e, want = _emulated_run(state, x, tgt, tasks)
    yp, _, pert = _emulated_run(state, x, tgt, tasks, perturb_seed=1)
    cat = lambda d: torch.cat([d[k].reshape(-1) for k in got])
    errs = {k: _rel2(got[k], want[k]) for k in got}
    noise = {k: _rel2(pert[k], want[k]) for k in got}
    out_err, out_noise = _rel2(y.float(), ye), _rel2(yp, ye)
    whole, whole_noise = _rel2(cat(got), cat(want)), _rel2(cat(pert), cat(want))
    worst = max(errs, key=lambda k: errs[k] / (3 * noise[k] + 3e-2))
    record('bf16_emulated_grads', case=case, loss=float(loss), loss_emu=loss_e, out=out_err, out_noise=out_noise, whole=whole,
           whole_noise=whole_noise, worst=errs[worst], worst_noise=noise[worst], worst_name=worst,
           by_tensor={k: [round(errs[k], 4), round(noise[k], 4)] for k in errs})
    assert abs(float(loss) - loss_e) < 2e-3 * abs(loss_e), (float(loss), loss_e)
    assert out_err < 1.5 * out_noise + 5e-3, (out_err, out_noise)
    assert whole < 1.5 * whole_noise + 2e-2, (whole, whole_noise)
    assert errs[worst] < 3 * noise[worst] + 3e-2, (worst, errs[worst], noise[worst])
    # the loss end of the network, where bf16 leaves the comparison sharp
    for k in got:
        if k.startswith('conv_out.expert'):
            assert errs[k] < 5e-2, (k, errs[k])
            assert _rel2(1.1 * got[k], want[k]) > 5e-2          # a systematic 10 % error there is outside the bound


def test_net_golden_bf16_gradients():
    """G3 (reference Net, mult_chan 2) in bf16: output in 2-norm and the whole parameter gradient by direction and norm against
    the reference's float32 ones.  2 ... 32 channels a layer and 4 voxels on the deepest level: one bf16 rounding moves a whole
    BatchNorm statistic there (measured: output 5.2e-2, gradient 2-norm error 0.49)."""
    from repmode_amd.nn_modules.RepMode import Net
    g = load_golden('g3_net_mc2.npz')
    net = Net(Opts(), mult_chan=int(g['mult_chan']), dtype=torch.bfloat16)
    net.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('p.')})
    net.to(DEV).train()
    x, tgt = torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['target']).to(DEV)
    y = net(x, torch.from_numpy(g['tasks']))
    torch.nn.functional.mse_loss(y, tgt).backward()
    gm = torch.cat([p.grad.reshape(-1).float().cpu() for _, p in net.named_parameters()])
    gr = torch.cat([torch.from_numpy(g['d.' + k]).reshape(-1) for k, _ in net.named_parameters()])
    out = _rel2(y, torch.from_numpy(g['y']))
    cos, ratio = _cos_ratio(gm, gr)
    record('net_golden_bf16', whole_grad=_rel2(gm, gr), cos=cos, ratio=ratio, out=out)
    assert out < 0.1, out
    assert cos > 0.8 and abs(ratio - 1.0) < 0.15, (cos, ratio)


@pytest.mark.timeout(1200)
def test_bf16_predict_vs_oracle_at_mult_chan_32():
    """Sliding-window inference in bf16 with the eval-mode filter cache and the folded BatchNorm (fnet_model.py:149-223)
    on a mult_chan-32 network and a 32x96x128 volume (6 overlapping 32x64x64 patches, batches of 4) against the oracle's
    float32 predict from the same state."""
    from repmode_amd.model import Model
    torch.manual_seed(0)
    opts = Opts()
    opts.batch_size_eval = 4
    m = Model(opts, nn_module='RepMode', lr=1e-3, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():                      # non-trivial running statistics, so that the eval-mode BatchNorm does something
        for mod in m.net.modules():
            if isinstance(mod, torch.nn.BatchNorm3d):
                mod.running_mean.copy_((torch.rand(mod.running_mean.shape, generator=gen) * 0.2 - 0.1).to(DEV))
                mod.running_var.copy_((torch.rand(mod.running_var.shape, generator=gen) * 0.4 + 0.8).to(DEV))
    ref = orc.Net(opts, mult_chan=32)
    ref.load_state_dict({k: v.cpu() for k, v in m.net.state_dict().items()})
    sig = torch.randn(1, 1, 32, 96, 128, generator=gen)
    task = torch.tensor([9])
    got = m.predict(sig, task, (32, 64, 64))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    want = orc.predict(ref, sig, task, (32, 64, 64), opts.batch_size_eval)
    e_max, e_2 = rel_err(got, want), _rel2(got, want)
    record('bf16_predict', max=e_max, two_norm=e_2)
    assert e_2 < 2e-2, (e_max, e_2)
    assert e_max < 5e-2, (e_max, e_2)


@pytest.mark.timeout(900)
def test_plugin_called_the_way_the_reference_harness_calls_it(tmp_path, monkeypatch):
    """INTEGRATION.md section 1's three-line shim written into ``<tmp>/fnet/nn_modules/RepModeAMD.py``, found through
    ``importlib.import_module('fnet.nn_modules.' + nn_module).Net(opts)`` (fnet_model.py:52) and driven by the call order of
    fnet_model.py:98-113 spelled out here: tensors moved to the device (the task ids too), ``net.train()``,
    ``optimizer.zero_grad()``, forward + ``MSELoss(reduction='none')`` -> mean under ``torch.cuda.amp.autocast()``,
    ``GradScaler.scale(loss).backward()``, ``scaler.step``, ``scaler.update``.  Finite, no skipped scaler step, and the
    losses agree with repmode_amd.Model (bf16) on the same data from the same seed."""
    pkg = tmp_path / 'fnet' / 'nn_modules'
    pkg.mkdir(parents=True)
    (tmp_path / 'fnet' / '__init__.py').write_text('')
    (pkg / '__init__.py').write_text('')
    (pkg / 'RepModeAMD.py').write_text(
        'import sys\nsys.path.insert(0, %r)\nfrom repmode_amd.nn_modules.RepMode import Net  # noqa: F401\n' % ROOT)
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == 'fnet' or k.startswith('fnet.')]:
        monkeypatch.delitem(sys.modules, k)
    importlib.invalidate_caches()
    opts = Opts()
    opts.gpu_ids = 0
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    net = importlib.import_module('fnet.nn_modules.' + 'RepModeAMD').Net(opts)        # fnet_model.py:52
    net.to(device)                                                                     # :53
    optimizer = torch.optim.Adam(net.parameters(), lr=1e-3)                            # :55
    criterion = torch.nn.MSELoss(reduction='none')                                     # :36
    scaler = torch.cuda.amp.GradScaler()                                               # :46
    gen = torch.Generator().manual_seed(11)
    batches = [(torch.randn(4, 1, 16, 32, 32, generator=gen), torch.randn(4, 1, 16, 32, 32, generator=gen),
                torch.tensor([3, 7, 3, 11])) for _ in range(3)]
    losses = []
    for signal, target, task in batches:
        signal, target, task = signal.to(device), target.to(device), task.to(device)   # :98-100
        net.train()                                                                    # :102
        optimizer.zero_grad()                                                          # :105
        with torch.cuda.amp.autocast():                                                # :106
            output = net(signal, task)
            loss_nomean = criterion(output, target)
            loss = torch.mean(loss_nomean)
        scaler.scale(loss).backward()                                                  # :111
        scaler.step(optimizer)
        scaler.update()
        assert output.shape == signal.shape and output.dtype == torch.float32
        losses.append(loss.item())                                                     # :117
        for p in net.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()
    assert all(np.isfinite(losses))
    assert scaler.get_scale() == 65536.0              # the initial scale: no step was skipped for an inf / nan gradient
    # the build's own harness on the same data from the same seed
    from repmode_amd.model import Model
    torch.manual_seed(0)
    m = Model(Opts(), nn_module='RepMode', lr=1e-3, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
    own = []
    for signal, target, task in batches:
        m.do_train_iter(signal, target, task)
        own.append(float(m.last_loss))
    record('reference_harness_call', losses=losses, own=own)
    for a, b in zip(losses, own):
        assert abs(a - b) < 2e-2 * abs(b), (losses, own)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_deterministic_mode_is_bitwise_reproducible(dtype):
    """REPMODE_DETERMINISTIC / ops.set_deterministic(True): two train steps from the same state on the same batch (four
    16x32x32 patches, three distinct tasks: merged, two-tensor and per-expert blocks, every kernel family of the step) give
    bitwise identical losses, outputs, gradients and updated parameters -- and agree with the default (atomics) mode within
    the tolerance that mode's own run-to-run noise needs."""
    from repmode_amd import ops
    from repmode_amd.model import Model
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(4, 1, 16, 32, 32, generator=gen)
    t = torch.randn(4, 1, 16, 32, 32, generator=gen)
    tasks = torch.tensor([3, 7, 11, 3])

    def run(det):
        ops.set_deterministic(det)
        torch.manual_seed(0)
        m = Model(Opts(), nn_module='RepMode', lr=1e-3, gpu_ids=0, mult_chan=8, dtype=dtype)
        out = []
        for _ in range(2):
            o, per = m.do_train_iter(x, t, tasks)
            out += [float(m.last_loss), o.detach().float().cpu().clone(), per.detach().float().cpu().clone()]
        out += [p.grad.detach().float().cpu().clone() for p in m.net.parameters()]
        out += [p.detach().float().cpu().clone() for p in m.net.parameters()]
        return out

    try:
        a, b, c = run(True), run(True), run(False)
    finally:
        ops.set_deterministic(False)
    for u, v in zip(a, b):
        assert (u == v) if isinstance(u, float) else torch.equal(u, v)
    # against the default mode: the losses (two steps) within 1 %
    assert abs(a[0] - c[0]) < 1e-2 * abs(c[0]) and abs(a[3] - c[3]) < 1e-2 * abs(c[3])


@pytest.mark.timeout(900)
def test_batch24_train_iters_bf16_against_the_float32_path():
    """BASELINE configs[2] under test: ``Model.do_train_iter`` at batch 24 of 1x32x64x64, all 12
    tasks, mult_chan 32 -- the per-expert levels with 12 slots, the filter gradient's grid plans, the wave-specialised kernels
    at three times the headline batch.  Three bf16 steps against three float32 steps of the same kernels' parity mode from the
    same seeded state and data (the float32 path is what the goldens pin to the reference): losses within 2 %."""
    from repmode_amd.model import Model
    gen = torch.Generator().manual_seed(24)
    xs = [torch.randn(24, 1, 32, 64, 64, generator=gen) for _ in range(3)]
    ts = [torch.randn(24, 1, 32, 64, 64, generator=gen) for _ in range(3)]
    tasks = torch.arange(24) % 12
    losses = {}
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        m = Model(Opts(), nn_module='RepMode', lr=1e-3, gpu_ids=0, mult_chan=32, dtype=dtype)
        out = []
        for x, t in zip(xs, ts):
            y, per_sample = m.do_train_iter(x.to(DEV), t.to(DEV), tasks)
            assert y.shape == x.shape and torch.isfinite(y).all() and per_sample.shape == (24,)
            out.append(float(m.last_loss))
        losses[dtype] = out
        del m
        torch.cuda.empty_cache()
    record('batch24_train_iters', f32=losses[torch.float32], bf16=losses[torch.bfloat16])
    for a, b in zip(losses[torch.bfloat16], losses[torch.float32]):
        assert abs(a - b) < 2e-2 * abs(b), losses
