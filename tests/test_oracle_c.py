"""Cross-check of the two oracles: the plain-C restatement (oracle/mode_block_ref.c, explicit loops
and explicit backward formulas, double accumulation) against golden vectors captured from the
reference, and against the PyTorch-based oracle's autograd.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err
from oracle import repmode_oracle as orc

F = ctypes.POINTER(ctypes.c_float)
Dp = ctypes.POINTER(ctypes.c_double)
Ip = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope='module')
def cref():
    path = os.path.join(ROOT, 'oracle', 'libmode_block_ref.so')
    if not os.path.exists(path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    return ctypes.CDLL(path)


def fp(a):
    return a.ctypes.data_as(F)


def dp(a):
    return a.ctypes.data_as(Dp)


@pytest.mark.parametrize('name', ['g1_block_8_16.npz', 'g1_block_16_1_final.npz', 'g1_block_1_32.npz'])
def test_c_oracle_forward_vs_reference_golden(cref, name):
    g = load_golden(name)
    k5, k3, k1, a3, a5 = (np.ascontiguousarray(g['p.' + k]) for k in (
        'expert_conv5x5_conv', 'expert_conv3x3_conv', 'expert_conv1x1_conv', 'expert_avg3x3_conv', 'expert_avg5x5_conv'))
    gw, gb = np.ascontiguousarray(g['p.gate.weight']), np.ascontiguousarray(g['p.gate.bias'])
    co, ci = k5.shape[:2]
    x = np.ascontiguousarray(g['x'])
    n, _, D, H, W = x.shape
    tasks = g['tasks'].astype(np.int32)
    gp = np.zeros((n, 5, co), np.float32)
    cref.ref_gate_probs(fp(gw), fp(gb), tasks.ctypes.data_as(Ip), n, 12, co, fp(gp))
    assert rel_err(gp, g['g']) < 1e-6
    y = np.zeros((n, co, D, H, W), np.float32)
    for s in range(n):
        w = np.zeros((co, ci, 125), np.float32)
        cref.ref_merge_filter(fp(k5), fp(k3), fp(k1), fp(a3), fp(a5), fp(gp[s]), co, ci, fp(w))
        if 'w_merged' in g:
            assert rel_err(w.reshape(co, ci, 5, 5, 5), g['w_merged'][s]) < 1e-6
        cref.ref_conv5(fp(x[s]), fp(w), ci, co, D, H, W, fp(y[s]))
    assert rel_err(y, g['y_pre']) < 2e-6


def test_c_oracle_backward_vs_torch_oracle(cref):
    """Explicit dgrad / wgrad / GatRep-backward formulas == autograd of the PyTorch oracle."""
    gen = torch.Generator().manual_seed(3)
    co, ci, D, H, W, n = 6, 5, 3, 5, 7, 3
    u = lambda *s: (torch.rand(*s, generator=gen) - 0.5)
    ps = [u(co, ci, 5, 5, 5), u(co, ci, 3, 3, 3), u(co, ci, 1, 1, 1), u(co, ci, 1, 1, 1), u(co, ci, 1, 1, 1),
          u(5 * co, 12), u(5 * co)]
    tasks = torch.tensor([2, 7, 2])
    x = torch.randn(n, ci, D, H, W, generator=gen)
    r = torch.randn(n, co, D, H, W, generator=gen)
    ref = [p.clone().requires_grad_(True) for p in ps]
    xr = x.clone().requires_grad_(True)
    (orc.mode_conv_pre_bn(xr, *ref, tasks) * r).sum().backward()

    npp = [np.ascontiguousarray(p.numpy()) for p in ps]
    gp = np.zeros((n, 5, co), np.float32)
    t32 = tasks.numpy().astype(np.int32)
    cref.ref_gate_probs(fp(npp[5]), fp(npp[6]), t32.ctypes.data_as(Ip), n, 12, co, fp(gp))
    dk5, dk3 = np.zeros((co, ci, 125)), np.zeros((co, ci, 27))
    dk1, da3, da5 = np.zeros((co, ci)), np.zeros((co, ci)), np.zeros((co, ci))
    dgw, dgb = np.zeros((5 * co, 12)), np.zeros(5 * co)
    dx = np.zeros((n, ci, D, H, W), np.float32)
    xn, rn = np.ascontiguousarray(x.numpy()), np.ascontiguousarray(r.numpy())
    for s in range(n):
        w = np.zeros((co, ci, 125), np.float32)
        cref.ref_merge_filter(*(fp(a) for a in npp[:5]), fp(gp[s]), co, ci, fp(w))
        cref.ref_conv5_dgrad(fp(rn[s]), fp(w), ci, co, D, H, W, fp(dx[s]))
        dw = np.zeros((co, ci, 125))
        cref.ref_conv5_wgrad_acc(fp(xn[s]), fp(rn[s]), ci, co, D, H, W, dp(dw))
        dl = np.zeros((5, co))
        cref.ref_gatrep_bwd_acc(dp(dw), *(fp(a) for a in npp[:5]), fp(gp[s]), co, ci, dp(dk5), dp(dk3), dp(dk1),
                                dp(da3), dp(da5), dp(dl))
        dgw[:, int(tasks[s])] += dl.reshape(-1)
        dgb += dl.reshape(-1)
    assert rel_err(dx, xr.grad) < 1e-5
    got = [dk5.reshape(co, ci, 5, 5, 5), dk3.reshape(co, ci, 3, 3, 3), dk1.reshape(co, ci, 1, 1, 1),
           da3.reshape(co, ci, 1, 1, 1), da5.reshape(co, ci, 1, 1, 1), dgw, dgb]
    for name, a, b in zip(['k5', 'k3', 'k1', 'a3', 'a5', 'gate_w', 'gate_b'], got, ref):
        assert rel_err(a, b.grad) < 1e-5, name


ASAN_SCRIPT = r"""
import ctypes, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
F = ctypes.POINTER(ctypes.c_float); Dp = ctypes.POINTER(ctypes.c_double); Ip = ctypes.POINTER(ctypes.c_int)
fp = lambda a: a.ctypes.data_as(F); dp = lambda a: a.ctypes.data_as(Dp)
rng = np.random.RandomState(0)
co, ci, D, H, W, n = 6, 5, 3, 5, 7, 2                      # ragged on purpose: every loop bound differs
f = lambda *s: np.ascontiguousarray(rng.rand(*s).astype(np.float32) - 0.5)
k5, k3, k1, a3, a5, gw, gb = f(co, ci, 125), f(co, ci, 27), f(co, ci), f(co, ci), f(co, ci), f(5 * co, 12), f(5 * co)
x, r = f(n, ci, D, H, W), f(n, co, D, H, W)
tasks = np.array([2, 11], np.int32)
gp = np.zeros((n, 5, co), np.float32)
lib.ref_gate_probs(fp(gw), fp(gb), tasks.ctypes.data_as(Ip), n, 12, co, fp(gp))
dk5, dk3 = np.zeros((co, ci, 125)), np.zeros((co, ci, 27))
dk1, da3, da5 = np.zeros((co, ci)), np.zeros((co, ci)), np.zeros((co, ci))
for s in range(n):
    w = np.zeros((co, ci, 125), np.float32)
    lib.ref_merge_filter(fp(k5), fp(k3), fp(k1), fp(a3), fp(a5), fp(gp[s]), co, ci, fp(w))
    y = np.zeros((co, D, H, W), np.float32); dx = np.zeros((ci, D, H, W), np.float32)
    lib.ref_conv5(fp(x[s]), fp(w), ci, co, D, H, W, fp(y))
    lib.ref_conv5_dgrad(fp(r[s]), fp(w), ci, co, D, H, W, fp(dx))
    dw = np.zeros((co, ci, 125)); dl = np.zeros((5, co))
    lib.ref_conv5_wgrad_acc(fp(x[s]), fp(r[s]), ci, co, D, H, W, dp(dw))
    lib.ref_gatrep_bwd_acc(dp(dw), fp(k5), fp(k3), fp(k1), fp(a3), fp(a5), fp(gp[s]), co, ci, dp(dk5), dp(dk3), dp(dk1), dp(da3), dp(da5), dp(dl))
    assert np.isfinite(y).all() and np.isfinite(dx).all() and np.isfinite(dw).all()
print('ASAN_RUN_OK')
"""


def test_c_oracle_under_address_sanitizer():
    """SURVEY.md section 5 (race detection / sanitizers: absent in the reference; the build's counterpart on the CPU side): the
    plain-C restatement compiled with -fsanitize=address runs every entry point on ragged shapes in a subprocess (libasan
    preloaded) without a report -- no out-of-bounds tap, halo or gradient index in the explicit loops the parity tests trust."""
    import sys
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip('no libasan next to this gcc')
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(asan), ASAN_OPTIONS='detect_leaks=0:abort_on_error=0')
    p = subprocess.run([sys.executable, '-c', ASAN_SCRIPT, os.path.join(ROOT, 'oracle', 'libmode_block_ref_asan.so')], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'ASAN_RUN_OK' in p.stdout and 'AddressSanitizer' not in p.stderr, p.stderr[-2000:]
