"""Operators of the MoDE hot path: thin autograd wrappers over the C ABI (include/repmode_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only; every FLOP of the
MoDE block runs in the hand-written HIP kernels of librepmode_hip.so.  Tensors handed to the
library are channels-last (NDHWC) and contiguous.  There is no CPU / eager fallback: calling an
operator on a non-HIP tensor raises.
"""
import ctypes

import torch

from . import _lib

NUM_EXPERTS = 5
TAPS = 125

_DTYPE_CODE = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def dtype_code(dtype):
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError('repmode_amd computes in float32 or bfloat16, got %s' % dtype)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_hip(t, what):
    if not t.is_cuda:
        raise _lib.RepModeHipError(
            '%s is on %s: repmode_amd runs on MI355X (HIP) tensors only and has no CPU fallback' % (what, t.device))


class TaskPlan:
    """Which merged filter each sample uses.

    The reference turns ``task`` into a one-hot matrix (RepMode.py:44-49) and merges one filter per
    SAMPLE (:182-190).  The filter depends on the task only, so the batch is grouped into "slots"
    (distinct tasks): ``slot_task[s]`` = task id of slot s, ``sample_slot[n]`` = slot of sample n.
    In eval mode the reference applies sample 0's filter to the whole batch (RepMode.py:209-210),
    i.e. one slot.
    """

    def __init__(self, tasks, num_tasks, device, training=True):
        if torch.is_tensor(tasks):
            if tasks.dim() == 2:                       # one-hot rows as the reference's MoDEConv takes
                tasks = tasks.argmax(dim=1)
            host = [int(v) for v in tasks.detach().cpu().tolist()]   # device tensor: one sync (the
            # reference does N of them in one_hot_task_embedding); pass a CPU tensor to avoid it
        else:
            host = [int(v) for v in tasks]
        for v in host:
            if not 0 <= v < num_tasks:
                raise ValueError('task id %d outside [0, %d)' % (v, num_tasks))
        self.tasks_host = host
        self.num_tasks = num_tasks
        self.training = training
        if training:
            uniq = sorted(set(host))
            index = {t: i for i, t in enumerate(uniq)}
            slots = [index[t] for t in host]
        else:
            uniq = [host[0]]
            slots = [0] * len(host)
        self.nslots = len(uniq)
        self.n = len(host)
        self.slot_task_host = uniq
        self.slot_task = torch.tensor(uniq, dtype=torch.int32).to(device, non_blocking=True)
        self.sample_slot = torch.tensor(slots, dtype=torch.int32).to(device, non_blocking=True)


def gate_softmax(gate_w, gate_b, plan, co):
    """g[s, e, o] (float32) -- RepMode.py:198-200."""
    g = torch.empty((plan.nslots, NUM_EXPERTS, co), dtype=torch.float32, device=gate_w.device)
    _lib.call('repmode_gate_softmax', _ptr(gate_w), _ptr(gate_b), _ptr(plan.slot_task), plan.nslots,
              plan.num_tasks, co, _ptr(g), _stream())
    return g


def gatrep_merge(k5, k3, k1, a3, a5, g, dtype, want_wf=True, want_wd=False):
    """Merged per-slot filters in the conv kernels' layouts -- RepMode.py:171-192."""
    co, ci = k5.shape[0], k5.shape[1]
    code = dtype_code(dtype)
    s = g.shape[0]
    wf = wd = None
    if want_wf:
        wf = torch.empty((s, TAPS, _lib.padded_channels(co, code, False), _lib.padded_channels(ci, code, True)),
                         dtype=dtype, device=k5.device)
    if want_wd:
        wd = torch.empty((s, TAPS, _lib.padded_channels(ci, code, False), _lib.padded_channels(co, code, True)),
                         dtype=dtype, device=k5.device)
    _lib.call('repmode_gatrep_fwd', _ptr(k5), _ptr(k3), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(g), s, co, ci, code,
              _ptr(wf) if want_wf else None, _ptr(wd) if want_wd else None, _stream())
    return wf, wd


def conv5(x_cl, w, sample_slot, cout, out_f32=False):
    """y[n] = x[n] (*) w[sample_slot[n]], 5^3 'same' cross-correlation, NDHWC -- RepMode.py:204-210."""
    n, d, h, wd_, cin = x_cl.shape
    code = dtype_code(x_cl.dtype)
    out_dtype = torch.float32 if (out_f32 or x_cl.dtype == torch.float32) else x_cl.dtype
    y = torch.empty((n, d, h, wd_, cout), dtype=out_dtype, device=x_cl.device)
    _lib.call('repmode_conv5', _ptr(x_cl), _ptr(w), _ptr(sample_slot), _ptr(y), n, d, h, wd_, cin, cout, code,
              1 if out_dtype == torch.float32 else 0, _stream())
    return y


def conv5_wgrad(x_cl, dy_cl, plan, cout):
    """dw[s, tap, o, i] (float32) summed over the samples of each slot."""
    n, d, h, wd_, cin = x_cl.shape
    dw = torch.empty((plan.nslots, TAPS, cout, cin), dtype=torch.float32, device=x_cl.device)
    _lib.call('repmode_conv5_wgrad', _ptr(x_cl), _ptr(dy_cl), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
              n, d, h, wd_, cin, cout, dtype_code(x_cl.dtype), _stream())
    return dw


class _ModeConv3d(torch.autograd.Function):
    """Fused gate-softmax + GatRep + per-slot 5^3 convolution, forward and backward."""

    @staticmethod
    def forward(ctx, x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32):
        _require_hip(x_cl, 'input')
        co = k5.shape[0]
        g = gate_softmax(gate_w, gate_b, plan, co)
        wf, _ = gatrep_merge(k5, k3, k1, a3, a5, g, x_cl.dtype, want_wf=True, want_wd=False)
        y = conv5(x_cl, wf, plan.sample_slot, co, out_f32)
        ctx.save_for_backward(x_cl, k5, k3, k1, a3, a5, g)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, k5, k3, k1, a3, a5, g = ctx.saved_tensors
        plan = ctx.plan
        co, ci = k5.shape[0], k5.shape[1]
        dy = dy.to(x_cl.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            # the filter is re-merged in the data-gradient layout instead of being kept from forward
            _, wd = gatrep_merge(k5, k3, k1, a3, a5, g, x_cl.dtype, want_wf=False, want_wd=True)
            # deep levels (small volumes) split the channel reduction over workgroups -> float output
            dx = conv5(dy, wd, plan.sample_slot, ci, out_f32=x_cl.shape[3] < 32)
            if dx.dtype != x_cl.dtype:
                dx = dx.to(x_cl.dtype)
            del wd
        dw = conv5_wgrad(x_cl, dy, plan, co)
        dk5, dk3, dk1 = torch.empty_like(k5), torch.empty_like(k3), torch.empty_like(k1)
        da3, da5 = torch.empty_like(a3), torch.empty_like(a5)
        # gate.weight is [5*Co, T], gate.bias [5*Co]; shapes are recovered from g / plan
        dgw = torch.empty((NUM_EXPERTS * co, plan.num_tasks), dtype=torch.float32, device=k5.device)
        dgb = torch.empty((NUM_EXPERTS * co,), dtype=torch.float32, device=k5.device)
        dg_ws = torch.empty_like(g)
        _lib.call('repmode_gatrep_bwd', _ptr(dw), _ptr(k5), _ptr(k3), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(g),
                  _ptr(plan.slot_task), plan.nslots, plan.num_tasks, co, ci, _ptr(dk5), _ptr(dk3), _ptr(dk1),
                  _ptr(da3), _ptr(da5), _ptr(dgw), _ptr(dgb), _ptr(dg_ws), _stream())
        return dx, dk5, dk3, dk1, da3, da5, dgw, dgb, None, None


def mode_conv3d(x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32=False):
    """The MoDE block up to (not including) BN/ReLU, on a channels-last tensor.

    x_cl: [N, D, H, W, Ci] float32 or bfloat16 (HIP).  Expert / gate parameters: float32, the
    reference's shapes.  Returns [N, D, H, W, Co] in x's dtype (float32 when ``out_f32``).
    """
    x_cl = x_cl.contiguous()
    ps = [p.contiguous() for p in (k5, k3, k1, a3, a5, gate_w, gate_b)]
    for p in ps:
        if p.dtype != torch.float32:
            raise TypeError('MoDE parameters must be float32')
    if plan.n != x_cl.shape[0]:
        raise ValueError('task plan is for %d samples, input has %d' % (plan.n, x_cl.shape[0]))
    return _ModeConv3d.apply(x_cl, *ps, plan, out_f32)
