"""Operators of the MoDE hot path: thin autograd wrappers over the C ABI (include/repmode_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only; every FLOP of the
MoDE block runs in the hand-written HIP kernels of librepmode_hip.so.  Tensors handed to the
library are channels-last (NDHWC) and contiguous.  There is no CPU / eager fallback: calling an
operator on a non-HIP tensor raises.
"""
import contextlib
import os

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
    return t.data_ptr()            # plain int: ctypes converts it for a c_void_p parameter (no wrapper object)


def _stream():
    # raw handle of the calling thread's current stream (the Stream-object route costs ~10 us per call, 600 calls a step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _require_hip(t, what):
    if not t.is_cuda:
        raise _lib.RepModeHipError(
            '%s is on %s: repmode_amd runs on MI355X (HIP) tensors only and has no CPU fallback' % (what, t.device))


# ---- a second HIP stream for independent launches of one layer.  On the deep levels a conv / filter-gradient launch
# has 128-256 workgroups for 512 slots, and a layer's data gradient, filter gradient and 1x1-expert GEMMs do not depend
# on each other: issued on two streams they share the chip instead of queueing behind each other's tails.  Opt-in
# (REPMODE_FORK_MAX_W=16): launched kernel by kernel the host cannot feed two streams fast enough (no gain); inside a
# HIP graph (Model(hip_graph=True)) it is worth 2.5 % of the step (same box: 14.28 -> 13.93 ms), at the price of
# per-launch durations that no longer describe one kernel.
FORK_MAX_W = int(os.environ.get('REPMODE_FORK_MAX_W', '0'))      # layers with W <= this fork (0: never)
_SIDE_STREAMS = {}


def _fork(t):
    """(main, side) streams for the device of tensor ``t``; the side stream is ordered after everything issued on the
    current stream so far.  Rules for the caller: every launch that touches a tensor on the side stream lies between
    ``_fork`` and ``_join``, and ``_join`` comes before the function returns (so no tensor is freed, and no result
    consumed, while the side stream still works on it)."""
    main = torch.cuda.current_stream(t.device)
    side = _SIDE_STREAMS.get(t.device.index)
    if side is None:
        side = _SIDE_STREAMS[t.device.index] = torch.cuda.Stream(t.device)
    side.wait_stream(main)
    return main, side


def _join(main, side):
    main.wait_stream(side)


def _forks(x_cl):
    return 0 < x_cl.shape[3] <= FORK_MAX_W and x_cl.is_cuda


class ZeroPool:
    """One pre-zeroed buffer per train step for the float accumulation targets of the atomics-based kernels (split-K
    conv outputs, chunked filter gradients): ~60 memset launches per step become one.

    The first step with a given key records the requested sizes in order; later steps allocate the total once
    (``torch.zeros``), hand out views in the same order and tell the kernels to skip their own clearing.  Any
    divergence from the recorded sequence falls back to plain allocations for the rest of the step.  A fresh buffer
    is allocated every step, so views held across steps (saved activations, a returned output) stay valid.
    """

    ALIGN = 64        # floats (256 bytes)

    def __init__(self):
        self.plans = {}
        self.key = None
        self.req = []
        self.buf = None

    def begin(self, key, device):
        self.end()
        self.key, self.req, self.pos, self.off = key, [], 0, 0
        plan = self.plans.get(key)
        self.plan = plan
        self.buf = None
        if plan:
            total = sum(-(-n // self.ALIGN) * self.ALIGN for n in plan)
            self.buf = torch.zeros(total, dtype=torch.float32, device=device)

    def end(self):
        if self.key is not None and self.req:
            self.plans[self.key] = self.req
        self.key, self.req, self.buf = None, [], None

    def take(self, shape, device):
        """(float32 tensor of ``shape``, prezeroed?)"""
        n = 1
        for v in shape:
            n *= int(v)
        if self.key is not None:
            self.req.append(n)
            if self.buf is not None and self.pos < len(self.plan) and self.plan[self.pos] == n:
                # an independent tensor over the pool's storage (not a view: views would share ONE version counter,
                # and an in-place torch op on any of them would invalidate every saved one for autograd)
                t = torch.empty(0, dtype=torch.float32, device=device).set_(
                    self.buf.untyped_storage(), self.off, tuple(int(v) for v in shape))
                self.off += -(-n // self.ALIGN) * self.ALIGN
                self.pos += 1
                return t, True
            self.buf = None          # sequence differs from the recorded one: plain allocations from here on
        return torch.empty(shape, dtype=torch.float32, device=device), False


ZERO_POOL = ZeroPool()


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
        self.sample_task = torch.tensor(host, dtype=torch.int32).to(device, non_blocking=True)   # task id per sample
        self.bn_counted = False     # set by Net.forward once num_batches_tracked of every BN layer is advanced


def gate_softmax(gate_w, gate_b, plan, co):
    """g[s, e, o] (float32) -- RepMode.py:198-200."""
    g = torch.empty((plan.nslots, NUM_EXPERTS, co), dtype=torch.float32, device=gate_w.device)
    _lib.call('repmode_gate_softmax', _ptr(gate_w), _ptr(gate_b), _ptr(plan.slot_task), plan.nslots,
              plan.num_tasks, co, _ptr(g), _stream())
    return g


def expert_frags(k5, k3, dtype, want_wd=True):
    """The raw 5^3 / 3^3 experts as two un-merged slots of the conv kernels' layouts (slot 1 = K3, valid for
    ``centre3`` convolutions only).  bf16: one layout kernel per role; float32 (parity mode): GatRep with one-hot
    gate probabilities, which writes all taps."""
    co, ci = k5.shape[0], k5.shape[1]
    if dtype != torch.bfloat16:
        return gatrep_merge(k5, k3, k5.new_zeros((co, ci)), k5.new_zeros((co, ci)), k5.new_zeros((co, ci)),
                            _expert_selector(co, k5.device), dtype, want_wf=True, want_wd=want_wd)
    code = dtype_code(dtype)
    wf = torch.empty((2, TAPS, _lib.padded_channels(co, code, False), _lib.padded_channels(ci, code, True)),
                     dtype=dtype, device=k5.device)
    wd = torch.empty((2, TAPS, _lib.padded_channels(ci, code, False), _lib.padded_channels(co, code, True)),
                     dtype=dtype, device=k5.device) if want_wd else None
    _lib.call('repmode_expert_frags', _ptr(k5), _ptr(k3), co, ci, _ptr(wf), _ptr(wd) if want_wd else None, _stream())
    return wf, wd


def gate_softmax_samples(gate_w, gate_b, plan, co):
    """g[n, e, o] per SAMPLE (the same kernel with one "slot" per sample)."""
    g = torch.empty((plan.n, NUM_EXPERTS, co), dtype=torch.float32, device=gate_w.device)
    _lib.call('repmode_gate_softmax', _ptr(gate_w), _ptr(gate_b), _ptr(plan.sample_task), plan.n, plan.num_tasks, co,
              _ptr(g), _stream())
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


GRAD_SINK = None    # a distributed.GradReducer: parameter gradients are then written straight into its buckets


def _grad_out(param):
    """Output buffer for the gradient of ``param``: its slice of the data-parallel reducer's communication bucket when
    there is one (no per-gradient bucket copy -- 193 small kernels per step under DistributedDataParallel), else a
    fresh tensor."""
    sink = GRAD_SINK
    if sink is not None:
        v = sink.grad_buffer(param)
        if v is not None:
            return v
    return torch.empty_like(param)


def conv5(x_cl, w, sample_slot, cout, out_f32=False, out=None, centre3=False, accumulate=False, dxc=False):
    """y[n] = x[n] (*) w[sample_slot[n]], 5^3 'same' cross-correlation, NDHWC -- RepMode.py:204-210.
    ``accumulate`` (float ``out`` given): add to ``out`` instead of overwriting it.  ``dxc``: only the centre x tap
    of every (dz, dy) row (thin layers with their x taps folded into channels, see ``thin_conv_*``)."""
    n, d, h, wd_, cin = x_cl.shape
    code = dtype_code(x_cl.dtype)
    out_dtype = torch.float32 if (out_f32 or x_cl.dtype == torch.float32) else x_cl.dtype
    if out is not None:
        y = out
    elif out_dtype == torch.float32 and x_cl.dtype == torch.bfloat16:
        # float output = the kernel may split the reduction and add with atomics: a pre-zeroed pool view saves its memset
        y, pre = ZERO_POOL.take((n, d, h, wd_, cout), x_cl.device)
        accumulate = accumulate or pre
    else:
        y = torch.empty((n, d, h, wd_, cout), dtype=out_dtype, device=x_cl.device)
    assert y.dtype == out_dtype and y.is_contiguous()
    assert not accumulate or out_dtype == torch.float32
    _lib.call('repmode_conv5_ex', _ptr(x_cl), _ptr(w), _ptr(sample_slot), _ptr(y), n, d, h, wd_, cin, cout, code,
              1 if out_dtype == torch.float32 else 0, (1 if centre3 else 0) | (2 if accumulate else 0) | (4 if dxc else 0), _stream())
    return y


def _thin_pack(w, to_rows):
    """Re-pack a thin layer's merged filter [S, 125, rowsP, redP] (bf16) for the conv kernel's dx-centre mode."""
    s_, _, rp, kp = w.shape
    assert (kp == 16 and not to_rows) or (rp == 32 and to_rows)
    out = torch.empty_like(w)             # only the 25 (dz, dy, dx=2) taps are written -- and read
    _lib.call('repmode_thin_pack', _ptr(w), _ptr(out), s_, (rp // 32) * (kp // 16), 1 if to_rows else 0, _stream())
    return out


def _shift5(t_cl):
    """[N, D, H, W, 1] (float or bf16) -> [N, D, H, W, 8] bf16 with channel dx = the input shifted by dx - 2 along x."""
    n, d, h, w_, _ = t_cl.shape
    out = torch.empty((n, d, h, w_, 8), dtype=torch.bfloat16, device=t_cl.device)
    _lib.call('repmode_shift5', _ptr(t_cl), dtype_code(t_cl.dtype), _ptr(out), n * d * h, w_, _stream())
    return out


def thin_conv_in1(x_cl, w, sample_slot, cout, out_f32=False):
    """conv5 for a ONE-channel input (bf16 filter ``w`` [S, 125, CoP, 16]): the five x taps become channels and the
    general kernel runs 25 instead of 125 taps (csrc/thin.hip).  Used for the first layer's forward and (with the
    data-gradient filter) for the last layer's data gradient."""
    return conv5(_shift5(x_cl.contiguous()), _thin_pack(w, False), sample_slot, cout, out_f32, dxc=True)


def thin_conv_out1(x_cl, w, sample_slot):
    """conv5 for ONE output channel (bf16 filter ``w`` [S, 125, 32, CiP]): the five x taps become output rows, then a
    5-tap diagonal sum (csrc/thin.hip).  Returns float [N, D, H, W, 1]."""
    n, d, h, w_, _ = x_cl.shape
    y5 = conv5(x_cl, _thin_pack(w, True), sample_slot, 5, out_f32=True, dxc=True)
    y = torch.empty((n, d, h, w_, 1), dtype=torch.float32, device=x_cl.device)
    _lib.call('repmode_unshift5', _ptr(y5), _ptr(y), n * d * h, w_, _stream())
    return y


def conv5_wgrad(x_cl, dy_cl, plan, cout, centre3=False, expert_layout=None, out=None):
    """dw[s, tap, o, i] (float32) summed over the samples of each slot."""
    n, d, h, wd_, cin = x_cl.shape
    if expert_layout is not None:
        # single slot, gradient written directly in the experts' parameter layout: [Co, Ci, 5,5,5] or [Co, Ci, 3,3,3]
        k = expert_layout
        dw = out if out is not None else torch.empty((cout, cin, k, k, k), dtype=torch.float32, device=x_cl.device)
        assert dw.shape == (cout, cin, k, k, k) and dw.dtype == torch.float32 and dw.is_contiguous()
        _lib.call('repmode_conv5_wgrad_ex', _ptr(x_cl), _ptr(dy_cl), _ptr(plan.sample_slot), 1, _ptr(dw),
                  n, d, h, wd_, cin, cout, dtype_code(x_cl.dtype), 2 if k == 5 else 3, _stream())
        return dw
    dw, pre = ZERO_POOL.take((plan.nslots, TAPS, cout, cin), x_cl.device)
    if x_cl.dtype == torch.bfloat16 and (cin == 1) != (cout == 1) and not centre3:
        # thin layer: taps stand in for the missing channel dimension (conv5_wgrad_thin)
        a_t, b_t, c, flip = (dy_cl, x_cl, cout, 0) if cin == 1 else (x_cl, dy_cl, cin, 1)
        _lib.call('repmode_conv5_wgrad_thin', _ptr(a_t), _ptr(b_t), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
                  n, d, h, wd_, c, flip | (2 if pre else 0), _stream())
        return dw
    _lib.call('repmode_conv5_wgrad_ex', _ptr(x_cl), _ptr(dy_cl), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
              n, d, h, wd_, cin, cout, dtype_code(x_cl.dtype), (1 if centre3 else 0) | (8 if pre else 0), _stream())
    return dw


_EVAL_FILTERS = None     # dict inside ``eval_filter_cache()``: (expert storage, tasks, dtype) -> merged forward filter


@contextlib.contextmanager
def eval_filter_cache():
    """Inside this context the merged filter of an eval-mode MoDE block (one slot: RepMode.py:209-210) is computed once
    per (block, task, dtype) instead of once per forward: sliding-window inference re-uses it for every batch of
    patches of a volume.  The parameters must not change inside the context."""
    global _EVAL_FILTERS
    prev, _EVAL_FILTERS = _EVAL_FILTERS, {}
    try:
        yield
    finally:
        _EVAL_FILTERS = prev


def _merged_filters(k5, k3, k1, a3, a5, gate_w, gate_b, plan, dtype, want_wd):
    """(g, forward filter, data-gradient filter or None) of a MoDE block: gate softmax + GatRep, or the filter kept by
    ``eval_filter_cache`` for this block and task."""
    cache = _EVAL_FILTERS if not (plan.training or want_wd or torch.is_grad_enabled()) else None
    key = (k5.data_ptr(), tuple(plan.slot_task_host), dtype) if cache is not None else None
    if cache is not None and key in cache:
        g, wf = cache[key]
        return g, wf, None
    g = gate_softmax(gate_w, gate_b, plan, k5.shape[0])
    wf, wd = gatrep_merge(k5, k3, k1, a3, a5, g, dtype, want_wf=True, want_wd=want_wd)
    if cache is not None:
        cache[key] = (g, wf)
    return g, wf, wd


def _filter_and_expert_grads(dw, k5, k3, k1, a3, a5, g, plan):
    """GatRep backward: per-slot filter gradient dw [S, 125, Co, Ci] -> (dk5, dk3, dk1, da3, da5, dgate_w, dgate_b)."""
    co, ci = k5.shape[0], k5.shape[1]
    dk5, dk3, dk1, da3, da5 = _grad_out(k5), _grad_out(k3), _grad_out(k1), _grad_out(a3), _grad_out(a5)
    # gate.weight is [5*Co, T], gate.bias [5*Co]; shapes are recovered from g / plan
    dgw = torch.empty((NUM_EXPERTS * co, plan.num_tasks), dtype=torch.float32, device=k5.device)
    dgb = torch.empty((NUM_EXPERTS * co,), dtype=torch.float32, device=k5.device)
    dg_ws = torch.empty_like(g)
    _lib.call('repmode_gatrep_bwd', _ptr(dw), _ptr(k5), _ptr(k3), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(g),
              _ptr(plan.slot_task), plan.nslots, plan.num_tasks, co, ci, _ptr(dk5), _ptr(dk3), _ptr(dk1),
              _ptr(da3), _ptr(da5), _ptr(dgw), _ptr(dgb), _ptr(dg_ws), _stream())
    return dk5, dk3, dk1, da3, da5, dgw, dgb


class _ModeConv3d(torch.autograd.Function):
    """Fused gate-softmax + GatRep + per-slot 5^3 convolution, forward and backward."""

    @staticmethod
    def forward(ctx, x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32):
        _require_hip(x_cl, 'input')
        co = k5.shape[0]
        # the data-gradient filter comes out of the same pass over the experts (one launch, one read of the
        # weights); these are the shallow levels, whose merged filters are small next to the activations
        g, wf, wd = _merged_filters(k5, k3, k1, a3, a5, gate_w, gate_b, plan, x_cl.dtype, ctx.needs_input_grad[0])
        ci = k5.shape[1]
        thin = x_cl.dtype == torch.bfloat16 and (ci == 1) != (co == 1)
        if thin and ci == 1:                                  # first layer: x taps folded into input channels
            y = thin_conv_in1(x_cl, wf, plan.sample_slot, co, out_f32)
        elif thin:                                            # last layer: x taps folded into output rows
            y = thin_conv_out1(x_cl, wf, plan.sample_slot)
            if not out_f32:
                y = y.to(x_cl.dtype)
        else:
            y = conv5(x_cl, wf, plan.sample_slot, co, out_f32)
        ctx.save_for_backward(x_cl, k5, k3, k1, a3, a5, g, wd)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, k5, k3, k1, a3, a5, g, wd = ctx.saved_tensors
        plan = ctx.plan
        co, ci = k5.shape[0], k5.shape[1]
        dy = dy.to(x_cl.dtype).contiguous()
        dx = None
        # the filter gradient and the GatRep backward do not depend on the data gradient: second stream on the deep levels
        fork = ctx.needs_input_grad[0] and _forks(x_cl)
        if fork:
            main, side = _fork(x_cl)
            with torch.cuda.stream(side):
                grads = _filter_and_expert_grads(conv5_wgrad(x_cl, dy, plan, co), k5, k3, k1, a3, a5, g, plan)
        else:
            grads = _filter_and_expert_grads(conv5_wgrad(x_cl, dy, plan, co), k5, k3, k1, a3, a5, g, plan)
        if ctx.needs_input_grad[0]:
            # deep levels (small volumes) split the channel reduction over workgroups -> float output
            if x_cl.dtype == torch.bfloat16 and co == 1 and ci != 1:     # the last layer: dy has one channel
                dx = thin_conv_in1(dy, wd, plan.sample_slot, ci, out_f32=x_cl.shape[3] < 32)
            elif x_cl.dtype == torch.bfloat16 and ci == 1 and co != 1:   # (first layer, if its input ever needs a gradient)
                dx = thin_conv_out1(dy, wd, plan.sample_slot)
            else:
                dx = conv5(dy, wd, plan.sample_slot, ci, out_f32=x_cl.shape[3] < 32)
            if dx.dtype != x_cl.dtype:
                dx = dx.to(x_cl.dtype)
            del wd
        if fork:
            _join(main, side)
        return (dx,) + grads + (None, None)


class _ModeConv3dPair(torch.autograd.Function):
    """``_ModeConv3d`` for a skip connection: the block's input is the channel concatenation of two tensors
    (RepMode.py:106 ``torch.cat((x_skip, up), 1)``), which is never materialised -- the conv kernel reads its channel
    chunks from either tensor, the data gradient is written to two tensors, and the filter gradient is computed in
    two channel ranges of one buffer (csrc: repmode_conv5_pair, repmode_conv5_wgrad_part)."""

    @staticmethod
    def forward(ctx, xa, xb, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32):
        _require_hip(xa, 'input')
        co, ci = k5.shape[0], k5.shape[1]
        ca = xa.shape[-1]
        n, d, h, w_ = xa.shape[:4]
        g, wf, wd = _merged_filters(k5, k3, k1, a3, a5, gate_w, gate_b, plan, xa.dtype,
                                    ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        code = dtype_code(xa.dtype)
        out_dtype = torch.float32 if (out_f32 or xa.dtype == torch.float32) else xa.dtype
        flags = 0
        if out_dtype == torch.float32 and xa.dtype == torch.bfloat16:
            y, pre = ZERO_POOL.take((n, d, h, w_, co), xa.device)
            flags = 2 if pre else 0
        else:
            y = torch.empty((n, d, h, w_, co), dtype=out_dtype, device=xa.device)
        _lib.call('repmode_conv5_pair', _ptr(xa), _ptr(xb), ca, _ptr(wf), _ptr(plan.sample_slot), _ptr(y), None, 0,
                  n, d, h, w_, ci, co, code, 1 if out_dtype == torch.float32 else 0, flags, _stream())
        ctx.save_for_backward(xa, xb, k5, k3, k1, a3, a5, g, wd)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, k5, k3, k1, a3, a5, g, wd = ctx.saved_tensors
        plan = ctx.plan
        co, ci = k5.shape[0], k5.shape[1]
        ca, cb = xa.shape[-1], xb.shape[-1]
        n, d, h, w_ = xa.shape[:4]
        dt = xa.dtype
        code = dtype_code(dt)
        dy = dy.to(dt).contiguous()

        def filter_grads():
            # the two channel ranges of one (cleared) buffer
            dw, pre = ZERO_POOL.take((plan.nslots, TAPS, co, ci), xa.device)
            if not pre:
                dw.zero_()
            for part, off in ((xa, 0), (xb, ca)):
                _lib.call('repmode_conv5_wgrad_part', _ptr(part), _ptr(dy), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
                          n, d, h, w_, part.shape[-1], ci, off, co, code, 8, _stream())
            return _filter_and_expert_grads(dw, k5, k3, k1, a3, a5, g, plan)

        fork = wd is not None and _forks(xa)
        if fork:
            main, side = _fork(xa)
            with torch.cuda.stream(side):
                grads = filter_grads()
        else:
            grads = filter_grads()
        dxa = dxb = None
        if wd is not None:
            f32 = w_ < 32 or dt == torch.float32        # deep levels: split reduction -> float output (as _ModeConv3d)
            flags = 0
            if f32 and dt == torch.bfloat16:
                (dxa, pa), (dxb, pb) = ZERO_POOL.take((n, d, h, w_, ca), xa.device), ZERO_POOL.take((n, d, h, w_, cb), xa.device)
                if pa and pb:
                    flags = 2
                elif pa or pb:                           # (cannot happen with a consistent pool; stay correct anyway)
                    dxa.zero_(); dxb.zero_(); flags = 2
            else:
                odt = torch.float32 if f32 else dt
                dxa = torch.empty((n, d, h, w_, ca), dtype=odt, device=xa.device)
                dxb = torch.empty((n, d, h, w_, cb), dtype=odt, device=xa.device)
            _lib.call('repmode_conv5_pair', _ptr(dy), None, 0, _ptr(wd), _ptr(plan.sample_slot), _ptr(dxa), _ptr(dxb), ca,
                      n, d, h, w_, co, ci, code, 1 if f32 else 0, flags, _stream())
            if dxa.dtype != dt:
                dxa, dxb = dxa.to(dt), dxb.to(dt)
        if fork:
            _join(main, side)
        return (dxa, dxb) + grads + (None, None)


def pair_supported(xa_cl, xb_cl, plan):
    """The two-tensor path covers the merged formulation with channel counts on tile boundaries."""
    kc = 16 if xa_cl.dtype == torch.bfloat16 else 8
    return (xa_cl.shape[-1] % 32 == 0 and xb_cl.shape[-1] % kc == 0 and xa_cl.shape[:4] == xb_cl.shape[:4]
            and xa_cl.dtype == xb_cl.dtype and not use_unmerged(xa_cl, plan))


def mode_conv3d_pair(xa_cl, xb_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32=False):
    """``mode_conv3d`` of the channel concatenation (xa, xb) without building it; falls back to cat + mode_conv3d
    where the two-tensor kernels do not apply (per-expert formulation, odd channel counts)."""
    if not pair_supported(xa_cl, xb_cl, plan):
        return mode_conv3d(torch.cat((xa_cl, xb_cl), dim=-1), k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32)
    ps = [p.contiguous() for p in (k5, k3, k1, a3, a5, gate_w, gate_b)]
    for p in ps:
        if p.dtype != torch.float32:
            raise TypeError('MoDE parameters must be float32')
    if plan.n != xa_cl.shape[0]:
        raise ValueError('task plan is for %d samples, input has %d' % (plan.n, xa_cl.shape[0]))
    return _ModeConv3dPair.apply(xa_cl.contiguous(), xb_cl.contiguous(), *ps, plan, out_f32)


class _BnRelu(torch.autograd.Function):
    """BatchNorm3d + ReLU on a channels-last tensor [..., C] (RepMode.py:146-149, 212; :80-84; :97-101)."""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, running_mean, running_var, training, momentum, eps, out_dtype):
        _require_hip(x_cl, 'input')
        c = x_cl.shape[-1]
        m = x_cl.numel() // c
        out = torch.empty(x_cl.shape, dtype=out_dtype, device=x_cl.device)
        save_mean = torch.empty(c, dtype=torch.float32, device=x_cl.device)
        save_invstd = torch.empty_like(save_mean)
        _lib.call('repmode_bn_relu_fwd', _ptr(x_cl), _ptr(out), _ptr(weight), _ptr(bias), _ptr(running_mean),
                  _ptr(running_var), _ptr(save_mean), _ptr(save_invstd), m, c, float(eps), float(momentum),
                  1 if training else 0, dtype_code(x_cl.dtype), dtype_code(out_dtype), _stream())
        ctx.save_for_backward(x_cl, weight, bias, save_mean, save_invstd)
        ctx.training = training
        ctx.mark_non_differentiable(running_mean, running_var)
        return out

    @staticmethod
    def backward(ctx, dy):
        x_cl, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        c = x_cl.shape[-1]
        m = x_cl.numel() // c
        dy = dy.contiguous()
        dx = torch.empty_like(x_cl)
        tot = torch.empty(2 * c, dtype=torch.float32, device=x_cl.device)    # [0:c) = dbeta, [c:2c) = dgamma
        _lib.call('repmode_bn_relu_bwd', _ptr(x_cl), _ptr(dy), _ptr(weight), _ptr(bias), _ptr(save_mean),
                  _ptr(save_invstd), _ptr(dx), _ptr(tot), m, c, 1 if ctx.training else 0, dtype_code(x_cl.dtype),
                  dtype_code(dy.dtype), _stream())
        return dx, tot[c:], tot[:c], None, None, None, None, None, None


def bn_relu(x_cl, bn, training=None, out_dtype=None, count=True):
    """``relu(batch_norm(x))`` with the parameters / running statistics of a ``torch.nn.BatchNorm3d`` module
    (kept as the parameter container so that the state_dict matches the reference).  The module's own state decides
    what ``nn.BatchNorm3d.forward`` would do (the reference calls the module itself, RepMode.py:212): ``bn.training``
    selects batch or running statistics (``training`` is accepted for the callers that pass the block's mode, and
    must agree), ``momentum=None`` is the cumulative moving average 1 / num_batches_tracked, and a module without
    running statistics always normalises with batch statistics.  ``count=False``: the caller has already advanced
    ``num_batches_tracked`` (the network does it for all its BN layers in one launch)."""
    x_cl = x_cl.contiguous()
    if out_dtype is None:
        out_dtype = x_cl.dtype
    train_mode = bn.training
    use_batch_stats = train_mode or not bn.track_running_stats
    if count and train_mode and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    if not bn.track_running_stats:
        # no running buffers: the kernel still wants two vectors to update -- scratch ones, discarded
        c = x_cl.shape[-1]
        rm = torch.zeros(c, dtype=torch.float32, device=x_cl.device)
        rv = torch.ones(c, dtype=torch.float32, device=x_cl.device)
        momentum = 0.0
    else:
        rm, rv = bn.running_mean, bn.running_var
        if bn.momentum is None:
            # cumulative average (nn.BatchNorm: exponential_average_factor = 1 / num_batches_tracked); one host read,
            # only on this non-default configuration
            momentum = 1.0 / max(int(bn.num_batches_tracked), 1) if train_mode else 0.0
        else:
            momentum = bn.momentum
    if not bn.affine:
        raise _lib.RepModeHipError('bn_relu: BatchNorm without affine parameters is not supported by the HIP kernel')
    return _BnRelu.apply(x_cl, bn.weight, bn.bias, rm, rv, use_batch_stats, momentum, bn.eps, out_dtype)


def k2_weight_frags(weight, rows, red, red_major, dtype, both=False):
    """2x2x2 filter parameter (float, [rows][red][2][2][2] or, ``red_major``, [red][rows][2][2][2]) -> the
    fragment-major operand [8][rowsP/32][redP/KC][32][KC] of the k2s2 kernel in ``dtype`` (one launch).  ``both``:
    also the operand with rows and red exchanged (the stage's data-gradient filter) from the same launch."""
    code = dtype_code(dtype)
    rp, kp = _lib.padded_channels(rows, code, False), _lib.padded_channels(red, code, True)
    out = torch.empty((8, rp, kp), dtype=dtype, device=weight.device)
    if not both:
        _lib.call('repmode_k2_frags', _ptr(weight), rows, red, 1 if red_major else 0, code, _ptr(out), _stream())
        return out
    out_t = torch.empty((8, _lib.padded_channels(red, code, False), _lib.padded_channels(rows, code, True)), dtype=dtype,
                        device=weight.device)
    _lib.call('repmode_k2_frags2', _ptr(weight), rows, red, 1 if red_major else 0, code, _ptr(out), _ptr(out_t), _stream())
    return out, out_t


def k2s2(in_cl, w_frag, cout, scatter):
    """Gather (fine -> coarse) or scatter (coarse -> fine) 2x2x2 stride-2 GEMM, see include/repmode_hip.h."""
    n, a, b, c, cin = in_cl.shape
    d, h, w = (a, b, c) if scatter else (a // 2, b // 2, c // 2)
    oshape = (n, 2 * d, 2 * h, 2 * w, cout) if scatter else (n, d, h, w, cout)
    out = torch.empty(oshape, dtype=in_cl.dtype, device=in_cl.device)
    _lib.call('repmode_k2s2', _ptr(in_cl), _ptr(w_frag), _ptr(out), n, d, h, w, cin, cout, dtype_code(in_cl.dtype),
              1 if scatter else 0, _stream())
    return out


def k2s2_wgrad(coarse_cl, fine_cl, param_layout=0):
    """dw[p, a, b] = sum_m coarse[m, a] * fine[fine(m, p), b] as float [8, A, B] (``param_layout`` 0), or directly in
    a parameter's layout: 1 -> [A, B, 2, 2, 2], 2 -> [B, A, 2, 2, 2].  bf16: HIP kernel; float32: a library GEMM on
    gathered patches (parity mode only)."""
    n, d, h, w, ca = coarse_cl.shape
    cb = fine_cl.shape[-1]
    shape = {0: (8, ca, cb), 1: (ca, cb, 2, 2, 2), 2: (cb, ca, 2, 2, 2)}[param_layout]
    if coarse_cl.dtype == torch.bfloat16 and param_layout != 2:
        # the kernel accumulates tap-major (atomics into the parameter layout, 32-byte stride, measured 5x slower);
        # layout 1 is one small transpose launch behind it
        dw8, pre = ZERO_POOL.take((8, ca, cb), coarse_cl.device)
        _lib.call('repmode_k2s2_wgrad_ex', _ptr(coarse_cl), _ptr(fine_cl), _ptr(dw8), n, d, h, w, ca, cb, 4 if pre else 0,
                  _stream())
        if param_layout == 0:
            return dw8
        dw = torch.empty(shape, dtype=torch.float32, device=coarse_cl.device)
        _lib.call('repmode_tap_transpose', _ptr(dw8), _ptr(dw), ca * cb, 8, _stream())
        return dw
    if coarse_cl.dtype == torch.bfloat16:
        dw = torch.empty(shape, dtype=torch.float32, device=coarse_cl.device)
        _lib.call('repmode_k2s2_wgrad_ex', _ptr(coarse_cl), _ptr(fine_cl), _ptr(dw), n, d, h, w, ca, cb, param_layout,
                  _stream())
        return dw
    g = _gather_patches(fine_cl)                                       # [M, 8*B]
    dw8 = (coarse_cl.view(-1, ca).t() @ g).view(ca, 8, cb).permute(1, 0, 2)   # [8, A, B]
    if param_layout == 0:
        return dw8
    dw = dw8.view(2, 2, 2, ca, cb)
    return (dw.permute(3, 4, 0, 1, 2) if param_layout == 1 else dw.permute(4, 3, 0, 1, 2)).contiguous()


def _gather_patches(x_cl):
    """fine [N,2d,2h,2w,C] -> [N*d*h*w, 8*C], taps ordered (pz, py, px) -- only the weight gradients need it."""
    n, a, b, c, ch = x_cl.shape
    return x_cl.view(n, a // 2, 2, b // 2, 2, c // 2, 2, ch).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, 8 * ch)


class _Down2(torch.autograd.Function):
    """Conv3d(C, C, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:81)."""

    @staticmethod
    def forward(ctx, x_cl, weight):
        _require_hip(x_cl, 'input')
        co, ci = weight.shape[:2]
        if ctx.needs_input_grad[0]:
            wf, wb = k2_weight_frags(weight, co, ci, False, x_cl.dtype, both=True)   # forward + data-gradient filters
        else:
            wf, wb = k2_weight_frags(weight, co, ci, False, x_cl.dtype), None
        ctx.save_for_backward(x_cl, weight, wb)
        return k2s2(x_cl, wf, co, scatter=False)

    @staticmethod
    def backward(ctx, dy):
        x_cl, weight, wb = ctx.saved_tensors
        co, ci = weight.shape[:2]
        dy = dy.to(x_cl.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = k2s2(dy, wb, ci, scatter=True)
        dw = k2s2_wgrad(dy, x_cl, param_layout=1)                         # [Co, Ci, 2, 2, 2]
        return dx, dw


class _Up2(torch.autograd.Function):
    """ConvTranspose3d(Ci, Co, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:98)."""

    @staticmethod
    def forward(ctx, x_cl, weight):
        _require_hip(x_cl, 'input')
        ci, co = weight.shape[:2]
        if ctx.needs_input_grad[0]:
            wf, wb = k2_weight_frags(weight, co, ci, True, x_cl.dtype, both=True)
        else:
            wf, wb = k2_weight_frags(weight, co, ci, True, x_cl.dtype), None
        ctx.save_for_backward(x_cl, weight, wb)
        return k2s2(x_cl, wf, co, scatter=True)

    @staticmethod
    def backward(ctx, dy):
        x_cl, weight, wb = ctx.saved_tensors
        ci, co = weight.shape[:2]
        dy = dy.to(x_cl.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = k2s2(dy, wb, ci, scatter=False)
        dw = k2s2_wgrad(x_cl, dy, param_layout=1)                         # [Ci, Co, 2, 2, 2]
        return dx, dw


def down2(x_cl, weight):
    return _Down2.apply(x_cl.contiguous(), weight)


def up2(x_cl, weight):
    return _Up2.apply(x_cl.contiguous(), weight)


_SLOT_IDS = {}


class _SingleSlot:
    """All samples share one filter: used when the experts themselves are the 'slots'."""

    def __init__(self, n, device, slot=0):
        self.nslots = 1
        self.n = n
        key = (n, slot, str(device))
        if key not in _SLOT_IDS:                      # constant index vectors: built once per shape
            _SLOT_IDS[key] = torch.full((n,), slot, dtype=torch.int32, device=device)
        self.sample_slot = _SLOT_IDS[key]


_ONEHOT2 = {}


def _expert_selector(co, device):
    """g for two pseudo-slots that select the raw experts: slot 0 = conv5x5, slot 1 = zero-padded conv3x3."""
    key = (co, str(device))
    if key not in _ONEHOT2:
        g = torch.zeros((2, NUM_EXPERTS, co), dtype=torch.float32, device=device)
        g[0, 0] = 1.0
        g[1, 1] = 1.0
        _ONEHOT2[key] = g
    return _ONEHOT2[key]


def box_sum(in3=None, in5=None, out=None, add=(), out_dtype=torch.float32):
    """box3(in3) + box5(in5) [+ up to two more float tensors ``add``], stored in ``out_dtype``: zero-padded k^3 box
    means of float channels-last tensors -- the avg-pool experts' spatial part (RepMode.py:139-142, 176-180:
    w1x1 * 1/k^3 broadcast over the k^3 support)."""
    ref = in3 if in3 is not None else in5
    n, d, h, w, c = ref.shape
    if out is None:
        out = torch.empty(ref.shape, dtype=out_dtype, device=ref.device)
    add = list(add) + [None, None]
    _lib.call('repmode_box_sum_ex', _ptr(in3) if in3 is not None else None, _ptr(in5) if in5 is not None else None,
              _ptr(add[0]) if add[0] is not None else None, _ptr(add[1]) if add[1] is not None else None,
              _ptr(out), dtype_code(out.dtype), n, d, h, w, c, _stream())
    return out


def tap_transpose(dw_taps, shape, out=None):
    """Tap-major filter gradient [125, Co, Ci] -> the expert parameter's [Co, Ci, k, k, k] (k = 5: all taps; k = 3:
    the centred 27)."""
    co, ci, k = shape[0], shape[1], shape[2]
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dw_taps.device)
    assert tuple(out.shape) == tuple(shape) and out.dtype == torch.float32 and out.is_contiguous()
    _lib.call('repmode_tap_transpose', _ptr(dw_taps), _ptr(out), co * ci, k ** 3, _stream())
    return out


def gate_bwd(g, dg, slot_task, num_tasks):
    """(dgate_w [5*Co, T], dgate_b [5*Co]) from probabilities g and their gradients dg, both [S, 5, Co]."""
    s_, _, co = g.shape
    dgw = torch.empty((NUM_EXPERTS * co, num_tasks), dtype=torch.float32, device=g.device)
    dgb = torch.empty((NUM_EXPERTS * co,), dtype=torch.float32, device=g.device)
    _lib.call('repmode_gate_bwd', _ptr(g), _ptr(dg), _ptr(slot_task), s_, num_tasks, co, _ptr(dgw), _ptr(dgb), _stream())
    return dgw, dgb


def expert_mix_fwd(p, gn):
    """y = sum_e g[n, e, :] * P_e  (P: float [5, N, D, H, W, Co], gn: float [N, 5, Co])."""
    _, n, d, h, w, co = p.shape
    y = torch.empty((n, d, h, w, co), dtype=torch.float32, device=p.device)
    _lib.call('repmode_expert_mix_fwd', _ptr(p), _ptr(gn), _ptr(y), n, d * h * w, co, _stream())
    return y


def expert_mix_bwd(dy, p, gn, dtype):
    """(dg [N,5,Co], dye_lo [2,N,D,H,W,Co] in ``dtype``, dye_hi [3, M(+pad), Co] float) from dy, the expert outputs and
    g; M = N*D*H*W voxel rows.  dye_hi's three matrices feed batched GEMMs; rocBLAS picks a pathological kernel when
    such a GEMM has exactly 256 x 256 (or 128 x 256) outputs (118 us instead of 18), so for small M each matrix gets
    8 rows of zero padding."""
    _, n, d, h, w, co = p.shape
    m = n * d * h * w
    pad = 8 if m <= 512 else 0
    dg, pre = ZERO_POOL.take((n, NUM_EXPERTS, co), p.device)
    lo = torch.empty((2, n, d, h, w, co), dtype=dtype, device=p.device)
    if pad:
        hi, hpre = ZERO_POOL.take((3, m + pad, co), p.device)
        if not hpre:
            hi[:, m:].zero_()
    else:
        hi = torch.empty((3, m, co), dtype=torch.float32, device=p.device)
    _lib.call('repmode_expert_mix_bwd_ex', _ptr(dy), _ptr(p), _ptr(gn), _ptr(dg), _ptr(lo), _ptr(hi), (m + pad) * co, n,
              d * h * w, co, dtype_code(dtype) | (16 if pre else 0), _stream())
    return dg, lo, hi


class _ModeConv3dUnmerged(torch.autograd.Function):
    """The same MoDE block by linearity of the convolution (property 3):

        y[n] = sum_e g[n, e, :] * conv(x[n], K_e)

    The experts are shared by all samples, so nothing is merged per task: the 5^3 and 3^3 experts are
    laid out once as two pseudo-slots (``expert_frags``) and the HIP conv kernels run them for the
    whole batch; the three 1x1 experts (conv1x1, avg3, avg5) are plain GEMMs on x and its box means.
    Used on the deep levels, where the weights (84 % of the network's parameters) dwarf the
    activations and the per-task merged filters / filter gradients of the merged path are pure HBM
    traffic (measured 10 of 25 ms per step at 8 distinct tasks).  Backward needs no per-task filter
    gradient either: expert gradients come from the gate-scaled output gradient, gate gradients from
    <dy, P_e>.
    """

    @staticmethod
    def forward(ctx, x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan):
        _require_hip(x_cl, 'input')
        co, ci = k5.shape[0], k5.shape[1]
        n = x_cl.shape[0]
        dev = x_cl.device
        gn = gate_softmax_samples(gate_w, gate_b, plan, co)              # g per SAMPLE [N, 5, Co]
        # (forward and data-gradient layouts of the two raw experts from one pass over the weights)
        wf2, wd2 = expert_frags(k5, k3, x_cl.dtype, want_wd=ctx.needs_input_grad[0])
        s0, s1 = _SingleSlot(n, dev, 0), _SingleSlot(n, dev, 1)
        d, h, w = x_cl.shape[1:4]
        p, pre = ZERO_POOL.take((NUM_EXPERTS, n, d, h, w, co), dev)                       # expert outputs P_e
        xb = torch.empty((3, n, d, h, w, ci), dtype=torch.float32, device=dev)

        def small_experts():
            # the 3^3 expert, and the three 1x1 experts as ONE batched GEMM:
            # [x | box3(x) | box5(x)] @ [K1 | A3 | A5]^T  -> P_2..P_4
            conv5(x_cl, wf2, s1.sample_slot, co, out_f32=True, out=p[1], centre3=True, accumulate=pre)   # 3x3x3 support
            xb[0].copy_(x_cl)
            box_sum(in3=xb[0], out=xb[1])
            box_sum(in5=xb[0], out=xb[2])
            w1 = torch.stack((k1.view(co, ci), a3.view(co, ci), a5.view(co, ci)))         # [3, Co, Ci]
            torch.bmm(xb.view(3, -1, ci), w1.transpose(1, 2), out=p[2:].view(3, -1, co))
            return w1

        # the 5^3 expert's conv on this stream, the four small experts beside it on the second one
        fork = _forks(x_cl)
        if fork:
            main, side = _fork(x_cl)
            with torch.cuda.stream(side):
                w1 = small_experts()
        else:
            w1 = small_experts()
        conv5(x_cl, wf2, s0.sample_slot, co, out_f32=True, out=p[0], accumulate=pre)
        if fork:
            _join(main, side)
        y = expert_mix_fwd(p, gn)
        ctx.save_for_backward(x_cl, k5, k3, k1, a3, a5, gn, xb, w1, p, wd2)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, k5, k3, k1, a3, a5, gn, xb, w1, p, wd2 = ctx.saved_tensors
        plan = ctx.plan
        co, ci = k5.shape[0], k5.shape[1]
        n = x_cl.shape[0]
        dev = x_cl.device
        dt = x_cl.dtype
        dy = dy.float().contiguous()
        # ---- gate: dg[n,e,o] = <dy, P_e>, softmax Jacobian, Linear grads (RepMode.py:198-200)
        dg, d01, dhi = expert_mix_bwd(dy, p, gn, dt)            # <dy, P_e>, and the gate-scaled dy per expert
        dgw, dgb = gate_bwd(gn, dg, plan.sample_task, plan.num_tasks)       # one "slot" per sample
        s0, s1 = _SingleSlot(n, dev, 0), _SingleSlot(n, dev, 1)

        def expert_grads():
            # filter gradients of the gate-scaled dy, all samples in one slot.  Large layers (every workgroup owns its
            # outputs: no atomics) write the parameters' [Co][Ci][taps] layout directly, each wave transposing its tile
            # through LDS; the others accumulate tap-major and are transposed by a second launch.
            one = _SingleSlot(n, dev, 0)
            tiles = ((co + 31) // 32) * ((ci + 31) // 32)
            if dt == torch.bfloat16 and tiles * 5 >= 512:
                dk5 = conv5_wgrad(x_cl, d01[0], one, co, expert_layout=5, out=_grad_out(k5))
            else:
                dk5 = tap_transpose(conv5_wgrad(x_cl, d01[0], one, co)[0], k5.shape, out=_grad_out(k5))
            if dt == torch.bfloat16 and tiles * 3 >= 512:
                dk3 = conv5_wgrad(x_cl, d01[1], one, co, expert_layout=3, out=_grad_out(k3))
            else:
                dk3 = tap_transpose(conv5_wgrad(x_cl, d01[1], one, co, centre3=True)[0], k3.shape, out=_grad_out(k3))
            d1 = torch.bmm(dhi[:, :xb[0].numel() // ci].transpose(1, 2), xb.view(3, -1, ci))   # [3, Co, Ci]
            return dk5, dk3, d1[0].reshape(k1.shape), d1[1].reshape(a3.shape), d1[2].reshape(a5.shape)

        # the expert gradients do not depend on the data gradient: second stream
        fork = ctx.needs_input_grad[0] and _forks(x_cl)
        if fork:
            main, side = _fork(x_cl)
            with torch.cuda.stream(side):
                dk5, dk3, dk1, da3, da5 = expert_grads()
        else:
            dk5, dk3, dk1, da3, da5 = expert_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            dxf = conv5(d01[0], wd2, s0.sample_slot, ci, out_f32=True)
            shp = dxf.shape
            conv5(d01[1], wd2, s1.sample_slot, ci, out_f32=True, out=dxf, centre3=True, accumulate=True)
            # 1x1 experts: one batched GEMM gives the three partial data gradients; the zero-padded box
            # mean is self-adjoint, so the avg experts' parts go back through box3 / box5 -- summed with the
            # two conv parts and cast in the same kernel
            m = dxf.numel() // ci
            t = torch.bmm(dhi, w1)                                                        # [3, M(+pad), Ci]
            tv = [t[e, :m].view(shp) for e in range(3)]
            dx = box_sum(in3=tv[1], in5=tv[2], add=(dxf, tv[0]), out_dtype=dt)
            del wd2
        if fork:
            _join(main, side)
        return dx, dk5, dk3, dk1, da3, da5, dgw, dgb, None


def use_unmerged(x_cl, plan):
    """Heuristic: small volumes (levels 3-4) with several distinct tasks in the batch."""
    return plan.training and plan.nslots > 2 and x_cl.shape[3] <= 8


def mode_conv3d(x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32=False, mode='auto'):
    """The MoDE block up to (not including) BN/ReLU, on a channels-last tensor.

    x_cl: [N, D, H, W, Ci] float32 or bfloat16 (HIP).  Expert / gate parameters: float32, the
    reference's shapes.  Returns [N, D, H, W, Co] in x's dtype (float32 when ``out_f32``, and always
    float32 from the 'unmerged' formulation).  ``mode``: 'merged' (per-task GatRep + one conv), 'unmerged'
    (per-expert convs, see _ModeConv3dUnmerged) or 'auto'.
    """
    x_cl = x_cl.contiguous()
    ps = [p.contiguous() for p in (k5, k3, k1, a3, a5, gate_w, gate_b)]
    for p in ps:
        if p.dtype != torch.float32:
            raise TypeError('MoDE parameters must be float32')
    if plan.n != x_cl.shape[0]:
        raise ValueError('task plan is for %d samples, input has %d' % (plan.n, x_cl.shape[0]))
    if mode == 'auto':
        mode = 'unmerged' if use_unmerged(x_cl, plan) else 'merged'
    if mode == 'unmerged':
        return _ModeConv3dUnmerged.apply(x_cl, *ps, plan)
    return _ModeConv3d.apply(x_cl, *ps, plan, out_f32)
