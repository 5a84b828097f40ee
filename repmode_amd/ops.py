"""Operators of the MoDE hot path.

Two layers, both over the C ABI of librepmode_hip.so (include/repmode_hip.h):

  * the OPERATOR SEAM -- ``torch.ops.repmode.*`` (csrc/torch/repmode_ops.cpp, librepmode_torch.so): C++ ops with C++
    autograd that run a whole MoDE block (gate softmax + GatRep + per-slot conv [+ BatchNorm + ReLU]) or a stride-2
    stage per call and choose the formulation per layer.  ``mode_conv3d``, ``mode_conv3d_pair``, ``bn_relu``, ``down2``,
    ``up2`` below are their Python faces; the network (nn_modules/RepMode.py) calls the block-level ops directly.
  * thin wrappers of single kernels through ctypes (``conv5``, ``gatrep_merge``, ``conv5_wgrad`` ...): what the
    kernel-level parity tests and the micro-benchmarks call.

PyTorch is used for device memory, streams and autograd bookkeeping only; every FLOP of the MoDE block runs in the
hand-written HIP kernels.  Tensors handed to the library are channels-last (NDHWC) and contiguous.  There is no CPU /
eager fallback: calling an operator on a non-HIP tensor raises, a missing library raises at load.
"""
import contextlib
import os

import torch

from . import _lib

NUM_EXPERTS = 5
TAPS = 125

_DTYPE_CODE = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_MODE_CODE = {'auto': 0, 'merged': 1, 'unmerged': 2, 'pair': 3}


def dtype_code(dtype):
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError('repmode_amd computes in float32 or bfloat16, got %s' % dtype)


def _ptr(t):
    return t.data_ptr()            # plain int: ctypes converts it for a c_void_p parameter (no wrapper object)


def _stream():
    # raw handle of the calling thread's current stream (the Stream-object route costs ~10 us per call)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _require_hip(t, what):
    if not t.is_cuda:
        raise _lib.RepModeHipError(
            '%s is on %s: repmode_amd runs on MI355X (HIP) tensors only and has no CPU fallback' % (what, t.device))


_TOPS = None


def torch_ops():
    """``torch.ops.repmode`` (loads librepmode_torch.so on first use; raises when it has not been built)."""
    global _TOPS
    if _TOPS is None:
        _lib.load_torch_ops()
        _TOPS = torch.ops.repmode
    return _TOPS


# ---- a second HIP stream for independent launches of one layer.  On the deep levels a conv / filter-gradient launch
# has 128-256 workgroups for 512 slots, and a layer's data gradient, filter gradient and 1x1-expert GEMMs do not depend
# on each other: issued on two streams they share the chip instead of queueing behind each other's tails
# (csrc/torch/repmode_ops.cpp: Fork).  Layers with W <= fork_max_w fork; 0 (REPMODE_FORK_MAX_W unset): never.
def set_fork_max_w(w):
    torch_ops().set_fork_max_w(int(w))


def get_fork_max_w():
    return int(torch_ops().get_fork_max_w())


def set_bn_epilogue(mask):
    """BatchNorm work taken over by the forward conv's epilogue (csrc/conv5_igemm.hip, repmode_conv5_epi), a bit mask:
    1 = eval mode without autograd: the whole BatchNorm + ReLU (scale folded into the merged filter, bias + ReLU in the
    epilogue) on the layers whose conv writes the element-typed tensor (default ON); 2 = training: the batch statistics of
    those layers from the conv launch instead of a statistics pass (default OFF: measured 1.5 % slower per step, see
    csrc/torch/repmode_ops.cpp).  REPMODE_BN_EPILOGUE sets the initial mask."""
    torch_ops().set_bn_epilogue(int(mask))


def set_prepare(on):
    """A training forward pass merges the forward filters of all its merged-formulation blocks with ONE launch up front
    (they depend on parameters and tasks only; default) or block by block (REPMODE_PREPARE=0)."""
    torch_ops().set_prepare(bool(on))


def set_dual_launch(on):
    """Per-expert formulation: the 5x5x5 and the 3x3x3 expert's convolutions as one launch (default) or two."""
    torch_ops().set_dual_launch(bool(on))


def set_deep_conv(on):
    """Per-expert levels: the two experts' convolutions through the uniform-grid kernel of csrc/conv5_deep.hip (default)
    or the general kernel's dual-expert launch (REPMODE_DEEP=0)."""
    torch_ops().set_deep_conv(bool(on))


def set_deep_fwd_min(v):
    """Forward pair through conv5_deep only on the level-3 tile with samples x input channels >= v (0: always)."""
    torch_ops().set_deep_fwd_min(int(v))


def get_deep_conv():
    return bool(torch_ops().get_deep_conv())


def set_deep_mode(mask):
    """Per-expert blocks as ONE launch per direction (csrc/deep_mode.hip) where ``deep_mode_plan`` takes the shape: bit 0 the
    forward, bit 1 the data gradient (default 3; 0 = round 4's five launches; REPMODE_DEEP_MODE)."""
    torch_ops().set_deep_mode(int(mask))


def get_deep_mode():
    return int(torch_ops().get_deep_mode())


def set_deterministic(on):
    """Run-to-run bitwise reproducible results (REPMODE_DETERMINISTIC=1): every float sum of the path gets a fixed order --
    see ``repmode_set_deterministic`` in include/repmode_hip.h; slower (the splits it removes are there for parallelism)."""
    _lib.call('repmode_set_deterministic', 1 if on else 0)


def get_deterministic():
    return bool(_lib.load().repmode_get_deterministic())


def set_reserve_cus(n):
    """CUs the persistent grids (conv5_ws_kernel, the stream-K filter gradient) leave free for a communication kernel beside
    them (``repmode_set_reserve_cus``; also REPMODE_RESERVE_CUS).  Results do not depend on it."""
    _lib.call('repmode_set_reserve_cus', int(n))


def get_reserve_cus():
    return int(_lib.load().repmode_get_reserve_cus())


def set_conv_pipe(mode):
    """The convolution's pipelined one-wave-per-SIMD form on volumes 32 or more voxels wide (csrc/conv5_igemm.hip,
    ``conv5_pipe_kernel`` / ``conv5_ws_kernel``): bit 0 on, bit 1 one channel sub-tile per wave everywhere, bit 2 also on grids
    smaller than the chip (tests), bit 3 the wave-specialised kernel, bit 4 its items along z first, bit 5 row-stationary tap order
    on 32-channel layers; default 57.  0: the two-workgroup form everywhere.
    Results do not depend on bits 0-4; bit 5 changes the float summation order of the taps."""
    _lib.call('repmode_set_conv_pipe', int(mode))


def get_conv_pipe():
    return int(_lib.load().repmode_get_conv_pipe())


def set_wgrad_ws(mode):
    """The bf16 filter gradient's wave-specialised form (csrc/conv5_wgrad.hip): 0 never, 1 where a workgroup has a long tile
    loop, 2 wherever the tile allows, 3 (default) as 1 plus the stream-K form (persistent workgroups over the launch's
    tile-step sequence) on the merged levels' launches that are eligible for it."""
    _lib.call('repmode_set_wgrad_ws', int(mode))


def get_wgrad_ws():
    return int(_lib.load().repmode_get_wgrad_ws())


def set_bn_fused(on):
    """BatchNorm passes as ONE launch with a grid-wide barrier where the tensor fits the grid's registers (csrc/bnrelu.hip):
    1 (default) on, 0 the two-launch passes."""
    _lib.call('repmode_set_bn_fused', int(on))


def get_bn_fused():
    return int(_lib.load().repmode_get_bn_fused())


def set_wgrad_col(mode):
    """The bf16 filter gradient's column-walking form (csrc/conv5_wgrad_col.hip: all 125 taps of a (slot, 16 co, 16 ci) tile
    in one workgroup, a ring of x planes in LDS): 0 never, 1 (default) on the shapes it was measured to win, 2 wherever eligible."""
    _lib.call('repmode_set_wgrad_col', int(mode))


def get_wgrad_col():
    return int(_lib.load().repmode_get_wgrad_col())


def set_wgrad_col_split(q):
    """The column form's tap split: 0 (default) chosen by the library, 1 / 2 / 4 that many workgroups per unit."""
    _lib.call('repmode_set_wgrad_col_split', int(q))


def get_wgrad_col_split():
    return int(_lib.load().repmode_get_wgrad_col_split())


def set_thin_kernels(on):
    """The one-channel first / last layers through their own kernels (csrc/thin_conv.hip; default) or round 2's fold of the x
    taps around the general kernel (REPMODE_THIN=0).  The operator library's switch (the ``thin_conv_*`` wrappers here
    take ``folded``)."""
    torch_ops().set_thin_kernels(bool(on))


def set_overlap(on):
    """Overlap of the HBM-bound GatRep kernels with the convolutions on the library's own streams (the step's filter
    preparation beside the first convolutions, a layer's GatRep backward beside its data-gradient conv).  Off by default
    (REPMODE_OVERLAP=1 turns it on): measured 1.6 % slower than launching in line, see csrc/torch/repmode_ops.cpp."""
    torch_ops().set_overlap(bool(on))


def set_tail_jobs(on):
    """Backward pass: a block's gate backward and gradient-layout transposes ride in the first workgroups of the block's
    data-gradient conv launch (default, csrc/tail_jobs.h) instead of being launched on their own (REPMODE_TAIL=0)."""
    torch_ops().set_tail_jobs(bool(on))


def get_tail_jobs():
    return bool(torch_ops().get_tail_jobs())


def get_overlap():
    return bool(torch_ops().get_overlap())


class ZeroPool:
    """One pre-zeroed buffer per train step for the float accumulation targets of the atomics-based kernels (split-K
    conv outputs, chunked filter gradients): ~60 memset launches per step become one.  The state lives in the operator
    library (csrc/torch/repmode_ops.cpp: ZeroPool); this is its handle.

    The first step with a given key records the requested sizes in order; later steps allocate the total once, hand
    out independent tensors over that storage in the same order and tell the kernels to skip their own clearing.  Any
    divergence from the recorded sequence falls back to plain allocations for the rest of the step.  A fresh buffer is
    allocated every step, so tensors held across steps (saved activations, a returned output) stay valid.
    """

    def __init__(self):
        self._like = {}

    def _like_for(self, device):
        device = torch.device(device)
        t = self._like.get(device)
        if t is None:
            t = self._like[device] = torch.empty(0, device=device)
        return t

    def begin(self, key, device):
        torch_ops().zero_pool_begin(repr(key), self._like_for(device))

    def end(self):
        torch_ops().zero_pool_end()

    def has_plan(self, key):
        return bool(torch_ops().zero_pool_has_plan(repr(key)))

    def take(self, shape, device):
        """(float32 tensor of ``shape``, prezeroed?)"""
        t, pre = torch_ops().zero_pool_take([int(v) for v in shape], self._like_for(device))
        return t, bool(pre)


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
        self.task0 = uniq[0]
        # the three index vectors travel as ONE copy out of pinned memory: a pageable source makes the copy synchronous --
        # the host would wait for the previous step's kernels at the top of every step and lose its lead over the GPU.
        # (Each vector starts on a 16-int boundary of the device buffer.)
        pad = lambda v: v + [0] * (-len(v) % 16)
        packed = torch.tensor(pad(uniq) + pad(slots) + host, dtype=torch.int32)
        dev = torch.device(device)
        if dev.type == 'cuda':
            packed = packed.pin_memory()
        buf = packed.to(dev, non_blocking=True)
        o1 = len(pad(uniq))
        o2 = o1 + len(pad(slots))
        self.slot_task = buf[:len(uniq)]
        self.sample_slot = buf[o1:o1 + len(slots)]
        self.sample_task = buf[o2:o2 + len(host)]   # task id per sample
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


def _ptr_array(ctype, tensors):
    import ctypes
    return (ctype * len(tensors))(*[t.data_ptr() if torch.is_tensor(t) else (t or 0) for t in tensors])


def adam_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step):
    """``torch.optim.Adam``'s update of a list of float tensors, in place, through ``repmode_adam_multi`` (csrc/adam.hip;
    fnet_model.py:55, 112).  ``step``: the 1-based count of this update.  Test-facing: the product calls the ``adam_step`` op."""
    import ctypes
    mx = 40
    for b0 in range(0, len(params), mx):
        sl = slice(b0, b0 + mx)
        n = len(params[sl])
        numel = (ctypes.c_long * n)(*[p.numel() for p in params[sl]])
        _lib.call('repmode_adam_multi', n, _ptr_array(ctypes.c_void_p, params[sl]), _ptr_array(ctypes.c_void_p, grads[sl]),
                  _ptr_array(ctypes.c_void_p, exp_avgs[sl]), _ptr_array(ctypes.c_void_p, exp_avg_sqs[sl]), numel,
                  float(lr), float(beta1), float(beta2), float(eps), int(step), _stream())


def adam_expert_frags(k5s, k3s, lr, beta1, beta2, eps, step, want_wd=True):
    """``repmode_adam_expert_frags``: Adam on the 5x5x5 / 3x3x3 experts of some blocks and their bf16 conv operands in one pass.
    ``k5s`` / ``k3s``: lists of (param, grad, exp_avg, exp_avg_sq).  Returns [(wf, wd)] per block (as ``expert_frags``)."""
    import ctypes
    code = dtype_code(torch.bfloat16)
    outs, cos, cis = [], [], []
    for (p5, _, _, _) in k5s:
        co, ci = p5.shape[0], p5.shape[1]
        wf = torch.empty((2, TAPS, _lib.padded_channels(co, code, False), _lib.padded_channels(ci, code, True)), dtype=torch.bfloat16,
                         device=p5.device)
        wd = torch.empty((2, TAPS, _lib.padded_channels(ci, code, False), _lib.padded_channels(co, code, True)), dtype=torch.bfloat16,
                         device=p5.device) if want_wd else None
        outs.append((wf, wd)); cos.append(co); cis.append(ci)
    n = len(k5s)
    cols = lambda lst, j: _ptr_array(ctypes.c_void_p, [t[j] for t in lst])
    _lib.call('repmode_adam_expert_frags', n, cols(k5s, 0), cols(k5s, 1), cols(k5s, 2), cols(k5s, 3), cols(k3s, 0), cols(k3s, 1),
              cols(k3s, 2), cols(k3s, 3), (ctypes.c_int * n)(*cos), (ctypes.c_int * n)(*cis),
              _ptr_array(ctypes.c_void_p, [o[0] for o in outs]), _ptr_array(ctypes.c_void_p, [o[1] for o in outs]),
              float(lr), float(beta1), float(beta2), float(eps), int(step), _stream())
    return outs


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


def set_grad_sink(reducer):
    """Where the MoDE gradient kernels put parameter gradients: the communication buckets of a
    ``distributed.GradReducer`` (no per-gradient bucket copy -- 193 small kernels per step under
    DistributedDataParallel), or fresh tensors (``None``).  Process-wide: one training process per GPU."""
    if reducer is None:
        torch_ops().grad_sink_clear()
        return
    entries = [e for b in reducer.buckets for e in b.entries]
    torch_ops().grad_sink_set([e.param for e in entries], [e.bucket.flat for e in entries], [int(e.off) for e in entries])


def _grad_out(param):
    """Output buffer for the gradient of ``param``: its slice of the registered reducer's bucket when there is one and
    the parameter holds no gradient yet, else a fresh tensor (what the C++ gradient code calls)."""
    return torch_ops().grad_out(param)


def conv5(x_cl, w, sample_slot, cout, out_f32=False, out=None, centre3=False, accumulate=False, dxc=False):
    """y[n] = x[n] (*) w[sample_slot[n]], 5^3 'same' cross-correlation, NDHWC -- RepMode.py:204-210.
    ``accumulate`` (float ``out`` given): add to ``out`` instead of overwriting it.  ``dxc``: only the centre x tap
    of every (dz, dy) row (thin layers with their x taps folded into channels, see ``thin_conv_*``)."""
    n, d, h, wd_, cin = x_cl.shape
    code = dtype_code(x_cl.dtype)
    out_dtype = torch.float32 if (out_f32 or x_cl.dtype == torch.float32) else x_cl.dtype
    if out is not None:
        y = out
    else:
        y = torch.empty((n, d, h, wd_, cout), dtype=out_dtype, device=x_cl.device)
    assert y.dtype == out_dtype and y.is_contiguous()
    assert not accumulate or out_dtype == torch.float32
    _lib.call('repmode_conv5_ex', _ptr(x_cl), _ptr(w), _ptr(sample_slot), _ptr(y), n, d, h, wd_, cin, cout, code,
              1 if out_dtype == torch.float32 else 0, (1 if centre3 else 0) | (2 if accumulate else 0) | (4 if dxc else 0), _stream())
    return y


def conv5_pair(xa, xb, w, sample_slot, cout, cout1=0):
    """``repmode_conv5_pair`` directly: input channels split over xa | xb (xb None: one tensor), output channels over two
    tensors at ``cout1`` (0: one).  Element-typed output.  Returns (y, y2 or None)."""
    n, d, h, wd_, ca = xa.shape
    cb = xb.shape[-1] if xb is not None else 0
    y = torch.empty((n, d, h, wd_, cout1 if cout1 else cout), dtype=xa.dtype, device=xa.device)
    y2 = torch.empty((n, d, h, wd_, cout - cout1), dtype=xa.dtype, device=xa.device) if cout1 else None
    _lib.call('repmode_conv5_pair', _ptr(xa), _ptr(xb) if xb is not None else None, ca if xb is not None else 0, _ptr(w),
              _ptr(sample_slot), _ptr(y), _ptr(y2) if y2 is not None else None, cout1, n, d, h, wd_, ca + cb, cout,
              dtype_code(xa.dtype), 0, 0, _stream())
    return y, y2


def conv5_deep_supported(x_cl):
    """Whether ``conv5_deep`` takes this input (bf16, x extent <= 8, channels a multiple of 8)."""
    return bool(_lib.load().repmode_conv5_deep_supported(x_cl.shape[3], x_cl.shape[4], dtype_code(x_cl.dtype))) \
        if x_cl.dtype in (torch.float32, torch.bfloat16) else False


def conv5_deep(x_cl, w2, cout, two_in=False, out=None, zeroed=False):
    """The per-expert formulation's two convolutions on a deep level as one uniform grid (csrc/conv5_deep.hip).
    ``w2``: ``expert_frags``' two slots.  Forward form: x [N,...] -> float [2N, ...] (P5, then P3); data-gradient form
    (``two_in``): x [2N, ...] (G5, then G3) -> float [N, ...] = conv(G5, slot 0) + conv(G3, slot 1)."""
    nn, d, h, wd_, cin = x_cl.shape
    n = nn // 2 if two_in else nn
    shape = (n if two_in else 2 * n, d, h, wd_, cout)
    if out is None:
        y = torch.empty(shape, dtype=torch.float32, device=x_cl.device)
    else:
        y = out
        assert tuple(y.shape) == shape and y.dtype == torch.float32 and y.is_contiguous()
    _lib.call('repmode_conv5_deep', _ptr(x_cl), _ptr(w2), _ptr(y), n, d, h, wd_, cin, cout,
              (1 if two_in else 0) | (2 if zeroed else 0), _stream())
    return y


def deep_mode_plan(direction, n, d, h, w, cin, cout, dtype=torch.bfloat16):
    """0: ``deep_mode_fwd`` (direction 0) / ``deep_mode_dgrad`` (1) does not take the shape; 1: plain stores; k > 1: k workgroups
    add into every output element (float outputs, zero on entry)."""
    if dtype != torch.bfloat16:
        return 0
    return int(_lib.load().repmode_deep_mode_plan(direction, n, d, h, w, cin, cout, dtype_code(dtype)))


def box_expand(x_cl):
    """float [3, N, D, H, W, C] = (x, box3(x) / 27, box5(x) / 125): the 1x1 experts' operands (RepMode.py:139-142, 176-180)."""
    n, d, h, w, c = x_cl.shape
    out = torch.empty((3, n, d, h, w, c), dtype=torch.float32, device=x_cl.device)
    _lib.call('repmode_box_expand', _ptr(x_cl), dtype_code(x_cl.dtype), _ptr(out), n, d, h, w, c, _stream())
    return out


def box_pair(in3, in5):
    """(box3(in3) / 27, box5(in5) / 125) of two float channels-last tensors, one launch."""
    n, d, h, w, c = in3.shape
    out = torch.empty((2, n, d, h, w, c), dtype=torch.float32, device=in3.device)
    _lib.call('repmode_box_pair', _ptr(in3), _ptr(in5), _ptr(out[0]), _ptr(out[1]), n, d, h, w, c, _stream())
    return out[0], out[1]


def deep_mode_fwd(x_cl, wf, xs, k1, a3, a5, gn):
    """A per-expert MoDE block's forward in one launch (csrc/deep_mode.hip): x bf16 [N,D,H,W,Ci], ``wf`` = ``expert_frags``'
    forward role, ``xs`` = ``box_expand(x)``, the 1x1 experts' parameters [Co,Ci], ``gn`` [N,5,Co] the gate probabilities per
    sample -> (P [5,N,D,H,W,Co] float, y [N,D,H,W,Co] float)."""
    n, d, h, w, ci = x_cl.shape
    co = gn.shape[2]
    plan = deep_mode_plan(0, n, d, h, w, ci, co)
    alloc = torch.zeros if plan > 1 else torch.empty
    p = alloc((NUM_EXPERTS, n, d, h, w, co), dtype=torch.float32, device=x_cl.device)
    y = alloc((n, d, h, w, co), dtype=torch.float32, device=x_cl.device)
    _lib.call('repmode_deep_mode_fwd', _ptr(x_cl), _ptr(wf), _ptr(xs), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(gn), _ptr(p), _ptr(y),
              n, d, h, w, ci, co, _stream())
    return p, y


def deep_mode_dgrad(lo, wd, s0, s1, s2, k1, a3, a5, cin, out_dtype=torch.bfloat16):
    """A per-expert MoDE block's data gradient in one launch: ``lo`` bf16 [2,N,D,H,W,Co] (the conv experts' gate-scaled output
    gradients), ``wd`` = ``expert_frags``' data-gradient role, s0 / s1 / s2 float [N,D,H,W,Co] = G_2, box3(G_3)/27,
    box5(G_4)/125 -> dx [N,D,H,W,Ci] (float when the plan splits the reduction over workgroups)."""
    _, n, d, h, w, co = lo.shape
    plan = deep_mode_plan(1, n, d, h, w, cin, co)
    if plan > 1:
        dx = torch.zeros((n, d, h, w, cin), dtype=torch.float32, device=lo.device)
    else:
        dx = torch.empty((n, d, h, w, cin), dtype=out_dtype, device=lo.device)
    _lib.call('repmode_deep_mode_dgrad', _ptr(lo), _ptr(wd), _ptr(s0), _ptr(s1), _ptr(s2), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(dx),
              dtype_code(dx.dtype), n, d, h, w, cin, co, _stream())
    return dx


def conv5_merged(x_cl, w2, k1, a3, a5, g, sample_slot, cout):
    """EXPERIMENT (not on the product path): the forward conv of a merged-formulation block with GatRep inside the kernel --
    ``w2`` = ``expert_frags``' two un-merged slots, ``k1 / a3 / a5`` the 1x1 experts' parameters, ``g`` [S, 5, Co] the gate
    probabilities.  bf16 in, float out."""
    n, d, h, wd_, cin = x_cl.shape
    y = torch.empty((n, d, h, wd_, cout), dtype=torch.float32, device=x_cl.device)
    _lib.call('repmode_conv5_merged', _ptr(x_cl), _ptr(w2), _ptr(k1), _ptr(a3), _ptr(a5), _ptr(g), _ptr(sample_slot), _ptr(y),
              n, d, h, wd_, cin, cout, 0, _stream())
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


def thin_conv_in1(x_cl, w, sample_slot, cout, out_f32=False, bias=None, relu=False, folded=False):
    """conv5 for a ONE-channel input (bf16 filter ``w`` [S, 125, CoP, 16]): the first layer's forward and (with the
    data-gradient filter) the last layer's data gradient.  Default: the layer's own kernel (csrc/thin_conv.hip: the 125
    taps are the GEMM's reduction dimension); ``folded``: round 2's form -- the five x taps become channels and the general
    kernel runs 25 instead of 125 taps (csrc/thin.hip)."""
    if folded:
        assert bias is None and not relu
        return conv5(_shift5(x_cl.contiguous()), _thin_pack(w, False), sample_slot, cout, out_f32, dxc=True)
    n, d, h, w_, c1 = x_cl.shape
    assert c1 == 1 and x_cl.dtype == torch.bfloat16 and x_cl.is_contiguous()
    y = torch.empty((n, d, h, w_, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x_cl.device)
    _lib.call('repmode_conv5_thin_in1', _ptr(x_cl), _ptr(w), _ptr(sample_slot), _ptr(y), n, d, h, w_, cout, 1 if out_f32 else 0,
              _ptr(bias) if bias is not None else None, 1 if relu else 0, _stream())
    return y


def thin_conv_out1(x_cl, w, sample_slot, folded=False):
    """conv5 for ONE output channel (bf16 filter ``w`` [S, 125, 32, CiP]).  Default: the layer's own kernel
    (csrc/thin_conv.hip: the 25 (dz, dy) tap rows are the GEMM's row dimension); ``folded``: round 2's form -- the five x
    taps become output rows of the general kernel, then a 5-tap diagonal sum (csrc/thin.hip).  Returns float [N, D, H, W, 1]."""
    n, d, h, w_, cin = x_cl.shape
    if not folded:
        assert x_cl.dtype == torch.bfloat16 and x_cl.is_contiguous()
        y = torch.empty((n, d, h, w_, 1), dtype=torch.float32, device=x_cl.device)
        _lib.call('repmode_conv5_thin_out1', _ptr(x_cl), _ptr(w), _ptr(sample_slot), _ptr(y), n, d, h, w_, cin, _stream())
        return y
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
    dw = torch.empty((plan.nslots, TAPS, cout, cin), dtype=torch.float32, device=x_cl.device)      # (cleared by the call)
    if x_cl.dtype == torch.bfloat16 and (cin == 1) != (cout == 1) and not centre3:
        # thin layer: taps stand in for the missing channel dimension (conv5_wgrad_thin)
        a_t, b_t, c, flip = (dy_cl, x_cl, cout, 0) if cin == 1 else (x_cl, dy_cl, cin, 1)
        _lib.call('repmode_conv5_wgrad_thin', _ptr(a_t), _ptr(b_t), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
                  n, d, h, wd_, c, flip, _stream())
        return dw
    _lib.call('repmode_conv5_wgrad_ex', _ptr(x_cl), _ptr(dy_cl), _ptr(plan.sample_slot), plan.nslots, _ptr(dw),
              n, d, h, wd_, cin, cout, dtype_code(x_cl.dtype), 1 if centre3 else 0, _stream())
    return dw


@contextlib.contextmanager
def eval_filter_cache():
    """Inside this context the merged filter of an eval-mode MoDE block (one slot: RepMode.py:209-210) is computed once
    per (block, task, dtype) instead of once per forward: sliding-window inference re-uses it for every batch of
    patches of a volume.  The parameters must not change inside the context."""
    torch_ops().eval_cache_begin()
    try:
        yield
    finally:
        torch_ops().eval_cache_end()


def _plan_args(plan):
    return (plan.slot_task, plan.sample_slot, plan.sample_task, plan.nslots, plan.num_tasks, plan.training, plan.task0)


def pair_supported(xa_cl, xb_cl, plan):
    """The two-tensor path covers the merged formulation with channel counts on tile boundaries."""
    kc = 16 if xa_cl.dtype == torch.bfloat16 else 8
    return (xa_cl.shape[-1] % 32 == 0 and xb_cl.shape[-1] % kc == 0 and xa_cl.shape[:4] == xb_cl.shape[:4]
            and xa_cl.dtype == xb_cl.dtype and not use_unmerged(xa_cl, plan))


def mode_conv3d_pair(xa_cl, xb_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32=False, force=False):
    """``mode_conv3d`` of the channel concatenation (xa, xb) without building it (a skip connection, RepMode.py:106);
    concatenates where the two-tensor kernels do not apply (per-expert formulation, odd channel counts) unless
    ``force`` (then such a call raises)."""
    return torch_ops().mode_conv3d(xa_cl, xb_cl, k5, k3, k1, a3, a5, gate_w, gate_b, *_plan_args(plan), out_f32,
                                   _MODE_CODE['pair' if force else 'auto'])


def bn_relu(x_cl, bn, training=None, out_dtype=None, count=True):
    """``relu(batch_norm(x))`` with the parameters / running statistics of a ``torch.nn.BatchNorm3d`` module
    (kept as the parameter container so that the state_dict matches the reference).  The module's own state decides
    what ``nn.BatchNorm3d.forward`` would do (the reference calls the module itself, RepMode.py:212): ``bn.training``
    selects batch or running statistics (``training`` is accepted for callers that pass the block's mode),
    ``momentum=None`` is the cumulative moving average 1 / num_batches_tracked, and a module without running
    statistics always normalises with batch statistics.  ``count=False``: the caller has already advanced
    ``num_batches_tracked`` (the network does it for all its BN layers in one launch)."""
    if out_dtype is None:
        out_dtype = x_cl.dtype
    args = bn_args(bn, x_cl.device, count)
    return torch_ops().bn_relu(x_cl, *args, dtype_code(out_dtype))


def bn_args(bn, device, count=True):
    """(weight, bias, running_mean, running_var, batch_stats, momentum, eps) of a BatchNorm3d module for the ops."""
    train_mode = bn.training
    if not bn.affine:
        raise _lib.RepModeHipError('BatchNorm without affine parameters is not supported by the HIP kernel')
    if not bn.track_running_stats:
        # no running buffers: the kernel still wants two vectors to update -- scratch ones, discarded
        c = bn.num_features
        return (bn.weight, bn.bias, torch.zeros(c, dtype=torch.float32, device=device),
                torch.ones(c, dtype=torch.float32, device=device), True, 0.0, bn.eps)
    if count and train_mode:
        bn.num_batches_tracked.add_(1)
    if bn.momentum is None:
        # cumulative average (nn.BatchNorm: exponential_average_factor = 1 / num_batches_tracked); one host read,
        # only on this non-default configuration
        momentum = 1.0 / max(int(bn.num_batches_tracked), 1) if train_mode else 0.0
    else:
        momentum = bn.momentum
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, train_mode, momentum, bn.eps)


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
        dw8 = torch.empty((8, ca, cb), dtype=torch.float32, device=coarse_cl.device)
        _lib.call('repmode_k2s2_wgrad_ex', _ptr(coarse_cl), _ptr(fine_cl), _ptr(dw8), n, d, h, w, ca, cb, 0, _stream())
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


def down2(x_cl, weight):
    """Conv3d(C, C, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:81), with autograd."""
    return torch_ops().down2(x_cl, weight)


def up2(x_cl, weight):
    """ConvTranspose3d(Ci, Co, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:98), with autograd."""
    return torch_ops().up2(x_cl, weight)


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
    """(dg [N,5,Co], dye_lo [2,N,D,H,W,Co] in ``dtype``, dye_hi [3, M, Co] float) from dy, the expert outputs and g;
    M = N*D*H*W voxel rows.  dye_lo: the two conv experts' gate-scaled output gradients (operands of the conv kernels),
    dye_hi: the three 1x1 experts' (operands of repmode_gemm3)."""
    _, n, d, h, w, co = p.shape
    m = n * d * h * w
    dg = torch.empty((n, NUM_EXPERTS, co), dtype=torch.float32, device=p.device)
    lo = torch.empty((2, n, d, h, w, co), dtype=dtype, device=p.device)
    hi = torch.empty((3, m, co), dtype=torch.float32, device=p.device)
    _lib.call('repmode_expert_mix_bwd_ex', _ptr(dy), _ptr(p), _ptr(gn), _ptr(dg), _ptr(lo), _ptr(hi), m * co, n,
              d * h * w, co, dtype_code(dtype), _stream())
    return dg, lo, hi


def use_unmerged(x_cl, plan):
    """Heuristic of the operator library: small volumes (levels 3-4) with several distinct tasks in the batch take the
    per-expert formulation  y[n] = sum_e g[n, e, :] * conv(x[n], K_e)  (linearity, SURVEY.md section 4 property 3):
    the experts are shared by all samples, so nothing is merged per task -- on the deep levels the weights (84 % of the
    parameters) dwarf the activations and per-task merged filters / filter gradients are pure HBM traffic."""
    return plan.training and plan.nslots > 2 and x_cl.shape[3] <= int(torch_ops().get_unmerged_max_w())


def mode_conv3d(x_cl, k5, k3, k1, a3, a5, gate_w, gate_b, plan, out_f32=False, mode='auto'):
    """The MoDE block up to (not including) BN/ReLU, on a channels-last tensor, with autograd (RepMode.py:194-210).

    x_cl: [N, D, H, W, Ci] float32 or bfloat16 (HIP).  Expert / gate parameters: float32, the reference's shapes.
    Returns [N, D, H, W, Co] in x's dtype (float32 when ``out_f32``, and always float32 from the per-expert
    formulation).  ``mode``: 'merged' (per-task GatRep + one conv), 'unmerged' (per-expert convs) or 'auto'.
    """
    return torch_ops().mode_conv3d(x_cl, None, k5, k3, k1, a3, a5, gate_w, gate_b, *_plan_args(plan), out_f32,
                                   _MODE_CODE[mode])


def patch_gather(volume, starts, patch_size):
    """The crops ``volume[s : s + patch]`` of a device-resident float volume [D, H, W] for every origin in ``starts``,
    stacked as [nb, 1, pd, ph, pw] -- one launch per 32 patches (fnet_model.py:196-205 slices and stacks one by one)."""
    import ctypes
    d, h, w = (int(v) for v in volume.shape[-3:])
    vol = volume.reshape(d, h, w)
    if vol.dtype != torch.float32 or not vol.is_contiguous():
        vol = vol.float().contiguous()
    pd, ph, pw = (int(v) for v in patch_size)
    nb = len(starts)
    out = torch.empty((nb, 1, pd, ph, pw), dtype=torch.float32, device=vol.device)
    st = (ctypes.c_int * (3 * nb))(*[int(v) for s in starts for v in s])
    for lo in range(0, nb, 32):             # REPMODE_PATCH_MAX per launch
        hi = min(nb, lo + 32)
        _lib.call('repmode_patch_gather', _ptr(vol), d, h, w, ctypes.byref(st, 3 * lo * 4), hi - lo, pd, ph, pw,
                  _ptr(out[lo:hi]), _stream())
    return out


def patch_blend(out, gauss, starts, pred_sum, weight_sum):
    """``pred_sum[patch] += out[n] * gauss; weight_sum[patch] += gauss`` for the patches of a batch in batch order
    (fnet_model.py:207-217), in place, one launch per 32 patches.  out: [nb, 1, pd, ph, pw] float32 or bf16."""
    import ctypes
    d, h, w = (int(v) for v in pred_sum.shape[-3:])
    pd, ph, pw = (int(v) for v in gauss.shape[-3:])
    nb = len(starts)
    if out.numel() != nb * pd * ph * pw:
        raise ValueError('patch_blend: out %s does not hold %d patches of %s' % (tuple(out.shape), nb, (pd, ph, pw)))
    for t in (pred_sum, weight_sum, gauss):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError('patch_blend: pred_sum, weight_sum and gauss must be contiguous float32 tensors')
    if pred_sum.numel() != d * h * w or weight_sum.numel() != d * h * w:
        raise ValueError('patch_blend: pred_sum / weight_sum must hold ONE volume (batch and channel dims of 1)')
    out = out.contiguous()
    code = dtype_code(out.dtype)
    st = (ctypes.c_int * (3 * nb))(*[int(v) for s in starts for v in s])
    for lo in range(0, nb, 32):
        hi = min(nb, lo + 32)
        _lib.call('repmode_patch_blend', _ptr(out[lo:hi]), code, _ptr(gauss), ctypes.byref(st, 3 * lo * 4), hi - lo, pd, ph, pw,
                  _ptr(pred_sum), _ptr(weight_sum), d, h, w, _stream())
