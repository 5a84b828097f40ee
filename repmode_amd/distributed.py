"""Data-parallel training across the GPUs of one node: one process per GPU, gradients all-reduced by
RCCL over xGMI (``torch.distributed`` backend "nccl" IS RCCL on ROCm), overlapped with backward.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(fnet_model.py:40-44), which is replaced, not translated.  Samples are independent units of the
MoDE path (each has its own merged filter and its own conv, RepMode.py:183-189, 204-208); the only
couplings are training-mode BatchNorm statistics -- kept per rank, as DataParallel's replicas
would -- and the gradient sum.  So the path shards with exactly one collective per step: an
all-reduce (SUM / world) of the 193 gradient tensors (123,877,633 elements at mult_chan=32), issued
per bucket as soon as the bucket's gradients exist.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce is bound by one link, so
buckets are sized to keep several collectives in flight under the remaining backward work instead
of one large tail transfer: fp32 buckets with a 48 MB cap (a tensor larger than the cap is a bucket of its own: the 5x5x5
experts of levels 3-4 are 16-131 MB each, so DDP builds 6-7 buckets per step; ``comm_info`` reports their sizes).  88 % of the gradient bytes belong
to the deep levels (enc4 / bottleneck / dec4) whose backward finishes in the first third of the
backward pass -- the level 0-1 encoder backward (most of the FLOPs) hides their transfer.

``GradReducer`` (opt-in: ``Model(distributed='reducer')``) is that scheme without DistributedDataParallel's
per-gradient bucket copies: the gradient kernels of the MoDE blocks write straight into their bucket slices
(``ops._grad_out``), autograd adopts those tensors as ``param.grad``, the few remaining gradients (BatchNorm, the
stride-2 convs, the gate) are gathered by one multi-tensor copy per bucket, and the bucket is all-reduced (AVG)
asynchronously as soon as its last gradient exists.  Measured on one rank (tools/ddp_overhead.py): plain step 13.86 ms,
DDP 14.64, GradReducer 14.69, GradReducer without the collectives 13.90 -- i.e. the 0.8 ms a single-rank group adds
is its (pointless) all-reduce kernels, not the 193 copies, and the two schemes are equal; the stock wrapper stays the
default until the reducer has run on eight GPUs.
"""
import os

import torch
import torch.distributed as dist

BUCKET_MB = 48


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradDtypeSwitch:
    """State of the switchable communication hook: ``dtype`` None = float32 buckets, 'bf16' = bfloat16 on the wire.  One
    wrapper for the whole run: the buckets' dtype rule (Model._apply_grad_dtype_rule) flips this instead of building a second
    DistributedDataParallel over the same network (the rebuild broadcast rank 0's parameters AND BatchNorm
    running statistics to every rank, against the per-rank-statistics rule, while the first wrapper's hooks were still alive)."""

    def __init__(self, dtype=None):
        self.dtype = dtype


def _switch_hook(state, bucket):
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    if state.dtype == 'bf16':
        # buckets travel as bfloat16 (half the bytes per xGMI link: a ring all-reduce of 495 MB of float32 gradients is
        # bound by ONE ~153 GB/s link); the sum is formed in bf16 by the collective, the result returns to float32
        return default_hooks.bf16_compress_hook(None, bucket)
    return default_hooks.allreduce_hook(None, bucket)


def wrap_ddp(net, device=None, grad_compress=None):
    """DistributedDataParallel with the settings the MoDE path wants:
      * every parameter gets a gradient every step (all experts and the whole gate matrix take part;
        unused task columns receive exact zeros) -> ``find_unused_parameters=False``;
      * BatchNorm running statistics stay per rank -> ``broadcast_buffers=False``;
      * gradients live inside the communication buckets -> ``gradient_as_bucket_view=True``.
    ``grad_compress``: None = float32 buckets through the wrapper's own all-reduce (the default: what the reference's
    DataParallel + Adam average in, fnet_model.py:40-44, 112); 'bf16' / 'auto' = the switchable hook above, starting as
    bfloat16 / float32 (``ddp.grad_dtype_switch``).
    """
    from torch.nn.parallel import DistributedDataParallel as DDP
    kwargs = dict(broadcast_buffers=False, find_unused_parameters=False, gradient_as_bucket_view=True,
                  bucket_cap_mb=BUCKET_MB)
    if device is not None and device.type == 'cuda':
        ddp = DDP(net, device_ids=[device.index], output_device=device.index, **kwargs)
    else:
        ddp = DDP(net, **kwargs)
    ddp.grad_dtype_switch = None
    if grad_compress in ('bf16', torch.bfloat16, 'auto'):
        ddp.grad_dtype_switch = GradDtypeSwitch('bf16' if grad_compress != 'auto' else None)
        ddp.register_comm_hook(ddp.grad_dtype_switch, _switch_hook)
    elif grad_compress:
        raise ValueError('grad_compress: None, "bf16" or "auto", got %r' % (grad_compress,))
    return ddp


XGMI_LINK_GBPS = 153.0      # one xGMI link, per direction (MI355X: 7 links per GPU, point to point)
# of the measured backward pass (REPMODE_COMPRESS_IF_RING_OVER: another fraction; 0 = always compress)
COMPRESS_IF_RING_OVER = float(os.environ.get('REPMODE_COMPRESS_IF_RING_OVER', '0.6'))
# transports the link model speaks for (REPMODE_GRAD_RULE_BACKENDS=nccl,gloo lets a one-GPU box walk the rule's bf16 branch)
RULE_BACKENDS = tuple(os.environ.get('REPMODE_GRAD_RULE_BACKENDS', 'nccl').split(','))


def ring_allreduce_ms(nbytes, world, link_gbps=XGMI_LINK_GBPS):
    """Per-link bound of a ring all-reduce of ``nbytes`` over ``world`` GPUs: every byte crosses a link 2 (W - 1) / W times."""
    if world <= 1:
        return 0.0
    return 2.0 * (world - 1) / world * nbytes / (link_gbps * 1e9) * 1e3


def pick_grad_dtype(grad_bytes_fp32, world, backward_ms, backend):
    """The rule that chooses the gradient buckets' dtype: bfloat16 when the per-link ring estimate of
    the float32 all-reduce exceeds COMPRESS_IF_RING_OVER of the measured backward pass -- it could no longer hide under it
    -- and the transport is RCCL (the link model is xGMI's; gloo keeps float32).  Returns 'bf16' or None."""
    if backend not in RULE_BACKENDS or world <= 1 or backward_ms <= 0:
        return None
    return 'bf16' if ring_allreduce_ms(grad_bytes_fp32, world) > COMPRESS_IF_RING_OVER * backward_ms else None


def comm_info(model):
    """What the process group actually looks like, for the bench line's ``config.comm``: backend, the ranks a collective
    saw (an all-reduce of ones), bucket size / count and the buckets' dtype."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    dev = model.device
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    nparam = sum(p.numel() for p in model.net.parameters() if p.requires_grad)
    info = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_seen': int(ones.item()),
            'bucket_mb': BUCKET_MB, 'dtype': 'bf16' if model.grad_compress == 'bf16' else 'f32',
            'dtype_rule': getattr(model, 'grad_compress_rule', None), 'grad_bytes_fp32': 4 * nparam,
            'ring_estimate_ms_fp32': ring_allreduce_ms(4 * nparam, dist.get_world_size())}
    if model.reducer is not None:
        info['n_buckets'] = len(model.reducer.buckets)
        info['scheme'] = 'GradReducer'
    elif model.ddp is not None:
        info['scheme'] = 'DistributedDataParallel'
        try:
            # the buckets DDP actually built (bytes each): a tensor larger than the cap travels as a bucket of its own -- the
            # 5x5x5 experts of levels 3-4 are 16-131 MB each, so 495 MB of gradients are 6-7 buckets, not 495 / 48
            data = model.ddp._get_ddp_logging_data()
            sizes = data.get('rebuilt_bucket_sizes') or data.get('bucket_sizes') or ''
            sizes = [int(v) for v in str(sizes).split(',') if v.strip()]
            info['n_buckets'] = len(sizes)
            info['bucket_mbytes'] = [round(v / 2 ** 20, 1) for v in sizes]
        except Exception:                         # (a private accessor: informative only)
            info['n_buckets'] = None
    return info


class _Bucket:
    __slots__ = ('flat', 'entries', 'ready', 'stray', 'work', 'comm')

    def __init__(self):
        self.flat, self.entries, self.ready, self.stray, self.work, self.comm = None, [], 0, [], None, None


class _Entry:
    __slots__ = ('param', 'name', 'bucket', 'off', 'numel', 'shape', 'stride', 'ptr')


class GradReducer:
    """Bucketed, backward-overlapped gradient averaging for one-process-per-GPU data parallelism.

    Parameters are packed (reverse registration order ~ the order backward produces them) into flat float32 buckets
    of ``bucket_mb``; ``param.grad`` lives inside the bucket:
      * ``grad_buffer(param)`` hands a gradient kernel its bucket slice as output buffer (``ops.GRAD_SINK``); autograd
        adopts the returned tensor as ``param.grad`` without a copy;
      * a post-accumulate hook counts a bucket's gradients; gradients that did not land in the bucket are gathered by
        ONE ``torch._foreach_copy_`` and re-pointed at their slice; the complete bucket is all-reduced (AVG) with
        ``async_op=True`` on the collective stream while backward continues;
      * ``finish()`` (after ``backward()``, before ``optimizer.step()``) waits for the collectives.
    Every parameter must receive a gradient in every backward pass (true of the MoDE network: all experts and the
    whole gate matrix take part) -- ``finish()`` raises otherwise; gradients must be cleared with
    ``zero_grad(set_to_none=True)`` between steps (an existing ``.grad`` is accumulated into, never aliased).
    BatchNorm running statistics stay per rank; parameters and buffers are broadcast from rank 0 once.
    """

    ALIGN = 64          # elements: bucket slices start on 256-byte boundaries (the kernels store 16 bytes per lane)

    def __init__(self, net, bucket_mb=BUCKET_MB, group=None, sync_params=True, always_reduce=False, comm_dtype=None):
        """``comm_dtype=torch.bfloat16``: a complete bucket is cast to bf16, all-reduced in bf16 and cast back in ``finish()``
        -- half the bytes per xGMI link for two streaming casts per bucket."""
        self.group = group
        self.comm_dtype = comm_dtype
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.reduce = self.active and (self.world > 1 or always_reduce)
        named = [(k, p) for k, p in net.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError('GradReducer: the module has no trainable parameters')
        dev, dt = named[0][1].device, named[0][1].dtype
        for k, p in named:
            if p.device != dev or p.dtype != dt or not p.is_contiguous():
                raise ValueError('GradReducer: parameters must be contiguous and share one device / dtype (%s)' % k)
        if self.active and self.world > 1 and sync_params:
            with torch.no_grad():
                for t in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(t, 0, group=group)
        backend = dist.get_backend(group) if self.active else None
        self.avg_op = backend == 'nccl'                      # RCCL averages in the collective; gloo: SUM, then divide
        cap = max(1, int(bucket_mb * (1 << 20)) // named[0][1].element_size())
        self.buckets, self.by_ptr, self.by_param = [], {}, {}
        cur, used = _Bucket(), 0
        for k, p in reversed(named):
            e = _Entry()
            e.param, e.name, e.bucket, e.off, e.numel = p, k, cur, used, p.numel()
            e.shape, e.stride = tuple(p.shape), tuple(p.stride())
            cur.entries.append(e)
            used += -(-p.numel() // self.ALIGN) * self.ALIGN
            if used >= cap:
                self._close(cur, used, dev, dt)
                cur, used = _Bucket(), 0
        if cur.entries:
            self._close(cur, used, dev, dt)
        self.fired = []
        self.copied = self.last_copied = 0       # gradients gathered by a copy in the current / the last finished pass
        self._handles = [e.param.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for e in b.entries]

    def _close(self, b, used, dev, dt):
        b.flat = torch.zeros(used, dtype=dt, device=dev)
        base, esz = b.flat.data_ptr(), b.flat.element_size()
        for e in b.entries:
            e.ptr = base + e.off * esz
            self.by_ptr[e.param.data_ptr()] = e
            self.by_param[e.param] = e
        self.buckets.append(b)

    @staticmethod
    def _view(e):
        return e.bucket.flat.as_strided(e.shape, e.stride, e.off)

    def grad_buffer(self, param):
        """The bucket slice of ``param`` (matched by storage address: autograd hands backward() an alias of the
        parameter) as a fresh tensor, or None when the parameter is not managed / already holds a gradient."""
        e = self.by_ptr.get(param.data_ptr())
        if e is None or e.param.grad is not None or tuple(param.shape) != e.shape:
            return None
        return self._view(e)

    def _on_grad(self, param):
        e = self.by_param[param]
        b = e.bucket
        if param.grad.data_ptr() != e.ptr:
            b.stray.append(e)
        b.ready += 1
        if b.ready == len(b.entries):
            self._fire(b)

    def _fire(self, b):
        if b.stray:
            self.copied += len(b.stray)
            views = [self._view(e) for e in b.stray]
            torch._foreach_copy_(views, [e.param.grad for e in b.stray])
            for e, v in zip(b.stray, views):
                e.param.grad = v
            b.stray = []
        if self.reduce:
            op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
            buf = b.flat
            if self.comm_dtype is not None and self.comm_dtype != b.flat.dtype:
                if b.comm is None:
                    b.comm = torch.empty_like(b.flat, dtype=self.comm_dtype)
                b.comm.copy_(b.flat)
                buf = b.comm
            b.work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        b.ready = 0
        self.fired.append(b)

    def finish(self):
        """Wait for this backward pass's collectives; gradients are the across-rank averages afterwards."""
        if len(self.fired) != len(self.buckets):
            done = set(id(b) for b in self.fired)
            missing = [e.name for b in self.buckets if id(b) not in done for e in b.entries if e.param.grad is None]
            self.fired, self.copied = [], 0
            for b in self.buckets:
                b.ready, b.stray = 0, []
            raise RuntimeError('GradReducer: %d parameters received no gradient in this backward pass (e.g. %s); '
                               'every parameter must take part in every step' % (len(missing), ', '.join(missing[:4])))
        for b in self.fired:
            if b.work is not None:
                b.work.wait()
                b.work = None
                if b.comm is not None:
                    b.flat.copy_(b.comm)
                if not self.avg_op and self.world > 1:
                    b.flat.div_(self.world)
        self.fired = []
        self.last_copied, self.copied = self.copied, 0

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def shard_batch(global_batch, rank, world):
    """Contiguous per-rank slice [lo, hi) of a global batch (weak scaling keeps hi-lo fixed)."""
    per = global_batch // world
    return rank * per, (rank + 1) * per


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (used for step timing)."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
