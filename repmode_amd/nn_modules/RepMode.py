"""MI355X-native counterpart of the reference plug-in ``fnet/nn_modules/RepMode.py``.

Same boundary as the reference module:
  * ``Net(opts, mult_chan=32, in_channels=1, out_channels=1)`` -- RepMode.py:8-15; ``opts`` needs
    ``.adopted_datasets`` (len = number of tasks) and ``.gpu_ids``;
  * ``net(signal[N,1,D,H,W] float, task int64[N]) -> [N,1,D,H,W]`` -- RepMode.py:51-71;
  * the same module tree / parameter names, hence the same 309-key ``state_dict``: reference
    checkpoints load unchanged.

Different inside: activations are channels-last (NDHWC) in HBM, tasks are grouped into slots and the
gate/GatRep/convolution (forward and backward) of every MoDE block run in the HIP kernels of
``librepmode_hip.so`` through ``repmode_amd.ops``.  No CPU path: a CPU tensor raises.
"""
import math
import os

import torch

from .. import ops

NUM_EXPERTS = 5


def _resolve_dtype(x, override):
    """float32 (exact-f32 MFMA, parity mode) or bfloat16 (throughput mode).

    An explicit override (ctor argument or REPMODE_AMD_DTYPE) wins; inside ``torch.autocast`` the
    block computes in bfloat16 (the reference trains under fp16 autocast, fnet_model.py:106; bf16
    needs no loss scaling); otherwise the input's dtype.
    """
    if override is not None:
        return override
    env = os.environ.get('REPMODE_AMD_DTYPE')
    if env:
        return {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'f32': torch.float32, 'float32': torch.float32}[env]
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def _to_cl(x):
    """logical NCDHW (any strides) -> contiguous NDHWC; free when x is already channels_last_3d."""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _from_cl(x_cl):
    """contiguous NDHWC -> logical NCDHW view (channels_last_3d strides)."""
    return x_cl.permute(0, 4, 1, 2, 3)


def _kaiming_param(*shape):
    w = torch.nn.Parameter(torch.empty(*shape))
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # RepMode.py:156-159
    return w


class MoDEConv(torch.nn.Module):
    """RepMode.py:123-214.  ``t`` may be a ``TaskPlan``, int task ids [N] or one-hot rows [N, T]."""

    def __init__(self, num_experts, num_tasks, in_chan, out_chan, kernel_size=5, stride=1, padding='same',
                 conv_type='normal', dtype=None):
        super().__init__()
        if num_experts != NUM_EXPERTS or kernel_size != 5 or stride != 1 or padding != 'same':
            raise ValueError('the MoDE block is 5 experts on a 5x5x5 stride-1 same-padded filter (RepMode.py:22,114)')
        assert conv_type in ['normal', 'final']
        self.num_experts, self.num_tasks = num_experts, num_tasks
        self.in_chan, self.out_chan = in_chan, out_chan
        self.kernel_size, self.stride, self.padding, self.conv_type = kernel_size, stride, padding, conv_type
        self.compute_dtype = dtype
        self.expert_conv5x5_conv = _kaiming_param(out_chan, in_chan, 5, 5, 5)
        self.expert_conv3x3_conv = _kaiming_param(out_chan, in_chan, 3, 3, 3)
        self.expert_conv1x1_conv = _kaiming_param(out_chan, in_chan, 1, 1, 1)
        self.register_buffer('expert_avg3x3_pool', torch.full((3, 3, 3), 1.0 / 27))
        self.expert_avg3x3_conv = _kaiming_param(out_chan, in_chan, 1, 1, 1)
        self.register_buffer('expert_avg5x5_pool', torch.full((5, 5, 5), 1.0 / 125))
        self.expert_avg5x5_conv = _kaiming_param(out_chan, in_chan, 1, 1, 1)
        if conv_type == 'normal':
            self.subsequent_layer = torch.nn.Sequential(
                torch.nn.BatchNorm3d(out_chan), torch.nn.ReLU(inplace=True))
        else:
            self.subsequent_layer = torch.nn.Identity()
        self.gate = torch.nn.Linear(num_tasks, num_experts * out_chan, bias=True)

    def forward(self, x, t, x2=None):
        """``x2``: more input channels, to follow x's (a skip connection's ``torch.cat((x, x2), 1)`` that is never built).
        One call into the operator library: cast + channels-last, gate softmax, GatRep, per-slot convolution and -- for a
        'normal' block -- BatchNorm3d + ReLU (RepMode.py:194-214)."""
        plan = t if isinstance(t, ops.TaskPlan) else ops.TaskPlan(t, self.num_tasks, x.device, self.training)
        dtype = _resolve_dtype(x, self.compute_dtype)
        # float output where a later stage reduces it in f32 anyway: the final layer (the operator adds the deep levels,
        # whose reduction is split over workgroups with f32 atomics: repmode_conv5_elem_out)
        out_f32 = self.conv_type == 'final'
        if self.conv_type == 'normal':
            bn = ops.bn_args(self.subsequent_layer[0], x.device, count=not getattr(t, 'bn_counted', False))
        else:
            bn = (None, None, None, None, False, 0.0, 0.0)
        return ops.torch_ops().mode_block(
            x, x2, self.expert_conv5x5_conv, self.expert_conv3x3_conv, self.expert_conv1x1_conv, self.expert_avg3x3_conv,
            self.expert_avg5x5_conv, self.gate.weight, self.gate.bias, *bn, plan.slot_task, plan.sample_slot, plan.sample_task,
            plan.nslots, plan.num_tasks, plan.training, plan.task0, ops.dtype_code(dtype), out_f32)


class MoDESubNet2Conv(torch.nn.Module):                        # RepMode.py:111-120
    def __init__(self, num_experts, num_tasks, n_in, n_out, dtype=None):
        super().__init__()
        self.conv1 = MoDEConv(num_experts, num_tasks, n_in, n_out, kernel_size=5, padding='same', dtype=dtype)
        self.conv2 = MoDEConv(num_experts, num_tasks, n_out, n_out, kernel_size=5, padding='same', dtype=dtype)

    def forward(self, x, t, x2=None):
        return self.conv2(self.conv1(x, t, x2), t)


class Down2(torch.nn.Module):
    """``Conv3d(C, C, kernel_size=2, stride=2, bias=False)`` (RepMode.py:81) on channels-last data:
    non-overlapping 2x2x2 patches make it a gather GEMM (HIP kernel ``repmode_k2s2``)."""

    def __init__(self, chan):
        super().__init__()
        self.weight = _kaiming_param(chan, chan, 2, 2, 2)      # same shape / init as nn.Conv3d

    def forward(self, x):
        return _from_cl(ops.down2(_to_cl(x), self.weight))

    def forward_bn_relu(self, x, bn, count):
        """The stage with its BatchNorm3d + ReLU (RepMode.py:80-84) in one operator call."""
        return ops.torch_ops().stage2_bn_relu(x, self.weight, *ops.bn_args(bn, x.device, count), False, ops.dtype_code(x.dtype))


class Up2(torch.nn.Module):
    """``ConvTranspose3d(Ci, Co, kernel_size=2, stride=2, bias=False)`` (RepMode.py:98): every input
    voxel emits a disjoint 2x2x2 output patch -> a scatter GEMM (HIP kernel ``repmode_k2s2``)."""

    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.weight = _kaiming_param(in_chan, out_chan, 2, 2, 2)   # same shape / init as nn.ConvTranspose3d

    def forward(self, x):
        return _from_cl(ops.up2(_to_cl(x), self.weight))

    def forward_bn_relu(self, x, bn, count):
        """The stage with its BatchNorm3d + ReLU (RepMode.py:97-101) in one operator call."""
        return ops.torch_ops().stage2_bn_relu(x, self.weight, *ops.bn_args(bn, x.device, count), True, ops.dtype_code(x.dtype))


class MoDEEncoderBlock(torch.nn.Module):                       # RepMode.py:74-89
    def __init__(self, num_experts, num_tasks, in_chan, out_chan, dtype=None):
        super().__init__()
        self.in_chan, self.out_chan = in_chan, out_chan
        self.conv_more = MoDESubNet2Conv(num_experts, num_tasks, in_chan, out_chan, dtype=dtype)
        self.conv_down = torch.nn.Sequential(Down2(out_chan), torch.nn.BatchNorm3d(out_chan),
                                             torch.nn.ReLU(inplace=True))

    def forward(self, x, t):
        x_skip = self.conv_more(x, t)
        # stride-2 conv as a GEMM + BN + ReLU (RepMode.py:81-83)
        return self.conv_down[0].forward_bn_relu(x_skip, self.conv_down[1], not getattr(t, 'bn_counted', False)), x_skip


class MoDEDecoderBlock(torch.nn.Module):                       # RepMode.py:92-108
    def __init__(self, num_experts, num_tasks, in_chan, out_chan, dtype=None):
        super().__init__()
        self.in_chan, self.out_chan = in_chan, out_chan
        self.convt = torch.nn.Sequential(Up2(in_chan, out_chan), torch.nn.BatchNorm3d(out_chan),
                                         torch.nn.ReLU(inplace=True))
        self.conv_less = MoDESubNet2Conv(num_experts, num_tasks, in_chan, out_chan, dtype=dtype)

    def forward(self, x, x_skip, t):
        up = self.convt[0].forward_bn_relu(x, self.convt[1], not getattr(t, 'bn_counted', False))   # RepMode.py:98-100
        return self.conv_less(x_skip, t, up)                   # = cat((x_skip, up), 1): skip first, RepMode.py:106


class Net(torch.nn.Module):
    """RepMode.py:8-71.  ``dtype``: None (follow autocast / input), torch.float32 or torch.bfloat16."""

    def __init__(self, opts, mult_chan=32, in_channels=1, out_channels=1, dtype=None):
        super().__init__()
        self.opts = opts
        self.mult_chan, self.in_channels, self.out_channels = mult_chan, in_channels, out_channels
        self.num_tasks = len(self.opts.adopted_datasets)       # RepMode.py:21
        self.num_experts = NUM_EXPERTS
        self.gpu_ids = [self.opts.gpu_ids] if isinstance(self.opts.gpu_ids, int) else self.opts.gpu_ids
        self.compute_dtype = dtype
        e, t, m = self.num_experts, self.num_tasks, in_channels * mult_chan
        self.encoder_block1 = MoDEEncoderBlock(e, t, in_channels, m, dtype)
        self.encoder_block2 = MoDEEncoderBlock(e, t, m, m * 2, dtype)
        self.encoder_block3 = MoDEEncoderBlock(e, t, m * 2, m * 4, dtype)
        self.encoder_block4 = MoDEEncoderBlock(e, t, m * 4, m * 8, dtype)
        self.bottle_block = MoDESubNet2Conv(e, t, m * 8, m * 16, dtype)
        self.decoder_block4 = MoDEDecoderBlock(e, t, m * 16, m * 8, dtype)
        self.decoder_block3 = MoDEDecoderBlock(e, t, m * 8, m * 4, dtype)
        self.decoder_block2 = MoDEDecoderBlock(e, t, m * 4, m * 2, dtype)
        self.decoder_block1 = MoDEDecoderBlock(e, t, m * 2, m, dtype)
        self.conv_out = MoDEConv(e, t, mult_chan, out_channels, kernel_size=5, padding='same', conv_type='final',
                                 dtype=dtype)

    def forward(self, x, t):
        if any(s % 16 for s in x.shape[-3:]):
            raise ValueError('patch dims must be multiples of 16 (4 stride-2 levels), got %s' % (tuple(x.shape[-3:]),))
        # integer task ids -> slot plan, built once and shared by the 19 MoDE blocks (replaces the
        # one-hot embedding of RepMode.py:44-49,53)
        plan = t if isinstance(t, ops.TaskPlan) else ops.TaskPlan(t, self.num_tasks, x.device, self.training)
        if self.training and torch.is_grad_enabled():
            # one pooled memset per step for the kernels' float accumulation buffers (ops.ZeroPool)
            ops.ZERO_POOL.begin(('train', tuple(x.shape), plan.nslots, str(x.device)), x.device)
        else:
            ops.ZERO_POOL.end()
        if self.training and not plan.bn_counted:
            # num_batches_tracked of the 26 BatchNorm layers: one multi-tensor launch instead of 26
            pairs = getattr(self, '_bn_counters', None)
            # (cached; rebuilt when the buffers were replaced, e.g. by .to(device): the probe is the last BN's buffer)
            if pairs is None or (pairs and pairs[-1][1] is not pairs[-1][0].num_batches_tracked):
                pairs = [(m, m.num_batches_tracked) for m in self.modules()
                         if isinstance(m, torch.nn.BatchNorm3d) and m.track_running_stats]
                object.__setattr__(self, '_bn_counters', pairs)
            # a BatchNorm layer frozen on its own (bn.eval() under a training network) does not count batches
            counters = [c for m, c in pairs if m.training]
            if counters:
                torch._foreach_add_(counters, 1)
            plan.bn_counted = True
        if self.training:
            self._prepare_filters(x, plan)
        x, s1 = self.encoder_block1(x, plan)
        x, s2 = self.encoder_block2(x, plan)
        x, s3 = self.encoder_block3(x, plan)
        x, s4 = self.encoder_block4(x, plan)
        x = self.bottle_block(x, plan)
        x = self.decoder_block4(x, s4, plan)
        x = self.decoder_block3(x, s3, plan)
        x = self.decoder_block2(x, s2, plan)
        x = self.decoder_block1(x, s1, plan)
        y = self.conv_out(x, plan).float()
        if self.training:
            ops.torch_ops().finish_prepared(y)
        return y

    def _prepare_filters(self, x, plan):
        """The forward filters of all 19 MoDE blocks depend on the parameters and the batch's tasks only: their gate softmax +
        GatRep (or expert layout) launches are issued now, on the library's preparation stream, and run beside the first
        convolutions instead of in front of each block (csrc/torch/repmode_ops.cpp: prepare_filters)."""
        blocks = getattr(self, '_mode_blocks', None)
        if blocks is None:
            e, d = 'encoder_block%d', 'decoder_block%d'
            order = [(getattr(self, e % (l + 1)).conv_more, l) for l in range(4)] + [(self.bottle_block, 4)] + \
                    [(getattr(self, d % (l + 1)).conv_less, l) for l in (3, 2, 1, 0)]
            blocks = [(c, l) for sub, l in order for c in (sub.conv1, sub.conv2)] + [(self.conv_out, 0)]
            object.__setattr__(self, '_mode_blocks', blocks)
        grad = torch.is_grad_enabled()
        mods = [b for b, _ in blocks]
        need_dx = [int(grad and (i > 0 or x.requires_grad)) for i in range(len(mods))]
        dtype = _resolve_dtype(x, self.compute_dtype)
        ops.torch_ops().prepare_filters(
            [m.expert_conv5x5_conv for m in mods], [m.expert_conv3x3_conv for m in mods], [m.expert_conv1x1_conv for m in mods],
            [m.expert_avg3x3_conv for m in mods], [m.expert_avg5x5_conv for m in mods], [m.gate.weight for m in mods],
            [m.gate.bias for m in mods], [x.shape[-1] >> l for _, l in blocks], need_dx, plan.slot_task, plan.sample_slot,
            plan.sample_task, plan.nslots, plan.num_tasks, plan.training, ops.dtype_code(dtype))
        # the eight stride-2 stages' operands (RepMode.py:81, 98) from ONE launch as well (one ~5 us layout launch in front of
        # each stage before); a stage takes its entry when it runs
        stages = getattr(self, '_stage_convs', None)
        if stages is None:
            stages = [(m, isinstance(m, Up2)) for m in self.modules() if isinstance(m, (Down2, Up2))]
            object.__setattr__(self, '_stage_convs', stages)
        if stages:
            ops.torch_ops().prepare_stage_filters([m.weight for m, _ in stages], [int(u) for _, u in stages], ops.dtype_code(dtype), grad)
