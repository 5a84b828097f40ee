"""The optimizer of the train step -- ``torch.optim.Adam(net.parameters(), lr)`` of the reference harness
(fnet/fnet_model.py:55, stepped at :112) -- running through this build's own kernels (csrc/adam.hip).

``Adam`` IS a ``torch.optim.Adam`` (same constructor defaults, same ``state`` layout: ``step`` / ``exp_avg`` / ``exp_avg_sq``
per parameter, so ``state_dict()`` / ``load_state_dict()`` interchange with the stock optimizer and with the reference's
checkpoints, fnet_model.py:57-65); only ``step()`` differs: one call into the operator library, which updates every
parameter with two kernel families and -- for the 5x5x5 / 3x3x3 experts of the blocks that run the per-expert formulation --
emits the updated experts as the convolution kernels' bf16 operands in the same pass (``repmode_adam_expert_frags``), so the
next forward pass has no layout launch.  No CPU path: parameters must live on the HIP device.
"""
import torch


def _foreign_step_hook(optimizer, args, kwargs):
    """Any OTHER optimizer stepping (the reference harness's own ``torch.optim.Adam``, a stock fused one): its foreach / fused
    kernels write the parameters without moving autograd's version counters (measured: torch's fused Adam left the expert
    operands kept across steps looking current -- stale filters, a 4 % worse loss after 55 steps), so the operator library's
    store of expert operands is emptied and the next forward pass lays them out again."""
    if not isinstance(optimizer, Adam):
        from . import _lib
        if _lib.torch_ops_loaded():
            torch.ops.repmode.clear_frag_store()


_hook_handle = None


def install_foreign_step_hook():
    """Registered once, when the operator library is loaded (``_lib.load_torch_ops``)."""
    global _hook_handle
    if _hook_handle is None:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        _hook_handle = register_optimizer_step_post_hook(_foreign_step_hook)


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """``capturable``: the step count ALSO lives on the device (one int64 for the whole parameter list) and the kernels take
        the step's constants from device memory (``repmode_adam_hyper_dev``), so that ``step()`` can be part of a captured HIP
        graph: a replay advances the count itself.  The host counters of ``state`` are brought up to date by ``sync_steps()``
        (``state_dict()`` calls it).  The learning rate is a launch argument: changing it needs a new capture."""
        # (the stock single-tensor bookkeeping: `step` counters are host scalars -- no device read to learn the step count)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False,
                         capturable=False)
        self.device_step = bool(capturable)
        self._step_dev = self._hyper_dev = None

    def sync_steps(self):
        """Device step count -> the host counters of ``state`` (after graph replays, which run no host code).  One device read."""
        if self._step_dev is not None:
            t = int(self._step_dev.item())
            for st in self.state.values():
                if 'step' in st:
                    st['step'] = torch.tensor(float(t))

    def state_dict(self):
        self.sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """As torch.optim.Adam; `step` counters that a fused / capturable optimizer kept on the device come to the host (one
        copy at load time, so that ``step()`` never reads the device to learn the step count)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:            # (a stock optimizer's groups say fused / capturable; this one's bookkeeping is neither)
            group['fused'], group['capturable'], group['foreach'] = False, False, False
        for st in self.state.values():
            if torch.is_tensor(st.get('step')) and st['step'].is_cuda:
                st['step'] = st['step'].detach().cpu()
        if self._step_dev is not None:
            # A captured step holds the device count and the constants' buffer BY ADDRESS: they stay where they
            # are and take the loaded count in place.  Counters that disagree cannot be one device count: the buffers go, and the
            # next step says so (as a first step would).
            loaded = {int(st['step']) for st in self.state.values() if 'step' in st}
            if len(loaded) <= 1:
                self._step_dev.fill_(loaded.pop() if loaded else 0)
            else:
                self._step_dev = self._hyper_dev = None

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops as ops_
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group['weight_decay'] != 0 or group['amsgrad'] or group.get('maximize', False):
                raise RuntimeError('repmode_amd.optim.Adam: the reference trains with plain Adam (no weight decay / amsgrad / maximize)')
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            beta1, beta2 = group['betas']
            lr = float(group['lr'])
            if self.device_step:
                if len(self.param_groups) != 1:
                    raise RuntimeError('repmode_amd.optim.Adam(capturable=True): one parameter group (one device step count)')
                if self._step_dev is None:
                    # first step (launch by launch, never under capture): the device count starts where the host counters are
                    t0 = {int(s) for s in steps}
                    if len(t0) != 1:
                        raise RuntimeError('repmode_amd.optim.Adam(capturable=True): every parameter must have seen the same '
                                           'number of steps')
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError('repmode_amd.optim.Adam(capturable=True): run one step launch by launch before capturing')
                    self._step_dev = torch.full((1,), t0.pop(), dtype=torch.int64, device=params[0].device)
                    self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=params[0].device)
                for s in steps:
                    s += 1              # (host mirror: exact while steps run launch by launch; sync_steps() after replays)
                ops_.torch_ops().adam_step_dev(params, grads, exp_avgs, exp_avg_sqs, lr, float(beta1), float(beta2), float(group['eps']),
                                               self._step_dev, self._hyper_dev)
                continue
            # parameters that have seen the same number of updates go together (normally: all of them)
            by_step = {}
            for i, s in enumerate(steps):
                if s.is_cuda:                                       # (a counter created under another optimizer's policy)
                    s = self.state[params[i]]['step'] = s.detach().cpu()
                s += 1                                              # (host scalar tensor, in place: the state's counter)
                by_step.setdefault(int(s), []).append(i)
            for t, idx in by_step.items():
                sel = (lambda xs: xs) if len(idx) == len(params) else (lambda xs: [xs[i] for i in idx])
                ops_.torch_ops().adam_step(sel(params), sel(grads), sel(exp_avgs), sel(exp_avg_sqs), lr, float(beta1), float(beta2),
                                           float(group['eps']), t)
        return loss
