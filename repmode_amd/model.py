"""Training / inference harness around the MI355X MoDE network -- counterpart of the reference's
``fnet/fnet_model.py`` ``Model`` for the rows of SURVEY.md section 8 that call the hot path:

  * ``do_train_iter(signal, target, task)``  -- fnet_model.py:96-132 (zero_grad, autocast forward,
    MSELoss('none') -> mean, backward, Adam step, per-sample loss);
  * ``predict(signal, task, patch_size)``    -- fnet_model.py:149-223 (50 %-overlap tiling, LIFO
    batches of ``opts.batch_size_eval``, Gaussian-weighted blend);
  * ``get_state/save_state/load_state``      -- fnet_model.py:57-94 (same dict keys, so checkpoints
    are interchangeable through ``nn_state``).

Differences that are the point of the build: bfloat16 instead of fp16 autocast (no GradScaler), the
task ids stay on the host until the slot plan is built (no device->host sync per sample), the loss
stays on the device (no ``.item()`` per iteration unless asked), the Gaussian blend accumulates on
the GPU, and data-parallel training is one process per GPU with RCCL all-reduce (DistributedData
Parallel) instead of single-process ``nn.DataParallel`` (fnet_model.py:40-44).
"""
import math
import os

import numpy as np
import torch

from .nn_modules import RepMode as _repmode


def get_gaussian(patch_size, sigma_scale=1.0 / 8):
    """Importance map of fnet_model.py:242-252.

    A centred delta filtered by scipy's separable ``gaussian_filter(mode='constant')`` equals the
    outer product of three 1-D Gaussians truncated at 4 sigma (scipy's default); it is
    max-normalised and zeros are replaced by the smallest non-zero value.
    """
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    gm = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
    gm = (gm / np.max(gm)).astype(np.float32)
    gm[gm == 0] = np.min(gm[gm != 0])
    return gm


def patch_grid(img_size, patch_size, overlap=0.5):
    """(starts, ends) of every patch in enumeration order -- fnet_model.py:156-193."""
    strides = [int(math.ceil(p * (1 - overlap))) for p in patch_size]
    steps = [int(math.ceil((i - p) / s + 1)) for i, p, s in zip(img_size, patch_size, strides)]
    grid = []
    for a in range(steps[0]):
        for b in range(steps[1]):
            for c in range(steps[2]):
                starts = [i * s for i, s in zip((a, b, c), strides)]
                ends = [min(s + p, im) for s, p, im in zip(starts, patch_size, img_size)]
                starts = [max(e - p, 0) for e, p in zip(ends, patch_size)]
                grid.append((starts, ends))
    return grid


class Model(object):
    def __init__(self, opts, nn_module='RepMode', init_weights=True, lr=0.001, criterion_fn=torch.nn.MSELoss,
                 gpu_ids=0, mult_chan=32, dtype=torch.bfloat16, distributed=False):
        self.opts = opts
        self.nn_module = nn_module
        self.lr = lr
        self.count_iter = 0
        self.count_epoch = 0
        self.gpu_ids = [gpu_ids] if isinstance(gpu_ids, int) else gpu_ids
        if self.gpu_ids[0] < 0:
            raise RuntimeError('repmode_amd.Model needs a HIP device; there is no CPU path')
        self.device = torch.device('cuda', self.gpu_ids[0])
        self.patch_size = (32, 128, 128)                       # fnet_model.py:34
        self.mult_chan = mult_chan
        self.dtype = dtype
        self.distributed = distributed
        self.criterion = criterion_fn(reduction='none')        # fnet_model.py:36
        self._init_model()

    def _init_model(self):
        if self.nn_module != 'RepMode':
            raise ValueError('only the RepMode network is provided (got %r)' % self.nn_module)
        self.net = _repmode.Net(self.opts, mult_chan=self.mult_chan, dtype=self.dtype).to(self.device)
        from . import distributed as dist_
        from . import ops as ops_
        self.ddp = self.reducer = None
        if self.distributed in ('reducer', 'reducer-always'):
            # gradients are produced inside the communication buckets and averaged under backward (distributed.py);
            # measured equal to the stock wrapper on one rank, not yet run on eight -> opt-in
            self.reducer = dist_.GradReducer(self.net, always_reduce=self.distributed == 'reducer-always')
        elif self.distributed:
            self.ddp = dist_.wrap_ddp(self.net, self.device)
        # process-wide (one training process per GPU): where the MoDE gradient kernels put the parameter gradients
        ops_.GRAD_SINK = self.reducer
        try:
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=self.lr, fused=True)
        except (RuntimeError, TypeError):
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=self.lr)

    # ---- checkpoint: fnet_model.py:57-94 (same keys)
    def get_state(self):
        return dict(nn_module=self.nn_module, opts=self.opts, nn_state=self.net.state_dict(),
                    optimizer_state=self.optimizer.state_dict(), count_iter=self.count_iter,
                    count_epoch=self.count_epoch)

    def save_state(self, path_save):
        dirname = os.path.dirname(path_save)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
        state = self.get_state()
        state['nn_state'] = {k: v.cpu() for k, v in state['nn_state'].items()}
        torch.save(state, path_save)

    def load_state(self, path_load):
        state = torch.load(path_load, map_location='cpu', weights_only=False)
        self.net.load_state_dict(state['nn_state'])
        if 'optimizer_state' in state:
            self.optimizer.load_state_dict(state['optimizer_state'])
        self.count_iter = state.get('count_iter', 0)
        self.count_epoch = state.get('count_epoch', 0)

    # ---- training step: fnet_model.py:96-132
    def do_train_iter(self, signal, target, task, sync=False):
        """One optimisation step.  ``task`` should be a CPU int tensor (as the DataLoader yields it):
        the slot plan is then built without touching the device.  Returns (output, per-sample loss);
        both stay on the device unless ``sync`` (then they are copied to the host like the reference
        does)."""
        signal = signal.to(self.device, non_blocking=True)
        target = target.to(self.device, non_blocking=True)
        module = self.ddp if self.ddp is not None else self.net
        if not module.training:
            module.train()               # (walks the whole module tree: ~0.6 ms, not needed every step)
        self.optimizer.zero_grad(set_to_none=True)
        if torch.is_tensor(task) and not task.is_cuda:
            task = [int(t) for t in task.tolist()]   # plain ints: DDP's input scatter would move a tensor to the GPU
        output = module(signal, task)
        loss_nomean = self.criterion(output, target)
        loss = torch.mean(loss_nomean)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        self.count_iter += 1
        loss_sample = torch.mean(loss_nomean.detach(), dim=(1, 2, 3, 4))
        self.last_loss = loss.detach()
        if sync:
            return output.detach().cpu(), loss_sample.cpu()
        return output.detach(), loss_sample

    # ---- sliding-window inference: fnet_model.py:149-223
    def predict(self, signal, task, patch_size=None):
        patch_size = tuple(patch_size or self.patch_size)
        signal = signal.to(self.device)
        self.net.eval()
        img_size = signal.shape[-3:]
        gauss = torch.from_numpy(get_gaussian(patch_size)).to(self.device)
        pred_sum = torch.zeros(signal.shape, device=self.device)
        weight_sum = torch.zeros(signal.shape, device=self.device)
        patches = patch_grid(img_size, patch_size)
        task_id = int(task.reshape(-1)[0]) if torch.is_tensor(task) else int(task)
        bs = int(self.opts.batch_size_eval)
        while patches:                                          # LIFO batches, fnet_model.py:196-200
            batch = [patches.pop() for _ in range(min(bs, len(patches)))]
            crops = torch.cat([signal[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] for s, e in batch], dim=0)
            with torch.no_grad():
                out = self.net(crops, [task_id] * len(batch))
            for i, (s, e) in enumerate(batch):
                pred_sum[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] += out[i:i + 1] * gauss
                weight_sum[:, :, s[0]:e[0], s[1]:e[1], s[2]:e[2]] += gauss
        return (pred_sum / weight_sum).cpu()
