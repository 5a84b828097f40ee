"""Device-resident training volumes and the batch sampler of the hot path's input side.

The reference keeps every (signal, target) volume pair in host memory, crops / flips one sample at a time in
``SSPDataset.__getitem__`` -> ``data_aug`` (fnet/data/SSPdataset.py:117-155) on DataLoader workers, collates, and copies
the batch to the GPU every iteration.  At 14 ms per step that host pipeline is the bottleneck, so here the volumes live
in HBM (a whole training set of this kind is a few GB; the GPU has 288) and ONE kernel launch crops and flips a whole
batch (``repmode_crop_flip``, csrc/pipeline.hip).  The random decisions stay on the host and follow the reference's
numpy call order exactly, so a seeded run picks the same crops and flips as the reference's ``data_aug``:

    per sample:  np.random.randint(0, size - patch + 1) for z, y, x      (SSPdataset.py:141-144)
                 np.random.uniform(0, 1, size=3) <= random_flip_prob     (SSPdataset.py:150-151)

Only what touches the train step is here; file formats, the dataset split and the CSV bookkeeping of the reference's
data package are out of scope.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def draw_augmentation(img_size, patch_size, random_flip_prob, rng=np.random):
    """(starts[3], flip mask) of one sample, consuming ``rng`` exactly as ``data_aug`` does (SSPdataset.py:137-155).
    Flip mask: bit 0 = z, bit 1 = y, bit 2 = x (``torch.flip`` dims 1, 2, 3 of the [1, D, H, W] sample)."""
    starts = [int(rng.randint(0, i - c + 1)) for i, c in zip(img_size, patch_size)]
    p = rng.uniform(0, 1, size=3)
    mask = 0
    for axis in range(3):
        if p[axis] <= random_flip_prob:
            mask |= 1 << axis
    return starts, mask


class DeviceVolumes:
    """A training set resident on the GPU: ``add(signal, target, task)`` uploads a volume pair once; ``sample_batch`` returns
    (signal [N,1,pd,ph,pw], target, task ids) cropped and flipped on the device by one launch."""

    def __init__(self, device, patch_size=(32, 64, 64), random_flip_prob=0.5):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.RepModeHipError('DeviceVolumes keeps the volumes in HBM; there is no CPU path')
        self.patch_size = tuple(int(v) for v in patch_size)
        self.random_flip_prob = float(random_flip_prob)
        self.signal, self.target, self.task = [], [], []

    def add(self, signal, target, task):
        """signal / target: [D, H, W] or [1, D, H, W] arrays or tensors (already normalised, as the reference stores them)."""
        s = torch.as_tensor(np.asarray(signal) if not torch.is_tensor(signal) else signal, dtype=torch.float32)
        t = torch.as_tensor(np.asarray(target) if not torch.is_tensor(target) else target, dtype=torch.float32)
        s, t = s.reshape(s.shape[-3:]), t.reshape(t.shape[-3:])
        if s.shape != t.shape:
            raise ValueError('signal %s and target %s differ in shape' % (tuple(s.shape), tuple(t.shape)))
        if any(a < p for a, p in zip(s.shape, self.patch_size)):
            raise ValueError('volume %s is smaller than the patch %s' % (tuple(s.shape), self.patch_size))
        self.signal.append(s.contiguous().to(self.device))
        self.target.append(t.contiguous().to(self.device))
        self.task.append(int(task))

    def __len__(self):
        return len(self.signal)

    def crop_flip(self, indices, starts, flips):
        """The batch for explicit (volume index, starts, flip mask) triples: one ``repmode_crop_flip`` launch."""
        n = len(indices)
        pd, ph, pw = self.patch_size
        sig = torch.empty((n, 1, pd, ph, pw), dtype=torch.float32, device=self.device)
        tgt = torch.empty_like(sig)
        ptr_t = ctypes.c_void_p * n
        sv = ptr_t(*[self.signal[i].data_ptr() for i in indices])
        tv = ptr_t(*[self.target[i].data_ptr() for i in indices])
        dims = (ctypes.c_int * (3 * n))(*[int(v) for i in indices for v in self.signal[i].shape])
        st = (ctypes.c_int * (3 * n))(*[int(v) for s in starts for v in s])
        fl = (ctypes.c_int * n)(*[int(f) for f in flips])
        stream = torch._C._cuda_getCurrentRawStream(self.device.index if self.device.index is not None else torch.cuda.current_device())
        for lo in range(0, n, 32):          # REPMODE_CROP_MAX_SAMPLES per launch
            hi = min(n, lo + 32)
            _lib.call('repmode_crop_flip', ctypes.byref(sv, lo * ctypes.sizeof(ctypes.c_void_p)),
                      ctypes.byref(tv, lo * ctypes.sizeof(ctypes.c_void_p)), ctypes.byref(dims, 3 * lo * 4),
                      ctypes.byref(st, 3 * lo * 4), ctypes.byref(fl, lo * 4), hi - lo, pd, ph, pw,
                      sig[lo:hi].data_ptr(), tgt[lo:hi].data_ptr(), stream)
        return sig, tgt

    def sample_batch(self, indices, rng=np.random, augment=True):
        """(signal, target, task) for the volumes ``indices`` (what a DataLoader batch sampler would yield), augmented like
        the reference's training set: per sample, in order, the random crop then the flips."""
        starts, flips = [], []
        for i in indices:
            if augment:
                s, f = draw_augmentation(self.signal[i].shape, self.patch_size, self.random_flip_prob, rng)
            else:
                s, f = [0, 0, 0], 0
            starts.append(s)
            flips.append(f)
        sig, tgt = self.crop_flip(indices, starts, flips)
        return sig, tgt, torch.tensor([self.task[i] for i in indices], dtype=torch.int64)
