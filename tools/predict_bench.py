#!/usr/bin/env python3
"""BASELINE configs[4]: whole-volume sliding-window inference on a synthetic 64x624x924 Z-stack (patch 32x128x128,
50 % overlap, batches of 8, Gaussian blend) through repmode_amd.model.Model.predict.
    python tools/predict_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repmode_amd.model import Model, patch_grid
opts = bench.Opts(); opts.batch_size_eval = 8
torch.manual_seed(0)
m = Model(opts, lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
vol = torch.randn(1, 1, 64, 624, 924)
npatch = len(patch_grid(vol.shape[-3:], (32, 128, 128)))
small = torch.randn(1, 1, 32, 128, 256)
m.predict(small, torch.tensor([3]), patch_size=(32, 128, 128))          # warm-up (allocator, kernels)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = m.predict(vol, torch.tensor([3]), patch_size=(32, 128, 128))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('predict 64x624x924: %d patches, %.2f s -> %.1f M output voxels/s, %.1f M computed voxels/s (finite: %s)' % (
    npatch, dt, vol.numel() / dt / 1e6, npatch * 32 * 128 * 128 / dt / 1e6, bool(torch.isfinite(out).all())))
