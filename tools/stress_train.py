#!/usr/bin/env python3
"""Training with a different random task mix every step (1..8 distinct tasks per batch: merged / per-expert path
switches, ZeroPool keys change), to check stability: finite decreasing loss, bounded memory, steady step time.
    python tools/stress_train.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repmode_amd.model import Model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
torch.manual_seed(0)
m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 1, 32, 64, 64, generator=g).cuda()
t = torch.tanh(x * 0.5) + 0.1 * torch.randn(8, 1, 32, 64, 64, generator=g).cuda()
losses, t0 = [], time.perf_counter()
for s in range(steps):
    k = 1 + int(torch.randint(0, 8, (1,), generator=g))                 # number of distinct tasks this step
    pool = torch.randperm(12, generator=g)[:k]
    task = pool[torch.randint(0, k, (8,), generator=g)]
    m.do_train_iter(x, t, task)
    if s % 25 == 24 or s == steps - 1:
        torch.cuda.synchronize()
        losses.append(float(m.last_loss))
        print('step %4d  tasks %-26s loss %.5f  %.1f ms/step  mem %.2f GB (peak %.2f)' % (
            s + 1, sorted(set(task.tolist())), losses[-1], (time.perf_counter() - t0) / 25 * 1e3,
            torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30))
        t0 = time.perf_counter()
assert all(l == l and l < 1e3 for l in losses), losses
assert losses[-1] < losses[0], losses
print('ok: loss %.4f -> %.4f' % (losses[0], losses[-1]))
