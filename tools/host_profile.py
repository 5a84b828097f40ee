#!/usr/bin/env python3
"""Where the HOST time of a train step goes (cProfile over a few steps; the GPU runs behind).
    python tools/host_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repmode_amd.model import Model
m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
x = torch.randn(8, 1, 32, 64, 64, device='cuda'); t = torch.randn(8, 1, 32, 64, 64, device='cuda')
task = torch.arange(8) % 12
for _ in range(5): m.do_train_iter(x, t, task)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): m.do_train_iter(x, t, task)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host issue %.2f ms/step, with final sync %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.do_train_iter(x, t, task)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
