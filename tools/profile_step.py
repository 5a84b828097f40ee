#!/usr/bin/env python3
"""torch.profiler view of the train step (host ops + device kernels), for finding small-op overheads.
    python tools/profile_step.py > gpurun_out/torch_profile.txt
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from repmode_amd.model import Model

opts = bench.Opts()
torch.manual_seed(0)
m = Model(opts, lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
x = torch.randn(8, 1, 32, 64, 64, device='cuda')
t = torch.randn(8, 1, 32, 64, 64, device='cuda')
task = torch.arange(8) % 12
for _ in range(3):
    m.do_train_iter(x, t, task)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=False) as prof:
    for _ in range(2):
        m.do_train_iter(x, t, task)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=70))
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=70))
