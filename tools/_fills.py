import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from repmode_amd.model import Model
m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
x = torch.randn(8, 1, 32, 64, 64, device='cuda'); t = torch.randn(8, 1, 32, 64, 64, device='cuda')
task = torch.arange(8) % 12
for _ in range(2): m.do_train_iter(x, t, task)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    m.do_train_iter(x, t, task); torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
names = [e.name for e in evs]
i0 = max(i for i, n in enumerate(names) if n == 'aten::mse_loss')
for e in evs[i0:i0 + 80]:
    ks = [k.name[:60] for k in e.kernels]
    print('%-45s %s' % (e.name[:45], ks[:2]))
