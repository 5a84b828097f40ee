#!/usr/bin/env python3
"""Where the HOST time of a train step goes: host-only enqueue time of a forward / a whole step with the GPU idle at the
start (so nothing can block on a full queue), the raw cost of one kernel launch and of one allocation on this box, and a
cProfile of a few steps.
    python tools/host_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repmode_amd import ops
from repmode_amd.model import Model
m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
x = torch.randn(8, 1, 32, 64, 64, device='cuda'); t = torch.randn(8, 1, 32, 64, 64, device='cuda')
task = torch.arange(8) % 12
for _ in range(5): m.do_train_iter(x, t, task)
torch.cuda.synchronize()
T = ops.torch_ops()
print('launch cost %.1f us (1000 tiny launches, C++ loop over the C ABI), small alloc %.2f us' % (T.debug_launch_cost(x, 1000), T.debug_alloc_cost(x, 1000)))
torch.cuda.synchronize()
for name, fn in (('train step', lambda: m.do_train_iter(x, t, task)),):
    hs = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print('%s: host enqueue %.2f ms (GPU idle at start; min of 6: %.2f)' % (name, 1e3 * sum(hs) / len(hs), 1e3 * min(hs)))
m.net.train()
with torch.no_grad():
    hs = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.net(x, task); hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print('forward (no_grad): host enqueue %.2f ms (min %.2f)' % (1e3 * sum(hs) / len(hs), 1e3 * min(hs)))
# pieces of the step
import torch.nn.functional as F
hs = {'zero_grad': [], 'forward': [], 'loss': [], 'backward': [], 'adam': []}
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.optimizer.zero_grad(set_to_none=True); t1 = time.perf_counter()
    out = m.net(x, task); t2 = time.perf_counter()
    ln = m.criterion(out, t); loss = ln.mean(); t3 = time.perf_counter()
    loss.backward(); t4 = time.perf_counter()
    m.optimizer.step(); t5 = time.perf_counter()
    for k, v in zip(hs, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): hs[k].append(v)
torch.cuda.synchronize()
print('host ms per piece (GPU idle at start): ' + ', '.join('%s %.2f' % (k, 1e3 * min(v)) for k, v in hs.items()))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.do_train_iter(x, t, task)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
