#!/usr/bin/env python3
"""Where do bf16 gradients of the whole network stand against the oracle's float32 autograd?  One train-mode forward +
backward of the mult_chan-32 network on the G4b inputs (three 16x64x64 patches, tasks 3, 7, 3) from the reference's seed-0
state: per parameter tensor the norm ratio, cosine and 2-norm relative error of the HIP path in float32 and in bfloat16,
grouped by block in backward order (the loss end first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import Opts, load_golden
from oracle import repmode_oracle as orc
from repmode_amd.nn_modules.RepMode import Net

g = load_golden('g4b_model_train_iter.npz')
tasks = torch.from_numpy(g['tasks'])
x, t = torch.from_numpy(g['xs'][0]), torch.from_numpy(g['targets'][0])
torch.manual_seed(0)
ref = orc.Net(Opts(), mult_chan=32).train()
torch.set_num_threads(min(32, os.cpu_count() or 1))
torch.nn.functional.mse_loss(ref(x, tasks), t).backward()
refg = {k: p.grad for k, p in ref.named_parameters()}
res = {}
for dt in (torch.float32, torch.bfloat16):
    net = Net(Opts(), mult_chan=32, dtype=dt)
    net.load_state_dict(ref.state_dict())
    net.to('cuda:0').train()
    torch.nn.functional.mse_loss(net(x.cuda(), tasks), t.cuda()).backward()
    res[dt] = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
    del net


def stats(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(a.norm() / b.norm()), float(a @ b / (a.norm() * b.norm())), float((a - b).norm() / b.norm())


print('%-58s %8s | f32: ratio cos rel2 | bf16: ratio cos rel2' % ('parameter', 'numel'))
for k in reversed(list(refg)):
    f, b = stats(res[torch.float32][k], refg[k]), stats(res[torch.bfloat16][k], refg[k])
    print('%-58s %8d | %6.3f %7.4f %8.1e | %6.3f %7.4f %8.1e' % (k, refg[k].numel(), *f, *b))
for dt in res:
    a = torch.cat([res[dt][k].reshape(-1) for k in refg]); b = torch.cat([refg[k].reshape(-1) for k in refg])
    print(dt, 'whole gradient: ratio %.3f cos %.4f rel2 %.2e' % stats(a, b))
