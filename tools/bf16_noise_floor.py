#!/usr/bin/env python3
"""What a network-level bf16 comparison can resolve (CPU only; writes profiles/r04_bf16_noise.txt when redirected):
the bf16-EMULATING oracle (oracle.Net(emulate=torch.bfloat16): float32 arithmetic between the rounding points of the HIP
path) against ITSELF with every parameter perturbed by 1e-6 relative -- float-summation-order noise.  Block by block in the
forward pass, and per parameter tensor in the backward pass.  The differences are the floor under any comparison of two
correct bf16 implementations of RepMode.py:194-214 (tests/test_bf16_end_to_end_gpu.py calibrates its bounds on it).
    python tools/bf16_noise_floor.py [mult_chan]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from conftest import Opts
from oracle import repmode_oracle as orc

mc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gen = torch.Generator().manual_seed(11)
x = torch.randn(3, 1, 16, 64, 64, generator=gen)
tgt = torch.randn(3, 1, 16, 64, 64, generator=gen)
tasks = torch.tensor([3, 7, 3])


def run(seed):
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=mc, emulate=torch.bfloat16)
    net.train()
    if seed:
        g2 = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=g2))
    acts = {}
    for name, m in net.named_modules():
        if isinstance(m, orc.MoDEConv):
            m.register_forward_hook(lambda mod, i, o, name=name: acts.__setitem__(name, o.detach().clone()))
    y = net(x, tasks)
    torch.nn.functional.mse_loss(y, tgt).backward()
    return acts, {k: p.grad.clone() for k, p in net.named_parameters()}


rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
a0, g0 = run(0)
a1, g1 = run(1)
print('bf16-emulating oracle, mult_chan %d, 3 patches of 16x64x64, tasks 3 7 3: parameters perturbed by 1e-6 relative' % mc)
print('-- forward: relative 2-norm difference of every MoDE block\'s output')
for k in a0:
    print('%-48s %.2e' % (k, rel(a1[k], a0[k])))
print('-- backward: relative 2-norm difference of parameter gradients (expert_conv5x5_conv of every block, in forward order)')
for k in g0:
    if k.endswith('expert_conv5x5_conv'):
        print('%-62s %.3f' % (k, rel(g1[k], g0[k])))
cat = lambda g: torch.cat([v.reshape(-1) for v in g.values()])
print('whole parameter gradient: %.3f   worst tensor: %.3f' % (rel(cat(g1), cat(g0)), max(rel(g1[k], g0[k]) for k in g0)))
