"""CPU check of the host-side mathematics of the 'unmerged' (per-expert, linearity) MoDE formulation in
repmode_amd/ops.py: the HIP kernels are replaced by oracle-based stand-ins (test infrastructure), so
what is verified is the orchestration -- gate scaling, box means, 1x1 experts as GEMMs, gradient
routing -- against the oracle's autograd of the reference arithmetic."""
import pytest
import torch

from oracle import repmode_oracle as orc


@pytest.fixture
def fake_kernels(monkeypatch):
    from repmode_amd import ops

    def gate(gw, gb, plan, co):
        return orc.gate_probs(gw, gb, torch.tensor(plan.slot_task_host), co)

    def gate_samples(gw, gb, plan, co):
        return orc.gate_probs(gw, gb, torch.tensor(plan.tasks_host), co)

    def gate_bwd(g, dg, slot_task, num_tasks):
        # softmax Jacobian + Linear gradients, one "slot" per row of g
        s_, _, co = g.shape
        dl = (g * (dg - (g * dg).sum(dim=1, keepdim=True))).reshape(s_, 5 * co)
        dgw = torch.zeros(5 * co, num_tasks)
        dgw.index_add_(1, slot_task.long(), dl.t().contiguous())
        return dgw, dl.sum(dim=0)

    def tap_t(dw_taps, shape, out=None):
        co, ci, k = shape[0], shape[1], shape[2]
        w = dw_taps.view(5, 5, 5, co, ci)
        if k == 3:
            w = w[1:4, 1:4, 1:4]
        return out.copy_(w.permute(3, 4, 0, 1, 2))

    def merge(k5, k3, k1, a3, a5, g, dtype, want_wf=True, want_wd=False):
        w = orc.merge_filters(orc.expert_bank(k5, k3, k1, a3, a5), g)          # [S,Co,Ci,5,5,5]
        # the data-gradient filter: taps flipped, channel roles swapped
        return (w if want_wf else None), (w.flip(3, 4, 5).transpose(1, 2).contiguous() if want_wd else None)

    def xfrags(k5, k3, dtype, want_wd=True):
        co, ci = k5.shape[:2]
        z = torch.zeros(co, ci, 1, 1, 1)
        g = torch.zeros(2, 5, co)
        g[0, 0] = 1.0
        g[1, 1] = 1.0
        return merge(k5, k3, z, z, z, g, dtype, want_wf=True, want_wd=want_wd)

    def conv5(x_cl, w, slot, cout, out_f32=False, out=None, centre3=False, accumulate=False):
        ws = w[slot.long()]
        y = orc.conv_per_sample(x_cl.float().permute(0, 4, 1, 2, 3), ws).permute(0, 2, 3, 4, 1).contiguous()
        if out is not None:
            if accumulate:
                out.add_(y)
            else:
                out.copy_(y)
            return out
        return y

    def wgrad(x_cl, dy_cl, plan, cout, centre3=False, expert_layout=None, out=None):
        with torch.enable_grad():
            n, ci = x_cl.shape[0], x_cl.shape[-1]
            wt = torch.zeros(1, cout, ci, 5, 5, 5, requires_grad=True)
            y = orc.conv_per_sample(x_cl.detach().float().permute(0, 4, 1, 2, 3), wt.expand(n, -1, -1, -1, -1, -1))
            (y * dy_cl.detach().float().permute(0, 4, 1, 2, 3)).sum().backward()
        if expert_layout == 5:
            return out.copy_(wt.grad[0])
        if expert_layout == 3:
            return out.copy_(wt.grad[0][:, :, 1:4, 1:4, 1:4])
        return wt.grad.reshape(1, cout, ci, 125).permute(0, 3, 1, 2).contiguous()

    def box(in3=None, in5=None, out=None, add=(), out_dtype=torch.float32):
        def one(t, k):
            c = t.shape[-1]
            w = torch.ones(c, 1, k, k, k) / k ** 3
            return torch.nn.functional.conv3d(t.permute(0, 4, 1, 2, 3), w, padding=k // 2,
                                              groups=c).permute(0, 2, 3, 4, 1).contiguous()
        res = 0
        if in3 is not None:
            res = res + one(in3, 3)
        if in5 is not None:
            res = res + one(in5, 5)
        for a in add:
            res = res + a
        if out is not None:
            out.copy_(res)
            return out
        return res.to(out_dtype)

    def mix_fwd(p, gn):
        return (p * gn.permute(1, 0, 2)[:, :, None, None, None, :]).sum(0)

    def mix_bwd(dy, p, gn, dtype):
        dg = (p * dy[None]).sum((2, 3, 4)).permute(1, 0, 2).contiguous()
        dye = dy[None] * gn.permute(1, 0, 2)[:, :, None, None, None, :]
        return dg, dye[:2].to(dtype).contiguous(), dye[2:].reshape(3, -1, dye.shape[-1]).contiguous()

    for name, fn in dict(gate_softmax=gate, gate_softmax_samples=gate_samples, gate_bwd=gate_bwd, tap_transpose=tap_t,
                         gatrep_merge=merge, expert_frags=xfrags, conv5=conv5, conv5_wgrad=wgrad, box_sum=box,
                         expert_mix_fwd=mix_fwd, expert_mix_bwd=mix_bwd).items():
        monkeypatch.setattr(ops, name, fn)
    monkeypatch.setattr(ops, '_require_hip', lambda *a: None)
    return ops


@pytest.mark.parametrize('co,ci,shape', [(6, 5, (3, 4, 5)), (4, 4, (2, 4, 4)), (1, 7, (2, 3, 3))])
def test_unmerged_matches_reference_arithmetic(fake_kernels, co, ci, shape):
    ops = fake_kernels
    gen = torch.Generator().manual_seed(co * 10 + ci)
    u = lambda *s: torch.rand(*s, generator=gen) - 0.5
    ps = [u(co, ci, 5, 5, 5), u(co, ci, 3, 3, 3), u(co, ci, 1, 1, 1), u(co, ci, 1, 1, 1), u(co, ci, 1, 1, 1),
          u(5 * co, 12), u(5 * co)]
    tasks = [4, 9, 4, 0]
    x = torch.randn(4, ci, *shape, generator=gen)
    r = torch.randn(4, co, *shape, generator=gen)
    ref = [p.clone().requires_grad_(True) for p in ps]
    xr = x.clone().requires_grad_(True)
    yr = orc.mode_conv_pre_bn(xr, *ref, torch.tensor(tasks))
    (yr * r).sum().backward()
    dev = [p.clone().requires_grad_(True) for p in ps]
    xd = x.permute(0, 2, 3, 4, 1).contiguous().requires_grad_(True)
    plan = ops.TaskPlan(torch.tensor(tasks), 12, 'cpu', True)
    assert ops.use_unmerged(torch.empty(4, 2, 4, 8, 3), plan) and not ops.use_unmerged(torch.empty(4, 2, 4, 16, 3), plan)
    y = ops.mode_conv3d(xd, *dev, plan, mode='unmerged')
    (y * r.permute(0, 2, 3, 4, 1)).sum().backward()
    err = lambda a, b: float((a.detach() - b.detach()).abs().max() / b.detach().abs().max())
    assert err(y.permute(0, 4, 1, 2, 3), yr) < 1e-5
    assert err(xd.grad.permute(0, 4, 1, 2, 3), xr.grad) < 1e-5
    for a, b in zip(dev, ref):
        assert err(a.grad, b.grad) < 1e-5
