"""CPU-only checks of the host side: the C ABI library loads and exports exactly what
include/repmode_hip.h declares, the slot plan, the module surface (309-key state_dict, parameter
shapes) and that the product path refuses to run without a HIP device.  No compute calls."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, Opts, load_golden


def _header_functions():
    text = open(os.path.join(ROOT, 'include', 'repmode_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(repmode_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from repmode_amd import _lib
    lib = _lib.load()
    declared = _header_functions()
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), 'librepmode_hip.so does not export %s' % name
    assert set(declared) == set(_lib.EXPORTS), (set(declared) ^ set(_lib.EXPORTS))
    assert lib.repmode_abi_version() == _lib.ABI_VERSION == 11


def test_padded_channels():
    from repmode_amd import _lib
    assert _lib.padded_channels(1, _lib.BF16, True) == 16
    assert _lib.padded_channels(17, _lib.BF16, True) == 32
    assert _lib.padded_channels(1, _lib.F32, True) == 8
    assert _lib.padded_channels(1, _lib.BF16, False) == 32
    assert _lib.padded_channels(33, _lib.F32, False) == 64


def test_argument_validation_without_gpu():
    """Argument errors are reported through the int return code + message, never a crash."""
    from repmode_amd import _lib
    with pytest.raises(_lib.RepModeHipError, match='null pointer'):
        _lib.call('repmode_conv5', None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, None)
    with pytest.raises(_lib.RepModeHipError, match='null pointer'):
        _lib.call('repmode_gate_softmax', None, None, None, 1, 12, 4, None, None)
    buf = ctypes.create_string_buffer(16)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.RepModeHipError):
            _lib.call('repmode_device_arch', 0, buf, 16)


def test_deep_mode_plan_answers_without_gpu():
    """repmode_deep_mode_plan (host logic only: which shapes the one-launch per-expert block takes, and whether its outputs have
    one writer): the network's level 3 / 4 layers at the benchmarked batch, what it refuses, and that calling the kernels with a
    refused shape or null pointers is an error code, not a crash."""
    from repmode_amd import _lib
    lib = _lib.load()
    plan = lambda *a: lib.repmode_deep_mode_plan(*a)
    if torch.cuda.is_available():
        pytest.skip('answers depend on the device\'s CU count; the GPU suite checks them there')
    # (no device: the plan assumes 256 CUs)
    assert plan(0, 8, 4, 8, 8, 256, 256, _lib.BF16) == 1          # enc4.conv2 forward: 32 planes x 8 channel tiles = one per CU
    assert plan(1, 8, 4, 8, 8, 128, 256, _lib.BF16) == 2          # enc4.conv1 data gradient: 4 channel tiles -> two slices
    assert plan(0, 8, 2, 4, 4, 256, 512, _lib.BF16) == 1          # bottle.conv1 forward: one-sample tiles, no split
    assert plan(0, 8, 2, 4, 4, 512, 512, _lib.BF16) == 2          # bottle.conv2 forward: two slices of the reduction
    assert plan(1, 8, 2, 4, 4, 512, 512, _lib.BF16) == 4
    assert plan(0, 2, 6, 10, 8, 16, 32, _lib.BF16) == 0 and plan(0, 2, 8, 8, 8, 16, 32, _lib.BF16) == 0
    assert plan(0, 2, 4, 8, 8, 12, 32, _lib.BF16) == 0 and plan(0, 2, 4, 8, 8, 16, 32, _lib.F32) == 0
    with pytest.raises(_lib.RepModeHipError, match='null pointer'):
        _lib.call('repmode_deep_mode_fwd', None, None, None, None, None, None, None, None, None, 2, 4, 8, 8, 16, 32, None)
    with pytest.raises(_lib.RepModeHipError, match='null pointer'):
        _lib.call('repmode_box_pair', None, None, None, None, 1, 4, 8, 8, 16, None)
    with pytest.raises(_lib.RepModeHipError):
        _lib.call('repmode_concat_channels', 16, 16, 16, 4, 24, 16, None)      # a row piece that is no multiple of 16 bytes


def test_task_plan_grouping():
    from repmode_amd.ops import TaskPlan
    p = TaskPlan(torch.tensor([7, 2, 7, 11, 2]), 12, 'cpu', training=True)
    assert p.nslots == 3 and p.slot_task_host == [2, 7, 11]
    assert p.sample_slot.tolist() == [1, 0, 1, 2, 0] and p.slot_task.tolist() == [2, 7, 11]
    assert p.slot_task.dtype == torch.int32
    # eval: the first sample's task for the whole batch (RepMode.py:209-210)
    e = TaskPlan(torch.tensor([5, 5, 9]), 12, 'cpu', training=False)
    assert e.nslots == 1 and e.slot_task_host == [5] and e.sample_slot.tolist() == [0, 0, 0]
    # one-hot rows as the reference's MoDEConv receives them
    oh = torch.zeros(2, 12)
    oh[0, 3] = oh[1, 8] = 1
    assert TaskPlan(oh, 12, 'cpu').slot_task_host == [3, 8]
    with pytest.raises(ValueError):
        TaskPlan([12], 12, 'cpu')
    with pytest.raises(ValueError):
        TaskPlan([-1], 12, 'cpu')


def test_module_surface_matches_reference_state_dict():
    from repmode_amd.nn_modules.RepMode import Net
    g = load_golden('g3_net_mc2.npz')
    ref = {k[2:]: v for k, v in g.items() if k.startswith('p.')}
    net = Net(Opts(), mult_chan=2)
    sd = net.state_dict()
    assert len(sd) == 309 and list(sd) == list(ref)
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k].shape, k
    net.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()})      # reference checkpoint loads
    with torch.device('meta'):
        big = Net(Opts(), mult_chan=32)
    assert sum(p.numel() for p in big.parameters()) == 123_877_633
    assert len(list(big.parameters())) == 193


def test_init_distribution_matches_reference():
    """kaiming_uniform_(a=sqrt(5)) => U(+-1/sqrt(fan_in)) (RepMode.py:156-159)."""
    from repmode_amd.nn_modules.RepMode import MoDEConv
    torch.manual_seed(0)
    blk = MoDEConv(5, 12, 16, 64)
    for p, k in [(blk.expert_conv5x5_conv, 5), (blk.expert_conv3x3_conv, 3), (blk.expert_conv1x1_conv, 1)]:
        bound = 1.0 / (16 * k ** 3) ** 0.5
        assert p.abs().max() <= bound and p.abs().max() > 0.9 * bound
    assert torch.allclose(blk.expert_avg3x3_pool, torch.full((3, 3, 3), 1 / 27.))
    assert torch.allclose(blk.expert_avg5x5_pool, torch.full((5, 5, 5), 1 / 125.))
    assert blk.gate.weight.shape == (5 * 64, 12)


def test_no_cpu_fallback():
    from repmode_amd import _lib
    from repmode_amd.nn_modules.RepMode import Net
    net = Net(Opts(), mult_chan=2)
    with pytest.raises(RuntimeError, match='no CPU fallback'):       # TORCH_CHECK of the operator library
        net(torch.randn(1, 1, 16, 16, 16), torch.tensor([0]))
    with pytest.raises(ValueError, match='multiples of 16'):
        net(torch.randn(1, 1, 16, 16, 20), torch.tensor([0]))


def test_product_code_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under repmode_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'repmode_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), os.path.join(dirpath, f)


def test_predict_helpers_match_golden():
    from repmode_amd.model import get_gaussian, patch_grid
    import numpy as np
    g = load_golden('g5_predict.npz')
    assert np.array_equal(get_gaussian((16, 32, 32)), g['gauss_16x32x32'])
    grid = patch_grid((64, 624, 924), (32, 128, 128))
    assert len(grid) == 378 and int(g['patches_64x624x924_nbatches']) == 48
    assert grid[0] == ([0, 0, 0], [32, 128, 128]) and grid[-1] == ([32, 496, 796], [64, 624, 924])


def test_zero_pool_semantics():
    """The operator library's ZeroPool (csrc/torch/repmode_ops.cpp) on CPU tensors: the first step of a key records,
    later steps hand out zeroed, non-overlapping, independently versioned tensors in the recorded order, and any
    divergence falls back to plain allocations."""
    import torch
    from repmode_amd.ops import ZERO_POOL as pool
    dev = torch.device('cpu')
    key = ('test_zero_pool_semantics', 1)
    shapes = [(3, 5), (7,), (2, 2, 4)]
    assert not pool.has_plan(key)
    pool.begin(key, dev)
    first = [pool.take(s, dev) for s in shapes]
    assert all(not pre for _, pre in first)                      # recording step: nothing pooled
    pool.begin(key, dev)                                         # (begin() closes the previous step)
    assert pool.has_plan(key)
    got = [pool.take(s, dev) for s in shapes]
    assert all(pre for _, pre in got)
    ts = [t for t, _ in got]
    assert all(float(t.abs().sum()) == 0.0 and tuple(t.shape) == s and t.dtype == torch.float32 for t, s in zip(ts, shapes))
    v1 = ts[1]._version
    ts[0].fill_(1.0)                                             # in-place op on one tensor ...
    assert ts[1]._version == v1 and float(ts[1].sum()) == 0.0    # ... neither touches nor re-versions another
    ptrs = sorted((t.data_ptr(), t.numel() * 4) for t in ts)
    assert all(a + n <= b for (a, n), (b, _) in zip(ptrs, ptrs[1:]))
    assert all(p % 256 == ptrs[0][0] % 256 for p, _ in ptrs)     # 64-float granules
    extra, pre = pool.take((4,), dev)                            # more requests than recorded: plain allocation
    assert not pre
    pool.begin(key, dev)
    _, pre0 = pool.take(shapes[0], dev)
    _, pre1 = pool.take((9, 9), dev)                             # diverges from the recorded sequence
    _, pre2 = pool.take(shapes[2], dev)
    assert pre0 and not pre1 and not pre2
    pool.end()
    assert pool.take((3,), dev)[1] is False                      # inactive pool


def test_operator_library_schemas():
    """librepmode_torch.so loads without a GPU and registers the ops of the seam with the
    argument lists the Python side passes; a CPU tensor is refused with a RuntimeError that says why (TORCH_CHECK)."""
    import pytest
    import torch
    from repmode_amd import ops
    t = ops.torch_ops()
    for name in ('mode_block', 'mode_conv3d', 'bn_relu', 'down2', 'up2', 'stage2_bn_relu', 'zero_pool_begin', 'zero_pool_end',
                 'grad_sink_set', 'grad_sink_clear', 'eval_cache_begin', 'eval_cache_end', 'set_fork_max_w'):
        assert hasattr(t, name), name
    schema = str(torch.ops.repmode.mode_block.default._schema)
    assert 'Tensor? x2' in schema and 'Tensor slot_task' in schema and 'bool out_f32' in schema
    plan = ops.TaskPlan([3, 5], 12, 'cpu', training=True)
    co, ci = 4, 2
    ps = [torch.zeros(co, ci, 5, 5, 5), torch.zeros(co, ci, 3, 3, 3)] + [torch.zeros(co, ci, 1, 1, 1)] * 3 + \
         [torch.zeros(5 * co, 12), torch.zeros(5 * co)]
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.mode_conv3d(torch.zeros(2, 4, 4, 4, ci), *ps, plan)
    ops.set_fork_max_w(16)
    assert ops.get_fork_max_w() == 16
    ops.set_fork_max_w(0)
    with ops.eval_filter_cache():
        assert int(t.eval_cache_size()) == 0




def test_seeded_init_matches_reference_fixture():
    """``torch.manual_seed(0)`` + this build's construction order gives the reference's initial state bit for bit (same
    parameters, same order of random draws): checked against the fingerprint the REAL fnet_model.Model (mult_chan 32)
    produced when g4b was captured, for repmode_amd's Net and the oracle's.  This is what lets the full-size train-step
    scalars be pinned without storing the 0.5 GB state."""
    import numpy as np
    import torch
    from oracle import repmode_oracle as orc
    from repmode_amd.nn_modules.RepMode import Net
    g = load_golden('g4b_model_train_iter.npz')
    keys = [str(k) for k in g['state_fingerprint_keys']]
    for cls in (Net, orc.Net):
        torch.manual_seed(0)
        sd = cls(Opts(), mult_chan=32).state_dict()
        assert sorted(k for k in sd if sd[k].dtype.is_floating_point) == keys
        finger = np.asarray([float(sd[k].double().sum()) for k in keys])
        assert np.array_equal(finger, g['state_fingerprint']), cls


def test_wgrad_lds_layout_is_conflict_free_in_the_bank_model():
    """The filter-gradient kernel's LDS layout (csrc/conv5_wgrad.hip, WgTile) against the bank model of ds_read_b128
    (tools/lds_bank_check.py): no lane group of any B-window or A-fragment read of any tile puts two lanes on one
    16-byte slot; round 1's layout cost every read one extra cycle per lane group.  The model restates the kernel's
    constants, so the source is checked to still carry them."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('lds_bank_check', os.path.join(root, 'tools', 'lds_bank_check.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    src = open(os.path.join(root, 'repmode_amd', 'csrc', 'conv5_wgrad.hip')).read()
    for line in ('static constexpr int RG = TX >= 32 ? 8 : TX >= 16 ? 4 : 2;',
                 'static constexpr bool SWZ = TX < 16;',
                 'static constexpr int ROW_C = TZ * HY * RG * 16 + 32;',
                 'static constexpr int DYS = TV * 2 + 32;'):
        assert line in src, line
    for t in chk.TILES:
        assert ('conv5_wgrad_bf16_kernel<%d, %d, %d' % t) in src or ('launch_wgrad_bf16<%d, %d, %d>' % t) in src, t
        assert chk.wgrad_x(*t) == 0 and chk.wgrad_dy(*t) == 0, t
        assert chk.round1_x(*t) == 4 * 5 * t[0] * t[1] * t[2] // 32        # one extra cycle per lane group and read


def test_task_plan_index_vectors_share_one_buffer():
    """ops.TaskPlan packs slot_task | sample_slot | sample_task into ONE int32 buffer (one pinned host-to-device copy per step on a
    GPU): each vector starts on a 16-int boundary of it and holds what the separate tensors held."""
    import torch
    from repmode_amd import ops
    tasks = [7, 3, 7, 11, 0, 3, 3, 9, 7]
    plan = ops.TaskPlan(torch.tensor(tasks), 12, 'cpu', training=True)
    uniq = sorted(set(tasks))
    assert plan.nslots == len(uniq) and plan.slot_task_host == uniq
    assert plan.slot_task.tolist() == uniq
    assert plan.sample_slot.tolist() == [uniq.index(t) for t in tasks]
    assert plan.sample_task.tolist() == tasks
    for v in (plan.slot_task, plan.sample_slot, plan.sample_task):
        assert v.dtype == torch.int32 and v.is_contiguous() and v.storage_offset() % 16 == 0
    assert plan.slot_task.untyped_storage().data_ptr() == plan.sample_task.untyped_storage().data_ptr()
    # eval mode: one slot (the first sample's task), RepMode.py:209-210
    ev = ops.TaskPlan(torch.tensor(tasks), 12, 'cpu', training=False)
    assert ev.nslots == 1 and ev.slot_task.tolist() == [7] and ev.sample_slot.tolist() == [0] * len(tasks)


def test_adam_bookkeeping_and_state_dict_interchange(monkeypatch):
    """repmode_amd.optim.Adam on the host side (the kernels are the GPU tests'): it IS a torch.optim.Adam -- same state keys,
    `step` counters as host scalars, parameters grouped by the number of updates they have seen, state dicts loadable by the
    stock optimizer and back (counters a fused optimizer kept on the device come to the host: no device read per step)."""
    import torch
    from repmode_amd import ops, optim
    calls = []

    class FakeOps:
        def adam_step(self, p, g, m, v, lr, b1, b2, eps, t):
            calls.append((len(p), t, lr, b1, b2, eps))

    monkeypatch.setattr(ops, 'torch_ops', lambda: FakeOps())
    ps = [torch.nn.Parameter(torch.randn(3)) for _ in range(4)]
    opt = optim.Adam(ps, lr=2e-3)
    assert isinstance(opt, torch.optim.Adam)
    for p in ps:
        p.grad = torch.randn(3)
    opt.step(); opt.step()
    ps[0].grad = None
    opt.step()
    for p in ps:
        p.grad = torch.randn(3)
    opt.step()
    assert [(c[0], c[1]) for c in calls] == [(4, 1), (4, 2), (3, 3), (1, 3), (3, 4)]
    assert calls[0][2:] == (2e-3, 0.9, 0.999, 1e-8)
    sd = opt.state_dict()
    assert sorted(sd['state'][0]) == ['exp_avg', 'exp_avg_sq', 'step'] and float(sd['state'][0]['step']) == 3.0
    assert not sd['state'][1]['step'].is_cuda and sd['param_groups'][0]['fused'] is False
    stock = torch.optim.Adam(ps, lr=2e-3)
    stock.load_state_dict(sd)
    assert float(stock.state_dict()['state'][1]['step']) == 4.0
    back = optim.Adam(ps, lr=2e-3)
    back.load_state_dict(stock.state_dict())
    assert float(back.state_dict()['state'][0]['step']) == 3.0
    with pytest.raises(RuntimeError):
        bad = optim.Adam(ps, lr=1e-3)
        bad.param_groups[0]['weight_decay'] = 0.1
        bad.step()
