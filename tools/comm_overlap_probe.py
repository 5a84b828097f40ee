#!/usr/bin/env python3
"""What do RCCL's all-reduce kernels cost BESIDE the backward pass's convolution launches on one MI355X?  (VERDICT round 2,
weak #10: a side-stream kernel next to a conv launch starves -- conv workgroups hold every CU's LDS and registers.)

A one-rank RCCL group still launches its all-reduce kernels, so ``Model(distributed='reducer-always')`` executes the ten
48 MB bucket collectives under backward exactly where an 8-GPU run would issue them (what is missing is the link time, not
the launches).  Measured here, all in ONE process on one box:
  plain        train step without collectives
  beside       train step with the collectives under backward (reducer-always)
  masked-*     the same with the COMPUTE stream restricted by a CU mask (hipExtStreamCreateWithCUMask), so that RCCL's
               stream always finds free CUs:  top32 = CUs 224..255 left free, spread16 = every 16th CU left free
  alone        every bucket's all-reduce on an otherwise idle GPU (HIP events)
and, under ``rocprofv3 --kernel-trace`` (tools/r3_comm.sh), the RCCL kernels' own durations beside the convs.
    python tools/comm_overlap_probe.py [steps] [bf16]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
import torch
import torch.distributed as dist
from conftest import Opts
from repmode_amd.model import Model

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
compress = 'bf16' if 'bf16' in sys.argv else None
only = [a for a in sys.argv[1:] if a in ('plain', 'beside', 'masked-top32', 'masked-spread16', 'alone')]
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(8, 1, 32, 64, 64, device=dev, generator=gen)
t = torch.randn(8, 1, 32, 64, 64, device=dev, generator=gen)
tasks = torch.arange(8) % 12


def masked_stream(free):
    hip = ctypes.CDLL('libamdhip64.so')
    mask = (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
    for cu in free:
        mask[cu // 32] &= ~(1 << (cu % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, mask)
    assert rc == 0, 'hipExtStreamCreateWithCUMask failed: %d' % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(distributed, stream=None, label=''):
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16, distributed=distributed, grad_compress=compress)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(8):
            m.do_train_iter(x, t, tasks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.do_train_iter(x, t, tasks)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    print('%-18s %.3f ms/step' % (label, ms), flush=True)
    return m, ms


res = {}
if not only or 'plain' in only:
    _, res['plain'] = run(False, label='plain')
if not only or 'beside' in only:
    m, res['beside'] = run('reducer-always', label='beside' + (' (bf16 buckets)' if compress else ''))
    if not only or 'alone' in only:
        torch.cuda.synchronize()
        tot = 0.0
        for i, b in enumerate(m.reducer.buckets):
            buf = b.comm if b.comm is not None else b.flat
            for _ in range(3):
                dist.all_reduce(buf, op=dist.ReduceOp.AVG)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dist.all_reduce(buf, op=dist.ReduceOp.AVG)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            tot += us
            print('  bucket %2d  %6.1f MB  all-reduce alone %7.1f us' % (i, buf.numel() * buf.element_size() / 1e6, us), flush=True)
        print('  all buckets alone: %.1f us per step' % tot)
    del m
if not only or 'masked-top32' in only:
    _, res['masked-top32'] = run('reducer-always', masked_stream(range(224, 256)), 'masked-top32')
if not only or 'masked-spread16' in only:
    _, res['masked-spread16'] = run('reducer-always', masked_stream(range(15, 256, 16)), 'masked-spread16')
    _, res['plain-spread16'] = run(False, masked_stream(range(15, 256, 16)), 'plain, 240 CUs')
if 'plain' in res:
    for k, v in res.items():
        if k != 'plain':
            print('%-18s +%.3f ms over plain' % (k, v - res['plain']))
dist.destroy_process_group()
