#!/usr/bin/env python3
"""Headline benchmark: train-step voxels/s of the RepMode U-Net on 32x64x64 patches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one full optimisation step of the hot path on one batch of synthetic patches per GPU:
forward (19 MoDE blocks through the HIP kernels) + backward (data and filter gradients, GatRep
backward) + Adam over all 123.9 M parameters, exactly the work of fnet_model.py:105-113.  Per-GPU
batch is fixed (weak scaling); ``value`` = all ranks' input voxels / max-over-ranks step time.
Prints ONE JSON line on rank 0.

Also reported in the same line:
  roofline      the conv5_igemm kernel (forward + data-gradient launches): algorithmic FLOPs
                (2 * voxels * Cin * Cout * 125 per launch) / HIP-event duration on the launch
                stream, summed over the timed region, against the dense bf16 MFMA peak.
  cpu_baseline  the CPU oracle (oracle/repmode_oracle.py, a port -- the reference's Python cannot
                travel) timed on this box's host cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PATCH = (32, 64, 64)
PER_GPU_BATCH = 8
PROF_EVERY = 10          # the per-launch HIP events of the roofline are taken on every 10th timed step (launched kernel by kernel)
MULT_CHAN = 32
NUM_TASKS = 12
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md


class Opts:
    adopted_datasets = ['alpha_tubulin', 'beta_actin', 'desmoplakin', 'dna', 'fibrillarin', 'lamin_b1',
                        'membrane_caax_63x', 'myosin_iib', 'sec61_beta', 'st6gal1', 'tom20', 'zo1']
    gpu_ids = 0
    batch_size_eval = 8


def cpu_baseline(seconds_budget=30.0):
    """Oracle train step (forward + backward + Adam, fp32) on the host cores.  Bounded: one batch-1
    step of the full-size network on the headline patch (the per-voxel cost does not depend on the
    batch size); a tiny warm-up first so thread pools and allocators are initialised."""
    from oracle import repmode_oracle as orc
    # the oracle's PyTorch-CPU ops stop scaling (and then regress) well before a big host's core
    # count: 32 threads is the measured sweet spot region; `cores` reports what was actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = orc.Net(Opts(), mult_chan=MULT_CHAN)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    net.train()
    tasks = torch.tensor([3])
    x = torch.randn(1, 1, 16, 32, 32)
    orc.train_step(net, opt, x, torch.randn_like(x), tasks)            # warm-up, small patch
    x = torch.randn(1, 1, *PATCH)
    tgt = torch.randn_like(x)
    # as many batch-1 steps as fit in ~12 s (at least one): the sample stays bounded on a slow host
    nsteps, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(net, opt, x, tgt, tasks)
        nsteps += 1
        dt = time.perf_counter() - t0
        if dt + dt / nsteps > 12.0 or dt > seconds_budget:
            break
    vox = PATCH[0] * PATCH[1] * PATCH[2]
    return {'value': vox * nsteps / dt, 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d train step(s) (fwd+bwd+Adam, fp32, vectorised oracle) of the full mult_chan=32 network on '
                      'batch 1 of 1x32x64x64; %.1f s' % (nsteps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='patches per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-inputs', action='store_true', help='inputs start in pinned host memory every step (PCIe-inclusive rate)')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-launch HIP events')
    ap.add_argument('--graph', action='store_true', help='replay the train step as one HIP graph (N = 1; see DESIGN.md 3.5; REPMODE_FORK_MAX_W=16 adds the two-stream layers)')
    ap.add_argument('--prof-all', action='store_true', help='record HIP events for every library kernel, not only conv5_igemm')
    ap.add_argument('--dump-launches', default=None, help='write per-launch (kind, ms, TFLOP/s or TB/s) of the last timed step as JSON')
    args = ap.parse_args()

    from repmode_amd import _lib, distributed as dist_
    from repmode_amd.model import Model

    rank, world, local = dist_.init_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch N>1 through torch.distributed.run' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a MI355X; the product path has no CPU fallback')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32

    opts = Opts()
    opts.gpu_ids = local
    torch.manual_seed(0)                       # reference default seed (config.py:47); same init on every rank
    model = Model(opts, nn_module='RepMode', lr=1e-4, gpu_ids=local, mult_chan=MULT_CHAN, dtype=dtype,
                  distributed=world > 1, hip_graph=world == 1 and args.graph)
    b = args.batch
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    signal = torch.randn(b, 1, *PATCH, device=device, generator=gen)
    target = torch.randn(b, 1, *PATCH, device=device, generator=gen)
    if args.host_inputs:
        # the PCIe-inclusive variant: the batch is handed over in (pinned) host memory every step, as a DataLoader
        # would; never the headline `value` (inputs resident in HBM), reported in DESIGN.md section 5
        signal, target = signal.cpu().pin_memory(), target.cpu().pin_memory()
    task = (torch.arange(b) + rank * b) % NUM_TASKS            # CPU int tensor, like the DataLoader's

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.do_train_iter(signal, target, task)
    barrier()
    # HIP events around the dominant kernel's launches only by default, and on every PROF_EVERY-th timed step (each
    # event pair costs ~4 us of stream time: ~0.5 ms per step if every launch of every step were bracketed)
    sample = 1 if (args.prof_all or args.dump_launches) else PROF_EVERY
    if not args.no_prof:
        _lib.prof_enable(1 if (args.prof_all or args.dump_launches) else 2)
    profiled_steps = 0
    t0 = time.perf_counter()
    for step in range(args.steps):
        if not args.no_prof:
            on = step % sample == 0
            _lib.prof_pause(not on)
            profiled_steps += on
        # (a profiled step is launched kernel by kernel: the library's event pairs are not part of the captured graph)
        model.do_train_iter(signal, target, task, eager=not args.no_prof and on)
    t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (== dt when the host is the limiter)
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    dt = dist_.max_over_ranks(dt, device)
    loss = float(model.last_loss)

    vox_per_step = world * b * PATCH[0] * PATCH[1] * PATCH[2]
    out = {
        'metric': 'train-step voxels/sec (RepMode U-Net, 32x64x64 patch)',
        'value': vox_per_step * args.steps / dt,
        'unit': 'voxels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': args.dtype,
        'data': 'synthetic' + (' (inputs in pinned host memory every step: PCIe-inclusive)' if args.host_inputs else ''),
        'config': {'workload': 'RepMode U-Net (mult_chan 32, 12 tasks, 123.9M params) full train step '
                               '(fwd + bwd + Adam), batch %d x 1x32x64x64 per GPU' % b,
                   'global_batch': world * b, 'patch': list(PATCH), 'parallelism': 'dp%d' % world,
                   'final_loss': loss, 'host_issue_ms_per_step': 1e3 * t_issue / args.steps,
                   'hip_graph': bool(model.hip_graph), 'steps_launched_kernel_by_kernel': profiled_steps},
    }
    if rank == 0:
        if not args.no_prof:
            kinds = {}
            for kind in ('conv5_igemm', 'conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd'):
                n, ms, work = _lib.prof_summary(kind)
                if n == 0:
                    continue          # kind not recorded (default: the dominant kernel only; --prof-all for all)
                kinds[kind] = {'launches': n, 'ms_per_step': ms / max(profiled_steps, 1),
                               'rate': (work / (ms * 1e-3) / 1e12) if ms > 0 else None}   # TFLOP/s or TB/s
            n, ms, flops = _lib.prof_summary('conv5_igemm')
            achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak = PEAK_TFLOPS[args.dtype]
            # HBM bytes per launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE,
            # separate runs, gfx950 read-side correction applied by profiles/pmc_summary.py); null if not collected
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
            if args.dtype == 'bf16' and b == PER_GPU_BATCH and os.path.exists(pmc):
                traffic = json.load(open(pmc)).get('conv5_igemm', {}).get('hbm_bytes_per_launch')
            out['roofline'] = {'kernel': 'conv5_igemm_kernel', 'bound': 'mfma', 'achieved': achieved, 'peak': peak,
                               'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
                               'launches': n, 'avg_launch_ms': ms / max(n, 1),
                               'flops_per_launch': flops / max(n, 1)}
            out['kernels'] = kinds
            if args.dump_launches:
                recs = _lib.prof_records()
                per = len(recs) // max(profiled_steps, 1)
                last = [{'kind': k, 'us': ms * 1e3, 'rate': (w / (ms * 1e-3) / 1e12) if ms > 0 else None,
                         'work': w} for k, ms, w in recs[-per:]]
                with open(args.dump_launches, 'w') as f:
                    json.dump(last, f, indent=0)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
