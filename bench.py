#!/usr/bin/env python3
"""Headline benchmark: train-step voxels/s of the RepMode U-Net on 32x64x64 patches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one full optimisation step of the hot path on one batch of synthetic patches per GPU:
forward (19 MoDE blocks through the HIP kernels) + backward (data and filter gradients, GatRep
backward) + Adam over all 123.9 M parameters, exactly the work of fnet_model.py:105-113.
``value`` = all ranks' input voxels / max-over-ranks step time.  Prints ONE JSON line on rank 0.

Workloads (BASELINE.json ``configs``):
  N = 1   configs[1]: batch 8 of 1x32x64x64, bf16, 8 distinct tasks  -- the configuration the metric is quoted on.
  N > 1   configs[3]: 24 patches per rank (global 192 at 8 GPUs), all 12 tasks on every rank, gradients all-reduced
          over RCCL under backward.  Per-GPU work is the same at N = 2, 4, 8 (weak scaling); it is NOT the N = 1
          line's batch, so the line also carries ``config.no_comm_value``: the same ranks, same batch, same run,
          stepping WITHOUT the gradient all-reduce -- the denominator for the collectives' cost at this batch
          (``value / no_comm_value`` is the scaling efficiency that the N = 1 line cannot give).  ``config.comm``: what the
          process group looked like (backend, ranks a collective saw, the buckets DDP built and their sizes, their dtype, and
          ``efficiency`` = value / no_comm_value).  The buckets are float32 (``--grad-compress fp32``, the default: what the
          reference averages in); ``--grad-compress auto`` lets the rule (distributed.pick_grad_dtype) choose from the first
          four distributed steps: with a shorter --warmup the missing ones run as setup BEFORE the warmup
          (``config.setup_steps_before_warmup``), never inside the timed region.
  ``config.ms_per_step_unprofiled``: the mean GPU time (HIP events at the step boundaries) of the timed steps that were NOT
          launched kernel by kernel for the roofline's event pairs -- the steady-state step; ``ms_per_step`` / ``value`` keep
          every timed step, the profiled ones included (pessimistic by about 1 %).
  ``--batch B`` overrides the per-GPU batch for either.

Also reported in the same line:
  roofline      the MoDE convolution, forward + data-gradient launches, of every block but the two one-channel ends (the
                dominant kernel family): conv5_ws_kernel (levels 0-2: 22 launches, 93 % of the conv FLOPs) and deep_mode_kernel
                (levels 3-4, round 5: a per-expert block as one launch per direction), together and each under ``by_kernel``;
                beside them the filter gradient (entry `conv5_wgrad_bf16_kernel`: that kernel on levels 0-1 and 3-4 and, since
                round 6, conv5_wgrad_col_kernel on level 2 -- all 20 multi-channel filter-gradient launches), the step's second-largest family, with its own
                algorithmic bytes: algorithmic FLOPs (the layer's merged 125-tap convolution, 2 * voxels * Cin * Cout * 125,
                once per layer and direction) / HIP-event duration on the launch stream, against the dense bf16 MFMA peak;
                ``all_conv_kernels``: the same + the one-channel layers' kernels (+ conv5_deep / conv5_igemm where round 4's
                launches still run).  ``traffic``: HBM bytes per launch from rocprofv3 PMC passes of THIS build
                (profiles/*_pmc_traffic.json carries the hash of the kernel sources it was taken on), else null.
  fwd           forward only (BASELINE's ">= 40 % MFMA on fused GatRep+Conv3d forward"): voxels/s of the whole forward pass, and
                the MFMA fraction of EVERY kernel the MoDE blocks launch in a pass as ONE unit -- gate softmax + GatRep, the
                convolutions, and the helper kernels (box means, 1x1-expert GEMMs, gate mix, skip concatenation): their summed
                event-timed durations against the forward's algorithmic conv FLOPs.
  cpu_baseline  the CPU oracle (oracle/repmode_oracle.py, a port -- the reference's Python cannot travel) timed on
                this box's host cores, rank 0 at N = 1 only: batch 2 of the same patches, both organisations of the
                arithmetic (the reference's per-sample Python loop, and the vectorised one).
"""
import argparse
import hashlib
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PATCH = (32, 64, 64)
BATCH_1GPU = 8           # BASELINE configs[1]
BATCH_MULTI = 24         # BASELINE configs[2] / configs[3]: 24 per GPU, global 192 on 8
PROF_EVERY = 10          # the per-launch HIP events of the roofline are taken on every 10th timed step
MULT_CHAN = 32
NUM_TASKS = 12
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
FWD_FLOP_PER_VOXEL = 2083520.0                    # whole forward; MoDE convs alone 2,072,000
FWD_CONV_FLOP_PER_VOXEL = 2072000.0
# the forward / data-gradient convolution kernels of the MoDE blocks
CONV_KINDS = ('conv5_ws', 'conv5_igemm', 'deep_mode', 'deep_mode_dgrad', 'conv5_deep', 'conv5_thin')
# the MoDE convolution, forward and data gradient, of every block but the two one-channel ends: the wave-specialised kernel
# (levels 0-2), the per-expert blocks' one-launch kernel (levels 3-4, round 5) and whatever still goes through the general kernel
MAIN_KINDS = ('conv5_ws', 'conv5_igemm', 'deep_mode', 'deep_mode_dgrad')
# round 4's `roofline` covered conv5_ws + the per-expert levels' FORWARD pair (their data gradient ran in conv5_deep, outside it):
# the same launches of this build, for a like-for-like figure (`roofline.round4_scope`)
ROUND4_SCOPE = ('conv5_ws', 'conv5_igemm', 'deep_mode')
# the MoDE blocks' small kernels (box means, the 1x1 experts' GEMMs, the gate mix, the skip concatenation of the per-expert
# decoder block): they are part of "GatRep + conv as one unit"
HELPER_KIND = 'helper'
KERNEL_SYMBOL = {'conv5_ws': 'conv5_ws_kernel', 'conv5_igemm': 'conv5_igemm_kernel', 'deep_mode': 'deep_mode_kernel',
                 'deep_mode_dgrad': 'deep_mode_kernel', 'conv5_wgrad': 'conv5_wgrad_bf16_kernel'}


class Opts:
    adopted_datasets = ['alpha_tubulin', 'beta_actin', 'desmoplakin', 'dna', 'fibrillarin', 'lamin_b1',
                        'membrane_caax_63x', 'myosin_iib', 'sec61_beta', 'st6gal1', 'tom20', 'zo1']
    gpu_ids = 0
    batch_size_eval = 8


def kernel_source_hash():
    """Hash of the kernel sources + header: identifies the build a profile under profiles/ was taken on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'repmode_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, 'repmode_amd', 'csrc', '*.h')) +
                    glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kind, batch, dtype):
    """HBM bytes per launch of ``kind`` from the newest profiles/*_pmc_traffic.json taken on THIS build and workload."""
    want = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get('kernel_source_hash') == want and d.get('batch') == batch and d.get('dtype') == dtype:
            return d.get(kind, {}).get('hbm_bytes_per_launch'), os.path.basename(f)
    return None, None


def conv_algorithmic_bytes(batch, nslots, deep_mode):
    """Algorithmic HBM bytes per train step of the MoDE convolution's forward / data-gradient launches, per kernel kind, as
    {kind: (bytes, launches)}: every tensor once -- the bf16 input, the output (bf16 where the kernel library
    writes the element type -- `repmode_conv5_elem_out`: levels 0-2 at the benchmarked batch --, float where the reduction is
    split over workgroups or the consumer wants float) and the filter (merged levels: one per slot, 125 taps x Cin x Cout
    bf16; per-expert levels 3-4: 125 + 27 taps bf16 + three 1x1 experts float, shared by the batch).
    ``deep_mode``: the per-expert blocks run as one launch per direction (round 5: forward reads x, writes float y; the data
    gradient reads the two gate-scaled bf16 output gradients and writes bf16 dx) -- the float P_e tensors the forward keeps
    for the gate gradient and the box-mean operands are NOT algorithmic: they show up in traffic_over_algorithmic.
    Otherwise (round 4's launches): the forward pair through conv5_igemm (two float outputs); its data gradient runs in
    conv5_deep, which has no entry."""
    from repmode_amd import _lib
    lib = _lib.load()
    dims = [(PATCH[0] >> l, PATCH[1] >> l, PATCH[2] >> l) for l in range(5)]
    v = [d * h * w for d, h, w in dims]
    merged = [(0, 32, 32), (0, 64, 32), (0, 32, 32), (1, 32, 64), (1, 64, 64), (1, 128, 64), (1, 64, 64),
              (2, 64, 128), (2, 128, 128), (2, 256, 128), (2, 128, 128)]
    pair = [(3, 128, 256), (3, 256, 256), (3, 512, 256), (3, 256, 256), (4, 256, 512), (4, 512, 512)]
    acc = {'conv5_ws': [0.0, 0], 'conv5_igemm': [0.0, 0], 'deep_mode': [0.0, 0], 'deep_mode_dgrad': [0.0, 0]}
    for l, ci, co in merged:
        for a, b in ((ci, co), (co, ci)):                       # forward, data gradient
            elem = lib.repmode_conv5_elem_out(batch, *dims[l], a, b, _lib.BF16) != 0
            t = batch * v[l] * (a * 2 + b * (2 if elem else 4)) + nslots * 125 * ci * co * 2
            k = 'conv5_ws' if elem else 'conv5_igemm'
            acc[k][0] += t
            acc[k][1] += 1
    for l, ci, co in pair:
        filt = (125 + 27) * ci * co * 2 + 3 * ci * co * 4
        if deep_mode:
            acc['deep_mode'][0] += batch * v[l] * (ci * 2 + co * 4) + filt                  # forward
            acc['deep_mode'][1] += 1
            acc['deep_mode_dgrad'][0] += batch * v[l] * (2 * co * 2 + ci * 2) + filt        # data gradient
            acc['deep_mode_dgrad'][1] += 1
        else:
            acc['conv5_igemm'][0] += batch * v[l] * (ci * 2 + 2 * co * 4) + (125 + 27) * ci * co * 2
            acc['conv5_igemm'][1] += 1
    return {k: (b, n) for k, (b, n) in acc.items()}


def conv5_wgrad_algorithmic_bytes(batch, nslots):
    """Algorithmic HBM bytes per train step of the filter-gradient launches (conv5_wgrad_bf16_kernel; autograd of
    RepMode.py:207): both operands once -- the layer's bf16 input and bf16 output gradient -- and the float32 filter gradient
    once per slot (merged levels 0-2: 125 taps x Cin x Cout per distinct task; the per-expert pair of levels 3-4: 125 + 27 taps,
    shared by the batch, two gate-scaled output gradients).  The two one-channel layers have their own kernel and kind."""
    v = [PATCH[0] * PATCH[1] * PATCH[2] // 8 ** l for l in range(5)]
    merged = [(0, 32, 32), (0, 64, 32), (0, 32, 32), (1, 32, 64), (1, 64, 64), (1, 128, 64), (1, 64, 64),
              (2, 64, 128), (2, 128, 128), (2, 256, 128), (2, 128, 128)]
    pair = [(3, 128, 256), (3, 256, 256), (3, 512, 256), (3, 256, 256), (4, 256, 512), (4, 512, 512)]
    total = 0.0
    for l, ci, co in merged:
        total += batch * v[l] * (ci + co) * 2 + nslots * 125 * ci * co * 4
    for l, ci, co in pair:
        total += batch * v[l] * (ci + 2 * co) * 2 + (125 + 27) * ci * co * 4
    return total


def cpu_baseline():
    """Oracle train step (forward + backward + Adam, fp32) on the host cores: batch 2 of the headline patches (two
    different tasks), up to 32 host threads (see below), >= 3 timed steps (~17 s on the GPU box) of each organisation after a small
    warm-up -- the reference's
    own (one merged filter + one batch-1 conv per sample in a Python loop, RepMode.py:182-190, 204-208) and the
    vectorised restatement (gather + one contraction + one grouped conv).  Bounded: ~2 x 17 s on a 32+ core host (40 s cap each)."""
    from oracle import repmode_oracle as orc
    # the oracle's PyTorch-CPU ops stop scaling well before a big host's hardware-thread count and then regress badly
    # (all 256+ threads of the GPU box: a batch-2 step did not finish in 5 minutes; 32 threads: ~7 s) -- 32 is the
    # measured sweet spot; `cores` reports the threads actually used, the sample text the host's count
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    n = 2
    vox = n * PATCH[0] * PATCH[1] * PATCH[2]
    res = {}
    for style in ('vectorised', 'reference-style'):
        orc.REFERENCE_STYLE = style == 'reference-style'
        torch.manual_seed(0)
        net = orc.Net(Opts(), mult_chan=MULT_CHAN)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        net.train()
        tasks = torch.tensor([3, 7])
        x = torch.randn(n, 1, 16, 32, 32)
        orc.train_step(net, opt, x, torch.randn_like(x), tasks)            # warm-up, small patch
        x = torch.randn(n, 1, *PATCH)
        tgt = torch.randn_like(x)
        nsteps, t0 = 0, time.perf_counter()
        while True:
            orc.train_step(net, opt, x, tgt, tasks)
            nsteps += 1
            dt = time.perf_counter() - t0
            # at least three timed steps per organisation (the driver's run had timed two), bounded at ~40 s
            if (nsteps >= 3 and dt + dt / nsteps > 8.0) or (nsteps >= 3 and dt > 10.0) or dt + dt / nsteps > 40.0:
                break
        res[style] = {'value': vox * nsteps / dt, 'steps': nsteps, 'seconds': dt}
        del net, opt
    orc.REFERENCE_STYLE = False
    best = max(res, key=lambda k: res[k]['value'])
    return {'value': res[best]['value'], 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'variant': best,
            'reference_style_value': res['reference-style']['value'], 'vectorised_value': res['vectorised']['value'],
            'host_hw_threads': os.cpu_count(),
            'sample': 'full mult_chan=32 network, fp32 train steps (fwd+bwd+Adam) on batch 2 of 1x32x64x64, tasks (3, 7): '
                      '%d step(s) in %.1f s per-sample loop as the reference organises it, %d step(s) in %.1f s vectorised; '
                      '`value` is the faster of the two' % (res['reference-style']['steps'], res['reference-style']['seconds'],
                                                            res['vectorised']['steps'], res['vectorised']['seconds'])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--batch', type=int, default=0,
                    help='patches per GPU (default: 8 on one GPU = configs[1], 24 per rank on several = configs[3])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-inputs', action='store_true', help='inputs start in pinned host memory every step (PCIe-inclusive rate)')
    ap.add_argument('--device-volumes', action='store_true',
                    help='every step draws a fresh batch from device-resident volumes (repmode_amd.data.DeviceVolumes: random crop + '
                         'flips like the reference\'s data_aug, one launch) instead of re-using one synthetic batch')
    ap.add_argument('--no-prof', action='store_true', help='do not record per-launch HIP events')
    ap.add_argument('--no-fwd', action='store_true', help='skip the forward-only measurement')
    ap.add_argument('--grad-compress', default='fp32', choices=['fp32', 'bf16', 'auto'],
                    help="N > 1: the gradient buckets' dtype -- float32 (default: what the reference averages in), bfloat16, or "
                         "'auto' = chosen by distributed.pick_grad_dtype from the first measured backward passes")
    ap.add_argument('--graph', action='store_true', help='replay the train step as one HIP graph (N = 1; DESIGN.md 3.5)')
    ap.add_argument('--prof-all', action='store_true', help='record HIP events for every library kernel, not only conv5_igemm')
    ap.add_argument('--dump-launches', default=None, help='write per-launch (kind, ms, TFLOP/s or TB/s) of the last timed step as JSON')
    args = ap.parse_args()

    from repmode_amd import _lib, ops as ops_, distributed as dist_
    from repmode_amd.model import Model

    # REPMODE_BENCH_SHARE_GPU=1: developer check of the N > 1 code path on a one-GPU box -- every rank on cuda:0, gloo
    # transport (RCCL refuses two ranks on one device); never a measurement
    share_gpu = bool(os.environ.get('REPMODE_BENCH_SHARE_GPU'))
    if share_gpu:
        os.environ['LOCAL_RANK'] = '0'
    rank, world, local = dist_.init_from_env(backend='gloo' if share_gpu else None)
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
                  distributed=world > 1, hip_graph=world == 1 and args.graph, grad_compress=args.grad_compress)
    b = args.batch or (BATCH_1GPU if world == 1 else BATCH_MULTI)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    signal = torch.randn(b, 1, *PATCH, device=device, generator=gen)
    target = torch.randn(b, 1, *PATCH, device=device, generator=gen)
    if args.host_inputs:
        # the PCIe-inclusive variant: the batch is handed over in (pinned) host memory every step, as a DataLoader
        # would; never the headline `value` (inputs resident in HBM), reported in DESIGN.md section 5
        signal, target = signal.cpu().pin_memory(), target.cpu().pin_memory()
    task = (torch.arange(b) + rank * b) % NUM_TASKS            # CPU int tensor, like the DataLoader's
    volumes = None
    if args.device_volumes:
        # the input side of SURVEY 8f.4 inside the timed region: 12 synthetic 64x160x160 volume pairs (one per task) live in
        # HBM; each step crops + flips a new batch of them (numpy draws in the reference's order, one HIP launch)
        import numpy as np
        from repmode_amd.data import DeviceVolumes
        volumes = DeviceVolumes(device, PATCH, 0.5)
        for t_ in range(NUM_TASKS):
            v = torch.randn(2, 64, 160, 160, generator=torch.Generator().manual_seed(t_))
            volumes.add(v[0], v[1], t_)
        rng = np.random.RandomState(rank)
        vol_idx = [int(t_) for t_ in task]

    def next_batch():
        if volumes is None:
            return signal, target, task
        return volumes.sample_batch(vol_idx, rng)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # N > 1 with --grad-compress auto: the gradient buckets' dtype is chosen from the first Model.RULE_STEPS distributed steps:
    # a one-time setup (one synchronisation) that a short --warmup must not push into the timed region
    settle = max(0, Model.RULE_STEPS - args.warmup) if (world > 1 and args.grad_compress == 'auto') else 0
    for _ in range(settle + args.warmup):
        model.do_train_iter(*next_batch())
    barrier()
    # HIP events around the dominant kernel's launches only by default, and on every PROF_EVERY-th timed step (each
    # event pair costs ~4 us of stream time: ~0.5 ms per step if every launch of every step were bracketed)
    sample = 1 if (args.prof_all or args.dump_launches) else PROF_EVERY
    if not args.no_prof:
        _lib.prof_enable(1 if (args.prof_all or args.dump_launches) else 2)
    profiled_steps = 0
    overlap = ops_.get_overlap()
    hosted = ops_.get_tail_jobs()
    # one event per step boundary (on the stream the step is launched on): the GPU time of every timed step, so that the line
    # can also quote the steady-state step -- the mean over the steps that were NOT launched kernel by kernel
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_profiled = []
    t0 = time.perf_counter()
    for step in range(args.steps):
        step_ev[step].record()
        if not args.no_prof:
            on = step % sample == 0
            _lib.prof_pause(not on)
            profiled_steps += on
            # a step whose launches are timed one by one runs them one after the other: with the GatRep kernels beside
            # the convolutions on a second stream, an event pair would time the pair of kernels, not the kernel
            ops_.set_overlap(overlap and not on)
            # likewise the small jobs that otherwise ride in the first workgroups of the data-gradient conv launches (gate
            # backward, layout transposes: csrc/tail_jobs.h) are launched on their own on such a step, so that the events
            # around a conv5_igemm launch time the convolution alone
            ops_.set_tail_jobs(hosted and not on)
        # (a profiled step is launched kernel by kernel: the library's event pairs are not part of a captured graph)
        step_profiled.append(bool(not args.no_prof and on))
        model.do_train_iter(*next_batch(), eager=not args.no_prof and on)
    step_ev[args.steps].record()
    t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (includes waiting on a full queue)
    barrier()
    dt = time.perf_counter() - t0
    dt = dist_.max_over_ranks(dt, device)
    ops_.set_overlap(overlap)
    ops_.set_tail_jobs(hosted)
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    plain = [m for m, p_ in zip(step_ms, step_profiled) if not p_]
    ms_unprofiled = sum(plain) / len(plain) if plain else None
    loss = float(model.last_loss)
    # what ENQUEUEING one step costs the host: timed with the GPU idle at the start of the step, so that nothing waits
    # on a full queue (over many back-to-back steps the host runs ahead until the runtime blocks it, and
    # `host_issue_ms_per_step` then reads as the GPU's own step time)
    _lib.prof_pause(True)
    enq = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.do_train_iter(signal, target, task)
        enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    train_prof = {}
    if not args.no_prof and rank == 0:
        for kind in ('conv5_ws', 'conv5_igemm', 'deep_mode', 'deep_mode_dgrad', 'conv5_deep', 'conv5_thin', 'conv5_wgrad', 'gatrep_fwd',
                     'gatrep_bwd', HELPER_KIND):
            train_prof[kind] = _lib.prof_summary(kind)
        train_recs = _lib.prof_records() if args.dump_launches else None
    _lib.prof_enable(False)

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
        'data': 'synthetic' + (' (inputs in pinned host memory every step: PCIe-inclusive)' if args.host_inputs else '') +
                (' (a fresh crop + flip batch from device-resident volumes every step)' if args.device_volumes else ''),
        'config': {'workload': 'RepMode U-Net (mult_chan 32, 12 tasks, 123.9M params) full train step '
                               '(fwd + bwd + Adam), batch %d x 1x32x64x64 per GPU (%s)'
                               % (b, 'BASELINE configs[1]' if (world == 1 and b == BATCH_1GPU) else
                                  'BASELINE configs[2]: batch 24, 12 tasks, one GPU' if (world == 1 and b == BATCH_MULTI) else
                                  'BASELINE configs[3]: global batch %d' % (world * b) if b == BATCH_MULTI else 'custom batch'),
                   'global_batch': world * b, 'patch': list(PATCH), 'parallelism': 'dp%d' % world,
                   'distinct_tasks_per_rank': len(set(task.tolist())),
                   'final_loss': loss, 'host_enqueue_ms_per_step': 1e3 * sorted(enq)[len(enq) // 2],
                   'host_issue_ms_per_step': 1e3 * t_issue / args.steps, 'kernel_overlap': overlap,
                   'hip_graph': bool(model.hip_graph), 'steps_launched_kernel_by_kernel': profiled_steps,
                   'ms_per_step_unprofiled': ms_unprofiled,
                   'setup_steps_before_warmup': settle},
    }

    # ---- N > 1: what the process group looked like (so that a scaling record can show that RCCL saw N ranks), then the same
    # ranks, batch and run without the gradient all-reduce (what the collectives cost at this batch)
    if world > 1:
        out['config']['comm'] = dist_.comm_info(model)
        k = max(5, min(args.steps, 20))
        module = model.ddp if model.ddp is not None else None
        ctx = module.no_sync() if module is not None else None
        if ctx is not None:
            with ctx:
                for _ in range(3):
                    model.do_train_iter(signal, target, task)
                barrier()
                t1 = time.perf_counter()
                for _ in range(k):
                    model.do_train_iter(signal, target, task)
                barrier()
                dt_nc = dist_.max_over_ranks(time.perf_counter() - t1, device)
            out['config']['no_comm_value'] = vox_per_step * k / dt_nc
            out['config']['no_comm_ms_per_step'] = 1e3 * dt_nc / k
            out['config']['comm']['efficiency'] = out['value'] / out['config']['no_comm_value']
            out['config']['note'] = ('per-GPU batch is 24 at every N > 1 (weak scaling, configs[3]) but 8 on the N = 1 line '
                                     '(configs[1]): use value / no_comm_value, not value / (N x the N = 1 value), as the '
                                     'scaling efficiency')

    # ---- forward only (train-mode forward, no autograd graph): whole-pass rate + gate/GatRep/conv as one unit
    if rank == 0 and not args.no_fwd and not args.host_inputs:
        net = model.net
        net.train()
        kf = 10
        with torch.no_grad():
            for _ in range(3):
                net(signal, task)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(kf):
                net(signal, task)
            torch.cuda.synchronize()
            dt_f = time.perf_counter() - t1
            _lib.prof_enable(1)
            for _ in range(2):
                net(signal, task)
            torch.cuda.synchronize()
        # every launch of the MoDE blocks in a forward pass: the convolution kernels (conv5_ws, deep_mode, conv5_igemm /
        # conv5_deep where round 4's launches still run, the thin layers' own), gate softmax + GatRep (+ expert layout), and the
        # blocks' helper kernels (box means, gemm3, expert_mix, the per-expert decoder block's concatenation)
        n_c, ms_c, fl_c = (sum(v) for v in zip(*(_lib.prof_summary(k) for k in CONV_KINDS)))
        n_g, ms_g, _ = _lib.prof_summary('gatrep_fwd')
        n_h, ms_h, _ = _lib.prof_summary(HELPER_KIND)
        _lib.prof_enable(False)
        vps = b * PATCH[0] * PATCH[1] * PATCH[2] * kf / dt_f
        peak = PEAK_TFLOPS[args.dtype]
        ms_unit = ms_c + ms_g + ms_h
        unit_tflops = fl_c / (ms_unit * 1e-3) / 1e12 if ms_unit > 0 else 0.0
        out['fwd'] = {'value': vps, 'unit': 'voxels/s', 'ms_per_pass': 1e3 * dt_f / kf,
                      'whole_pass_tflops': vps * FWD_FLOP_PER_VOXEL / 1e12,
                      'whole_pass_frac': vps * FWD_FLOP_PER_VOXEL / 1e12 / peak,
                      'gatrep_conv_unit': {
                          'what': 'EVERY kernel the 19 MoDE blocks launch in one forward pass up to (not including) BatchNorm: gate '
                                  'softmax + GatRep (+ expert layout), every convolution launch (conv5_ws, deep_mode, conv5_igemm, '
                                  'conv5_deep, the one-channel layers\' kernels) and the helper kernels (box means, 1x1-expert GEMMs, '
                                  'gate mix, skip concatenation of the per-expert decoder block), event-timed one by one; FLOPs = '
                                  'the MoDE convs\' algorithmic 2*V*Cin*Cout*125',
                          'conv_ms': ms_c / 2, 'gatrep_ms': ms_g / 2, 'helper_ms': ms_h / 2, 'conv_launches': n_c // 2,
                          'gatrep_launches': n_g // 2, 'helper_launches': n_h // 2, 'achieved': unit_tflops, 'peak': peak,
                          'unit': 'TFLOP/s', 'frac': unit_tflops / peak,
                          'conv_only_frac': (fl_c / (ms_c * 1e-3) / 1e12 / peak) if ms_c > 0 else 0.0}}

    if rank == 0:
        if not args.no_prof:
            kinds = {}
            for kind, (n, ms, work) in train_prof.items():
                if n == 0:
                    continue          # kind not recorded (default: the dominant kernel only; --prof-all for all)
                kinds[kind] = {'launches': n, 'ms_per_step': ms / max(profiled_steps, 1),
                               'rate': (work / (ms * 1e-3) / 1e12) if ms > 0 else None}   # TFLOP/s or TB/s
            # the MoDE convolution of conv5_igemm.hip: conv5_ws_kernel (levels 0-1, since round 3) + conv5_igemm_kernel (levels 2-4)
            n, ms, flops = (sum(v) for v in zip(*(train_prof[k] for k in MAIN_KINDS)))
            achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak = PEAK_TFLOPS[args.dtype]
            # (the PMC passes know kernel SYMBOLS: both directions of deep_mode_kernel share one per-launch average)
            tr_parts = [(pmc_traffic({'deep_mode_dgrad': 'deep_mode'}.get(k, k), b, args.dtype), train_prof[k][0])
                        for k in MAIN_KINDS if train_prof[k][0]]
            traffic_file = next((t[1] for t, _ in tr_parts if t[1]), None)
            traffic = None
            if tr_parts and all(t[0] for t, _ in tr_parts):
                traffic = sum(t[0] * c for t, c in tr_parts) / sum(c for _, c in tr_parts)
            nslots = len(set(task.tolist()))
            alg = conv_algorithmic_bytes(b, nslots, train_prof['deep_mode'][0] + train_prof['deep_mode_dgrad'][0] > 0)
            alg_bytes = sum(alg[k][0] for k in MAIN_KINDS)
            alg_n = max(sum(alg[k][1] for k in MAIN_KINDS), 1)
            # the filter gradient -- the step's largest kernel family after conv5_ws -- beside them: its launches per profiled
            # step carry the whole step's algorithmic bytes (a skip connection's layer is two launches over one filter gradient)
            wg_n = train_prof['conv5_wgrad'][0]
            alg['conv5_wgrad'] = (conv5_wgrad_algorithmic_bytes(b, nslots) * max(profiled_steps, 1), wg_n)
            by_kernel = {}

            def entry(kinds_, traffic_kind):
                kn, kms, kfl = (sum(v) for v in zip(*(train_prof[k] for k in kinds_)))
                if not kn:
                    return None
                ab, an = sum(alg[k][0] for k in kinds_), sum(alg[k][1] for k in kinds_)
                ktr = pmc_traffic(traffic_kind, b, args.dtype)[0] if traffic_kind else None
                return {'launches': kn, 'avg_launch_ms': kms / kn, 'achieved': kfl / (kms * 1e-3) / 1e12,
                        'frac': kfl / (kms * 1e-3) / 1e12 / peak, 'traffic': ktr, 'algorithmic_bytes_per_launch': ab / max(an, 1),
                        'traffic_over_algorithmic': (ktr / (ab / an)) if ktr and an else None}
            for sym, kinds_, tk in (('conv5_ws_kernel', ('conv5_ws',), 'conv5_ws'),
                                    ('deep_mode_kernel', ('deep_mode', 'deep_mode_dgrad'), 'deep_mode'),
                                    ('conv5_igemm_kernel', ('conv5_igemm',), 'conv5_igemm'),
                                    ('conv5_wgrad_bf16_kernel', ('conv5_wgrad',), 'conv5_wgrad')):
                e = entry(kinds_, tk)
                if e:
                    by_kernel[sym] = e
            if 'deep_mode_kernel' in by_kernel:
                # (the PMC pass cannot tell the two directions of one kernel symbol apart: no traffic on the halves)
                by_kernel['deep_mode_kernel']['forward'] = entry(('deep_mode',), None)
                by_kernel['deep_mode_kernel']['data_gradient'] = entry(('deep_mode_dgrad',), None)
            r4n, r4ms, r4fl = (sum(v) for v in zip(*(train_prof[k] for k in ROUND4_SCOPE)))
            out['roofline'] = {'kernel': ' + '.join(dict.fromkeys(KERNEL_SYMBOL[k] for k in MAIN_KINDS if train_prof[k][0])) +
                                         ' (the MoDE convolution, forward and data gradient, of every block but the one-channel ends)',
                               'by_kernel': by_kernel, 'bound': 'mfma', 'achieved': achieved, 'peak': peak,
                               'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_file,
                               'launches': n, 'avg_launch_ms': ms / max(n, 1),
                               'flops_per_launch': flops / max(n, 1),
                               'algorithmic_bytes_per_launch': alg_bytes / alg_n,
                               'traffic_over_algorithmic': (traffic / (alg_bytes / alg_n)) if traffic else None,
                               'round4_scope': {'what': "rounds 1-4 quoted `roofline` over conv5_ws + the per-expert levels' FORWARD "
                                                        'launches only (their data gradient ran in conv5_deep, outside the figure): '
                                                        'the same scope on this build, for comparison with BENCH_r01..r04',
                                                'launches': r4n, 'frac': (r4fl / (r4ms * 1e-3) / 1e12 / peak) if r4ms > 0 else None}}
            # all forward / data-gradient convolution kernels together (the dominant one above + the deep levels' + the thin layers')
            n_a, ms_a, fl_a = (sum(v) for v in zip(*(train_prof[k] for k in CONV_KINDS)))
            out['roofline']['all_conv_kernels'] = {'kernels': [k for k in CONV_KINDS if train_prof[k][0]], 'launches': n_a,
                                                   'achieved': (fl_a / (ms_a * 1e-3) / 1e12) if ms_a > 0 else 0.0,
                                                   'frac': (fl_a / (ms_a * 1e-3) / 1e12 / peak) if ms_a > 0 else 0.0,
                                                   'ms_per_step': ms_a / max(profiled_steps, 1)}
            out['kernels'] = kinds
            if args.dump_launches:
                per = len(train_recs) // max(profiled_steps, 1)
                last = [{'kind': k, 'us': ms * 1e3, 'rate': (w / (ms * 1e-3) / 1e12) if ms > 0 else None,
                         'work': w} for k, ms, w in train_recs[-per:]]
                with open(args.dump_launches, 'w') as f:
                    json.dump(last, f, indent=0)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()               # (rank 0 prints after its forward-only leg: every rank leaves the group together)
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
