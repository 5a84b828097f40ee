#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (--kernel-trace --stats --output-format csv) of bench.py.

    python profiles/analyze_trace.py gpurun_out/profN [steps_in_trace]

Prints the per-kernel totals (rocprof's own *_kernel_stats.csv, shortened, per train step) and the time per
step of each library kernel family.  Per-launch algorithmic rates come from the library's own HIP-event
records instead: `python bench.py --prof-all --dump-launches f.json` + `profiles/launch_table.py f.json`.
"""
import csv
import glob
import re
import sys

def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', name)[:78]


def main():
    d = sys.argv[1]
    stats = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
    trace = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    tr = list(csv.DictReader(open(trace)))
    # train steps in the trace: counted from the trace itself (a step ends with a run of consecutive fused-Adam launches;
    # bench.py runs warm-up + timed + 5 host-enqueue-timing steps), the command-line number only as a fall-back
    steps = count_steps(tr) or (int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('== per-kernel totals over %d traced steps (%.2f ms/step of GPU time) ==' % (steps, tot / 1e6 / steps))
    for r in rows[:24]:
        print('%-80s calls %5s  %8.3f ms/step  avg %8.1f us  %5.1f%%' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3,
            100 * float(r['TotalDurationNs']) / tot))
    ig = [r for r in rows if 'conv5_igemm_kernel' in r['Name'] or 'conv5_ws_kernel' in r['Name'] or 'conv5_pipe_kernel' in r['Name'] or
          'deep_mode_kernel' in r['Name']]
    if ig:
        calls = sum(int(r['Calls']) for r in ig)
        print('conv5_ws_kernel + deep_mode_kernel + conv5_igemm_kernel, all %d instantiations: %d calls, %.1f us average  (compare bench.py roofline.avg_launch_ms)'
              % (len(ig), calls, sum(float(r['TotalDurationNs']) for r in ig) / calls / 1e3))
    for key in ('conv5_ws', 'conv5_igemm', 'deep_mode', 'conv5_deep', 'thin_', 'conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd', 'bn_', 'k2s2', 'expert_mix', 'box_sum'):
        ks = [r for r in tr if key in r['Kernel_Name']]
        if not ks:
            continue
        t = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ks) / steps
        print('%-12s %6.1f launches/step  %8.0f us/step' % (key, len(ks) / steps, t))
    families(tr)


FAMILIES = [('deep_mode level 3 forward (one launch per per-expert block)', ('deep_mode_kernel<MCfg<1, 8, 8', 'true>')),
            ('deep_mode level 3 data gradient', ('deep_mode_kernel<MCfg<1, 8, 8', 'false>')),
            ('deep_mode level 4 forward', ('deep_mode_kernel<MCfg<2, 4, 4', 'true>')),
            ('deep_mode level 4 data gradient', ('deep_mode_kernel<MCfg<2, 4, 4', 'false>')),
            ('operand check (expert_frags_verify)', 'expert_frags_verify'), ('skip concatenation', 'concat2'),
            ('conv5_deep level 3 (per-expert pair)', 'conv5_deep_kernel<DCfg<4, 8, 8'), ('conv5_deep level 4 (per-expert pair)', 'conv5_deep_kernel<DCfg<2, 4, 4'),
            ('thin layers (own kernels)', 'thin_in1_kernel'), ('thin layers (own kernels)', 'thin_out1_kernel'),
            ('conv5_ws level 2 (16-voxel bricks)', ('conv5_ws_kernel', ', 16>')),
            ('conv5_ws level 0-1 (wave-specialised)', 'conv5_ws_kernel'), ('conv5_pipe level 0-1', 'conv5_pipe_kernel'),
            ('conv5_igemm level 0-1', 'conv5_igemm_kernel<unsigned short, Cfg<4, 4, 32'),
            ('conv5_igemm level 2', 'conv5_igemm_kernel<unsigned short, Cfg<4, 4, 16'),
            ('conv5_igemm level 3', 'conv5_igemm_kernel<unsigned short, Cfg<4, 8, 8'),
            ('conv5_igemm level 4', 'conv5_igemm_kernel<unsigned short, Cfg<2, 4, 4'),
            ('conv5_wgrad level 0-1', 'conv5_wgrad_col_kernel<8, 32'), ('conv5_wgrad level 2', 'conv5_wgrad_col_kernel<16, 16'),
            ('conv5_wgrad level 2', 'conv5_wgrad_col_kernel<8, 16'), ('conv5_wgrad level 3', 'conv5_wgrad_col_kernel<8, 8'),
            ('conv5_wgrad level 4', 'conv5_wgrad_col_kernel<4, 8'),
            ('conv5_wgrad level 0-1', 'conv5_wgrad_bf16_kernel<1, 8, 32'), ('conv5_wgrad level 2', 'conv5_wgrad_bf16_kernel<1, 8, 16'),
            ('conv5_wgrad level 3', 'conv5_wgrad_bf16_kernel<2, 8, 8'), ('conv5_wgrad level 4', 'conv5_wgrad_bf16_kernel<2, 4, 8'),
            ('conv5_wgrad_thin', 'wgrad_thin'), ('gatrep_fwd', 'gatrep_fwd'), ('gatrep_bwd', 'gatrep_bwd'),
            ('expert_frags', 'expert_frags'), ('gate softmax / backward', 'gate_'), ('BatchNorm+ReLU', 'bn_'),
            ('k2s2 (stride-2 stages)', 'k2'), ('Adam (torch fused)', 'FusedAdam'), ('Adam + expert operands (adam.hip)', 'adam_'), ('box_sum', 'box_sum'),
            ('expert_mix', 'expert_mix'), ('tap_transpose', 'tap_transpose'), ('thin-layer helpers', 'shift5'),
            ('thin-layer helpers', 'thin_pack'), ('1x1 experts (gemm3)', 'gemm3'), ('box_sum', 'box_expand'), ('box_sum', 'box_'),
            ('loss (fused MSE)', 'mse_'), ('crop + flip', 'crop_flip'), ('rocBLAS', 'Cijk'), ('cat', 'CatArray'),
            ('pooled memset / fills', 'FillFunctor'), ('other PyTorch elementwise', 'at::native')]


def adam_step_ends(tr):
    """Indices (in start-time order) of the last fused-Adam launch of every train step, and the sorted trace."""
    tr = sorted(tr, key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(tr) if 'FusedAdam' in r['Kernel_Name'] or 'adam_' in r['Kernel_Name']]
    if len(adam) < 2:
        return tr, []
    per = 1
    while per < len(adam) and adam[per] == adam[per - 1] + 1:
        per += 1                                   # Adam launches per step (consecutive dispatches)
    return tr, [adam[i] for i in range(per - 1, len(adam), per)]


def count_steps(tr):
    return len(adam_step_ends(tr)[1])


def families(tr):
    """GPU time per train step by kernel family and U-Net level, over the steady steps of the trace (a step ends with
    its last fused-Adam launch; the first three steps are skipped), plus the idle time between launches."""
    tr, ends = adam_step_ends(tr)
    if len(ends) < 6:
        return
    t, c, n, idle = {}, {}, 0, 0.0
    for k in range(3, len(ends) - 1):
        step = tr[ends[k] + 1:ends[k + 1] + 1]
        n += 1
        for a, b in zip(step[:-1], step[1:]):
            idle += max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
        for r in step:
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            fam = next((f for f, pat in FAMILIES if (all(q in name for q in pat) if isinstance(pat, tuple) else pat in name)), 'other')
            t[fam] = t.get(fam, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            c[fam] = c.get(fam, 0) + 1
    tot = sum(t.values()) / n
    print('== GPU time per step by family, %d steady steps: %.2f ms busy + %.2f ms idle between launches ==' % (n, tot / 1e3, idle / n / 1e3))
    for fam in sorted(t, key=lambda f: -t[f]):
        print('%-28s %8.1f us/step  %6.1f launches  %5.1f%%' % (fam, t[fam] / n, c[fam] / n, 100 * t[fam] / n / tot))


if __name__ == '__main__':
    main()
