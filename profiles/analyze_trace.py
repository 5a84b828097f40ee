#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (--kernel-trace --stats --output-format csv) of bench.py.

    python profiles/analyze_trace.py gpurun_out/profN [steps_in_trace]

Prints (a) the per-kernel totals (rocprof's own *_kernel_stats.csv, shortened) and (b) for the last
traced train step the conv5_igemm forward launches per network layer with their algorithmic
TFLOP/s (2 * voxels * Cin * Cout * 125 / duration), which is what bench.py's roofline aggregates.
"""
import csv
import glob
import re
import sys

LAYERS = [('enc1.c1', 1, 32, 0), ('enc1.c2', 32, 32, 0), ('enc2.c1', 32, 64, 1), ('enc2.c2', 64, 64, 1),
          ('enc3.c1', 64, 128, 2), ('enc3.c2', 128, 128, 2), ('enc4.c1', 128, 256, 3), ('enc4.c2', 256, 256, 3),
          ('bot.c1', 256, 512, 4), ('bot.c2', 512, 512, 4), ('dec4.c1', 512, 256, 3), ('dec4.c2', 256, 256, 3),
          ('dec3.c1', 256, 128, 2), ('dec3.c2', 128, 128, 2), ('dec2.c1', 128, 64, 1), ('dec2.c2', 64, 64, 1),
          ('dec1.c1', 64, 32, 0), ('dec1.c2', 32, 32, 0), ('out', 32, 1, 0)]
BATCH, VOX0 = 8, 32 * 64 * 64


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', name)[:78]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    stats = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
    trace = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('== per-kernel totals over %d traced steps (%.2f ms/step of GPU time) ==' % (steps, tot / 1e6 / steps))
    for r in rows[:24]:
        print('%-80s calls %5s  %8.3f ms/step  avg %8.1f us  %5.1f%%' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3,
            100 * float(r['TotalDurationNs']) / tot))
    tr = list(csv.DictReader(open(trace)))
    conv = [r for r in tr if 'conv5_igemm' in r['Kernel_Name']]
    per = len(conv) // steps
    last = conv[(steps - 1) * per:]
    print('\n== conv5_igemm, forward launches of the last step (batch %d, 32x64x64) ==' % BATCH)
    total = 0.0
    for (name, ci, co, lvl), r in zip(LAYERS, last[:len(LAYERS)]):
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        flops = 2.0 * BATCH * (VOX0 // 8 ** lvl) * ci * co * 125
        total += dur
        print('%-8s %4d->%-4d L%d  %8.1f us  %8.1f TFLOP/s' % (name, ci, co, lvl, dur, flops / dur / 1e6))
    rest = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last[len(LAYERS):])
    print('forward %.0f us, data-gradient launches %.0f us (%d launches)' % (total, rest, len(last) - len(LAYERS)))
    for key in ('conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd'):
        ks = [r for r in tr if key in r['Kernel_Name']]
        per = len(ks) // steps
        t = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ks[(steps - 1) * per:])
        print('%-12s %4d launches/step  %8.0f us/step' % (key, per, t))


if __name__ == '__main__':
    main()
