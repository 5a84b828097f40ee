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
    for key in ('conv5_igemm', 'conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd', 'bn_', 'k2s2', 'expert_mix', 'box_sum'):
        ks = [r for r in tr if key in r['Kernel_Name']]
        if not ks:
            continue
        t = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ks) / steps
        print('%-12s %6.1f launches/step  %8.0f us/step' % (key, len(ks) / steps, t))


if __name__ == '__main__':
    main()
