#!/usr/bin/env python3
"""Format the per-launch records of one train step written by `bench.py --prof-all --dump-launches f.json`:
kernel family, HIP-event duration, algorithmic rate (TFLOP/s for the conv kernels, TB/s for GatRep)."""
import json
import sys

L = json.load(open(sys.argv[1]))
tot = {}
for i, l in enumerate(L):
    unit = 'TB/s' if l['kind'].startswith('gatrep') or l['kind'] == 'helper' else 'TFLOP/s'
    work = l['work'] / (1e6 if unit == 'TB/s' else 1e9)
    print('%3d %-18s %8.1f us  %8.1f %-7s  %10.2f %s' % (i, l['kind'], l['us'], l['rate'] or 0, unit, work,
                                                          'MB' if unit == 'TB/s' else 'GFLOP'))
    t = tot.setdefault(l['kind'], [0, 0.0, 0.0])
    t[0] += 1
    t[1] += l['us']
    t[2] += l['work']
print()
for k, (n, us, w) in tot.items():
    print('%-18s %3d launches  %8.0f us/step  %8.1f %s aggregate' % (k, n, us, w / us / 1e6,
                                                                  'TB/s' if k.startswith('gatrep') or k == 'helper' else 'TFLOP/s'))
