#!/usr/bin/env python3
"""Per-kernel totals of tools/step_launches.py listings, side by side:  python tools/kernel_totals.py a.txt b.txt"""
import re, sys, collections
def load(f):
    tot = collections.OrderedDict()
    tail = ''
    for l in open(f):
        m = re.match(r'\s*([\d.]+) us\s+gap\s+([\d.-]+)\s+grid\s+(\d+) x\s+(\d+) x\s+(\d+)\s+wg\s+(\d+)\s+(.*)', l)
        if not m:
            tail = l.strip(); continue
        name = re.sub(r'<.*', '', m.group(7).strip())
        d = tot.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += float(m.group(1))
    return tot, tail
a, ta = load(sys.argv[1]); b, tb = load(sys.argv[2])
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0, 0])[1] + b.get(k, [0, 0])[1])):
    x, y = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
    print('%-52s n %3d %8.1f us | n %3d %8.1f us | %+7.1f' % (k[:52], x[0], x[1], y[0], y[1], y[1] - x[1]))
print(ta); print(tb)
