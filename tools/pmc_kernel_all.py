#!/usr/bin/env python3
"""Per kernel name: sums of two rocprofv3 --pmc counters and their ratio:  python tools/pmc_kernel_all.py <out dir> <num> <den>"""
import csv, glob, os, re, sys
from collections import defaultdict
d, num, den = sys.argv[1], sys.argv[2], sys.argv[3]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*', '', name)[:70]
        tot[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == den:
            cnt[name] += 1
rows = sorted(tot.items(), key=lambda kv: -kv[1].get(den, 0))
for name, c in rows[:28]:
    if c.get(den, 0) > 0:
        print('%-72s %5d launches  %s %12.0f  %s %12.0f  ratio %.2f' % (name, cnt[name], num, c.get(num, 0), den, c[den], c.get(num, 0) / c[den]))
