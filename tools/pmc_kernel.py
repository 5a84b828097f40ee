#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel-name substring:  python tools/pmc_kernel.py <out dir> <substring>"""
import csv, glob, os, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
tot, n = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
for k in sorted(tot):
    print('  %-28s %16.0f  per launch %14.0f  (%d launches)' % (k, tot[k], tot[k] / n[k], n[k]))
