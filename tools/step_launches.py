#!/usr/bin/env python3
"""Every launch of the last complete train step of a rocprofv3 kernel trace of bench.py, in order: duration, idle gap before
it, grid.  (Steps are delimited by the fused Adam launches, as in profiles/analyze_trace.py.)
    python tools/step_launches.py <trace dir> [max duration us to list, default: all]"""
import csv, glob, re, sys
d = sys.argv[1]
cap = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
tr = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(tr) if 'FusedAdam' in r['Kernel_Name'] or 'adam_' in r['Kernel_Name'].lower() or 'FusedOptimizer' in r['Kernel_Name']]
# groups of consecutive adam launches
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
lo, hi = ends[-2] + 1, ends[-1] + 1
prev = int(tr[lo - 1]['End_Timestamp'])
tot = gap_tot = 0.0
for r in tr[lo:hi]:
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev) / 1e3
    prev = int(r['End_Timestamp'])
    tot += dur; gap_tot += max(gap, 0)
    if dur <= cap:
        print('%8.1f us  gap %6.1f  grid %8s x %4s x %3s  wg %4s  %s' % (dur, gap, r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''), r.get('Workgroup_Size_X', '?'), name[:110]))
print('%d launches, %.1f us busy, %.1f us idle' % (hi - lo, tot, gap_tot))
