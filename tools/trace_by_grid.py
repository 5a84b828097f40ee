#!/usr/bin/env python3
"""Per (kernel, grid, workgroup) average duration of a rocprofv3 --kernel-trace CSV directory: the per-launch view of a
microbenchmark (tools/*_microbench.py run under rocprofv3), where a kernel's launches differ by layer size.
    python tools/trace_by_grid.py <dir> [name filter] [skip first N launches of each group]"""
import csv, glob, re, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tr = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
groups, order = {}, []
for r in tr:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    name = re.sub(r'\(.*', '', name)
    if flt and not re.search(flt, name):
        continue
    key = (name[:70], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
    if key not in groups:
        groups[key] = []; order.append(key)
    groups[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for key in order:
    v = groups[key][skip:] or groups[key]
    v2 = sorted(v)
    print('%-70s grid %8s %5s %3s wg %5s  n %4d  avg %8.1f us  med %8.1f  min %8.1f' % (*key, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
