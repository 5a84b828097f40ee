#!/usr/bin/env python3
"""The numbers DESIGN.md / README.md quote, read back from the committed profiles/rNN_* files:  python tools/profile_numbers.py [r03]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
P = os.path.join(ROOT, 'profiles')
for f in ('bench_line', 'bench_line_driver_args', 'bench_line_b24', 'bench_line_device_volumes'):
    d = json.load(open(os.path.join(P, '%s_%s.json' % (tag, f))))
    r, fw = d.get('roofline', {}), d.get('fwd', {})
    print('%-28s %7.3f ms  %6.1f M voxels/s  roofline %.4f (%.0f TFLOP/s, %.1f us/launch)  all conv %.4f  traffic %.1f MB = %.3f x  fwd unit %s  fwd pass %s ms' % (
        f, d['ms_per_step'], d['value'] / 1e6, r.get('frac', 0), r.get('achieved', 0), r.get('avg_launch_ms', 0) * 1e3,
        r.get('all_conv_kernels', {}).get('frac', 0), (r.get('traffic') or 0) / 1e6, r.get('traffic_over_algorithmic') or 0,
        '%.4f (conv alone %.4f)' % (fw['gatrep_conv_unit']['frac'], fw['gatrep_conv_unit']['conv_only_frac']) if fw else '-',
        '%.3f' % fw['ms_per_pass'] if fw else '-'))
    for k, v in r.get('by_kernel', {}).items():
        print('    %-20s %3d launches  %6.1f us  %6.0f TFLOP/s  %.4f  traffic %.1f MB = %.3f x' % (
            k, v['launches'], v['avg_launch_ms'] * 1e3, v['achieved'], v['frac'], (v.get('traffic') or 0) / 1e6, v.get('traffic_over_algorithmic') or 0))
    if 'cpu_baseline' in d:
        print('    cpu_baseline %.0f voxels/s on %d threads; host enqueue %.2f ms' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config']['host_enqueue_ms_per_step']))
p = json.load(open(os.path.join(P, '%s_pmc_traffic.json' % tag)))
print('pmc hash', p['kernel_source_hash'])
for k, v in p.items():
    if isinstance(v, dict) and 'mfma_util_percent' in v and v['mfma_util_percent'] > 1:
        print('    %-14s %4d launches  %.1f MB/launch  MFMA busy %.1f %%' % (k, v['launches_profiled'], v['hbm_bytes_per_launch'] / 1e6, v['mfma_util_percent']))
for f in ('bench_trace_summary', 'bench_trace_summary_conv_alone'):
    for l in open(os.path.join(P, '%s_%s.txt' % (tag, f))):
        if 'instantiations' in l or 'busy +' in l:
            print(f, '|', l.strip()[:160])
print(open(os.path.join(P, '%s_predict.txt' % tag)).read().strip().split('\n')[-1])
