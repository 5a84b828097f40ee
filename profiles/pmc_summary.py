#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (one counter per pass, as MI355X_MICROARCH.md prescribes) into
per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
    rocprofv3 --pmc WRITE_SIZE ...                                   (separate run)
    python profiles/pmc_summary.py gpurun_out > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, which is how these
kernels read -> the read side is doubled.  WRITE_SIZE is used as reported (uncalibrated).
"""
import collections
import csv
import glob
import json
import sys

KINDS = ['conv5_igemm', 'conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd']


def load(d, counter):
    f = sorted(glob.glob('%s/pmc_%s/*/*_counter_collection.csv' % (d, counter)))[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        for k in KINDS:
            if k in r['Kernel_Name']:
                agg[k][0] += 1
                agg[k][1] += float(r['Counter_Value'])
    return agg


def main():
    d = sys.argv[1]
    rd, wr = load(d, 'FETCH_SIZE'), load(d, 'WRITE_SIZE')
    out = {}
    for k in KINDS:
        n = rd[k][0]
        if not n:
            continue
        read_b = 2.0 * rd[k][1] * 1024 / n
        write_b = wr[k][1] * 1024 / max(wr[k][0], 1)
        out[k] = {'launches_profiled': n, 'hbm_read_bytes_per_launch': read_b, 'hbm_write_bytes_per_launch': write_b,
                  'hbm_bytes_per_launch': read_b + write_b,
                  'note': 'FETCH_SIZE x2 (gfx950 wide-load correction) + WRITE_SIZE, KiB -> bytes, averaged over launches'}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
