#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (one counter set per pass, as MI355X_MICROARCH.md prescribes) into per-kernel-family HBM
traffic per launch and MFMA utilisation.

    python profiles/pmc_summary.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/, pmc_MFMA/> <batch> <dtype>  > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half
of the bytes of a wide (16 B/lane) coalesced streaming read, which is how these kernels read -> the read side is doubled.
WRITE_SIZE is used as reported (uncalibrated).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs)
with cycles = GRBM_GUI_ACTIVE / 8: the per-dispatch value rocprofv3 writes is the SUM over the 8 XCDs (checked: it is
17-22 k per microsecond of kernel time = 8 x the 2.1-2.7 GHz clock; and SQ_VALU_MFMA_BUSY_CYCLES of a conv launch is
exactly 32 cycles x its number of 32x32x16 MFMAs).  Same quantity as rocprofv3's derived MfmaUtil (which takes the max
over XCDs instead), summed over the family's launches.  The output carries the hash of the kernel sources (bench.kernel_source_hash) the passes were taken on: bench.py
only quotes `traffic` from a file whose hash matches the build it runs.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = ['conv5_ws', 'conv5_pipe', 'conv5_igemm', 'deep_mode', 'conv5_deep', 'thin_in1', 'thin_out1', 'conv5_wgrad', 'gatrep_fwd', 'gatrep_bwd', 'bn_', 'k2s2', 'gemm3', 'multi_tensor_apply', 'adam_']


def rows(d, name):
    fs = sorted(glob.glob('%s/pmc_%s/**/*_counter_collection.csv' % (d, name), recursive=True))
    if not fs:
        return []
    return list(csv.DictReader(open(fs[-1])))


def per_kind(rs, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        if r['Counter_Name'] != counter:
            continue
        for k in KINDS:
            if k in r['Kernel_Name']:
                agg[k][0] += 1
                agg[k][1] += float(r['Counter_Value'])
    return agg


def main():
    d, batch, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import bench
    rd, wr = per_kind(rows(d, 'FETCH_SIZE'), 'FETCH_SIZE'), per_kind(rows(d, 'WRITE_SIZE'), 'WRITE_SIZE')
    mf = rows(d, 'MFMA')
    busy, act = per_kind(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_kind(mf, 'GRBM_GUI_ACTIVE')
    out = {'kernel_source_hash': bench.kernel_source_hash(), 'batch': batch, 'dtype': dtype,
           'command': 'REPMODE_TAIL=0 rocprofv3 --pmc <counter(s)> --kernel-trace --output-format csv -- python bench.py --batch %d --steps 1 --warmup 1 '
                      '--no-cpu-baseline --no-prof --no-fwd   (one pass per counter set: FETCH_SIZE | WRITE_SIZE | '
                      'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; REPMODE_TAIL=0: the conv launches alone, without the small jobs they '
                      'otherwise host -- what bench.py times on its event-timed steps)' % batch}
    for k in KINDS:
        n = rd[k][0]
        if not n:
            continue
        read_b = 2.0 * rd[k][1] * 1024 / n
        write_b = wr[k][1] * 1024 / max(wr[k][0], 1)
        e = {'launches_profiled': n, 'hbm_read_bytes_per_launch': read_b, 'hbm_write_bytes_per_launch': write_b,
             'hbm_bytes_per_launch': read_b + write_b,
             'note': 'FETCH_SIZE x2 (gfx950 wide-load correction) + WRITE_SIZE, KiB -> bytes, averaged over launches'}
        if act[k][1] > 0:
            e['mfma_util_percent'] = 100.0 * busy[k][1] / (act[k][1] / 8.0 * 1024)
            e['mfma_note'] = 'sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) over %d launches' % act[k][0]
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
