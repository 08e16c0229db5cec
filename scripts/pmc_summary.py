#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name.
usage: pmc_summary.py <dir> [<dir>...]            text summary
       pmc_summary.py --traffic <dir> [scene width height aa bounces queue_gib] > out.json
                                                  per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE
                                                  passes; the workload is recorded so that bench.py only quotes the
                                                  figures for the workload they were measured on

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced
read at half its bytes (128-byte requests tallied at 64), so it is doubled here; WRITE_SIZE is
exact for 16-byte-per-lane stores and float atomics."""
import collections
import csv
import glob
import json
import sys


def load(d):
    out = {}
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            calls[k].add(row['Dispatch_Id'])
        out[f] = (agg, calls)
    return out


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0]


if len(sys.argv) > 2 and sys.argv[1] == '--traffic':
    res = collections.defaultdict(dict)
    for f, (agg, calls) in load(sys.argv[2]).items():
        for k, v in agg.items():
            n = len(calls[k])
            if 'FETCH_SIZE' in v:
                res[short(k)]['fetch_bytes_per_launch'] = 2.0 * v['FETCH_SIZE'] * 1024 / n
            if 'WRITE_SIZE' in v:
                res[short(k)]['write_bytes_per_launch'] = v['WRITE_SIZE'] * 1024 / n
            res[short(k)]['launches'] = n
            # issue counters of the same record (the SQ pass): bench.py turns them into the kernel's vector-ALU busy
            # fraction with the launch time it measures itself
            if 'SQ_ACTIVE_INST_VALU' in v:
                res[short(k)]['valu_active_inst_per_launch'] = v['SQ_ACTIVE_INST_VALU'] / n
            if 'SQ_WAIT_ANY' in v and v.get('SQ_WAVE_CYCLES'):
                res[short(k)]['wave_time_waiting'] = v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']
    for k, v in res.items():
        if 'fetch_bytes_per_launch' in v and 'write_bytes_per_launch' in v:
            v['hbm_bytes_per_launch'] = v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']
    if len(sys.argv) >= 9:
        a = sys.argv[3:9]
        res['_workload'] = {'scene': a[0], 'width': int(a[1]), 'height': int(a[2]), 'aa': int(a[3]), 'bounces': int(a[4]), 'queue_gib': float(a[5])}
    print(json.dumps(res, indent=1, sort_keys=True))
    sys.exit(0)

for d in sys.argv[1:]:
    for f, (agg, calls) in load(d).items():
        print('==', f)
        for k, v in agg.items():
            print(f'  {short(k)[:60]}  dispatches={len(calls[k])}')
            for c, x in sorted(v.items()):
                print(f'      {c:32s} {x:.6g}')
