#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name.  usage: pmc_summary.py <dir> [<dir>...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'][:60]
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            calls[k].add(row['Dispatch_Id'])
        print('==', f)
        for k, v in agg.items():
            print(f'  {k}  dispatches={len(calls[k])}')
            for c, x in sorted(v.items()):
                print(f'      {c:32s} {x:.6g}')
