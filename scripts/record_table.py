#!/usr/bin/env python3
"""Per-kernel table from a measurement record directory (scripts/profile_record.sh):
time per image from the kernel trace, HBM traffic and utilisation figures from the counter passes.
usage: record_table.py <dir>"""
import collections
import csv
import json
import re
import sys

d = sys.argv[1]
bench = json.loads(open(d + '/bench.json').read().strip().splitlines()[-1])
steps_traced = json.loads(open(d + '/bench_traced.json').read().strip().splitlines()[-1])
images_traced = steps_traced['steps'] + steps_traced['warmup']
txt = open(d + '/pmc_summary.txt').read()
cur, data = None, collections.defaultdict(dict)
for line in txt.splitlines():
    m = re.match(r'\s{2}(\S.*?)\s+dispatches=(\d+)', line)
    if m:
        cur = m.group(1)
        data[cur]['n'] = int(m.group(2))
        continue
    m = re.match(r'\s{6}(\S+)\s+([0-9.e+]+)', line)
    if m and cur:
        data[cur][m.group(1)] = float(m.group(2))
ns = {}
for row in csv.DictReader(open(d + '/kernel_stats.csv')):
    nm = row['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    ns[nm] = float(row['AverageNs'])   # per launch; scaled to one image below (the traced run also renders untimed images)
traffic = json.load(open(d + '/hbm_traffic.json'))
print('value %.0f Msamples/s, %.1f ms per image, roofline %s' % (bench['value'], bench['ms_per_step'], json.dumps(bench['roofline'])))
print('| kernel | ms | HBM traffic | vector ALU busy | lanes active in VALU instr. | waves parked (SQ_WAIT_ANY) | L1 hit | fp32 FMA+MUL |')
tot_t = tot_b = 0
names = sorted((k for k in ns if k.startswith(('k_trace', 'k_shade', 'k_shadow', 'k_bounce')) and 'TestCount' not in k and k in data and k in traffic),
               key=lambda k: -ns[k] * traffic[k]['launches'])
for k in names:
    c = data[k]
    t = ns[k] * 1e-9 * traffic[k]['launches']   # launches per image = dispatches of the one-image counter passes
    simd_cycles = t * 2.4e9 * 1024
    gb = traffic[k]['hbm_bytes_per_launch'] * traffic[k]['launches'] / 1e9
    tot_t += t
    tot_b += gb
    print('| `%s` | %.1f | %.1f GB (%.1f TB/s) | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.1f TFLOP/s |' % (
        k, t * 1e3, gb, gb / t / 1e3, 100 * c['SQ_ACTIVE_INST_VALU'] * 2 / simd_cycles,   # (a SIMD takes a 64-wide vector instruction in two cycles)
        100 * c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64), 100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
        100 * (1 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum']),
        (2 * c['SQ_INSTS_VALU_FMA_F32'] + c['SQ_INSTS_VALU_MUL_F32']) * 64 / t / 1e12))
print('kernels %.1f ms, %.0f GB per image' % (tot_t * 1e3, tot_b))
