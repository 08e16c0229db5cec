#!/usr/bin/env python3
"""Static instruction mix of every kernel in an assembly dump of ptx.hip (scripts/regs.sh writes one).
usage: isa_mix.py /tmp/ptx_regs.s [kernel-name-substring]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for n, (i, name) in enumerate(starts):
    end = next((j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end')), len(lines))
    if want not in name:
        continue
    c = collections.Counter()
    tot = 0
    for l in lines[i + 1:end]:
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        op = t.split()[0]
        tot += 1
        if op.startswith('v_'):
            if 'f64' in op:
                c['v_f64'] += 1
            elif op.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log')):
                c['v_trans'] += 1
            elif op.startswith('v_div'):
                c['v_div'] += 1
            elif op.startswith(('v_cndmask', 'v_cmp')):
                c['v_cmp_sel'] += 1
            elif op.startswith(('v_mov', 'v_accvgpr')):
                c['v_mov'] += 1
            elif op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')):
                c['v_lane'] += 1
            else:
                c['v_other'] += 1
        elif op.startswith('s_waitcnt'):
            c['s_waitcnt'] += 1
        elif op.startswith(('s_load', 's_buffer_load')):
            c['s_load'] += 1
        elif op.startswith('s_'):
            c['s_other'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            c['vmem'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        else:
            c['other'] += 1
    print('%-46s total %5d  %s' % (name.replace('_ZN12_GLOBAL__N_1', '')[:46], tot, dict(sorted(c.items()))))
