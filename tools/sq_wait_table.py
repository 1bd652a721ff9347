#!/usr/bin/env python3
"""Where the waves of each launch of the SHIPPED pipeline spend their cycles: shares of SQ_WAVE_CYCLES that are parked
(SQ_WAIT_ANY: s_waitcnt / s_barrier), issue-stalled (SQ_WAIT_INST_ANY: the instruction cannot issue - matrix / vector pipe busy,
operand dependency, memory queue full) and issuing (SQ_ACTIVE_INST_ANY), plus the vector-ALU, LDS and vector-memory shares, from
two rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 1 --dtype D --modes none` (tools/gpu_sq_counters.sh), last step.
Usage: sq_wait_table.py sq_pass1.csv sq_pass2.csv > table.md"""
import collections
import csv
import re
import sys


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return list(d.values())


def last_step(disp):
    last = max(i for i, v in enumerate(disp) if 'stft2_kernel' in v['name'] and 'istft2' not in v['name'])
    return disp[last:]


if __name__ == '__main__':
    a, b = last_step(load(sys.argv[1])), last_step(load(sys.argv[2]))
    print('| kernel (launch order) | us | parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | of which VALU | LDS | VMEM | LDS issue stall |')
    print('|---|---|---|---|---|---|---|---|---|')
    for va, vb in zip(a, b):
        n = re.sub(r'\(anonymous namespace\)::|^void ', '', va['name']).split('(')[0]
        wc, wc2 = va.get('SQ_WAVE_CYCLES', 0) or 1, vb.get('SQ_WAVE_CYCLES', 0) or 1
        f = lambda d, k, w: d.get(k, 0.0) / w  # noqa: E731
        print(f"| `{n}` | {(va['t1'] - va['t0']) / 1e3:.1f} | {f(va, 'SQ_WAIT_ANY', wc):.2f} | {f(va, 'SQ_WAIT_INST_ANY', wc):.2f} | "
              f"{f(va, 'SQ_ACTIVE_INST_ANY', wc):.2f} | {f(va, 'SQ_ACTIVE_INST_VALU', wc):.2f} | {f(vb, 'SQ_ACTIVE_INST_LDS', wc2):.2f} | "
              f"{f(vb, 'SQ_ACTIVE_INST_VMEM', wc2):.2f} | {f(vb, 'SQ_WAIT_INST_LDS', wc2):.2f} |")
