#!/usr/bin/env python3
"""Print every kernel dispatch of a rocprofv3 --pmc counter_collection.csv whose name matches a filter, with its counters."""
import csv, sys, collections
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = int(r['Dispatch_Id'])
    e = d.setdefault(k, {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
flt = sys.argv[2] if len(sys.argv) > 2 else 'wino'
rows = [v for v in d.values() if flt in v['name']]
if rows:
    ctrs = [c for c in rows[0] if c not in ('name', 't0', 't1')]
    print('kernel'.ljust(44), 'us'.rjust(8), ' '.join(c[-22:].rjust(22) for c in ctrs))
    for v in rows:
        nm = v['name'].replace('void (anonymous namespace)::', '').split('(')[0][:44]
        print(nm.ljust(44), f"{(v['t1']-v['t0'])/1e3:8.1f}", ' '.join(f"{v.get(c,0):22.5g}" for c in ctrs))
