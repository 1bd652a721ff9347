#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per conv layer of the LAST step (32 conv/tconv launches)."""
import csv, sys, collections
sys.path.insert(0, '.')
from lass_amd import arch

def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return d

def layers(B=16):
    order = []
    for r in arch.conv_layer_table(1024):
        if r['kind'] in ('3x3', 'tconv'):
            order.append([r['name'], r['macs']])
        elif r['name'].endswith('.shortcut'):
            order[-1][1] += r['macs']; order[-1][0] += '+sc'
    return order

if __name__ == '__main__':
    d = load(sys.argv[1])
    convs = [v for v in d.values() if 'conv_kernel' in v['name']][-32:]
    ctrs = [c for c in convs[0] if c not in ('name', 't0', 't1')]
    print('layer'.ljust(28), 'ms'.rjust(7), 'TF'.rjust(6), ' '.join(c[-18:].rjust(18) for c in ctrs))
    for (name, macs), v in zip(layers(), convs):
        dt = (v['t1'] - v['t0']) / 1e9
        print(name.ljust(28), f"{dt*1e3:7.3f} {2*16*macs/dt/1e12:6.1f}", ' '.join(f"{v[c]:18.4g}" for c in ctrs))
