#!/usr/bin/env python3
"""Per-layer MFMA utilisation from a rocprofv3 --pmc pass over tools/conv_bench.py (one iteration, B=16, T=1024):
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)          (MI355X_MICROARCH.md: the SQ
counter counts cycles per SIMD summed over the chip, GRBM_GUI_ACTIVE is summed over the 8 XCDs)
and the rates each conv launch sustained.  Usage: pmc_table.py counter_collection.csv [f32|bf16] > table.md"""
import collections
import csv
import sys

sys.path.insert(0, '.')
from lass_amd import arch


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return d


if __name__ == '__main__':
    d = load(sys.argv[1])
    mode = sys.argv[2] if len(sys.argv) > 2 else 'f32'
    convs = [v for v in d.values() if ('conv_kernel' in v['name'] or 'wino_kernel' in v['name'] or 'conv_bf16_kernel' in v['name'])
             and 'relayout' not in v['name']]
    # shortcut rows of arch.conv_layer_table come AFTER conv2: attach them
    rows = arch.conv_layer_table(1024)
    seq = []
    for i, r in enumerate(rows):
        if r['kind'] == '3x3':
            sc = 0
            for j in (i + 1, i + 2):
                if j < len(rows) and rows[j]['name'].endswith('.shortcut') and r['name'].endswith('.conv2') and \
                        rows[j]['name'].rsplit('.', 1)[0] == r['name'].rsplit('.', 1)[0]:
                    sc = rows[j]['macs']
            seq.append((r['name'] + ('+sc' if sc else ''), r['macs'], sc, r['kind'], r['h']))
        elif r['kind'] == 'tconv':
            seq.append((r['name'], r['macs'], 0, 'tconv', r['h']))
    # conv_bench runs every job (block = conv1 + conv2, .up = one launch) once as warm-up and once timed (--iters 1):
    # 2 x k consecutive dispatches per job, of which the last k are kept
    kept, pos, i = [], 0, 0
    while i < len(seq):
        k = 1 if seq[i][3] == 'tconv' else 2
        kept += convs[pos + k:pos + 2 * k]
        pos += 2 * k
        i += k
    assert pos == len(convs), (pos, len(convs))
    convs = kept
    B = 16
    peak = 157.3 if mode == 'f32' else 2500.0
    print(f'| launch | ms | algorithmic TFLOP/s | executed TFLOP/s | executed / {peak:g} peak | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES) | clock GHz |')
    print('|---|---|---|---|---|---|---|')
    tot_ms = tot_alg = tot_exe = 0.0
    for (name, macs, sc, kind, h), v in zip(seq, convs):
        dt = (v['t1'] - v['t0']) / 1e9
        alg = 2.0 * B * (macs + sc)
        wino = mode == 'f32' and kind == '3x3' and h % 2 == 0 and 'wino' in v['name']
        exe = 2.0 * B * (macs * (4.0 / 9.0 if wino else 1.0) + sc) * (3.0 if mode == 'bf16x3' else 1.0)
        gui = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024.0) if gui else float('nan')
        clk = gui / dt / 1e9 if dt > 0 else float('nan')
        tot_ms += dt * 1e3; tot_alg += alg; tot_exe += exe
        print(f'| {name} | {dt*1e3:.3f} | {alg/dt/1e12:.1f} | {exe/dt/1e12:.1f} | {exe/dt/1e12/peak:.3f} | {busy:.3f} | {clk:.2f} |')
    print(f'| **all {len(seq)} launches** | {tot_ms:.3f} | {tot_alg/tot_ms/1e9:.1f} | {tot_exe/tot_ms/1e9:.1f} | {tot_exe/tot_ms/1e9/peak:.3f} | | |')
