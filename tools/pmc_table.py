#!/usr/bin/env python3
"""Per-launch MFMA utilisation of the SHIPPED pipeline from a rocprofv3 --pmc pass over
`bench.py --steps 1 --warmup 1 --dtype D --modes none` (B=16, 10 s clips): the conv / transposed-conv launches of the LAST
step (everything behind the last stft2_kernel dispatch), in launch order = lass_separate's order (api.hip separate_impl):
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)          (MI355X_MICROARCH.md: the SQ
counter counts cycles per SIMD summed over the chip, GRBM_GUI_ACTIVE is summed over the 8 XCDs)
and the rates each launch sustained.  Usage: pmc_table.py counter_collection.csv [f32|bf16|bf16x3] > table.md"""
import collections
import csv
import re
import sys

sys.path.insert(0, '.')
from lass_amd import arch

CONV = re.compile(r'wino4_kernel|wino32_kernel|wino_kernel|conv_kernel|conv_bf16_kernel|fused_bf16_kernel')


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return list(d.values())


def last_step(disp):
    last = max(i for i, v in enumerate(disp) if 'stft2_kernel' in v['name'] and 'istft2' not in v['name'])
    return [v for v in disp[last:] if CONV.search(v['name']) and 'relayout' not in v['name'] and 'weights' not in v['name']]


def launch_sequence(convs):
    """(name, 3x3 macs, shortcut macs, kind, h) per launch, following the dispatches (a fused block is one launch)."""
    rows = arch.conv_layer_table(1024)
    by = {r['name']: r for r in rows}
    blocks = [e.name for e in arch.ENCODERS] + [d.name for d in arch.DECODERS]
    seq, it = [], iter(convs)
    for blk in blocks:
        v = next(it)
        c1, c2 = by[blk + '.conv1'], by[blk + '.conv2']
        sc = by[blk + '.shortcut']['macs'] if blk + '.shortcut' in by else 0
        if blk + '.up' in by:
            if 'dec6u_fused' in v['name']:  # bf16, round 4: the transposed conv runs inside the fused decoder kernel
                seq.append((blk + ' (up + conv1 + conv2, one kernel)', c1['macs'] + c2['macs'] + by[blk + '.up']['macs'], sc, '3x3', c1['h'], v))
                continue
            seq.append((blk + '.up', by[blk + '.up']['macs'], 0, 'tconv', by[blk + '.up']['h'], v))
            v = next(it)
        if 'fused' in v['name']:
            seq.append((blk + ' (conv1+conv2, one kernel)', c1['macs'] + c2['macs'], sc, '3x3', c1['h'], v))
        else:
            seq.append((blk + '.conv1', c1['macs'], 0, '3x3', c1['h'], v))
            seq.append((blk + '.conv2' + ('+sc' if sc else ''), c2['macs'], sc, '3x3', c2['h'], next(it)))
    rest = list(it)
    assert not rest, ('unmatched launches', len(rest))
    return seq


if __name__ == '__main__':
    mode = sys.argv[2] if len(sys.argv) > 2 else 'f32'
    seq = launch_sequence(last_step(load(sys.argv[1])))
    B = 16
    peak = 157.3 if mode == 'f32' else 2500.0
    print(f'| launch | kernel | ms | algorithmic TFLOP/s | executed TFLOP/s | executed / {peak:g} peak | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES) | clock GHz |')
    print('|---|---|---|---|---|---|---|---|')
    tot_ms = tot_alg = tot_exe = 0.0
    cls = {'3x3': 0.0, 'tconv': 0.0}
    for name, macs, sc, kind, h, v in seq:
        dt = (v['t1'] - v['t0']) / 1e9
        alg = 2.0 * B * (macs + sc)
        wino = mode == 'f32' and kind == '3x3' and h % 2 == 0 and 'wino' in v['name']
        wfac = 0.25 if (wino and 'wino4_kernel' in v['name']) else (4.0 / 9.0 if wino else 1.0)  # F(4x4,3x3) / F(2x2,3x3) / direct
        exe = 2.0 * B * (macs * wfac + sc) * (3.0 if mode == 'bf16x3' else 1.0)
        gui = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024.0) if gui else float('nan')
        clk = gui / dt / 1e9 if dt > 0 else float('nan')
        tot_ms += dt * 1e3; tot_alg += alg; tot_exe += exe; cls[kind] += dt * 1e3
        kn = re.sub(r'\(anonymous namespace\)::|^void ', '', v['name']).split('(')[0]
        print(f'| {name} | `{kn}` | {dt*1e3:.3f} | {alg/dt/1e12:.1f} | {exe/dt/1e12:.1f} | {exe/dt/1e12/peak:.3f} | {busy:.3f} | {clk:.2f} |')
    print(f'| **all {len(seq)} launches** | | {tot_ms:.3f} | {tot_alg/tot_ms/1e9:.1f} | {tot_exe/tot_ms/1e9:.1f} | {tot_exe/tot_ms/1e9/peak:.3f} | | |')
    print(f'\nconv3x3 class {cls["3x3"]:.3f} ms, transposed convs {cls["tconv"]:.3f} ms (under the counters; bench.py\'s `conv_ms` is the un-profiled figure)')
