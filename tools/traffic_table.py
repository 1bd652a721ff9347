#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the SHIPPED pipeline from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`bench.py --steps 1 --warmup 1 --dtype D --modes none`: the conv launches of the last step in lass_separate's order, with
the gfx950 unit corrections of tools/traffic_summary.py (both counters in units of 1024 B, FETCH_SIZE reporting half of the
bytes read; fetch_calibration.json).  Usage: traffic_table.py fetch.csv write.csv [f32|bf16] > t.md"""
import sys

sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from pmc_table import last_step, launch_sequence, load

FETCH_B, WRITE_B = 2048.0, 1024.0  # bytes per counter unit (fetch_calibration.json: measured on gfx950)

if __name__ == '__main__':
    mode = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    fseq = launch_sequence(last_step(load(sys.argv[1])))
    wseq = launch_sequence(last_step(load(sys.argv[2])))
    print('| launch | ms | fetch GB | write GB | (fetch + write) / ms, TB/s |')
    print('|---|---|---|---|---|')
    tf = tw = tms = 0.0
    for (name, _, _, _, _, f), (_, _, _, _, _, w) in zip(fseq, wseq):
        ms = (f['t1'] - f['t0']) / 1e6
        fb, wb = f.get('FETCH_SIZE', 0.0) * FETCH_B, w.get('WRITE_SIZE', 0.0) * WRITE_B
        tf += fb; tw += wb; tms += ms
        print(f'| {name} | {ms:.3f} | {fb / 1e9:.3f} | {wb / 1e9:.3f} | {(fb + wb) / ms / 1e9:.2f} |')
    print(f'| **all {len(fseq)} launches** | {tms:.3f} | {tf / 1e9:.3f} | {tw / 1e9:.3f} | {(tf + tw) / tms / 1e9:.2f} |')
