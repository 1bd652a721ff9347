#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of `bench.py --steps 1 --warmup 1 --modes none`)
into profiles/<round>/conv_traffic_<dtype>.json: HBM-side bytes per launch of the dominant kernel class (conv3x3_mfma:
the 26 Winograd / bf16 3x3 launches of a step), last step only.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in units of 1024 B; FETCH_SIZE reports exactly 1/2 of
the bytes read.  tools/fetch_calib.hip measured that factor for every access shape these kernels use (4 / 8 / 16 B per
lane buffer loads, 128-byte row segments, LDS-DMA): 0.5000 in all cases, WRITE_SIZE 1.000 (profiles/r02/
fetch_calibration.json) - so ONE corrected number is reported: traffic = FETCH_SIZE / 0.5 + WRITE_SIZE.
Usage: traffic_summary.py fetch.csv write.csv out.json dtype [fetch_calibration.json]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_dispatch(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        e = d.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'v': 0.0})
        e['v'] += float(r['Counter_Value'])
    return list(d.values())


def is_conv3x3(name):
    if 'wino_kernel' in name or 'wino32_kernel' in name or 'wino4_kernel' in name or 'fused_bf16_kernel' in name:
        return True
    if 'conv_bf16_kernel<9' in name or 'conv_bf16_kernelILi9E' in name:
        return True
    return ('conv_kernel' in name) and 'relayout' not in name and ('<9,' in name or 'ILi9E' in name)


def last_step(entries):
    """The dispatches behind the last stft2_kernel (the last step of the run)."""
    last = max(i for i, e in enumerate(entries) if 'stft2_kernel' in e['name'] and 'istft2' not in e['name'])
    return entries[last:]


def main(fetch_csv, write_csv, out_json, dtype, calib_json=None):
    ratio_f, ratio_w = 0.5, 1.0
    calib = None
    if calib_json:
        calib = json.load(open(calib_json))
        rf = list(calib['fetch_ratio'].values())
        rw = list(calib['write_ratio'].values())
        assert max(rf) - min(rf) < 1e-3 and max(rw) - min(rw) < 1e-3, "access shapes disagree: per-kernel factors needed"
        ratio_f, ratio_w = sum(rf) / len(rf), sum(rw) / len(rw)
    f = [e for e in last_step(per_dispatch(fetch_csv, 'FETCH_SIZE')) if is_conv3x3(e['name'])]
    w = [e for e in last_step(per_dispatch(write_csv, 'WRITE_SIZE')) if is_conv3x3(e['name'])]
    launches_per_step = len(f)  # 26 (f32, bf16x3) or 24 (bf16: encoder_block1 and decoder_block6 are one kernel each)
    assert launches_per_step in (24, 25, 26) and len(w) == launches_per_step, (len(f), len(w))
    fetch = sum(e['v'] for e in f) * 1024.0 / ratio_f
    write = sum(e['v'] for e in w) * 1024.0 / ratio_w
    res = {
        "kernel_class": "conv3x3_mfma", "dtype": dtype, "launches": launches_per_step,
        "fetch_bytes_per_launch": fetch / launches_per_step,
        "write_bytes_per_launch": write / launches_per_step,
        "traffic_bytes_per_launch": (fetch + write) / launches_per_step,
        "fetch_calibration": {"fetch_ratio": ratio_f, "write_ratio": ratio_w,
                              "source": calib_json or "MI355X_MICROARCH.md (0.5 / 1.0)"},
        "workload": "bench.py --steps 1 --warmup 1 (B=16, 10 s clips), last step",
        "source_hash": __import__("__graft_entry__")._src_hash(),  # kernel sources this was measured at (bench.py checks it)
        "note": "L2 memory-side request bytes (Infinity-Cache hits included), corrected by the measured counter ratio",
    }
    json.dump(res, open(out_json, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main(*sys.argv[1:6])
