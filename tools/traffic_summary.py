#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of `bench.py --steps 1 --warmup 1`) into
profiles/<round>/conv_traffic.json: HBM bytes per launch of the dominant kernel class (conv3x3_mfma), last step only.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B; FETCH_SIZE reads
exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream - our conv kernels read 4 B/lane (activations) and
16 B/lane (weights), an access mix the guide calls uncalibrated, so both the raw and the x2-corrected figure are kept.
"""
import csv, json, sys, collections

def per_dispatch(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 'v': 0.0})
        e['v'] += float(r['Counter_Value'])
    return list(d.values())

def main(fetch_csv, write_csv, out_json, launches_per_step=26):
    f = [e for e in per_dispatch(fetch_csv, 'FETCH_SIZE') if (('conv_kernel' in e['name'] and '<1,' not in e['name'] and 'ILi1E' not in e['name']) or 'wino_kernel' in e['name'])]
    w = [e for e in per_dispatch(write_csv, 'WRITE_SIZE') if (('conv_kernel' in e['name'] and '<1,' not in e['name'] and 'ILi1E' not in e['name']) or 'wino_kernel' in e['name'])]
    f, w = f[-launches_per_step:], w[-launches_per_step:]
    fetch = sum(e['v'] for e in f) * 1024.0
    write = sum(e['v'] for e in w) * 1024.0
    res = {
        "kernel_class": "conv3x3_mfma", "launches": launches_per_step,
        "fetch_bytes_raw_per_launch": fetch / launches_per_step,
        "fetch_bytes_x2_per_launch": 2 * fetch / launches_per_step,
        "write_bytes_per_launch": write / launches_per_step,
        "traffic_bytes_per_launch": (2 * fetch + write) / launches_per_step,
        "traffic_bytes_per_launch_raw": (fetch + write) / launches_per_step,
        "note": "FETCH_SIZE x2 per the gfx950 correction for coalesced streams; raw kept because 4-B/lane loads are uncalibrated",
    }
    json.dump(res, open(out_json, 'w'), indent=1)
    print(json.dumps(res))

if __name__ == '__main__':
    main(*sys.argv[1:4])
