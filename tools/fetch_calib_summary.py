#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/fetch_calib -> profiles/rNN/fetch_calibration.json:
counter bytes (KiB units x 1024) / bytes really moved, per access shape (second repetition of each kernel)."""
import collections
import csv
import json
import sys

MOVED = float(4 << 28)


def per_kernel(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0], "v": 0.0})
        e["v"] += float(r["Counter_Value"])
    out = {}
    for e in d.values():  # later dispatches overwrite earlier: keeps the second repetition
        out[e["name"]] = e["v"] * 1024.0
    return out


def main(fetch_csv, write_csv, out_json):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    res = {"moved_bytes_per_kernel": MOVED,
           "fetch_ratio": {k: v / MOVED for k, v in f.items() if k.startswith("rd_")},
           "write_ratio": {k: v / MOVED for k, v in w.items() if k.startswith("wr_")},
           "fetch_of_write_kernels": {k: v / MOVED for k, v in f.items() if k.startswith("wr_")},
           "note": "ratio = counter x 1024 B / bytes moved; the traffic summary divides each kernel's counters by the "
                   "ratio of its own access shape"}
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
