#!/usr/bin/env python3
"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as gfx950 cannot hold both in
one pass) of `bench.py --steps 1 --warmup 0` into per-dispatch HBM traffic.
Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a
coalesced streaming read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (counters are in KB).  Calibration
in this access pattern: k_leaf_hash_lde writes 2^23 x 32 B = 262144 KB (WRITE_SIZE reads exactly that) and
reads cols x 2^23 x 8 B (2 * FETCH_SIZE matches within 0.1 %)."""
import csv
import json
import sys
from collections import OrderedDict


def load(path):
    out = OrderedDict()
    for r in csv.DictReader(open(path)):
        key = int(r["Dispatch_Id"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        v = out.setdefault(key, [name, int(r["Grid_Size"]), 0.0])
        v[2] += float(r["Counter_Value"])
    return out


def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv), load(write_csv)
    rows = []
    for d, (name, grid, fv) in f.items():
        if d in w and w[d][0] == name:
            rows.append({"dispatch": d, "kernel": name, "grid": grid, "FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": w[d][2],
                         "hbm_bytes": (2.0 * fv + w[d][2]) * 1024.0})
    big = [r for r in rows if r["hbm_bytes"] > 5e7]
    json.dump({"correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "dispatches": big}, open(out_json, "w"), indent=1)
    for r in big:
        print("%5d %-34s grid=%-10d fetch=%10.0f KB write=%10.0f KB  hbm=%8.3f GB" % (
            r["dispatch"], r["kernel"][:34], r["grid"], r["FETCH_SIZE_KB"], r["WRITE_SIZE_KB"], r["hbm_bytes"] / 1e9))


if __name__ == "__main__":
    main(*sys.argv[1:4])
