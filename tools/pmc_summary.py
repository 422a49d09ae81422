#!/usr/bin/env python3
"""Summarise a rocprofv3 counter-collection CSV (one --pmc pass of `bench.py`) per kernel: average counter value per launch.

  python tools/pmc_summary.py [--sum] <counter_collection.csv> [<more.csv> ...] > profiles/<round>_<what>.json

--sum: the run launches every kernel several times on inputs of different sizes (config E walks the exome in gene chunks): also
report each counter's total over the run's launches (`<counter>_total`), and keep the library kernels (rocPRIM sort / unique).

Counters are summed over the SEs / XCDs as rocprofv3 reports them (one row per dispatch and counter).  FETCH_SIZE and
WRITE_SIZE are also given in bytes (x 1024); for FETCH_SIZE the gfx950 correction of MI355X_MICROARCH.md (HBM section:
wide coalesced streaming reads are tallied at half their bytes) is reported next to the raw value as `fetch_bytes_x2` -
the true figure lies between the two for kernels that mix 4-16 B gathers with streaming reads.
"""
import collections
import csv
import json
import sys


KEEP_ALL = False


def short_name(name):
    if "mp::" not in name:
        if not KEEP_ALL:
            return None
        return name.split("(")[0].split("<")[0].replace("void ", "")[:80]
    s = name.split("mp::", 1)[1]
    return s.split("(")[0]


def main():
    global KEEP_ALL
    args = sys.argv[1:]
    want_sum = "--sum" in args
    if want_sum:
        args.remove("--sum")
        KEEP_ALL = True
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in args:
        for r in csv.DictReader(open(path)):
            k = short_name(r["Kernel_Name"])
            if k is None:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    out = {}
    for k, cs in acc.items():
        o = dict(meta[k])
        o["launches_averaged"] = max(len(v) for v in cs.values())
        for c, v in cs.items():
            o[c] = sum(v) / len(v)
            if want_sum:
                o[c + "_total"] = sum(v)
        if "FETCH_SIZE" in o:
            o["fetch_bytes_raw"] = o["FETCH_SIZE"] * 1024.0
            o["fetch_bytes_x2"] = o["FETCH_SIZE"] * 2048.0
        if "WRITE_SIZE" in o:
            o["write_bytes"] = o["WRITE_SIZE"] * 1024.0
        if want_sum and "FETCH_SIZE_total" in o:
            o["fetch_bytes_raw_total"] = o["FETCH_SIZE_total"] * 1024.0
        if want_sum and "WRITE_SIZE_total" in o:
            o["write_bytes_total"] = o["WRITE_SIZE_total"] * 1024.0
        out[k] = o
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
