#!/usr/bin/env python3
"""Turn rocprofv3 counter-collection CSVs (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py`) into
profiles/<round>_pmc_traffic.json, which bench.py reads to fill roofline.traffic.

  python tools/pmc_traffic.py <round> <config> <fetch_counter_collection.csv> <write_counter_collection.csv>

Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of
1024 bytes per count as printed; FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x on gfx950 - the replay /
sequence kernels read 4-16 B per lane in gathers, an access width the guide lists as uncalibrated, so the raw value is
kept and flagged "uncalibrated"; WRITE_SIZE is exact for dword-per-lane stores. Values are averages per launch.
"""
import csv
import collections
import json
import sys


def rows_of(path):
    """rocprofv3 counter-collection output: CSV (--output-format csv) or the default rocpd sqlite database."""
    if path.endswith(".db"):
        import sqlite3
        db = sqlite3.connect(path)
        for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            yield {"Kernel_Name": name, "Counter_Name": counter, "Counter_Value": value}
    else:
        for r in csv.DictReader(open(path)):
            yield r


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in rows_of(path):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "mp::" not in name:
            continue
        short = name.split("mp::")[1].split("<")[0].split("(")[0]
        acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    rnd, config, fpath, wpath = sys.argv[1:5]
    fetch, nf = per_kernel(fpath, "FETCH_SIZE")
    write, _ = per_kernel(wpath, "WRITE_SIZE")
    out = {"round": rnd, "config": config, "unit": "bytes per launch", "launches_averaged": nf,
           "note": "FETCH_SIZE / WRITE_SIZE x 1024; gather-width reads are uncalibrated on gfx950 (guide), no 2x applied",
           "kernels": {k: {"fetch_bytes": fetch[k] * 1024.0, "write_bytes": write.get(k, 0.0) * 1024.0,
                           "hbm_bytes": (fetch[k] + write.get(k, 0.0)) * 1024.0} for k in fetch}}
    path = "profiles/%s_pmc_traffic.json" % rnd
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, json.dumps(out["kernels"]))


if __name__ == "__main__":
    main()
