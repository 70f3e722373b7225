#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter.

    python scripts/pmc_summary.py OUT.json DIR [DIR ...]      (DIR = a rocprofv3 -d output directory, one per pass)
"""
import csv
import glob
import json
import os
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if not k.startswith("void rtd::") and "rtd::" not in k:
                        continue
                    e = acc.setdefault(k, {"dispatches": set(), "counters": {}, "lds": row["LDS_Block_Size"], "scratch": row["Scratch_Size"],
                                           "vgpr": row["VGPR_Count"], "wg": row["Workgroup_Size"], "ns": {}})
                    e["counters"][row["Counter_Name"]] = e["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    e["dispatches"].add((d, row["Dispatch_Id"]))
                    e["ns"][(d, row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    res = {}
    for k, e in acc.items():
        per_dir = {}
        for (d, _), ns in e["ns"].items():
            per_dir[d] = per_dir.get(d, 0) + ns
        res[k] = {"workgroup": e["wg"], "lds_bytes": e["lds"], "scratch_bytes": e["scratch"], "vgprs": e["vgpr"],
                  "kernel_seconds_per_pass": {os.path.basename(d.rstrip("/")): v * 1e-9 for d, v in per_dir.items()},
                  "counters": e["counters"]}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
