"""Summarise rocprofv3 --pmc output: mean counter value per launch, per kernel.

  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -d gpurun_out/pmc --output-format csv -- python3 tools/kernel_times.py
  python tools/pmc_summary.py gpurun_out/pmc [name-filter] > gpurun_out/pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            # one row per (dispatch, counter); dimensions (XCC/SE/...) are summed by rocprofv3 unless split on request
            per_dispatch = collections.defaultdict(float)
            names = {}
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (did, cname), v in per_dispatch.items():
                acc[names[did]][cname].append(v)
    out = {}
    for k, cs in acc.items():
        if flt and flt not in k:
            continue
        out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "median": sorted(v)[len(v) // 2]} for c, v in cs.items()}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
