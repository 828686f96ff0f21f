#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (pmc_counter_collection.csv) per kernel: mean counter value per dispatch.
usage: summarize_pmc.py <dir-with-pass-subdirs> [kernel-substring ...]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main(root, filt):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in sorted(glob.glob(root + "/*/pmc_counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"][:40]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in sorted(acc, key=lambda n: -sum(dur[n])):
        if filt and not any(x in name for x in filt):
            continue
        print(f"## {name}: {len(dur[name])} profiled dispatches, mean duration {sum(dur[name])/len(dur[name]):.1f} us")
        for c, v in sorted(acc[name].items()):
            print(f"  {c:28s} mean/dispatch {sum(v)/len(v):16.1f}")
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
