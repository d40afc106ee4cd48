#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output for one kernel-name substring: mean counter values per
dispatch (counter_collection.csv) and duration stats (kernel_trace.csv / kernel_stats.csv)."""
import collections
import csv
import glob
import sys


def main(root, needle="stft_kernel"):
    for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(list)
        meta = None
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Kernel_Name"][:60], "vgpr", r["VGPR_Count"], "sgpr", r["SGPR_Count"], "lds",
                        r["LDS_Block_Size"], "grid", r["Grid_Size"], "wg", r["Workgroup_Size"])
        print(f, meta)
        for k in sorted(acc):
            v = acc[k]
            print("  %-34s n=%-4d mean=%.6g min=%.6g max=%.6g" % (k, len(v), sum(v) / len(v), min(v), max(v)))
    for f in sorted(glob.glob(root + "/**/*_kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if needle in r["Name"]:
                print(f, {k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev")})


if __name__ == "__main__":
    main(*sys.argv[1:])
