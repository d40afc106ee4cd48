#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the tracked profiles/ directory.

    python tools/make_profile.py <round-tag> <kernel-trace-dir> <fetch-dir> <write-dir> [<sq-dir>]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim rows),
profiles/<tag>_pmc.txt (mean counter values per dispatch of the STFT kernel) and
profiles/traffic.json (HBM bytes per launch, what bench.py reports as roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE are in KiB and
are collected in separate --pmc passes; on gfx950 FETCH_SIZE counts exactly half of the bytes of
a coalesced streaming read (128-B requests tallied at 64 B), so the read side is doubled;
WRITE_SIZE is exact for streaming stores.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDLE = "stft_kernel"


def counters(d):
    acc = collections.defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if NEEDLE in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"], "grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"],
                        "vgpr_field": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"]}
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}, meta


def main(tag, kt, fetch, write, *extra):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows = []
    for f in glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows[:12])
    avg_ns = None
    for r in rows[1:]:
        if NEEDLE in r[0]:
            avg_ns = float(r[3])
    lines = []
    traffic = {}
    for d in (fetch, write) + tuple(extra):
        c, meta = counters(d)
        lines.append("# %s  %s" % (os.path.basename(d.rstrip("/")), json.dumps(meta)))
        for k in sorted(c):
            lines.append("%-28s mean=%.6g  (n=%d dispatches)" % (k, c[k][0], c[k][1]))
        if "FETCH_SIZE" in c:
            traffic["fetch_size_kib_raw"] = c["FETCH_SIZE"][0]
        if "WRITE_SIZE" in c:
            traffic["write_size_kib_raw"] = c["WRITE_SIZE"][0]
    if "fetch_size_kib_raw" in traffic and "write_size_kib_raw" in traffic:
        rd = traffic["fetch_size_kib_raw"] * 1024 * 2  # gfx950: FETCH_SIZE reads 1/2 for streaming reads
        wr = traffic["write_size_kib_raw"] * 1024
        traffic.update(read_bytes=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
                       correction="FETCH_SIZE KiB x1024 x2 (gfx950 half-count) + WRITE_SIZE KiB x1024",
                       algorithmic_bytes_per_launch=112497 * 3072, kernel_avg_ns_rocprof=avg_ns)
        lines.append("HBM bytes per launch = %.4g read + %.4g write = %.4g (algorithmic %d)" %
                     (rd, wr, rd + wr, 112497 * 3072))
    open(os.path.join(out, tag + "_pmc.txt"), "w").write("\n".join(lines) + "\n")
    tj = os.path.join(out, "traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    allt["power"] = traffic
    allt["_source"] = tag
    json.dump(allt, open(tj, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:])
