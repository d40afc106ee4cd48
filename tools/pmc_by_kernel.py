#!/usr/bin/env python3
"""Per-kernel-instantiation means of every counter under a tools/pmc_sizes.sh output directory, with the
gfx950 corrections (FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024: MI355X_MICROARCH.md section HBM)."""
import collections, csv, glob, re, sys

def short(name):
    m = re.search(r"stft_kernel<(\d+), *(\d+), *(\d+)>", name)
    return "stft_kernel<%s,%s,%s>" % m.groups() if m else None

def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = "vgpr %s lds %s grid %s wg %s" % (r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    dur = {}
    for f in sorted(glob.glob(root + "/**/*_kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k: dur[k] = (int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]))
    for k in sorted(acc):
        print("%s  %s" % (k, meta[k]))
        if k in dur:
            print("  duration: calls=%d avg=%.1f us min=%.1f max=%.1f" % (dur[k][0], dur[k][1] / 1e3, dur[k][2] / 1e3, dur[k][3] / 1e3))
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        for n in sorted(c):
            print("  %-24s mean=%.6g (n=%d)" % (n, c[n], len(acc[k][n])))
        if "FETCH_SIZE" in c:
            print("  read bytes  (FETCH_SIZE KiB x1024 x2) = %.4g" % (c["FETCH_SIZE"] * 2048))
        if "WRITE_SIZE" in c:
            print("  write bytes (WRITE_SIZE KiB x1024)    = %.4g" % (c["WRITE_SIZE"] * 1024))
        if "SQ_WAVE_CYCLES" in c:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if n in c: print("  %-24s / SQ_WAVE_CYCLES = %.3f" % (n, c[n] / c["SQ_WAVE_CYCLES"]))
        if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print("  LDS bank-conflict share = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))

if __name__ == "__main__":
    main(sys.argv[1])
