#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the tracked profiles/ directory.

    python tools/make_profile.py <round-tag> <prof-dir>          (prof-dir as written by tools/profile_round.sh:
                                                                   kt/ fetch/ write/ sq1/ sq2/ sq3/)
    python tools/make_profile.py <round-tag> <kernel-trace-dir> <fetch-dir> <write-dir> [<sq-dir>]   (round-1 layout)

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
NEEDLE = "stft_kernel<10, 2, 0>"   # the timed kernel; bench.py's clock-settle phase runs the <10, 2, 2> variant
SETTLE = "stft_kernel<10, 2, 2>"


def counters(d, needle=None, grid=None, not_grid=None):
    needle = needle or NEEDLE
    acc = collections.defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid) and (not_grid is None or r["Grid_Size"] != not_grid):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"], "grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"],
                        "vgpr_field": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"]}
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}, meta


def launch_trace(kt, path):
    """Per-launch durations of the STFT kernels in start order (kernel trace): the clock transient of a back-to-back
    run (docs/EXPERIMENTS.md section 5.1) through bench.py's settle phase (S rows: the dB-output variant) into the warm-up and
    timed launches (T rows)."""
    rows = []
    for f in glob.glob(os.path.join(kt, "**", "*_kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if NEEDLE in r["Kernel_Name"] or SETTLE in r["Kernel_Name"]]
    if not rows:
        return
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    timed = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if NEEDLE in r["Kernel_Name"]]
    with open(path, "w") as fh:
        fh.write("# S = settle phase (dB-output variant), T = warm-up + timed launches of the bench kernel; start (ms since the "
                 "first launch), then up to 10 consecutive kernel durations in us\n")
        if timed:
            fh.write("# T launches: n=%d mean %.1f us, min %.1f, max %.1f\n" % (len(timed), sum(timed) / len(timed), min(timed), max(timed)))
        i = 0
        while i < len(rows):
            kind = "T" if NEEDLE in rows[i]["Kernel_Name"] else "S"
            j = i
            while j < len(rows) and j < i + 10 and (NEEDLE in rows[j]["Kernel_Name"]) == (kind == "T"):
                j += 1
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:j]]
            fh.write("%s %4d  %7.2f ms  %s\n" % (kind, i, (int(rows[i]["Start_Timestamp"]) - t0) / 1e6, " ".join("%5.1f" % x for x in d)))
            i = j


def main(tag, kt, fetch=None, write=None, *extra):
    if fetch is None:  # one directory written by tools/profile_round.sh
        root = kt
        kt, fetch, write = (os.path.join(root, d) for d in ("kt", "fetch", "write"))
        extra = tuple(os.path.join(root, d) for d in ("sq1", "sq2", "sq3") if os.path.isdir(os.path.join(root, d)))
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows = []
    for f in glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
    rows = [[c if len(c) <= 120 else c[:117] + "..." for c in r] for r in rows]  # torch's kernel names run to kilobytes
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows[:12])
    kt_long = os.path.join(os.path.dirname(kt.rstrip("/")), "kt200")
    launch_trace(kt_long if os.path.isdir(kt_long) else kt, os.path.join(out, tag + "_launch_trace.txt"))
    kt_cold = os.path.join(os.path.dirname(kt.rstrip("/")), "ktcold")
    if os.path.isdir(kt_cold):
        launch_trace(kt_cold, os.path.join(out, tag + "_launch_trace_cold.txt"))
    for f in glob.glob(os.path.join(kt_long, "**", "*_kernel_stats.csv"), recursive=True):
        r200 = [[c if len(c) <= 120 else c[:117] + "..." for c in r] for r in csv.reader(open(f))]
        with open(os.path.join(out, tag + "_kernel_stats_200steps.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(r200[:4])
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
    # the byte-output variant (bench.py --out-kind db_u8): stft_kernel<10, 2, 3>, 256 * 4 + 512 * 1 = 1,536 B per frame
    u8 = {}
    root = os.path.dirname(fetch.rstrip("/"))
    fu, wu = os.path.join(root, "fetch_u8"), os.path.join(root, "write_u8")
    if os.path.isdir(fu) and os.path.isdir(wu):
        cf, meta = counters(fu, "stft_kernel<10, 2, 3>")
        cw, _ = counters(wu, "stft_kernel<10, 2, 3>")
        if "FETCH_SIZE" in cf and "WRITE_SIZE" in cw:
            rd, wr = cf["FETCH_SIZE"][0] * 1024 * 2, cw["WRITE_SIZE"][0] * 1024
            u8 = dict(fetch_size_kib_raw=cf["FETCH_SIZE"][0], write_size_kib_raw=cw["WRITE_SIZE"][0], read_bytes=rd, write_bytes=wr,
                      hbm_bytes_per_launch=rd + wr, correction="FETCH_SIZE KiB x1024 x2 (gfx950 half-count) + WRITE_SIZE KiB x1024",
                      algorithmic_bytes_per_launch=112497 * 1536)
            lines.append("# db_u8 variant  %s" % json.dumps(meta))
            lines.append("HBM bytes per launch (stft_kernel<10,2,3>) = %.4g read + %.4g write = %.4g (algorithmic %d)" %
                         (rd, wr, rd + wr, 112497 * 1536))
    # the default line's extra kernels: configs[2] at full size (stft_kernel<11,2,0>, 5,752,832 frames x 6,144 B) and configs[4]'s
    # chunk launch (stft_kernel<10,2,0> on 7,813 utterances x 622 frames x 3,072 B: every launch of that kernel whose grid is not
    # the config-2 launch's)
    extra_t = {}
    fx, wx = os.path.join(root, "fetch_x"), os.path.join(root, "write_x")
    if os.path.isdir(fx) and os.path.isdir(wx):
        c2grid = None
        _, m2 = counters(fetch)
        c2grid = m2.get("grid")
        for key, needle, kw, alg in (("batched", "stft_kernel<11, 2, 0>", {}, 5752832 * 6144),
                                     ("config5", NEEDLE, {"not_grid": c2grid}, 7813 * 622 * 3072)):
            cf, meta = counters(fx, needle, **kw)
            cw, _ = counters(wx, needle, **kw)
            if "FETCH_SIZE" in cf and "WRITE_SIZE" in cw:
                rd, wr = cf["FETCH_SIZE"][0] * 1024 * 2, cw["WRITE_SIZE"][0] * 1024
                extra_t[key] = dict(fetch_size_kib_raw=cf["FETCH_SIZE"][0], write_size_kib_raw=cw["WRITE_SIZE"][0], read_bytes=rd,
                                    write_bytes=wr, hbm_bytes_per_launch=rd + wr, dispatches=[cf["FETCH_SIZE"][1], cw["WRITE_SIZE"][1]],
                                    correction="FETCH_SIZE KiB x1024 x2 (gfx950 half-count) + WRITE_SIZE KiB x1024",
                                    algorithmic_bytes_per_launch=alg, grid=meta.get("grid"))
                lines.append("# %s  %s" % (key, json.dumps(meta)))
                lines.append("HBM bytes per launch (%s) = %.5g read + %.5g write = %.5g (algorithmic %d: %.3fx)" %
                             (key, rd, wr, rd + wr, alg, (rd + wr) / alg))
    # the default line's extra objects (configs[2], configs[4]) under the kernel trace
    ktx = os.path.join(root, "ktx")
    for f in glob.glob(os.path.join(ktx, "**", "*_kernel_stats.csv"), recursive=True):
        rx = [[c if len(c) <= 120 else c[:117] + "..." for c in r] for r in csv.reader(open(f))]
        keep = [rx[0]] + [r for r in rx[1:] if "stft_kernel" in r[0]]
        with open(os.path.join(out, tag + "_kernel_stats_default_line.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(keep)
    open(os.path.join(out, tag + "_pmc.txt"), "w").write("\n".join(lines) + "\n")
    tj = os.path.join(out, "traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    allt["power"] = traffic
    if u8:
        allt["db_u8"] = u8
    allt.update(extra_t)
    allt["_source"] = tag
    json.dump(allt, open(tj, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:])
