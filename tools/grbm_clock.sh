#!/bin/bash
# effective clock of >= 10 ms dispatches: GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back);
# counter pass and kernel-trace pass are separate runs of the same program
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
OUT=gpurun_out/grbm
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 tools/bench_sizes.py long > $OUT/kt.log 2>&1 &&
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE -d $OUT/grbm -o grbm -- python3 tools/bench_sizes.py long > $OUT/grbm.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
def short(n):
    m = re.search(r"stft_kernel<(\d+), *(\d+), *(\d+)>", n); return "stft_kernel<%s,%s,%s>" % m.groups() if m else None
dur = {}
for f in glob.glob("gpurun_out/grbm/kt/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r["Name"]): dur[short(r["Name"])] = float(r["AverageNs"])
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/grbm/grbm/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r["Kernel_Name"]) and r["Counter_Name"] == "GRBM_GUI_ACTIVE": acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    g = sum(acc[k]) / len(acc[k])
    if k in dur: print("%s: GRBM_GUI_ACTIVE %.4g / 8 / %.3f ms = %.0f MHz (n=%d dispatches)" % (k, g, dur[k] / 1e6, g / 8 / (dur[k] / 1e3), len(acc[k])))
PY
