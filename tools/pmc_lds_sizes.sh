#!/bin/bash
# LDS bank conflicts per kernel instantiation over the size sweep; usage: tools/pmc_lds_sizes.sh <tag>
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $OUT/s -o s -- python3 tools/bench_sizes.py > $OUT/s.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print("%-50s LDS instrs %.3g  array cycles/instr %.2f  conflict share %.2f" % (
        k, c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_IDX_ACTIVE", 0) / max(c.get("SQ_INSTS_LDS", 1), 1),
        c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
