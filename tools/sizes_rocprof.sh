#!/bin/bash
# rocprofv3 --kernel-trace --stats over the size sweep: per-instantiation kernel durations beside profiles/<tag>_sizes.jsonl
# usage: tools/sizes_rocprof.sh <tag>   -> gpurun_out/<tag>_sizes_kernel_stats.csv
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
OUT=gpurun_out/prof_$1_sizes
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o s -- python3 tools/bench_sizes.py > $OUT/s.log 2>&1
python3 - $OUT gpurun_out/$1_sizes_kernel_stats.csv <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
keep = [rows[0]] + [[c if len(c) <= 100 else c[:97] + "..." for c in r] for r in rows[1:] if "stft_kernel" in r[0]]
csv.writer(open(sys.argv[2], "w", newline="")).writerows(keep)
print(open(sys.argv[2]).read())
PY
grep fft $OUT/s.log > gpurun_out/$1_sizes_under_rocprof.jsonl
