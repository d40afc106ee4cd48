#!/bin/bash
# the randomized-layout parity sweep (tests/test_parity_gpu.py::test_randomized_layouts and ::test_randomized_formats_strides_and_epilogues) from a handful of other seeds
# usage: tools/fuzz_layouts.sh "<seed> <seed> ..." [cases per seed]
set -o pipefail
for s in $1; do
  SG_FUZZ_SEED=$s SG_FUZZ_CASES=${2:-96} timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -q -x -k randomized 2>&1 | tail -n 3 | sed "s/^/seed $s: /" || exit 1
done
