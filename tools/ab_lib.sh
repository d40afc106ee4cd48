#!/bin/bash
# same-box A/B of two builds of libspectro.so: the in-tree one against gpurun_stage/<name>.so (SPECTRO_LIB override)
#   usage: tools/ab_lib.sh <other.so> <bench_sizes mode> [<mode> ...]
OTHER=$1; shift
for rep in 1 2; do
  for m in "$@"; do
    echo "== in-tree $m"; python3 tools/bench_sizes.py $m 2>&1 | grep '^{'
    echo "== $OTHER $m"; SPECTRO_LIB=$OTHER python3 tools/bench_sizes.py $m 2>&1 | grep '^{'
  done
done
