#!/bin/bash
# frames-per-run sweep for the LDS-ring load variants (tools/bench_sizes.py r03gen)
for rl in 16 24 32 48 64 96; do echo "SG_RUN_LEN=$rl"; SG_RUN_LEN=$rl python3 tools/bench_sizes.py r03gen 2>&1 | grep -E '"hop": (300|735)'; done
