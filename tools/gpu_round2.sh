#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/valubench tools/valubench.hip 2>/dev/null
./tools/valubench 2>&1 | tee gpurun_out/valubench_r02.log
IFS='|' read -ra SETS <<< "$1"; shift
for fl in "${SETS[@]}"; do
  echo "== flags: $fl"
  SG_EXTRA_HIPCC_FLAGS="$fl" python -m spectrogram_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  for mode in "$@"; do
    timeout -k 10 300 python tools/bench_sizes.py $mode 2>&1 | grep -E "fft|rror" || true
  done
done
SG_EXTRA_HIPCC_FLAGS="" python -m spectrogram_amd.build --force > /dev/null
