#!/bin/bash
# usage: tools/gpu_round3.sh "<flags>|..." "<extra_lds list>" modes...
set -o pipefail
mkdir -p gpurun_out
IFS='|' read -ra SETS <<< "$1"; shift
XL="$1"; shift
for fl in "${SETS[@]}"; do
  SG_EXTRA_HIPCC_FLAGS="$fl" python -m spectrogram_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  for x in $XL; do
    echo "== flags: $fl extra_lds=$x"
    for mode in "$@"; do
      SG_EXTRA_LDS=$x timeout -k 10 300 python tools/bench_sizes.py $mode 2>&1 | grep -E "fft|rror" || true
    done
  done
done
SG_EXTRA_HIPCC_FLAGS="" python -m spectrogram_amd.build --force > /dev/null
