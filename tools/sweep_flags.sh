#!/bin/bash
# usage: tools/sweep_flags.sh "<flags1>|<flags2>|..." <bench_sizes mode>
IFS='|' read -ra SETS <<< "$1"
for fl in "${SETS[@]}"; do
  echo "== flags: $fl"
  SG_EXTRA_HIPCC_FLAGS="$fl" python -m spectrogram_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  python tools/bench_sizes.py $2 2>&1 | grep fft
done
SG_EXTRA_HIPCC_FLAGS="" python -m spectrogram_amd.build --force > /dev/null
