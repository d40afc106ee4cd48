#!/bin/bash
# remaining-work priority (SG_PRIO_REM) x run length on the bench workload
for fl in "-DSG_PRIO_REM=0" "-DSG_PRIO_REM=1"; do
  SG_EXTRA_HIPCC_FLAGS="$fl" python -m spectrogram_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; continue; }
  echo "== $fl"
  python tools/runlen_sweep.py ${1:-0,28}
done
SG_EXTRA_HIPCC_FLAGS="" python -m spectrogram_amd.build --force > /dev/null
