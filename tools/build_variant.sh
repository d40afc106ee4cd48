#!/bin/bash
# An alternative build of libspectro.so for same-box A/B runs (tools/lib_ab.py, SPECTRO_LIB): the kernel units of the
# listed sizes are recompiled with extra flags, every other object is the in-tree one.
#   usage: tools/build_variant.sh <name> "<extra hipcc flags>" [log2n ...]     -> gpurun_stage/libspectro_<name>.so
set -e
NAME=$1; FLAGS=$2; shift 2
SIZES=${@:-10 11}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/spectrogram_amd/lib/obj; TMP=/tmp/sgvar_$NAME; mkdir -p $TMP $ROOT/gpurun_stage
BASE="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$ROOT/include -fno-gpu-rdc -mllvm -structurizecfg-skip-uniform-regions"
rm -f $TMP/*.o $TMP/failed
for l in $SIZES; do
  (/opt/rocm/bin/hipcc $BASE $FLAGS -DSG_LOG2N=$l -c $ROOT/spectrogram_amd/csrc/stft_inst.hip -o $TMP/stft_n$l.o || touch $TMP/failed) &
done
# SG_VARIANT_API=1: the host side sees the flags too (diagnostic builds whose launcher differs: SG_STAMP, SG_STALL)
if [ -n "$SG_VARIANT_API" ]; then
  (/opt/rocm/bin/hipcc $BASE $FLAGS -c $ROOT/spectrogram_amd/csrc/spectro_api.hip -o $TMP/spectro_api.o || touch $TMP/failed) &
fi
wait
if [ -f $TMP/failed ]; then echo "build_variant: a unit failed to compile" >&2; exit 1; fi
OBJS=""
for f in $OBJ/*.o; do b=$(basename $f); if [ -f $TMP/$b ]; then OBJS="$OBJS $TMP/$b"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/gpurun_stage/libspectro_$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built $ROOT/gpurun_stage/libspectro_$NAME.so
