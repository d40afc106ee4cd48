#!/bin/bash
# Round 6: PC sampling of one bench_sizes.py mode (rocprofv3 beta feature).  Stochastic samples carry the reason a wave
# did not issue (barrier / waitcnt / arbiter ...); host-trap samples only the PC, which is enough to tell s_barrier from
# s_waitcnt vmcnt / lgkmcnt.  Never combined with --pmc or trace domains; the program sits directly after "--".
#   usage: tools/pcsamp.sh <tag> <mode> <stochastic|host_trap>   -> gpurun_out/pcsamp_<tag>/
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1; MODE=${2:-n2048}; METHOD=${3:-stochastic}
OUT=gpurun_out/pcsamp_$TAG
mkdir -p $OUT
if [ "$METHOD" = stochastic ]; then UNIT="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else UNIT="--pc-sampling-unit time --pc-sampling-interval 100"; fi
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD $UNIT --output-format csv -d $OUT/$METHOD -o pcs -- python3 tools/bench_sizes.py $MODE > $OUT/$METHOD.log 2>&1
echo "pcsamp $METHOD rc=$?" | tee -a $OUT/rc.txt
ls -la $OUT/$METHOD 2>/dev/null | head
