#!/bin/bash
# LDS counters of the bench kernel (one pass); usage: tools/pmc_lds.sh <tag> [bench args]
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-streaming --no-power --steps 20 --warmup 2"
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $OUT/l -o l -- $B "$@" > $OUT/l.log 2>&1
python3 tools/pmc_summary.py $OUT 2>&1 | grep -v "kernel_stats" | tee $OUT/summary.txt
