#!/bin/bash
# deeper stall counters (one group per pass); usage: tools/pmc_round2.sh <tag> [bench args]
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-streaming --no-power --steps 20 --warmup 2"
rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES -d $OUT/a -o a -- $B "$@" > $OUT/a.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT/b -o b -- $B "$@" > $OUT/b.log 2>&1
# (TA_* / TCP_* / TD_* groups are not collected: a TA_* pass aborted and hung rocprofv3 on this pool, tools/README.md)
python3 tools/pmc_summary.py $OUT 2>&1 | grep -v "kernel_stats" | tee $OUT/summary.txt
