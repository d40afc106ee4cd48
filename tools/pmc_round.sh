#!/bin/bash
# rocprofv3 passes over bench.py (program directly after "--": no env/bash hop; one counter group per pass,
# no trace domains combined with --pmc).  usage: tools/pmc_round.sh [--deep] <tag> [bench args...]
#   --deep: the stall / FIFO / level counters as well (the former pmc_round2.sh)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
DEEP=0; if [ "$1" = "--deep" ]; then DEEP=1; shift; fi
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-streaming --no-power --no-extra-configs "$@" > $OUT/kt.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq1 -o sq1 -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-streaming --no-power --no-extra-configs "$@" > $OUT/sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $OUT/sq2 -o sq2 -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-streaming --no-power --no-extra-configs "$@" > $OUT/sq2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/sq3 -o sq3 -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-streaming --no-power --no-extra-configs "$@" > $OUT/sq3.log 2>&1
if [ $DEEP = 1 ]; then
  B="python3 bench.py --no-cpu-baseline --no-streaming --no-power --no-extra-configs --steps 20 --warmup 2"
  rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES -d $OUT/a -o a -- $B "$@" > $OUT/a.log 2>&1
  rocprofv3 --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT/b -o b -- $B "$@" > $OUT/b.log 2>&1
  # (TA_* / TCP_* / TD_* groups are not collected: a TA_* pass aborted and hung rocprofv3 on this pool, tools/README.md)
fi
python3 tools/pmc_summary.py $OUT 2>&1 | tee $OUT/summary.txt
