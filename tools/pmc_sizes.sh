#!/bin/bash
# rocprofv3 passes over one tools/bench_sizes.py mode (one counter group per run; program directly after "--").
#   usage: tools/pmc_sizes.sh <tag> <bench_sizes mode>      -> gpurun_out/pmc_<tag>/{kt,fetch,write,sq1,sq2}, summary.txt
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1; MODE=$2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
B="python3 tools/bench_sizes.py $MODE"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1 &&
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $B > $OUT/fetch.log 2>&1 &&
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- $B > $OUT/write.log 2>&1 &&
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq1 -o sq1 -- $B > $OUT/sq1.log 2>&1 &&
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d $OUT/sq2 -o sq2 -- $B > $OUT/sq2.log 2>&1
python3 tools/pmc_by_kernel.py $OUT | tee $OUT/summary.txt
