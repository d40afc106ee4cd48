#!/bin/bash
# Round 6: where do the waves of stft_kernel<11,2,0> (N = 2048, hop 512: config 3, the reference's fftSize) spend the time
# they are not issuing?  rocprofv3 passes over `tools/bench_sizes.py n2048` (one counter group per run, the program directly
# after "--"): kernel stats, the SQ wait counters, the in-flight-level counters behind VmemLatency / LdsLatency, and -- when
# the pool allows it -- stochastic PC sampling, whose samples carry the reason a wave did not issue.
#   usage: tools/n2048_stalls.sh <tag> [mode]    -> gpurun_out/stalls_<tag>/
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1; MODE=${2:-n2048}
OUT=gpurun_out/stalls_$TAG
mkdir -p $OUT
B="python3 tools/bench_sizes.py $MODE"
$B > $OUT/plain.log 2>&1 &&
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1 &&
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq1 -o sq1 -- $B > $OUT/sq1.log 2>&1 &&
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/sq2 -o sq2 -- $B > $OUT/sq2.log 2>&1 &&
rocprofv3 --output-format csv --pmc VmemLatency -d $OUT/vl -o vl -- $B > $OUT/vl.log 2>&1 &&
rocprofv3 --output-format csv --pmc LdsLatency -d $OUT/ll -o ll -- $B > $OUT/ll.log 2>&1 &&
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH -d $OUT/sq3 -o sq3 -- $B > $OUT/sq3.log 2>&1
echo "pmc passes rc=$?" | tee $OUT/rc.txt
python3 tools/pmc_by_kernel.py $OUT > $OUT/summary.txt 2>&1
