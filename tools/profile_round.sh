#!/bin/bash
# The profile set behind bench.py's roofline object (run on the GPU box; outputs under gpurun_out/prof_<tag>).
# One rocprofv3 run per counter group: FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2, so they cannot share a
# pass, and asking for more counters than a block has slots aborts inside the first GPU call with
# rocprofiler_create_counter_config error 38 ("Request exceeds the capabilities of the hardware to collect").
# The program sits directly after "--" (no env / bash / taskset hop) and --pmc is never combined with trace domains.
#   usage: tools/profile_round.sh <tag>     then, back home:  python tools/make_profile.py <tag> gpurun_out/prof_<tag>
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python3 bench.py --full-line --no-cpu-baseline --no-streaming --no-power --no-extra-configs"
BX="python3 bench.py --full-line --no-cpu-baseline --no-streaming --no-power"   # with the configs[2] / configs[4] objects of the default line
# the driver's command line (--steps 20 --warmup 5): the kernel average the bench line must agree with ...
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 20 --warmup 5 > $OUT/kt.log 2>&1
# ... and a 220-launch run for the launch-by-launch trace (clock transient under sustained load)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt200 -o kt200 -- $B --steps 200 --warmup 20 > $OUT/kt200.log 2>&1
# ... and the same 220 launches from a cold start (no settle phase): the transient itself
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/ktcold -o ktcold -- $B --steps 200 --warmup 20 --settle-ms 0 > $OUT/ktcold.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $B --steps 20 --warmup 2 > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- $B --steps 20 --warmup 2 > $OUT/write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq1 -o sq1 -- $B --steps 20 --warmup 2 > $OUT/sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $OUT/sq2 -o sq2 -- $B --steps 20 --warmup 2 > $OUT/sq2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/sq3 -o sq3 -- $B --steps 20 --warmup 2 > $OUT/sq3.log 2>&1
# the byte-output variant's HBM traffic (traffic.json "db_u8": --out-kind db_u8 lines used to carry traffic: null)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch_u8 -o fetch -- $B --steps 20 --warmup 2 --out-kind db_u8 > $OUT/fetch_u8.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write_u8 -o write -- $B --steps 20 --warmup 2 --out-kind db_u8 > $OUT/write_u8.log 2>&1
# the default line's extra objects under the kernel trace: configs[2] at full size (stft_kernel<11,2,0>, 12 launches of
# 5,752,832 frames) and configs[4] chunks (stft_kernel<10,2,0> on 7,813-utterance chunks, mixed into that kernel's row)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/ktx -o ktx -- $BX --steps 20 --warmup 5 > $OUT/ktx.log 2>&1
# ... and their HBM traffic: configs[2]'s stft_kernel<11,2,0> launch and configs[4]'s 7,813-utterance chunk launch (traffic.json
# "batched" / "config5"; the chunk launches are told from the config-2 launches of the same kernel by their grid)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch_x -o fetch -- $BX --steps 4 --warmup 1 --settle-ms 0 > $OUT/fetch_x.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write_x -o write -- $BX --steps 4 --warmup 1 --settle-ms 0 > $OUT/write_x.log 2>&1
grep -h '"metric"' $OUT/kt.log | head -1 > $OUT/bench_under_rocprof.json
grep -h '"metric"' $OUT/ktx.log | head -1 > $OUT/bench_extra_under_rocprof.json
ls $OUT
