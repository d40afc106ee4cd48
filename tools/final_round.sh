#!/bin/bash
# The round's record run on one GPU box: parity suite, profile set, bench lines, size sweep.
# usage: tools/final_round.sh <tag>       (outputs under gpurun_out/; condense at home with tools/make_profile.py)
set -o pipefail
TAG=$1
mkdir -p gpurun_out
./tools/xlanetest > gpurun_out/xlanetest_$TAG.txt 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_$TAG.txt 2>&1 || { tail -n 20 gpurun_out/tests_$TAG.txt; exit 1; }
tail -n 1 gpurun_out/tests_$TAG.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n1_driver.json 2> gpurun_out/bench_$TAG.err && echo bench20 ok
python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_n1.json 2>> gpurun_out/bench_$TAG.err && echo bench200 ok
python bench.py --workload config5 > gpurun_out/${TAG}_bench_config5_n1.json 2>> gpurun_out/bench_$TAG.err && echo config5 ok
(python tools/bench_sizes.py && python tools/bench_sizes.py k10 && python tools/bench_sizes.py u8 && python tools/bench_sizes.py lf && python tools/bench_sizes.py generic && python tools/bench_sizes.py ref) 2>/dev/null | grep fft > gpurun_out/${TAG}_sizes.jsonl && echo sizes ok
tools/profile_round.sh $TAG > gpurun_out/profile_$TAG.log 2>&1 && echo profiles ok
tools/pmc_lds_sizes.sh ${TAG}lds > gpurun_out/${TAG}_lds_conflicts.txt 2>&1 && echo lds ok
