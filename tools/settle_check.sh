#!/bin/bash
# bench.py with and without the clock-settle phase, three runs each (avg_launch_ms and value)
for s in 50 0 50 0 50 0; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming --no-power --settle-ms $s 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('settle_ms=$s  ms_per_step %.4f  avg_launch_ms %.4f  value %.4g  frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'], d['roofline']['frac']))"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-streaming --no-power 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('200 steps settled  ms_per_step %.4f  avg_launch_ms %.4f  value %.4g' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value']))"
