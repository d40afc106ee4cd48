#!/bin/bash
# same-box A/B of the working tree against a worktree of HEAD in ./_head (both pre-built): bench.py driver command, alternating
for i in 1 2 3; do
  for d in . _head; do
    (cd $d && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming --no-power 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$d', 'ms_per_step %.4f avg_launch_ms %.4f value %.4g' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value']))")
  done
done
for d in . _head; do (cd $d && python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-streaming --no-power 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$d', '200 steps: ms_per_step %.4f value %.4g' % (d['ms_per_step'], d['value']))"); done
