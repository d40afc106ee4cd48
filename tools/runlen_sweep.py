#!/usr/bin/env python3
"""Sweep SG_RUN_LEN (frames per wave-run) for the bench workload: the driver's 20-step command three times and one
200-step run per point.  usage: tools/runlen_sweep.py 12,14,16,...  [extra bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def bench(rl, steps, warm, extra):
    e = dict(os.environ)
    if rl:
        e["SG_RUN_LEN"] = str(rl)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warm),
                        "--no-cpu-baseline", "--no-streaming", "--no-power"] + extra, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)["roofline"]["avg_launch_ms"] * 1e3
    return float("nan")

if __name__ == "__main__":
    extra = sys.argv[2:]
    for rl in [int(x) for x in sys.argv[1].split(",")]:
        s = sorted(bench(rl, 20, 5, extra) for _ in range(3))
        print("run_len=%-3d  20-step avg us: %s   200-step: %.1f" % (rl, " ".join("%.1f" % x for x in s), bench(rl, 200, 20, extra)), flush=True)
