#!/usr/bin/env python3
"""Tuning sweeps on the GPU box: rebuild with extra hipcc flags and/or sweep SG_RUN_LEN,
run bench.py for each point, print one table.  Not part of the product."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(env, steps=100):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "10",
                        "--no-cpu-baseline", "--no-streaming", "--no-power"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            return j["value"], j["roofline"]["avg_launch_ms"], j["ms_per_step"]
    return None, r.stderr[-300:], None


def rebuild(flags):
    e = dict(os.environ)
    e["SG_EXTRA_HIPCC_FLAGS"] = flags
    subprocess.check_call([sys.executable, "-m", "spectrogram_amd.build", "--force"], cwd=ROOT, env=e,
                          stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    flag_sets = sys.argv[1].split("|") if len(sys.argv) > 1 else [""]
    run_lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    extra = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    for fl in flag_sets:
        rebuild(fl)
        for rl in run_lens:
            for xl in extra:
                env = {}
                if rl:
                    env["SG_RUN_LEN"] = str(rl)
                if xl:
                    env["SG_EXTRA_LDS"] = str(xl)
                v, a, m = bench(env)
                print("flags=%-22s run_len=%-3d extra_lds=%-6d frames/s=%s avg_launch_ms=%.6s ms_per_step=%.6s" %
                      (fl, rl, xl, v and "%.4g" % v, a, m), flush=True)
    rebuild("")
